// kernels/problem.h -- QuadraticProblem evaluations: plain SpMM, cost + Riemannian gradient, Riemannian Hessian-vector product.
// Part of kernels.h (included inside namespace dpgo, in this order: common.h, problem.h, tcg.h, persist.h, multilevel.h, dense.h, manifold.h, rtr.h, agent.h, init.h).
#pragma once

// ================================================================ K1: plain SpMM
// OUT = V*Q (+ Gadd).  QuadraticProblem::EucGrad / EucHessianEta
// (src/QuadraticProblem.cpp:43-54) and, with a rectangular coupling matrix, PoseGraph::constructG
// (src/PoseGraph.cpp:493-580).
template <int D, int R, int SPLIT, int NTS = 0>
__global__ __launch_bounds__(kBlock) void k_spmm(BsrDev Q, const double* __restrict__ V,
                                                 const double* __restrict__ Gadd, double* __restrict__ OUT,
                                                 int n) {
  using GEO = Geo<D, R, SPLIT>;
  const LaneId L = lane_id<D, SPLIT>();
  const int ntiles = (n + GEO::P - 1) / GEO::P;
  const TileIter ti_ = tile_iter(ntiles);
  for (int tile = ti_.first; tile < ti_.last; tile += ti_.step) {
    const int i = tile * GEO::P + L.wave * GEO::G + L.g;
    const bool okp = (L.g < GEO::G) && (i < n);
    const bool ok = okp && (L.s == 0);
    double acc[R];
    spmm_col<D, R, SPLIT>(Q.rowptr, Q.colidx, Q.vals, V, i, L.s, L.c, okp, acc);
    if (ok) {
      const size_t off = (size_t)i * GEO::T + L.c * R;
      if (Gadd) {
#pragma unroll
        for (int a = 0; a < R; ++a) acc[a] += ld_stream<NTS>(Gadd + off + a);
      }
      store_col_stream<NTS, R>(OUT + off, acc);
    }
  }
}

// ================================================================ K1 on symmetric storage (common.h, spmm_sym_pre)
// DPGO_SPMM_SYM_BATCH: blocks whose loads the gather keeps in flight (as DPGO_HESS_BATCH for the tCG-step kernel);
// DPGO_SPMM_SYM_SPAN: the product leaves through the wave's LDS tile as lane-linear 16-byte pieces (even tile sizes) instead
// of five 8-byte stores per lane at a 40-byte stride -- non-temporal 8-byte stores reach HBM as partial lines (PMC: 20.6 MB
// written for 16 MB of output).
#ifndef DPGO_SPMM_SYM_BATCH
#define DPGO_SPMM_SYM_BATCH 4
#endif
#ifndef DPGO_SPMM_SYM_SPAN
#define DPGO_SPMM_SYM_SPAN 1
#endif
template <int D, int R, int NTS>
__global__ __launch_bounds__(kBlock) void k_spmm_sym(BsrSymDev Q, const double* __restrict__ V,
                                                     const double* __restrict__ Gadd, double* __restrict__ OUT, int n) {
  using GEO = Geo<D, R, 1>;
  constexpr bool kSpanOut = Span<D, R, 1>::kOk && DPGO_SPMM_SYM_SPAN;
  __shared__ __attribute__((aligned(16))) double os[kSpanOut ? kWaves : 1][kSpanOut ? GEO::G : 1][kSpanOut ? GEO::T : 1];
  const LaneId L = lane_id<D, 1>();
  const int ntiles = (n + GEO::P - 1) / GEO::P;
  const TileIter ti_ = tile_iter(ntiles);
  for (int tk = ti_.first; tk < ti_.last; tk += ti_.step) {
    const int tile = tile_of(Q, tk);
    const int i = tile * GEO::P + L.wave * GEO::G + L.g;
    const bool ok = (L.g < GEO::G) && (i < n);
    const SymIdx si = sym_idx_load<D>(Q, i, L.c, ok);
    double out[R];
    spmm_sym_pre<D, R, DPGO_SPMM_SYM_BATCH>(si, Q, V, L.c, out);
    const size_t off = (size_t)i * GEO::T + L.c * R;
    if (ok && Gadd) {
#pragma unroll
      for (int a = 0; a < R; ++a) out[a] += ld_stream<NTS>(Gadd + off + a);
    }
    if constexpr (kSpanOut) {
      const int p0 = tile * GEO::P + L.wave * GEO::G;
      const int npose = (n - p0) < GEO::G ? (n - p0) : GEO::G;
      const int valid = npose > 0 ? npose * GEO::T : 0;
      if (ok) store_col<R>(&os[L.wave][L.g][L.c * R], out);
      wave_sync();
      span_from_lds<D, R, NTS>(OUT + (size_t)p0 * GEO::T, &os[L.wave][0][0], valid);
      wave_sync();  // (the tile is rewritten by the wave's next row block)
    } else if (ok) {
      store_col_stream<NTS, R>(OUT + off, out);
    }
  }
}

// refresh of the transposed upper copy after Q's values changed: uvalsT[u][p][q] = vals[src[u]][q][p]
template <int D>
__global__ __launch_bounds__(kBlock) void k_sym_refresh(const double* __restrict__ vals, const int32_t* __restrict__ src,
                                                        double* __restrict__ uvalsT, int nu) {
  constexpr int B = D + 1, BB = B * B;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < (size_t)nu * BB; e += (size_t)gridDim.x * kBlock) {
    const size_t u = e / BB;
    const int p = (int)(e % BB) / B, q = (int)(e % BB) % B;
    uvalsT[e] = vals[(size_t)src[u] * BB + q * B + p];
  }
}

// lower block l (slot lsrc[l] of Q) must be the transpose of its upper block: Q[i,j][p][q] = Q[j,i][q][p] = uvalsT[u][p][q]
template <int D>
__global__ __launch_bounds__(kBlock) void k_sym_check(const double* __restrict__ vals, const int32_t* __restrict__ lsrc,
                                                      const int32_t* __restrict__ lslot, const double* __restrict__ uvalsT,
                                                      int nl, int* __restrict__ flag) {
  constexpr int BB = (D + 1) * (D + 1);
  bool bad = false;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < (size_t)nl * BB; e += (size_t)gridDim.x * kBlock) {
    const size_t l = e / BB;
    const int k = (int)(e % BB);
    const double a = vals[(size_t)lsrc[l] * BB + k], b = uvalsT[(size_t)lslot[l] * BB + k];
    if (!(fabs(a - b) <= 1e-12 * (fabs(a) + fabs(b)))) bad = true;
  }
  if (bad) atomicOr(flag, 1);
}

// ================================================================ K1+K2: cost + Riemannian gradient
// One pass over Q gives f(X) = 0.5<XQ,X> + <X,G> (src/QuadraticProblem.cpp:29-41),
// EG = XQ + G (:43-47), S = sym(Y^T EG_rot) (cached for the Hessian, ROPTLIB EucGradToGrad),
// RG = proj_X(EG) (:71-79) and |RG|^2 (:81-83).
// partials: [0] sum(XQ.X)  [1] sum(X.G)  [2] |RG|^2
template <int D, int R, int SPLIT, class MAT = BsrDev>
__global__ __launch_bounds__(kBlock) void k_grad(MAT Q, const double* __restrict__ X,
                                                 const double* __restrict__ Gm, double* __restrict__ RG,
                                                 double* __restrict__ S, double* __restrict__ EGout,
                                                 double* __restrict__ partials, const DevState* __restrict__ st,
                                                 int n) {
  using GEO = Geo<D, R, SPLIT>;
  __shared__ double sm[kWaves][2][GEO::G][GEO::T];
  __shared__ double red[kWaves * kNP];
  if (st && st->rtr_stop) return;
  const LaneId L = lane_id<D, SPLIT>();
  const int ntiles = (n + GEO::P - 1) / GEO::P;
  double part[3] = {0.0, 0.0, 0.0};
  const TileIter ti_ = tile_iter(ntiles);
  for (int tk = ti_.first; tk < ti_.last; tk += ti_.step) {
    const int tile = tile_of(Q, tk);
    const int i = tile * GEO::P + L.wave * GEO::G + L.g;
    const bool okp = (L.g < GEO::G) && (i < n);
    const bool ok = okp && (L.s == 0);
    double eg[R], x[R];
    const size_t off = (size_t)i * GEO::T + L.c * R;
    double* ys = ok ? &sm[L.wave][0][L.g][0] : nullptr;
    double* ws = ok ? &sm[L.wave][1][L.g][0] : nullptr;
    q_gather<D, R, SPLIT>(Q, X, i, L.s, L.c, okp, eg);  // (plain or symmetric storage of Q, common.h)
    if (ok) {
      load_col<R>(X + off, x);
#pragma unroll
      for (int a = 0; a < R; ++a) part[0] = fma(eg[a], x[a], part[0]);
      if (Gm) {
#pragma unroll
        for (int a = 0; a < R; ++a) {
          const double gv = Gm[off + a];
          part[1] = fma(x[a], gv, part[1]);
          eg[a] += gv;
        }
      }
      store_col<R>(ys + L.c * R, x);
      store_col<R>(ws + L.c * R, eg);
    }
    wave_sync();
    if (ok) {
      double out[R], s[D];
      proj_col<D, R>(ys, ws, L.c, eg, out, s);
#pragma unroll
      for (int a = 0; a < R; ++a) part[2] = fma(out[a], out[a], part[2]);
      if (RG) store_col<R>(RG + off, out);
      if (EGout) store_col<R>(EGout + off, eg);
      if (S && L.c < D) {
#pragma unroll
        for (int a = 0; a < D; ++a) S[(size_t)i * D * D + L.c * D + a] = s[a];
      }
    }
    wave_sync();
  }
  store_partials<3>(part, partials, red);
}

// ================================================================ K1+K3+K2: Riemannian Hessian-vector product
// HV = proj_X( V*Q - V_rot * S ),  S = sym(Y^T EG_rot)   (QuadraticProblem::EucHessianEta,
// src/QuadraticProblem.cpp:49-54, + ROPTLIB Stiefel::EucHvToHv + ProductManifold::Projection).
// partials: [0] <V,HV>   [1] <V,Gdot> (if Gdot != null; used for the RTR model decrease)
// When `st` is given the kernel is a tCG step and exits early once tCG has finished.
template <int D, int R, int SPLIT, class MAT = BsrDev>
__global__ __launch_bounds__(kBlock) void k_hess(MAT Q, const double* __restrict__ X,
                                                 const double* __restrict__ S, const double* __restrict__ V,
                                                 const double* __restrict__ Gdot, double* __restrict__ HV,
                                                 double* __restrict__ partials, const DevState* __restrict__ st,
                                                 int check_tcg, int n) {
  using GEO = Geo<D, R, SPLIT>;
  __shared__ double sm[kWaves][3][GEO::G][GEO::T];
  __shared__ double red[kWaves * kNP];
  if (st) {
    if (st->rtr_stop) return;
    if (check_tcg && st->tcg_done) return;
  }
  const LaneId L = lane_id<D, SPLIT>();
  const int ntiles = (n + GEO::P - 1) / GEO::P;
  double part[2] = {0.0, 0.0};
  const TileIter ti_ = tile_iter(ntiles);
  for (int tk = ti_.first; tk < ti_.last; tk += ti_.step) {
    const int tile = tile_of(Q, tk);
    const int i = tile * GEO::P + L.wave * GEO::G + L.g;
    const bool okp = (L.g < GEO::G) && (i < n);
    const bool ok = okp && (L.s == 0);
    double h[R], v[R], x[R];
    const size_t off = (size_t)i * GEO::T + L.c * R;
    double* ys = ok ? &sm[L.wave][0][L.g][0] : nullptr;
    double* vs = ok ? &sm[L.wave][1][L.g][0] : nullptr;
    double* hs = ok ? &sm[L.wave][2][L.g][0] : nullptr;
    q_gather<D, R, SPLIT>(Q, V, i, L.s, L.c, okp, h);
    if (ok) {
      load_col<R>(X + off, x);
      load_col<R>(V + off, v);
      store_col<R>(ys + L.c * R, x);
      store_col<R>(vs + L.c * R, v);
    }
    wave_sync();
    if (ok) {
      if (L.c < D) {
        // h[:,c] -= sum_a V[:,a] * S[a][c]   (S symmetric: row c of S_i)
#pragma unroll
        for (int a = 0; a < D; ++a) {
          const double sac = S[(size_t)i * D * D + L.c * D + a];
#pragma unroll
          for (int k = 0; k < R; ++k) h[k] = fma(-vs[a * R + k], sac, h[k]);
        }
      }
      store_col<R>(hs + L.c * R, h);
    }
    wave_sync();
    if (ok) {
      double out[R], s[D];
      proj_col<D, R>(ys, hs, L.c, h, out, s);
#pragma unroll
      for (int a = 0; a < R; ++a) part[0] = fma(v[a], out[a], part[0]);
      if (Gdot) {
#pragma unroll
        for (int a = 0; a < R; ++a) part[1] = fma(v[a], Gdot[off + a], part[1]);
      }
      store_col<R>(HV + off, out);
    }
    wave_sync();
  }
  store_partials<2>(part, partials, red);
}
