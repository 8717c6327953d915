// kernels/multilevel.h -- opt-in two-level (aggregation multigrid) preconditioner.
// Part of kernels.h (included inside namespace dpgo, in this order: common.h, problem.h, tcg.h, tcg_pipe.h, multilevel.h, manifold.h, rtr.h, agent.h).
#pragma once

// ================================================================ two-level (aggregation multigrid) preconditioner
// Optional replacement of the block-Jacobi solve inside QuadraticProblem::PreConditioner (the reference applies an
// exact CHOLMOD solve of Q + 0.1 I there, src/QuadraticProblem.cpp:56-69).  One cycle for A = Q + shift I:
//   x1 = w Dinv r;  rc = P^T (r - A x1);  xc = Ac^-1 rc;  x = x1 + P xc;  z = proj_X( x + w Dinv (r - A x) )
// Aggregates are runs of k consecutive poses; P's blocks are relative poses composed along the odometry chain
// (host setup, oracle: amg_prolongation_blocks); Ac = P^T A P is kept as a dense inverse in HBM (<= 3200 unknowns,
// Infinity-Cache resident).  The four products with A, P, P^T run on the block-SpMM kernel (k_spmm with -A and the
// rectangular P / P^T); the kernels below are the three pieces that are not an SpMM.  `gate`: the solver's state
// record -- launches enqueued after tCG finished return at once.
template <int D, int R>
__global__ __launch_bounds__(kBlock) void k_ml_presmooth(const double* __restrict__ V, const double* __restrict__ dinv,
                                                         double omega, double* __restrict__ OUT,
                                                         const DevState* __restrict__ gate, int n) {
  using GEO = Geo<D, R>;
  if (gate && (gate->tcg_done || gate->rtr_stop)) return;
  __shared__ double sm[kWaves][GEO::G][GEO::T];
  const LaneId L = lane_id<D>();
  const int ntiles = (n + GEO::P - 1) / GEO::P;
  const TileIter ti_ = tile_iter(ntiles);
  for (int tile = ti_.first; tile < ti_.last; tile += ti_.step) {
    const int i = tile * GEO::P + L.wave * GEO::G + L.g;
    const bool ok = (L.g < GEO::G) && (i < n);
    const size_t off = (size_t)i * GEO::T + L.c * R;
    double* vs = ok ? &sm[L.wave][L.g][0] : nullptr;
    double v[R], z[R];
    if (ok) {
      load_col<R>(V + off, v);
      store_col<R>(vs + L.c * R, v);
    }
    wave_sync();
    if (ok) {
      jacobi_col<D, R>(vs, dinv + (size_t)i * GEO::BB + L.c * GEO::B, z);
#pragma unroll
      for (int a = 0; a < R; ++a) z[a] *= omega;
      store_col<R>(OUT + off, z);
    }
    wave_sync();
  }
}

// z = proj_X( x + w Dinv res ),  partial <z, r> into slot 1 of the update kernel's partial-sum region (the launch
// uses the update kernel's grid, so every workgroup entry is rewritten).
template <int D, int R>
__global__ __launch_bounds__(kBlock) void k_ml_finish(const double* __restrict__ X, const double* __restrict__ xv,
                                                      const double* __restrict__ res, const double* __restrict__ r,
                                                      const double* __restrict__ dinv, double omega,
                                                      double* __restrict__ Z, double* __restrict__ pout,
                                                      const DevState* __restrict__ gate, int n) {
  using GEO = Geo<D, R>;
  if (gate && (gate->tcg_done || gate->rtr_stop)) return;
  __shared__ double sm[kWaves][3][GEO::G][GEO::T];
  __shared__ double red[kWaves * kNP];
  const LaneId L = lane_id<D>();
  const int ntiles = (n + GEO::P - 1) / GEO::P;
  const TileIter ti_ = tile_iter(ntiles);
  double part[1] = {0.0};
  for (int tile = ti_.first; tile < ti_.last; tile += ti_.step) {
    const int i = tile * GEO::P + L.wave * GEO::G + L.g;
    const bool ok = (L.g < GEO::G) && (i < n);
    const size_t off = (size_t)i * GEO::T + L.c * R;
    double* ys = ok ? &sm[L.wave][0][L.g][0] : nullptr;
    double* vs = ok ? &sm[L.wave][1][L.g][0] : nullptr;
    double* zs = ok ? &sm[L.wave][2][L.g][0] : nullptr;
    double x[R], v[R], z[R];
    if (ok) {
      load_col<R>(X + off, x);
      load_col<R>(res + off, v);
      store_col<R>(ys + L.c * R, x);
      store_col<R>(vs + L.c * R, v);
    }
    wave_sync();
    if (ok) {
      double xc[R];
      load_col<R>(xv + off, xc);
      jacobi_col<D, R>(vs, dinv + (size_t)i * GEO::BB + L.c * GEO::B, z);
#pragma unroll
      for (int a = 0; a < R; ++a) z[a] = fma(omega, z[a], xc[a]);
      store_col<R>(zs + L.c * R, z);
    }
    wave_sync();
    if (ok) {
      double out[R], s[D], rr[R];
      proj_col<D, R>(ys, zs, L.c, z, out, s);
      store_col<R>(Z + off, out);
      load_col<R>(r + off, rr);
#pragma unroll
      for (int a = 0; a < R; ++a) part[0] = fma(out[a], rr[a], part[0]);
    }
    wave_sync();
  }
  block_allreduce<1>(part, red);
  if (threadIdx.x == 0 && pout) pout[blockIdx.x * kNP + 1] = part[0];
}

// Dense coarse solve: OUT (N x R, R contiguous) = M (N x N, row-major) * V (N x R).  One wave per output row;
// M streams once (Infinity-Cache / HBM), V is re-read by every wave through L2.  M is STORED in fp32 (it is a
// preconditioner: iteration counts are unchanged, the dominant stream of the cycle halves); accumulation is fp64.
template <int R>
__global__ __launch_bounds__(kBlock) void k_ml_dense_apply(const float* __restrict__ M, const double* __restrict__ V,
                                                           double* __restrict__ OUT, const DevState* __restrict__ gate,
                                                           int N) {
  if (gate && (gate->tcg_done || gate->rtr_stop)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int row = blockIdx.x * kWaves + wave; row < N; row += gridDim.x * kWaves) {
    const float* __restrict__ m = M + (size_t)row * N;
    double acc[R];
#pragma unroll
    for (int a = 0; a < R; ++a) acc[a] = 0.0;
    for (int j = lane; j < N; j += 64) {
      const double mv = (double)m[j];
#pragma unroll
      for (int a = 0; a < R; ++a) acc[a] = fma(mv, V[(size_t)j * R + a], acc[a]);
    }
#pragma unroll
    for (int a = 0; a < R; ++a) acc[a] = wave_reduce_lane63(acc[a]);
    if (lane == 63) {
#pragma unroll
      for (int a = 0; a < R; ++a) OUT[(size_t)row * R + a] = acc[a];
    }
  }
}

// vals_out = -(Q + shift I) on Q's pattern (the SpMM kernel then yields r - A v in one pass: OUT = v (-A) + r)
template <int D>
__global__ __launch_bounds__(kBlock) void k_ml_neg_shift(BsrDev Q, double shift, double* __restrict__ vals_out, int n) {
  constexpr int B = D + 1, BB = B * B;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    for (int t = Q.rowptr[i]; t < Q.rowptr[i + 1]; ++t) {
      const bool diag = Q.colidx[t] == i;
#pragma unroll
      for (int e = 0; e < BB; ++e) {
        double v = -Q.vals[(size_t)t * BB + e];
        if (diag && (e / B) == (e % B)) v -= shift;
        vals_out[(size_t)t * BB + e] = v;
      }
    }
  }
}

// ---------------------------------------------------------------- fused form of the cycle (5 launches per tCG iteration)
// k_tcg_update(ml_omega) writes x1 = w Dinv r;  k_ml_restrict: rc = P^T (r - A x1) in one pass (needs aggregates that
// do not straddle workgroup tiles: P % k == 0);  k_ml_coarse_prolong: xc = Ac^-1 rc and x = x1 + P xc, one workgroup
// per aggregate;  k_ml_post: z = proj_X(x + w Dinv (r - A x)) in the SpMM's epilogue, with the partial sums <r,r>, <z,r>
// for the next k_tcg_hess (slots 0 and 1 of every entry of ITS grid).
template <int D, int R, int SPLIT>
__global__ __launch_bounds__(kBlock) void k_ml_restrict(BsrDev Q, const double* __restrict__ x1,
                                                        const double* __restrict__ r, const double* __restrict__ Pb,
                                                        double shift, int k, double* __restrict__ rc,
                                                        const DevState* __restrict__ gate, int n) {
  using GEO = Geo<D, R, SPLIT>;
  if (gate && (gate->tcg_done || gate->rtr_stop)) return;
  __shared__ double res_s[kWaves][GEO::G][GEO::T];  // residual tiles (per wave)
  __shared__ double t_s[GEO::P][GEO::T];            // P_i^T res_i of every pose of the workgroup tile
  const LaneId L = lane_id<D, SPLIT>();
  const int ntiles = (n + GEO::P - 1) / GEO::P;
  const TileIter ti_ = tile_iter(ntiles);
  for (int tile = ti_.first; tile < ti_.last; tile += ti_.step) {
    const int lp = L.wave * GEO::G + L.g;  // pose slot inside the workgroup tile
    const int i = tile * GEO::P + lp;
    const bool okp = (L.g < GEO::G) && (i < n);
    const bool ok = okp && (L.s == 0);
    const size_t off = (size_t)i * GEO::T + L.c * R;
    double h[R];
    spmm_col<D, R, SPLIT>(Q.rowptr, Q.colidx, Q.vals, x1, i, L.s, L.c, okp, h);
    if (ok) {
      double xr[R], rr[R];
      load_col<R>(x1 + off, xr);
      load_col<R>(r + off, rr);
#pragma unroll
      for (int a = 0; a < R; ++a) h[a] = rr[a] - h[a] - shift * xr[a];
      store_col<R>(&res_s[L.wave][L.g][L.c * R], h);
    }
    wave_sync();
    if (L.s == 0 && L.g < GEO::G) {
      double t[R];
#pragma unroll
      for (int a = 0; a < R; ++a) t[a] = 0.0;
      if (ok) {  // row c of P_i^T res_i = sum_c' P_i[c'][c] res_i[c'][:]
        const double* __restrict__ pb = Pb + (size_t)i * GEO::BB;
#pragma unroll
        for (int cc = 0; cc < GEO::B; ++cc) {
          const double pv = pb[cc * GEO::B + L.c];
#pragma unroll
          for (int a = 0; a < R; ++a) t[a] = fma(pv, res_s[L.wave][L.g][cc * R + a], t[a]);
        }
      }
      store_col<R>(&t_s[lp][L.c * R], t);  // zeros for poses beyond n
    }
    __syncthreads();
    if (ok && (i % k) == 0) {  // the aggregate's first pose sums its members (all inside this tile)
      double acc[R];
#pragma unroll
      for (int a = 0; a < R; ++a) acc[a] = 0.0;
      for (int m = 0; m < k && lp + m < GEO::P; ++m) {
#pragma unroll
        for (int a = 0; a < R; ++a) acc[a] += t_s[lp + m][L.c * R + a];
      }
      store_col<R>(rc + (size_t)(i / k) * GEO::T + L.c * R, acc);
    }
    __syncthreads();
  }
}

// One workgroup per aggregate a: waves 0..B-1 compute the B rows of xc_a = (Ac^-1 rc)_a, then the workgroup writes
// x_i = x1_i + P_i xc_a for the aggregate's poses.
template <int D, int R>
__global__ __launch_bounds__(kBlock) void k_ml_coarse_prolong(const float* __restrict__ M, const double* __restrict__ rc,
                                                              const double* __restrict__ x1,
                                                              const double* __restrict__ Pb, int k,
                                                              double* __restrict__ x, const DevState* __restrict__ gate,
                                                              int n, int nc) {
  constexpr int B = D + 1, T = B * R, BB = B * B;
  if (gate && (gate->tcg_done || gate->rtr_stop)) return;
  __shared__ double xc_s[B][R];
  __shared__ double part_s[kWaves][B][R];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int N = nc * B;
  for (int a = blockIdx.x; a < nc; a += gridDim.x) {
    // every wave takes a quarter of the columns and ALL B rows of this aggregate: one read of rc per workgroup
    // (rc is 10x the bytes of a matrix row: read per row it dominated the L2 traffic of the cycle)
    {
      const float* __restrict__ m = M + (size_t)(a * B) * N;
      double acc[B][R];
#pragma unroll
      for (int c = 0; c < B; ++c)
#pragma unroll
        for (int q = 0; q < R; ++q) acc[c][q] = 0.0;
      for (int j = wave * 64 + lane; j < N; j += kBlock) {
        double rv[R];
#pragma unroll
        for (int q = 0; q < R; ++q) rv[q] = rc[(size_t)j * R + q];
#pragma unroll
        for (int c = 0; c < B; ++c) {
          const double mv = (double)m[(size_t)c * N + j];
#pragma unroll
          for (int q = 0; q < R; ++q) acc[c][q] = fma(mv, rv[q], acc[c][q]);
        }
      }
#pragma unroll
      for (int c = 0; c < B; ++c)
#pragma unroll
        for (int q = 0; q < R; ++q) {
          const double sum = wave_reduce_lane63(acc[c][q]);
          if (lane == 63) part_s[wave][c][q] = sum;
        }
    }
    __syncthreads();
    if (threadIdx.x < B * R) {  // fixed-order sum over the waves
      const int c = threadIdx.x / R, q = threadIdx.x % R;
      double sum = part_s[0][c][q];
#pragma unroll
      for (int w2 = 1; w2 < kWaves; ++w2) sum += part_s[w2][c][q];
      xc_s[c][q] = sum;
    }
    __syncthreads();
    for (int tsk = threadIdx.x; tsk < k * B; tsk += kBlock) {  // (pose, row c) tasks of the aggregate
      const int i = a * k + tsk / B, c = tsk % B;
      if (i < n) {
        const double* __restrict__ pb = Pb + (size_t)i * BB + c * B;
        const size_t off = (size_t)i * T + c * R;
#pragma unroll
        for (int q = 0; q < R; ++q) {
          double v = x1[off + q];
#pragma unroll
          for (int cc = 0; cc < B; ++cc) v = fma(pb[cc], xc_s[cc][q], v);
          x[off + q] = v;
        }
      }
    }
    __syncthreads();
  }
}

template <int D, int R, int SPLIT>
__global__ __launch_bounds__(kBlock) void k_ml_post(BsrDev Q, const double* __restrict__ X,
                                                    const double* __restrict__ xv, const double* __restrict__ r,
                                                    const double* __restrict__ dinv, double omega, double shift,
                                                    double* __restrict__ Z, double* __restrict__ pout,
                                                    const DevState* __restrict__ gate, int n) {
  using GEO = Geo<D, R, SPLIT>;
  if (gate && (gate->tcg_done || gate->rtr_stop)) return;
  __shared__ double sm[kWaves][3][GEO::G][GEO::T];
  __shared__ double red[kWaves * kNP];
  const LaneId L = lane_id<D, SPLIT>();
  const int ntiles = (n + GEO::P - 1) / GEO::P;
  const TileIter ti_ = tile_iter(ntiles);
  double part[2] = {0.0, 0.0};
  for (int tile = ti_.first; tile < ti_.last; tile += ti_.step) {
    const int i = tile * GEO::P + L.wave * GEO::G + L.g;
    const bool okp = (L.g < GEO::G) && (i < n);
    const bool ok = okp && (L.s == 0);
    const size_t off = (size_t)i * GEO::T + L.c * R;
    double* ys = ok ? &sm[L.wave][0][L.g][0] : nullptr;
    double* vs = ok ? &sm[L.wave][1][L.g][0] : nullptr;
    double* zs = ok ? &sm[L.wave][2][L.g][0] : nullptr;
    double h[R], xr[R], rr[R], z[R];
    spmm_col<D, R, SPLIT>(Q.rowptr, Q.colidx, Q.vals, xv, i, L.s, L.c, okp, h);
    if (ok) {
      double x[R];
      load_col<R>(X + off, x);
      load_col<R>(xv + off, xr);
      load_col<R>(r + off, rr);
#pragma unroll
      for (int a = 0; a < R; ++a) {
        h[a] = rr[a] - h[a] - shift * xr[a];  // r - A x
        part[0] = fma(rr[a], rr[a], part[0]);
      }
      store_col<R>(ys + L.c * R, x);
      store_col<R>(vs + L.c * R, h);
    }
    wave_sync();
    if (ok) {
      jacobi_col<D, R>(vs, dinv + (size_t)i * GEO::BB + L.c * GEO::B, z);
#pragma unroll
      for (int a = 0; a < R; ++a) z[a] = fma(omega, z[a], xr[a]);
      store_col<R>(zs + L.c * R, z);
    }
    wave_sync();
    if (ok) {
      double out[R], s[D];
      proj_col<D, R>(ys, zs, L.c, z, out, s);
      store_col<R>(Z + off, out);
#pragma unroll
      for (int a = 0; a < R; ++a) part[1] = fma(out[a], rr[a], part[1]);
    }
    wave_sync();
  }
  store_partials<2>(part, pout, red);
}
