// kernels/manifold.h -- manifold operations: stand-alone preconditioner / projection, qf retraction, polar projection, rounding to SE(d).
// Part of kernels.h (included inside namespace dpgo, in this order: common.h, problem.h, tcg.h, persist.h, multilevel.h, dense.h, manifold.h, rtr.h, agent.h, init.h).
#pragma once

// ================================================================ K6: preconditioner (stand-alone)
// Z = proj_X( V * Dinv )   (QuadraticProblem::PreConditioner, src/QuadraticProblem.cpp:56-69, with
// the block-Jacobi factor in place of the CHOLMOD solve); dinv == nullptr -> Z = proj_X(V).
template <int D, int R>
__global__ __launch_bounds__(kBlock) void k_precond(const double* __restrict__ X, const double* __restrict__ V,
                                                    const double* __restrict__ dinv, double* __restrict__ Z,
                                                    int n) {
  using GEO = Geo<D, R>;
  __shared__ double sm[kWaves][3][GEO::G][GEO::T];
  const LaneId L = lane_id<D>();
  const int ntiles = (n + GEO::P - 1) / GEO::P;
  const TileIter ti_ = tile_iter(ntiles);
  for (int tile = ti_.first; tile < ti_.last; tile += ti_.step) {
    const int i = tile * GEO::P + L.wave * GEO::G + L.g;
    const bool ok = (L.g < GEO::G) && (i < n);
    const size_t off = (size_t)i * GEO::T + L.c * R;
    double* ys = ok ? &sm[L.wave][0][L.g][0] : nullptr;
    double* vs = ok ? &sm[L.wave][1][L.g][0] : nullptr;
    double* zs = ok ? &sm[L.wave][2][L.g][0] : nullptr;
    double v[R], x[R], z[R];
    if (ok) {
      load_col<R>(X + off, x);
      load_col<R>(V + off, v);
      store_col<R>(ys + L.c * R, x);
      store_col<R>(vs + L.c * R, v);
    }
    wave_sync();
    if (ok) {
      if (dinv) {
        jacobi_col<D, R>(vs, dinv + (size_t)i * GEO::BB + L.c * GEO::B, z);
      } else {
#pragma unroll
        for (int a = 0; a < R; ++a) z[a] = v[a];
      }
      store_col<R>(zs + L.c * R, z);
    }
    wave_sync();
    if (ok) {
      double out[R], s[D];
      proj_col<D, R>(ys, zs, L.c, z, out, s);
      store_col<R>(Z + off, out);
    }
    wave_sync();
  }
}

// ================================================================ K4: retraction
// X2 = R_X(scale * eta): Stiefel factor = Q of the thin QR of Y + eta with diag(R) > 0 (modified
// Gram-Schmidt; ROPTLIB Stiefel::qfRetraction), Euclidean factor p + eta.  Each lane c < D rebuilds
// q_0..q_c from the LDS tile (identical arithmetic in all lanes of the pose).
template <int D, int R>
__global__ __launch_bounds__(kBlock) void k_retract(const double* __restrict__ X, const double* __restrict__ eta,
                                                    double scale, double* __restrict__ X2,
                                                    const DevState* __restrict__ st, int n) {
  using GEO = Geo<D, R>;
  __shared__ double sm[kWaves][GEO::G][GEO::T];
  if (st && st->rtr_stop) return;
  const LaneId L = lane_id<D>();
  const int ntiles = (n + GEO::P - 1) / GEO::P;
  const TileIter ti_ = tile_iter(ntiles);
  for (int tile = ti_.first; tile < ti_.last; tile += ti_.step) {
    const int i = tile * GEO::P + L.wave * GEO::G + L.g;
    const bool ok = (L.g < GEO::G) && (i < n);
    const size_t off = (size_t)i * GEO::T + L.c * R;
    double* as = ok ? &sm[L.wave][L.g][0] : nullptr;
    double a[R];
    if (ok) {
      double x[R], e[R];
      load_col<R>(X + off, x);
      load_col<R>(eta + off, e);
#pragma unroll
      for (int k = 0; k < R; ++k) a[k] = fma(scale, e[k], x[k]);
      store_col<R>(as + L.c * R, a);
    }
    wave_sync();
    if (ok) {
      qf_col<D, R>(as, L.c, a);
      store_col<R>(X2 + off, a);
    }
    wave_sync();
  }
}

// ================================================================ K5: polar projection
// LiftedSEManifold::project (src/manifold/LiftedSEManifold.cpp:34-45; JacobiSVD U V^T,
// src/DPGO_utils.cpp:480-486).  out = polar( a*A + b*Bm + c*Cm ) per pose when project != 0:
// U V^T = M (M^T M)^{-1/2}; the D x D symmetric eigenproblem is solved by cyclic Jacobi sweeps
// in registers.  One lane per pose column; every lane c < D of a pose repeats the small solve.
template <int D, int R>
__global__ __launch_bounds__(kBlock) void k_axpby_project(double a, const double* __restrict__ A, double b,
                                                          const double* __restrict__ Bm, double c,
                                                          const double* __restrict__ Cm, int project,
                                                          double* __restrict__ out, int n) {
  using GEO = Geo<D, R>;
  __shared__ double sm[kWaves][GEO::G][GEO::T];
  const LaneId L = lane_id<D>();
  const int ntiles = (n + GEO::P - 1) / GEO::P;
  const TileIter ti_ = tile_iter(ntiles);
  for (int tile = ti_.first; tile < ti_.last; tile += ti_.step) {
    const int i = tile * GEO::P + L.wave * GEO::G + L.g;
    const bool ok = (L.g < GEO::G) && (i < n);
    const size_t off = (size_t)i * GEO::T + L.c * R;
    double* ms = ok ? &sm[L.wave][L.g][0] : nullptr;
    double m[R];
    if (ok) {
#pragma unroll
      for (int k = 0; k < R; ++k) {
        double v = a * A[off + k];
        if (Bm) v = fma(b, Bm[off + k], v);
        if (Cm) v = fma(c, Cm[off + k], v);
        m[k] = v;
      }
      store_col<R>(ms + L.c * R, m);
    }
    wave_sync();
    if (ok) {
      if (project && L.c < D) {
        // C = M^T M (D x D), eigen-decompose C = W diag(lam) W^T, out col c = sum_a M[:,a] * F[a][c],
        // F = W diag(lam^-1/2) W^T.
        double Cmat[D][D], W[D][D];
#pragma unroll
        for (int p = 0; p < D; ++p)
#pragma unroll
          for (int q = 0; q < D; ++q) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < R; ++k) s = fma(ms[p * R + k], ms[q * R + k], s);
            Cmat[p][q] = s;
            W[p][q] = (p == q) ? 1.0 : 0.0;
          }
        for (int sweep = 0; sweep < 12; ++sweep) {
          double offn = 0.0;
#pragma unroll
          for (int p = 0; p < D; ++p)
#pragma unroll
            for (int q = p + 1; q < D; ++q) offn += Cmat[p][q] * Cmat[p][q];
          double dn = 0.0;
#pragma unroll
          for (int p = 0; p < D; ++p) dn += Cmat[p][p] * Cmat[p][p];
          if (offn <= 1e-32 * dn) break;
#pragma unroll
          for (int p = 0; p < D; ++p)
#pragma unroll
            for (int q = p + 1; q < D; ++q) {
              const double apq = Cmat[p][q];
              if (apq != 0.0) {
                const double th = (Cmat[q][q] - Cmat[p][p]) / (2.0 * apq);
                const double t = (th >= 0.0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
                const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
#pragma unroll
                for (int k = 0; k < D; ++k) {
                  const double ckp = Cmat[k][p], ckq = Cmat[k][q];
                  Cmat[k][p] = cs * ckp - sn * ckq;
                  Cmat[k][q] = sn * ckp + cs * ckq;
                }
#pragma unroll
                for (int k = 0; k < D; ++k) {
                  const double cpk = Cmat[p][k], cqk = Cmat[q][k];
                  Cmat[p][k] = cs * cpk - sn * cqk;
                  Cmat[q][k] = sn * cpk + cs * cqk;
                }
#pragma unroll
                for (int k = 0; k < D; ++k) {
                  const double wkp = W[k][p], wkq = W[k][q];
                  W[k][p] = cs * wkp - sn * wkq;
                  W[k][q] = sn * wkp + cs * wkq;
                }
              }
            }
        }
        // singular values are clamped at 1e-14 sigma_max: a rank-deficient block (which the reference's SVD maps to
        // SOME finite U V^T) stays finite here too instead of dividing by zero
        double lmax = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) lmax = fmax(lmax, Cmat[k][k]);
        const double lfloor = fmax(1e-28 * lmax, 1e-300);
        double F[D];  // column c of F
#pragma unroll
        for (int p = 0; p < D; ++p) {
          double s = 0.0;
#pragma unroll
          for (int k = 0; k < D; ++k) {
            double wck = 0.0;
#pragma unroll
            for (int cc = 0; cc < D; ++cc) wck = (cc == L.c) ? W[cc][k] : wck;
            s += W[p][k] * wck / sqrt(fmax(Cmat[k][k], lfloor));
          }
          F[p] = s;
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
          double s = 0.0;
#pragma unroll
          for (int p = 0; p < D; ++p) s = fma(ms[p * R + k], F[p], s);
          m[k] = s;
        }
      }
      store_col<R>(out + off, m);
    }
    wave_sync();
  }
}

// ================================================================ K12: rounding to SE(d)
// PGOAgent::getTrajectoryInLocalFrame / getTrajectoryInGlobalFrame (src/PGOAgent.cpp:718-767):
//   T_i = [ projectToRotationGroup(Ya^T Y_i) | Ya^T p_i - t0 ],  t0 = Ya^T pa,
// anchor (Ya, pa) = the global anchor, or pose 0 of X (local frame).  projectToRotationGroup
// (src/DPGO_utils.cpp:464-478: U V^T, last column of U negated when det U det V < 0) is evaluated as
// M V diag(s_k / sigma_k) V^T from the eigen-decomposition M^T M = V diag(sigma^2) V^T (cyclic Jacobi), with
// s_k = -1 on the SMALLEST singular value when det M < 0.  One lane per pose; output tiles [n][d+1][d]
// (= the reference's d x (d+1)n column-major Matrix).
struct AnchorArg {
  double v[4 * 6];  // (d+1) x r tile, same layout as a pose tile of X
  int use;          // 0: take pose 0 of X
};
template <int D, int R>
__global__ __launch_bounds__(kBlock) void k_round(const double* __restrict__ X, AnchorArg anchor,
                                                  double* __restrict__ T, int n) {
  constexpr int B = D + 1, TS = B * R;
  double Ya[D][R], pa[R];
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int k = 0; k < R; ++k) Ya[a][k] = anchor.use ? anchor.v[a * R + k] : X[a * R + k];
#pragma unroll
  for (int k = 0; k < R; ++k) pa[k] = anchor.use ? anchor.v[D * R + k] : X[D * R + k];
  double t0[D];
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < R; ++k) s = fma(Ya[a][k], pa[k], s);
    t0[a] = s;
  }
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const double* __restrict__ x = X + (size_t)i * TS;
    double M[D][D], tt[D];
#pragma unroll
    for (int a = 0; a < D; ++a) {
#pragma unroll
      for (int b = 0; b < D; ++b) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < R; ++k) s = fma(Ya[a][k], x[b * R + k], s);
        M[a][b] = s;
      }
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < R; ++k) s = fma(Ya[a][k], x[D * R + k], s);
      tt[a] = s - t0[a];
    }
    double det;
    if constexpr (D == 2) {
      det = M[0][0] * M[1][1] - M[0][1] * M[1][0];
    } else {
      det = M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
            M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
    }
    double C[D][D], W[D][D];
#pragma unroll
    for (int p = 0; p < D; ++p)
#pragma unroll
      for (int q = 0; q < D; ++q) {
        double s = 0.0;
#pragma unroll
        for (int a = 0; a < D; ++a) s = fma(M[a][p], M[a][q], s);
        C[p][q] = s;
        W[p][q] = (p == q) ? 1.0 : 0.0;
      }
    for (int sweep = 0; sweep < 16; ++sweep) {
      double offn = 0.0, dn = 0.0;
#pragma unroll
      for (int p = 0; p < D; ++p) {
        dn += C[p][p] * C[p][p];
#pragma unroll
        for (int q = p + 1; q < D; ++q) offn += C[p][q] * C[p][q];
      }
      if (offn <= 1e-32 * dn) break;
#pragma unroll
      for (int p = 0; p < D; ++p)
#pragma unroll
        for (int q = p + 1; q < D; ++q) {
          const double apq = C[p][q];
          if (apq != 0.0) {
            const double th = (C[q][q] - C[p][p]) / (2.0 * apq);
            const double t = (th >= 0.0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
            const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
#pragma unroll
            for (int k = 0; k < D; ++k) {
              const double ckp = C[k][p], ckq = C[k][q];
              C[k][p] = cs * ckp - sn * ckq;
              C[k][q] = sn * ckp + cs * ckq;
            }
#pragma unroll
            for (int k = 0; k < D; ++k) {
              const double cpk = C[p][k], cqk = C[q][k];
              C[p][k] = cs * cpk - sn * cqk;
              C[q][k] = sn * cpk + cs * cqk;
            }
#pragma unroll
            for (int k = 0; k < D; ++k) {
              const double wkp = W[k][p], wkq = W[k][q];
              W[k][p] = cs * wkp - sn * wkq;
              W[k][q] = sn * wkp + cs * wkq;
            }
          }
        }
    }
    int kmin = 0;
#pragma unroll
    for (int k = 1; k < D; ++k) kmin = (C[k][k] < C[kmin][kmin]) ? k : kmin;
    double sc[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const double lam = C[k][k] > 0.0 ? C[k][k] : 0.0;
      const double inv = lam > 0.0 ? 1.0 / sqrt(lam) : 0.0;
      sc[k] = (det < 0.0 && k == kmin) ? -inv : inv;
    }
    // F = W diag(sc) W^T ; Rot = M F
    double F[D][D];
#pragma unroll
    for (int p = 0; p < D; ++p)
#pragma unroll
      for (int q = 0; q < D; ++q) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) s = fma(W[p][k] * sc[k], W[q][k], s);
        F[p][q] = s;
      }
    double* __restrict__ o = T + (size_t)i * B * D;
#pragma unroll
    for (int c = 0; c < D; ++c)
#pragma unroll
      for (int a = 0; a < D; ++a) {
        double s = 0.0;
#pragma unroll
        for (int p = 0; p < D; ++p) s = fma(M[a][p], F[p][c], s);
        o[c * D + a] = s;
      }
#pragma unroll
    for (int a = 0; a < D; ++a) o[D * D + a] = tt[a];
  }
}
