// kernels/tcg_pipe.h -- opt-in pipelined tCG step (one launch and one reduction per iteration, small blocks).
// Part of kernels.h (included inside namespace dpgo, in this order: common.h, problem.h, tcg.h, tcg_pipe.h, multilevel.h, manifold.h, rtr.h, agent.h).
#pragma once

// ================================================================ pipelined tCG step (small, latency-bound blocks)
// ONE launch and ONE reduction per tCG iteration (oracle: tcg_pipelined; Ghysels & Vanroose's pipelined PCG
// mapped onto ROPTLIB's tCG_TR bookkeeping).  Beside r, z = P r, delta, H delta the kernel keeps w = H z,
// m = P w, q = P H delta, t = H q; the only operator application of iteration j is n = H m on a vector that the
// PREVIOUS launch completed, so no grid-wide dependency sits inside the launch:
//   prologue : <r,r>, <z,r>, <z,w> of the previous launch -> stop test, beta, <delta,H delta> = mu - beta^2 (..)_prev,
//              alpha, trust-region boundary / negative curvature
//   per tile : n = H m (gather);  delta = -z + b delta, H delta = -w + b H delta, q = -m + b q, t = -n + b t;
//              eta += a delta, r += a H delta, z += a q, w += a t;  m = P w;  partial sums of the new r, z, w
// mode 1 (init, after k_tcg_update(first)): w = H z, m = P w, partial <z,w>.  m is double-buffered: the gather of
// this launch reads m while other workgroups already write the next one.
// A launch of the two-kernel scheme costs ~8 us on a 2500-pose block whatever it computes, so halving the
// launches nearly halves the iteration; the price is 21 instead of 14 vector streams per iteration, which is
// why blocks in the bandwidth regime (SPLIT = 1) keep the two-kernel scheme.
template <int D, int R, int SPLIT>
__global__ __launch_bounds__(kBlock) void k_tcg_pipe(BsrDev Q, const double* __restrict__ X,
                                                     const double* __restrict__ S, const double* __restrict__ dinv,
                                                     const double* __restrict__ m, double* __restrict__ m_out,
                                                     double* __restrict__ z,
                                                     double* __restrict__ w, double* __restrict__ delta,
                                                     double* __restrict__ Hd, double* __restrict__ q,
                                                     double* __restrict__ t, double* __restrict__ eta,
                                                     double* __restrict__ r, const double* __restrict__ pin, int nb_in,
                                                     double* __restrict__ pout, const DevState* __restrict__ sin,
                                                     DevState* __restrict__ sout, int step_kind, int n,
                                                     unsigned long long* hflag, unsigned gen) {
  // step_kind: 0 = iteration j >= 1, 1 = init (w0, m0), 2 = iteration 0
  const int init = (step_kind == 1);
  const bool first = (step_kind == 2);
  using GEO = Geo<D, R, SPLIT>;
  using SPN = Span<D, R, SPLIT>;
  static_assert(SPN::kOk, "span layout needs an even tile size");
  __shared__ __attribute__((aligned(16))) double sm[kWaves][4][GEO::G][GEO::T];
  __shared__ double red[kWaves * kNP];
  const LaneId L = lane_id<D, SPLIT>();
  const int lane = threadIdx.x & 63;
  const int ntiles = (n + GEO::P - 1) / GEO::P;
  const TileIter ti_ = tile_iter(ntiles);
  double* ys = &sm[L.wave][0][0][0];
  double* vs = &sm[L.wave][1][0][0];
  double* hs = &sm[L.wave][2][0][0];
  double* os = &sm[L.wave][3][0][0];

  // ---- requests that do not depend on each other go out first (see k_tcg_hess_span)
  DevState st;
  load_state(st, sin);
  PartialRaw<3> praw;
  partials_issue<3>(pin, nb_in, praw);
  const double* __restrict__ gsrc = init ? z : m;  // the vector H is applied to

  RowIdx ri;
  dbl2 xv[SPN::NIT], mv[SPN::NIT], zv[SPN::NIT], wv[SPN::NIT], dv[SPN::NIT], hv[SPN::NIT], qv[SPN::NIT],
      tv[SPN::NIT], ev[SPN::NIT], rv[SPN::NIT];
  double srow[D], drow[GEO::B];
  int p0 = 0, valid = 0, i = 0;
  bool okp = false, ok = false;
  auto prefetch = [&](int tile) {
    p0 = tile * GEO::P + L.wave * GEO::G;
    const int npose = (n - p0) < GEO::G ? (n - p0) : GEO::G;
    valid = npose > 0 ? npose * GEO::T : 0;
    i = p0 + L.g;
    okp = (L.g < GEO::G) && (i < n);
    ok = okp && (L.s == 0);
    ri = row_idx_load<D, SPLIT>(Q.rowptr, Q.colidx, i, L.s, L.c, okp);
    const size_t base = (size_t)p0 * GEO::T;
#pragma unroll
    for (int it = 0; it < SPN::NIT; ++it) {
      const int pc = lane + 64 * it;
      if (2 * pc < valid) {
        xv[it] = reinterpret_cast<const dbl2*>(X + base)[pc];
        zv[it] = reinterpret_cast<const dbl2*>(z + base)[pc];
        if (!init) {
          mv[it] = reinterpret_cast<const dbl2*>(m + base)[pc];
          wv[it] = reinterpret_cast<const dbl2*>(w + base)[pc];
          ev[it] = reinterpret_cast<const dbl2*>(eta + base)[pc];
          rv[it] = reinterpret_cast<const dbl2*>(r + base)[pc];
          dv[it] = reinterpret_cast<const dbl2*>(delta + base)[pc];
          hv[it] = reinterpret_cast<const dbl2*>(Hd + base)[pc];
          qv[it] = reinterpret_cast<const dbl2*>(q + base)[pc];
          tv[it] = reinterpret_cast<const dbl2*>(t + base)[pc];
        }
      }
    }
    if (ok) {
      if (L.c < D) {
#pragma unroll
        for (int a = 0; a < D; ++a) srow[a] = S[(size_t)i * D * D + L.c * D + a];
      }
      if (dinv) {
#pragma unroll
        for (int k = 0; k < GEO::B; ++k) drow[k] = dinv[(size_t)i * GEO::BB + L.c * GEO::B + k];
      }
    }
  };
  int tile = ti_.first;
  bool have = tile < ti_.last;
  if (have) prefetch(tile);

  // ---- scalar prologue
  if (st.rtr_stop || st.tcg_done) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      store_state(sout, st);
      publish_progress(hflag, gen, st);
    }
    return;
  }
  double pr[3];
  partials_finish<3>(praw, pr, red);
  double alpha = 0.0, beta = 0.0, tau = 0.0;
  int mode = 0;  // 0: full step, 1: boundary / negative curvature (eta += tau delta, stop), 2: nothing to do
  if (init) {
    // pr[0..1] = <r0,r0>, <z0,r0> from k_tcg_update(first)
    st.norm_r0 = sqrt(pr[0]);
    st.z_r = pr[1];
    st.d_Pd = pr[1];
    st.e_Pd = 0.0;
    st.e_Pe = 0.0;
    st.d_Hd = 0.0;
    st.n_hess += 1;  // w0 = H z0
    if (st.max_inner <= 0) {
      st.tcg_done = 1;
      mode = 2;
    }
  } else {
    if (first) {
      st.d_Hd = pr[2];  // delta_0 = -z_0: <delta,H delta> = <z,w>
    } else {
      const double norm_r = sqrt(pr[0]);
      const double pw = (st.theta == 1.0) ? st.norm_r0 : pow(st.norm_r0, st.theta);  // theta = 1 (reference default)
      if (st.tcg_j >= st.min_inner && norm_r <= st.norm_r0 * (pw < st.kappa ? pw : st.kappa)) {
        st.tcg_status = (st.kappa < pw) ? TCG_LCON : TCG_SCON;
        st.tcg_done = 1;
        mode = 2;
      } else {
        beta = pr[1] / st.z_r;
        st.e_Pd = beta * (st.e_Pd + st.alpha * st.d_Pd);
        st.d_Pd = pr[1] + beta * beta * st.d_Pd;
        st.z_r = pr[1];
        st.tcg_j += 1;
        if (st.tcg_j >= st.max_inner) {
          st.tcg_done = 1;
          st.tcg_status = TCG_MAXITER;
          mode = 2;
        } else {
          st.d_Hd = pr[2] - beta * beta * st.d_Hd;
        }
      }
    }
    if (mode == 0) {
      const double d_Hd = st.d_Hd;
      alpha = st.z_r / d_Hd;
      const double e_Pe_new = st.e_Pe + 2.0 * alpha * st.e_Pd + alpha * alpha * st.d_Pd;
      st.alpha = alpha;
      const double D2 = st.Delta * st.Delta;
      if (d_Hd <= 0.0 || e_Pe_new >= D2) {
        tau = (-st.e_Pd + sqrt(st.e_Pd * st.e_Pd + st.d_Pd * (D2 - st.e_Pe))) / st.d_Pd;
        st.tcg_status = (d_Hd < 0.0) ? TCG_NEGCURV : TCG_EXCREGION;
        st.tcg_done = 1;
        mode = 1;
      } else {
        st.e_Pe = e_Pe_new;
        st.n_hess += 1;  // n = H m below
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    store_state(sout, st);
    publish_progress(hflag, gen, st);
  }
  if (mode == 2) return;

  double part[3] = {0.0, 0.0, 0.0};
  while (have) {
    const size_t base = (size_t)p0 * GEO::T;
    if (mode == 1) {  // eta += tau * delta_j, delta_j = beta delta_{j-1} - z_j
#pragma unroll
      for (int it = 0; it < SPN::NIT; ++it) {
        const int pc = lane + 64 * it;
        if (2 * pc < valid) {
          dbl2 dn, e = ev[it];
          dn.x = first ? -zv[it].x : fma(beta, dv[it].x, -zv[it].x);
          dn.y = first ? -zv[it].y : fma(beta, dv[it].y, -zv[it].y);
          e.x = fma(tau, dn.x, e.x);
          e.y = fma(tau, dn.y, e.y);
          reinterpret_cast<dbl2*>(eta + base)[pc] = e;
        }
      }
    } else {
      // (1) X and the gathered vector's own tile -> lane = (pose, column) layout
#pragma unroll
      for (int it = 0; it < SPN::NIT; ++it) {
        const int pc = lane + 64 * it;
        if (2 * pc < valid) {
          reinterpret_cast<dbl2*>(ys)[pc] = xv[it];
          reinterpret_cast<dbl2*>(vs)[pc] = init ? zv[it] : mv[it];
        }
      }
      double h[R];
      spmm_col_pre<D, R, SPLIT>(ri, Q.colidx, Q.vals, gsrc, L.s, L.c, h);
      wave_sync();
      if (ok) {
        if (L.c < D) {
          const double* vt = vs + L.g * GEO::T;
#pragma unroll
          for (int a = 0; a < D; ++a) {
#pragma unroll
            for (int k = 0; k < R; ++k) h[k] = fma(-vt[a * R + k], srow[a], h[k]);
          }
        }
        store_col<R>(hs + L.g * GEO::T + L.c * R, h);
      }
      wave_sync();
      if (ok) {
        double hz[R], sdummy[D];
        proj_col<D, R>(ys + L.g * GEO::T, hs + L.g * GEO::T, L.c, h, hz, sdummy);
        store_col<R>(os + L.g * GEO::T + L.c * R, hz);  // n = H (gathered vector), own rows
      }
      wave_sync();
      // (2) recurrences in span layout; the new w goes back to LDS for the preconditioner
#pragma unroll
      for (int it = 0; it < SPN::NIT; ++it) {
        const int pc = lane + 64 * it;
        if (2 * pc < valid) {
          const dbl2 nv = reinterpret_cast<const dbl2*>(os)[pc];
          dbl2 wn, zn = zv[it];
          if (init) {
            wn = nv;
          } else {
            dbl2 dn, hn, qn, tn, e = ev[it], rn = rv[it];
            if (first) {
              dn.x = -zv[it].x, dn.y = -zv[it].y;
              hn.x = -wv[it].x, hn.y = -wv[it].y;
              qn.x = -mv[it].x, qn.y = -mv[it].y;
              tn.x = -nv.x, tn.y = -nv.y;
            } else {
              dn.x = fma(beta, dv[it].x, -zv[it].x), dn.y = fma(beta, dv[it].y, -zv[it].y);
              hn.x = fma(beta, hv[it].x, -wv[it].x), hn.y = fma(beta, hv[it].y, -wv[it].y);
              qn.x = fma(beta, qv[it].x, -mv[it].x), qn.y = fma(beta, qv[it].y, -mv[it].y);
              tn.x = fma(beta, tv[it].x, -nv.x), tn.y = fma(beta, tv[it].y, -nv.y);
            }
            e.x = fma(alpha, dn.x, e.x), e.y = fma(alpha, dn.y, e.y);
            rn.x = fma(alpha, hn.x, rn.x), rn.y = fma(alpha, hn.y, rn.y);
            zn.x = fma(alpha, qn.x, zn.x), zn.y = fma(alpha, qn.y, zn.y);
            wn.x = fma(alpha, tn.x, wv[it].x), wn.y = fma(alpha, tn.y, wv[it].y);
            reinterpret_cast<dbl2*>(delta + base)[pc] = dn;
            reinterpret_cast<dbl2*>(Hd + base)[pc] = hn;
            reinterpret_cast<dbl2*>(q + base)[pc] = qn;
            reinterpret_cast<dbl2*>(t + base)[pc] = tn;
            reinterpret_cast<dbl2*>(eta + base)[pc] = e;
            reinterpret_cast<dbl2*>(r + base)[pc] = rn;
            reinterpret_cast<dbl2*>(z + base)[pc] = zn;
            part[0] = fma(rn.x, rn.x, part[0]);
            part[0] = fma(rn.y, rn.y, part[0]);
            part[1] = fma(zn.x, rn.x, part[1]);
            part[1] = fma(zn.y, rn.y, part[1]);
          }
          reinterpret_cast<dbl2*>(w + base)[pc] = wn;
          reinterpret_cast<dbl2*>(hs)[pc] = wn;
          part[2] = fma(zn.x, wn.x, part[2]);
          part[2] = fma(zn.y, wn.y, part[2]);
        }
      }
      wave_sync();
      // (3) m = P w = proj_X(w Dinv)
      double zz[R];
      if (ok) {
        const double* wt = hs + L.g * GEO::T;
        if (dinv) {
          jacobi_col<D, R>(wt, drow, zz);
        } else {
#pragma unroll
          for (int a = 0; a < R; ++a) zz[a] = wt[L.c * R + a];
        }
        store_col<R>(vs + L.g * GEO::T + L.c * R, zz);
      }
      wave_sync();
      if (ok) {
        double out[R], sdummy[D];
        proj_col<D, R>(ys + L.g * GEO::T, vs + L.g * GEO::T, L.c, zz, out, sdummy);
        store_col<R>(os + L.g * GEO::T + L.c * R, out);
      }
      wave_sync();
#pragma unroll
      for (int it = 0; it < SPN::NIT; ++it) {
        const int pc = lane + 64 * it;
        if (2 * pc < valid) reinterpret_cast<dbl2*>(m_out + base)[pc] = reinterpret_cast<const dbl2*>(os)[pc];
      }
      wave_sync();
    }
    tile += ti_.step;
    have = tile < ti_.last;
    if (have) prefetch(tile);
  }
  if (mode == 0) {
    if (init) {  // carry <r0,r0>, <z0,r0> is not needed again: the state holds them
      part[0] = 0.0;
      part[1] = 0.0;
    }
    store_partials<3>(part, pout, red);
  }
}
