// kernels/tcg.h -- the two kernels of the tCG loop (fused Hessian step and residual / iterate update): shared scalar prologues,
// the generic-layout kernels (odd tile size) and the span kernels (even tile size: every 3-D case).
// Part of kernels.h (included inside namespace dpgo, in this order: common.h, problem.h, tcg.h, persist.h, multilevel.h, dense.h, manifold.h, rtr.h, agent.h, init.h).
#pragma once

// ================================================================ K7b + K1/K3/K2 fused: one tCG step
// One launch per tCG iteration:
//  (i)   prologue = the scalar half of the direction update (ROPTLIB tCG_TR): convergence test
//        |r| <= |r0| min(|r0|^theta, kappa), beta = z_r'/z_r, e_Pd / d_Pd recurrences, from the
//        <r,r>, <z,r> partials of k_tcg_update (first = 1: norm_r0, z_r, d_Pd initialisation);
//  (ii)  Hz = proj_X( z Q - z_rot S ): the block-SpMM gathers the preconditioned residual z;
//  (iii) row-local, in place:  delta <- beta*delta - z,   H delta <- beta*(H delta) - Hz
//        (H is linear on the tangent space, so this equals H applied to the new delta; it lets the
//        direction update ride in the SpMM epilogue instead of costing a second gather or a separate
//        kernel: 3 -> 2 launches per tCG iteration), and the <delta, H delta> partial.
// The oracle has the same option (hess_recurrence) for trajectory-level parity tests.
// ---------------------------------------------------------------- tCG scalar prologues (shared)
// Direction-update scalars (ROPTLIB tCG_TR): returns false when this launch has nothing left to do.
__device__ __forceinline__ bool tcg_hess_prologue(DevState& st, const double* __restrict__ pin, int nb_in, int first,
                                                  double* red, double& beta, const PartialRaw<2>* early = nullptr) {
  double pr[2];
  if (early)
    partials_finish<2>(*early, pr, red);
  else
    load_partials<2>(pin, nb_in, pr, red);
  const double r_r = pr[0], z_r_new = pr[1];
  beta = 0.0;
  if (first) {
    st.norm_r0 = sqrt(r_r);
    st.z_r = z_r_new;
    st.d_Pd = z_r_new;
    st.e_Pd = 0.0;
    if (st.max_inner <= 0) {
      st.tcg_done = 1;
      return false;
    }
    return true;
  }
  const double norm_r = sqrt(r_r);
  const double pw = (st.theta == 1.0) ? st.norm_r0 : pow(st.norm_r0, st.theta);  // theta = 1 (reference default)
  if (st.tcg_j >= st.min_inner && norm_r <= st.norm_r0 * (pw < st.kappa ? pw : st.kappa)) {
    st.tcg_status = (st.kappa < pw) ? TCG_LCON : TCG_SCON;
    st.tcg_done = 1;
    return false;
  }
  beta = z_r_new / st.z_r;
  st.e_Pd = beta * (st.e_Pd + st.alpha * st.d_Pd);
  st.d_Pd = z_r_new + beta * beta * st.d_Pd;
  st.z_r = z_r_new;
  st.tcg_j += 1;
  if (st.tcg_j >= st.max_inner) {
    st.tcg_done = 1;
    st.tcg_status = TCG_MAXITER;
    return false;
  }
  return true;
}

// Step-length scalars: mode 0 = normal step, 1 = boundary step (eta += tau*delta, stop), 2 = initialisation.
__device__ __forceinline__ int tcg_update_prologue(DevState& st, const double* __restrict__ pin, int nb_in, int first,
                                                   double* red, double& alpha, double& tau,
                                                   const PartialRaw<1>* early = nullptr) {
  alpha = 0.0;
  tau = 0.0;
  if (first) {
    st.tcg_done = 0;
    st.tcg_j = 0;
    st.tcg_status = TCG_MAXITER;
    st.e_Pe = 0.0;
    st.e_Pd = 0.0;
    return 2;
  }
  double dh[1];
  if (early)
    partials_finish<1>(*early, dh, red);
  else
    load_partials<1>(pin, nb_in, dh, red);
  const double d_Hd = dh[0];
  alpha = st.z_r / d_Hd;
  const double e_Pe_new = st.e_Pe + 2.0 * alpha * st.e_Pd + alpha * alpha * st.d_Pd;
  st.n_hess += 1;
  st.alpha = alpha;
  const double D2 = st.Delta * st.Delta;
  if (d_Hd <= 0.0 || e_Pe_new >= D2) {
    tau = (-st.e_Pd + sqrt(st.e_Pd * st.e_Pd + st.d_Pd * (D2 - st.e_Pe))) / st.d_Pd;
    st.tcg_status = (d_Hd < 0.0) ? TCG_NEGCURV : TCG_EXCREGION;
    st.tcg_done = 1;
    return 1;
  }
  st.e_Pe = e_Pe_new;
  return 0;
}

template <int D, int R, int SPLIT>
__global__ __launch_bounds__(kBlock, DPGO_LB_HESS) void k_tcg_hess(BsrDev Q, const double* __restrict__ X,
                                                     const double* __restrict__ S, const double* __restrict__ z,
                                                     double* __restrict__ delta, double* __restrict__ Hd,
                                                     const double* __restrict__ pin, int nb_in,
                                                     double* __restrict__ pout, const DevState* __restrict__ sin,
                                                     DevState* __restrict__ sout, int first, int n,
                                                     unsigned long long* hflag, unsigned gen) {
  // generic layout (odd tile size: the 2-D cases with odd r); even tile sizes run k_tcg_hess_span
  using GEO = Geo<D, R, SPLIT>;
  __shared__ __attribute__((aligned(16))) double sm[kWaves][3][GEO::G][GEO::T];
  __shared__ double red[kWaves * kNP];
  DevState st;
  load_state(st, sin);
  gen = state_gen(st, gen);
  if (st.rtr_stop || st.tcg_done) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      store_state(sout, st);
      publish_progress(hflag, gen, st);
    }
    return;
  }
  double beta;
  const bool go = tcg_hess_prologue(st, pin, nb_in, first, red, beta);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    store_state(sout, st);
    publish_progress(hflag, gen, st);
  }
  if (!go) return;

  const LaneId L = lane_id<D, SPLIT>();
  const int ntiles = (n + GEO::P - 1) / GEO::P;
  double part[1] = {0.0};
  const TileIter ti_ = tile_iter(ntiles);
  for (int tile = ti_.first; tile < ti_.last; tile += ti_.step) {
    const int i = tile * GEO::P + L.wave * GEO::G + L.g;
    const bool okp = (L.g < GEO::G) && (i < n);
    const bool ok = okp && (L.s == 0);
    double h[R], zc[R], x[R];
    const size_t off = (size_t)i * GEO::T + L.c * R;
    double* ys = ok ? &sm[L.wave][0][L.g][0] : nullptr;
    double* vs = ok ? &sm[L.wave][1][L.g][0] : nullptr;
    double* hs = ok ? &sm[L.wave][2][L.g][0] : nullptr;
    // issue the epilogue's loads first: they overlap the gather's index -> tile latency chain
    double srow[D], dl[R], hd[R];
    if (ok) {
      load_col<R>(X + off, x);
      load_col<R>(z + off, zc);
      if (L.c < D) {
#pragma unroll
        for (int a = 0; a < D; ++a) srow[a] = S[(size_t)i * D * D + L.c * D + a];
      }
      if (!first) {
        load_col<R>(delta + off, dl);
        load_col<R>(Hd + off, hd);
      }
    }
    spmm_col<D, R, SPLIT>(Q.rowptr, Q.colidx, Q.vals, z, i, L.s, L.c, okp, h);
    if (ok) {
      store_col<R>(ys + L.c * R, x);
      store_col<R>(vs + L.c * R, zc);
    }
    wave_sync();
    if (ok) {
      if (L.c < D) {
#pragma unroll
        for (int a = 0; a < D; ++a) {
#pragma unroll
          for (int k = 0; k < R; ++k) h[k] = fma(-vs[a * R + k], srow[a], h[k]);
        }
      }
      store_col<R>(hs + L.c * R, h);
    }
    wave_sync();
    if (ok) {
      double hz[R], s[D];
      proj_col<D, R>(ys, hs, L.c, h, hz, s);
      if (first) {
#pragma unroll
        for (int a = 0; a < R; ++a) {
          dl[a] = -zc[a];
          hd[a] = -hz[a];
        }
      } else {
#pragma unroll
        for (int a = 0; a < R; ++a) {
          dl[a] = fma(beta, dl[a], -zc[a]);
          hd[a] = fma(beta, hd[a], -hz[a]);
        }
      }
#pragma unroll
      for (int a = 0; a < R; ++a) part[0] = fma(dl[a], hd[a], part[0]);
      store_col<R>(delta + off, dl);
      store_col<R>(Hd + off, hd);
    }
    wave_sync();
  }
  store_partials<1>(part, pout, red);
}

// ================================================================ span kernels (pose tile size even: all 3-D cases)
// Same arithmetic as k_tcg_hess / k_tcg_update; the differences are purely about memory:
//  * own-tile vectors move as lane-linear 16-byte pieces (Span<>), element-wise recurrences run in span layout;
//  * the FIRST tile's global loads (row pointer, column indices, vector pieces) are issued before the scalar
//    prologue (state record + partial-sum reduction), so the two dependent-latency chains overlap -- this is
//    what matters for small blocks (multi-GPU strong scaling), where a kernel is a chain of ~15 memory latencies.
template <int D, int R, int SPLIT, int NTS = 0>
__global__ __launch_bounds__(kBlock) void k_tcg_hess_span(BsrDev Q, const double* __restrict__ X,
                                                          const double* __restrict__ S, const double* __restrict__ z,
                                                          double* __restrict__ delta, double* __restrict__ Hd,
                                                          const double* __restrict__ pin, int nb_in,
                                                          double* __restrict__ pout, const DevState* __restrict__ sin,
                                                          DevState* __restrict__ sout, int first, int n,
                                                          unsigned long long* hflag, unsigned gen) {
  using GEO = Geo<D, R, SPLIT>;
  using SPN = Span<D, R, SPLIT>;
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

  // ---- per-tile prefetch state
  RowIdx ri;
  dbl2 xv[SPN::NIT], zv[SPN::NIT], dv[SPN::NIT], hv[SPN::NIT];
  double srow[D];
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
    const dbl2* X2 = reinterpret_cast<const dbl2*>(X + base);
    const dbl2* z2 = reinterpret_cast<const dbl2*>(z + base);
    const dbl2* d2 = reinterpret_cast<const dbl2*>(delta + base);
    const dbl2* h2 = reinterpret_cast<const dbl2*>(Hd + base);
#pragma unroll
    for (int it = 0; it < SPN::NIT; ++it) {
      const int pc = lane + 64 * it;
      if (2 * pc < valid) {
        xv[it] = ld_stream<NTS>(X2 + pc);
        zv[it] = z2[pc];
        if (!first) {
          dv[it] = ld_stream<NTS>(d2 + pc);
          hv[it] = ld_stream<NTS>(h2 + pc);
        }
      }
    }
    if (ok && L.c < D) {
#pragma unroll
      for (int a = 0; a < D; ++a) srow[a] = ld_stream<NTS>(S + (size_t)i * D * D + L.c * D + a);
    }
  };
  // ---- everything the prologue needs is requested before the first wait: state record (scalar loads), the
  // previous kernel's partial sums (small blocks only: the registers would cost the big-block kernel an
  // occupancy step), then the first tile.  A small-block launch is a chain of dependent memory round trips
  // (rocprof: 9.4 us for 2500 poses); this takes two of them off the chain.
  DPGO_TL_DECL;
  DPGO_STAMP(0, 0);
  DevState st;
  load_state(st, sin);
  gen = state_gen(st, gen);
  [[maybe_unused]] PartialRaw<2> praw;
  if constexpr (SPLIT > 1) partials_issue<2>(pin, nb_in, praw);
  int tile = ti_.first;
  bool have = tile < ti_.last;
  if (have) prefetch(tile);

  DPGO_STAMP(0, 1);
  // ---- scalar prologue
  if (st.rtr_stop || st.tcg_done) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      store_state(sout, st);
      publish_progress(hflag, gen, st);
    }
    return;
  }
  double beta;
  const bool go = tcg_hess_prologue(st, pin, nb_in, first, red, beta, (SPLIT > 1) ? &praw : nullptr);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    store_state(sout, st);
    publish_progress(hflag, gen, st);
  }
  if (!go) return;
  DPGO_STAMP(0, 2);

  double part[1] = {0.0};
  while (have) {
#pragma unroll
    for (int it = 0; it < SPN::NIT; ++it) {
      const int pc = lane + 64 * it;
      if (2 * pc < valid) {
        reinterpret_cast<dbl2*>(ys)[pc] = xv[it];
        reinterpret_cast<dbl2*>(vs)[pc] = zv[it];
      }
    }
    DPGO_STAMP(0, 3);
    double h[R];
    spmm_col_pre<D, R, SPLIT>(ri, Q.colidx, Q.vals, z, L.s, L.c, h);
    wave_sync();
    DPGO_STAMP(0, 4);
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
      store_col<R>(os + L.g * GEO::T + L.c * R, hz);
    }
    wave_sync();
    {
      const size_t base = (size_t)p0 * GEO::T;
      dbl2* d2 = reinterpret_cast<dbl2*>(delta + base);
      dbl2* h2 = reinterpret_cast<dbl2*>(Hd + base);
#pragma unroll
      for (int it = 0; it < SPN::NIT; ++it) {
        const int pc = lane + 64 * it;
        if (2 * pc < valid) {
          const dbl2 hzv = reinterpret_cast<const dbl2*>(os)[pc];
          dbl2 dn, hn;
          if (first) {
            dn.x = -zv[it].x;
            dn.y = -zv[it].y;
            hn.x = -hzv.x;
            hn.y = -hzv.y;
          } else {
            dn.x = fma(beta, dv[it].x, -zv[it].x);
            dn.y = fma(beta, dv[it].y, -zv[it].y);
            hn.x = fma(beta, hv[it].x, -hzv.x);
            hn.y = fma(beta, hv[it].y, -hzv.y);
          }
          st_stream<NTS>(d2 + pc, dn);
          st_stream<NTS>(h2 + pc, hn);
          part[0] = fma(dn.x, hn.x, part[0]);
          part[0] = fma(dn.y, hn.y, part[0]);
        }
      }
    }
    wave_sync();
    DPGO_STAMP(0, 5);
    tile += ti_.step;
    have = tile < ti_.last;
    if (have) prefetch(tile);
  }
  store_partials<1>(part, pout, red);
  DPGO_STAMP(0, 6);
  DPGO_COMMIT(0);
}

// k_tcg_hess_span on the symmetric storage of Q (spmm_sym_pre, common.h): big blocks only (one pose per D+1 lanes).  The
// own-tile pieces of delta / H delta are requested after the gather instead of one tile ahead and z is re-read from LDS
// (registers).  The gather keeps the loads of DPGO_HESS_BATCH blocks in flight per wave, which costs the third wave per SIMD
// (217 VGPRs) and still wins: 2 waves x 4 blocks in flight against 3 x 1 (39.7 -> 37.3 us).  In-kernel timeline of a tile
// (tools/timeline_tiles.py, profiles/r05_v4_timeline_tiles.txt): ~8 us, of which the gather's two round trips 3 us and the
// epilogue behind it 4 us.  Tried on top and measured neutral, i.e. the launch is bound by what the memory system
// delivers to 512 resident workgroups, not by one wave's chain of round trips: delta / H delta requested in front of the
// gather (37.4..37.7 us), a three-stage pipeline over the tiles (row extents two tiles ahead, indices + own X, z, S one tile
// ahead: tile 7 us, launch 37.4..38.0 us), the previous kernel's partial sums requested in front of the first tile.
template <int D, int R, int NTS>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(DPGO_SYM_WAVES, DPGO_SYM_WAVES))) void k_tcg_hess_sym(BsrSymDev Q, const double* __restrict__ X,
                                                          const double* __restrict__ S, const double* __restrict__ z,
                                                          double* __restrict__ delta, double* __restrict__ Hd,
                                                          const double* __restrict__ pin, int nb_in,
                                                          double* __restrict__ pout, const DevState* __restrict__ sin,
                                                          DevState* __restrict__ sout, int first, int n,
                                                          unsigned long long* hflag, unsigned gen) {
  constexpr int SPLIT = 1;
  using GEO = Geo<D, R, SPLIT>;
  using SPN = Span<D, R, SPLIT>;
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

  // ---- per-tile prefetch state
  SymIdx si;
  dbl2 xv[SPN::NIT], zv[SPN::NIT];
  double srow[D];
  int p0 = 0, valid = 0, i = 0;
  bool okp = false, ok = false;
  auto prefetch = [&](int tk) {
    const int tile = tile_of(Q, tk);  // (the walk over the tiles: BsrSymDevT::tord)
    p0 = tile * GEO::P + L.wave * GEO::G;
    const int npose = (n - p0) < GEO::G ? (n - p0) : GEO::G;
    valid = npose > 0 ? npose * GEO::T : 0;
    i = p0 + L.g;
    okp = (L.g < GEO::G) && (i < n);
    ok = okp && (L.s == 0);
    si = sym_idx_load<D>(Q, i, L.c, okp);
    const size_t base = (size_t)p0 * GEO::T;
    const dbl2* X2 = reinterpret_cast<const dbl2*>(X + base);
    const dbl2* z2 = reinterpret_cast<const dbl2*>(z + base);
#pragma unroll
    for (int it = 0; it < SPN::NIT; ++it) {
      const int pc = lane + 64 * it;
      if (2 * pc < valid) {
        xv[it] = ld_stream<NTS>(X2 + pc);
        zv[it] = z2[pc];
      }
    }
    if (ok && L.c < D) {
#pragma unroll
      for (int a = 0; a < D; ++a) srow[a] = ld_stream<NTS>(S + (size_t)i * D * D + L.c * D + a);
    }
  };
  // ---- everything the prologue needs is requested before the first wait: state record (scalar loads), the
  // previous kernel's partial sums (small blocks only: the registers would cost the big-block kernel an
  // occupancy step), then the first tile.  A small-block launch is a chain of dependent memory round trips
  // (rocprof: 9.4 us for 2500 poses); this takes two of them off the chain.
  DPGO_TL_DECL;
  DPGO_TL_TILES_DECL;
  DPGO_STAMP(0, 0);
  DevState st;
  load_state(st, sin);
  gen = state_gen(st, gen);
  [[maybe_unused]] PartialRaw<2> praw;
  int tile = ti_.first;
  bool have = tile < ti_.last;
  if (have) prefetch(tile);

  DPGO_STAMP(0, 1);
  // ---- scalar prologue
  if (st.rtr_stop || st.tcg_done) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      store_state(sout, st);
      publish_progress(hflag, gen, st);
    }
    return;
  }
  double beta;
  const bool go = tcg_hess_prologue(st, pin, nb_in, first, red, beta, (SPLIT > 1) ? &praw : nullptr);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    store_state(sout, st);
    publish_progress(hflag, gen, st);
  }
  if (!go) return;
  DPGO_STAMP(0, 2);
  DPGO_STAMP_ENTRY;
  DPGO_STAMP_AT(1);

  double part[1] = {0.0};
  while (have) {
    DPGO_STAMP_TILE(0);
#pragma unroll
    for (int it = 0; it < SPN::NIT; ++it) {
      const int pc = lane + 64 * it;
      if (2 * pc < valid) {
        reinterpret_cast<dbl2*>(ys)[pc] = xv[it];
        reinterpret_cast<dbl2*>(vs)[pc] = zv[it];
      }
    }
    DPGO_STAMP(0, 3);
    DPGO_STAMP_TILE(1);
    double h[R];
    spmm_sym_pre<D, R, DPGO_HESS_BATCH>(si, Q, z, L.c, h);
    DPGO_TL_USE(h[0]);
    DPGO_STAMP_TILE(2);
    // delta, H delta of the own tile: requested after the gather (its accumulators need the registers), consumed after
    // the projection
    dbl2 dv[SPN::NIT], hv[SPN::NIT];
    if (!first) {
      const size_t base = (size_t)p0 * GEO::T;
      const dbl2* d2 = reinterpret_cast<const dbl2*>(delta + base);
      const dbl2* h2 = reinterpret_cast<const dbl2*>(Hd + base);
#pragma unroll
      for (int it = 0; it < SPN::NIT; ++it) {
        const int pc = lane + 64 * it;
        if (2 * pc < valid) {
          dv[it] = ld_stream<NTS>(d2 + pc);
          hv[it] = ld_stream<NTS>(h2 + pc);
        }
      }
    }
    wave_sync();
    DPGO_STAMP(0, 4);
    DPGO_STAMP_TILE(3);
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
      store_col<R>(os + L.g * GEO::T + L.c * R, hz);
    }
    wave_sync();
    DPGO_STAMP_TILE(4);
    {
      const size_t base = (size_t)p0 * GEO::T;
      dbl2* d2 = reinterpret_cast<dbl2*>(delta + base);
      dbl2* h2 = reinterpret_cast<dbl2*>(Hd + base);
#pragma unroll
      for (int it = 0; it < SPN::NIT; ++it) {
        const int pc = lane + 64 * it;
        if (2 * pc < valid) {
          const dbl2 hzv = reinterpret_cast<const dbl2*>(os)[pc];
          const dbl2 zl = reinterpret_cast<const dbl2*>(vs)[pc];
          dbl2 dn, hn;
          if (first) {
            dn.x = -zl.x;
            dn.y = -zl.y;
            hn.x = -hzv.x;
            hn.y = -hzv.y;
          } else {
            dn.x = fma(beta, dv[it].x, -zl.x);
            dn.y = fma(beta, dv[it].y, -zl.y);
            hn.x = fma(beta, hv[it].x, -hzv.x);
            hn.y = fma(beta, hv[it].y, -hzv.y);
          }
          st_stream<NTS>(d2 + pc, dn);
          st_stream<NTS>(h2 + pc, hn);
          part[0] = fma(dn.x, hn.x, part[0]);
          part[0] = fma(dn.y, hn.y, part[0]);
        }
      }
    }
    wave_sync();
    DPGO_STAMP(0, 5);
    tile += ti_.step;
    have = tile < ti_.last;
    if (have) prefetch(tile);
    DPGO_STAMP_TILE(5);
    DPGO_TILE_NEXT;
  }
  store_partials<1>(part, pout, red);
  DPGO_STAMP(0, 6);
  DPGO_STAMP_AT(2);
  DPGO_COMMIT(0);
}

// k_tcg_hess_sym with the own-tile streams moved by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction, straight
// into the wave's LDS tiles, no staging registers) -- VERDICT r5 item 3a.  X and z of the NEXT tile are requested at the top
// of the current tile (double-buffered tiles), delta / H delta of the current tile in front of the gather; the h / proj(h)
// tiles share one buffer.  Seven tiles per wave (70 KB per workgroup: two workgroups per CU, as the register version).
// Same arithmetic in the same order as k_tcg_hess_sym: bit-identical results.  WAVES: waves per SIMD it is compiled for,
// HB: blocks whose loads the gather keeps in flight.  DPGO_HESS_DMA selects it (solve.hip); measurements in DESIGN.md.
// NT = 1: non-temporal (aux = 2), for operands a launch touches once -- as ld_stream<1> of the register version
template <int D, int R, int NT = 0>
__device__ __forceinline__ void span_dma(const double* __restrict__ src, double* lds_tile, int valid) {
  using SPN = Span<D, R, 1>;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int it = 0; it < SPN::NIT; ++it) {
    const int pc = lane + 64 * it;
    if (2 * pc < valid) {
      if constexpr (NT)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const dbl2*>(src) + pc),
                                         (__attribute__((address_space(3))) void*)(lds_tile + 128 * it), 16, 0, 2);
      else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const dbl2*>(src) + pc),
                                         (__attribute__((address_space(3))) void*)(lds_tile + 128 * it), 16, 0, 0);
    }
  }
}
template <int D, int R, int NTS, int WAVES, int HB, int DBUF>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_tcg_hess_sym_dma(
    BsrSymDev Q, const double* __restrict__ X, const double* __restrict__ S, const double* __restrict__ z,
    double* __restrict__ delta, double* __restrict__ Hd, const double* __restrict__ pin, int nb_in,
    double* __restrict__ pout, const DevState* __restrict__ sin, DevState* __restrict__ sout, int first, int n,
    unsigned long long* hflag, unsigned gen) {
  using GEO = Geo<D, R, 1>;
  using SPN = Span<D, R, 1>;
  // DBUF = 1: X[2], z[2], delta, H delta, h / proj(h) -- the next tile's X, z land while this tile is worked on (70 KB per
  // workgroup: 2 per CU); DBUF = 0: one X and one z tile, requested when the previous tile is done with them (51 KB: 3 per CU)
  constexpr int NX = DBUF ? 2 : 1, NT = 2 * NX + 3;
  __shared__ __attribute__((aligned(16))) double sm[kWaves][NT][GEO::G][GEO::T];
  __shared__ double red[kWaves * kNP];
  const LaneId L = lane_id<D, 1>();
  const int lane = threadIdx.x & 63;
  const int ntiles = (n + GEO::P - 1) / GEO::P;
  const TileIter ti_ = tile_iter(ntiles);
  auto tile_ptr = [&](int k) { return &sm[L.wave][k][0][0]; };
  double* ds = tile_ptr(2 * NX);
  double* es = tile_ptr(2 * NX + 1);
  double* hs = tile_ptr(2 * NX + 2);

  auto span_of = [&](int tile, int& p0, int& valid) {
    p0 = tile * GEO::P + L.wave * GEO::G;
    const int npose = (n - p0) < GEO::G ? (n - p0) : GEO::G;
    valid = npose > 0 ? npose * GEO::T : 0;
  };
  // ---- per-tile register state: row extents / preloaded indices and the S rows (requested one tile ahead, as before)
  SymIdx si;
  double srow[D];
  int p0 = 0, valid = 0, i = 0;
  bool okp = false, ok = false;
  auto prefetch_idx = [&](int tile) {
    span_of(tile, p0, valid);
    i = p0 + L.g;
    okp = (L.g < GEO::G) && (i < n);
    ok = okp;
    si = sym_idx_load<D>(Q, i, L.c, okp);
    if (ok && L.c < D) {
#pragma unroll
      for (int a = 0; a < D; ++a) srow[a] = ld_stream<NTS>(S + (size_t)i * D * D + L.c * D + a);
    }
  };
  auto dma_xz = [&](int tile, int buf) {
    int q0, v;
    span_of(tile, q0, v);
    const size_t base = (size_t)q0 * GEO::T;
    span_dma<D, R, NTS>(X + base, tile_ptr(buf), v);
    span_dma<D, R>(z + base, tile_ptr(NX + buf), v);  // (z is gathered by the neighbours' rows: kept in L2)
  };
  DevState st;
  load_state(st, sin);
  gen = state_gen(st, gen);
  int tk = ti_.first;
  bool have = tk < ti_.last;
  int tile = have ? tile_of(Q, tk) : 0;
  int buf = 0;
  if (have) {
    dma_xz(tile, 0);
    prefetch_idx(tile);
  }
  // ---- scalar prologue
  if (st.rtr_stop || st.tcg_done) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      store_state(sout, st);
      publish_progress(hflag, gen, st);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (no DMA may land in LDS a later workgroup owns)
    return;
  }
  double beta;
  const bool go = tcg_hess_prologue(st, pin, nb_in, first, red, beta, nullptr);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    store_state(sout, st);
    publish_progress(hflag, gen, st);
  }
  if (!go) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  double part[1] = {0.0};
  while (have) {
    const int tk2 = tk + ti_.step;
    const bool have2 = tk2 < ti_.last;
    const int tile2 = have2 ? tile_of(Q, tk2) : 0;
    const size_t base = (size_t)p0 * GEO::T;
    if (DBUF && have2) dma_xz(tile2, buf ^ 1);  // the next tile's X, z: in flight during this tile's gather and epilogue
    if (!first) {
      span_dma<D, R, NTS>(delta + base, ds, valid);
      span_dma<D, R, NTS>(Hd + base, es, valid);
    }
    double* ys = tile_ptr(buf);
    double* vs = tile_ptr(NX + buf);
    double h[R];
    spmm_sym_pre<D, R, HB>(si, Q, z, L.c, h);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every DMA piece of this wave has landed
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
    double hz[R];
    if (ok) {
      double sdummy[D];
      proj_col<D, R>(ys + L.g * GEO::T, hs + L.g * GEO::T, L.c, h, hz, sdummy);
    }
    wave_sync();  // every lane of a pose has read the h tile before proj(h) replaces it
    if (ok) store_col<R>(hs + L.g * GEO::T + L.c * R, hz);
    wave_sync();
    {
      dbl2* d2 = reinterpret_cast<dbl2*>(delta + base);
      dbl2* h2 = reinterpret_cast<dbl2*>(Hd + base);
#pragma unroll
      for (int it = 0; it < SPN::NIT; ++it) {
        const int pc = lane + 64 * it;
        if (2 * pc < valid) {
          const dbl2 hzv = reinterpret_cast<const dbl2*>(hs)[pc];
          const dbl2 zl = reinterpret_cast<const dbl2*>(vs)[pc];
          dbl2 dn, hn;
          if (first) {
            dn.x = -zl.x;
            dn.y = -zl.y;
            hn.x = -hzv.x;
            hn.y = -hzv.y;
          } else {
            const dbl2 dv = reinterpret_cast<const dbl2*>(ds)[pc];
            const dbl2 hv = reinterpret_cast<const dbl2*>(es)[pc];
            dn.x = fma(beta, dv.x, -zl.x);
            dn.y = fma(beta, dv.y, -zl.y);
            hn.x = fma(beta, hv.x, -hzv.x);
            hn.y = fma(beta, hv.y, -hzv.y);
          }
          st_stream<NTS>(d2 + pc, dn);
          st_stream<NTS>(h2 + pc, hn);
          part[0] = fma(dn.x, hn.x, part[0]);
          part[0] = fma(dn.y, hn.y, part[0]);
        }
      }
    }
    wave_sync();
    tk = tk2;
    have = have2;
    tile = tile2;
    if (DBUF) buf ^= 1;
    if (!DBUF && have) dma_xz(tile, 0);  // (the wave is done with its X / z tiles: wave_sync above)
    if (have) prefetch_idx(tile);
  }
  store_partials<1>(part, pout, red);
}

#ifndef DPGO_UPDATE_WAVES
#define DPGO_UPDATE_WAVES 0  // waves per SIMD k_tcg_update_span is compiled for (0: the compiler's choice -- 203 VGPRs = 2 waves)
#endif
// ML = 1: the launch is known to be in multilevel mode (ml_omega > 0: z receives the unprojected pre-smoothing step, the
// iterate is not read) -- the instance the 100k loop launches: without the iterate's pieces and the projection it fits
// DPGO_UPDATE_WAVES_ML waves per SIMD.  ML = 0: the mode is the runtime argument (block-Jacobi / no preconditioner / either).
#ifndef DPGO_UPDATE_WAVES_ML
#define DPGO_UPDATE_WAVES_ML 3
#endif
template <int D, int R, int ML = 0>
__global__ __launch_bounds__(kBlock)
    __attribute__((amdgpu_waves_per_eu(ML ? DPGO_UPDATE_WAVES_ML : (DPGO_UPDATE_WAVES ? DPGO_UPDATE_WAVES : 1),
                                       ML ? DPGO_UPDATE_WAVES_ML : (DPGO_UPDATE_WAVES ? DPGO_UPDATE_WAVES : 8))))
void k_tcg_update_span(const double* __restrict__ X, const double* __restrict__ g,
                                                            const double* __restrict__ dinv,
                                                            const double* __restrict__ delta,
                                                            const double* __restrict__ Hd, double* __restrict__ eta,
                                                            double* __restrict__ r, double* __restrict__ z,
                                                            const double* __restrict__ pin, int nb_in,
                                                            double* __restrict__ pout, const DevState* __restrict__ sin,
                                                            DevState* __restrict__ sout, int first, int n,
                                                            unsigned long long* hflag, unsigned gen,
                                                            double ml_omega, float* __restrict__ z32 = nullptr) {
  // ml_omega > 0 (fused multilevel preconditioner): z receives the pre-smoothing step w Dinv r, unprojected -- into z32
  // instead, rounded to fp32, when the cycle keeps its internal vectors in that storage (kernel-uniform)
  using GEO = Geo<D, R>;
  using SPN = Span<D, R, 1>;
  __shared__ __attribute__((aligned(16))) double sm[kWaves][4][GEO::G][GEO::T];
  __shared__ double red[kWaves * kNP];
  const LaneId L = lane_id<D>();
  const int lane = threadIdx.x & 63;
  const int ntiles = (n + GEO::P - 1) / GEO::P;
  const TileIter ti_ = tile_iter(ntiles);
  double* ys = &sm[L.wave][0][0][0];
  double* rs = &sm[L.wave][1][0][0];
  double* zs = &sm[L.wave][2][0][0];
  double* os = &sm[L.wave][3][0][0];

  const bool ml_mode = ML ? true : (ml_omega > 0.0);  // (ML: compile-time)
  dbl2 xv[SPN::NIT], ev[SPN::NIT], dv[SPN::NIT], hv[SPN::NIT], rv[SPN::NIT];
  double drow[GEO::B];
  int p0 = 0, valid = 0, i = 0;
  bool ok = false;
  auto prefetch = [&](int tile) {
    p0 = tile * GEO::P + L.wave * GEO::G;
    const int npose = (n - p0) < GEO::G ? (n - p0) : GEO::G;
    valid = npose > 0 ? npose * GEO::T : 0;
    i = p0 + L.g;
    ok = (L.g < GEO::G) && (i < n);
    const size_t base = (size_t)p0 * GEO::T;
    const dbl2* X2 = reinterpret_cast<const dbl2*>(X + base);
    const dbl2* g2 = reinterpret_cast<const dbl2*>(g + base);
    const dbl2* e2 = reinterpret_cast<const dbl2*>(eta + base);
    const dbl2* d2 = reinterpret_cast<const dbl2*>(delta + base);
    const dbl2* h2 = reinterpret_cast<const dbl2*>(Hd + base);
    const dbl2* r2 = reinterpret_cast<const dbl2*>(r + base);
#pragma unroll
    for (int it = 0; it < SPN::NIT; ++it) {
      const int pc = lane + 64 * it;
      if (2 * pc < valid) {
        // (the iterate is only needed by the tangent projection of the block-Jacobi / unpreconditioned z; the multilevel
        // pre-smoothing step x1 = w Dinv r is not projected: 16 MB less per launch at 100k poses)
        if (!ml_mode) xv[it] = X2[pc];
        if (first) {
          rv[it] = g2[pc];
        } else {
          ev[it] = e2[pc];
          dv[it] = d2[pc];
          hv[it] = h2[pc];
          rv[it] = r2[pc];
        }
      }
    }
    if (ok && dinv) {
#pragma unroll
      for (int k = 0; k < GEO::B; ++k) drow[k] = dinv[(size_t)i * GEO::BB + L.c * GEO::B + k];
    }
  };
  DPGO_TL_DECL;
  DPGO_STAMP(1, 0);
  // state record and partial sums are requested before the first tile (see k_tcg_hess_span)
  DevState st;
  load_state(st, sin);
  gen = state_gen(st, gen);
  PartialRaw<1> praw;
  partials_issue<1>(pin, nb_in, praw);
  int tile = ti_.first;
  bool have = tile < ti_.last;
  if (have) prefetch(tile);

  if (st.rtr_stop || (!first && st.tcg_done)) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      store_state(sout, st);
      publish_progress(hflag, gen, st);
    }
    return;
  }
  double alpha, tau;
  const int mode = tcg_update_prologue(st, pin, nb_in, first, red, alpha, tau, &praw);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    store_state(sout, st);
    publish_progress(hflag, gen, st);
  }

  DPGO_STAMP(1, 2);
  double part[2] = {0.0, 0.0};
  while (have) {
    const size_t base = (size_t)p0 * GEO::T;
    dbl2* eta2 = reinterpret_cast<dbl2*>(eta + base);
    if (mode == 1) {  // workgroup-uniform: eta += tau * delta, then tCG stops
#pragma unroll
      for (int it = 0; it < SPN::NIT; ++it) {
        const int pc = lane + 64 * it;
        if (2 * pc < valid) {
          dbl2 e = ev[it];
          e.x = fma(tau, dv[it].x, e.x);
          e.y = fma(tau, dv[it].y, e.y);
          eta2[pc] = e;
        }
      }
    } else {
      dbl2* r2 = reinterpret_cast<dbl2*>(r + base);
#pragma unroll
      for (int it = 0; it < SPN::NIT; ++it) {
        const int pc = lane + 64 * it;
        if (2 * pc < valid) {
          if (!ml_mode) reinterpret_cast<dbl2*>(ys)[pc] = xv[it];
          dbl2 rr = rv[it];
          if (mode == 2) {
            dbl2 zero;
            zero.x = 0.0;
            zero.y = 0.0;
            eta2[pc] = zero;
          } else {
            dbl2 e = ev[it];
            e.x = fma(alpha, dv[it].x, e.x);
            e.y = fma(alpha, dv[it].y, e.y);
            rr.x = fma(alpha, hv[it].x, rr.x);
            rr.y = fma(alpha, hv[it].y, rr.y);
            eta2[pc] = e;
          }
          r2[pc] = rr;
          reinterpret_cast<dbl2*>(rs)[pc] = rr;
          part[0] = fma(rr.x, rr.x, part[0]);
          part[0] = fma(rr.y, rr.y, part[0]);
        }
      }
      wave_sync();
      double zz[R];
      if (ok) {
        const double* rt = rs + L.g * GEO::T;
        if (dinv) {
          jacobi_col<D, R>(rt, drow, zz);
        } else {
#pragma unroll
          for (int a = 0; a < R; ++a) zz[a] = rt[L.c * R + a];
        }
        store_col<R>(zs + L.g * GEO::T + L.c * R, zz);
      }
      wave_sync();
      if (ok) {
        double out[R], sdummy[D];
        if (ml_mode) {
#pragma unroll
          for (int a = 0; a < R; ++a) out[a] = ml_omega * zz[a];
        } else {
          proj_col<D, R>(ys + L.g * GEO::T, zs + L.g * GEO::T, L.c, zz, out, sdummy);
        }
        const double* rt = rs + L.g * GEO::T + L.c * R;
#pragma unroll
        for (int a = 0; a < R; ++a) part[1] = fma(out[a], rt[a], part[1]);
        store_col<R>(os + L.g * GEO::T + L.c * R, out);
      }
      wave_sync();
      if (z32) {
        float2* zf = reinterpret_cast<float2*>(z32 + base);
#pragma unroll
        for (int it = 0; it < SPN::NIT; ++it) {
          const int pc = lane + 64 * it;
          if (2 * pc < valid) {
            const dbl2 v = reinterpret_cast<const dbl2*>(os)[pc];
            zf[pc] = make_float2((float)v.x, (float)v.y);
          }
        }
      } else {
        dbl2* z2 = reinterpret_cast<dbl2*>(z + base);
#pragma unroll
        for (int it = 0; it < SPN::NIT; ++it) {
          const int pc = lane + 64 * it;
          if (2 * pc < valid) z2[pc] = reinterpret_cast<const dbl2*>(os)[pc];
        }
      }
      wave_sync();
    }
    DPGO_STAMP(1, 5);
    tile += ti_.step;
    have = tile < ti_.last;
    if (have) prefetch(tile);
  }
  if (mode != 1) store_partials<2>(part, pout, red);
  DPGO_STAMP(1, 6);
  DPGO_COMMIT(1);
}

// ================================================================ K7a: tCG residual / iterate update
// ROPTLIB SolversTR::tCG_TR, first half of one inner iteration (and, with first = 1, its
// initialisation r = g, eta = 0, z = P(r)):
//   d_Hd (from k_hess partials) -> alpha, e_Pe';  boundary / negative curvature -> eta += tau*delta, stop
//   else eta += alpha*delta; r += alpha*Hd; z = P(r);  partials: [0] <r,r>  [1] <z,r>
template <int D, int R>
__global__ __launch_bounds__(kBlock, DPGO_LB_UPDATE) void k_tcg_update(const double* __restrict__ X, const double* __restrict__ g,
                                                       const double* __restrict__ dinv,
                                                       const double* __restrict__ delta,
                                                       const double* __restrict__ Hd, double* __restrict__ eta,
                                                       double* __restrict__ r, double* __restrict__ z,
                                                       const double* __restrict__ pin, int nb_in,
                                                       double* __restrict__ pout, const DevState* __restrict__ sin,
                                                       DevState* __restrict__ sout, int first, int n,
                                                       unsigned long long* hflag, unsigned gen, double ml_omega) {
  // generic layout (odd tile size); even tile sizes run k_tcg_update_span
  using GEO = Geo<D, R>;
  __shared__ __attribute__((aligned(16))) double sm[kWaves][3][GEO::G][GEO::T];
  __shared__ double red[kWaves * kNP];
  DevState st;
  load_state(st, sin);
  gen = state_gen(st, gen);
  if (st.rtr_stop || (!first && st.tcg_done)) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      store_state(sout, st);
      publish_progress(hflag, gen, st);
    }
    return;
  }
  double alpha, tau;
  const int mode = tcg_update_prologue(st, pin, nb_in, first, red, alpha, tau);  // 0: step, 1: boundary step, 2: init
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    store_state(sout, st);
    publish_progress(hflag, gen, st);
  }

  const LaneId L = lane_id<D>();
  const int ntiles = (n + GEO::P - 1) / GEO::P;
  double part[2] = {0.0, 0.0};
  const TileIter ti_ = tile_iter(ntiles);
  for (int tile = ti_.first; tile < ti_.last; tile += ti_.step) {
    const int i = tile * GEO::P + L.wave * GEO::G + L.g;
    const bool ok = (L.g < GEO::G) && (i < n);
    const size_t off = (size_t)i * GEO::T + L.c * R;
    if (mode == 1) {  // workgroup-uniform
      if (ok) {
        double e[R], dl[R];
        load_col<R>(eta + off, e);
        load_col<R>(delta + off, dl);
#pragma unroll
        for (int a = 0; a < R; ++a) e[a] = fma(tau, dl[a], e[a]);
        store_col<R>(eta + off, e);
      }
      continue;
    }
    double* ys = ok ? &sm[L.wave][0][L.g][0] : nullptr;
    double* rs = ok ? &sm[L.wave][1][L.g][0] : nullptr;
    double* zs = ok ? &sm[L.wave][2][L.g][0] : nullptr;
    double rr[R], x[R], zz[R], drow[GEO::B];
    if (ok) {
      // all of this pose's loads are issued back to back (independent addresses)
      load_col<R>(X + off, x);
      if (dinv) {
#pragma unroll
        for (int k = 0; k < GEO::B; ++k) drow[k] = dinv[(size_t)i * GEO::BB + L.c * GEO::B + k];
      }
      if (mode == 2) {
        load_col<R>(g + off, rr);
        double e[R];
#pragma unroll
        for (int a = 0; a < R; ++a) e[a] = 0.0;
        store_col<R>(eta + off, e);
      } else {
        double e[R], dl[R], hd[R];
        load_col<R>(eta + off, e);
        load_col<R>(delta + off, dl);
        load_col<R>(Hd + off, hd);
        load_col<R>(r + off, rr);
#pragma unroll
        for (int a = 0; a < R; ++a) {
          e[a] = fma(alpha, dl[a], e[a]);
          rr[a] = fma(alpha, hd[a], rr[a]);
        }
        store_col<R>(eta + off, e);
      }
      store_col<R>(r + off, rr);
#pragma unroll
      for (int a = 0; a < R; ++a) part[0] = fma(rr[a], rr[a], part[0]);
      store_col<R>(ys + L.c * R, x);
      store_col<R>(rs + L.c * R, rr);
    }
    wave_sync();
    if (ok) {
      if (dinv) {
        jacobi_col<D, R>(rs, drow, zz);
      } else {
#pragma unroll
        for (int a = 0; a < R; ++a) zz[a] = rr[a];
      }
      store_col<R>(zs + L.c * R, zz);
    }
    wave_sync();
    if (ok) {
      double out[R], s[D];
      if (ml_omega > 0.0) {
#pragma unroll
        for (int a = 0; a < R; ++a) out[a] = ml_omega * zz[a];
      } else {
        proj_col<D, R>(ys, zs, L.c, zz, out, s);
      }
#pragma unroll
      for (int a = 0; a < R; ++a) part[1] = fma(out[a], rr[a], part[1]);
      store_col<R>(z + off, out);
    }
    wave_sync();
  }
  if (mode != 1) store_partials<2>(part, pout, red);
}
