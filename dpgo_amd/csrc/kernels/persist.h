// kernels/persist.h -- persistent single-XCD tCG kernel for small blocks (the latency regime of multi-GPU strong scaling).
// Part of kernels.h (included inside namespace dpgo, in this order: common.h, problem.h, tcg.h, persist.h, multilevel.h, dense.h, manifold.h, rtr.h, agent.h, init.h).
#pragma once

// ================================================================ one launch per tCG run
// Below ~6k poses a tCG iteration of the two-kernel scheme (k_tcg_hess | k_tcg_update) is half kernel boundary
// (~4 us from the last instruction of one kernel to the first of the next, twice per iteration).
// This kernel runs ROPTLIB's whole tCG_TR loop (same arithmetic, same scalar recurrences as the
// two-kernel scheme) in ONE launch, with in-kernel barriers between the phases of an iteration:
//   phase A: Hz = proj_X(z Q - z_rot S) on the workgroup's own rows (the gather reads the neighbours' z),
//            delta <- beta delta - z,  H delta <- beta H delta - Hz,  partial <delta, H delta>          | barrier
//   phase B: alpha / boundary test;  eta += alpha delta,  r += alpha H delta,  z = proj_X(r Dinv),
//            partials <r,r>, <z,r>                                                                       | barrier
// What makes the barriers cheap is that every participant sits on ONE XCD: the XCD's L2 is then the coherence point,
// so a barrier is one 16-byte store and a poll of the other participants' stores -- no atomics, no cache write-back /
// invalidate (which is what the 4-7 us of a chip-wide barrier are made of, MI355X_MICROARCH.md price list) -- and it
// carries the partial sums of the step's dot products with it (xcd_allreduce), provided that
//   * a producer's stores have reached the L2 before it arrives        (s_waitcnt vmcnt(0) + workgroup barrier),
//   * consumers read other workgroups' data with L1-bypassing loads    (nontemporal / agent-scope atomic loads).
// Placement is undefined by HIP, so it is ESTABLISHED at run time, not assumed: the launch has 8x the wanted workgroups,
// each reads its XCC id; the first arrival fixes the target XCD, workgroups elsewhere leave at once, and the ones on
// the target wait until every launched workgroup has reported before they count themselves.  Every spin is bounded;
// a time-out raises PersistCtrl::error and the host reruns the outer iteration with the two-kernel scheme.
struct PersistCtrl {          // zeroed (target = -1) by the host before every launch
  unsigned long long counts;  // [31:0] workgroups that have started, [63:32] of them on the target XCD (one atomic)
  int target;                 // XCC id of the participants
  unsigned bar;               // unused (kept for layout)
  int error;                  // a spin ran out: results invalid
  unsigned iters;             // diagnostic: tCG iterations executed
  unsigned members;           // diagnostic: participants
  unsigned pad;
  // diagnostic timeline of participant 0 (100 MHz wall clock ticks, summed over the iterations after the first):
  // [0] phase A (Hessian step)  [1] all-reduce after A  [2] phase B (update)  [3] all-reduce after B  [4] iterations
  unsigned long long ticks[8];
};

constexpr unsigned kSpinLimit = 1u << 21;  // polls (with s_sleep) before a spin gives up: ~0.5 s

__device__ __forceinline__ int xcc_id() {
  // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, size 4)
  return (int)(__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 0xf);
}

__device__ __forceinline__ bool spin_until_ge(unsigned* p, unsigned want, int* error) {
  for (unsigned it = 0; it < kSpinLimit; ++it) {
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
    if ((it & 63u) == 63u && __hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
    __builtin_amdgcn_s_sleep(1);
  }
  __hip_atomic_store(error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return false;
}

// Every vector the launch itself writes (eta, r, z, delta, H delta) is read with L1-bypassing loads: the two phases
// map poses to workgroups differently, so "own rows" of one phase may have been written by another workgroup.
template <int R>
__device__ __forceinline__ void load_col_nt(const double* p, double (&v)[R]) {
#pragma unroll
  for (int a = 0; a < R; ++a) v[a] = __builtin_nontemporal_load(p + a);
}

// Barrier + all-reduce of the participants in one step, without atomics: every workgroup publishes its K partial sums
// as K granules {tag, value} -- one naturally aligned 16-byte store each, so tag and value arrive together
// (MI355X_MICROARCH.md, hand-off granules) -- and thread t of every workgroup polls participant t's granules with
// 16-byte L1-bypassing loads until the tag of THIS step shows up.  The sums are then formed in the same fixed order by
// everybody.  tag = tagbase + epoch is unique per launch and step (no clearing between launches); two buffers
// alternate, which suffices because nobody can be more than one step ahead of the slowest participant.
constexpr int kPersistMax = kBlock;  // participants <= threads of a workgroup (one poller per participant)
template <int K>
__device__ __forceinline__ bool xcd_allreduce(dbl2* gran, int rank, int members, double tagbase, unsigned& epoch,
                                              double (&val)[K], double* red, int* error, int* ok_s) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's vector stores have been acknowledged by the L2
  if (threadIdx.x == 0) *ok_s = 1;
  __syncthreads();
  epoch += 1;
  const double tag = tagbase + (double)epoch;
  dbl2* buf = gran + (size_t)(epoch & 1u) * kPersistMax * 2;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if ((int)threadIdx.x == k) {  // block_allreduce left the workgroup's sums in every thread
      dbl2 gv;
      gv.x = tag;
      gv.y = val[k];
      buf[rank * 2 + k] = gv;
    }
  }
  double v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = 0.0;
  if ((int)threadIdx.x < members) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      bool got = false;
      for (unsigned it = 0; it < kSpinLimit; ++it) {
        const dbl2 gv = __builtin_nontemporal_load(buf + threadIdx.x * 2 + k);
        if (gv.x == tag) {
          v[k] = gv.y;
          got = true;
          break;
        }
        if ((it & 255u) == 255u && __hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (!got) {
        __hip_atomic_store(error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *ok_s = 0;
      }
    }
  }
  block_allreduce<K>(v, red);  // two workgroup barriers inside: ok_s is settled afterwards
#pragma unroll
  for (int k = 0; k < K; ++k) val[k] = v[k];
  return *ok_s != 0;
}

template <int D, int R, int SPLIT>
__global__ __launch_bounds__(kBlock) void k_tcg_persist(BsrDev Q, const double* __restrict__ X,
                                                        const double* __restrict__ S, const double* __restrict__ g,
                                                        const double* __restrict__ dinv, double* delta, double* Hd,
                                                        double* eta, double* r, double* z, dbl2* gran, double tagbase,
                                                        const DevState* __restrict__ sin, DevState* __restrict__ sout,
                                                        PersistCtrl* ctrl, int n, unsigned long long* hflag,
                                                        unsigned gen) {
  using GEO = Geo<D, R, SPLIT>;   // SpMM phase: SPLIT lane groups per pose
  using GEU = Geo<D, R, 1>;       // update phase: one lane per (pose, column)
  __shared__ __attribute__((aligned(16))) double sm[kWaves][4][GEU::G][GEU::T];
  __shared__ double red[kWaves * kNP];
  __shared__ int ok_s, rank_s, members_s;

  // ---- establish the participants: the workgroups that landed on the target XCD
  if (threadIdx.x == 0) {
    const int xcc = xcc_id();
    int tgt = -1;
    // first arrival fixes the target (agent-scope CAS); everybody reads the winner back
    const int prev = atomicCAS(&ctrl->target, -1, xcc);
    tgt = (prev == -1) ? xcc : prev;
    // one atomic reports the arrival and, on the target XCD, takes a rank
    const bool mine = (xcc == tgt);
    const unsigned long long old = atomicAdd(&ctrl->counts, 1ull | (mine ? (1ull << 32) : 0ull));
    int rank = mine ? (int)(old >> 32) : -1;
    int members = 0;
    if (rank >= 0) {
      // every launched workgroup has reported => the member count is final
      bool done = false;
      for (unsigned it = 0; it < kSpinLimit && !done; ++it) {
        const unsigned long long v = __hip_atomic_load(&ctrl->counts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(v & 0xffffffffull) >= gridDim.x) {
          members = (int)(v >> 32);
          done = true;
        } else {
          __builtin_amdgcn_s_sleep(1);
        }
      }
      if (!done) {
        __hip_atomic_store(&ctrl->error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        rank = -1;
      }
      if (rank == 0) ctrl->members = (unsigned)members;
    }
    rank_s = rank;
    members_s = members;
  }
  __syncthreads();
  const int rank = rank_s, members = members_s;
  if (rank < 0 || members <= 0) return;  // not on the target XCD (or time-out: error flag is set)

  DevState st;
  load_state(st, sin);
  if (st.rtr_stop) {
    if (rank == 0 && threadIdx.x == 0) {
      store_state(sout, st);
      publish_progress(hflag, gen, st);
    }
    return;
  }
  unsigned epoch = 0;
  const LaneId L = lane_id<D, SPLIT>();
  const LaneId U = lane_id<D, 1>();
  const int ntiles_s = (n + GEO::P - 1) / GEO::P;
  const int ntiles_u = (n + GEU::P - 1) / GEU::P;

  // ---- phase B: (first) r = g, eta = 0 | eta += alpha delta, r += alpha H delta;  z = proj_X(r Dinv);  partials
  auto phase_update = [&](bool first, double alpha, double (&part)[2]) {
    part[0] = part[1] = 0.0;
    for (int tile = rank; tile < ntiles_u; tile += members) {
      const int i = tile * GEU::P + U.wave * GEU::G + U.g;
      const bool ok = (U.g < GEU::G) && (i < n);
      const size_t off = (size_t)i * GEU::T + U.c * R;
      double* ys = ok ? &sm[U.wave][0][U.g][0] : nullptr;
      double* rs = ok ? &sm[U.wave][1][U.g][0] : nullptr;
      double* zs = ok ? &sm[U.wave][2][U.g][0] : nullptr;
      double rr[R], x[R], zz[R], drow[GEU::B];
      if (ok) {
        load_col<R>(X + off, x);
        if (dinv) {
#pragma unroll
          for (int k = 0; k < GEU::B; ++k) drow[k] = dinv[(size_t)i * GEU::BB + U.c * GEU::B + k];
        }
        if (first) {
          load_col<R>(g + off, rr);
          double e[R];
#pragma unroll
          for (int a = 0; a < R; ++a) e[a] = 0.0;
          store_col<R>(eta + off, e);
        } else {
          double e[R], dl[R], hd[R];
          load_col_nt<R>(eta + off, e);
          load_col_nt<R>(delta + off, dl);
          load_col_nt<R>(Hd + off, hd);
          load_col_nt<R>(r + off, rr);
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
        store_col<R>(ys + U.c * R, x);
        store_col<R>(rs + U.c * R, rr);
      }
      wave_sync();
      if (ok) {
        if (dinv) {
          jacobi_col<D, R>(rs, drow, zz);
        } else {
#pragma unroll
          for (int a = 0; a < R; ++a) zz[a] = rr[a];
        }
        store_col<R>(zs + U.c * R, zz);
      }
      wave_sync();
      if (ok) {
        double out[R], s[D];
        proj_col<D, R>(ys, zs, U.c, zz, out, s);
#pragma unroll
        for (int a = 0; a < R; ++a) part[1] = fma(out[a], rr[a], part[1]);
        store_col<R>(z + off, out);
      }
      wave_sync();
    }
    block_allreduce<2>(part, red);
  };

  // ---- phase A: Hz on the own rows (gather of z bypasses L1), direction recurrences, partial <delta, H delta>
  auto phase_hess = [&](bool first, double beta, double (&part)[1]) {
    part[0] = 0.0;
    for (int tile = rank; tile < ntiles_s; tile += members) {
      const int i = tile * GEO::P + L.wave * GEO::G + L.g;
      const bool okp = (L.g < GEO::G) && (i < n);
      const bool ok = okp && (L.s == 0);
      double h[R], zc[R], x[R];
      const size_t off = (size_t)i * GEO::T + L.c * R;
      double* ys = ok ? &sm[L.wave][0][L.g][0] : nullptr;
      double* vs = ok ? &sm[L.wave][1][L.g][0] : nullptr;
      double* hs = ok ? &sm[L.wave][2][L.g][0] : nullptr;
      double srow[D], dl[R], hd[R];
      if (ok) {
        load_col<R>(X + off, x);
        load_col_nt<R>(z + off, zc);
        if (L.c < D) {
#pragma unroll
          for (int a = 0; a < D; ++a) srow[a] = S[(size_t)i * D * D + L.c * D + a];
        }
        if (!first) {
          load_col_nt<R>(delta + off, dl);
          load_col_nt<R>(Hd + off, hd);
        }
      }
      spmm_col<D, R, SPLIT, true>(Q.rowptr, Q.colidx, Q.vals, z, i, L.s, L.c, okp, h);
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
    block_allreduce<1>(part, red);
  };

  // ---- tCG_TR (ROPTLIB): the scalar logic of tcg_update_prologue / tcg_hess_prologue, evaluated redundantly (and
  // identically: same partials, same summation order) by every participant
  st.tcg_done = 0;
  st.tcg_j = 0;
  st.tcg_status = TCG_MAXITER;
  st.e_Pe = 0.0;
  st.e_Pd = 0.0;
  bool alive = true;
  double pr[2];
  phase_update(true, 0.0, pr);
  alive = xcd_allreduce<2>(gran, rank, members, tagbase, epoch, pr, red, &ctrl->error, &ok_s);
  if (alive) {
    st.norm_r0 = sqrt(pr[0]);
    st.z_r = pr[1];
    st.d_Pd = pr[1];
    st.e_Pd = 0.0;
    if (st.max_inner <= 0) st.tcg_done = 1;
  }
  double beta = 0.0;
  bool first = true;
  unsigned iters = 0;
  unsigned long long tk[5] = {0, 0, 0, 0, 0};
  while (alive && !st.tcg_done) {
    const unsigned long long t0 = wall_clock64();
    double dh[1];
    phase_hess(first, beta, dh);
    const unsigned long long t1 = wall_clock64();
    if (!(alive = xcd_allreduce<1>(gran, rank, members, tagbase, epoch, dh, red, &ctrl->error, &ok_s))) break;
    const unsigned long long t2 = wall_clock64();
    const double d_Hd = dh[0];
    const double alpha = st.z_r / d_Hd;
    const double e_Pe_new = st.e_Pe + 2.0 * alpha * st.e_Pd + alpha * alpha * st.d_Pd;
    st.n_hess += 1;
    st.alpha = alpha;
    iters += 1;
    const double D2 = st.Delta * st.Delta;
    if (d_Hd <= 0.0 || e_Pe_new >= D2) {  // negative curvature / trust-region boundary: eta += tau delta, stop
      const double tau = (-st.e_Pd + sqrt(st.e_Pd * st.e_Pd + st.d_Pd * (D2 - st.e_Pe))) / st.d_Pd;
      st.tcg_status = (d_Hd < 0.0) ? TCG_NEGCURV : TCG_EXCREGION;
      st.tcg_done = 1;
      for (int tile = rank; tile < ntiles_u; tile += members) {
        const int i = tile * GEU::P + U.wave * GEU::G + U.g;
        if ((U.g < GEU::G) && (i < n)) {
          const size_t off = (size_t)i * GEU::T + U.c * R;
          double e[R], dl[R];
          load_col_nt<R>(eta + off, e);
          load_col_nt<R>(delta + off, dl);
#pragma unroll
          for (int a = 0; a < R; ++a) e[a] = fma(tau, dl[a], e[a]);
          store_col<R>(eta + off, e);
        }
      }
      break;
    }
    st.e_Pe = e_Pe_new;
    phase_update(false, alpha, pr);
    const unsigned long long t3 = wall_clock64();
    if (!(alive = xcd_allreduce<2>(gran, rank, members, tagbase, epoch, pr, red, &ctrl->error, &ok_s))) break;
    const unsigned long long t4 = wall_clock64();
    if (!first) {
      tk[0] += t1 - t0;
      tk[1] += t2 - t1;
      tk[2] += t3 - t2;
      tk[3] += t4 - t3;
      tk[4] += 1;
    }
    const double norm_r = sqrt(pr[0]), z_r_new = pr[1];
    const double pw = (st.theta == 1.0) ? st.norm_r0 : pow(st.norm_r0, st.theta);
    if (st.tcg_j >= st.min_inner && norm_r <= st.norm_r0 * (pw < st.kappa ? pw : st.kappa)) {
      st.tcg_status = (st.kappa < pw) ? TCG_LCON : TCG_SCON;
      st.tcg_done = 1;
      break;
    }
    beta = z_r_new / st.z_r;
    st.e_Pd = beta * (st.e_Pd + st.alpha * st.d_Pd);
    st.d_Pd = z_r_new + beta * beta * st.d_Pd;
    st.z_r = z_r_new;
    st.tcg_j += 1;
    if (st.tcg_j >= st.max_inner) {
      st.tcg_done = 1;
      st.tcg_status = TCG_MAXITER;
    }
    first = false;
  }
  if (rank == 0 && threadIdx.x == 0) {
    if (alive) {
      store_state(sout, st);
      publish_progress(hflag, gen, st);
    }
    ctrl->iters = iters;
#pragma unroll
    for (int q = 0; q < 5; ++q) ctrl->ticks[q] = tk[q];
  }
}
