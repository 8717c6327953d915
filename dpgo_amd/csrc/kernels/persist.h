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

// Tiles a workgroup may own.  Its rows of X, S, Dinv and of every tCG vector stay in LDS for the whole launch, the row
// pointers / preloaded column indices in registers: a phase is then ONE hop of loads (the gathered z tiles and the Q
// blocks, issued together) instead of a chain of dependent ones -- in this regime a kernel is latency, not bytes.
constexpr int kResidentTiles = 3;

template <int D, int R, int SPLIT>
__global__ __launch_bounds__(kBlock) void k_tcg_persist(BsrDev Q, const double* __restrict__ X,
                                                        const double* __restrict__ S, const double* __restrict__ g,
                                                        const double* __restrict__ dinv, double* eta, double* z,
                                                        dbl2* gran, double tagbase, const DevState* __restrict__ sin,
                                                        DevState* __restrict__ sout, PersistCtrl* ctrl, int n,
                                                        unsigned long long* hflag, unsigned gen) {
  using GEO = Geo<D, R, SPLIT>;
  constexpr int P = GEO::P, G = GEO::G, T = GEO::T, B = GEO::B, BB = GEO::BB, MT = kResidentTiles;
  __shared__ __attribute__((aligned(16))) double Xs[MT][P][T], Es[MT][P][T], Rs[MT][P][T], Ds[MT][P][T], Hs[MT][P][T],
      Zs[MT][P][T];
  __shared__ double Ss[MT][P][D * D], Vs[MT][P][BB];
  __shared__ double ex[kWaves][G][T];  // wave-private exchange tile (columns of one pose meet here)
  __shared__ double red[kWaves * kNP];
  __shared__ int ok_s, rank_s, members_s;

  // ---- establish the participants: the workgroups that landed on the target XCD
  if (threadIdx.x == 0) {
    const int xcc = xcc_id();
    // first arrival fixes the target (agent-scope CAS); everybody reads the winner back
    const int prev = atomicCAS(&ctrl->target, -1, xcc);
    const int tgt = (prev == -1) ? xcc : prev;
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
      const int ntiles = (n + P - 1) / P;
      if (!done || members > kPersistMax || (long long)members * MT < ntiles) {  // placement did not hold: give up
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
  if (rank < 0 || members <= 0) return;  // not on the target XCD (or the error flag is set)

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
  const int lp = L.wave * G + L.g;  // pose slot inside a workgroup tile
  const int ntiles = (n + P - 1) / P;

  // ---- resident data of the workgroup's tiles
  RowIdx ri[MT];
  int pose[MT];
  bool okp[MT], own[MT];
#pragma unroll
  for (int k = 0; k < MT; ++k) {
    const int tile = rank + k * members;
    pose[k] = tile * P + lp;
    okp[k] = (tile < ntiles) && (L.g < G) && (pose[k] < n);
    own[k] = okp[k] && (L.s == 0);
    ri[k] = row_idx_load<D, SPLIT>(Q.rowptr, Q.colidx, pose[k], L.s, L.c, okp[k]);
    if (own[k]) {
      const size_t off = (size_t)pose[k] * T + L.c * R;
#pragma unroll
      for (int a = 0; a < R; ++a) {
        Xs[k][lp][L.c * R + a] = X[off + a];
        Rs[k][lp][L.c * R + a] = g[off + a];  // r0 = g
        Es[k][lp][L.c * R + a] = 0.0;         // eta0 = 0
        Ds[k][lp][L.c * R + a] = 0.0;
        Hs[k][lp][L.c * R + a] = 0.0;
      }
      if (L.c < D) {
#pragma unroll
        for (int a = 0; a < D; ++a) Ss[k][lp][L.c * D + a] = S[(size_t)pose[k] * D * D + L.c * D + a];
      }
#pragma unroll
      for (int a = 0; a < B; ++a) Vs[k][lp][L.c * B + a] = dinv ? dinv[(size_t)pose[k] * BB + L.c * B + a] : 0.0;
    }
  }
  wave_sync();

  // ---- phase B: (first) r = g, eta = 0 | eta += alpha delta, r += alpha H delta;  z = proj_X(r Dinv);  partials
  auto phase_update = [&](bool first, double alpha, double (&part)[2]) {
    part[0] = part[1] = 0.0;
#pragma unroll
    for (int k = 0; k < MT; ++k) {
      double rr[R], zz[R];
      if (own[k]) {
#pragma unroll
        for (int a = 0; a < R; ++a) {
          const int e = L.c * R + a;
          double rv = Rs[k][lp][e];
          if (!first) {
            Es[k][lp][e] = fma(alpha, Ds[k][lp][e], Es[k][lp][e]);
            rv = fma(alpha, Hs[k][lp][e], rv);
            Rs[k][lp][e] = rv;
          }
          rr[a] = rv;
          part[0] = fma(rv, rv, part[0]);
        }
      }
      wave_sync();  // the pose's B columns of r are in LDS
      if (own[k]) {
        if (dinv) {
          jacobi_col<D, R>(&Rs[k][lp][0], &Vs[k][lp][L.c * B], zz);
        } else {
#pragma unroll
          for (int a = 0; a < R; ++a) zz[a] = rr[a];
        }
        store_col<R>(&ex[L.wave][L.g][L.c * R], zz);
      }
      wave_sync();
      if (own[k]) {
        double out[R], s[D];
        proj_col<D, R>(&Xs[k][lp][0], &ex[L.wave][L.g][0], L.c, zz, out, s);
        const size_t off = (size_t)pose[k] * T + L.c * R;
#pragma unroll
        for (int a = 0; a < R; ++a) {
          part[1] = fma(out[a], rr[a], part[1]);
          Zs[k][lp][L.c * R + a] = out[a];
          z[off + a] = out[a];  // the copy the other workgroups gather
        }
      }
      wave_sync();
    }
    block_allreduce<2>(part, red);
  };

  // ---- phase A: Hz on the own rows (one hop: Q blocks + gathered z tiles), direction recurrences, <delta, H delta>
  auto phase_hess = [&](bool first, double beta, double (&part)[1]) {
    part[0] = 0.0;
#pragma unroll
    for (int k = 0; k < MT; ++k) {
      if (rank + k * members >= ntiles) break;  // workgroup-uniform
      double h[R];
      spmm_col_pre<D, R, SPLIT, true>(ri[k], Q.colidx, Q.vals, z, L.s, L.c, h);
      if (own[k]) {
        if (L.c < D) {
#pragma unroll
          for (int a = 0; a < D; ++a) {
            const double sac = Ss[k][lp][L.c * D + a];
#pragma unroll
            for (int q = 0; q < R; ++q) h[q] = fma(-Zs[k][lp][a * R + q], sac, h[q]);
          }
        }
        store_col<R>(&ex[L.wave][L.g][L.c * R], h);
      }
      wave_sync();
      if (own[k]) {
        double hz[R], s[D];
        proj_col<D, R>(&Xs[k][lp][0], &ex[L.wave][L.g][0], L.c, h, hz, s);
#pragma unroll
        for (int a = 0; a < R; ++a) {
          const int e = L.c * R + a;
          const double zc = Zs[k][lp][e];
          const double dn = first ? -zc : fma(beta, Ds[k][lp][e], -zc);
          const double hn = first ? -hz[a] : fma(beta, Hs[k][lp][e], -hz[a]);
          Ds[k][lp][e] = dn;
          Hs[k][lp][e] = hn;
          part[0] = fma(dn, hn, part[0]);
        }
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
#pragma unroll
      for (int k = 0; k < MT; ++k) {
        if (own[k]) {
#pragma unroll
          for (int a = 0; a < R; ++a) Es[k][lp][L.c * R + a] = fma(tau, Ds[k][lp][L.c * R + a], Es[k][lp][L.c * R + a]);
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
  // the step eta leaves the launch (the retraction and the model decrease read it); r, delta, H delta die here
#pragma unroll
  for (int k = 0; k < MT; ++k) {
    if (own[k]) {
      const size_t off = (size_t)pose[k] * T + L.c * R;
#pragma unroll
      for (int a = 0; a < R; ++a) eta[off + a] = Es[k][lp][L.c * R + a];
    }
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
