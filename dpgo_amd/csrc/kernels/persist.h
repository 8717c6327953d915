// kernels/persist.h -- persistent whole-chip tCG kernel for blocks in the latency regime (what a GPU runs when a graph is
// cut over many agents / GPUs: <= ~65k poses).
// Part of kernels.h (included inside namespace dpgo, in this order: common.h, problem.h, tcg.h, persist.h, multilevel.h, dense.h, manifold.h, rtr.h, agent.h, init.h).
#pragma once

// ================================================================ one launch per tCG run
// Below a few ten thousand poses a tCG iteration of the two-kernel scheme (k_tcg_hess | k_tcg_update) is made of kernel
// boundaries and prologues, not of bytes (2 500 poses 14.4 us, 12 500 poses 20.8 us per iteration
// whatever they compute).  This kernel runs ROPTLIB's whole tCG_TR loop of one outer iteration (same
// arithmetic, same scalar recurrences as the two-kernel scheme) in ONE launch on up to 256 workgroups:
//   phase A: Hz = proj_X(z Q - z_rot S) on the workgroup's own rows (the gather reads the neighbours' z),
//            delta <- beta delta - z,  H delta <- beta H delta - Hz,  partial <delta, H delta>          | all-reduce
//   phase B: alpha / boundary test;  eta += alpha delta,  r += alpha H delta,  z = proj_X(r Dinv),
//            partials <r,r>, <z,r>                                                                       | all-reduce
// * Every tCG vector of the workgroup's rows (r, eta, delta, H delta, z, and X, S, Dinv, the row pointers and preloaded
//   column indices) lives in REGISTERS for the whole launch -- one lane = one column of one pose, as everywhere; only
//   the columns of a pose meet through a wave-private LDS tile.  The single vector that crosses workgroups is z.
// * Placement-independent hand-off (MI355X_MICROARCH.md, "Workgroup dispatch ... visibility"; cdna_hip_programming.md
//   Guideline 16): the 8 XCDs' L2s are not coherent and HIP promises nothing about where a workgroup runs, so z is stored
//   WRITE-THROUGH (agent-scope relaxed atomic stores = sc1) and gathered with agent-scope loads (sc1: never served by
//   this CU's L1 or a stale L2 line), every storing wave drains its stores (s_waitcnt vmcnt(0)) before the workgroup
//   publishes, and the publish IS the all-reduce: each workgroup stores its K partial sums as 8-byte granules
//   {epoch, 32-bit half} (one atomic store each, so tag and payload arrive together), ONE wave per workgroup sweeps all
//   participants' granules until every tag carries this step's epoch, and everybody forms the sums in the same fixed
//   order -- so every workgroup takes the same data-dependent decisions (negative curvature, boundary, kappa/theta stop).
//   Seeing a participant's step-e granules implies its z stores of step e have reached memory.
// * Every participant must be resident: the host sizes the grid to what the chip holds at once (and reserves those
//   slots process-wide, so that concurrently solved agents never wait for each other's workgroups); every spin is
//   bounded, a time-out raises PersistCtrl::error and the host reruns the outer iteration with the two-kernel scheme.
struct PersistCtrl {
  int error;       // a spin ran out: results invalid
  unsigned iters;  // diagnostic: tCG iterations executed
  unsigned members;
  unsigned pad;
  // diagnostic timeline of participant 0 (100 MHz wall clock ticks, summed over the iterations after the first):
  // [0] phase A (Hessian step)  [1] all-reduce after A  [2] phase B (update)  [3] all-reduce after B  [4] iterations
  unsigned long long ticks[8];
};

constexpr unsigned kSpinLimit = 1u << 21;  // polls (with s_sleep) before a spin gives up: ~0.5 s
constexpr int kPersistMax = kBlock;        // participants (workgroups) of one launch
constexpr int kGranVals = 2;               // partial sums per all-reduce (max)
constexpr int kGranRows = 2 * kGranVals;   // 8-byte words per participant: {epoch, low half}, {epoch, high half} per value
// granule table: [2 buffers][kGranRows][kPersistMax] words -- a row is contiguous over the participants, so a sweep is
// kGranRows coalesced loads per 64 participants
constexpr size_t kGranWords = (size_t)2 * kGranRows * kPersistMax;

__device__ __forceinline__ double ld_agent(const double* p) {
  // agent-scope relaxed load (global_load_dwordx2 sc1): coherent with other workgroups' write-through stores
  const unsigned long long b =
      __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __longlong_as_double((long long)b);
}
__device__ __forceinline__ void st_agent(double* p, double v) {
  // agent-scope relaxed store (global_store_dwordx2 sc1): write-through, visible to every XCD once acknowledged
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

// Barrier + all-reduce of the launch's workgroups in one step.  PRECONDITION: every thread has executed
// `s_waitcnt vmcnt(0)` after its last store that other workgroups read, and a workgroup barrier followed (the
// block_allreduce that produced `val` provides it; `val` is identical in all threads of the workgroup).
template <int K>
__device__ __forceinline__ bool chip_allreduce(unsigned long long* gran, int rank, int members, unsigned salt, unsigned& step,
                                               double (&val)[K], double* red, int* error, int* ok_s) {
  static_assert(K <= kGranVals, "granule rows");
  step += 1;
  const unsigned long long epoch = (unsigned long long)(salt | step);  // never 0; unique per launch and step
  unsigned long long* buf = gran + (size_t)(step & 1u) * kGranRows * kPersistMax;
  if (threadIdx.x == 0) *ok_s = 1;
  if ((int)threadIdx.x < 2 * K) {
    const int k = threadIdx.x >> 1, half = threadIdx.x & 1;
    const unsigned long long bits = (unsigned long long)__double_as_longlong(val[k]);
    const unsigned long long w = (epoch << 32) | (half ? (bits >> 32) : (bits & 0xffffffffull));
    __hip_atomic_store(buf + (size_t)threadIdx.x * kPersistMax + rank, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();  // ok_s initialised before a poller may clear it
  if (threadIdx.x < 64) {  // ONE wave sweeps: lane l takes participants l, l + 64, l + 128, l + 192
    double acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0.0;
    bool fine = true;
#pragma unroll
    for (int q = 0; q < kPersistMax / 64; ++q) {
      const int t = (int)threadIdx.x + 64 * q;
      if (q * 64 >= members) break;  // wave-uniform
      if (t < members) {
        bool got = false;
        for (unsigned it = 0; it < kSpinLimit; ++it) {
          unsigned long long w[2 * K];
#pragma unroll
          for (int j = 0; j < 2 * K; ++j)
            w[j] = __hip_atomic_load(buf + (size_t)j * kPersistMax + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          bool all = true;
#pragma unroll
          for (int j = 0; j < 2 * K; ++j) all = all && ((w[j] >> 32) == epoch);
          if (all) {
#pragma unroll
            for (int k = 0; k < K; ++k)
              acc[k] += __longlong_as_double((long long)((w[2 * k] & 0xffffffffull) | (w[2 * k + 1] << 32)));
            got = true;
            break;
          }
          if ((it & 255u) == 255u && __hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
          __builtin_amdgcn_s_sleep(1);
        }
        fine = fine && got;
      }
    }
    if (!fine) {
      __hip_atomic_store(error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *ok_s = 0;
    }
    // lane l holds the sum over its participants (ascending); fixed DPP tree over the lanes: the same bits everywhere
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = wave_reduce_lane63(acc[k]);
    if (threadIdx.x == 63) {
#pragma unroll
      for (int k = 0; k < K; ++k) red[k] = acc[k];
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) val[k] = red[k];
  const bool ok = *ok_s != 0;
  __syncthreads();  // red / ok_s may be rewritten by the next reduction
  return ok;
}

// MT = tiles (of Geo::P poses) a workgroup owns: tile = rank + k * members, k < MT.
template <int D, int R, int SPLIT, int MT>
__global__ __launch_bounds__(kBlock) void k_tcg_persist(BsrDev Q, const double* __restrict__ X,
                                                        const double* __restrict__ S, const double* __restrict__ g,
                                                        const double* __restrict__ dinv, double* __restrict__ eta, double* z,
                                                        unsigned long long* gran, unsigned salt,
                                                        const DevState* __restrict__ sin, DevState* __restrict__ sout,
                                                        PersistCtrl* ctrl, int n, unsigned long long* hflag, unsigned gen) {
  using GEO = Geo<D, R, SPLIT>;
  constexpr int P = GEO::P, G = GEO::G, T = GEO::T, B = GEO::B, BB = GEO::BB;
  // resident in LDS: the poses' X (projections need all rotation columns of a pose) and z (Hessian correction);
  // ex: two wave-private exchange tiles (the columns of one pose meet here)
  __shared__ __attribute__((aligned(16))) double Xs[MT][P][T], Zs[MT][P][T];
  __shared__ __attribute__((aligned(16))) double ex[2][kWaves][G][T];
  __shared__ double red[kWaves * kNP];
  __shared__ int ok_s;

  const int rank = blockIdx.x, members = gridDim.x;
  DevState st;
  load_state(st, sin);
  if (st.rtr_stop) {
    if (rank == 0 && threadIdx.x == 0) {
      store_state(sout, st);
      publish_progress(hflag, gen, st);
    }
    return;
  }
  unsigned step = 0;
  const LaneId L = lane_id<D, SPLIT>();
  const int lp = L.wave * G + L.g;  // pose slot inside a workgroup tile
  const int ntiles = (n + P - 1) / P;
  const int co = L.c * R;

  // ---- resident data of the workgroup's rows (registers; X and z also in LDS)
  RowIdx ri[MT];
  int pose[MT];
  bool okp[MT], own[MT];
  double rr[MT][R], ee[MT][R], dl[MT][R], hd[MT][R], zc[MT][R], srow[MT][D], drow[MT][B];
#pragma unroll
  for (int k = 0; k < MT; ++k) {
    const int tile = rank + k * members;
    pose[k] = tile * P + lp;
    okp[k] = (tile < ntiles) && (L.g < G) && (pose[k] < n);
    own[k] = okp[k] && (L.s == 0);
    ri[k] = row_idx_load<D, SPLIT>(Q.rowptr, Q.colidx, pose[k], L.s, L.c, okp[k]);
#pragma unroll
    for (int a = 0; a < R; ++a) rr[k][a] = ee[k][a] = dl[k][a] = hd[k][a] = zc[k][a] = 0.0;
#pragma unroll
    for (int a = 0; a < D; ++a) srow[k][a] = 0.0;
#pragma unroll
    for (int a = 0; a < B; ++a) drow[k][a] = 0.0;
    if (own[k]) {
      const size_t off = (size_t)pose[k] * T + co;
#pragma unroll
      for (int a = 0; a < R; ++a) {
        Xs[k][lp][co + a] = X[off + a];
        rr[k][a] = g[off + a];  // r0 = g
      }
      if (L.c < D) {
#pragma unroll
        for (int a = 0; a < D; ++a) srow[k][a] = S[(size_t)pose[k] * D * D + L.c * D + a];
      }
      if (dinv) {
#pragma unroll
        for (int a = 0; a < B; ++a) drow[k][a] = dinv[(size_t)pose[k] * BB + L.c * B + a];
      }
    }
  }
  wave_sync();

  // ---- phase B: (first) r = g, eta = 0 | eta += alpha delta, r += alpha H delta;  z = proj_X(r Dinv);  partials
  auto phase_update = [&](bool first, double alpha, double (&part)[2]) {
    part[0] = part[1] = 0.0;
#pragma unroll
    for (int k = 0; k < MT; ++k) {
      if (rank + k * members >= ntiles) break;  // workgroup-uniform
      double zz[R];
      if (own[k]) {
#pragma unroll
        for (int a = 0; a < R; ++a) {
          if (!first) {
            ee[k][a] = fma(alpha, dl[k][a], ee[k][a]);
            rr[k][a] = fma(alpha, hd[k][a], rr[k][a]);
          }
          part[0] = fma(rr[k][a], rr[k][a], part[0]);
        }
        if (dinv) store_col<R>(&ex[0][L.wave][L.g][co], rr[k]);
      }
      if (dinv) {
        wave_sync();  // the pose's B columns of r are in LDS
        if (own[k]) jacobi_col<D, R>(&ex[0][L.wave][L.g][0], drow[k], zz);
      } else {
#pragma unroll
        for (int a = 0; a < R; ++a) zz[a] = rr[k][a];
      }
      if (own[k]) store_col<R>(&ex[1][L.wave][L.g][co], zz);
      wave_sync();
      if (own[k]) {
        double out[R], s[D];
        proj_col<D, R>(&Xs[k][lp][0], &ex[1][L.wave][L.g][0], L.c, zz, out, s);
        const size_t off = (size_t)pose[k] * T + co;
#pragma unroll
        for (int a = 0; a < R; ++a) {
          part[1] = fma(out[a], rr[k][a], part[1]);
          zc[k][a] = out[a];
          Zs[k][lp][co + a] = out[a];
          st_agent(z + off + a, out[a]);  // the copy the other workgroups gather
        }
      }
      // ex[0] / ex[1] of the next tile are written only after this tile's reads: with block-Jacobi the next tile's first
      // wave_sync stands between them; without a preconditioner this one does
      if (!dinv) wave_sync();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's z stores have been acknowledged
    block_allreduce<2>(part, red);
  };

  // ---- phase A: Hz on the own rows (one hop: Q blocks + gathered z tiles of ALL owned tiles are requested before the
  // first epilogue), direction recurrences, <delta, H delta>
  auto phase_hess = [&](bool first, double beta, double (&part)[1]) {
    part[0] = 0.0;
    double h[MT][R];
#pragma unroll
    for (int k = 0; k < MT; ++k) {
      if (rank + k * members >= ntiles) break;  // workgroup-uniform
      spmm_col_pre<D, R, SPLIT, 2>(ri[k], Q.colidx, Q.vals, z, L.s, L.c, h[k]);
    }
#pragma unroll
    for (int k = 0; k < MT; ++k) {
      if (rank + k * members >= ntiles) break;
      double* xt = &ex[k & 1][L.wave][L.g][0];
      if (own[k]) {
        if (L.c < D) {
#pragma unroll
          for (int a = 0; a < D; ++a) {
#pragma unroll
            for (int q = 0; q < R; ++q) h[k][q] = fma(-Zs[k][lp][a * R + q], srow[k][a], h[k][q]);
          }
        }
        store_col<R>(xt + co, h[k]);
      }
      wave_sync();
      if (own[k]) {
        double hz[R], s[D];
        proj_col<D, R>(&Xs[k][lp][0], xt, L.c, h[k], hz, s);
#pragma unroll
        for (int a = 0; a < R; ++a) {
          const double dn = first ? -zc[k][a] : fma(beta, dl[k][a], -zc[k][a]);
          const double hn = first ? -hz[a] : fma(beta, hd[k][a], -hz[a]);
          dl[k][a] = dn;
          hd[k][a] = hn;
          part[0] = fma(dn, hn, part[0]);
        }
      }
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
  alive = chip_allreduce<2>(gran, rank, members, salt, step, pr, red, &ctrl->error, &ok_s);
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
    if (!(alive = chip_allreduce<1>(gran, rank, members, salt, step, dh, red, &ctrl->error, &ok_s))) break;
    const unsigned long long t2 = wall_clock64();
    const double d_Hd = dh[0];
    const double alpha = st.z_r / d_Hd;
    const double e_Pe_new = st.e_Pe + 2.0 * alpha * st.e_Pd + alpha * alpha * st.d_Pd;
    st.n_hess += 1;
    st.alpha = alpha;
    st.d_Hd = d_Hd;
    iters += 1;
    const double D2 = st.Delta * st.Delta;
    if (d_Hd <= 0.0 || e_Pe_new >= D2) {  // negative curvature / trust-region boundary: eta += tau delta, stop
      const double tau = (-st.e_Pd + sqrt(st.e_Pd * st.e_Pd + st.d_Pd * (D2 - st.e_Pe))) / st.d_Pd;
      st.tcg_status = (d_Hd < 0.0) ? TCG_NEGCURV : TCG_EXCREGION;
      st.tcg_done = 1;
#pragma unroll
      for (int k = 0; k < MT; ++k) {
#pragma unroll
        for (int a = 0; a < R; ++a) ee[k][a] = fma(tau, dl[k][a], ee[k][a]);
      }
      break;
    }
    st.e_Pe = e_Pe_new;
    phase_update(false, alpha, pr);
    const unsigned long long t3 = wall_clock64();
    if (!(alive = chip_allreduce<2>(gran, rank, members, salt, step, pr, red, &ctrl->error, &ok_s))) break;
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
      const size_t off = (size_t)pose[k] * T + co;
#pragma unroll
      for (int a = 0; a < R; ++a) eta[off + a] = ee[k][a];
    }
  }
  if (rank == 0 && threadIdx.x == 0) {
    // a participant that passed every all-reduce of the run saw everybody's granules of the last step, so nobody can
    // still fail: the error flag is final here
    if (alive && !__hip_atomic_load(&ctrl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
      store_state(sout, st);
      publish_progress(hflag, gen, st);
    } else {
      // time-out: hand the UNCHANGED state on, poisoned, so that every kernel enqueued behind this launch exits and the
      // host resumes from this state with the two-kernel scheme (kPersistPoison in dpgo_hip.hip)
      DevState s0;
      load_state(s0, sin);
      s0.rtr_stop = 3;
      store_state(sout, s0);
      publish_progress(hflag, gen, s0);
    }
    ctrl->iters = iters;
    ctrl->members = (unsigned)members;
#pragma unroll
    for (int q = 0; q < 5; ++q) ctrl->ticks[q] = tk[q];
  }
}
