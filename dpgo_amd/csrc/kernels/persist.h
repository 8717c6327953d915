// kernels/persist.h -- persistent whole-chip solve kernel for blocks in the latency regime (what a GPU runs when a graph is
// cut over many agents / GPUs: <= ~32k poses): QuadraticOptimizer::optimize in ONE launch (k_rtr_persist).
// Part of kernels.h (included inside namespace dpgo, in this order: common.h, problem.h, tcg.h, persist.h, multilevel.h, dense.h, manifold.h, rtr.h, agent.h, init.h).
#pragma once

// ================================================================ one launch per solve
// Below a few ten thousand poses a tCG iteration of the multi-launch scheme (k_tcg_hess | k_tcg_update) is made of kernel
// boundaries and prologues, not of bytes (2 500 poses 14.4 us, 12 500 poses 20.8 us per iteration
// whatever they compute).  k_rtr_persist (below) runs the whole solve -- ROPTLIB's SolversTR::Run with its tCG_TR loops
// (same arithmetic, same scalar recurrences as the multi-launch scheme) -- in ONE launch on up to 256
// workgroups.  A tCG iteration inside it:
//   phase A: Hz = proj_X(z Q - z_rot S) on the workgroup's own rows (the gather reads the neighbours' z),
//            delta <- beta delta - z,  H delta <- beta H delta - Hz,  partial <delta, H delta>          | all-reduce
//   phase B: alpha / boundary test;  eta += alpha delta,  r += alpha H delta,  z = proj_X(r Dinv),
//            partials <r,r>, <z,r>                                                                       | all-reduce
//   (the additive two-level preconditioner, ADD below, keeps this shape since round 6: the restriction of H delta rides
//    on the first reduction as a payload, the coarse solve and the prolongation sit in phase B)
// * The gather of phase A takes the poses of the workgroup's OWN tiles from the LDS tiles phase B wrote (layouts with
//   64-pose tiles, persist_local) and crosses the chip only for the others.
// * Every tCG vector of the workgroup's rows (r, eta, delta, H delta, z, and S, Dinv, the row's column indices and block
//   columns of Q) lives in REGISTERS for the whole launch -- one lane = one column of one pose, as everywhere; the iterate,
//   the trial point and the two gradients are LDS tiles; only the columns of a pose meet through a wave-private LDS tile.
//   The vectors that cross workgroups: z per tCG iteration, the trial point and the step once per outer iteration.
// * Placement-independent hand-off (MI355X_MICROARCH.md, "Workgroup dispatch ... visibility"; cdna_hip_programming.md
//   Guideline 16): the 8 XCDs' L2s are not coherent and HIP promises nothing about where a workgroup runs, so z is stored
//   WRITE-THROUGH (agent-scope relaxed atomic stores = sc1) and gathered with agent-scope loads (sc1: never served by
//   this CU's L1 or a stale L2 line), every storing wave drains its stores (s_waitcnt vmcnt(0)) before the workgroup
//   publishes, and the publish IS the all-reduce: each workgroup stores its K partial sums as 8-byte granules
//   {epoch, 32-bit half} (one atomic store each, so tag and payload arrive together), thread t of every workgroup sweeps
//   participant t's granules until they carry this step's epoch, and everybody forms the sums in the same fixed
//   order -- so every workgroup takes the same data-dependent decisions (negative curvature, boundary, kappa/theta stop).
//   Seeing a participant's step-e granules implies its z stores of step e have reached memory.
// * Every participant must be resident: the host sizes the grid to what the chip holds at once (and reserves those
//   slots process-wide, so that concurrently solved agents never wait for each other's workgroups); every spin is
//   bounded, a time-out raises PersistCtrl::error, the caller's iterate is left untouched and the host reruns the solve with
//   the multi-launch scheme.
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
constexpr int kPersistPoison = 3;          // DevState::rtr_stop of a launch in which a participant timed out
// s_sleep units (64 clocks each) before the first granule sweep of a reduction and between sweeps; packed into one
// kernel argument (first << 8 | between) so that they can be tuned at run time (DPGO_POLL_FIRST / DPGO_POLL_SLEEP)
constexpr int kPollFirstSleep = 24, kPollSleep = 3;
constexpr int kPollFirstPaySleep = 0;  // before the first sweep of a reduction with a payload (0: as a plain one); bits 16..23
__device__ __forceinline__ void sleep_units(int n) {
  for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
}
constexpr int kPersistMax = kBlock;        // participants (workgroups) of one launch
constexpr int kGranVals = 4;               // partial sums per all-reduce (max)
// A reduction can also CARRY a payload: up to kGranPay doubles per participant that are not summed but handed, participant
// t's to thread t of every workgroup (an all-gather riding on the all-reduce: same granules, same sweep, same epoch -- the
// additive preconditioner's restricted vectors, (D+1) R <= 24 doubles per aggregate).
constexpr int kGranPay = 24;
constexpr int kGranRows = 2 * (kGranVals + kGranPay);  // 8-byte words per participant: {epoch, low half}, {epoch, high half} per value
// granule table: [2 buffers][kGranVals + kGranPay values][kPersistMax participants] cells of 2 words (chip_allreduce) -- a
// row is contiguous over the participants, so a sweep is one coalesced 16-byte load per value and 64 participants
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

// Column loads / tile stores of the exchanged vector through a buffer descriptor: 16-byte pieces with the sc1 policy
// (aux bit 4 on gfx950; cdna_hip_programming.md Guideline 16 R1) -- an R-double column is R/2 16-byte loads (+ one 8-byte
// load), a pose tile leaves as 16-byte write-through stores instead of 8-byte ones (each of which is a fabric write of its
// own).  Only dword alignment is required of multi-dword global accesses.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr int kAuxSc1 = 16;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t vec_rsrc(double* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)bytes, 0x00020000);
}
template <int R>
__device__ __forceinline__ void ld_col_agent(__amdgpu_buffer_rsrc_t rz, int byte_off, double (&x)[R]) {
#pragma unroll
  for (int a = 0; a + 1 < R; a += 2) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rz, byte_off + 8 * a, 0, kAuxSc1);
    x[a] = __longlong_as_double((long long)(((unsigned long long)v.y << 32) | v.x));
    x[a + 1] = __longlong_as_double((long long)(((unsigned long long)v.w << 32) | v.z));
  }
  if constexpr (R & 1) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rz, byte_off + 8 * (R - 1), 0, kAuxSc1);
    x[R - 1] = __longlong_as_double((long long)(((unsigned long long)v.y << 32) | v.x));
  }
}

// Barrier + all-reduce of the launch's workgroups in one step, two workgroup barriers in all.  part: in = this THREAD's
// partial sums, out = the sums over the launch (identical bits in every thread of every workgroup).  PRECONDITION: every
// thread has executed `s_waitcnt vmcnt(0)` after its last store that other workgroups read (the first barrier below then
// orders the whole workgroup's stores before the publish).  red: 2 x (2 * kWaves * kGranVals) doubles.  *ok_s: 1 at kernel start.
// PAY > 0: `pay_ws` = the workgroup's payload as per-wave partial sums in LDS, [kWaves][PAY] (written by every wave BEFORE
// the call; the waves are added in order behind the first barrier), `got` = out: participant threadIdx.x's payload
// (threads >= members: zeros).
template <int K, int PAY = 0>
__device__ __forceinline__ bool chip_allreduce(unsigned long long* gran, int rank, int members, unsigned salt, unsigned& step,
                                               double (&part)[K], double* red, int* error, int* ok_s, int poll,
                                               const double* pay_ws = nullptr, double* got = nullptr) {
  static_assert(K <= kGranVals && PAY <= kGranPay, "granule rows");
  step += 1;
  const unsigned long long epoch = (unsigned long long)(salt | step);  // never 0; unique per launch and step
  unsigned long long* buf = gran + (size_t)(step & 1u) * kGranRows * kPersistMax;
  // Table of one buffer: CELLS of 16 bytes, cell (value v, participant t) at word 2 (v kPersistMax + t) = the value's two
  // granules {epoch, low half}, {epoch, high half} side by side.  Each granule is still ONE 8-byte store (tag and payload
  // arrive together, whatever happens to the pair); the sweep reads a cell with one 16-byte load -- half the load
  // instructions, and 8-byte accesses move at 0.54-0.70x the rate of 16-byte ones (MI355X_MICROARCH.md) -- and checks
  // both tags.  Values 0 .. kGranVals-1: the partial sums; kGranVals ..: the payload.
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rg =
      __builtin_amdgcn_make_buffer_rsrc(buf, 0, (int)(kGranRows * kPersistMax * sizeof(unsigned long long)), 0x00020000);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // two sets of scratch alternate: a wave that runs ahead into the next reduction must not overwrite sums a slower wave
  // of this one still reads (only wave-level barriers separate the two)
  red += (step & 1u) * (2 * kWaves * kGranVals);
#pragma unroll
  for (int k = 0; k < K; ++k) part[k] = wave_reduce_lane63(part[k]);
  if (lane == 63) {
#pragma unroll
    for (int k = 0; k < K; ++k) red[wave * K + k] = part[k];
  }
  __syncthreads();
  if ((int)threadIdx.x < 2 * K) {  // the workgroup's sums (waves in order), one granule per 32-bit half
    const int k = threadIdx.x >> 1, half = threadIdx.x & 1;
    double sum = red[k];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) sum += red[w * K + k];
    const unsigned long long bits = (unsigned long long)__double_as_longlong(sum);
    const unsigned long long w = (epoch << 32) | (half ? (bits >> 32) : (bits & 0xffffffffull));
    __hip_atomic_store(buf + ((size_t)k * kPersistMax + rank) * 2 + half, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if constexpr (PAY > 0) {  // the payload's granules: values kGranVals .. of the table
    const int g = (int)threadIdx.x - 2 * K;
    if (g >= 0 && g < 2 * PAY) {
      const int e = g >> 1, half = g & 1;
      double sum = pay_ws[e];
#pragma unroll
      for (int w = 1; w < kWaves; ++w) sum += pay_ws[w * PAY + e];
      const unsigned long long bits = (unsigned long long)__double_as_longlong(sum);
      const unsigned long long w = (epoch << 32) | (half ? (bits >> 32) : (bits & 0xffffffffull));
      __hip_atomic_store(buf + ((size_t)(kGranVals + e) * kPersistMax + rank) * 2 + half, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // thread t sweeps participant t's cells (one pass = K (+ PAY) 16-byte loads in flight, a row of the table per load instruction)
  double v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = 0.0;
  if constexpr (PAY > 0) {
#pragma unroll
    for (int e = 0; e < PAY; ++e) got[e] = 0.0;
  }
  if ((int)threadIdx.x < members) {
    const int t = threadIdx.x;
    bool got_all = false;
    // a granule needs ~1 us to cross the chip: polls before that only load the fabric the granules travel on (sweeping
    // at once made the reduction 1 us SLOWER at 196 workgroups), so the first sweep waits and the later ones back off
    sleep_units((PAY > 0 && ((poll >> 16) & 0xff)) ? ((poll >> 16) & 0xff) : ((poll >> 8) & 0xff));
    for (unsigned it = 0; it < kSpinLimit; ++it) {
      u32x4 c[K];
      [[maybe_unused]] u32x4 cp[PAY > 0 ? PAY : 1];
#pragma unroll
      for (int k = 0; k < K; ++k) c[k] = __builtin_amdgcn_raw_buffer_load_b128(rg, (k * kPersistMax + t) * 16, 0, kAuxSc1);
      if constexpr (PAY > 0) {
#pragma unroll
        for (int e = 0; e < PAY; ++e)
          cp[e] = __builtin_amdgcn_raw_buffer_load_b128(rg, ((kGranVals + e) * kPersistMax + t) * 16, 0, kAuxSc1);
      }
      const unsigned ep = (unsigned)epoch;
      bool all = true;
#pragma unroll
      for (int k = 0; k < K; ++k) all = all && (c[k].y == ep) && (c[k].w == ep);
      if constexpr (PAY > 0) {
#pragma unroll
        for (int e = 0; e < PAY; ++e) all = all && (cp[e].y == ep) && (cp[e].w == ep);
      }
      if (all) {
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = __hiloint2double((int)c[k].z, (int)c[k].x);
        if constexpr (PAY > 0) {
#pragma unroll
          for (int e = 0; e < PAY; ++e) got[e] = __hiloint2double((int)cp[e].z, (int)cp[e].x);
        }
        got_all = true;
        break;
      }
      if ((it & 255u) == 255u && __hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      sleep_units(poll & 0xff);
    }
    if (!got_all) {
      __hip_atomic_store(error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *ok_s = 0;  // sticky: the launch is abandoned
    }
  }
  // fixed tree (DPP inside a wave, waves in order): the same bits in every thread of every workgroup
  double* red2 = red + kWaves * K;
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = wave_reduce_lane63(v[k]);
  if (lane == 63) {
#pragma unroll
    for (int k = 0; k < K; ++k) red2[wave * K + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double sum = red2[k];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) sum += red2[w * K + k];
    part[k] = sum;
  }
  return *ok_s != 0;
}

// ---------------------------------------------------------------- the gather of the persistent kernel
// Row c of (Q z)_i for the lane (pose i, slice s, column c), with EVERY load of a tile issued before the first use: a
// phase then costs one memory round trip, not one per block (with one wave per SIMD nothing else hides them).  Outer-
// product form (cf. spmm_sym_pre): the lane loads only column c of a gathered tile (R doubles) and of the block, keeps the
// (D+1) x R partial, and one reduce-scatter over the pose's lanes per row leaves row c in lane c; slices (SPLIT lane
// groups taking every SPLIT-th block) are summed by a fixed shuffle tree into slice 0.  The first 2 (D+1) blocks of a row
// are covered by unconditional loads (absent blocks: a valid address and a zero block column); longer rows (hubs) take
// the loop.  QRES: the block columns stay in registers for the whole launch (Q does not change); otherwise they are
// loaded with the tiles (plain loads: L2-resident).
template <int D, int SPLIT>
struct GatherGeo {
  static constexpr int B = D + 1;
  static constexpr int NPRE = 2 * B;                       // blocks of a row covered without a loop
  static constexpr int NB = (NPRE + SPLIT - 1) / SPLIT;    // of them, per lane group
};
template <int D, int R, int SPLIT, bool QRES>
struct GatherOps {
  using GG = GatherGeo<D, SPLIT>;
  int t0, deg;
  int j[GG::NB];  // gathered pose of block m (own pose / pose 0 when the block is absent)
  // the same pose as a slot of the workgroup's OWN LDS tiles ((tile index) P + slot), -1 = another workgroup's: its
  // z is read from the tile this workgroup wrote itself instead of crossing the chip (DPGO_PERSIST_LOCAL; set by the kernel)
  int loc[GG::NB];
  double q[QRES ? GG::NB : 1][GG::B];
};
template <int D, int R, int SPLIT, bool QRES>
__device__ __forceinline__ void gather_setup(GatherOps<D, R, SPLIT, QRES>& go, const BsrDev& Q, int i, int s, int c, bool okp) {
  using GG = GatherGeo<D, SPLIT>;
  constexpr int B = GG::B, BB = B * B;
  go.t0 = okp ? Q.rowptr[i] : 0;
  go.deg = (okp ? Q.rowptr[i + 1] : 0) - go.t0;
#pragma unroll
  for (int m = 0; m < GG::NB; ++m) {
    const int blk = s + m * SPLIT;
    const bool on = blk < go.deg && blk < GG::NPRE;
    go.j[m] = on ? Q.colidx[go.t0 + blk] : (okp ? i : 0);
    go.loc[m] = -1;
    if constexpr (QRES) {
#pragma unroll
      for (int cc = 0; cc < B; ++cc) go.q[m][cc] = on ? Q.vals[(size_t)(go.t0 + blk) * BB + cc * B + c] : 0.0;
    }
  }
}
// own_tiles (LDS, [tiles][P][T] of the gathered vector, complete and ordered by a workgroup barrier; nullptr: none): the
// columns of poses this workgroup owns come from there
template <int D, int R, int SPLIT, bool QRES>
__device__ __forceinline__ void gather_issue(const GatherOps<D, R, SPLIT, QRES>& go, const BsrDev& Q,
                                             __amdgpu_buffer_rsrc_t rz, int s, int c,
                                             double (&xc)[GatherGeo<D, SPLIT>::NB][R],
                                             double (&qc)[GatherGeo<D, SPLIT>::NB][D + 1],
                                             const double* own_tiles = nullptr) {
  using GG = GatherGeo<D, SPLIT>;
  constexpr int B = GG::B, BB = B * B, T = B * R;
#pragma unroll
  for (int m = 0; m < GG::NB; ++m) {
    if (own_tiles && go.loc[m] >= 0) {
      load_col<R>(own_tiles + go.loc[m] * T + c * R, xc[m]);
    } else {
      ld_col_agent<R>(rz, (go.j[m] * T + c * R) * 8, xc[m]);
    }
    if constexpr (QRES) {
#pragma unroll
      for (int cc = 0; cc < B; ++cc) qc[m][cc] = go.q[m][cc];
    } else {
      const int blk = s + m * SPLIT;
      const bool on = blk < go.deg && blk < GG::NPRE;
      const double* qp = Q.vals + (size_t)(go.t0 + (on ? blk : 0)) * BB + c;  // (absent block: the row's first, times 0)
#pragma unroll
      for (int cc = 0; cc < B; ++cc) {
        const double v = qp[cc * B];
        qc[m][cc] = on ? v : 0.0;
      }
    }
  }
}
template <int D, int R, int SPLIT, bool QRES>
__device__ __forceinline__ void gather_finish(const GatherOps<D, R, SPLIT, QRES>& go, const BsrDev& Q,
                                              __amdgpu_buffer_rsrc_t rz, int s, int c, const double (&xc)[GatherGeo<D, SPLIT>::NB][R],
                                              const double (&qc)[GatherGeo<D, SPLIT>::NB][D + 1], double (&h)[R]) {
  using GG = GatherGeo<D, SPLIT>;
  constexpr int B = GG::B, BB = B * B, T = B * R;
  double acc[B][R];
#pragma unroll
  for (int cc = 0; cc < B; ++cc)
#pragma unroll
    for (int a = 0; a < R; ++a) acc[cc][a] = 0.0;
#pragma unroll
  for (int m = 0; m < GG::NB; ++m)
#pragma unroll
    for (int cc = 0; cc < B; ++cc)
#pragma unroll
      for (int a = 0; a < R; ++a) acc[cc][a] = fma(xc[m][a], qc[m][cc], acc[cc][a]);
  for (int t = go.t0 + GG::NPRE + s; t < go.t0 + go.deg; t += SPLIT) {  // rows with more than 2 (D+1) blocks
    double xr[R], qr[B];
    ld_col_agent<R>(rz, (Q.colidx[t] * T + c * R) * 8, xr);
#pragma unroll
    for (int cc = 0; cc < B; ++cc) qr[cc] = Q.vals[(size_t)t * BB + cc * B + c];
#pragma unroll
    for (int cc = 0; cc < B; ++cc)
#pragma unroll
      for (int a = 0; a < R; ++a) acc[cc][a] = fma(xr[a], qr[cc], acc[cc][a]);
  }
  pose_reduce_scatter<D, R>(acc, c, h);
  if constexpr (SPLIT > 1) {  // fixed-order tree over the slices; the sum lands in slice 0
#pragma unroll
    for (int o = SPLIT / 2; o >= 1; o >>= 1) {
#pragma unroll
      for (int a = 0; a < R; ++a) h[a] += __shfl_down(h[a], o * B);
    }
  }
}

// MT = tiles (of Geo::P poses) a workgroup owns: tile = rank + k * members, k < MT.  Every variant keeps up to 512
// registers per lane, i.e. one workgroup per CU; the host reserves resident slots accordingly (2 of the 2 per CU).
constexpr int persist_slots_per_wg(int, int, bool = false) { return 2; }

// ADD: the preconditioner is the ADDITIVE two-level combination  z = proj_X( w Dinv r + P A_c^-1 P^T r )  on the handle's
// two-level hierarchy with ONE aggregate per workgroup tile (at most Geo::P poses: 16 with 4 lane groups per pose, 64 with
// one pose per (D+1) lanes -- blocks up to ~14 000 poses in 3-D): block-Jacobi plus the coarse-grid correction of
// the residual itself, so nothing inside the preconditioner applies an operator to a vector other workgroups hold -- the
// only exchange is the restricted residual rc (n / P coarse nodes x (D+1) R doubles: an all-gather every workgroup reads
// in full), and it rides on a reduction the iteration needs anyway.  Per iteration (round 2-5 form, DPGO_ADD_PAYLOAD=0):
// phase A | all-reduce <delta, H delta> | r, eta update, x1 = Dinv r, rc = P^T r | all-reduce <r, r> (rc visible) |
// xc = (own rows of A_c^-1, resident in LDS) rc, z = proj_X(w x1 + P xc) | all-reduce <z, r> (z visible): three reductions
// instead of the V-cycle's five launches.  The oracle restates the operator (precond = "amg_additive").
//
// Round 6 (DPGO_ADD_PAYLOAD=1, the default): TWO reductions per iteration, as block-Jacobi.  The restriction is linear and
// tCG's residual is a recurrence, r <- r + alpha H delta, so rc <- rc + alpha P^T(H delta): the restriction of H delta is
// formed in phase A -- BEFORE alpha is known -- and travels as the PAYLOAD of the <delta, H delta> reduction
// (chip_allreduce<K, PAY>: an all-gather on the same granules and the same sweep); thread t of every workgroup keeps
// aggregate t's rows of rc in registers for the whole tCG run and adds alpha times what it swept.  Phase B then runs
// straight through -- r, eta, x1, rc, xc, z -- and <r, r>, <z, r> leave in ONE reduction.  Only the first residual of a
// tCG run (r0 = g) is restricted directly (payload of its <r, r> reduction).  Same operator; rc differs from P^T r by the
// rounding of the recurrence, as r itself does from g + H eta.
#ifndef DPGO_ADD_PAYLOAD
#define DPGO_ADD_PAYLOAD 1
#endif
constexpr bool kAddPayload = DPGO_ADD_PAYLOAD != 0;
struct AddDev {
  const double* Pb;    // prolongation blocks of level 0, [n][D+1][D+1] row-major
  const double* Minv;  // dense inverse of the coarse operator, row-major, leading dimension lda
  int lda, nc;         // nc coarse nodes = pose tiles
  double* rc;          // [nc][D+1][R] restricted residual (written and gathered inside the launch)
  double w;            // weight of the block-Jacobi term
  // graph aggregates: pose of every (aggregate, slot), nc x tile entries, -1 = empty slot; nullptr: aggregate a = the
  // poses [a tile, (a + 1) tile)
  const int32_t* perm;
  // graph aggregates: aggregate of every pose, its position in the member list, the member list's row pointer (a pose's
  // slot in its aggregate's tile = mem_pos - agg_ptr[aggregate]); nullptr with index runs
  const int32_t *lab, *mem_pos, *agg_ptr;
};
#ifndef DPGO_PERSIST_LOCAL
#define DPGO_PERSIST_LOCAL 1  // tCG's gather reads the poses of the workgroup's own tiles from LDS (0: everything from memory)
#endif
// Used by the layouts with one pose per (D+1) lanes (64-pose tiles: on a 12 500-pose slab three quarters of a graph
// aggregate's neighbours, 3 of 7 of an index tile's, are the workgroup's own -- Hessian phase 3.8 -> 2.25 us additive,
// 3.2 -> 1.9 block-Jacobi).  With 4 lane groups per pose (16-pose tiles) few neighbours are local and the test costs what
// it saves (torus3D block-Jacobi 10.0 -> 10.3 us per iteration): off.
template <int SPLIT>
constexpr bool persist_local() { return DPGO_PERSIST_LOCAL != 0 && SPLIT == 1; }

// Trust-region parameters of a solve (src/QuadraticOptimizer.cpp:64-78)
struct RtrArgs {
  double tol, Delta0, Delta_max;
  int max_inner, max_outer, accept_tiny;
};

// THE WHOLE SOLVE in one launch (QuadraticOptimizer::optimize, src/QuadraticOptimizer.cpp:26-108; ROPTLIB SolversTR::Run):
// cost and gradient at the initial iterate, then per outer iteration the tCG loop above, the retraction, cost / gradient
// at the trial point, the model decrease (H eta), the rho test and the radius update -- k_grad, k_rtr_begin, k_retract,
// k_hess and k_rtr_update of the multi-launch scheme, same arithmetic, evaluated on the registers / LDS tiles the tCG
// loop already holds.  Between workgroups travel, besides z: the trial point x2 and the step eta (write-through, gathered
// for x2 Q and eta Q); their dot products ride on two more reductions per outer iteration.  X is read at the start; the
// result is left in the trial-point buffer and committed to X by k_persist_commit (below) behind this kernel -- a launch
// in which ANY participant timed out leaves the caller's iterate untouched, the host reruns the solve with the multi-launch
// scheme.  The host's part of a solve: two launches, one 200-byte read-back.
template <int D, int R, int SPLIT, int MT, bool ADD = false>
__global__ __launch_bounds__(kBlock, 1) void k_rtr_persist(BsrDev Q, double* X, const double* __restrict__ Glin,
                                                           const double* __restrict__ dinv, double* xbuf, double* ebuf,
                                                           double* z, unsigned long long* gran, unsigned salt,
                                                           DevState* __restrict__ sout, PersistCtrl* ctrl, int n,
                                                           unsigned long long* hflag, unsigned gen, int poll, RtrArgs ra,
                                                           AddDev add) {
  using GEO = Geo<D, R, SPLIT>;
  constexpr int P = GEO::P, G = GEO::G, T = GEO::T, B = GEO::B, BB = GEO::BB;
  static_assert(!ADD || MT == 1, "additive preconditioner: one aggregate = one tile per workgroup");
  // resident in LDS: the poses' X (projections need all rotation columns of a pose) and z (Hessian correction; after the
  // tCG loop the same tiles hold the trial point x2); ex: two wave-private exchange tiles (the columns of one pose meet here)
  __shared__ __attribute__((aligned(16))) double Xs[MT][P][T], Zs[MT][P][T];
  // Riemannian gradient at the current iterate and at the trial point (touched once per outer iteration, each lane its own
  // column: kept out of the registers the tCG loop needs)
  __shared__ __attribute__((aligned(16))) double G1s[MT][P][T], G2s[MT][P][T];
  // (own-tile gathers, persist_local: the step eta of the workgroup's poses, for the H eta gather of the rho test)
  constexpr bool kLoc = persist_local<SPLIT>();
  __shared__ __attribute__((aligned(16))) double Es[kLoc ? MT : 1][kLoc ? P : 1][kLoc ? T : 1];
  __shared__ __attribute__((aligned(16))) double ex[2][kWaves][G][T];
  __shared__ double red[2 * 2 * kWaves * kGranVals];
  __shared__ int ok_s;
  // additive preconditioner: P_i^T r_i of the tile's poses, the aggregate's coarse solution, its per-wave partial sums;
  // dynamic LDS: the (D+1) rows of A_c^-1 this workgroup's aggregate needs, (D+1) x N_c doubles
  __shared__ double ts[ADD ? P : 1][ADD ? T : 1], xc_s[ADD ? T : 1], xw_s[ADD ? kWaves : 1][ADD ? T : 1];
  __shared__ double tw_s[ADD ? kWaves : 1][ADD ? T : 1];  // per-wave sums of P_i^T r_i
  extern __shared__ double Ms[];

  const int rank = blockIdx.x, members = gridDim.x;
  const unsigned long long t_entry = wall_clock64();  // (diagnostic timeline, PersistCtrl::ticks[5..7])
  if (threadIdx.x == 0) ok_s = 1;  // (ordered before its first use by the barriers of the first all-reduce)
  DevState st;
  unsigned step = 0;
  const LaneId L = lane_id<D, SPLIT>();
  const int lp = L.wave * G + L.g;  // pose slot inside a workgroup tile
  // (additive preconditioner: a workgroup's tile = an aggregate; with graph aggregates its poses are anywhere)
  const int ntiles = ADD ? add.nc : (n + P - 1) / P;
  const int co = L.c * R;
  __shared__ int pidx_s[ADD ? P : 1];  // ADD: pose of every slot of the tile (-1: empty), for the publishing lanes

  // ---- resident data of the workgroup's rows (registers; X and z also in LDS)
  // block columns resident in registers (2 blocks per lane group with SPLIT = 4; with SPLIT = 1 the row's first 2 (D+1)
  // blocks: 3-D only where one tile per workgroup leaves the room, 2-D -- 18 doubles per tile -- always)
  // (the non-resident form is instantiated for 3-D only: its <D = 2, R = 3, SPLIT = 1, MT = 2> instance returned a
  // deterministic but wrong step -- relative error 2e-4 against the oracle on kitti_00, every other (d, r) instance and
  // the resident form of the same instance agree to 4e-14; tools/diag_layout.py)
  constexpr bool QRES = (SPLIT > 1) || (MT == 1) || (D == 2);
  const __amdgpu_buffer_rsrc_t rz = vec_rsrc(z, (size_t)n * T * sizeof(double));
  const __amdgpu_buffer_rsrc_t rX = vec_rsrc(X, (size_t)n * T * sizeof(double));
  const __amdgpu_buffer_rsrc_t rx2 = vec_rsrc(xbuf, (size_t)n * T * sizeof(double));
  const __amdgpu_buffer_rsrc_t reta = vec_rsrc(ebuf, (size_t)n * T * sizeof(double));
  using GG = GatherGeo<D, SPLIT>;
  GatherOps<D, R, SPLIT, QRES> go[MT];
  int pose[MT];
  bool okp[MT], own[MT];
  double rr[MT][R], ee[MT][R], dl[MT][R], hd[MT][R], zc[MT][R], srow[MT][D], drow[MT][B];
#pragma unroll
  for (int k = 0; k < MT; ++k) {
    const int tile = rank + k * members;
    pose[k] = tile * P + lp;
    if constexpr (ADD) {
      if (add.perm) pose[k] = (tile < ntiles && L.g < G) ? add.perm[tile * P + lp] : -1;
    }
    okp[k] = (tile < ntiles) && (L.g < G) && (pose[k] >= 0) && (pose[k] < n);
    if (!okp[k]) pose[k] = 0;  // (a valid row for the wave-cooperative helpers; never used)
    if constexpr (ADD) {
      if (L.s == 0 && L.c == 0 && L.g < G) pidx_s[lp] = okp[k] ? pose[k] : -1;
    }
    own[k] = okp[k] && (L.s == 0);
    gather_setup<D, R, SPLIT, QRES>(go[k], Q, pose[k], L.s, L.c, okp[k]);
    if constexpr (persist_local<SPLIT>()) {
#pragma unroll
      for (int m = 0; m < GG::NB; ++m) {
        const int j = go[k].j[m];
        int lc = -1;
        if constexpr (ADD) {
          if (add.lab) {  // (both lookups issued together: one round trip)
            const int la = add.lab[j], mp = add.mem_pos[j];
            if (la == rank) lc = mp - add.agg_ptr[rank];
          } else if (j / P == rank) {
            lc = j - rank * P;
          }
        } else {
#pragma unroll
          for (int kk = 0; kk < MT; ++kk)
            if (j / P == rank + kk * members) lc = kk * P + (j - (rank + kk * members) * P);
        }
        go[k].loc[m] = lc;
      }
    }
#pragma unroll
    for (int a = 0; a < R; ++a) rr[k][a] = ee[k][a] = dl[k][a] = hd[k][a] = zc[k][a] = 0.0;
#pragma unroll
    for (int a = 0; a < D; ++a) srow[k][a] = 0.0;
#pragma unroll
    for (int a = 0; a < B; ++a) drow[k][a] = 0.0;
    if (own[k]) {
      const size_t off = (size_t)pose[k] * T + co;
#pragma unroll
      for (int a = 0; a < R; ++a) Xs[k][lp][co + a] = X[off + a];
      if (dinv) {
#pragma unroll
        for (int a = 0; a < B; ++a) drow[k][a] = dinv[(size_t)pose[k] * BB + L.c * B + a];
      }
    }
  }
  wave_sync();
  [[maybe_unused]] double pcol[B], prow[B];  // column c / row c of the pose's prolongation block
  // (payload form) aggregate threadIdx.x's (D+1) x R entries of the restricted residual, kept for a whole tCG run, and of
  // the restricted H delta the last Hessian-step reduction carried
  [[maybe_unused]] double rct[ADD ? T : 1], hct[ADD ? T : 1];
  [[maybe_unused]] const int Nc = add.nc * B;
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rrc = vec_rsrc(add.rc, (size_t)(ADD ? add.nc : 0) * T * sizeof(double));
  if constexpr (ADD) {
#pragma unroll
    for (int cc = 0; cc < B; ++cc) {
      pcol[cc] = own[0] ? add.Pb[(size_t)pose[0] * BB + cc * B + L.c] : 0.0;
      prow[cc] = own[0] ? add.Pb[(size_t)pose[0] * BB + L.c * B + cc] : 0.0;
    }
    if (rank < ntiles) {
      // (all of a thread's loads in flight together -- Nc <= kPersistMax (D+1): at most D+1 per row -- instead of one
      // dependent round trip per element: the additive form's set-up was 9 us longer than block-Jacobi's)
      constexpr int NJ = (kPersistMax * B + kBlock - 1) / kBlock;
      double mv[B][NJ];
#pragma unroll
      for (int row = 0; row < B; ++row)
#pragma unroll
        for (int q = 0; q < NJ; ++q) {
          const int j = (int)threadIdx.x + q * kBlock;
          mv[row][q] = j < Nc ? add.Minv[(size_t)(rank * B + row) * add.lda + j] : 0.0;
        }
#pragma unroll
      for (int row = 0; row < B; ++row)
#pragma unroll
        for (int q = 0; q < NJ; ++q) {
          const int j = (int)threadIdx.x + q * kBlock;
          if (j < Nc) Ms[row * Nc + j] = mv[row][q];
        }
    }
    __syncthreads();
  }

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
          if constexpr (T % 2 != 0) st_agent(z + off + a, out[a]);  // the copy the other workgroups gather
        }
      }
      if constexpr (T % 2 == 0) {
        // the wave's poses are one contiguous span of z whose layout is the LDS tile's: it leaves as lane-linear 16-byte
        // write-through pieces
        wave_sync();
        const int p0w = (rank + k * members) * P + L.wave * G;
        const int npose = (n - p0w) < G ? (n - p0w) : G;
        const int pieces = npose > 0 ? npose * (T / 2) : 0;
        const dbl2* span = reinterpret_cast<const dbl2*>(&Zs[k][L.wave * G][0]);
#pragma unroll
        for (int it = 0; it < (G * (T / 2) + 63) / 64; ++it) {
          const int pc = (int)(threadIdx.x & 63) + 64 * it;
          if (pc < pieces) {
            const dbl2 v = span[pc];
            u32x4 w;
            w.x = (unsigned)__double2loint(v.x);
            w.y = (unsigned)__double2hiint(v.x);
            w.z = (unsigned)__double2loint(v.y);
            w.w = (unsigned)__double2hiint(v.y);
            __builtin_amdgcn_raw_buffer_store_b128(w, rz, (p0w * T + 2 * pc) * 8, 0, kAuxSc1);
          }
        }
      }
      // ex[0] / ex[1] of the next tile are written only after this tile's reads: with block-Jacobi the next tile's first
      // wave_sync stands between them; without a preconditioner this one does
      if (!dinv) wave_sync();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's z stores have been acknowledged
  };

  // ---- ADD: the wave's pose slots leave as lane-linear 16-byte write-through pieces, every pose to its own place (the
  // poses of a graph aggregate are not contiguous)
  [[maybe_unused]] auto publish_slots = [&](__amdgpu_buffer_rsrc_t rs, const double* wave_tile) {
    if constexpr (ADD && T % 2 == 0) {
      const dbl2* span = reinterpret_cast<const dbl2*>(wave_tile);
#pragma unroll
      for (int it = 0; it < (G * (T / 2) + 63) / 64; ++it) {
        const int pc = (int)(threadIdx.x & 63) + 64 * it;
        if (pc < G * (T / 2)) {
          const int ps = pc / (T / 2), sub = pc - ps * (T / 2);
          const int gp = pidx_s[L.wave * G + ps];
          if (gp >= 0) {
            const dbl2 v = span[pc];
            u32x4 w;
            w.x = (unsigned)__double2loint(v.x);
            w.y = (unsigned)__double2hiint(v.x);
            w.z = (unsigned)__double2loint(v.y);
            w.w = (unsigned)__double2hiint(v.y);
            __builtin_amdgcn_raw_buffer_store_b128(w, rs, (gp * T + 2 * sub) * 8, 0, kAuxSc1);
          }
        }
      }
    }
  };

  // ---- additive preconditioner, first half: r, eta update; x1 = Dinv r (kept in zc); rc = sum over the tile of P_i^T r_i
  // (payload form: `with_rc` = the restriction is wanted -- the first residual of a tCG run; it is left as per-wave sums in
  // tw_s for the reduction that carries it)
  // P_i^T v_i of the wave's poses (their B columns of v in `tile`), summed over the wave's G pose slots in a fixed order
  [[maybe_unused]] auto restrict_to_waves = [&](const double (*tile)[T]) {
    if constexpr (ADD) {
      if (L.s == 0 && L.g < G) {
        double t[R];
#pragma unroll
        for (int a = 0; a < R; ++a) t[a] = 0.0;
        if (own[0]) {
#pragma unroll
          for (int cc = 0; cc < B; ++cc) {  // row c of P_i^T v_i = sum_c' P_i[c'][c] v_i[c'][:]
#pragma unroll
            for (int a = 0; a < R; ++a) t[a] = fma(pcol[cc], tile[L.g][cc * R + a], t[a]);
          }
        }
        store_col<R>(&ts[lp][co], t);  // zeros for pose slots beyond n
      }
      wave_sync();
      if ((int)(threadIdx.x & 63) < T) {
        const int e = threadIdx.x & 63;
        double sum = ts[L.wave * G][e];
#pragma unroll
        for (int m = 1; m < G; ++m) sum += ts[L.wave * G + m][e];
        tw_s[L.wave][e] = sum;
      }
    }
  };
  auto phase_add_restrict = [&](bool first, double alpha, double (&part)[1], bool with_rc = true) {
    part[0] = 0.0;
    if constexpr (ADD) {
      if (own[0]) {
#pragma unroll
        for (int a = 0; a < R; ++a) {
          if (!first) {
            ee[0][a] = fma(alpha, dl[0][a], ee[0][a]);
            rr[0][a] = fma(alpha, hd[0][a], rr[0][a]);
          }
          part[0] = fma(rr[0][a], rr[0][a], part[0]);
        }
        store_col<R>(&ex[0][L.wave][L.g][co], rr[0]);
      }
      wave_sync();  // the pose's B columns of r are in LDS
      if (own[0]) jacobi_col<D, R>(&ex[0][L.wave][L.g][0], drow[0], zc[0]);  // x1 (unweighted), until z replaces it
      if constexpr (kAddPayload) {
        if (with_rc) restrict_to_waves(ex[0][L.wave]);
        return;
      }
      // fixed order: every wave adds up its own G pose slots, then the waves in order; the all-gathered coarse residual
      restrict_to_waves(ex[0][L.wave]);
      __syncthreads();
      if ((int)threadIdx.x < T && rank < ntiles) {
        double sum = tw_s[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) sum += tw_s[w][threadIdx.x];
        st_agent(add.rc + (size_t)rank * T + threadIdx.x, sum);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // acknowledged before the reduction publishes
    }
  };
  // ---- second half: xc = (own rows of A_c^-1) rc, z = proj_X(w x1 + P_i xc), <z, r>
  auto phase_add_correct = [&](double (&part)[1]) {
    part[0] = 0.0;
    if constexpr (ADD) {
      double acc[B][R];
#pragma unroll
      for (int row = 0; row < B; ++row)
#pragma unroll
        for (int a = 0; a < R; ++a) acc[row][a] = 0.0;
      // a coarse unknown's R right-hand sides: one thread each, ALL of a thread's columns requested before the first use
      // (nc <= kPersistMax aggregates: at most NJ = D + 1 columns per thread; one memory round trip, not one per column)
      if constexpr (kAddPayload) {
        // thread t holds aggregate t's B coarse unknowns (rct): B columns of the workgroup's rows of the inverse each
        const int t = threadIdx.x;
        const bool on = t < add.nc;
#pragma unroll
        for (int b = 0; b < B; ++b) {
#pragma unroll
          for (int row = 0; row < B; ++row) {
            const double mv = on ? Ms[row * Nc + (on ? t : 0) * B + b] : 0.0;
#pragma unroll
            for (int a = 0; a < R; ++a) acc[row][a] = fma(mv, rct[b * R + a], acc[row][a]);
          }
        }
      } else {
        constexpr int NJ = (kPersistMax * B + kBlock - 1) / kBlock;
        double rj[NJ][R];
#pragma unroll
        for (int m = 0; m < NJ; ++m) {
          const int j = (int)threadIdx.x + m * kBlock;
          ld_col_agent<R>(rrc, (j < Nc ? j : 0) * R * 8, rj[m]);
        }
#pragma unroll
        for (int m = 0; m < NJ; ++m) {
          const int j = (int)threadIdx.x + m * kBlock;
          const bool on = j < Nc;
#pragma unroll
          for (int row = 0; row < B; ++row) {
            const double mv = on ? Ms[row * Nc + j] : 0.0;
#pragma unroll
            for (int a = 0; a < R; ++a) acc[row][a] = fma(mv, rj[m][a], acc[row][a]);
          }
        }
      }
      if constexpr (kAddPayload) {  // the T wave sums as one reduce-scatter over the wave's rows (common.h)
        constexpr int TP = (T + 3) / 4 * 4;
        double flat[TP], rs[TP / 4];
#pragma unroll
        for (int e = 0; e < TP; ++e) flat[e] = e < T ? acc[e / R][e % R] : 0.0;
        wave_reduce_rows<TP>(flat, rs);
        if ((threadIdx.x & 15) == 15) {
          const int q = wave_rows_value(threadIdx.x & 63);
#pragma unroll
          for (int j = 0; j < TP / 4; ++j)
            if (4 * j + q < T) xw_s[threadIdx.x >> 6][4 * j + q] = rs[j];
        }
      } else {
#pragma unroll
        for (int row = 0; row < B; ++row)
#pragma unroll
          for (int a = 0; a < R; ++a) {
            const double v = wave_reduce_lane63(acc[row][a]);
            if ((threadIdx.x & 63) == 63) xw_s[threadIdx.x >> 6][row * R + a] = v;
          }
      }
      __syncthreads();
      if ((int)threadIdx.x < T) {
        double sum = xw_s[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) sum += xw_s[w][threadIdx.x];
        xc_s[threadIdx.x] = sum;
      }
      __syncthreads();
      double xx[R];
      if (own[0]) {
#pragma unroll
        for (int a = 0; a < R; ++a) {
          double v = add.w * zc[0][a];
#pragma unroll
          for (int cc = 0; cc < B; ++cc) v = fma(prow[cc], xc_s[cc * R + a], v);
          xx[a] = v;
        }
        store_col<R>(&ex[1][L.wave][L.g][co], xx);
      }
      wave_sync();
      if (own[0]) {
        double out[R], s[D];
        proj_col<D, R>(&Xs[0][lp][0], &ex[1][L.wave][L.g][0], L.c, xx, out, s);
        const size_t off = (size_t)pose[0] * T + co;
#pragma unroll
        for (int a = 0; a < R; ++a) {
          part[0] = fma(out[a], rr[0][a], part[0]);
          zc[0][a] = out[a];
          Zs[0][lp][co + a] = out[a];
          if constexpr (T % 2 != 0) st_agent(z + off + a, out[a]);
        }
      }
      if constexpr (T % 2 == 0) {
        wave_sync();
        publish_slots(rz, &Zs[0][L.wave * G][0]);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };

  // ---- phase A: Hz on the own rows (one hop: Q blocks + gathered z tiles of ALL owned tiles are requested before the
  // first epilogue), direction recurrences, <delta, H delta>
  auto phase_hess = [&](bool first, double beta, double (&part)[1]) {
    part[0] = 0.0;
    double h[MT][R];
    // (z of the workgroup's own poses: the LDS tiles the update phase wrote, ordered by the barriers of its reduction)
    const double* own_z = persist_local<SPLIT>() ? &Zs[0][0][0] : nullptr;
    if constexpr (SPLIT > 1) {  // few loads per tile: all tiles' requests go out before the first reduction
      double xc[MT][GG::NB][R], qc[MT][GG::NB][B];
#pragma unroll
      for (int k = 0; k < MT; ++k) gather_issue<D, R, SPLIT, QRES>(go[k], Q, rz, L.s, L.c, xc[k], qc[k], own_z);
#pragma unroll
      for (int k = 0; k < MT; ++k) gather_finish<D, R, SPLIT, QRES>(go[k], Q, rz, L.s, L.c, xc[k], qc[k], h[k]);
    } else {  // one pose per (D+1) lanes: a tile's 2 (D+1) R + 2 (D+1)^2 loads fill the register budget; tile after tile
#pragma unroll
      for (int k = 0; k < MT; ++k) {
        double xc[GG::NB][R], qc[GG::NB][B];
        gather_issue<D, R, SPLIT, QRES>(go[k], Q, rz, L.s, L.c, xc, qc, own_z);
        gather_finish<D, R, SPLIT, QRES>(go[k], Q, rz, L.s, L.c, xc, qc, h[k]);
      }
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
    if constexpr (ADD && kAddPayload) {  // P^T (H delta) of the tile rides on the reduction of <delta, H delta>
      if (own[0]) store_col<R>(&ex[1][L.wave][L.g][co], hd[0]);
      wave_sync();
      restrict_to_waves(ex[1][L.wave]);
    }
  };

  // ---- publish a pose tile to the other workgroups (write-through): even tile sizes leave as the wave's contiguous span
  // of lane-linear 16-byte pieces read from its LDS copy `wave_tile` ([G][T], already written by the pose's lanes), odd
  // ones column by column from registers
  auto publish_tile = [&](double* gbuf, __amdgpu_buffer_rsrc_t rs, const double* wave_tile, int tile, int k,
                          const double (&col)[R]) {
    if constexpr (ADD && T % 2 == 0) {
      wave_sync();
      publish_slots(rs, wave_tile);
    } else if constexpr (T % 2 == 0) {
      wave_sync();
      const int p0w = tile * P + L.wave * G;
      const int npose = (n - p0w) < G ? (n - p0w) : G;
      const int pieces = npose > 0 ? npose * (T / 2) : 0;
      const dbl2* span = reinterpret_cast<const dbl2*>(wave_tile);
#pragma unroll
      for (int it = 0; it < (G * (T / 2) + 63) / 64; ++it) {
        const int pc = (int)(threadIdx.x & 63) + 64 * it;
        if (pc < pieces) {
          const dbl2 v = span[pc];
          u32x4 w;
          w.x = (unsigned)__double2loint(v.x);
          w.y = (unsigned)__double2hiint(v.x);
          w.z = (unsigned)__double2loint(v.y);
          w.w = (unsigned)__double2hiint(v.y);
          __builtin_amdgcn_raw_buffer_store_b128(w, rs, (p0w * T + 2 * pc) * 8, 0, kAuxSc1);
        }
      }
    } else {
      if (own[k]) {
        const size_t off = (size_t)pose[k] * T + co;
#pragma unroll
        for (int a = 0; a < R; ++a) st_agent(gbuf + off + a, col[a]);
      }
    }
  };

  // ---- cost and Riemannian gradient (k_grad) at the point whose tiles are in LDS (Yt) and in memory behind `ry`:
  // partials [0] sum(YQ . Y)  [1] sum(Y . G)  [2] |rgrad|^2; rg = the gradient's own column, s = this lane's row of
  // S = sym(Y^T EG) (ROPTLIB caches it for the Hessian)
  // (the point's tiles Yt are complete and ordered by a workgroup barrier when this runs: the gather takes the
  // workgroup's own poses from them, as tCG's does)
  auto phase_grad = [&](__amdgpu_buffer_rsrc_t ry, const double (&Yt)[MT][P][T], double (&part)[3], double (&rg)[MT][P][T],
                        double (&sr)[MT][D]) {
    part[0] = part[1] = part[2] = 0.0;
    const double* own_y = kLoc ? &Yt[0][0][0] : nullptr;
#pragma unroll
    for (int k = 0; k < MT; ++k) {
      double eg[R];
      {
        double xc[GG::NB][R], qc[GG::NB][B];
        gather_issue<D, R, SPLIT, QRES>(go[k], Q, ry, L.s, L.c, xc, qc, own_y);
        gather_finish<D, R, SPLIT, QRES>(go[k], Q, ry, L.s, L.c, xc, qc, eg);
      }
      if (rank + k * members >= ntiles) continue;  // workgroup-uniform (the gather above is wave-cooperative)
      double* xt = &ex[k & 1][L.wave][L.g][0];
      if (own[k]) {
        const size_t off = (size_t)pose[k] * T + co;
#pragma unroll
        for (int a = 0; a < R; ++a) {
          const double xv = Yt[k][lp][co + a];
          part[0] = fma(eg[a], xv, part[0]);
          if (Glin) {
            const double gv = Glin[off + a];
            part[1] = fma(xv, gv, part[1]);
            eg[a] += gv;
          }
        }
        store_col<R>(xt + co, eg);
      }
      wave_sync();
      if (own[k]) {
        double out[R], s[D];
        proj_col<D, R>(&Yt[k][lp][0], xt, L.c, eg, out, s);
#pragma unroll
        for (int a = 0; a < R; ++a) {
          part[2] = fma(out[a], out[a], part[2]);
          rg[k][lp][co + a] = out[a];
        }
#pragma unroll
        for (int a = 0; a < D; ++a) sr[k][a] = s[a];
      }
    }
    wave_sync();  // (ex is free again)
  };

  // ==== QuadraticOptimizer::optimize: statistics at the initial iterate (k_grad + k_rtr_begin)
  bool alive = true;
  {
    double p3[3];
    if constexpr (kLoc) __syncthreads();  // (the X tiles of all waves are in LDS: the gather below reads other waves' rows)
    phase_grad(rX, Xs, p3, G1s, srow);
    alive = chip_allreduce<3>(gran, rank, members, salt, step, p3, red, &ctrl->error, &ok_s, poll);
    st.f1 = 0.5 * p3[0] + p3[1];
    st.ngf = sqrt(p3[2]);
    st.Delta = ra.Delta0;
    st.Delta_max = ra.Delta_max;
    st.tol = ra.tol;
    st.f2 = st.f1;
    st.rho = 0.0;
    st.fInit = st.f1;
    st.gnInit = st.ngf;
    st.xqx = p3[0];
    st.xg = p3[1];
    st.outer_iter = 0;
    st.rtr_stop = (st.ngf < ra.tol) ? 1 : 0;
    st.accepted_last = 0;
    st.n_accept = 0;
    st.accept_tiny = ra.accept_tiny;
    st.pad0 = 0;
    st.z_r = st.d_Pd = st.e_Pd = st.e_Pe = st.norm_r0 = st.alpha = st.d_Hd = 0.0;
    st.theta = 1.0;  // ROPTLIB RTRNewton defaults (SURVEY 8c' item 4)
    st.kappa = 0.1;
    st.tcg_j = 0;
    st.tcg_done = 0;
    st.tcg_status = TCG_MAXITER;
    st.max_inner = ra.max_inner;
    st.n_hess = 0;
    st.min_inner = 0;
  }
  unsigned iters = 0;
  // [5] kernel entry -> initial statistics reduced  [6] first updates (r0 = g, z0) of the outer iterations, summed
  // [7] retraction, trial point, H eta, rho test of the outer iterations, summed
  unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  tk[5] = wall_clock64() - t_entry;
  bool moved = false;  // an accepted step: X has to be written back

  // ==== SolversTR::Run: outer iterations
  while (alive && !st.rtr_stop && st.outer_iter < ra.max_outer) {
  // ---- tCG_TR (ROPTLIB): the scalar logic of tcg_update_prologue / tcg_hess_prologue, evaluated redundantly (and
  // identically: same partials, same summation order) by every participant
#pragma unroll
  for (int k = 0; k < MT; ++k) {
#pragma unroll
    for (int a = 0; a < R; ++a) {
      rr[k][a] = own[k] ? G1s[k][lp][co + a] : 0.0;  // r0 = g
      ee[k][a] = dl[k][a] = hd[k][a] = 0.0;
    }
  }
  st.tcg_done = 0;
  st.tcg_j = 0;
  st.tcg_status = TCG_MAXITER;
  st.e_Pe = 0.0;
  st.e_Pd = 0.0;
  double pr[2];
  // residual update + preconditioner + the reduction(s) that carry <r,r>, <z,r>: one all-reduce with block-Jacobi / no
  // preconditioner, two (the restricted residual becomes visible with the first) with the additive two-level one
  unsigned long long tmid = 0;  // diagnostic: end of the update phase proper
  auto update_and_reduce = [&](bool first, double alpha) -> bool {
    if constexpr (ADD && kAddPayload) {
      double p1[1], p2[1];
      if (first) {  // r0 = g: restricted directly, all-gathered with <r, r>
        phase_add_restrict(true, alpha, p1, true);
        tmid = wall_clock64();
        if (!chip_allreduce<1, T>(gran, rank, members, salt, step, p1, red, &ctrl->error, &ok_s, poll, &tw_s[0][0], rct))
          return false;
        phase_add_correct(p2);
        if (!chip_allreduce<1>(gran, rank, members, salt, step, p2, red, &ctrl->error, &ok_s, poll)) return false;
        pr[0] = p1[0];
        pr[1] = p2[0];
        return true;
      }
      phase_add_restrict(false, alpha, p1, false);
#pragma unroll
      for (int e = 0; e < T; ++e) rct[e] = fma(alpha, hct[e], rct[e]);  // rc <- rc + alpha P^T (H delta)
      phase_add_correct(p2);
      pr[0] = p1[0];
      pr[1] = p2[0];
      tmid = wall_clock64();
      return chip_allreduce<2>(gran, rank, members, salt, step, pr, red, &ctrl->error, &ok_s, poll);
    } else if constexpr (ADD) {
      double p1[1], p2[1];
      phase_add_restrict(first, alpha, p1);
      tmid = wall_clock64();
      if (!chip_allreduce<1>(gran, rank, members, salt, step, p1, red, &ctrl->error, &ok_s, poll)) return false;
      phase_add_correct(p2);
      if (!chip_allreduce<1>(gran, rank, members, salt, step, p2, red, &ctrl->error, &ok_s, poll)) return false;
      pr[0] = p1[0];
      pr[1] = p2[0];
      return true;
    } else {
      phase_update(first, alpha, pr);
      tmid = wall_clock64();
      return chip_allreduce<2>(gran, rank, members, salt, step, pr, red, &ctrl->error, &ok_s, poll);
    }
  };
  {
    const unsigned long long tu = wall_clock64();
    alive = update_and_reduce(true, 0.0);
    tk[6] += wall_clock64() - tu;
  }
  if (alive) {
    st.norm_r0 = sqrt(pr[0]);
    st.z_r = pr[1];
    st.d_Pd = pr[1];
    st.e_Pd = 0.0;
    if (st.max_inner <= 0) st.tcg_done = 1;
  }
  double beta = 0.0;
  bool first = true;
  while (alive && !st.tcg_done) {
    const unsigned long long t0 = wall_clock64();
    double dh[1];
    phase_hess(first, beta, dh);
    const unsigned long long t1 = wall_clock64();
    if constexpr (ADD && kAddPayload)
      alive = chip_allreduce<1, T>(gran, rank, members, salt, step, dh, red, &ctrl->error, &ok_s, poll, &tw_s[0][0], hct);
    else
      alive = chip_allreduce<1>(gran, rank, members, salt, step, dh, red, &ctrl->error, &ok_s, poll);
    if (!alive) break;
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
    // (additive preconditioner: its second phase and reduction are reported under "all-reduce after B")
    if (!(alive = update_and_reduce(false, alpha))) break;
    const unsigned long long t3 = tmid, t4 = wall_clock64();
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
  if (!alive) break;

  // ---- retraction x2 = R_x1(eta) (k_retract: qf of Y + eta, p + eta); x2 and eta are published for the two gathers
  // below; the x2 tiles take the place of z in LDS (z is dead until the next tCG run)
  const unsigned long long t_tail = wall_clock64();
  double p1[1] = {0.0};  // <eta, g1>
#pragma unroll
  for (int k = 0; k < MT; ++k) {
    if (rank + k * members >= ntiles) break;  // workgroup-uniform
    double a2[R];
    if (own[k]) {
#pragma unroll
      for (int a = 0; a < R; ++a) {
        a2[a] = Xs[k][lp][co + a] + ee[k][a];
        p1[0] = fma(ee[k][a], G1s[k][lp][co + a], p1[0]);
      }
      store_col<R>(&ex[0][L.wave][L.g][co], a2);
    }
    wave_sync();
    if (own[k]) {
      qf_col<D, R>(&ex[0][L.wave][L.g][0], L.c, a2);
      store_col<R>(&Zs[k][lp][co], a2);
      store_col<R>(&ex[1][L.wave][L.g][co], ee[k]);
      if constexpr (kLoc) store_col<R>(&Es[k][lp][co], ee[k]);
    }
    publish_tile(xbuf, rx2, &Zs[k][L.wave * G][0], rank + k * members, k, a2);
    publish_tile(ebuf, reta, &ex[1][L.wave][0][0], rank + k * members, k, ee[k]);
    wave_sync();  // ex of the next tile is rewritten only after this tile's reads
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // acknowledged before the reduction publishes
  if (!(alive = chip_allreduce<1>(gran, rank, members, salt, step, p1, red, &ctrl->error, &ok_s, poll))) break;
  const double eta_g = p1[0];

  // ---- cost / gradient at the trial point, and H eta = proj_x1(eta Q - eta_rot S1) for the model decrease (k_hess)
  double s2[MT][D], p4[4];
  {
    double p3[3];
    phase_grad(rx2, Zs, p3, G2s, s2);
    p4[0] = p3[0];
    p4[1] = p3[1];
    p4[2] = p3[2];
    p4[3] = 0.0;
  }
#pragma unroll
  for (int k = 0; k < MT; ++k) {
    double h[R];
    {
      double xc[GG::NB][R], qc[GG::NB][B];
      gather_issue<D, R, SPLIT, QRES>(go[k], Q, reta, L.s, L.c, xc, qc, kLoc ? &Es[0][0][0] : nullptr);
      gather_finish<D, R, SPLIT, QRES>(go[k], Q, reta, L.s, L.c, xc, qc, h);
    }
    if (rank + k * members >= ntiles) continue;
    if (own[k]) store_col<R>(&ex[0][L.wave][L.g][co], ee[k]);  // the pose's columns of eta
    wave_sync();
    if (own[k]) {
      if (L.c < D) {
#pragma unroll
        for (int a = 0; a < D; ++a) {
#pragma unroll
          for (int q = 0; q < R; ++q) h[q] = fma(-ex[0][L.wave][L.g][a * R + q], srow[k][a], h[q]);
        }
      }
      store_col<R>(&ex[1][L.wave][L.g][co], h);
    }
    wave_sync();
    if (own[k]) {
      double hz[R], s[D];
      proj_col<D, R>(&Xs[k][lp][0], &ex[1][L.wave][L.g][0], L.c, h, hz, s);
#pragma unroll
      for (int a = 0; a < R; ++a) p4[3] = fma(ee[k][a], hz[a], p4[3]);
    }
    wave_sync();
  }
  if (!(alive = chip_allreduce<4>(gran, rank, members, salt, step, p4, red, &ctrl->error, &ok_s, poll))) break;

  // ---- rho test, radius update, acceptance (k_rtr_update; identical in every workgroup)
  {
    const double f2 = 0.5 * p4[0] + p4[1];
    const double ngf2 = sqrt(p4[2]);
    const double eta_Heta = p4[3];
    const double rho = (st.f1 - f2) / (-(eta_g + 0.5 * eta_Heta));
    if (rho > 0.75) {
      if (st.tcg_status == TCG_EXCREGION || st.tcg_status == TCG_NEGCURV) st.Delta *= 2.0;
      if (st.Delta > st.Delta_max) st.Delta = st.Delta_max;
    } else if (rho < 0.25) {
      st.Delta *= 0.25;
    }
    const double sqeps = 1.4901161193847656e-08;  // sqrt(DBL_EPSILON)
    bool accept = rho > 0.1;
    if (!accept && st.accept_tiny) accept = (fabs(st.f1 - f2) / (fabs(st.f1) + 1.0) < sqeps) && (f2 < st.f1);
    st.f2 = f2;
    st.rho = rho;
    st.accepted_last = accept ? 1 : 0;
    st.outer_iter += 1;
    if (accept) {
      st.f1 = f2;
      st.ngf = ngf2;
      st.n_accept += 1;
      st.rtr_stop = (ngf2 < st.tol) ? 1 : 0;
      st.xqx = p4[0];
      st.xg = p4[1];
      moved = true;
#pragma unroll
      for (int k = 0; k < MT; ++k) {
        if (own[k]) {
#pragma unroll
          for (int a = 0; a < R; ++a) {
            Xs[k][lp][co + a] = Zs[k][lp][co + a];    // x1 <- x2
            G1s[k][lp][co + a] = G2s[k][lp][co + a];  // g1 <- g2
          }
#pragma unroll
          for (int a = 0; a < D; ++a) srow[k][a] = s2[k][a];  // S1 <- S2
        }
      }
      wave_sync();
    }
  }
  tk[7] += wall_clock64() - t_tail;
  }  // outer iterations

  // ==== hand the result back: the state record, and the iterate (own columns) into the trial-point buffer -- every
  // participant has finished gathering from it once the last reduction is through.  The CALLER'S X is written by
  // k_persist_commit, launched behind this kernel, and only if no participant of the launch timed out: a workgroup cannot
  // know on its own whether a slower one will still fail, so nothing here touches X.
  const bool good = alive && !__hip_atomic_load(&ctrl->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (good && moved) {
#pragma unroll
    for (int k = 0; k < MT; ++k) {
      if (own[k]) {
        const size_t off = (size_t)pose[k] * T + co;
#pragma unroll
        for (int a = 0; a < R; ++a) xbuf[off + a] = Xs[k][lp][co + a];
      }
    }
  }
  if (rank == 0 && threadIdx.x == 0) {
    if (!good) st.rtr_stop = kPersistPoison;  // time-out: X stays untouched and the host reruns the solve
    store_state(sout, st);
    store_state(sout + 1, st);
    publish_progress(hflag, gen, st);
    ctrl->iters = iters;
    ctrl->members = (unsigned)members;
#pragma unroll
    for (int q = 0; q < 8; ++q) ctrl->ticks[q] = tk[q];
  }
}

// Second half of the hand-over of a one-launch solve (same stream, right behind k_rtr_persist): the iterate the solve left in
// the trial-point buffer becomes the caller's X -- if and only if the launch completed on EVERY participant (state record
// not poisoned, no time-out flag) and a step was accepted.  A late time-out of some workgroups therefore leaves X exactly as
// the caller passed it, whatever the others had finished.
// It also REPORTS: workgroup 0 writes the state record and the control block into the handle's host-coherent pinned copies
// (hst, hct: NULL = not wanted), which the host reads after synchronising the stream -- two copy commands (and their
// boundaries) less behind every one-launch solve.
static __global__ __launch_bounds__(kBlock) void k_persist_commit(const DevState* __restrict__ st, const PersistCtrl* __restrict__ ctrl,
                                                           const double* __restrict__ xfin, double* __restrict__ X,
                                                           size_t count, DevState* hst = nullptr, PersistCtrl* hct = nullptr) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && hst && hct) {
#define X(f) hst->f = st->f;
    DPGO_STATE_FIELDS(X)
#undef X
    hct->error = ctrl->error;
    hct->iters = ctrl->iters;
    hct->members = ctrl->members;
    hct->pad = ctrl->pad;
#pragma unroll
    for (int q = 0; q < 8; ++q) hct->ticks[q] = ctrl->ticks[q];
    __threadfence_system();
  }
  if (st->rtr_stop == kPersistPoison || ctrl->error || st->n_accept <= 0) return;
  // (8-byte pieces: a caller's device pointer is only promised to be aligned for doubles; at most 5 MB)
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += (size_t)gridDim.x * kBlock) X[i] = xfin[i];
}
