// kernels/common.h -- shared definitions: solver state record, thread geometry, XCD-aware tile walk, span access, reductions,
// small dense pieces (projection, block-Jacobi), the block-SpMM gather core.
// Part of kernels.h (included inside namespace dpgo, in this order: common.h, problem.h, tcg.h, persist.h, multilevel.h, dense.h, manifold.h, rtr.h, agent.h, init.h).
#pragma once


constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr int kMaxGrid = 1024;  // default launch cap: 4 workgroups per CU on 256 CUs
constexpr int kPartialCap = 1024;  // capacity of the per-workgroup partial-sum regions (upper bound of any grid)
constexpr int kNP = 4;          // partial sums per workgroup (max over kernels)

enum : int { TCG_NEGCURV = 0, TCG_EXCREGION = 1, TCG_LCON = 2, TCG_SCON = 3, TCG_MAXITER = 4 };

// Device-resident solver state (two slots; kernels read slot `in`, workgroup 0 writes `in^1`).
struct DevState {
  // --- RTR (ROPTLIB SolversTR::Run; reference configuration src/QuadraticOptimizer.cpp:64-78)
  double f1, ngf, Delta, Delta_max, tol;
  double f2, rho, fInit, gnInit;
  double xqx, xg;  // sum(XQ.X), sum(X.G) of the last k_rtr_begin evaluation
  int outer_iter, rtr_stop, accepted_last, n_accept;
  int accept_tiny, pad0;
  // --- tCG (ROPTLIB SolversTR::tCG_TR)
  double z_r, d_Pd, e_Pd, e_Pe, norm_r0, alpha, theta, kappa;
  double d_Hd;  // <delta, H delta> of the last iteration (pipelined tCG derives the next one from it)
  int tcg_j, tcg_done, tcg_status, max_inner;
  int n_hess, min_inner;
};

// Progress word published by workgroup 0 into host-coherent pinned memory (system-scope relaxed
// store).  The host feeds tCG-step kernels just-in-time, a few iterations ahead of `j`, instead of
// synchronising every few iterations; it is a HINT only -- the device state above is the truth and
// kernels enqueued after tCG finished exit in their prologue.
//   [63:32] generation (one per tCG run)   [31:8] tcg_j   [1] rtr_stop   [0] tcg_done
__device__ __forceinline__ void publish_progress(unsigned long long* hflag, unsigned gen, const DevState& st) {
  if (hflag) {
    const unsigned long long w = ((unsigned long long)gen << 32) |
                                 ((unsigned long long)((unsigned)st.tcg_j & 0xFFFFFFu) << 8) |
                                 (st.rtr_stop ? 2ull : 0ull) | (st.tcg_done ? 1ull : 0ull);
    __hip_atomic_store(hflag, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// The generation a tCG run's progress words carry.  Launches that are replayed from an instantiated hipGraph (one steady
// tCG iteration, solve.hip) cannot take it as a kernel argument -- it changes with every outer iteration -- so the record
// carries it: a launch with gen != 0 (every direct launch; the first launches of a tCG run always are) writes it into
// the record, a launch with gen == 0 reads it from there.
__device__ __forceinline__ unsigned state_gen(DevState& st, unsigned gen) {
  if (gen) st.pad0 = (int)gen;
  return (unsigned)st.pad0;
}

// Field-wise state copies: copying the whole struct by value (read, modify, write back) is lowered through
// scratch memory (120 B/lane measured with -Rpass-analysis=kernel-resource-usage), i.e. extra memory
// round trips on the critical path of every solver kernel.  Field by field it is scalar loads into SGPRs.
#define DPGO_STATE_FIELDS(X)                                                                              \
  X(f1) X(ngf) X(Delta) X(Delta_max) X(tol) X(f2) X(rho) X(fInit) X(gnInit) X(xqx) X(xg) X(outer_iter)      \
  X(rtr_stop) X(accepted_last) X(n_accept) X(accept_tiny) X(pad0) X(z_r) X(d_Pd) X(e_Pd) X(e_Pe) X(norm_r0) \
  X(alpha) X(theta) X(kappa) X(d_Hd) X(tcg_j) X(tcg_done) X(tcg_status) X(max_inner) X(n_hess) X(min_inner)
__device__ __forceinline__ void load_state(DevState& st, const DevState* __restrict__ p) {
#define X(f) st.f = p->f;
  DPGO_STATE_FIELDS(X)
#undef X
}
__device__ __forceinline__ void store_state(DevState* __restrict__ p, const DevState& st) {
#define X(f) p->f = st.f;
  DPGO_STATE_FIELDS(X)
#undef X
}

// SPLIT > 1 (SpMM kernels only): SPLIT lane groups share one pose and take every SPLIT-th block of its
// row; the partial columns are summed with log2(SPLIT) shuffles.  It shortens the dependent
// index -> tile load chain per wave (latency-bound regime: small agents / many GPUs); SPLIT = 1 is the
// throughput layout used for big blocks.
template <int D, int R, int SPLIT = 1>
struct Geo {
  static constexpr int B = D + 1;
  static constexpr int T = B * R;         // doubles per pose tile
  static constexpr int BB = B * B;        // doubles per Q block
  static constexpr int LPP = B * SPLIT;   // lanes per pose
  static constexpr int G = 64 / LPP;      // poses per wavefront
  static constexpr int P = G * kWaves;    // poses per workgroup tile
};

struct LaneId {
  int wave, g, s, c;
};
template <int D, int SPLIT = 1>
__device__ __forceinline__ LaneId lane_id() {
  constexpr int B = D + 1, LPP = B * SPLIT;
  LaneId id;
  const int l = threadIdx.x & 63;
  id.wave = threadIdx.x >> 6;
  id.g = l / LPP;
  const int lp = l - id.g * LPP;
  id.s = lp / B;
  id.c = lp - id.s * B;
  return id;
}

// ---------------------------------------------------------------- XCD-aware tile walk
// MI355X has 8 XCDs with private 4 MiB L2s; workgroup b is observed to run on XCD b % 8
// (MI355X_MICROARCH.md, "Workgroup dispatch"; used for SPEED only -- any placement is correct).
// Give each XCD one contiguous eighth of the pose tiles so the X tiles gathered by the block-SpMM
// (own rows + graph neighbours, mostly nearby indices) stay in that XCD's L2 instead of being
// fetched by all eight.  Measured with FETCH_SIZE: 156 MB -> see profiles/ per launch at 100k poses.
struct TileIter {
  int first, last, step;
};
__device__ __forceinline__ TileIter tile_iter(int ntiles) {
  TileIter it;
  const int G = gridDim.x;
  if (G < 16 || ntiles < 16) {
    it.first = blockIdx.x;
    it.last = ntiles;
    it.step = G;
    return it;
  }
  const int x = blockIdx.x & 7, lb = blockIdx.x >> 3;
  const int nbx = (G - x + 7) >> 3;  // workgroups that land on this XCD
  const int lo = (int)(((long long)ntiles * x) >> 3), hi = (int)(((long long)ntiles * (x + 1)) >> 3);
  it.first = lo + lb;
  it.last = hi;
  it.step = nbx;
  return it;
}

// The pose tiles exchanged through LDS are private to one wavefront, so a wave-level barrier (plus a
// wavefront-scope fence that orders the DS operations) replaces __syncthreads(): no cross-wave stall.
// occupancy hints (waves per SIMD) for the two kernels of the tCG loop; A/B-tuned on MI355X
#ifndef DPGO_LB_HESS
#define DPGO_LB_HESS 1
#endif
#ifndef DPGO_HESS_BATCH
#define DPGO_HESS_BATCH 4  // blocks whose loads the gather of k_tcg_hess_sym issues before the first FMA (spmm_sym_pre's NB)
#endif
#ifndef DPGO_RESTRICT_WAVES
#define DPGO_RESTRICT_WAVES 4  // waves per SIMD k_ml_restrict (one pose per D+1 lanes) is compiled for
#endif
#ifndef DPGO_GATHER_BATCH
#define DPGO_GATHER_BATCH 1  // the same for the other kernels on the symmetric storage (q_gather: restriction, one-launch solve)
#endif
#ifndef DPGO_CYCLE_SPAN
// own-tile rows of the cycle's level-0 kernels (restriction, post-smoothing) as lane-linear span pieces through the wave's LDS
// tiles, like the tCG kernels (1) or per-lane column loads / stores (0).  Built, parity-green and measured in round 6
// (profiles/r06_ab_results.txt, three interleaved pairs at 100k poses): restriction 23.9 -> 28.8 us, post-smoothing
// 22.8 -> 32.9 us, 132.8 -> 155.5 us per product -- the extra LDS round trip and wave barrier per tile sit on the tile's
// dependent chain behind the gather, and the post-smoothing kernel spills 19 instead of 10 VGPRs at its 4 waves per SIMD.
// Negative: off.
#define DPGO_CYCLE_SPAN 0
#endif
#ifndef DPGO_SYM_WAVES
#define DPGO_SYM_WAVES 2  // waves per SIMD k_tcg_hess_sym is compiled for (<= 256 VGPRs; 3 with DPGO_HESS_BATCH=1)
#endif
#ifndef DPGO_LB_UPDATE
#define DPGO_LB_UPDATE 1
#endif
// Optional in-kernel timeline (diagnostic builds only, -DDPGO_TIMELINE): workgroup 0 / lane 0 stamps the 100 MHz
// wall clock at phase boundaries of the two tCG kernels into a global array read back by dpgo_debug_timeline.
#ifdef DPGO_TIMELINE
static __device__ long long g_timeline[2][16];
// Per tile: wave 0 of the first / middle / last workgroup stamps entry [0], prologue done [1], end [2] and, for its t-th
// tile, slots 4 + 6 t + I.  k_tcg_hess_sym (g_tl_tiles; I: 0 top of the loop, 1 LDS staged, 2 gather done, 3 own pieces
// requested + sync, 4 projected, 5 stored + next tile requested), k_ml_restrict (g_tl_restrict; I: 0 top, 1 gather done,
// 2 residual staged, 3 P^T res staged, 4 run sums written, 5 tile done), k_ml_post_ap (g_tl_post; I: 0 top, 1 gather done,
// 2 own rows staged, 3 smoothed + prolonged, 4 staged again, 5 projected + stored).
static __device__ long long g_tl_tiles[3][64];
static __device__ long long g_tl_restrict[3][64];
static __device__ long long g_tl_post[3][64];
#define DPGO_TL_TILES_DECL                                                                                              \
  const int tlw_ = blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x / 2 ? 1 : (blockIdx.x == gridDim.x - 1 ? 2 : -1)); \
  int tlt_ = 0;                                                                                                         \
  const long long tle_ = wall_clock64()
#define DPGO_STAMP_ENTRY_IN(ARR) do { if (tlw_ >= 0 && threadIdx.x == 0) ARR[tlw_][0] = tle_; } while (0)  // (launches that run)
#define DPGO_STAMP_AT_IN(ARR, SLOT) do { if (tlw_ >= 0 && threadIdx.x == 0 && (SLOT) < 64) ARR[tlw_][SLOT] = wall_clock64(); } while (0)
#define DPGO_STAMP_TILE_IN(ARR, I) DPGO_STAMP_AT_IN(ARR, 4 + 6 * tlt_ + (I))
#define DPGO_STAMP_ENTRY DPGO_STAMP_ENTRY_IN(g_tl_tiles)
#define DPGO_STAMP_AT(SLOT) DPGO_STAMP_AT_IN(g_tl_tiles, SLOT)
#define DPGO_STAMP_TILE(I) DPGO_STAMP_TILE_IN(g_tl_tiles, I)
#define DPGO_TILE_NEXT ++tlt_
#define DPGO_TL_USE(X) asm volatile("" ::"v"(X))  // (a stamp behind it is not scheduled ahead of X's producers)
#define DPGO_TL_DECL long long tl_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define DPGO_STAMP(K, I) tl_[I] = wall_clock64()
#define DPGO_COMMIT(K)                                     \
  do {                                                     \
    if (blockIdx.x == 0 && threadIdx.x == 0)               \
      for (int q_ = 0; q_ < 8; ++q_) g_timeline[K][q_] = tl_[q_]; \
  } while (0)
#else
#define DPGO_TL_DECL do { } while (0)
#define DPGO_TL_TILES_DECL do { } while (0)
#define DPGO_STAMP_AT(SLOT) do { } while (0)
#define DPGO_STAMP_ENTRY do { } while (0)
#define DPGO_STAMP_ENTRY_IN(ARR) do { } while (0)
#define DPGO_STAMP_AT_IN(ARR, SLOT) do { } while (0)
#define DPGO_STAMP_TILE_IN(ARR, I) do { } while (0)
#define DPGO_STAMP_TILE(I) do { } while (0)
#define DPGO_TILE_NEXT do { } while (0)
#define DPGO_TL_USE(X) do { } while (0)
#define DPGO_STAMP(K, I) do { } while (0)
#define DPGO_COMMIT(K) do { } while (0)
#endif
#ifndef DPGO_WAVE_SYNC
#define DPGO_WAVE_SYNC 1
#endif
__device__ __forceinline__ void wave_sync() {
#if DPGO_WAVE_SYNC
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#else
  __syncthreads();
#endif
}

// ---------------------------------------------------------------- span access
// A wave's G poses are ONE contiguous span of G*T doubles whose memory layout is exactly the LDS tile
// layout [pose][column][R].  When T is even (always in 3-D: T = 4r) the span is moved with lane-linear
// 16-byte accesses (1 KiB per wave instruction) instead of 8-byte accesses at a 40-byte stride
// (tools/stream_lab.hip: 6.1 -> 7.0 TB/s on streaming kernels); element-wise updates are done in
// that "span layout" and only the per-pose coupling uses the lane = (pose, column) layout.
typedef double dbl2 __attribute__((ext_vector_type(2)));
// Operands a tCG-step kernel touches exactly once per launch (own-tile X, delta, H delta, S): with NTS = 1 they move as
// non-temporal accesses, so that the lines the gather re-uses (Q blocks referenced from later rows, z of graph neighbours)
// are not pushed out of the XCD's 4 MiB L2 by them.  Pays when the launch is fed from HBM (100k poses, every operand
// rotating: symmetric storage 48.0 -> 41.2 us, plain 48.1 -> 45.8; 1M poses: 354 -> 332 us); costs when the working set
// sits in the Infinity Cache (100k back-to-back: 33.7 -> 37.4 us), so the host selects it by size (stream_nt).
template <int NTS, typename T>
__device__ __forceinline__ T ld_stream(const T* p) {
  if constexpr (NTS) return __builtin_nontemporal_load(p);
  else return *p;
}
template <int NTS, typename T>
__device__ __forceinline__ void st_stream(T* p, T v) {
  if constexpr (NTS) __builtin_nontemporal_store(v, p);
  else *p = v;
}
template <int D, int R, int SPLIT>
struct Span {
  using GEO = Geo<D, R, SPLIT>;
  static constexpr bool kOk = (GEO::T % 2 == 0);
  static constexpr int SP = GEO::G * GEO::T;  // doubles per wave span
  static constexpr int NPC = SP / 2;          // 16-byte pieces
  static constexpr int NIT = (NPC + 63) / 64; // pieces per lane
};

// Own-tile span moves between global memory and a wave's LDS tile (layout [pose][column][R] = the memory layout): pieces of
// two entries, lane-linear (16-byte accesses for fp64, 8-byte for the fp32 storage of the cycle's vectors -- converted on the
// way, the LDS tile is always fp64).  `valid` = entries of the span that exist (ragged last tile).  Wave-cooperative.
template <int D, int R, int NTS = 0, class XT>
__device__ __forceinline__ void span_to_lds(const XT* __restrict__ src, double* __restrict__ lds, int valid) {
  using SPN = Span<D, R, 1>;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int it = 0; it < SPN::NIT; ++it) {
    const int pc = lane + 64 * it;
    if (2 * pc < valid) {
      dbl2 v;
      if constexpr (sizeof(XT) == 8) {
        v = ld_stream<NTS>(reinterpret_cast<const dbl2*>(src) + pc);
      } else {
        const float2 f = ld_stream<NTS>(reinterpret_cast<const float2*>(src) + pc);
        v.x = (double)f.x;
        v.y = (double)f.y;
      }
      reinterpret_cast<dbl2*>(lds)[pc] = v;
    }
  }
}
template <int D, int R, int NTS = 0, class XT>
__device__ __forceinline__ void span_from_lds(XT* __restrict__ dst, const double* __restrict__ lds, int valid) {
  using SPN = Span<D, R, 1>;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int it = 0; it < SPN::NIT; ++it) {
    const int pc = lane + 64 * it;
    if (2 * pc < valid) {
      const dbl2 v = reinterpret_cast<const dbl2*>(lds)[pc];
      if constexpr (sizeof(XT) == 8) {
        st_stream<NTS>(reinterpret_cast<dbl2*>(dst) + pc, v);
      } else {
        st_stream<NTS>(reinterpret_cast<float2*>(dst) + pc, make_float2((float)v.x, (float)v.y));
      }
    }
  }
}

// ---------------------------------------------------------------- reductions
__device__ __forceinline__ double wave_allreduce(double v) {
  // xor butterfly: every lane ends with the same bits (each level adds a commutative pair)
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Wave sum that lands in lane 63, built from DPP row shifts / row broadcasts (VALU lane crossing, no LDS
// round trips: six dependent steps of a few cycles each instead of six ds_bpermute round trips per 32-bit
// half).  Fixed summation tree, hence deterministic.  Lanes other than 63 hold partial sums.
#ifndef DPGO_DPP_REDUCE
#define DPGO_DPP_REDUCE 1
#endif
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_shifted(double v) {
  // lanes without a source (or masked off) receive 0.0
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, BANK_MASK, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, BANK_MASK, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_reduce_lane63(double v) {
  double s = v;
  s += dpp_shifted<0x111, 0xf, 0xf>(v);  // row_shr:1
  s += dpp_shifted<0x112, 0xf, 0xf>(v);  // row_shr:2
  s += dpp_shifted<0x113, 0xf, 0xf>(v);  // row_shr:3   -> s[i] = v[i-3..i] within a row of 16
  s += dpp_shifted<0x114, 0xf, 0xe>(s);  // row_shr:4, banks 1..3
  s += dpp_shifted<0x118, 0xf, 0xc>(s);  // row_shr:8, banks 2..3 -> lane 15 of each row holds the row sum
  s += dpp_shifted<0x142, 0xa, 0xf>(s);  // row_bcast:15 into rows 1 and 3
  s += dpp_shifted<0x143, 0xc, 0xf>(s);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
  return s;
}

// Wave sums of N values at once (N a multiple of 4; the callers pad with zeros): a reduce-scatter over the four rows of
// the wavefront instead of N full reductions.  gfx950's v_permlane32_swap / v_permlane16_swap exchange half of one
// register with the other half of a second one, so a PAIR of values is folded over the two halves of the wave (then over
// the two rows of each half) by two swaps and one addition: N values -> N/2 -> N/4, each now the sum over all four rows of
// ONE of the original values, which one depending on the lane's row; five DPP row shifts finish every survivor inside
// its row.  N = 20: 15 pair steps + 5 row reductions, about a quarter of the instructions of 20 x wave_reduce_lane63.
// Result: lane 15 of row q (lane 16 q + 15) holds in out[j] the wave sum of v[4 j + kRowValue[q]].  Fixed tree.
__device__ __forceinline__ void swap_fold32(double x, double y, double& s) {  // s: lanes 0..31 sum x's halves, 32..63 y's
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  s = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ void swap_fold16(double x, double y, double& s) {  // s: even rows sum x's row pair, odd rows y's
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  s = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ int wave_rows_value(int lane) {  // kRowValue[row of the lane]: rows 0..3 -> 0, 2, 1, 3
  const int q = lane >> 4;
  return ((q & 1) << 1) | (q >> 1);
}
template <int N>
__device__ __forceinline__ void wave_reduce_rows(const double (&v)[N], double (&out)[N / 4]) {
  static_assert(N % 4 == 0, "pad to a multiple of 4");
  double w[N / 2];
#pragma unroll
  for (int i = 0; i < N / 2; ++i) swap_fold32(v[2 * i], v[2 * i + 1], w[i]);  // half h holds value 2 i + h
#pragma unroll
  for (int j = 0; j < N / 4; ++j) {
    double s;
    swap_fold16(w[2 * j], w[2 * j + 1], s);  // row parity p holds w[2 j + p]: value 4 j + 2 p + h
    double t = s;
    t += dpp_shifted<0x111, 0xf, 0xf>(s);  // row_shr:1
    t += dpp_shifted<0x112, 0xf, 0xf>(s);  // row_shr:2
    t += dpp_shifted<0x113, 0xf, 0xf>(s);  // row_shr:3
    t += dpp_shifted<0x114, 0xf, 0xe>(t);  // row_shr:4, banks 1..3
    t += dpp_shifted<0x118, 0xf, 0xc>(t);  // row_shr:8, banks 2..3 -> lane 15 of the row holds the sum
    out[j] = t;
  }
}

template <int K>
__device__ __forceinline__ void block_allreduce(double (&v)[K], double* red /* >= kWaves*K */) {
#if DPGO_DPP_REDUCE
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = wave_reduce_lane63(v[k]);
  __syncthreads();
  if ((threadIdx.x & 63) == 63) {
#pragma unroll
    for (int k = 0; k < K; ++k) red[(threadIdx.x >> 6) * K + k] = v[k];
  }
#else
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = wave_allreduce(v[k]);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) red[(threadIdx.x >> 6) * K + k] = v[k];
  }
#endif
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double s = red[k];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) s += red[w * K + k];
    v[k] = s;
  }
}

// Sum the per-workgroup partials of the previous kernel; identical result in every thread
// of every workgroup.
template <int K>
__device__ __forceinline__ void load_partials(const double* __restrict__ p, int nb, double (&out)[K],
                                              double* red) {
#pragma unroll
  for (int k = 0; k < K; ++k) out[k] = 0.0;
  for (int i = threadIdx.x; i < nb; i += kBlock) {
#pragma unroll
    for (int k = 0; k < K; ++k) out[k] += p[i * kNP + k];
  }
  block_allreduce<K>(out, red);
}

// Two-phase variant for latency-bound launches: the global loads are issued early (together with the other
// independent loads of the kernel prologue) and reduced later.
constexpr int kPartialTrips = kPartialCap / kBlock;
template <int K>
struct PartialRaw {
  double v[kPartialTrips][K];
};
template <int K>
__device__ __forceinline__ void partials_issue(const double* __restrict__ p, int nb, PartialRaw<K>& raw) {
#pragma unroll
  for (int t = 0; t < kPartialTrips; ++t) {
    const int i = threadIdx.x + t * kBlock;
#pragma unroll
    for (int k = 0; k < K; ++k) raw.v[t][k] = (i < nb) ? p[i * kNP + k] : 0.0;
  }
}
template <int K>
__device__ __forceinline__ void partials_finish(const PartialRaw<K>& raw, double (&out)[K], double* red) {
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double a = raw.v[0][k];
#pragma unroll
    for (int t = 1; t < kPartialTrips; ++t) a += raw.v[t][k];  // same order as load_partials
    out[k] = a;
  }
  block_allreduce<K>(out, red);
}

template <int K>
__device__ __forceinline__ void store_partials(double (&v)[K], double* __restrict__ p, double* red) {
  block_allreduce<K>(v, red);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) p[blockIdx.x * kNP + k] = v[k];
  }
}

// ---------------------------------------------------------------- small dense pieces
// Tangent projection of column c of W at Y (ROPTLIB Stiefel::ExtrProjection; the Euclidean
// factor -- column D -- is untouched).  ys / ws: pose tiles in LDS ([col][R]).
// Optionally returns s[a] = sym(Y^T W)[a][c].
template <int D, int R>
__device__ __forceinline__ void proj_col(const double* ys, const double* ws, int c, const double (&w)[R],
                                         double (&out)[R], double (&s)[D]) {
  if (c < D) {
#pragma unroll
    for (int a = 0; a < D; ++a) {
      double p = 0.0, q = 0.0;
#pragma unroll
      for (int k = 0; k < R; ++k) {
        p = fma(ys[a * R + k], ws[c * R + k], p);
        q = fma(ws[a * R + k], ys[c * R + k], q);
      }
      s[a] = 0.5 * (p + q);
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
      double v = w[k];
#pragma unroll
      for (int a = 0; a < D; ++a) v = fma(-ys[a * R + k], s[a], v);
      out[k] = v;
    }
  } else {
#pragma unroll
    for (int k = 0; k < R; ++k) out[k] = w[k];
#pragma unroll
    for (int a = 0; a < D; ++a) s[a] = 0.0;
  }
}

// Block-Jacobi: z[:,c] = sum_k v[:,k] * Dinv[k][c]   (Dinv symmetric; lane reads row c)
template <int D, int R>
__device__ __forceinline__ void jacobi_col(const double* vs /* LDS tile */, const double* __restrict__ dinv_row,
                                           double (&z)[R]) {
  constexpr int B = D + 1;
#pragma unroll
  for (int a = 0; a < R; ++a) z[a] = 0.0;
#pragma unroll
  for (int k = 0; k < B; ++k) {
    const double dk = dinv_row[k];
#pragma unroll
    for (int a = 0; a < R; ++a) z[a] = fma(vs[k * R + a], dk, z[a]);
  }
}

// Column c of the qf retraction of one pose (ROPTLIB Stiefel::qfRetraction: Q factor of the thin QR of the r x d block
// with diag(R) > 0, by modified Gram-Schmidt): as = the pose's tile [col][R] in LDS holding Y + eta; lane c < D rebuilds
// q_0 .. q_c (identical arithmetic in all lanes of the pose) and returns q_c in a; the Euclidean column (c = D) keeps a.
template <int D, int R>
__device__ __forceinline__ void qf_col(const double* as, int c, double (&a)[R]) {
  if (c < D) {
    double q[D][R];
#pragma unroll
    for (int k = 0; k < D; ++k) {
      if (k <= c) {
        double v[R];
#pragma unroll
        for (int t = 0; t < R; ++t) v[t] = as[k * R + t];
#pragma unroll
        for (int l = 0; l < D; ++l) {
          if (l < k) {
            double dp = 0.0;
#pragma unroll
            for (int t = 0; t < R; ++t) dp = fma(q[l][t], v[t], dp);
#pragma unroll
            for (int t = 0; t < R; ++t) v[t] = fma(-dp, q[l][t], v[t]);
          }
        }
        double nn = 0.0;
#pragma unroll
        for (int t = 0; t < R; ++t) nn = fma(v[t], v[t], nn);
        const double inv = 1.0 / sqrt(nn);
#pragma unroll
        for (int t = 0; t < R; ++t) q[k][t] = v[t] * inv;
        if (k == c) {
#pragma unroll
          for (int t = 0; t < R; ++t) a[t] = q[k][t];
        }
      }
    }
  }
}

template <int R>
__device__ __forceinline__ void load_col(const double* __restrict__ p, double (&v)[R]) {
#pragma unroll
  for (int a = 0; a < R; ++a) v[a] = p[a];
}
template <int R>
__device__ __forceinline__ void store_col(double* __restrict__ p, const double (&v)[R]) {
#pragma unroll
  for (int a = 0; a < R; ++a) p[a] = v[a];
}
template <int R, class XT>
__device__ __forceinline__ void load_col_t(const XT* __restrict__ p, double (&v)[R]) {
#pragma unroll
  for (int a = 0; a < R; ++a) v[a] = (double)p[a];
}
template <int R, class XT>
__device__ __forceinline__ void store_col_t(XT* __restrict__ p, const double (&v)[R]) {
#pragma unroll
  for (int a = 0; a < R; ++a) p[a] = (XT)v[a];
}
template <int NTS, int R>
__device__ __forceinline__ void store_col_stream(double* __restrict__ p, const double (&v)[R]) {
#pragma unroll
  for (int a = 0; a < R; ++a) st_stream<NTS>(p + a, v[a]);
}

// ---------------------------------------------------------------- block-SpMM core
// acc[:] = (V*Q)[i][c][:] = sum_j sum_k V_j[:,k] * Q[i,j][c][k]      (Q symmetric)
// replaces Eigen's dense x RowMajor-sparse product in src/QuadraticProblem.cpp:33,39,46,53.
//
// Wave-cooperative: must be called by ALL 64 lanes (lanes without a row pass ok = false).  The B
// lanes of a pose preload the row's first 2B column indices (one coalesced load each) and broadcast
// them with ds_bpermute, which removes the dependent colidx -> tile load from every iteration of the
// gather loop (the kernel is bound by that latency chain, not by HBM: tools/spmm_lab.hip, 27.2 -> 24.6 us
// at 100k poses).  Each lane streams row c of the Q block (32 B for D = 3: the quad reads the 128-B
// block exactly once, coalesced) and the full gathered tile V_j (160 B, L2-resident).
struct RowIdx {
  int t0, deg, ja, jb;
};
// Row pointer + preloaded column indices of pose i (wave-cooperative: call with all 64 lanes).
template <int D, int SPLIT>
__device__ __forceinline__ RowIdx row_idx_load(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                               int i, int s, int c, bool ok) {
  constexpr int B = D + 1, LPP = B * SPLIT;
  constexpr int NJ = (SPLIT == 1) ? 2 : 1;
  RowIdx ri;
  const int lp = s * B + c;
  ri.t0 = ok ? rowptr[i] : 0;
  const int t1 = ok ? rowptr[i + 1] : 0;
  ri.deg = t1 - ri.t0;
  ri.ja = (lp < ri.deg) ? colidx[ri.t0 + lp] : 0;
  ri.jb = (NJ == 2 && lp + LPP < ri.deg) ? colidx[ri.t0 + lp + LPP] : 0;
  return ri;
}

// LD: 0 = plain, 1 = nontemporal, 2 = agent-scope relaxed atomic (global_load sc1: coherent with write-through stores of
// workgroups on other XCDs -- the persistent tCG kernel, where V is rewritten by the other workgroups of the SAME launch)
template <int LD>
__device__ __forceinline__ double ld_tile(const double* __restrict__ p) {
  if constexpr (LD == 2) {
    const unsigned long long b =
        __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __longlong_as_double((long long)b);
  } else if constexpr (LD == 1) {
    return __builtin_nontemporal_load(p);
  } else {
    return *p;
  }
}
// VT: storage type of the block values (double; float for the reduced-precision operator copies of the multilevel cycle --
// every product and sum stays fp64)
template <int D, int R, int SPLIT, int NT = 0, class VT = double>
__device__ __forceinline__ void spmm_col_pre(const RowIdx& ri, const int32_t* __restrict__ colidx,
                                             const VT* __restrict__ vals, const double* __restrict__ V, int s,
                                             int c, double (&acc)[R]) {
  constexpr int B = D + 1, T = B * R, BB = B * B, LPP = B * SPLIT;
  constexpr int NJ = (SPLIT == 1) ? 2 : 1;   // preloaded indices per lane
  constexpr int NPRE = NJ * LPP;             // preloaded indices per pose (2B for SPLIT = 1)
#pragma unroll
  for (int a = 0; a < R; ++a) acc[a] = 0.0;
  const int lane = threadIdx.x & 63;
  const int lp = s * B + c;
  const int gbase = lane - lp;
  const int t0 = ri.t0, deg = ri.deg, t1 = ri.t0 + ri.deg;
  const int ja = ri.ja, jb = ri.jb;
  int maxdeg = deg;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) maxdeg = max(maxdeg, __shfl_xor(maxdeg, o));
  const int kmax = maxdeg < NPRE ? maxdeg : NPRE;
  if constexpr (SPLIT > 1) {
    // latency layout: the loads of two blocks are in flight together (same FMA order as the plain loop)
    for (int k0 = 0; k0 < kmax; k0 += 2 * SPLIT) {
      const int kA = k0 + s, kB = k0 + SPLIT + s;
      const int jA = __shfl(ja, gbase + (kA < LPP ? kA : 0));
      const int jB = __shfl(ja, gbase + (kB < LPP ? kB : 0));
      const bool okA = kA < deg && kA < NPRE, okB = kB < deg && kB < NPRE;
      double qa[B], qb[B], xa[T], xb[T];
      if (okA) {
        const VT* __restrict__ q = vals + (size_t)(t0 + kA) * BB + c * B;
        const double* __restrict__ x = V + (size_t)jA * T;
#pragma unroll
        for (int kk = 0; kk < B; ++kk) qa[kk] = (double)q[kk];
#pragma unroll
        for (int e = 0; e < T; ++e) xa[e] = ld_tile<NT>(x + e);
      }
      if (okB) {
        const VT* __restrict__ q = vals + (size_t)(t0 + kB) * BB + c * B;
        const double* __restrict__ x = V + (size_t)jB * T;
#pragma unroll
        for (int kk = 0; kk < B; ++kk) qb[kk] = (double)q[kk];
#pragma unroll
        for (int e = 0; e < T; ++e) xb[e] = ld_tile<NT>(x + e);
      }
      if (okA) {
#pragma unroll
        for (int kk = 0; kk < B; ++kk) {
#pragma unroll
          for (int a = 0; a < R; ++a) acc[a] = fma(xa[kk * R + a], qa[kk], acc[a]);
        }
      }
      if (okB) {
#pragma unroll
        for (int kk = 0; kk < B; ++kk) {
#pragma unroll
          for (int a = 0; a < R; ++a) acc[a] = fma(xb[kk * R + a], qb[kk], acc[a]);
        }
      }
    }
  } else
  for (int k0 = 0; k0 < kmax; k0 += SPLIT) {
    const int k = k0 + s;  // this slice's block
    const int src = (k < LPP) ? k : k - LPP;
    const int j = __shfl((NJ == 2 && k >= LPP) ? jb : ja, gbase + (src < LPP ? src : 0));
    if (k < deg && k < NPRE) {
      const VT* __restrict__ q = vals + (size_t)(t0 + k) * BB + c * B;
      const double* __restrict__ x = V + (size_t)j * T;
      double qk[B];
#pragma unroll
      for (int kk = 0; kk < B; ++kk) qk[kk] = (double)q[kk];
#pragma unroll
      for (int kk = 0; kk < B; ++kk) {
#pragma unroll
        for (int a = 0; a < R; ++a) acc[a] = fma(ld_tile<NT>(x + kk * R + a), qk[kk], acc[a]);
      }
    }
  }
  for (int t = t0 + NPRE + s; t < t1; t += SPLIT) {  // rows with more than NPRE blocks
    const int j = colidx[t];
    const VT* __restrict__ q = vals + (size_t)t * BB + c * B;
    const double* __restrict__ x = V + (size_t)j * T;
    double qk[B];
#pragma unroll
    for (int kk = 0; kk < B; ++kk) qk[kk] = (double)q[kk];
#pragma unroll
    for (int kk = 0; kk < B; ++kk) {
#pragma unroll
      for (int a = 0; a < R; ++a) acc[a] = fma(ld_tile<NT>(x + kk * R + a), qk[kk], acc[a]);
    }
  }
  if (SPLIT > 1) {  // fixed-order tree over the slices; the sum lands in slice 0
#pragma unroll
    for (int o = SPLIT / 2; o >= 1; o >>= 1) {
#pragma unroll
      for (int a = 0; a < R; ++a) acc[a] += __shfl_down(acc[a], o * B);
    }
  }
}

template <int D, int R, int SPLIT, int NT = 0, class VT = double>
__device__ __forceinline__ void spmm_col(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                         const VT* __restrict__ vals, const double* __restrict__ V,
                                         int i, int s, int c, bool ok, double (&acc)[R]) {
  const RowIdx ri = row_idx_load<D, SPLIT>(rowptr, colidx, i, s, c, ok);
  spmm_col_pre<D, R, SPLIT, NT>(ri, colidx, vals, V, s, c, acc);
}

// ---------------------------------------------------------------- kernel arguments
template <class VT>
struct BsrDevT {
  const int32_t* rowptr;
  const int32_t* colidx;
  const VT* vals;
};
using BsrDev = BsrDevT<double>;
using BsrDev32 = BsrDevT<float>;  // fp32 copy of the values (A P of the multilevel cycle, opt-in)

// ---------------------------------------------------------------- reduce-scatter over the lanes of a pose
// Every lane (pose, column c) holds a (D+1) x R partial acc[cc][a]; afterwards out[a] = sum over the pose's lanes of their
// acc[c][a], i.e. lane c keeps row c of the sum.  D = 3: butterfly inside the quad with DPP quad_perm (no LDS crossbar):
// after the xor-1 step a lane holds rows (c & 1) and 2 + (c & 1) summed over its pair, after the xor-2 step row c summed
// over the quad.  D = 2: three lanes per pose, shuffles.  Wave-cooperative (all 64 lanes).
template <int D, int R>
__device__ __forceinline__ void pose_reduce_scatter(const double (&acc)[D + 1][R], int c, double (&out)[R]) {
  constexpr int B = D + 1;
  if constexpr (B == 4) {
    const bool p1 = c & 1, p2 = c & 2;
#pragma unroll
    for (int a = 0; a < R; ++a) {
      const double r01 = (p1 ? acc[1][a] : acc[0][a]) + dpp_shifted<0xB1, 0xf, 0xf>(p1 ? acc[0][a] : acc[1][a]);
      const double r23 = (p1 ? acc[3][a] : acc[2][a]) + dpp_shifted<0xB1, 0xf, 0xf>(p1 ? acc[2][a] : acc[3][a]);
      out[a] = (p2 ? r23 : r01) + dpp_shifted<0x4E, 0xf, 0xf>(p2 ? r01 : r23);
    }
  } else {
    const int gbase = (int)(threadIdx.x & 63) - c;
#pragma unroll
    for (int cc = 0; cc < B; ++cc)
#pragma unroll
      for (int a = 0; a < R; ++a) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < B; ++k) v += __shfl(acc[cc][a], gbase + k);
        if (cc == c) out[a] = v;
      }
  }
}

// ---------------------------------------------------------------- block product on symmetric storage
// Q is symmetric, Q[j,i] = Q[i,j]^T.  Only the blocks (i, j >= i) are stored, TRANSPOSED (column c of a block is
// contiguous); block row i walks its upper blocks and a list of references (j < i, slot of block (j, i)) whose
// stored block is used through its rows.  HBM sees ~half of Q's values, and the gather is by outer products: lane c of
// a pose loads only column c of the gathered tile (R doubles instead of the whole (D+1) x R tile) and of the block, keeps
// the (D+1) x R partial  P_c[c'][a] = V_j[a][c] Q_ij[c'][c], and ONE cross-lane reduce-scatter per row (not per block)
// leaves row c of (Q V)_i in lane c.  The first (D+1) upper and lower column indices of a row are preloaded by the pose's
// lanes and broadcast by shuffles.  One pose per (D+1) lanes (SPLIT = 1) only.  100k-pose grid, plain product: 28.4 us
// against 36.6 us with Infinity-Cache-cold operands, 22.9 against 24.4 us warm (profiles/).
template <class VT>
struct BsrSymDevT {
  const int32_t* urow;   // [n + 1] upper blocks (j >= i) of every block row
  const int32_t* ucol;
  const VT* uvalsT;      // transposed blocks: uvalsT[u][p][q] = Q[i, j][q][p]
  const int32_t* lrow;   // [n + 1] lower references (j < i)
  const int32_t* lcol;
  const int32_t* lslot;  // upper slot of block (j, i)
  // The WALK over the workgroup tiles (NULL: index order): tord[k] = the tile processed k-th, a permutation INSIDE each XCD's
  // contiguous eighth (tile_iter): breadth-first over the tile graph, so that the tiles one XCD processes at the same time
  // are graph neighbours -- a row's lower references and the z tiles it gathers are then in flight in the same L2 instead
  // of being fetched once per round (host: sym_symbolic_setup; the pose order, i.e. the data layout, is untouched).
  const int32_t* tord;
};
// the k-th tile of a kernel's walk over storage A
template <class VT>
__device__ __forceinline__ int tile_of(const BsrDevT<VT>&, int k) { return k; }
template <class VT>
__device__ __forceinline__ int tile_of(const BsrSymDevT<VT>& A, int k) { return A.tord ? A.tord[k] : k; }
using BsrSymDev = BsrSymDevT<double>;
using BsrSymDev32 = BsrSymDevT<float>;  // fp32 copy of the values (level-0 restriction of the multilevel cycle, opt-in)
struct SymIdx {
  int u0, du, l0, dl, ju, jl, sl;
};
template <int D, class VT>
__device__ __forceinline__ SymIdx sym_idx_load(const BsrSymDevT<VT>& Q, int i, int c, bool ok) {
  SymIdx si;
  si.u0 = ok ? Q.urow[i] : 0;
  si.du = (ok ? Q.urow[i + 1] : 0) - si.u0;
  si.l0 = ok ? Q.lrow[i] : 0;
  si.dl = (ok ? Q.lrow[i + 1] : 0) - si.l0;
  si.ju = (c < si.du) ? Q.ucol[si.u0 + c] : 0;
  si.jl = (c < si.dl) ? Q.lcol[si.l0 + c] : 0;
  si.sl = (c < si.dl) ? Q.lslot[si.l0 + c] : 0;
  return si;
}
// wave-cooperative (all 64 lanes); out = row c of (Q V)_i for the lane (g, c) of pose i
// XT: storage type of the gathered vector (double; float for the cycle-internal vectors kept in fp32)
// NB: blocks whose loads are in flight together (registers: NB (D+1+R) doubles)
template <int D, int R, int NB_ = 1, class VT, class XT = double>
__device__ __forceinline__ void spmm_sym_pre(const SymIdx& si, const BsrSymDevT<VT>& Q, const XT* __restrict__ V, int c,
                                             double (&out)[R]) {
  constexpr int B = D + 1, T = B * R, BB = B * B;
  const int lane = threadIdx.x & 63, gbase = lane - c;
  double acc[B][R];
#pragma unroll
  for (int cc = 0; cc < B; ++cc)
#pragma unroll
    for (int a = 0; a < R; ++a) acc[cc][a] = 0.0;
  const int u0 = si.u0, du = si.du, l0 = si.l0, dl = si.dl;
  int mu = du, ml = dl;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    mu = max(mu, __shfl_xor(mu, o));
    ml = max(ml, __shfl_xor(ml, o));
  }
  const int lu = mu < B ? mu : B, ll = ml < B ? ml : B;
  auto fma_block = [&](const double (&q)[B], int j) {
    double xc[R];
#pragma unroll
    for (int a = 0; a < R; ++a) xc[a] = (double)V[(size_t)j * T + c * R + a];
#pragma unroll
    for (int cc = 0; cc < B; ++cc)
#pragma unroll
      for (int a = 0; a < R; ++a) acc[cc][a] = fma(xc[a], q[cc], acc[cc][a]);
  };
  // The first B upper blocks and lower references: all addresses are known up front (u0 + k, and the preloaded j / slot),
  // so the loads of NB blocks are issued back to back before the first FMA -- one memory round trip per NB blocks instead of
  // one per block (the accumulation order is unchanged: results are bit-identical for every NB).
  constexpr int NB = NB_ < B ? NB_ : B;
  for (int k0 = 0; k0 < lu; k0 += NB) {  // upper blocks: column c of Q[i,j] = row c of the transposed storage
    double q[NB][B], xc[NB][R];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int k = k0 + b;
      const int j = __shfl(si.ju, gbase + (k < B ? k : 0));
      if (k < B && k < du) {
#pragma unroll
        for (int pp = 0; pp < B; ++pp) q[b][pp] = (double)Q.uvalsT[(size_t)(u0 + k) * BB + c * B + pp];
#pragma unroll
        for (int a = 0; a < R; ++a) xc[b][a] = (double)V[(size_t)j * T + c * R + a];
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int k = k0 + b;
      if (k < B && k < du) {
#pragma unroll
        for (int cc = 0; cc < B; ++cc)
#pragma unroll
          for (int a = 0; a < R; ++a) acc[cc][a] = fma(xc[b][a], q[b][cc], acc[cc][a]);
      }
    }
  }
  for (int t = u0 + B; t < u0 + du; ++t) {
    double q[B];
#pragma unroll
    for (int pp = 0; pp < B; ++pp) q[pp] = (double)Q.uvalsT[(size_t)t * BB + c * B + pp];
    fma_block(q, Q.ucol[t]);
  }
  for (int k0 = 0; k0 < ll; k0 += NB) {  // lower references: column c of Q[i,j] = row c of Q[j,i] = strided in its storage
    double q[NB][B], xc[NB][R];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int k = k0 + b;
      const int j = __shfl(si.jl, gbase + (k < B ? k : 0));
      const int sb = __shfl(si.sl, gbase + (k < B ? k : 0));
      if (k < B && k < dl) {
#pragma unroll
        for (int pp = 0; pp < B; ++pp) q[b][pp] = (double)Q.uvalsT[(size_t)sb * BB + pp * B + c];
#pragma unroll
        for (int a = 0; a < R; ++a) xc[b][a] = (double)V[(size_t)j * T + c * R + a];
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int k = k0 + b;
      if (k < B && k < dl) {
#pragma unroll
        for (int cc = 0; cc < B; ++cc)
#pragma unroll
          for (int a = 0; a < R; ++a) acc[cc][a] = fma(xc[b][a], q[b][cc], acc[cc][a]);
      }
    }
  }
  for (int t = l0 + B; t < l0 + dl; ++t) {
    double q[B];
    const size_t sb = (size_t)Q.lslot[t];
#pragma unroll
    for (int pp = 0; pp < B; ++pp) q[pp] = (double)Q.uvalsT[sb * BB + pp * B + c];
    fma_block(q, Q.lcol[t]);
  }
  pose_reduce_scatter<D, R>(acc, c, out);
}

// storage-generic row gather: h = row c of (A V)_i for the lane (g, s, c) of node i
template <int D, int R, int SPLIT, class VT>
__device__ __forceinline__ void q_gather(const BsrDevT<VT>& A, const double* __restrict__ V, int i, int s, int c, bool okp,
                                         double (&h)[R]) {
  spmm_col<D, R, SPLIT>(A.rowptr, A.colidx, A.vals, V, i, s, c, okp, h);
}
template <int D, int R, int SPLIT, class VT, class XT>
__device__ __forceinline__ void q_gather(const BsrSymDevT<VT>& A, const XT* __restrict__ V, int i, int s, int c, bool okp,
                                         double (&h)[R]) {
  static_assert(SPLIT == 1, "symmetric storage: one node per D+1 lanes");
  const SymIdx si = sym_idx_load<D>(A, i, c, okp);
  spmm_sym_pre<D, R, DPGO_GATHER_BATCH>(si, A, V, c, h);
}

