// kernels.h -- hand-written HIP kernels (gfx950 / CDNA4) for the dpgo RBCD local solve.
//
// Thread mapping ("PC layout"): one lane owns one column c of one pose tile, i.e. the R
// contiguous doubles X[i][c][0..R) of the reference layout (r x (d+1)n column-major,
// include/DPGO/manifold/Poses.h:16-21).  A 64-wide wavefront holds G = 64/(D+1) poses
// (16 for 3-D, 21 for 2-D), a 256-thread workgroup 4G poses.  A wave's loads and stores of
// a dense vector are one contiguous span (G*(D+1)*R*8 bytes).  Operations that couple the
// columns of one pose (tangent projection, Riemannian Hessian correction, block-Jacobi,
// qf retraction) exchange the pose tile through a wave-private LDS slot.
//
// Scalars never leave the device inside a solve: every kernel that produces a dot product
// writes one partial per workgroup; the NEXT kernel's prologue re-reduces those partials in
// every workgroup in a fixed order (bit-identical in all workgroups, deterministic
// run-to-run), advances the tCG / RTR scalar recurrences redundantly in registers, and
// workgroup 0 publishes the new state to the other slot of a two-slot state buffer.
// A kernel boundary (~1.5 us on MI355X) is the cheapest grid-wide barrier on this chip
// (MI355X_MICROARCH.md, price list: barrier-xcd 4-7 us).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dpgo {

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
#ifndef DPGO_LB_UPDATE
#define DPGO_LB_UPDATE 1
#endif
// Optional in-kernel timeline (diagnostic builds only, -DDPGO_TIMELINE): workgroup 0 / lane 0 stamps the 100 MHz
// wall clock at phase boundaries of the two tCG kernels into a global array read back by dpgo_debug_timeline.
#ifdef DPGO_TIMELINE
__device__ long long g_timeline[2][16];
#define DPGO_TL_DECL long long tl_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define DPGO_STAMP(K, I) tl_[I] = wall_clock64()
#define DPGO_COMMIT(K)                                     \
  do {                                                     \
    if (blockIdx.x == 0 && threadIdx.x == 0)               \
      for (int q_ = 0; q_ < 8; ++q_) g_timeline[K][q_] = tl_[q_]; \
  } while (0)
#else
#define DPGO_TL_DECL do { } while (0)
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
template <int D, int R, int SPLIT>
struct Span {
  using GEO = Geo<D, R, SPLIT>;
  static constexpr bool kOk = (GEO::T % 2 == 0);
  static constexpr int SP = GEO::G * GEO::T;  // doubles per wave span
  static constexpr int NPC = SP / 2;          // 16-byte pieces
  static constexpr int NIT = (NPC + 63) / 64; // pieces per lane
};

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

template <int D, int R, int SPLIT>
__device__ __forceinline__ void spmm_col_pre(const RowIdx& ri, const int32_t* __restrict__ colidx,
                                             const double* __restrict__ vals, const double* __restrict__ V, int s,
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
        const double* __restrict__ q = vals + (size_t)(t0 + kA) * BB + c * B;
        const double* __restrict__ x = V + (size_t)jA * T;
#pragma unroll
        for (int kk = 0; kk < B; ++kk) qa[kk] = q[kk];
#pragma unroll
        for (int e = 0; e < T; ++e) xa[e] = x[e];
      }
      if (okB) {
        const double* __restrict__ q = vals + (size_t)(t0 + kB) * BB + c * B;
        const double* __restrict__ x = V + (size_t)jB * T;
#pragma unroll
        for (int kk = 0; kk < B; ++kk) qb[kk] = q[kk];
#pragma unroll
        for (int e = 0; e < T; ++e) xb[e] = x[e];
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
      const double* __restrict__ q = vals + (size_t)(t0 + k) * BB + c * B;
      const double* __restrict__ x = V + (size_t)j * T;
      double qk[B];
#pragma unroll
      for (int kk = 0; kk < B; ++kk) qk[kk] = q[kk];
#pragma unroll
      for (int kk = 0; kk < B; ++kk) {
#pragma unroll
        for (int a = 0; a < R; ++a) acc[a] = fma(x[kk * R + a], qk[kk], acc[a]);
      }
    }
  }
  for (int t = t0 + NPRE + s; t < t1; t += SPLIT) {  // rows with more than NPRE blocks
    const int j = colidx[t];
    const double* __restrict__ q = vals + (size_t)t * BB + c * B;
    const double* __restrict__ x = V + (size_t)j * T;
    double qk[B];
#pragma unroll
    for (int kk = 0; kk < B; ++kk) qk[kk] = q[kk];
#pragma unroll
    for (int kk = 0; kk < B; ++kk) {
#pragma unroll
      for (int a = 0; a < R; ++a) acc[a] = fma(x[kk * R + a], qk[kk], acc[a]);
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

template <int D, int R, int SPLIT>
__device__ __forceinline__ void spmm_col(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                         const double* __restrict__ vals, const double* __restrict__ V,
                                         int i, int s, int c, bool ok, double (&acc)[R]) {
  const RowIdx ri = row_idx_load<D, SPLIT>(rowptr, colidx, i, s, c, ok);
  spmm_col_pre<D, R, SPLIT>(ri, colidx, vals, V, s, c, acc);
}

// ---------------------------------------------------------------- kernel arguments
struct BsrDev {
  const int32_t* rowptr;
  const int32_t* colidx;
  const double* vals;
};

// ================================================================ K1: plain SpMM
// OUT = V*Q (+ Gadd).  QuadraticProblem::EucGrad / EucHessianEta
// (src/QuadraticProblem.cpp:43-54) and, with a rectangular coupling matrix, PoseGraph::constructG
// (src/PoseGraph.cpp:493-580).
template <int D, int R, int SPLIT>
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
        for (int a = 0; a < R; ++a) acc[a] += Gadd[off + a];
      }
      store_col<R>(OUT + off, acc);
    }
  }
}

// ================================================================ K1+K2: cost + Riemannian gradient
// One pass over Q gives f(X) = 0.5<XQ,X> + <X,G> (src/QuadraticProblem.cpp:29-41),
// EG = XQ + G (:43-47), S = sym(Y^T EG_rot) (cached for the Hessian, ROPTLIB EucGradToGrad),
// RG = proj_X(EG) (:71-79) and |RG|^2 (:81-83).
// partials: [0] sum(XQ.X)  [1] sum(X.G)  [2] |RG|^2
template <int D, int R, int SPLIT>
__global__ __launch_bounds__(kBlock) void k_grad(BsrDev Q, const double* __restrict__ X,
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
  for (int tile = ti_.first; tile < ti_.last; tile += ti_.step) {
    const int i = tile * GEO::P + L.wave * GEO::G + L.g;
    const bool okp = (L.g < GEO::G) && (i < n);
    const bool ok = okp && (L.s == 0);
    double eg[R], x[R];
    const size_t off = (size_t)i * GEO::T + L.c * R;
    double* ys = ok ? &sm[L.wave][0][L.g][0] : nullptr;
    double* ws = ok ? &sm[L.wave][1][L.g][0] : nullptr;
    spmm_col<D, R, SPLIT>(Q.rowptr, Q.colidx, Q.vals, X, i, L.s, L.c, okp, eg);
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
template <int D, int R, int SPLIT>
__global__ __launch_bounds__(kBlock) void k_hess(BsrDev Q, const double* __restrict__ X,
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
  for (int tile = ti_.first; tile < ti_.last; tile += ti_.step) {
    const int i = tile * GEO::P + L.wave * GEO::G + L.g;
    const bool okp = (L.g < GEO::G) && (i < n);
    const bool ok = okp && (L.s == 0);
    double h[R], v[R], x[R];
    const size_t off = (size_t)i * GEO::T + L.c * R;
    double* ys = ok ? &sm[L.wave][0][L.g][0] : nullptr;
    double* vs = ok ? &sm[L.wave][1][L.g][0] : nullptr;
    double* hs = ok ? &sm[L.wave][2][L.g][0] : nullptr;
    spmm_col<D, R, SPLIT>(Q.rowptr, Q.colidx, Q.vals, V, i, L.s, L.c, okp, h);
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
template <int D, int R, int SPLIT>
__global__ __launch_bounds__(kBlock, DPGO_LB_HESS) void k_tcg_hess(BsrDev Q, const double* __restrict__ X,
                                                     const double* __restrict__ S, const double* __restrict__ z,
                                                     double* __restrict__ delta, double* __restrict__ Hd,
                                                     const double* __restrict__ pin, int nb_in,
                                                     double* __restrict__ pout, const DevState* __restrict__ sin,
                                                     DevState* __restrict__ sout, int first, int n,
                                                     unsigned long long* hflag, unsigned gen) {
  using GEO = Geo<D, R, SPLIT>;
  __shared__ __attribute__((aligned(16))) double sm[kWaves][4][GEO::G][GEO::T];
  __shared__ double red[kWaves * kNP];
  DevState st;
  load_state(st, sin);
  if (st.rtr_stop || st.tcg_done) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      store_state(sout, st);
      publish_progress(hflag, gen, st);
    }
    return;
  }
  double pr[2];
  load_partials<2>(pin, nb_in, pr, red);
  const double r_r = pr[0], z_r_new = pr[1];
  double beta = 0.0;
  bool go = true;
  if (first) {
    st.norm_r0 = sqrt(r_r);
    st.z_r = z_r_new;
    st.d_Pd = z_r_new;
    st.e_Pd = 0.0;
    if (st.max_inner <= 0) {
      st.tcg_done = 1;
      go = false;
    }
  } else {
    const double norm_r = sqrt(r_r);
    const double pw = (st.theta == 1.0) ? st.norm_r0 : pow(st.norm_r0, st.theta);  // theta = 1 (reference default)
    if (st.tcg_j >= st.min_inner && norm_r <= st.norm_r0 * (pw < st.kappa ? pw : st.kappa)) {
      st.tcg_status = (st.kappa < pw) ? TCG_LCON : TCG_SCON;
      st.tcg_done = 1;
      go = false;
    } else {
      beta = z_r_new / st.z_r;
      st.e_Pd = beta * (st.e_Pd + st.alpha * st.d_Pd);
      st.d_Pd = z_r_new + beta * beta * st.d_Pd;
      st.z_r = z_r_new;
      st.tcg_j += 1;
      if (st.tcg_j >= st.max_inner) {
        st.tcg_done = 1;
        st.tcg_status = TCG_MAXITER;
        go = false;
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    store_state(sout, st);
    publish_progress(hflag, gen, st);
  }
  if (!go) return;

  const LaneId L = lane_id<D, SPLIT>();
  const int ntiles = (n + GEO::P - 1) / GEO::P;
  double part[1] = {0.0};
  const TileIter ti_ = tile_iter(ntiles);
  if constexpr (Span<D, R, SPLIT>::kOk) {
    // ---- span path (T even): own-tile vectors move as 16-byte pieces through the LDS tiles; the direction /
    // H-direction recurrences run in span layout
    using SPN = Span<D, R, SPLIT>;
    const int lane = threadIdx.x & 63;
    double* ys = &sm[L.wave][0][0][0];
    double* vs = &sm[L.wave][1][0][0];
    double* hs = &sm[L.wave][2][0][0];
    double* os = &sm[L.wave][3][0][0];
    for (int tile = ti_.first; tile < ti_.last; tile += ti_.step) {
      const int p0 = tile * GEO::P + L.wave * GEO::G;
      const int npose = (n - p0) < GEO::G ? (n - p0) : GEO::G;
      const int valid = npose > 0 ? npose * GEO::T : 0;
      const size_t base = (size_t)p0 * GEO::T;
      const int i = p0 + L.g;
      const bool okp = (L.g < GEO::G) && (i < n);
      const bool ok = okp && (L.s == 0);
      const dbl2* X2 = reinterpret_cast<const dbl2*>(X + base);
      const dbl2* z2 = reinterpret_cast<const dbl2*>(z + base);
      dbl2* d2 = reinterpret_cast<dbl2*>(delta + base);
      dbl2* h2 = reinterpret_cast<dbl2*>(Hd + base);
      // issue every own-tile load before the gather: they overlap its index -> tile latency chain
      dbl2 dv[SPN::NIT], hv[SPN::NIT];
      double srow[D];
#pragma unroll
      for (int it = 0; it < SPN::NIT; ++it) {
        const int pc = lane + 64 * it;
        if (2 * pc < valid) {
          reinterpret_cast<dbl2*>(ys)[pc] = X2[pc];
          reinterpret_cast<dbl2*>(vs)[pc] = z2[pc];
          if (!first) {
            dv[it] = d2[pc];
            hv[it] = h2[pc];
          }
        }
      }
      if (ok && L.c < D) {
#pragma unroll
        for (int a = 0; a < D; ++a) srow[a] = S[(size_t)i * D * D + L.c * D + a];
      }
      double h[R];
      spmm_col<D, R, SPLIT>(Q.rowptr, Q.colidx, Q.vals, z, i, L.s, L.c, okp, h);
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
        store_col<R>(os + L.g * GEO::T + L.c * R, hz);
      }
      wave_sync();
#pragma unroll
      for (int it = 0; it < SPN::NIT; ++it) {
        const int pc = lane + 64 * it;
        if (2 * pc < valid) {
          const dbl2 zv = reinterpret_cast<const dbl2*>(vs)[pc];
          const dbl2 hzv = reinterpret_cast<const dbl2*>(os)[pc];
          dbl2 dn, hn;
          if (first) {
            dn.x = -zv.x;
            dn.y = -zv.y;
            hn.x = -hzv.x;
            hn.y = -hzv.y;
          } else {
            dn.x = fma(beta, dv[it].x, -zv.x);
            dn.y = fma(beta, dv[it].y, -zv.y);
            hn.x = fma(beta, hv[it].x, -hzv.x);
            hn.y = fma(beta, hv[it].y, -hzv.y);
          }
          d2[pc] = dn;
          h2[pc] = hn;
          part[0] = fma(dn.x, hn.x, part[0]);
          part[0] = fma(dn.y, hn.y, part[0]);
        }
      }
      wave_sync();
    }
  } else {
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
  }
  store_partials<1>(part, pout, red);
}


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

// ================================================================ span kernels (pose tile size even: all 3-D cases)
// Same arithmetic as k_tcg_hess / k_tcg_update; the differences are purely about memory:
//  * own-tile vectors move as lane-linear 16-byte pieces (Span<>), element-wise recurrences run in span layout;
//  * the FIRST tile's global loads (row pointer, column indices, vector pieces) are issued before the scalar
//    prologue (state record + partial-sum reduction), so the two dependent-latency chains overlap -- this is
//    what matters for small blocks (multi-GPU strong scaling), where a kernel is a chain of ~15 memory latencies.
template <int D, int R, int SPLIT>
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
        xv[it] = X2[pc];
        zv[it] = z2[pc];
        if (!first) {
          dv[it] = d2[pc];
          hv[it] = h2[pc];
        }
      }
    }
    if (ok && L.c < D) {
#pragma unroll
      for (int a = 0; a < D; ++a) srow[a] = S[(size_t)i * D * D + L.c * D + a];
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
          d2[pc] = dn;
          h2[pc] = hn;
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

template <int D, int R>
__global__ __launch_bounds__(kBlock) void k_tcg_update_span(const double* __restrict__ X, const double* __restrict__ g,
                                                            const double* __restrict__ dinv,
                                                            const double* __restrict__ delta,
                                                            const double* __restrict__ Hd, double* __restrict__ eta,
                                                            double* __restrict__ r, double* __restrict__ z,
                                                            const double* __restrict__ pin, int nb_in,
                                                            double* __restrict__ pout, const DevState* __restrict__ sin,
                                                            DevState* __restrict__ sout, int first, int n,
                                                            unsigned long long* hflag, unsigned gen,
                                                            double ml_omega) {
  // ml_omega > 0 (fused multilevel preconditioner): z receives the pre-smoothing step w Dinv r, unprojected
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
        xv[it] = X2[pc];
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
          reinterpret_cast<dbl2*>(ys)[pc] = xv[it];
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
        if (ml_omega > 0.0) {
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
      dbl2* z2 = reinterpret_cast<dbl2*>(z + base);
#pragma unroll
      for (int it = 0; it < SPN::NIT; ++it) {
        const int pc = lane + 64 * it;
        if (2 * pc < valid) z2[pc] = reinterpret_cast<const dbl2*>(os)[pc];
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
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int N = nc * B;
  for (int a = blockIdx.x; a < nc; a += gridDim.x) {
    if (wave < B) {
      const float* __restrict__ m = M + (size_t)(a * B + wave) * N;
      double acc[R];
#pragma unroll
      for (int q = 0; q < R; ++q) acc[q] = 0.0;
      for (int j = lane; j < N; j += 64) {
        const double mv = (double)m[j];
#pragma unroll
        for (int q = 0; q < R; ++q) acc[q] = fma(mv, rc[(size_t)j * R + q], acc[q]);
      }
#pragma unroll
      for (int q = 0; q < R; ++q) acc[q] = wave_reduce_lane63(acc[q]);
      if (lane == 63) {
#pragma unroll
        for (int q = 0; q < R; ++q) xc_s[wave][q] = acc[q];
      }
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
  using GEO = Geo<D, R>;
  __shared__ __attribute__((aligned(16))) double sm[kWaves][4][GEO::G][GEO::T];
  __shared__ double red[kWaves * kNP];
  DevState st;
  load_state(st, sin);
  if (st.rtr_stop || (!first && st.tcg_done)) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      store_state(sout, st);
      publish_progress(hflag, gen, st);
    }
    return;
  }
  int mode = 0;  // 0: normal step, 1: boundary step (eta += tau*delta, stop), 2: init
  double alpha = 0.0, tau = 0.0;
  if (first) {
    mode = 2;
    st.tcg_done = 0;
    st.tcg_j = 0;
    st.tcg_status = TCG_MAXITER;
    st.e_Pe = 0.0;
    st.e_Pd = 0.0;
  } else {
    double dh[1];
    load_partials<1>(pin, nb_in, dh, red);
    const double d_Hd = dh[0];
    alpha = st.z_r / d_Hd;
    const double e_Pe_new = st.e_Pe + 2.0 * alpha * st.e_Pd + alpha * alpha * st.d_Pd;
    st.n_hess += 1;
    st.alpha = alpha;
    const double D2 = st.Delta * st.Delta;
    if (d_Hd <= 0.0 || e_Pe_new >= D2) {
      tau = (-st.e_Pd + sqrt(st.e_Pd * st.e_Pd + st.d_Pd * (D2 - st.e_Pe))) / st.d_Pd;
      mode = 1;
      st.tcg_status = (d_Hd < 0.0) ? TCG_NEGCURV : TCG_EXCREGION;
      st.tcg_done = 1;
    } else {
      st.e_Pe = e_Pe_new;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    store_state(sout, st);
    publish_progress(hflag, gen, st);
  }

  const LaneId L = lane_id<D>();
  const int ntiles = (n + GEO::P - 1) / GEO::P;
  double part[2] = {0.0, 0.0};
  const TileIter ti_ = tile_iter(ntiles);
  if constexpr (Span<D, R, 1>::kOk) {
    // ---- span path (T even): vectors move as 16-byte pieces; r, X staged straight into the LDS tiles
    using SPN = Span<D, R, 1>;
    const int lane = threadIdx.x & 63;
    double* ys = &sm[L.wave][0][0][0];
    double* rs = &sm[L.wave][1][0][0];
    double* zs = &sm[L.wave][2][0][0];
    double* os = &sm[L.wave][3][0][0];
    for (int tile = ti_.first; tile < ti_.last; tile += ti_.step) {
      const int p0 = tile * GEO::P + L.wave * GEO::G;
      const int npose = (n - p0) < GEO::G ? (n - p0) : GEO::G;
      const int valid = npose > 0 ? npose * GEO::T : 0;  // doubles of this wave's span inside the array
      const size_t base = (size_t)p0 * GEO::T;
      const int i = p0 + L.g;
      const bool ok = (L.g < GEO::G) && (i < n);
      dbl2* eta2 = reinterpret_cast<dbl2*>(eta + base);
      const dbl2* dl2 = reinterpret_cast<const dbl2*>(delta + base);
      if (mode == 1) {  // workgroup-uniform: eta += tau * delta, then tCG stops
#pragma unroll
        for (int it = 0; it < SPN::NIT; ++it) {
          const int pc = lane + 64 * it;
          if (2 * pc < valid) {
            dbl2 e = eta2[pc];
            const dbl2 dv = dl2[pc];
            e.x = fma(tau, dv.x, e.x);
            e.y = fma(tau, dv.y, e.y);
            eta2[pc] = e;
          }
        }
        continue;
      }
      double drow[GEO::B];
      if (ok && dinv) {
#pragma unroll
        for (int k = 0; k < GEO::B; ++k) drow[k] = dinv[(size_t)i * GEO::BB + L.c * GEO::B + k];
      }
      const dbl2* X2 = reinterpret_cast<const dbl2*>(X + base);
      const dbl2* g2 = reinterpret_cast<const dbl2*>(g + base);
      const dbl2* hd2 = reinterpret_cast<const dbl2*>(Hd + base);
      dbl2* r2 = reinterpret_cast<dbl2*>(r + base);
#pragma unroll
      for (int it = 0; it < SPN::NIT; ++it) {
        const int pc = lane + 64 * it;
        if (2 * pc < valid) {
          reinterpret_cast<dbl2*>(ys)[pc] = X2[pc];
          dbl2 rv;
          if (mode == 2) {
            rv = g2[pc];
            dbl2 zero;
            zero.x = 0.0;
            zero.y = 0.0;
            eta2[pc] = zero;
          } else {
            dbl2 e = eta2[pc];
            const dbl2 dv = dl2[pc], hv = hd2[pc];
            rv = r2[pc];
            e.x = fma(alpha, dv.x, e.x);
            e.y = fma(alpha, dv.y, e.y);
            rv.x = fma(alpha, hv.x, rv.x);
            rv.y = fma(alpha, hv.y, rv.y);
            eta2[pc] = e;
          }
          r2[pc] = rv;
          reinterpret_cast<dbl2*>(rs)[pc] = rv;
          part[0] = fma(rv.x, rv.x, part[0]);
          part[0] = fma(rv.y, rv.y, part[0]);
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
        if (ml_omega > 0.0) {
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
      dbl2* z2 = reinterpret_cast<dbl2*>(z + base);
#pragma unroll
      for (int it = 0; it < SPN::NIT; ++it) {
        const int pc = lane + 64 * it;
        if (2 * pc < valid) z2[pc] = reinterpret_cast<const dbl2*>(os)[pc];
      }
      wave_sync();
    }
  } else {
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
  }
  if (mode != 1) store_partials<2>(part, pout, red);
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
      if (L.c < D) {
        double q[D][R];
#pragma unroll
        for (int k = 0; k < D; ++k) {
          if (k <= L.c) {
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
            if (k == L.c) {
#pragma unroll
              for (int t = 0; t < R; ++t) a[t] = q[k][t];
            }
          }
        }
      }
      store_col<R>(X2 + off, a);
    }
    wave_sync();
  }
}

// ================================================================ K7c: RTR acceptance test
// ROPTLIB SolversTR::Run, tail of one outer iteration: rho = (f1 - f2) / -(<eta,g> + 0.5 <eta,H eta>),
// radius update, acceptance (rho > 0.1, or the tiny-decrease clause), and on acceptance
// x1 <- x2, g1 <- g2, S1 <- S2.
// pe: k_grad partials at x2;  ph: k_hess partials for V = eta, Gdot = g1.
template <int D, int R>
__global__ __launch_bounds__(kBlock) void k_rtr_update(double* __restrict__ x1, const double* __restrict__ x2,
                                                       double* __restrict__ g1, const double* __restrict__ g2,
                                                       double* __restrict__ S1, const double* __restrict__ S2,
                                                       const double* __restrict__ pe, int nb_e,
                                                       const double* __restrict__ ph, int nb_h,
                                                       const DevState* __restrict__ sin, DevState* __restrict__ sout,
                                                       int n) {
  using GEO = Geo<D, R>;
  __shared__ double red[kWaves * kNP];
  DevState st;
  load_state(st, sin);
  if (st.rtr_stop) {
    if (blockIdx.x == 0 && threadIdx.x == 0) store_state(sout, st);
    return;
  }
  double e3[3], h2[2];
  load_partials<3>(pe, nb_e, e3, red);
  load_partials<2>(ph, nb_h, h2, red);
  const double f2 = 0.5 * e3[0] + e3[1];
  const double ngf2 = sqrt(e3[2]);
  const double eta_Heta = h2[0], eta_g = h2[1];
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
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) store_state(sout, st);
  if (!accept) return;
  const size_t total = (size_t)n * GEO::T;
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += stride) {
    x1[e] = x2[e];
    g1[e] = g2[e];
  }
  const size_t totS = (size_t)n * D * D;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < totS; e += stride) S1[e] = S2[e];
}

// RTR start: f1, |g1| from k_grad partials at x1; initial radius; stop test.
__global__ void k_rtr_begin(const double* __restrict__ pe, int nb_e, DevState* __restrict__ s0, double tol,
                            double Delta0, double Delta_max, int max_inner, int accept_tiny) {
  __shared__ double red[kWaves * kNP];
  double e3[3];
  load_partials<3>(pe, nb_e, e3, red);
  if (threadIdx.x == 0) {
    DevState st;
    st.f1 = 0.5 * e3[0] + e3[1];
    st.ngf = sqrt(e3[2]);
    st.Delta = Delta0;
    st.Delta_max = Delta_max;
    st.tol = tol;
    st.f2 = st.f1;
    st.rho = 0.0;
    st.fInit = st.f1;
    st.gnInit = st.ngf;
    st.xqx = e3[0];
    st.xg = e3[1];
    st.outer_iter = 0;
    st.rtr_stop = (st.ngf < tol) ? 1 : 0;
    st.accepted_last = 0;
    st.n_accept = 0;
    st.accept_tiny = accept_tiny;
    st.pad0 = 0;
    st.z_r = st.d_Pd = st.e_Pd = st.e_Pe = st.norm_r0 = st.alpha = 0.0;
    st.theta = 1.0;   // ROPTLIB RTRNewton default (SURVEY 8c' item 4)
    st.kappa = 0.1;
    st.tcg_j = 0;
    st.tcg_done = 0;
    st.tcg_status = TCG_MAXITER;
    st.max_inner = max_inner;
    st.n_hess = 0;
    st.min_inner = 0;
    store_state(s0, st);
    store_state(s0 + 1, st);
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
        double F[D];  // column c of F
#pragma unroll
        for (int p = 0; p < D; ++p) {
          double s = 0.0;
#pragma unroll
          for (int k = 0; k < D; ++k) {
            double wck = 0.0;
#pragma unroll
            for (int cc = 0; cc < D; ++cc) wck = (cc == L.c) ? W[cc][k] : wck;
            s += W[p][k] * wck / sqrt(Cmat[k][k]);
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

// ================================================================ K11: pack public poses
template <int D, int R>
__global__ __launch_bounds__(kBlock) void k_gather_tiles(const double* __restrict__ src,
                                                         const int32_t* __restrict__ idx, int count,
                                                         double* __restrict__ dst) {
  constexpr int T = (D + 1) * R;
  const size_t total = (size_t)count * T;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (size_t)gridDim.x * kBlock) {
    const int k = (int)(e / T), w = (int)(e - (size_t)k * T);
    dst[e] = src[(size_t)idx[k] * T + w];
  }
}

// Block-Jacobi factors: Dinv_i = (Q_ii + shift I)^-1 by Gauss-Jordan on the SPD (D+1)x(D+1) block.
template <int D>
__global__ __launch_bounds__(kBlock) void k_build_dinv(BsrDev Q, double shift, double* __restrict__ dinv, int n) {
  constexpr int B = D + 1;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    double A[B][B], I[B][B];
    bool found = false;
    for (int t = Q.rowptr[i]; t < Q.rowptr[i + 1]; ++t) {
      if (Q.colidx[t] == i) {
#pragma unroll
        for (int p = 0; p < B; ++p)
#pragma unroll
          for (int q = 0; q < B; ++q) A[p][q] = Q.vals[(size_t)t * B * B + p * B + q];
        found = true;
      }
    }
    if (!found) {
#pragma unroll
      for (int p = 0; p < B; ++p)
#pragma unroll
        for (int q = 0; q < B; ++q) A[p][q] = 0.0;
    }
#pragma unroll
    for (int p = 0; p < B; ++p)
#pragma unroll
      for (int q = 0; q < B; ++q) I[p][q] = (p == q) ? 1.0 : 0.0;
#pragma unroll
    for (int p = 0; p < B; ++p) A[p][p] += shift;
#pragma unroll
    for (int p = 0; p < B; ++p) {
      const double inv = 1.0 / A[p][p];
#pragma unroll
      for (int q = 0; q < B; ++q) {
        A[p][q] *= inv;
        I[p][q] *= inv;
      }
#pragma unroll
      for (int k = 0; k < B; ++k) {
        if (k != p) {
          const double f = A[k][p];
#pragma unroll
          for (int q = 0; q < B; ++q) {
            A[k][q] = fma(-f, A[p][q], A[k][q]);
            I[k][q] = fma(-f, I[p][q], I[k][q]);
          }
        }
      }
    }
#pragma unroll
    for (int p = 0; p < B; ++p)
#pragma unroll
      for (int q = 0; q < B; ++q) dinv[(size_t)i * B * B + p * B + q] = 0.5 * (I[p][q] + I[q][p]);
  }
}


// ================================================================ K10: edge residuals + GNC-TLS weights
// One lane per (re-weightable) edge e = (i -> j): squared residual of computeMeasurementError
// (reference src/DPGO_utils.cpp:501-507), rSq = kappa |Y_i R - Y_j|_F^2 + tau |p_j - p_i - Y_i t|^2, then
// RobustCost::weight for GNC_TLS (src/DPGO_robust.cpp:80-92, eq. (14) of the GNC paper) unless the edge has a
// fixed weight.  counts[0..2] = inliers (w > 1 - tol) / outliers (w < tol) / undecided among the non-fixed edges
// (integer atomics: exact and order-independent).
struct EdgeDev {
  const int32_t* p1;
  const int32_t* p2;
  const double* Rm;     // m x D x D, row-major per edge
  const double* t;      // m x D
  const double* kappa;
  const double* tau;
  const uint8_t* fixed;
  const uint8_t* role;   // 0 private, 1 shared outgoing (p1 mine, other pose = neighbour slot), 2 shared incoming
  const int32_t* slot;   // neighbour-tile slot of the other pose (roles 1, 2)
  double* weight;
  double* rsq;
  int m;
};

template <int D, int R>
__global__ __launch_bounds__(kBlock) void k_edge_weights(EdgeDev E, const double* __restrict__ X,
                                                         const double* __restrict__ nbr, double mu, double barc,
                                                         double w_tol, int update_weights, int* __restrict__ counts) {
  constexpr int B = D + 1, T = B * R;
  for (int e = blockIdx.x * kBlock + threadIdx.x; e < E.m; e += gridDim.x * kBlock) {
    // shared edges (PGOAgent::computeMeasurementResidual, src/PGOAgent.cpp:1048-1102): the pose owned by the
    // neighbour comes from the public-pose buffer
    const int role = E.role[e];
    const double* __restrict__ xi = (role == 2) ? nbr + (size_t)E.slot[e] * T : X + (size_t)E.p1[e] * T;
    const double* __restrict__ xj = (role == 1) ? nbr + (size_t)E.slot[e] * T : X + (size_t)E.p2[e] * T;
    const double* __restrict__ Rm = E.Rm + (size_t)e * D * D;
    const double* __restrict__ tv = E.t + (size_t)e * D;
    double rot = 0.0, tr = 0.0;
#pragma unroll
    for (int a = 0; a < R; ++a) {
#pragma unroll
      for (int c = 0; c < D; ++c) {
        double v = -xj[c * R + a];
#pragma unroll
        for (int k = 0; k < D; ++k) v = fma(xi[k * R + a], Rm[k * D + c], v);
        rot = fma(v, v, rot);
      }
      double u = xj[D * R + a] - xi[D * R + a];
#pragma unroll
      for (int k = 0; k < D; ++k) u = fma(-xi[k * R + a], tv[k], u);
      tr = fma(u, u, tr);
    }
    const double rSq0 = E.kappa[e] * rot + E.tau[e] * tr;
    E.rsq[e] = rSq0;
    if (!E.fixed[e]) {
      double w = E.weight[e];
      if (update_weights) {
        const double r = sqrt(rSq0), rSq = r * r, bSq = barc * barc;
        const double upper = (mu + 1.0) / mu * bSq, lower = mu / (mu + 1.0) * bSq;
        if (rSq >= upper) w = 0.0;
        else if (rSq <= lower) w = 1.0;
        else w = sqrt(bSq * mu * (mu + 1.0) / rSq) - mu;
        E.weight[e] = w;
      }
      if (counts && role != 2) {  // a shared edge is counted by the agent that owns its source pose
        if (w < w_tol) atomicAdd(&counts[1], 1);
        else if (w > 1.0 - w_tol) atomicAdd(&counts[0], 1);
        else atomicAdd(&counts[2], 1);
      }
    }
  }
}

// ================================================================ K9: rebuild the values of Q from edge weights
// Gather form of constructConnectionLaplacianSE (reference src/DPGO_utils.cpp:272-344): the block-CSR pattern
// is fixed by the edge list, GNC changes values only.  One lane per BSR slot sums, in a fixed order, the
// contributions of the edges incident to that slot (host-built lists):
//   kind 0: +T Om T^T (diagonal, source pose)   kind 1: +Om (diagonal, destination pose)
//   kind 2: -T Om (block (i,j))                 kind 3: -Om T^T (block (j,i))
// with T = [R t; 0 1], Om = w diag(kappa.., tau).  vals = base + sign * sum.
template <int D>
__global__ __launch_bounds__(kBlock) void k_rebuild_Q(EdgeDev E, const int32_t* __restrict__ cptr,
                                                      const int32_t* __restrict__ cedge,
                                                      const uint8_t* __restrict__ ckind,
                                                      const double* __restrict__ base, double sign,
                                                      double* __restrict__ vals, int nnzb) {
  constexpr int B = D + 1, BB = B * B;
  for (int s = blockIdx.x * kBlock + threadIdx.x; s < nnzb; s += gridDim.x * kBlock) {
    double acc[BB];
#pragma unroll
    for (int q = 0; q < BB; ++q) acc[q] = 0.0;
    for (int k = cptr[s]; k < cptr[s + 1]; ++k) {
      const int e = cedge[k];
      const int kind = ckind[k];
      double Tm[B][B], om[B];
      const double w = E.weight[e];
#pragma unroll
      for (int p = 0; p < D; ++p) {
#pragma unroll
        for (int q = 0; q < D; ++q) Tm[p][q] = E.Rm[(size_t)e * D * D + p * D + q];
        Tm[p][D] = E.t[(size_t)e * D + p];
        Tm[D][p] = 0.0;
        om[p] = w * E.kappa[e];
      }
      Tm[D][D] = 1.0;
      om[D] = w * E.tau[e];
#pragma unroll
      for (int p = 0; p < B; ++p) {
#pragma unroll
        for (int q = 0; q < B; ++q) {
          double v;
          if (kind == 0) {
            v = 0.0;
#pragma unroll
            for (int kk = 0; kk < B; ++kk) v = fma(Tm[p][kk] * om[kk], Tm[q][kk], v);
          } else if (kind == 1) {
            v = (p == q) ? om[p] : 0.0;
          } else if (kind == 2) {
            v = -Tm[p][q] * om[q];
          } else {
            v = -om[p] * Tm[q][p];
          }
          acc[p * B + q] += v;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < BB; ++q) vals[(size_t)s * BB + q] = base[(size_t)s * BB + q] + sign * acc[q];
  }
}

}  // namespace dpgo
