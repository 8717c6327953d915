// kernels/multilevel.h -- aggregation-multigrid preconditioner (the device path's default): per-iteration cycle kernels and
// the on-device setup of the hierarchy (prolongation blocks, Galerkin operators, dense inverse of the coarsest operator).
// Part of kernels.h (included inside namespace dpgo, in this order: common.h, problem.h, tcg.h, persist.h, multilevel.h, dense.h, manifold.h, rtr.h, agent.h, init.h).
#pragma once

// ================================================================ multilevel preconditioner
// Replaces the exact CHOLMOD solve of Q + 0.1 I inside QuadraticProblem::PreConditioner (src/QuadraticProblem.cpp:56-69;
// factor built by PoseGraph::constructPreconditioner, src/PoseGraph.cpp:598-613) by one V(1,1) cycle for A_0 = Q + shift I:
//   level l:  x1 = w Dinv_l r_l;  r_{l+1} = P_l^T (r_l - A_l x1);  xc = cycle_{l+1}(r_{l+1});  x = x1 + P_l xc;
//             z_l = x + w Dinv_l (r_l - A_l x);        coarsest level: z = A_L^-1 r_L (dense inverse in HBM)
// followed by the tangent projection (:68).  Level l+1's nodes are runs of k_l consecutive level-l nodes; P_l's blocks are
// relative poses composed along the odometry chain (k_ml_build_P), A_{l+1} = P_l^T A_l P_l (k_ml_galerkin).
// Launches per tCG iteration for L levels (L-1 coarsenings): k_tcg_hess | k_tcg_update (writes x1 of level 0) |
// k_ml_restrict x (L-1) | k_ml_coarse_prolong | k_ml_post_mid x (L-2) | k_ml_post  =  2L + 1.
// `gate`: the solver's state record -- launches enqueued after tCG finished return at once.

// Which aggregate (node of the next level) a node belongs to: runs of k consecutive nodes, or -- level 0 of a two-level
// hierarchy with GRAPH aggregates (host: ml_graph_aggregates) -- a label per pose.
struct AggMap {
  const int32_t* lab;
  int k;
  __device__ __forceinline__ int of(int i) const { return lab ? lab[i] : i / k; }
};

// x1 = w Dinv v (stand-alone pre-smoothing step; inside the tCG loop k_tcg_update produces it)
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

// Restriction of level l:  rc = P^T (r - (A + shift I) x1)  in one pass over A (needs aggregates that do not straddle
// workgroup tiles: GEO::P % k == 0).  With `dinv_next` (the next level is not the dense one) the pre-smoothing step of
// level l+1, x1c = w Dinv_{l+1} rc, rides in the epilogue.  `rc32` (the next level is the dense one and its inverse is
// stored in fp32): rc is written in fp32 instead.  `res_out`: the residual r - A x1 itself is kept (k_ml_post_ap).
// `stop` (inside the tCG loop, level 0): the LAST workgroup -- the one with the fewest tiles -- also evaluates tCG's
// residual test |r| <= |r0| min(|r0|^theta, kappa) from the <r,r> partial sums k_tcg_update just wrote, i.e. one kernel
// earlier than the Hessian-step kernel's prologue would (tcg_hess_prologue: same test, same state fields).  When it
// holds it raises tcg_done in the state record the rest of the cycle and the next Hessian-step kernel are gated on and
// publishes it to the host: the final iteration of a converged tCG run no longer pays for a dense solve and a
// post-smoothing pass whose result nobody reads, and the host learns one cycle sooner that it can enqueue the outer
// iteration's launches (what it had already enqueued of the next tCG iteration exits in its prologues, as before).
struct TcgStopCheck {
  DevState* state = nullptr;  // the record `gate` points to; NULL: no check
  const double* pin = nullptr;
  int nb = 0;
  unsigned long long* hflag = nullptr;
  unsigned gen = 0;
};
// Waves per SIMD the restriction is compiled for.  One pose per D+1 lanes: 4 (<= 128 VGPRs), because the grid is sized
// for 4 resident workgroups per CU (1 024: the 1 563 tiles of the 100k block in two rounds) and the symmetric-storage
// variants came out at 130 VGPRs = 3 per CU, i.e. a quarter of the grid waited for a slot.
template <int D, int R, int SPLIT>
struct RestrictWaves {
  static constexpr int kMin = (SPLIT == 1 && D == 3 && R <= 5) ? DPGO_RESTRICT_WAVES : 1;  // (121 .. 130 VGPRs without)
};
// PT: storage type of the prolongation blocks (float with the fp32 operator copies of the cycle, like MAT's values);
// XT: storage type of the cycle-internal vectors x1 (read: own tile + gather) and res_out (written) -- float with them
template <int D, int R, int SPLIT, class MAT = BsrDev, class PT = double, class XT = double>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(RestrictWaves<D, R, SPLIT>::kMin, 8))) void k_ml_restrict(MAT A, const XT* __restrict__ x1,
                                                        const double* __restrict__ r, const PT* __restrict__ Pb,
                                                        double shift, int k, double* __restrict__ rc,
                                                        float* __restrict__ rc32,
                                                        const double* __restrict__ dinv_next, double omega,
                                                        double* __restrict__ x1c, const DevState* gate,
                                                        int n, XT* __restrict__ res_out = nullptr,
                                                        double* __restrict__ tbuf = nullptr,
                                                        const int32_t* __restrict__ tpos = nullptr,
                                                        TcgStopCheck stop = TcgStopCheck()) {
  // tbuf (graph aggregates: their members are anywhere): partial sums of P_i^T res_i over runs of same-aggregate nodes
  // are written out instead of the aggregate's sum -- to the slots tpos encodes, ordered by aggregate, so that
  // k_ml_agg_sum reads every aggregate's partial sums as one contiguous run
  using GEO = Geo<D, R, SPLIT>;
  DPGO_TL_TILES_DECL;
  if (gate && (gate->tcg_done || gate->rtr_stop)) return;
  DPGO_STAMP_ENTRY_IN(g_tl_restrict);
  __shared__ double res_s[kWaves][GEO::G][GEO::T];  // residual tiles (per wave)
  __shared__ double t_s[GEO::P][GEO::T];            // P_i^T res_i of every node of the workgroup tile
  const LaneId L = lane_id<D, SPLIT>();
  const int ntiles = (n + GEO::P - 1) / GEO::P;
  const TileIter ti_ = tile_iter(ntiles);
  if (stop.state && blockIdx.x == gridDim.x - 1) {  // workgroup-uniform
    double rr_sum[1];
    load_partials<1>(stop.pin, stop.nb, rr_sum, &t_s[0][0]);
    if (threadIdx.x == 0) {
      const DevState* st = stop.state;
      const double norm_r = sqrt(rr_sum[0]), nr0 = st->norm_r0, theta = st->theta, kappa = st->kappa;
      const double pw = (theta == 1.0) ? nr0 : pow(nr0, theta);
      const int j = st->tcg_j;
      if (j >= st->min_inner && norm_r <= nr0 * (pw < kappa ? pw : kappa)) {
        stop.state->tcg_status = (kappa < pw) ? TCG_LCON : TCG_SCON;
        stop.state->tcg_done = 1;
        if (stop.hflag) {
          const unsigned g_ = stop.gen ? stop.gen : (unsigned)st->pad0;  // (0: a replayed launch, see state_gen)
          const unsigned long long w = ((unsigned long long)g_ << 32) |
                                       ((unsigned long long)((unsigned)j & 0xFFFFFFu) << 8) | 1ull;
          __hip_atomic_store(stop.hflag, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
    __syncthreads();  // (t_s is reused by the tiles below)
  }
  DPGO_STAMP_AT_IN(g_tl_restrict, 1);
  for (int tk = ti_.first; tk < ti_.last; tk += ti_.step) {
    DPGO_STAMP_TILE_IN(g_tl_restrict, 0);
    const int tile = tile_of(A, tk);
    const int lp = L.wave * GEO::G + L.g;  // node slot inside the workgroup tile
    const int i = tile * GEO::P + lp;
    const bool okp = (L.g < GEO::G) && (i < n);
    const bool ok = okp && (L.s == 0);
    const size_t off = (size_t)i * GEO::T + L.c * R;
    // (loading the own rows BEFORE the gather was tried and is slower: 35.3 -> 44.3 us at 100k poses)
    // (the small ones -- the pose's column of P, its slot in the run sums -- in front of it: each its own round trip behind
    // the gather in the in-kernel timeline, 0.3 and 0.5 us shorter per tile when hoisted, and the gather that much longer:
    // launch 17.8 -> 18.5 us.  The cycle's kernels move their time between phases, not off the launch.)
    double h[R];
    q_gather<D, R, SPLIT>(A, x1, i, L.s, L.c, okp, h);
    DPGO_TL_USE(h[0]);
    DPGO_STAMP_TILE_IN(g_tl_restrict, 1);
    if constexpr (SPLIT == 1 && Span<D, R, 1>::kOk && DPGO_CYCLE_SPAN) {
      // (opt-in, measured slower: see DPGO_CYCLE_SPAN in common.h) the own rows of r and x1 and the kept residual move as
      // lane-linear pieces of the wave's span: r is staged in the wave's residual tile, x1 in its rows of t_s (free until
      // the P^T products below); same arithmetic
      const int p0 = tile * GEO::P + L.wave * GEO::G;
      const int npose = (n - p0) < GEO::G ? (n - p0) : GEO::G;
      const int valid = npose > 0 ? npose * GEO::T : 0;
      const size_t base = (size_t)p0 * GEO::T;
      double* rw = &res_s[L.wave][0][0];
      span_to_lds<D, R>(r + base, rw, valid);
      span_to_lds<D, R>(x1 + base, &t_s[L.wave * GEO::G][0], valid);
      wave_sync();
      if (ok) {
        const double* rr = &res_s[L.wave][L.g][L.c * R];
        const double* xr = &t_s[lp][L.c * R];
#pragma unroll
        for (int a = 0; a < R; ++a) h[a] = rr[a] - h[a] - shift * xr[a];
        store_col<R>(&res_s[L.wave][L.g][L.c * R], h);  // (a lane reads and writes its own column only)
      }
      wave_sync();
      if (res_out) span_from_lds<D, R>(res_out + base, rw, valid);  // kept for k_ml_post_ap, in its storage type
    } else {
      if (ok) {
        XT xr[R];  // (kept in its storage type until used: five registers instead of ten for the fp32 cycle vectors)
        double rr[R];
#pragma unroll
        for (int a = 0; a < R; ++a) xr[a] = x1[off + a];
        load_col<R>(r + off, rr);
#pragma unroll
        for (int a = 0; a < R; ++a) h[a] = rr[a] - h[a] - shift * (double)xr[a];
        store_col<R>(&res_s[L.wave][L.g][L.c * R], h);
        if (res_out) store_col_t<R>(res_out + off, h);  // kept for k_ml_post_ap (in its storage type; P^T res uses h itself)
      }
      wave_sync();
    }
    DPGO_STAMP_TILE_IN(g_tl_restrict, 2);
    if (L.s == 0 && L.g < GEO::G) {
      double t[R];
#pragma unroll
      for (int a = 0; a < R; ++a) t[a] = 0.0;
      if (ok) {  // row c of P_i^T res_i = sum_c' P_i[c'][c] res_i[c'][:]
        const PT* __restrict__ pb = Pb + (size_t)i * GEO::BB;
#pragma unroll
        for (int cc = 0; cc < GEO::B; ++cc) {
          const double pv = (double)pb[cc * GEO::B + L.c];
#pragma unroll
          for (int a = 0; a < R; ++a) t[a] = fma(pv, res_s[L.wave][L.g][cc * R + a], t[a]);
        }
      }
      store_col<R>(&t_s[lp][L.c * R], t);  // zeros for nodes beyond n (rows lp of a wave are private to it)
    }
    DPGO_STAMP_TILE_IN(g_tl_restrict, 3);
    if (tbuf) {  // kernel-uniform: graph aggregates
      // inside the wave's G consecutive nodes every RUN of nodes of one aggregate is added up here (fixed order) and
      // leaves ONE partial sum: tpos[i] = slot * 32 + length for the first node of a run, -1 otherwise (host:
      // ml_symbolic_setup).  Consecutive poses mostly share an aggregate, so k_ml_agg_sum reads a fraction of what one
      // value per node was (100k poses: 16 MB written + 16 MB read per cycle before).
      wave_sync();
      if (ok) {
        const int info = tpos[i];
        if (info >= 0) {
          const int len = info & 31;
          double acc[R];
#pragma unroll
          for (int a = 0; a < R; ++a) acc[a] = t_s[lp][L.c * R + a];
          for (int m = 1; m < len; ++m) {
#pragma unroll
            for (int a = 0; a < R; ++a) acc[a] += t_s[lp + m][L.c * R + a];
          }
          store_col<R>(tbuf + (size_t)(info >> 5) * GEO::T + L.c * R, acc);
        }
      }
      wave_sync();  // (the wave's rows of t_s are rewritten by its next tile)
      DPGO_STAMP_TILE_IN(g_tl_restrict, 4);
      DPGO_TILE_NEXT;
      continue;
    }
    __syncthreads();
    // Sum over the aggregate's members (all inside this tile).  Large aggregates: all threads first fold runs of 8
    // members in place (entry e of rows row0 .. row0+7 into row0), the head node then adds k/8 rows instead of k -- a
    // single node's lanes walking 64 rows serially was a visible part of the kernel.  Fixed order either way.
    const int mstep = (k >= 16) ? 8 : 1;  // kernel-uniform
    if (mstep == 8) {
      const int nseg = (k + 7) / 8, nagg = GEO::P / k;
      for (int tsk = threadIdx.x; tsk < nagg * nseg * GEO::T; tsk += kBlock) {
        const int e = tsk % GEO::T, sg = (tsk / GEO::T) % nseg, ag = tsk / (GEO::T * nseg);
        const int row0 = ag * k + sg * 8;
        double sacc = t_s[row0][e];
        for (int m = 1; m < 8 && sg * 8 + m < k; ++m) sacc += t_s[row0 + m][e];
        t_s[row0][e] = sacc;
      }
      __syncthreads();
    }
    const bool head = ok && (i % k) == 0;
    double acc[R];
    if (head) {
#pragma unroll
      for (int a = 0; a < R; ++a) acc[a] = 0.0;
      for (int m = 0; m < k && lp + m < GEO::P; m += mstep) {
#pragma unroll
        for (int a = 0; a < R; ++a) acc[a] += t_s[lp + m][L.c * R + a];
      }
      // (the lane's column offset is made opaque HERE: otherwise the two store addresses are formed once in front of the
      // tile loop and, at 4 waves per SIMD, spilled -- 16 bytes of scratch written per thread and launch also when the
      // graph-aggregate path above never comes here)
      int cr = L.c * R;
      asm volatile("" : "+v"(cr));
      if (rc32) {  // the dense level reads its right-hand side in the precision its inverse is stored in
#pragma unroll
        for (int a = 0; a < R; ++a) rc32[(size_t)(i / k) * GEO::T + cr + a] = (float)acc[a];
      } else {
        store_col<R>(rc + (size_t)(i / k) * GEO::T + cr, acc);
      }
    }
    __syncthreads();
    if (dinv_next) {  // kernel-uniform
      if (head) store_col<R>(&t_s[lp][L.c * R], acc);  // the aggregate's B columns meet in its head's slot
      wave_sync();                                     // the B lanes of one node sit in one wavefront
      if (head) {
        double z[R];
        jacobi_col<D, R>(&t_s[lp][0], dinv_next + (size_t)(i / k) * GEO::BB + L.c * GEO::B, z);
#pragma unroll
        for (int a = 0; a < R; ++a) z[a] *= omega;
        store_col<R>(x1c + (size_t)(i / k) * GEO::T + L.c * R, z);
      }
      __syncthreads();
    }
  }
}

// Graph aggregates: rc[a] = sum of aggregate a's partial sums, in a fixed order.  k_ml_restrict wrote them ordered by
// aggregate, so aggregate a is the contiguous run t[agg_ptr[a] .. agg_ptr[a+1]) (agg_ptr = the host's seg_ptr).  One workgroup
// per aggregate: thread (group g, element e) adds every NG-th member's element e, the partial sums meet in LDS and the
// first T threads add them up.  rc32: the dense level stores its right-hand side in fp32.
template <int D, int R>
__global__ __launch_bounds__(kBlock) void k_ml_agg_sum(const double* __restrict__ t, const int32_t* __restrict__ agg_ptr,
                                                       int na, double* __restrict__ rc, float* __restrict__ rc32,
                                                       const DevState* __restrict__ gate) {
  constexpr int T = (D + 1) * R, NG = kBlock / T;
  if (gate && (gate->tcg_done || gate->rtr_stop)) return;
  __shared__ double part[NG][T];
  const int e = threadIdx.x % T, g = threadIdx.x / T;
  for (int a = blockIdx.x; a < na; a += gridDim.x) {
    const int m0 = agg_ptr[a], m1 = agg_ptr[a + 1];
    if (g < NG) {
      double acc = 0.0;
#pragma unroll 8
      for (int m = m0 + g; m < m1; m += NG) acc += t[(size_t)m * T + e];
      part[g][e] = acc;
    }
    __syncthreads();
    if (threadIdx.x < T) {
      double acc = part[0][threadIdx.x];
#pragma unroll
      for (int q = 1; q < NG; ++q) acc += part[q][threadIdx.x];
      if (rc32)
        rc32[(size_t)a * T + threadIdx.x] = (float)acc;
      else
        rc[(size_t)a * T + threadIdx.x] = acc;
    }
    __syncthreads();
  }
}

// Coarsest level + prolongation to the level above it.  One workgroup per NODES coarsest nodes: their B rows each of
// xc = M rc (M = dense inverse, row-major with leading dimension lda), then x_i = x1_i + P_i xc_a for the aggregates'
// nodes.  Every wave takes a quarter of the columns and ALL rows of the workgroup's nodes: the right-hand side rc
// (R doubles per column: 10x the bytes of a matrix row) is read once per workgroup, not once per row.  A lane owns
// groups of 16 bytes' worth of columns (2 for an fp64 inverse, 4 for an fp32 one), so that the matrix rows AND the
// right-hand-side values move as 16-byte loads (with 8-byte loads the kernel is bound by load issue, not by bytes);
// NODES = 2 halves the right-hand-side loads per matrix byte once more.
// MT = float: the inverse AND the restricted residual it multiplies are STORED in fp32 (the 100k-pose block streams 156 MB
// instead of 313 MB through here every cycle, and every workgroup reads the whole right-hand side: that traffic halves
// too); every product and sum stays fp64.  The coarse-grid correction of a preconditioner does not need more: the
// Hessian-vector products to the tolerance are the same (oracle experiment in DESIGN.md section 5).
#ifndef DPGO_COARSE_WAVES
#define DPGO_COARSE_WAVES 2   // waves per SIMD the kernel is compiled for (<= 256 VGPRs)
#endif
#ifndef DPGO_COARSE_UNROLL
#define DPGO_COARSE_UNROLL 1  // streaming steps whose loads are in flight together (measured: 1 -> 49.4, 2 -> 56.6 us)
#endif
// stream_hint: the matrix is read ONCE per cycle by exactly one workgroup.  When the loop's working set exceeds the 256 MB
// Infinity Cache the inverse is streamed with non-temporal loads so that it does not push Q and the tCG vectors out -- also
// a small one: at 100k poses with merged aggregates (38 MB) plain loads make THIS kernel faster (13.2 -> 10.5 us) and the
// loop slower (bench step 4.39 -> 4.48 ms, same box).  (Requesting several steps' loads together, by hand, made the kernel
// slower at every depth tried -- 2 / 3 / 5 steps: 13.7 / 17.3 / 19.1 us against 13.2 --: it runs at the rate its source
// delivers, not at a latency.)
template <int D, int R, int NODES, class MT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(DPGO_COARSE_WAVES, DPGO_COARSE_WAVES))) void k_ml_coarse_prolong(const MT* __restrict__ M, int lda,
                                                              const MT* __restrict__ rc,
                                                              const double* __restrict__ x1,
                                                              const double* __restrict__ Pb, int k,
                                                              double* __restrict__ x, const DevState* __restrict__ gate,
                                                              int n, int nc, double* __restrict__ xc_out = nullptr,
                                                              int stream_hint = 1) {
  constexpr int B = D + 1, T = B * R, BB = B * B, NR = NODES * B;
  constexpr int CPL = 16 / (int)sizeof(MT);  // columns per lane and step
  struct alignas(16) Pack {
    MT v[CPL];
  };
  if (gate && (gate->tcg_done || gate->rtr_stop)) return;
  // partial sums of all 256 threads, one row per (matrix row, right-hand side): summed by a fixed two-stage tree through
  // LDS (40 wavefront reductions per wave and node group cost more instructions than the streaming loop itself)
  constexpr int NV = B * R, SEG = 8, SEGLEN = kBlock / SEG;  // one node's values at a time: 40 KB of LDS for d = 3, r = 5
  __shared__ double xc_s[NR][R];
  __shared__ double all_s[NV][kBlock];
  __shared__ double seg_s[NV][SEG];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int N = nc * B;
  const int ngroups = (nc + NODES - 1) / NODES;
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int a0 = grp * NODES;
    {
      // rows a0*B .. a0*B + NR - 1; rows past N (ragged last group) are padding rows of the lda x lda array: finite
      const MT* __restrict__ m = M + (size_t)(a0 * B) * lda;
      double acc[NR][R];
#pragma unroll
      for (int c = 0; c < NR; ++c)
#pragma unroll
        for (int q = 0; q < R; ++q) acc[c][q] = 0.0;
      const int npack = (N + CPL - 1) / CPL;  // lda is a multiple of 64 and >= N: columns past N hold zeros
      const dbl2* __restrict__ rc2 = reinterpret_cast<const dbl2*>(rc);  // 16-byte pieces: R of them per column pack
      struct alignas(16) RPack {
        MT v[CPL * R];
      };
      auto stream_rows = [&](auto nt) {
#pragma unroll DPGO_COARSE_UNROLL
        for (int j = wave * 64 + lane; j < npack; j += kBlock) {
          Pack mv[NR];
#pragma unroll
          for (int c = 0; c < NR; ++c) {
            const dbl2* src = reinterpret_cast<const dbl2*>(m + (size_t)c * lda + CPL * j);
            dbl2 raw;
            if constexpr (decltype(nt)::value)
              raw = __builtin_nontemporal_load(src);
            else
              raw = *src;
            __builtin_memcpy(&mv[c], &raw, 16);
          }
          RPack rp;  // rc of column CPL j + cc in [cc R, (cc + 1) R)
          if (CPL * j + CPL <= N) {
            dbl2 raw[R];
#pragma unroll
            for (int q = 0; q < R; ++q) raw[q] = rc2[(size_t)j * R + q];
            __builtin_memcpy(&rp, raw, sizeof(rp));
          } else {
#pragma unroll
            for (int e = 0; e < CPL * R; ++e)
              rp.v[e] = (CPL * j + e / R < N) ? rc[(size_t)(CPL * j) * R + e] : (MT)0;
          }
          double rv[CPL * R];
#pragma unroll
          for (int e = 0; e < CPL * R; ++e) rv[e] = (double)rp.v[e];
#pragma unroll
          for (int c = 0; c < NR; ++c) {
#pragma unroll
            for (int q = 0; q < R; ++q) {
#pragma unroll
              for (int cc = 0; cc < CPL; ++cc) acc[c][q] = fma((double)mv[c].v[cc], rv[cc * R + q], acc[c][q]);
            }
          }
        }
      };
      if (stream_hint)
        stream_rows(std::true_type{});
      else
        stream_rows(std::false_type{});
#pragma unroll
      for (int nd = 0; nd < NODES; ++nd) {
#pragma unroll
        for (int c = 0; c < B; ++c)
#pragma unroll
          for (int q = 0; q < R; ++q) all_s[c * R + q][threadIdx.x] = acc[nd * B + c][q];
        __syncthreads();
        for (int tsk = threadIdx.x; tsk < NV * SEG; tsk += kBlock) {  // stage 1: SEG segments of 32 consecutive threads
          const int v = tsk / SEG, sg = tsk % SEG;
          double sum = 0.0;
#pragma unroll 8
          for (int e = 0; e < SEGLEN; ++e)  // start rotated by the task index: the wave's reads spread over the LDS banks
            sum += all_s[v][sg * SEGLEN + ((e + tsk) & (SEGLEN - 1))];
          seg_s[v][sg] = sum;
        }
        __syncthreads();
        if (threadIdx.x < NV) {  // stage 2
          double sum = seg_s[threadIdx.x][0];
#pragma unroll
          for (int sg = 1; sg < SEG; ++sg) sum += seg_s[threadIdx.x][sg];
          xc_s[nd * B + threadIdx.x / R][threadIdx.x % R] = sum;
        }
      }
    }
    __syncthreads();
    if (xc_out) {  // the level above forms x = x1 + P xc itself (k_ml_post_ap): only the solution leaves
      if (threadIdx.x < NR * R && a0 * B + threadIdx.x / R < N)
        xc_out[(size_t)(a0 * B) * R + threadIdx.x] = xc_s[threadIdx.x / R][threadIdx.x % R];
      __syncthreads();
      continue;
    }
    for (int tsk = threadIdx.x; tsk < NODES * k * B; tsk += kBlock) {  // (node, row c) tasks of the aggregates
      const int i = a0 * k + tsk / B, c = tsk % B;
      if (i < n) {
        const int an = (tsk / B) / k;  // which of the workgroup's coarsest nodes
        const double* __restrict__ pb = Pb + (size_t)i * BB + c * B;
        const size_t off = (size_t)i * T + c * R;
#pragma unroll
        for (int q = 0; q < R; ++q) {
          double v = x1[off + q];
#pragma unroll
          for (int cc = 0; cc < B; ++cc) v = fma(pb[cc], xc_s[an * B + cc][q], v);
          x[off + q] = v;
        }
      }
    }
    __syncthreads();
  }
}

// Post-smoothing of an intermediate level l (0 < l < L-1) + prolongation to level l-1:
//   z_a = x_a + w Dinv_a (r_a - (A x)_a);    xf_i = x1f_i + Pf_i z_a  for the kf level-(l-1) nodes i of aggregate a.
template <int D, int R, int SPLIT>
__global__ __launch_bounds__(kBlock) void k_ml_post_mid(BsrDev A, const double* __restrict__ xv,
                                                        const double* __restrict__ r, const double* __restrict__ dinv,
                                                        double omega, const double* __restrict__ x1f,
                                                        const double* __restrict__ Pbf, int kf, double* __restrict__ xf,
                                                        int nf, const DevState* __restrict__ gate, int n) {
  using GEO = Geo<D, R, SPLIT>;
  if (gate && (gate->tcg_done || gate->rtr_stop)) return;
  __shared__ double sm[kWaves][2][GEO::G][GEO::T];
  const LaneId L = lane_id<D, SPLIT>();
  const int ntiles = (n + GEO::P - 1) / GEO::P;
  const TileIter ti_ = tile_iter(ntiles);
  for (int tile = ti_.first; tile < ti_.last; tile += ti_.step) {
    const int i = tile * GEO::P + L.wave * GEO::G + L.g;
    const bool okp = (L.g < GEO::G) && (i < n);
    const bool ok = okp && (L.s == 0);
    const size_t off = (size_t)i * GEO::T + L.c * R;
    double* vs = ok ? &sm[L.wave][0][L.g][0] : nullptr;
    double* zs = ok ? &sm[L.wave][1][L.g][0] : nullptr;
    double h[R], xr[R], z[R];
    spmm_col<D, R, SPLIT>(A.rowptr, A.colidx, A.vals, xv, i, L.s, L.c, okp, h);
    if (ok) {
      double rr[R];
      load_col<R>(xv + off, xr);
      load_col<R>(r + off, rr);
#pragma unroll
      for (int a = 0; a < R; ++a) h[a] = rr[a] - h[a];  // r - A x (the shift lives in A's values on coarse levels)
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
      for (int m = 0; m < kf; ++m) {
        const int fi = i * kf + m;
        if (fi < nf) {
          const double* __restrict__ pb = Pbf + (size_t)fi * GEO::BB + L.c * GEO::B;
          const size_t foff = (size_t)fi * GEO::T + L.c * R;
          double out[R];
          load_col<R>(x1f + foff, out);
#pragma unroll
          for (int cc = 0; cc < GEO::B; ++cc) {
            const double pv = pb[cc];
#pragma unroll
            for (int a = 0; a < R; ++a) out[a] = fma(pv, zs[cc * R + a], out[a]);
          }
          store_col<R>(xf + foff, out);
        }
      }
    }
    wave_sync();
  }
}

#ifndef DPGO_POST_RR_LDS
#define DPGO_POST_RR_LDS 1  // k_ml_post_ap: r for <z, r> re-read from LDS (0: kept in registers)
#endif
#ifndef DPGO_POST_WAVES
#define DPGO_POST_WAVES 4  // waves per SIMD the level-0 post-smoothing kernels are compiled for (<= 128 VGPRs, see grid_post())
#endif
// Post-smoothing of level 0 in the SpMM's epilogue, tangent projection, and the partial sums <r,r>, <z,r> for the next
// k_tcg_hess (slots 0 and 1 of every entry of ITS grid):  z = proj_X( x + w Dinv (r - (Q + shift I) x) ).
template <int D, int R, int SPLIT, class MAT = BsrDev>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(SPLIT == 1 ? DPGO_POST_WAVES : 1))) void k_ml_post(MAT Q, const double* __restrict__ X,
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
  for (int tk = ti_.first; tk < ti_.last; tk += ti_.step) {
    const int tile = tile_of(Q, tk);
    const int i = tile * GEO::P + L.wave * GEO::G + L.g;
    const bool okp = (L.g < GEO::G) && (i < n);
    const bool ok = okp && (L.s == 0);
    const size_t off = (size_t)i * GEO::T + L.c * R;
    double* ys = ok ? &sm[L.wave][0][L.g][0] : nullptr;
    double* vs = ok ? &sm[L.wave][1][L.g][0] : nullptr;
    double* zs = ok ? &sm[L.wave][2][L.g][0] : nullptr;
    double h[R], xr[R], rr[R], z[R];
    q_gather<D, R, SPLIT>(Q, xv, i, L.s, L.c, okp, h);
    if (ok) {  // (loading the own rows BEFORE the gather was tried and is slower: 30.4 -> 37.3 us at 100k poses)
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
  if (pout) store_partials<2>(part, pout, red);
}

// Level-0 post-smoothing of a TWO-level hierarchy without a gather from a pose vector:
//   x = x1 + P xc,   r - A x = (r - A x1) - (A P) xc = res1 - AP xc,   z = proj_X( x + w Dinv (res1 - AP xc) ).
// AP has about half of Q's blocks and its gather reads the coarse solution (a few hundred KB: L2-resident) instead of a 16 MB
// pose vector; x1 = w Dinv r is recomputed from r (read anyway for <r,r>, <z,r>), the prolongation x1 + P xc happens here and
// not in the dense kernel.  Same operator as k_ml_post up to summation order.
// VT: storage type of A P's values and of the prolongation blocks, RT: of the kept residual res1 (float: the fp32 copies of the cycle)
template <int D, int R, int SPLIT, class VT = double, class RT = VT>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(SPLIT == 1 ? DPGO_POST_WAVES : 1))) void k_ml_post_ap(BsrDevT<VT> AP, const double* __restrict__ X,
                                                       const double* __restrict__ r, const RT* __restrict__ res1,
                                                       const double* __restrict__ xc, const VT* __restrict__ Pb, AggMap am,
                                                       const double* __restrict__ dinv, double omega,
                                                       double* __restrict__ Z, double* __restrict__ pout,
                                                       const DevState* __restrict__ gate, int n) {
  using GEO = Geo<D, R, SPLIT>;
  DPGO_TL_TILES_DECL;
  if (gate && (gate->tcg_done || gate->rtr_stop)) return;
  DPGO_STAMP_ENTRY_IN(g_tl_post);
  __shared__ double sm[kWaves][3][GEO::G][GEO::T];
  __shared__ double red[kWaves * kNP];
  const LaneId L = lane_id<D, SPLIT>();
  const int ntiles = (n + GEO::P - 1) / GEO::P;
  const TileIter ti_ = tile_iter(ntiles);
  double part[2] = {0.0, 0.0};
  for (int tile = ti_.first; tile < ti_.last; tile += ti_.step) {
    DPGO_STAMP_TILE_IN(g_tl_post, 0);
    const int i = tile * GEO::P + L.wave * GEO::G + L.g;
    const bool okp = (L.g < GEO::G) && (i < n);
    const bool ok = okp && (L.s == 0);
    const size_t off = (size_t)i * GEO::T + L.c * R;
    double* ys = ok ? &sm[L.wave][0][L.g][0] : nullptr;
    double* vs = ok ? &sm[L.wave][1][L.g][0] : nullptr;
    double* zs = ok ? &sm[L.wave][2][L.g][0] : nullptr;
    double h[R], rr[R], xcol[R], z[R], dr[GEO::B];
    // (the aggregate's label and the pose's row of P requested HERE, in front of the gather, take 2.5 us out of the
    // smoothing + prolongation phase of a tile -- the label is a round trip of its own in front of the coarse tile's -- and
    // the gather grows by as much: launch 21.2 .. 22.7 -> 20.6 .. 22.4 us, cycle tail 59.1 -> 60.0 us; not kept)
    spmm_col<D, R, SPLIT>(AP.rowptr, AP.colidx, AP.vals, xc, i, L.s, L.c, okp, h);
    DPGO_TL_USE(h[0]);
    DPGO_STAMP_TILE_IN(g_tl_post, 1);
    constexpr bool kSpan = (SPLIT == 1) && Span<D, R, 1>::kOk && DPGO_CYCLE_SPAN;
    [[maybe_unused]] int valid = 0;
    [[maybe_unused]] size_t base = 0;
    if constexpr (kSpan) {
      // (opt-in, measured slower: see DPGO_CYCLE_SPAN in common.h) own rows of X, r and the kept residual as lane-linear
      // pieces of the wave's span, staged in the three tiles; the output leaves the same way (below); same arithmetic
      const int p0 = tile * GEO::P + L.wave * GEO::G;
      const int npose = (n - p0) < GEO::G ? (n - p0) : GEO::G;
      valid = npose > 0 ? npose * GEO::T : 0;
      base = (size_t)p0 * GEO::T;
      span_to_lds<D, R>(X + base, &sm[L.wave][0][0][0], valid);
      span_to_lds<D, R>(r + base, &sm[L.wave][1][0][0], valid);
      span_to_lds<D, R>(res1 + base, &sm[L.wave][2][0][0], valid);
      if (ok) {
#pragma unroll
        for (int q = 0; q < GEO::B; ++q) dr[q] = dinv[(size_t)i * GEO::BB + L.c * GEO::B + q];
      }
      wave_sync();
      if (ok) {
#pragma unroll
        for (int a = 0; a < R; ++a) {
          rr[a] = vs[L.c * R + a];
          h[a] = zs[L.c * R + a] - h[a];  // r - A x
          part[0] = fma(rr[a], rr[a], part[0]);
        }
        store_col<R>(zs + L.c * R, h);  // (a lane reads and writes its own column only)
      }
    } else if (ok) {
      double x[R], rs[R];
      load_col<R>(X + off, x);
      load_col<R>(r + off, rr);
      load_col_t<R>(res1 + off, rs);
#pragma unroll
      for (int q = 0; q < GEO::B; ++q) dr[q] = dinv[(size_t)i * GEO::BB + L.c * GEO::B + q];
#pragma unroll
      for (int a = 0; a < R; ++a) {
        h[a] = rs[a] - h[a];  // r - A x
        part[0] = fma(rr[a], rr[a], part[0]);
      }
      store_col<R>(ys + L.c * R, x);
      store_col<R>(vs + L.c * R, rr);
      store_col<R>(zs + L.c * R, h);
    }
    wave_sync();
    DPGO_STAMP_TILE_IN(g_tl_post, 2);
    if (ok) {
      double x1c[R], zc[R];
      jacobi_col<D, R>(vs, dr, x1c);  // x1 = w Dinv r
      jacobi_col<D, R>(zs, dr, zc);   // Dinv (r - A x)
      const VT* __restrict__ pb = Pb + (size_t)i * GEO::BB + L.c * GEO::B;
      const double* __restrict__ xa = xc + (size_t)am.of(i) * GEO::T;
#pragma unroll
      for (int a = 0; a < R; ++a) xcol[a] = omega * x1c[a];
#pragma unroll
      for (int cc = 0; cc < GEO::B; ++cc) {
        const double pv = (double)pb[cc];
#pragma unroll
        for (int a = 0; a < R; ++a) xcol[a] = fma(pv, xa[cc * R + a], xcol[a]);
      }
#pragma unroll
      for (int a = 0; a < R; ++a) z[a] = fma(omega, zc[a], xcol[a]);
      DPGO_TL_USE(z[0]);
    }
    DPGO_STAMP_TILE_IN(g_tl_post, 3);
    wave_sync();  // every lane of the pose has read vs / zs before zs is overwritten
    if (ok) store_col<R>(zs + L.c * R, z);
    wave_sync();
    DPGO_STAMP_TILE_IN(g_tl_post, 4);
    if (ok) {
      double out[R], sdummy[D];
      proj_col<D, R>(ys, zs, L.c, z, out, sdummy);
      // <z, r>: r re-read from the wave's tile instead of kept in registers across the smoothing steps (the kernel is
      // compiled for 4 waves per SIMD = 128 VGPRs and spilled 10 of them: ~15 MB of scratch traffic per launch)
      if constexpr (DPGO_POST_RR_LDS) {
#pragma unroll
        for (int a = 0; a < R; ++a) part[1] = fma(out[a], vs[L.c * R + a], part[1]);
      } else {
#pragma unroll
        for (int a = 0; a < R; ++a) part[1] = fma(out[a], rr[a], part[1]);
      }
      if constexpr (kSpan)
        store_col<R>(vs + L.c * R, out);  // (the r tile is free: its last readers were the smoothing steps above)
      else
        store_col<R>(Z + off, out);
    }
    wave_sync();
    if constexpr (kSpan) {
      span_from_lds<D, R>(Z + base, &sm[L.wave][1][0][0], valid);
      wave_sync();
    }
    DPGO_STAMP_TILE_IN(g_tl_post, 5);
    DPGO_TILE_NEXT;
  }
  if (pout) store_partials<2>(part, pout, red);
  DPGO_STAMP_AT_IN(g_tl_post, 2);
}

// ================================================================ on-device setup of the hierarchy
// (the analogue of PoseGraph::constructPreconditioner, src/PoseGraph.cpp:598-613; values-only: the coarse block
// patterns are symbolic and built once per pattern of Q on the host)

// Prolongation blocks of one coarsening.  One thread per PARENT node walks its `span` fine poses along the odometry
// chain, composing G(parent root -> pose): the relative pose T = [R t; 0 1] of the edge i-1 -> i is read off
// Q_{i-1,i} = -T Om = -[w kappa R, w tau t; 0, w tau] (src/DPGO_utils.cpp:307-329); a missing / zero-weight link restarts
// the chain at the identity.  Every `stride`-th pose is the root of a child node c = i / stride:  Pb[c] = G^T.
template <int D>
__global__ __launch_bounds__(kBlock) void k_ml_build_P(BsrDev Q, int n_fine, int stride, int span,
                                                       double* __restrict__ Pb, int n_parent) {
  constexpr int B = D + 1, BB = B * B;
  for (int a = blockIdx.x * kBlock + threadIdx.x; a < n_parent; a += gridDim.x * kBlock) {
    double G[B][B];
#pragma unroll
    for (int p = 0; p < B; ++p)
#pragma unroll
      for (int q = 0; q < B; ++q) G[p][q] = (p == q) ? 1.0 : 0.0;
    const long long first = (long long)a * span;
    const long long last = (first + span < (long long)n_fine) ? first + span : (long long)n_fine;
    for (long long il = first; il < last; ++il) {
      const int i = (int)il;
      if (il != first) {
        int tb = -1;
        for (int t = Q.rowptr[i - 1]; t < Q.rowptr[i]; ++t)
          if (Q.colidx[t] == i) tb = t;
        bool ok = tb >= 0;
        double wt = 0.0, wk = 0.0;
        const double* __restrict__ blk = Q.vals + (size_t)(tb >= 0 ? tb : 0) * BB;
        if (ok) {
          wt = -blk[D * B + D];
#pragma unroll
          for (int p = 0; p < D; ++p) wk = fma(blk[p * B], blk[p * B], wk);
          wk = sqrt(wk);
          ok = (wt > 0.0) && (wk > 0.0);
        }
        if (ok) {
          double Tm[B][B], Gn[B][B];
#pragma unroll
          for (int p = 0; p < B; ++p)
#pragma unroll
            for (int q = 0; q < B; ++q) Tm[p][q] = (p == q) ? 1.0 : 0.0;
#pragma unroll
          for (int p = 0; p < D; ++p) {
#pragma unroll
            for (int q = 0; q < D; ++q) Tm[p][q] = -blk[p * B + q] / wk;
            Tm[p][D] = -blk[p * B + D] / wt;
          }
#pragma unroll
          for (int p = 0; p < B; ++p)
#pragma unroll
            for (int q = 0; q < B; ++q) {
              double s = 0.0;
#pragma unroll
              for (int m = 0; m < B; ++m) s = fma(G[p][m], Tm[m][q], s);
              Gn[p][q] = s;
            }
#pragma unroll
          for (int p = 0; p < B; ++p)
#pragma unroll
            for (int q = 0; q < B; ++q) G[p][q] = Gn[p][q];
        } else {
#pragma unroll
          for (int p = 0; p < B; ++p)
#pragma unroll
            for (int q = 0; q < B; ++q) G[p][q] = (p == q) ? 1.0 : 0.0;
        }
      }
      if (i % stride == 0) {
        double* __restrict__ out = Pb + (size_t)(i / stride) * BB;
#pragma unroll
        for (int p = 0; p < B; ++p)
#pragma unroll
          for (int q = 0; q < B; ++q) out[p * B + q] = G[q][p];
      }
    }
  }
}

// Prolongation blocks for GRAPH aggregates.  Every aggregate carries a spanning tree (the breadth-first tree the host's
// aggregation grew it along): parent[i] = the node that discovered i (-1: the aggregate's root), pslot[i] = the slot of
// block (parent, i) in Q.  One thread per aggregate walks its members in discovery order (a parent precedes its children)
// and composes G(root -> i) = G(root -> parent) T(parent -> i);  Pb[i] = G^T.  T is read off the block as in k_ml_build_P;
// the block of an edge measured the other way round (i -> parent) is the transpose -(T' Om)^T: T = T'^-1.  A block that
// is neither (zero weight, several measurements summed) restarts the chain at the identity.
template <int D>
__global__ __launch_bounds__(kBlock) void k_ml_build_P_tree(BsrDev Q, const int32_t* __restrict__ agg_ptr,
                                                            const int32_t* __restrict__ agg_mem,
                                                            const int32_t* __restrict__ parent,
                                                            const int32_t* __restrict__ pslot, double* Pb, int na) {
  constexpr int B = D + 1, BB = B * B;
  for (int a = blockIdx.x * kBlock + threadIdx.x; a < na; a += gridDim.x * kBlock) {
    for (int m = agg_ptr[a]; m < agg_ptr[a + 1]; ++m) {
      const int i = agg_mem[m];
      const int par = parent[i];
      double G[B][B];
#pragma unroll
      for (int p = 0; p < B; ++p)
#pragma unroll
        for (int q = 0; q < B; ++q) G[p][q] = (p == q) ? 1.0 : 0.0;
      if (par >= 0) {
        const double* blk = Q.vals + (size_t)pslot[i] * BB;
        const double wt = -blk[D * B + D];
        double wk = 0.0;
#pragma unroll
        for (int p = 0; p < D; ++p) wk = fma(blk[p * B], blk[p * B], wk);
        wk = sqrt(wk);
        bool fwd = true, bwd = true;
#pragma unroll
        for (int q = 0; q < D; ++q) {
          fwd = fwd && (blk[D * B + q] == 0.0);
          bwd = bwd && (blk[q * B + D] == 0.0);
        }
        if ((wt > 0.0) && (wk > 0.0) && (fwd || bwd)) {
          double Tm[B][B];
#pragma unroll
          for (int p = 0; p < B; ++p)
#pragma unroll
            for (int q = 0; q < B; ++q) Tm[p][q] = (p == q) ? 1.0 : 0.0;
#pragma unroll
          for (int p = 0; p < D; ++p)
#pragma unroll
            for (int q = 0; q < D; ++q) Tm[p][q] = -blk[p * B + q] / wk;
          if (fwd) {
#pragma unroll
            for (int p = 0; p < D; ++p) Tm[p][D] = -blk[p * B + D] / wt;
          } else {  // T'^-1 = [R'^T, -R'^T t'; 0 1],  R'^T = Tm's rotation part,  t' = -blk[D][:] / wt
#pragma unroll
            for (int p = 0; p < D; ++p) {
              double sv = 0.0;
#pragma unroll
              for (int q = 0; q < D; ++q) sv = fma(Tm[p][q], blk[D * B + q] / wt, sv);
              Tm[p][D] = sv;
            }
          }
          const double* __restrict__ Pp = Pb + (size_t)par * BB;  // G(root -> parent)^T, written earlier by this thread
#pragma unroll
          for (int p = 0; p < B; ++p)
#pragma unroll
            for (int q = 0; q < B; ++q) {
              double sv = 0.0;
#pragma unroll
              for (int mm = 0; mm < B; ++mm) sv = fma(Pp[mm * B + p], Tm[mm][q], sv);
              G[p][q] = sv;
            }
        }
      }
      double* out = Pb + (size_t)i * BB;
#pragma unroll
      for (int p = 0; p < B; ++p)
#pragma unroll
        for (int q = 0; q < B; ++q) out[p * B + q] = G[q][p];
    }
  }
}

// The same, one WAVE per aggregate (graph aggregates of dozens to hundreds of poses: one thread walking 250 members one
// dependent load after the other took 1.24 ms at 100k poses, 0.37 ms on a 12 500-pose block of 230 aggregates).  The members
// are in breadth-first discovery order, so a parent precedes its children; the wave takes them in chunks of 64 and, inside a
// chunk, in passes: a lane computes once its parent's block is there (parent in an earlier chunk, or done in an earlier
// pass -- ballot mask).  Every block is computed by the same arithmetic as in the one-thread kernel: identical bits.
// mem_pos[i] = position of pose i in agg_mem.
template <int D>
__device__ __forceinline__ void ml_tree_block(const BsrDev& Q, const double* Pb, int i, int par, int slot, double* out) {
  constexpr int B = D + 1, BB = B * B;
  double G[B][B];
#pragma unroll
  for (int p = 0; p < B; ++p)
#pragma unroll
    for (int q = 0; q < B; ++q) G[p][q] = (p == q) ? 1.0 : 0.0;
  if (par >= 0) {
    const double* blk = Q.vals + (size_t)slot * BB;
    const double wt = -blk[D * B + D];
    double wk = 0.0;
#pragma unroll
    for (int p = 0; p < D; ++p) wk = fma(blk[p * B], blk[p * B], wk);
    wk = sqrt(wk);
    bool fwd = true, bwd = true;
#pragma unroll
    for (int q = 0; q < D; ++q) {
      fwd = fwd && (blk[D * B + q] == 0.0);
      bwd = bwd && (blk[q * B + D] == 0.0);
    }
    if ((wt > 0.0) && (wk > 0.0) && (fwd || bwd)) {
      double Tm[B][B];
#pragma unroll
      for (int p = 0; p < B; ++p)
#pragma unroll
        for (int q = 0; q < B; ++q) Tm[p][q] = (p == q) ? 1.0 : 0.0;
#pragma unroll
      for (int p = 0; p < D; ++p)
#pragma unroll
        for (int q = 0; q < D; ++q) Tm[p][q] = -blk[p * B + q] / wk;
      if (fwd) {
#pragma unroll
        for (int p = 0; p < D; ++p) Tm[p][D] = -blk[p * B + D] / wt;
      } else {
#pragma unroll
        for (int p = 0; p < D; ++p) {
          double sv = 0.0;
#pragma unroll
          for (int q = 0; q < D; ++q) sv = fma(Tm[p][q], blk[D * B + q] / wt, sv);
          Tm[p][D] = sv;
        }
      }
      const double* Pp = Pb + (size_t)par * BB;  // G(root -> parent)^T
#pragma unroll
      for (int p = 0; p < B; ++p)
#pragma unroll
        for (int q = 0; q < B; ++q) {
          double sv = 0.0;
#pragma unroll
          for (int mm = 0; mm < B; ++mm) sv = fma(Pp[mm * B + p], Tm[mm][q], sv);
          G[p][q] = sv;
        }
    }
  }
#pragma unroll
  for (int p = 0; p < B; ++p)
#pragma unroll
    for (int q = 0; q < B; ++q) out[p * B + q] = G[q][p];
}
template <int D>
__global__ __launch_bounds__(kBlock) void k_ml_build_P_tree_wave(BsrDev Q, const int32_t* __restrict__ agg_ptr,
                                                                 const int32_t* __restrict__ agg_mem,
                                                                 const int32_t* __restrict__ parent,
                                                                 const int32_t* __restrict__ pslot,
                                                                 const int32_t* __restrict__ mem_pos, double* Pb, int na) {
  constexpr int B = D + 1, BB = B * B;
  const int lane = threadIdx.x & 63;
  const int wave0 = blockIdx.x * kWaves + (threadIdx.x >> 6), nwaves = gridDim.x * kWaves;
  for (int a = wave0; a < na; a += nwaves) {
    const int m0 = agg_ptr[a], m1 = agg_ptr[a + 1];
    for (int c0 = m0; c0 < m1; c0 += 64) {
      const int m = c0 + lane;
      const bool have = m < m1;
      const int i = have ? agg_mem[m] : 0;
      const int par = have ? parent[i] : -1;
      const int slot = have ? pslot[i] : 0;
      const int pp = par >= 0 ? mem_pos[par] : -1;  // the parent's position in the member list
      bool done = !have;
      unsigned long long done_mask = __ballot(done);
      for (int pass = 0; pass < 65 && done_mask != ~0ull; ++pass) {
        const bool ready = !done && (pp < c0 || ((done_mask >> (pp - c0)) & 1ull));
        if (ready) {
          double out[BB];
          ml_tree_block<D>(Q, Pb, i, par, slot, out);
#pragma unroll
          for (int e = 0; e < BB; ++e) Pb[(size_t)i * BB + e] = out[e];
          done = true;
        }
        // the blocks just written are read by other lanes of THIS wave in the next pass / chunk
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        done_mask = __ballot(done);
      }
    }
  }
}

// Galerkin operator of a TWO-level hierarchy from A P (k_ml_build_AP, which runs first):
//   Ac[a][bc] = sum_{i in a} P_i^T (A P)[i][bc].
// One WAVE per coarse slot: the lanes take the aggregate's members 64 apart, each looks the block column bc up in its row of
// A P (a handful of entries) and accumulates its 4 x 4 product; the lanes' partial sums are added by the fixed DPP tree
// (deterministic).  The one-thread-per-slot kernel below rescans every member's block row once per coarse slot of the
// row and took 3.6 ms at 100k poses / 0.84 ms on a 12 500-pose block.
template <int D>
__global__ __launch_bounds__(kBlock) void k_ml_galerkin_ap(BsrDev AP, const double* __restrict__ Pb, AggMap am,
                                                           const int32_t* __restrict__ agg_ptr,
                                                           const int32_t* __restrict__ agg_mem, int n_fine,
                                                           const int32_t* __restrict__ slot_row,
                                                           const int32_t* __restrict__ ccol, double* __restrict__ cvals,
                                                           int cnnzb) {
  const int k = am.k;
  constexpr int B = D + 1, BB = B * B;
  const int lane = threadIdx.x & 63;
  const int wave0 = blockIdx.x * kWaves + (threadIdx.x >> 6), nwaves = gridDim.x * kWaves;
  for (int s = wave0; s < cnnzb; s += nwaves) {
    const int a = slot_row[s], bc = ccol[s];
    double acc[B][B];
#pragma unroll
    for (int p = 0; p < B; ++p)
#pragma unroll
      for (int q = 0; q < B; ++q) acc[p][q] = 0.0;
    const int m0 = am.lab ? agg_ptr[a] : a * k;
    const int m1 = am.lab ? agg_ptr[a + 1] : ((a * k + k < n_fine) ? a * k + k : n_fine);
    for (int m = m0 + lane; m < m1; m += 64) {
      const int i = am.lab ? agg_mem[m] : m;
      int t = -1;
      for (int u = AP.rowptr[i]; u < AP.rowptr[i + 1]; ++u)
        if (AP.colidx[u] == bc) t = u;
      if (t < 0) continue;
      const double* __restrict__ Pi = Pb + (size_t)i * BB;
      const double* __restrict__ av = AP.vals + (size_t)t * BB;
#pragma unroll
      for (int p = 0; p < B; ++p)
#pragma unroll
        for (int q = 0; q < B; ++q) {
          double sv = acc[p][q];
#pragma unroll
          for (int mm = 0; mm < B; ++mm) sv = fma(Pi[mm * B + p], av[mm * B + q], sv);
          acc[p][q] = sv;
        }
    }
#pragma unroll
    for (int p = 0; p < B; ++p)
#pragma unroll
      for (int q = 0; q < B; ++q) acc[p][q] = wave_reduce_lane63(acc[p][q]);
    if (lane == 63) {
      double* __restrict__ out = cvals + (size_t)s * BB;
#pragma unroll
      for (int p = 0; p < B; ++p)
#pragma unroll
        for (int q = 0; q < B; ++q) out[p * B + q] = acc[p][q];
    }
  }
}

// Galerkin operator, values only:  Ac[a][bc] = sum_{i in a} sum_{j in bc} P_i^T (A_ij + [i == j] shift I) P_j.
// One thread per coarse slot (its block row in slot_row) scans the k fine rows of aggregate a: fixed summation order.
template <int D>
__global__ __launch_bounds__(kBlock) void k_ml_galerkin(BsrDev A, double shift, const double* __restrict__ Pb, AggMap am,
                                                        const int32_t* __restrict__ agg_ptr,
                                                        const int32_t* __restrict__ agg_mem, int n_fine,
                                                        const int32_t* __restrict__ slot_row,
                                                        const int32_t* __restrict__ ccol, double* __restrict__ cvals,
                                                        int cnnzb) {
  const int k = am.k;
  constexpr int B = D + 1, BB = B * B;
  for (int s = blockIdx.x * kBlock + threadIdx.x; s < cnnzb; s += gridDim.x * kBlock) {
    const int a = slot_row[s], bc = ccol[s];
    double acc[B][B];
#pragma unroll
    for (int p = 0; p < B; ++p)
#pragma unroll
      for (int q = 0; q < B; ++q) acc[p][q] = 0.0;
    // members of aggregate a: a run of k nodes, or (graph aggregates) the list agg_mem[agg_ptr[a] ..)
    const int m0 = am.lab ? agg_ptr[a] : a * k;
    const int m1 = am.lab ? agg_ptr[a + 1] : ((a * k + k < n_fine) ? a * k + k : n_fine);
    for (int m = m0; m < m1; ++m) {
      const int i = am.lab ? agg_mem[m] : m;
      const double* __restrict__ Pi = Pb + (size_t)i * BB;
      for (int t = A.rowptr[i]; t < A.rowptr[i + 1]; ++t) {
        const int j = A.colidx[t];
        if (am.of(j) != bc) continue;
        const double* __restrict__ av = A.vals + (size_t)t * BB;
        const double* __restrict__ Pj = Pb + (size_t)j * BB;
        double AP[B][B];
#pragma unroll
        for (int p = 0; p < B; ++p)
#pragma unroll
          for (int q = 0; q < B; ++q) {
            double sv = 0.0;
#pragma unroll
            for (int m = 0; m < B; ++m) {
              const double am = av[p * B + m] + ((j == i && p == m) ? shift : 0.0);
              sv = fma(am, Pj[m * B + q], sv);
            }
            AP[p][q] = sv;
          }
#pragma unroll
        for (int p = 0; p < B; ++p)
#pragma unroll
          for (int q = 0; q < B; ++q) {
            double sv = acc[p][q];
#pragma unroll
            for (int m = 0; m < B; ++m) sv = fma(Pi[m * B + p], AP[m][q], sv);
            acc[p][q] = sv;
          }
      }
    }
    double* __restrict__ out = cvals + (size_t)s * BB;
#pragma unroll
    for (int p = 0; p < B; ++p)
#pragma unroll
      for (int q = 0; q < B; ++q) out[p * B + q] = acc[p][q];
  }
}

// A P of level 0, values only:  AP[i][a] = sum_{j in a} (Q_ij + [i == j] shift I) P_j.  One thread per block row.
template <int D>
__global__ __launch_bounds__(kBlock) void k_ml_build_AP(BsrDev Q, double shift, const double* __restrict__ Pb, AggMap am,
                                                        int n, BsrDev AP, double* __restrict__ apvals) {
  constexpr int B = D + 1, BB = B * B;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    for (int s = AP.rowptr[i]; s < AP.rowptr[i + 1]; ++s) {
      const int a = AP.colidx[s];
      double acc[B][B];
#pragma unroll
      for (int p = 0; p < B; ++p)
#pragma unroll
        for (int q = 0; q < B; ++q) acc[p][q] = 0.0;
      for (int t = Q.rowptr[i]; t < Q.rowptr[i + 1]; ++t) {
        const int j = Q.colidx[t];
        if (am.of(j) != a) continue;
        const double* __restrict__ Aij = Q.vals + (size_t)t * BB;
        const double* __restrict__ Pj = Pb + (size_t)j * BB;
#pragma unroll
        for (int p = 0; p < B; ++p)
#pragma unroll
          for (int m = 0; m < B; ++m) {
            const double av = Aij[p * B + m] + ((i == j && p == m) ? shift : 0.0);
#pragma unroll
            for (int q = 0; q < B; ++q) acc[p][q] = fma(av, Pj[m * B + q], acc[p][q]);
          }
      }
#pragma unroll
      for (int p = 0; p < B; ++p)
#pragma unroll
        for (int q = 0; q < B; ++q) apvals[(size_t)s * BB + p * B + q] = acc[p][q];
    }
  }
}

// Dense copy of the coarsest operator, symmetrised (0.5 (Ac + Ac^T), as the oracle), into a zero-filled lda x lda array
// whose padding rows carry a unit diagonal (written by the host wrapper with a separate launch of k_dense_pad_identity).
template <int D>
__global__ __launch_bounds__(kBlock) void k_ml_dense_assemble(BsrDev A, const int32_t* __restrict__ slot_row,
                                                              double* __restrict__ M, int lda, int nnzb) {
  constexpr int B = D + 1, BB = B * B;
  for (int s = blockIdx.x * kBlock + threadIdx.x; s < nnzb; s += gridDim.x * kBlock) {
    const int a = slot_row[s], bc = A.colidx[s];
    int st = -1;  // the transposed slot (bc, a); the pattern is symmetric
    for (int t = A.rowptr[bc]; t < A.rowptr[bc + 1]; ++t)
      if (A.colidx[t] == a) st = t;
    const double* __restrict__ v = A.vals + (size_t)s * BB;
    const double* __restrict__ vt = A.vals + (size_t)(st >= 0 ? st : s) * BB;
#pragma unroll
    for (int p = 0; p < B; ++p)
#pragma unroll
      for (int q = 0; q < B; ++q) {
        const double w = (st >= 0) ? 0.5 * (v[p * B + q] + vt[q * B + p]) : v[p * B + q];
        M[(size_t)(a * B + p) * lda + bc * B + q] = w;
      }
  }
}
