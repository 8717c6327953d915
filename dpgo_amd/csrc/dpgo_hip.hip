// dpgo_hip.hip -- host driver + C ABI (include/dpgo_hip.h) of the MI355X-native RBCD local solver.
//
// Replaces, behind the reference's own interface, QuadraticProblem (src/QuadraticProblem.cpp),
// QuadraticOptimizer (src/QuadraticOptimizer.cpp), the ROPTLIB RTRNewton / tCG_TR loop it drives and
// the LiftedSEManifold operations (src/manifold/*.cpp).  All vectors stay in HBM for the whole solve;
// the host only enqueues kernels and polls a 200-byte state record every few tCG iterations.
#include "kernels.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/dpgo_hip.h"

using namespace dpgo;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIPC(expr)                                                                              \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return fail(DPGO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" + __FILE__ + \
                                    ":" + std::to_string(__LINE__) + ")");                      \
  } while (0)

#define CHK(expr)                \
  do {                           \
    int rc_ = (expr);            \
    if (rc_ != DPGO_OK) return rc_; \
  } while (0)

// (d, r) pairs with compiled kernels
#define DPGO_FOR_DR(M) M(2, 2) M(2, 3) M(2, 4) M(2, 5) M(3, 3) M(3, 4) M(3, 5) M(3, 6)

bool supported(int d, int r) {
#define M(dd, rr) \
  if (d == dd && r == rr) return true;
  DPGO_FOR_DR(M)
#undef M
  return false;
}

// DISPATCH(d, r, body): body sees constexpr int D, R
#define DPGO_CASE(dd, rr, ...)   \
  case (dd) * 16 + (rr): {       \
    constexpr int D = dd, R = rr; \
    __VA_ARGS__;                 \
  } break;
#define DISPATCH(d, r, ...)                                                                         \
  switch ((d) * 16 + (r)) {                                                                         \
    DPGO_CASE(2, 2, __VA_ARGS__) DPGO_CASE(2, 3, __VA_ARGS__) DPGO_CASE(2, 4, __VA_ARGS__)          \
    DPGO_CASE(2, 5, __VA_ARGS__) DPGO_CASE(3, 3, __VA_ARGS__) DPGO_CASE(3, 4, __VA_ARGS__)          \
    DPGO_CASE(3, 5, __VA_ARGS__) DPGO_CASE(3, 6, __VA_ARGS__)                                       \
    default:                                                                                        \
      return fail(DPGO_ERR_UNSUPPORTED, "unsupported (d, r)");                                      \
  }

// launch a <D, R, SPLIT> kernel with the handle's split factor
#define LAUNCH_SPLIT(p, KERNEL, GRID, ...)                                                            \
  do {                                                                                                \
    if ((p)->split == 4)                                                                              \
      hipLaunchKernelGGL((KERNEL<D, R, 4>), dim3(GRID), dim3(kBlock), 0, (p)->stream, __VA_ARGS__);   \
    else if ((p)->split == 2)                                                                         \
      hipLaunchKernelGGL((KERNEL<D, R, 2>), dim3(GRID), dim3(kBlock), 0, (p)->stream, __VA_ARGS__);   \
    else                                                                                              \
      hipLaunchKernelGGL((KERNEL<D, R, 1>), dim3(GRID), dim3(kBlock), 0, (p)->stream, __VA_ARGS__);   \
  } while (0)

struct Bsr {
  int nrows = 0, ncols = 0, nnzb = 0;
  int32_t* rowptr = nullptr;
  int32_t* colidx = nullptr;
  double* vals = nullptr;
  BsrDev dev() const { return BsrDev{rowptr, colidx, vals}; }
};

int free_bsr(Bsr& m) {
  if (m.rowptr) HIPC(hipFree(m.rowptr));
  if (m.colidx) HIPC(hipFree(m.colidx));
  if (m.vals) HIPC(hipFree(m.vals));
  m = Bsr();
  return DPGO_OK;
}

}  // namespace

struct dpgo_problem_s {
  int r = 0, d = 0, n = 0, b = 0, T = 0;
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  Bsr Q;
  Bsr C;  // inter-agent coupling (rectangular), for G
  double* G0 = nullptr;
  double* G = nullptr;
  bool has_G = false;
  double* dinv = nullptr;
  double dinv_shift = -1.0;
  // work vectors
  double *x1 = nullptr, *x2 = nullptr, *g1 = nullptr, *g2 = nullptr, *eta = nullptr, *delta = nullptr,
         *Hd = nullptr, *rr = nullptr, *z = nullptr, *S1 = nullptr, *S2 = nullptr;
  // multilevel (aggregation multigrid) preconditioner: levels[0] = the pose level ... levels.back() = the dense level
  struct MlLevel {
    int n = 0;      // nodes
    int k = 0;      // aggregate size towards the next level (0 on the dense level)
    int split = 1;  // lane groups per node of this level's SpMM-family kernels
    Bsr A;          // level >= 1: Galerkin operator (level 0: Q + shift I, never formed)
    int32_t* slot_row = nullptr;  // level >= 1: block row of every slot of A
    double *dinv = nullptr, *Pb = nullptr;            // smoother factors; prolongation blocks towards level + 1
    double *r = nullptr, *x1 = nullptr, *x = nullptr;  // restricted residual, pre-smoothed iterate, corrected iterate
    // level 0 of a two-level hierarchy: AP = (Q + shift I) P (block rows = poses, block columns = level-1 nodes) and the
    // residual after pre-smoothing, so that the post-smoothing kernel gathers from the SMALL coarse vector:
    // r - A (x1 + P xc) = (r - A x1) - (A P) xc
    Bsr AP;
    double* res1 = nullptr;
    // level 0 of a two-level hierarchy with GRAPH aggregates (ml_graph_aggregates): label of every pose, members of every
    // aggregate in discovery order, the spanning tree the prolongation is composed along, P_i^T res_i of every pose
    bool graph = false;
    int32_t *lab = nullptr, *agg_ptr = nullptr, *agg_mem = nullptr, *parent = nullptr, *pslot = nullptr;
    int32_t* mem_pos = nullptr;    // position of every pose in agg_mem (k_ml_build_P_tree_wave)
    // restriction of graph aggregates: inside the G consecutive poses a wave of k_ml_restrict owns, every RUN of poses with
    // the same aggregate is added up in the wave and leaves ONE partial sum; seg_info[i] = slot * 32 + length for the first
    // pose of a run (-1 otherwise), slots ordered by aggregate, seg_ptr[a] .. seg_ptr[a+1] = the partial sums of aggregate a
    int32_t *seg_info = nullptr, *seg_ptr = nullptr;
    int nseg = 0;  // partial sums per restriction
    int32_t* tile_perm = nullptr;  // aggregates of at most one persistent tile: pose of every (aggregate, slot), -1 = empty
    int perm_tile = 0;             // slots per aggregate in tile_perm
    int merge_cap = 0;             // graph aggregates: fragments merged up to this many poses (0: plain greedy growth)
    double* tbuf = nullptr;
    AggMap agg() const { return AggMap{graph ? lab : nullptr, k}; }
  };
  std::vector<MlLevel> ml;
  std::vector<int32_t> h_rowptr, h_colidx;  // host copy of Q's block pattern (symbolic setup of the hierarchy)
  bool ml_symbolic = false, ml_ready = false, ml_user_ks = false;
  bool ml_additive_layout = false;  // the hierarchy is the one the additive preconditioner needs (one aggregate per workgroup tile)
  // layout of the additive preconditioner's one-launch solve for this block pattern (additive_plan): lane groups per pose
  // (0: the block does not fit), slots per workgroup tile = aggregate, growth size and merge bound of the graph aggregates
  // (graph = false: index runs of `tile` poses), number of aggregates = workgroups
  struct AddPlan {
    int split = 0, tile = 0, S = 0, cap = 0, na = 0;
    bool graph = true;
  } add_plan;
  bool add_plan_known = false;
  // the aggregation the plan was found with (host arrays), reused by the symbolic setup that follows: growing and merging
  // the aggregates of a 12 500-pose block is 1.4 ms of host time
  struct AggCache {
    int S = 0, cap = 0;
    std::vector<int32_t> lab, ptr, mem, parent, pslot;
  } add_agg;
  double ml_omega = 0.7, ml_shift = 1e-1;
  double* ml_dense = nullptr;  // inverse of the coarsest operator, row-major, leading dimension ml_lda
  float* ml_dense32 = nullptr;  // its fp32 storage (what the cycle streams when ml_coarse_bits == 32)
  // lower block triangle of the (exactly symmetric) inverse, packed 64 x 64 tiles: what a two-level cycle streams in 64-bit
  // mode (k_dense_sym_apply: half the bytes); chunk table, partial-sum buffers
  double *ml_packed = nullptr, *ml_pd = nullptr, *ml_pt = nullptr;
  DenseChunk* ml_chunks = nullptr;
  int* ml_chunk_first = nullptr;
  int ml_nchunks = 0;
  bool ml_use_dense_sym() const {
    static const int env = [] {
      const char* e = std::getenv("DPGO_ML_DENSE_SYM");
      return e ? std::atoi(e) : -1;
    }();
    if (!ml_use_ap() || ml_coarse_bits != 64 || !ml_packed || env == 0) return false;
    return env == 1 || ml_lda >= 3072;  // below, the row-streaming kernel (one launch, cache-resident inverse) is as fast
  }
  int ml_coarse_bits = 64;  // 32: opt-in (dpgo_problem_multilevel_coarse_bits)
  int ml_lda = 0;
  double *ml_W = nullptr, *ml_Rx = nullptr;  // Gauss-Jordan panels (setup only)
  // DPGO_PRECOND_AUTO: the multilevel cycle is currently selected.  Decided afresh at the first "auto" use after every
  // change of Q (a function of the problem only, so repeated runs reproduce): multilevel for a block without coupling
  // to other agents -- there the tCG budget, not the trust-region boundary, ends the local solves --, block-Jacobi
  // for a block of a multi-agent problem; then hysteresis on the share of the tCG budget each solve used.
  bool auto_ml = false, auto_decided = false;
  // The cost rule of a COUPLED block the additive one-launch solve can hold (dpgo_hip.h, DPGO_PRECOND_AUTO): Q -- and with it
  // the hierarchy -- is constant across RBCD sweeps, so the set-up is paid once; everything is counted in units of a tenth of
  // a block-Jacobi product (kAutoUnits*), a function of the solves' product counts only, so that repeated runs reproduce.
  struct AutoCost {
    long long jac_units = 0;  // block-Jacobi work since Q last changed (or since the last hand-back)
    int ref = 0;              // products of the block-Jacobi solve the additive form is measured against
    int state = 0;            // 0 block-Jacobi, 1 additive on trial (its first solve), 2 additive
    int backoff = 0;          // hand-backs so far: the next trial waits for 2^backoff set-ups' worth of block-Jacobi work
    int switches = 0;         // block-Jacobi -> additive transitions since Q last changed
    int last_used = -1, last_products = 0;  // the last auto solve, as the rule saw it
    int uj = 10, ua = 18;     // unit costs of a block-Jacobi / an additive product the rule last used (auto_units_*)
  } auto_cost;
  void auto_decide() {
    if (!auto_decided) {
      auto_ml = !(has_G || C.nnzb > 0);
      auto_decided = true;
      auto_cost = AutoCost();
    }
  }
  // two-level hierarchies: level-0 post-smoothing through A P and the coarse solution (k_ml_post_ap); DPGO_ML_AP=0 disables
  bool ml_use_ap() const {
    static const bool off = [] {
      const char* e = std::getenv("DPGO_ML_AP");
      return e && std::atoi(e) == 0;
    }();
    return ml.size() == 2 && ml[0].AP.vals != nullptr && (!off || ml[0].graph);  // (graph aggregates exist in this form only)
  }
  // symmetric copy of Q for the plain SpMM on Infinity-Cache-cold blocks (k_spmm_sym): upper blocks transposed + lower references
  struct SymQ {
    int nu = 0, nl = 0;
    int32_t *urow = nullptr, *ucol = nullptr, *usrc = nullptr, *lrow = nullptr, *lcol = nullptr, *lslot = nullptr,
            *lsrc = nullptr;
    double* uvalsT = nullptr;
    int* flag = nullptr;       // device: set by k_sym_check when a lower block is not the transpose of its upper one
    bool symbolic = false;     // pattern arrays belong to the current block pattern
    bool pattern_ok = false;   // the pattern is structurally symmetric
    bool ready = false;        // uvalsT holds the current values and they passed the symmetry check
    bool values_ok = false;
    BsrSymDev dev() const { return BsrSymDev{urow, ucol, uvalsT, lrow, lcol, lslot}; }
  } sym;
  int spmm_variant = DPGO_SPMM_AUTO;
  bool tcg_sym = false;  // the fused tCG-step kernel reads the symmetric copy (resolved before a solve / a kernel probe)
  int cap_hs = kMaxGrid; // launch cap of k_tcg_hess_sym
  bool sym_wanted() const {
    if (spmm_variant == DPGO_SPMM_PLAIN || split != 1) return false;
    if (spmm_variant == DPGO_SPMM_SYMMETRIC) return true;
    // AUTO: when the tCG loop's working set (Q and eight pose vectors) no longer fits the 256 MB Infinity Cache, i.e. when
    // Q's bytes come from HBM: there the half-size storage wins (k_tcg_hess 45.5 against 49.4 us, plain product 28.4 against
    // 36.6 us at 100k poses with cold operands), while on cache-resident operands the fused kernels gain nothing
    // (DESIGN.md section 3).  DPGO_SPMM_SYMMETRIC=0/1 in the environment overrides.
    if (const char* e = std::getenv("DPGO_SPMM_SYMMETRIC")) return std::atoi(e) != 0;
    return beyond_cache();
  }
  // what the tCG loop streams besides Q and the pose vectors (the multilevel cycle's dense inverse, A P, prolongation):
  // set by the solve that last chose a preconditioner
  size_t loop_extra_bytes = 0;
  // non-temporal single-use operands: when the launch is fed from HBM (same size rule as the symmetric storage)
  bool want_stream_nt() const {
    const char* e = std::getenv("DPGO_STREAM_NT");
    return e ? std::atoi(e) != 0 : beyond_cache();
  }
  bool beyond_cache() const {
    return sizeof(double) * ((size_t)Q.nnzb * b * b + 8 * (size_t)n * T) + sizeof(int32_t) * (size_t)Q.nnzb + loop_extra_bytes >
           ((size_t)256 << 20);
  }
  // persistent whole-chip tCG kernel (blocks in the latency regime, block-Jacobi / no preconditioner): kernels/persist.h
  bool persist = false;      // enabled for this handle (by size; DPGO_PERSIST=0/1, dpgo_problem_set_persistent)
  int persist_share = 1;     // agents solved concurrently on this device (> 1: the most compact layout is preferred)
  int persist_wgs = 0, persist_split = 0, persist_mt = 0;  // geometry of the current / last launch
  int persist_reserved = 0;  // resident-slot reservation held by the running solve
  bool persist_failed_once = false;
  bool stream_nt = false;  // single-use operands of the tCG-step kernel move non-temporally (ld_stream, common.h)
  bool persist_add = false;  // the reservation is for the additive-preconditioner variant
  bool persist_stream_ordered = false;  // set for the duration of a begin / end solve (see launch_rtr_persistent)
  size_t persist_lds_attr = 0;  // dynamic LDS size the additive instance's launch attribute was last raised to
  // a solve enqueued by dpgo_optimize_device_begin and not yet collected by ..._end
  struct Pending {
    bool active = false;    // begin has been called
    bool launched = false;  // the one-launch solve is in flight (else: the solve already ran, `result` holds its outcome)
    dpgo_ropt_params resolved{};
    bool is_auto = false;
    const double* dinv = nullptr;
    double* own_x1 = nullptr;
    std::chrono::steady_clock::time_point t0;
    dpgo_ropt_result result{};
  } pending;
  PersistCtrl* pctrl = nullptr;
  unsigned long long* pgran = nullptr;  // granule table of the in-kernel all-reduce (kGranWords 8-byte words)
  unsigned gran_cleared_at = 0;         // value of `gen` when the table was last cleared
  PersistCtrl* hctrl = nullptr;  // pinned
  double* partials = nullptr;  // 5 regions of kPartialCap*kNP
  DevState* dstate = nullptr;  // 2 slots
  DevState* hstate = nullptr;  // pinned
  unsigned long long* hflag = nullptr;  // pinned, host-coherent: device-published tCG progress word
  unsigned gen = 0;
  bool saw_rtr_stop = false;  // set from the progress word in just-in-time mode
  // re-weightable edges (GNC)
  int em = 0;
  int32_t *e_p1 = nullptr, *e_p2 = nullptr, *c_ptr = nullptr, *c_edge = nullptr;
  double *e_R = nullptr, *e_t = nullptr, *e_kappa = nullptr, *e_tau = nullptr, *e_w = nullptr, *e_rsq = nullptr,
         *q_base = nullptr;
  uint8_t *e_fixed = nullptr, *c_kind = nullptr, *e_role = nullptr;
  int32_t* e_slot = nullptr;
  // contributions of shared re-weightable edges to the coupling matrix C
  int32_t *g_ptr = nullptr, *g_edge = nullptr;
  uint8_t* g_kind = nullptr;
  double* c_base = nullptr;
  int n_shared_edges = 0;
  int* e_counts = nullptr;
  EdgeDev edges() const {
    return EdgeDev{e_p1, e_p2, e_R, e_t, e_kappa, e_tau, e_fixed, e_role, e_slot, e_w, e_rsq, em};
  }
  int cur = 0;
  size_t vec_bytes() const { return (size_t)n * T * sizeof(double); }
  double* pE() const { return partials; }
  double* pA() const { return partials + 1 * kPartialCap * kNP; }
  double* pB() const { return partials + 2 * kPartialCap * kNP; }
  double* pH() const { return partials + 3 * kPartialCap * kNP; }
  int grid() const {
    const int P = (64 / b) * kWaves;
    int tiles = (n + P - 1) / P;
    if (tiles < 1) tiles = 1;
    return tiles < cap_u ? tiles : cap_u;
  }
  int cap_u = kMaxGrid, cap_h = kMaxGrid;  // launch caps of the streaming / SpMM kernel families
  // entries of partial region B (<r,r>, <z,r>) that k_tcg_hess has to sum: written by k_tcg_update (its grid) or,
  // with the fused multilevel cycle, by k_ml_post (SpMM-family grid)
  bool zr_from_post = false;
  int nb_zr() const { return zr_from_post ? grid_post() : grid(); }
  int split = 1;  // lane groups per pose in the SpMM kernels (latency layout for small blocks)
  int grid_s() const {  // SpMM kernels (k_spmm, k_grad, k_hess, k_tcg_hess)
    const int P = (64 / (b * split)) * kWaves;
    int tiles = (n + P - 1) / P;
    if (tiles < 1) tiles = 1;
    const int cap = tcg_sym ? cap_hs : cap_h;
    return tiles < cap ? tiles : cap;
  }
  // level-0 restriction / post-smoothing of the multilevel cycle: their own resident-slot counts (lighter kernels than
  // k_tcg_hess: with 4 instead of 3 waves per SIMD the 1 563 tiles of the 100k block take 2 rounds instead of 3)
  int cap_restrict = kMaxGrid, cap_post = kMaxGrid;
  int grid_tiles(int cap) const {
    const int P = (64 / (b * split)) * kWaves;
    int tiles = (n + P - 1) / P;
    if (tiles < 1) tiles = 1;
    return tiles < cap ? tiles : cap;
  }
  int grid_restrict() const { return grid_tiles(cap_restrict); }
  int grid_post() const { return grid_tiles(cap_post); }
  int grid_spmm() const {  // plain k_spmm: no partial sums, higher occupancy than the fused tCG kernel
    const int P = (64 / (b * split)) * kWaves;
    int tiles = (n + P - 1) / P;
    if (tiles < 1) tiles = 1;
    return tiles < kMaxGrid ? tiles : kMaxGrid;
  }
  int grid_flat() const {  // elementwise kernels
    size_t total = (size_t)n * T;
    size_t g = (total + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    return g < (size_t)kMaxGrid ? (int)g : kMaxGrid;
  }
};

namespace {

int set_device(dpgo_problem_s* p) {
  HIPC(hipSetDevice(p->device));
  return DPGO_OK;
}

int upload_bsr(Bsr& m, int nrows, int ncols, int nnzb, int b, const int32_t* rowptr, const int32_t* colidx,
               const double* vals, hipStream_t s) {
  CHK(free_bsr(m));
  m.nrows = nrows;
  m.ncols = ncols;
  m.nnzb = nnzb;
  HIPC(hipMalloc(&m.rowptr, sizeof(int32_t) * (nrows + 1)));
  HIPC(hipMalloc(&m.colidx, sizeof(int32_t) * (nnzb > 0 ? nnzb : 1)));
  HIPC(hipMalloc(&m.vals, sizeof(double) * (size_t)(nnzb > 0 ? nnzb : 1) * b * b));
  HIPC(hipMemcpyAsync(m.rowptr, rowptr, sizeof(int32_t) * (nrows + 1), hipMemcpyHostToDevice, s));
  if (nnzb > 0) {
    HIPC(hipMemcpyAsync(m.colidx, colidx, sizeof(int32_t) * nnzb, hipMemcpyHostToDevice, s));
    if (vals) HIPC(hipMemcpyAsync(m.vals, vals, sizeof(double) * (size_t)nnzb * b * b, hipMemcpyHostToDevice, s));
  }
  HIPC(hipStreamSynchronize(s));
  return DPGO_OK;
}

template <class Tp>
int upload(Tp** dst, const Tp* src, size_t count, hipStream_t s) {
  HIPC(hipMalloc(dst, sizeof(Tp) * (count > 0 ? count : 1)));
  if (count > 0) HIPC(hipMemcpyAsync(*dst, src, sizeof(Tp) * count, hipMemcpyHostToDevice, s));
  return DPGO_OK;
}

struct TmpDev {
  std::vector<void*> ptrs;
  ~TmpDev() {
    for (auto q : ptrs) (void)hipFree(q);
  }
  int alloc(double** out, size_t bytes) {
    HIPC(hipMalloc(out, bytes));
    ptrs.push_back(*out);
    return DPGO_OK;
  }
};
int validate_bsr(int nrows, int ncols, int nnzb, const int32_t* rowptr, const int32_t* colidx, bool need_diag) {
  if (!rowptr || nnzb < 0 || (nnzb > 0 && !colidx)) return fail(DPGO_ERR_INVALID, "null BSR arrays");
  if (rowptr[0] != 0 || rowptr[nrows] != nnzb) return fail(DPGO_ERR_INVALID, "BSR rowptr does not span nnzb");
  for (int i = 0; i < nrows; ++i) {
    if (rowptr[i + 1] < rowptr[i]) return fail(DPGO_ERR_INVALID, "BSR rowptr not monotone");
    bool diag = false;
    for (int t = rowptr[i]; t < rowptr[i + 1]; ++t) {
      if (colidx[t] < 0 || colidx[t] >= ncols) return fail(DPGO_ERR_INVALID, "BSR column index out of range");
      if (t > rowptr[i] && colidx[t] <= colidx[t - 1])
        return fail(DPGO_ERR_INVALID, "BSR column indices must be sorted and unique within a block row");
      if (colidx[t] == i) diag = true;
    }
    if (need_diag && !diag) return fail(DPGO_ERR_INVALID, "BSR block row without diagonal block");
  }
  return DPGO_OK;
}

int build_dinv(dpgo_problem_s* p, double shift) {
  if (!p->Q.vals) return fail(DPGO_ERR_STATE, "Q not set");
  if (p->dinv_shift == shift) return DPGO_OK;
  const int g = (p->n + kBlock - 1) / kBlock;
  if (p->d == 2)
    hipLaunchKernelGGL(k_build_dinv<2>, dim3(g), dim3(kBlock), 0, p->stream, p->Q.dev(), shift, p->dinv, p->n);
  else
    hipLaunchKernelGGL(k_build_dinv<3>, dim3(g), dim3(kBlock), 0, p->stream, p->Q.dev(), shift, p->dinv, p->n);
  HIPC(hipGetLastError());
  p->dinv_shift = shift;
  return DPGO_OK;
}

int poll_state(dpgo_problem_s* p) {
  HIPC(hipMemcpyAsync(p->hstate, p->dstate + p->cur, sizeof(DevState), hipMemcpyDeviceToHost, p->stream));
  HIPC(hipStreamSynchronize(p->stream));
  return DPGO_OK;
}

int push_state(dpgo_problem_s* p) {
  HIPC(hipMemcpyAsync(p->dstate + 0, p->hstate, sizeof(DevState), hipMemcpyHostToDevice, p->stream));
  HIPC(hipMemcpyAsync(p->dstate + 1, p->hstate, sizeof(DevState), hipMemcpyHostToDevice, p->stream));
  HIPC(hipStreamSynchronize(p->stream));
  return DPGO_OK;
}

// ---- symmetric copy of Q (plain SpMM on cold blocks) ----
void sym_free(dpgo_problem_s* p) {
  auto& S = p->sym;
  void* ptrs[] = {S.urow, S.ucol, S.usrc, S.lrow, S.lcol, S.lslot, S.lsrc, S.uvalsT, S.flag};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  S = dpgo_problem_s::SymQ();
  p->tcg_sym = false;
}

template <typename T>
int sym_upload(T** dst, const std::vector<T>& v, hipStream_t stream) {
  HIPC(hipMalloc(dst, sizeof(T) * std::max<size_t>(1, v.size())));
  if (!v.empty()) HIPC(hipMemcpyAsync(*dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice, stream));
  return DPGO_OK;
}

// pattern arrays from the host copy of Q's block pattern; a pattern that is not structurally symmetric leaves
// pattern_ok = false (the plain kernel stays in use), it is not an error
int sym_symbolic_setup(dpgo_problem_s* p) {
  sym_free(p);
  auto& S = p->sym;
  S.symbolic = true;
  const int n = p->n;
  if ((int)p->h_rowptr.size() != n + 1) return DPGO_OK;
  const auto& rp = p->h_rowptr;
  const auto& ci = p->h_colidx;
  std::vector<int32_t> urow(n + 1, 0), lrow(n + 1, 0), ucol, usrc, lcol, lslot, lsrc;
  ucol.reserve(ci.size() / 2 + n);
  for (int i = 0; i < n; ++i) {
    for (int t = rp[i]; t < rp[i + 1]; ++t)
      if (ci[t] >= i) {
        ucol.push_back(ci[t]);
        usrc.push_back(t);
      }
    urow[i + 1] = (int32_t)ucol.size();
  }
  for (int i = 0; i < n; ++i) {
    for (int t = rp[i]; t < rp[i + 1]; ++t) {
      const int j = ci[t];
      if (j >= i) break;  // columns are sorted
      const auto b0 = ucol.begin() + urow[j], b1 = ucol.begin() + urow[j + 1];
      const auto it = std::lower_bound(b0, b1, (int32_t)i);
      if (it == b1 || *it != i) return DPGO_OK;  // block (i, j) without block (j, i)
      lcol.push_back(j);
      lslot.push_back((int32_t)(it - ucol.begin()));
      lsrc.push_back(t);
    }
    lrow[i + 1] = (int32_t)lcol.size();
  }
  if (ucol.size() + lcol.size() != ci.size()) return DPGO_OK;
  if (2 * lcol.size() + (size_t)n != ci.size()) return DPGO_OK;  // an upper block without its lower one
  S.nu = (int)ucol.size();
  S.nl = (int)lcol.size();
  CHK(sym_upload(&S.urow, urow, p->stream));
  CHK(sym_upload(&S.ucol, ucol, p->stream));
  CHK(sym_upload(&S.usrc, usrc, p->stream));
  CHK(sym_upload(&S.lrow, lrow, p->stream));
  CHK(sym_upload(&S.lcol, lcol, p->stream));
  CHK(sym_upload(&S.lslot, lslot, p->stream));
  CHK(sym_upload(&S.lsrc, lsrc, p->stream));
  HIPC(hipMalloc(&S.uvalsT, sizeof(double) * (size_t)std::max(1, S.nu) * p->b * p->b));
  HIPC(hipMalloc(&S.flag, sizeof(int)));
  HIPC(hipStreamSynchronize(p->stream));  // the host vectors go out of scope
  S.pattern_ok = true;
  return DPGO_OK;
}

// true when the symmetric copy is usable for Q's current values (refreshes it when they changed)
int sym_ensure(dpgo_problem_s* p, bool* usable) {
  *usable = false;
  auto& S = p->sym;
  if (!S.symbolic) CHK(sym_symbolic_setup(p));
  if (!S.pattern_ok) return DPGO_OK;
  if (!S.ready) {
    const size_t total = (size_t)S.nu * p->b * p->b;
    const int g = (int)std::min<size_t>(kMaxGrid, (total + kBlock - 1) / kBlock);
    HIPC(hipMemsetAsync(S.flag, 0, sizeof(int), p->stream));
    if (p->d == 2) {
      hipLaunchKernelGGL(k_sym_refresh<2>, dim3(g), dim3(kBlock), 0, p->stream, p->Q.vals, S.usrc, S.uvalsT, S.nu);
      hipLaunchKernelGGL(k_sym_check<2>, dim3(g), dim3(kBlock), 0, p->stream, p->Q.vals, S.lsrc, S.lslot, S.uvalsT, S.nl,
                         S.flag);
    } else {
      hipLaunchKernelGGL(k_sym_refresh<3>, dim3(g), dim3(kBlock), 0, p->stream, p->Q.vals, S.usrc, S.uvalsT, S.nu);
      hipLaunchKernelGGL(k_sym_check<3>, dim3(g), dim3(kBlock), 0, p->stream, p->Q.vals, S.lsrc, S.lslot, S.uvalsT, S.nl,
                         S.flag);
    }
    HIPC(hipGetLastError());
    int bad = 0;
    HIPC(hipMemcpyAsync(&bad, S.flag, sizeof(int), hipMemcpyDeviceToHost, p->stream));
    HIPC(hipStreamSynchronize(p->stream));
    S.values_ok = (bad == 0);
    S.ready = true;
  }
  *usable = S.values_ok;
  return DPGO_OK;
}

// ---- kernel launch helpers (templated on D, R through DISPATCH) ----
int launch_spmm_sym(dpgo_problem_s* p, const BsrSymDev& M, const double* V, const double* Gadd, double* OUT) {
  const int g = p->grid_spmm();
  DISPATCH(p->d, p->r,
           {
             if (p->want_stream_nt())
               hipLaunchKernelGGL((k_spmm_sym<D, R, 1>), dim3(g), dim3(kBlock), 0, p->stream, M, V, Gadd, OUT, p->n);
             else
               hipLaunchKernelGGL((k_spmm_sym<D, R, 0>), dim3(g), dim3(kBlock), 0, p->stream, M, V, Gadd, OUT, p->n);
           });
  HIPC(hipGetLastError());
  return DPGO_OK;
}

int launch_spmm(dpgo_problem_s* p, const Bsr& M, const double* V, const double* Gadd, double* OUT, int nrows = -1) {
  if (&M == &p->Q && nrows < 0 && p->sym_wanted()) {
    bool usable = false;
    CHK(sym_ensure(p, &usable));
    if (usable) return launch_spmm_sym(p, p->sym.dev(), V, Gadd, OUT);
  }
  const int rows = nrows >= 0 ? nrows : p->n;
  int g = p->grid_spmm();
  if (nrows >= 0) {  // rectangular operator with its own row count (restriction)
    const int P = (64 / (p->b * p->split)) * kWaves;
    g = std::max(1, std::min(kMaxGrid, (rows + P - 1) / P));
  }
  DISPATCH(p->d, p->r, {
    if (p->want_stream_nt() && p->split == 1 && &M == &p->Q)
      hipLaunchKernelGGL((k_spmm<D, R, 1, 1>), dim3(g), dim3(kBlock), 0, p->stream, M.dev(), V, Gadd, OUT, rows);
    else
      LAUNCH_SPLIT(p, k_spmm, g, M.dev(), V, Gadd, OUT, rows);
  });
  HIPC(hipGetLastError());
  return DPGO_OK;
}

bool outer_sym_enabled() {  // tuning knob: DPGO_OUTER_SYM=0 keeps the outer iteration on the plain copy of Q
  static const bool on = [] { const char* e = std::getenv("DPGO_OUTER_SYM"); return !e || std::atoi(e) != 0; }();
  return on;
}
// `sym`: inside a solve whose tCG-step kernel reads the symmetric copy of Q (p->tcg_sym, valid for the duration of the
// solve) the gradient and the rho-test Hessian read it too: 7 us less per launch at 100k poses with cold operands, and the
// outer iteration no longer streams the 91 MB of the plain copy through the Infinity Cache the tCG loop lives in.
int launch_grad(dpgo_problem_s* p, const double* X, double* RG, double* S, double* EG,
                const DevState* st = nullptr, bool sym = false) {
  const double* Gm = p->has_G ? p->G : nullptr;
  if (sym && p->split == 1) {
    DISPATCH(p->d, p->r, hipLaunchKernelGGL((k_grad<D, R, 1, BsrSymDev>), dim3(p->grid_s()), dim3(kBlock), 0, p->stream,
                                            p->sym.dev(), X, Gm, RG, S, EG, p->pE(), st, p->n));
  } else {
    DISPATCH(p->d, p->r, LAUNCH_SPLIT(p, k_grad, p->grid_s(), p->Q.dev(), X, Gm, RG, S, EG, p->pE(), st, p->n));
  }
  HIPC(hipGetLastError());
  return DPGO_OK;
}

int launch_hess(dpgo_problem_s* p, const double* X, const double* S, const double* V, const double* Gdot,
                double* HV, double* partials, const DevState* st, int check_tcg, bool sym = false) {
  if (sym && p->split == 1) {
    DISPATCH(p->d, p->r, hipLaunchKernelGGL((k_hess<D, R, 1, BsrSymDev>), dim3(p->grid_s()), dim3(kBlock), 0, p->stream,
                                            p->sym.dev(), X, S, V, Gdot, HV, partials, st, check_tcg, p->n));
  } else {
    DISPATCH(p->d, p->r,
             LAUNCH_SPLIT(p, k_hess, p->grid_s(), p->Q.dev(), X, S, V, Gdot, HV, partials, st, check_tcg, p->n));
  }
  HIPC(hipGetLastError());
  return DPGO_OK;
}

int launch_tcg_update(dpgo_problem_s* p, const double* dinv, int first, double* z_out = nullptr,
                      double ml_omega = 0.0) {
  const int g = p->grid();
  double* zt = z_out ? z_out : p->z;
  DISPATCH(p->d, p->r, {
    if constexpr (Span<D, R, 1>::kOk)
      hipLaunchKernelGGL((k_tcg_update_span<D, R>), dim3(g), dim3(kBlock), 0, p->stream, p->x1, p->g1, dinv, p->delta,
                         p->Hd, p->eta, p->rr, zt, p->pA(), p->grid_s(), p->pB(), p->dstate + p->cur,
                         p->dstate + (p->cur ^ 1), first, p->n, p->hflag, p->gen, ml_omega);
    else
      hipLaunchKernelGGL((k_tcg_update<D, R>), dim3(g), dim3(kBlock), 0, p->stream, p->x1, p->g1, dinv, p->delta,
                         p->Hd, p->eta, p->rr, zt, p->pA(), p->grid_s(), p->pB(), p->dstate + p->cur,
                         p->dstate + (p->cur ^ 1), first, p->n, p->hflag, p->gen, ml_omega);
  });
  HIPC(hipGetLastError());
  p->cur ^= 1;
  return DPGO_OK;
}

// fused direction update + Riemannian Hessian-vector product (one tCG step)
// the tCG-step kernel: span variant whenever the pose tile size is even (all 3-D cases)
#define LAUNCH_TCG_HESS(p, SIN, SOUT, FIRST, HFLAG, GEN)                                                          \
  do {                                                                                                            \
    if constexpr (Span<D, R, 1>::kOk) {                                                                           \
      if ((p)->tcg_sym && (p)->stream_nt)                                                                         \
        hipLaunchKernelGGL((k_tcg_hess_sym<D, R, 1>), dim3((p)->grid_s()), dim3(kBlock), 0, (p)->stream,          \
                           (p)->sym.dev(), (p)->x1, (p)->S1, (p)->z, (p)->delta, (p)->Hd, (p)->pB(), (p)->nb_zr(), \
                           (p)->pA(), SIN, SOUT, FIRST, (p)->n, HFLAG, GEN);                                      \
      else if ((p)->tcg_sym)                                                                                      \
        hipLaunchKernelGGL((k_tcg_hess_sym<D, R, 0>), dim3((p)->grid_s()), dim3(kBlock), 0, (p)->stream,          \
                           (p)->sym.dev(), (p)->x1, (p)->S1, (p)->z, (p)->delta, (p)->Hd, (p)->pB(), (p)->nb_zr(), \
                           (p)->pA(), SIN, SOUT, FIRST, (p)->n, HFLAG, GEN);                                      \
      else if ((p)->stream_nt && (p)->split == 1)                                                                 \
        hipLaunchKernelGGL((k_tcg_hess_span<D, R, 1, 1>), dim3((p)->grid_s()), dim3(kBlock), 0, (p)->stream,      \
                           (p)->Q.dev(), (p)->x1, (p)->S1, (p)->z, (p)->delta, (p)->Hd, (p)->pB(), (p)->nb_zr(),   \
                           (p)->pA(), SIN, SOUT, FIRST, (p)->n, HFLAG, GEN);                                      \
      else                                                                                                        \
      LAUNCH_SPLIT(p, k_tcg_hess_span, (p)->grid_s(), (p)->Q.dev(), (p)->x1, (p)->S1, (p)->z, (p)->delta, (p)->Hd, \
                   (p)->pB(), (p)->nb_zr(), (p)->pA(), SIN, SOUT, FIRST, (p)->n, HFLAG, GEN);                   \
    } else                                                                                                        \
      LAUNCH_SPLIT(p, k_tcg_hess, (p)->grid_s(), (p)->Q.dev(), (p)->x1, (p)->S1, (p)->z, (p)->delta, (p)->Hd,      \
                   (p)->pB(), (p)->nb_zr(), (p)->pA(), SIN, SOUT, FIRST, (p)->n, HFLAG, GEN);                   \
  } while (0)

int launch_tcg_hess(dpgo_problem_s* p, int first) {
  DISPATCH(p->d, p->r, LAUNCH_TCG_HESS(p, p->dstate + p->cur, p->dstate + (p->cur ^ 1), first, p->hflag, p->gen));
  HIPC(hipGetLastError());
  p->cur ^= 1;
  return DPGO_OK;
}

// which storage of Q the tCG-step kernel of the coming launches reads (Q does not change inside a solve)
int resolve_tcg_storage(dpgo_problem_s* p) {
  p->tcg_sym = false;
  p->stream_nt = p->want_stream_nt();
  bool span = false;
  DISPATCH(p->d, p->r, { span = Span<D, R, 1>::kOk; });
  if (!span || !p->sym_wanted()) return DPGO_OK;
  bool usable = false;
  CHK(sym_ensure(p, &usable));
  p->tcg_sym = usable;
  return DPGO_OK;
}

int launch_retract(dpgo_problem_s* p, const double* X, const double* eta, double scale, double* X2,
                   const DevState* st) {
  DISPATCH(p->d, p->r, hipLaunchKernelGGL((k_retract<D, R>), dim3(p->grid()), dim3(kBlock), 0, p->stream, X, eta,
                                          scale, X2, st, p->n));
  HIPC(hipGetLastError());
  return DPGO_OK;
}

int launch_rtr_update(dpgo_problem_s* p) {
  const int g = p->grid_s();
  DISPATCH(p->d, p->r,
           hipLaunchKernelGGL((k_rtr_update<D, R>), dim3(p->grid_flat()), dim3(kBlock), 0, p->stream, p->x1, p->x2,
                              p->g1, p->g2, p->S1, p->S2, p->pE(), g, p->pH(), g, p->dstate + p->cur,
                              p->dstate + (p->cur ^ 1), p->n));
  HIPC(hipGetLastError());
  p->cur ^= 1;
  return DPGO_OK;
}

int launch_precond(dpgo_problem_s* p, const double* X, const double* V, const double* dinv, double* Z) {
  DISPATCH(p->d, p->r, hipLaunchKernelGGL((k_precond<D, R>), dim3(p->grid()), dim3(kBlock), 0, p->stream, X, V,
                                          dinv, Z, p->n));
  HIPC(hipGetLastError());
  return DPGO_OK;
}

int launch_rtr_begin(dpgo_problem_s* p, double tol, double Delta0, double Dmax, int max_inner, int tiny) {
  hipLaunchKernelGGL(k_rtr_begin, dim3(1), dim3(kBlock), 0, p->stream, p->pE(), p->grid_s(), p->dstate, tol, Delta0,
                     Dmax, max_inner, tiny);
  HIPC(hipGetLastError());
  p->cur = 0;
  return DPGO_OK;
}

struct Counters {
  int spmm = 0;
  bool vcycle_for_additive = false;  // an outer iteration of an "additive" solve ran the V-cycle instead
};

// ---------------------------------------------------------------------------------------------------------
// Multilevel preconditioner: hierarchy setup (symbolic on the host once per block pattern, numeric on the device
// for every new set of Q values) and the per-iteration launches.  DESIGN.md section 5.
int ml_tile(int b, int split) { return (64 / (b * split)) * kWaves; }
// tiles of the additive preconditioner's layout (4 lane groups per pose, one tile = one aggregate per workgroup)
int additive_tile(const dpgo_problem_s* p) { return ml_tile(p->b, 4); }
int ml_level_split(int n) { return n < 40000 ? 4 : 1; }

// Aggregate sizes per coarsening.  Every k must divide the workgroup tile of its level (fused restriction); the
// coarsest operator is a dense inverse of at most kMlDense unknowns (Infinity-Cache resident), kMlDenseMax if that is
// what it takes to get there in one coarsening; otherwise one more level.
constexpr int kMlDense = 3200, kMlDenseMax = 6400;
// Two-level hierarchies use GRAPH aggregates (a single negative entry -S: breadth-first-grown aggregates of at most S
// poses, ml_graph_aggregates) whenever one coarsening with S <= kMlGraphMax reaches a dense level of about 2 500
// unknowns: compact aggregates need 40-60 % of the Hessian-vector products that index runs of the same size need
// (DESIGN.md section 5), and the dense level can then be small.  DPGO_ML_GRAPH=0: index runs as before.
constexpr int kMlGraphMax = 512, kMlGraphUnknownsPerPose = 1600;
int ml_default_graph_size(int n, int b) {
  if (const char* e = std::getenv("DPGO_ML_GRAPH"))
    if (std::atoi(e) == 0) return 0;
  if (const char* e = std::getenv("DPGO_ML_GRAPH_SIZE"))  // experiments: force the size
    if (std::atoi(e) >= 2) return std::atoi(e);
  const long long S = std::max<long long>(4, ((long long)n * b + kMlGraphUnknownsPerPose - 1) / kMlGraphUnknownsPerPose);
  return S <= kMlGraphMax ? (int)S : 0;
}
// Blocks whose plain greedy growth would use aggregates of >= kMlMergeFrom poses (n (d+1) >= ~100 000 unknowns: >= 25 600
// poses in 3-D) grow them to S = ceil(n (d+1) / 2 200) instead and MERGE the growth's fragments up to 3 S / 2
// (ml_merge_small_aggregates): the aggregates come out uniform (mean ~ S instead of ~0.55 S with a tail of fragments), the
// same coarse-space quality needs a quarter fewer of them -- 100k poses: 546 aggregates / 70 products to |rgrad| < 1e-2
// against 732 / 77, a dense level of 38 MB instead of 69 MB; 25k: 536 / 87 against 589 / 93 (oracle, round 4).  Smaller
// blocks keep the plain growth (same product counts either way; their hierarchies are what the committed vectors pin).
constexpr int kMlMergeFrom = 64, kMlMergedUnknownsPerPose = 2200;
std::vector<int> ml_default_ks(int n, int b, int split0) {
  if (const int S = ml_default_graph_size(n, b)) {
    const bool forced = std::getenv("DPGO_ML_GRAPH_SIZE") != nullptr;
    if (!forced && S >= kMlMergeFrom) {
      const int Sm = (int)(((long long)n * b + kMlMergedUnknownsPerPose - 1) / kMlMergedUnknownsPerPose);
      const int cap = Sm + Sm / 2;
      if (cap <= kMlGraphMax) return std::vector<int>{-Sm, -cap};
    }
    return std::vector<int>{-S};
  }
  std::vector<int> ks;
  int cur = n, split = split0;
  for (int guard = 0; guard < 16; ++guard) {
    const int P = ml_tile(b, split);
    int pick = 0;
    for (int limit : {kMlDense, kMlDenseMax}) {
      for (int k = 4; k <= P && !pick; ++k)
        if (P % k == 0 && (long long)((cur + k - 1) / k) * b <= limit) pick = k;
      if (pick) break;
    }
    if (pick) {
      ks.push_back(pick);
      return ks;
    }
    int k = 2;
    for (int c = 2; c <= 8; ++c)
      if (P % c == 0) k = c;
    ks.push_back(k);
    cur = (cur + k - 1) / k;
    split = ml_level_split(cur);
  }
  return ks;
}

void ml_free(dpgo_problem_s* p) {
  p->ml_additive_layout = false;
  for (auto& L : p->ml) {
    free_bsr(L.A);
    free_bsr(L.AP);
    void* ptrs[] = {L.slot_row, L.dinv, L.Pb, L.r, L.x1, L.x, L.res1, L.lab, L.agg_ptr, L.agg_mem, L.parent, L.pslot, L.tbuf, L.tile_perm, L.mem_pos,
                    L.seg_info, L.seg_ptr};
    for (void* q : ptrs)
      if (q) (void)hipFree(q);
  }
  p->ml.clear();
  void* ptrs[] = {p->ml_dense, p->ml_W, p->ml_Rx, p->ml_dense32, p->ml_packed, p->ml_pd, p->ml_pt, p->ml_chunks,
                  p->ml_chunk_first};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  p->ml_dense = p->ml_W = p->ml_Rx = nullptr;
  p->ml_dense32 = nullptr;
  p->ml_packed = p->ml_pd = p->ml_pt = nullptr;
  p->ml_chunks = nullptr;
  p->ml_chunk_first = nullptr;
  p->ml_nchunks = 0;
  p->ml_lda = 0;
  p->ml_symbolic = p->ml_ready = false;
}

// Symbolic setup: level sizes, block patterns of the Galerkin operators, buffers.
// Graph aggregates of at most S nodes, grown greedily: seeds in index order; a seed's aggregate takes unassigned nodes in
// breadth-first order (queue; a node's neighbours in the order of its block row) until it holds S.  lab = aggregate of
// every node, mem / ptr = members in discovery order, parent / pslot = the breadth-first tree (slot of block
// (parent, node) in the pattern).  Restated in oracle/dpgo_oracle.py (amg_graph_aggregates).
int ml_graph_aggregates(const std::vector<int32_t>& rowptr, const std::vector<int32_t>& colidx, int n, int S,
                        std::vector<int32_t>& lab, std::vector<int32_t>& ptr, std::vector<int32_t>& mem,
                        std::vector<int32_t>& parent, std::vector<int32_t>& pslot) {
  lab.assign(n, -1);
  parent.assign(n, -1);
  pslot.assign(n, 0);
  mem.clear();
  mem.reserve(n);
  ptr.assign(1, 0);
  int na = 0;
  for (int s = 0; s < n; ++s) {
    if (lab[s] >= 0) continue;
    const size_t first = mem.size();
    lab[s] = na;
    mem.push_back(s);
    for (size_t head = first; head < mem.size() && (int)(mem.size() - first) < S; ++head) {
      const int u = mem[head];
      for (int t = rowptr[u]; t < rowptr[u + 1] && (int)(mem.size() - first) < S; ++t) {
        const int v = colidx[t];
        if (lab[v] >= 0) continue;
        lab[v] = na;
        parent[v] = u;
        pslot[v] = t;
        mem.push_back(v);
      }
    }
    ptr.push_back((int32_t)mem.size());
    ++na;
  }
  return na;
}

// The greedy growth leaves fragments (pockets between full aggregates); where an aggregate is a WORKGROUP of the one-launch
// solve (additive preconditioner) every fragment costs a whole workgroup.  Passes over the aggregates in index order until
// nothing changes: an aggregate of at most S / 2 nodes joins the neighbouring aggregate (one it shares a block with) it has
// the most blocks in common with among those that still have room (sizes add up to at most `cap`; ties: the lower index).
// Afterwards the aggregates are renumbered in the order of their smallest member and every aggregate's breadth-first tree
// is rebuilt from that member (neighbours in block-row order).  In place; returns the number of aggregates.  Restated in
// oracle/dpgo_oracle.py (amg_merge_small_aggregates).
int ml_merge_small_aggregates(const std::vector<int32_t>& rowptr, const std::vector<int32_t>& colidx, int n, int S, int cap,
                              std::vector<int32_t>& lab, std::vector<int32_t>& ptr, std::vector<int32_t>& mem,
                              std::vector<int32_t>& parent, std::vector<int32_t>& pslot) {
  const int na = (int)ptr.size() - 1;
  std::vector<std::vector<int32_t>> members(na);
  for (int a = 0; a < na; ++a) members[a].assign(mem.begin() + ptr[a], mem.begin() + ptr[a + 1]);
  std::vector<int> cnt(na, 0);
  std::vector<int> touched;
  for (bool changed = true; changed;) {
    changed = false;
    for (int a = 0; a < na; ++a) {
      if (members[a].empty() || 2 * (int)members[a].size() > S) continue;
      touched.clear();
      for (int i : members[a])
        for (int t = rowptr[i]; t < rowptr[i + 1]; ++t) {
          const int c = lab[colidx[t]];
          if (c == a) continue;
          if (cnt[c]++ == 0) touched.push_back(c);
        }
      std::sort(touched.begin(), touched.end());
      int best = -1, best_n = 0;
      for (int c : touched) {
        if ((int)(members[c].size() + members[a].size()) <= cap && cnt[c] > best_n) best = c, best_n = cnt[c];
        cnt[c] = 0;
      }
      if (best >= 0) {
        members[best].insert(members[best].end(), members[a].begin(), members[a].end());
        for (int i : members[a]) lab[i] = best;
        members[a].clear();
        changed = true;
      }
    }
  }
  std::vector<int> alive;
  for (int a = 0; a < na; ++a)
    if (!members[a].empty()) {
      std::sort(members[a].begin(), members[a].end());
      alive.push_back(a);
    }
  std::sort(alive.begin(), alive.end(), [&](int x, int y) { return members[x][0] < members[y][0]; });
  std::vector<int32_t> new_lab(n, -1);
  parent.assign(n, -1);
  pslot.assign(n, 0);
  mem.clear();
  ptr.assign(1, 0);
  for (size_t k = 0; k < alive.size(); ++k) {
    const int a = alive[k];
    // (a merged aggregate is connected by construction, so the search from its smallest member reaches everything; should
    // the pattern not be symmetric, the members it misses become further roots in index order)
    for (int root : members[a]) {
      if (new_lab[root] >= 0) continue;
      size_t head = mem.size();
      new_lab[root] = (int32_t)k;
      mem.push_back(root);
      for (; head < mem.size(); ++head) {
        const int u = mem[head];
        for (int t = rowptr[u]; t < rowptr[u + 1]; ++t) {
          const int v = colidx[t];
          if (lab[v] != a || new_lab[v] >= 0) continue;
          new_lab[v] = (int32_t)k;
          parent[v] = u;
          pslot[v] = t;
          mem.push_back(v);
        }
      }
    }
    ptr.push_back((int32_t)mem.size());
  }
  lab.swap(new_lab);
  return (int)alive.size();
}

// ks_in: aggregate sizes per coarsening; {-S}: two levels, graph aggregates of at most S poses; {-S, -cap}: the same with
// the fragments of the greedy growth merged up to `cap` poses (ml_merge_small_aggregates).  perm_tile > 0 (graph
// aggregates): also build the (aggregate, slot) -> pose table of the additive preconditioner's persistent layout with
// that many slots per aggregate.
int ml_symbolic_setup(dpgo_problem_s* p, const std::vector<int>& ks_in, int perm_tile = 0) {
  ml_free(p);
  if ((int)p->h_rowptr.size() != p->n + 1) return fail(DPGO_ERR_STATE, "multilevel: Q's block pattern is not set");
  const int b = p->b, bb = b * b;
  const size_t tb = sizeof(double) * p->T;
  std::vector<int32_t> rowptr = p->h_rowptr, colidx = p->h_colidx;
  int cur = p->n;
  // a single negative entry -S: two levels, graph aggregates of at most S poses; two negative entries: merged up to -ks[1]
  const bool merged = ks_in.size() == 2 && ks_in[0] < 0 && ks_in[1] < 0;
  const bool graph = (ks_in.size() == 1 && ks_in[0] < 0) || merged;
  std::vector<int> ks = ks_in;
  if (merged) ks.pop_back();
  if (graph) ks[0] = -ks_in[0];
  const int merge_cap = merged ? -ks_in[1] : 0;
  if (merged && merge_cap < ks[0]) return fail(DPGO_ERR_INVALID, "multilevel: the merge bound is at least the growth size");
  for (int k : ks)
    if (k < 0) return fail(DPGO_ERR_INVALID, "multilevel: graph aggregates (a negative size) make a two-level hierarchy");
  p->ml.resize(ks.size() + 1);
  for (size_t l = 0; l <= ks.size(); ++l) {
    auto& L = p->ml[l];
    L.n = cur;
    L.split = (l == 0) ? p->split : ml_level_split(cur);
    L.k = (l < ks.size()) ? ks[l] : 0;
    if (l == 0 && graph) {
      if (L.k < 2) return fail(DPGO_ERR_INVALID, "multilevel: graph aggregates hold at least 2 poses");
      std::vector<int32_t> lab, ptr, mem, parent, pslot;
      int na;
      if (p->add_plan_known && p->add_agg.S == L.k && p->add_agg.cap == merge_cap && (int)p->add_agg.lab.size() == cur) {
        const auto& A = p->add_agg;  // (the additive plan of this pattern was found with exactly these aggregates)
        lab = A.lab, ptr = A.ptr, mem = A.mem, parent = A.parent, pslot = A.pslot;
        na = (int)ptr.size() - 1;
      } else {
        na = ml_graph_aggregates(rowptr, colidx, cur, L.k, lab, ptr, mem, parent, pslot);
        if (merge_cap) na = ml_merge_small_aggregates(rowptr, colidx, cur, L.k, merge_cap, lab, ptr, mem, parent, pslot);
      }
      L.graph = true;
      L.merge_cap = merge_cap;
      CHK(upload(&L.lab, lab.data(), lab.size(), p->stream));
      CHK(upload(&L.agg_ptr, ptr.data(), ptr.size(), p->stream));
      CHK(upload(&L.agg_mem, mem.data(), mem.size(), p->stream));
      CHK(upload(&L.parent, parent.data(), parent.size(), p->stream));
      CHK(upload(&L.pslot, pslot.data(), pslot.size(), p->stream));
      std::vector<int32_t> mpos(cur);
      for (int m = 0; m < cur; ++m) mpos[mem[m]] = m;
      CHK(upload(&L.mem_pos, mpos.data(), mpos.size(), p->stream));
      std::vector<int32_t> seg_info(cur, -1), seg_ptr(na + 1, 0);
      {  // runs of equal labels inside the level-0 kernels' wave chunks (G consecutive poses)
        const int G = 64 / (b * L.split);
        std::vector<std::pair<int32_t, int32_t>> runs;  // (aggregate, first pose), in pose order
        for (int i = 0; i < cur;) {
          int j = i + 1;
          while (j < cur && j % G != 0 && lab[j] == lab[i]) ++j;
          runs.emplace_back(lab[i], i);
          seg_info[i] = j - i;  // (length for now)
          i = j;
        }
        std::stable_sort(runs.begin(), runs.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
        for (size_t q = 0; q < runs.size(); ++q) {
          seg_info[runs[q].second] += (int32_t)q * 32;
          seg_ptr[runs[q].first + 1] += 1;
        }
        for (int a = 0; a < na; ++a) seg_ptr[a + 1] += seg_ptr[a];
        L.nseg = (int)runs.size();
      }
      CHK(upload(&L.seg_info, seg_info.data(), seg_info.size(), p->stream));
      CHK(upload(&L.seg_ptr, seg_ptr.data(), seg_ptr.size(), p->stream));
      std::vector<int32_t> tperm;
      // the layout of the additive preconditioner's persistent kernel: aggregate = workgroup tile of `perm_tile` slots
      if (!perm_tile && !merge_cap && L.k == additive_tile(p)) perm_tile = L.k;
      if (perm_tile) {
        if (std::max(L.k, merge_cap) > perm_tile) return fail(DPGO_ERR_INVALID, "multilevel: aggregates larger than the tile");
        tperm.assign((size_t)na * perm_tile, -1);
        for (int a = 0; a < na; ++a)
          for (int m = ptr[a]; m < ptr[a + 1]; ++m) tperm[(size_t)a * perm_tile + (m - ptr[a])] = mem[m];
        CHK(upload(&L.tile_perm, tperm.data(), tperm.size(), p->stream));
        L.perm_tile = perm_tile;
      }
      HIPC(hipMalloc(&L.tbuf, tb * cur));
      // pattern of A P: the aggregates the block columns of every row fall into
      std::vector<int32_t> arow(cur + 1, 0), acol;
      acol.reserve(colidx.size());
      for (int i = 0; i < cur; ++i) {
        const size_t first = acol.size();
        for (int t = rowptr[i]; t < rowptr[i + 1]; ++t) acol.push_back(lab[colidx[t]]);
        std::sort(acol.begin() + first, acol.end());
        acol.erase(std::unique(acol.begin() + first, acol.end()), acol.end());
        arow[i + 1] = (int32_t)acol.size();
      }
      CHK(upload_bsr(L.AP, cur, na, (int)acol.size(), b, arow.data(), acol.data(), nullptr, p->stream));
      HIPC(hipMalloc(&L.res1, tb * cur));
      HIPC(hipMalloc(&L.Pb, sizeof(double) * (size_t)cur * bb));
      HIPC(hipMalloc(&L.x1, tb * cur));
      HIPC(hipMalloc(&L.x, tb * cur));
      // pattern of the dense level's operator: the aggregates of the block columns of every member's row
      std::vector<int32_t> crow(na + 1, 0), ccol;
      std::vector<int32_t> mark(na, -1);
      for (int a = 0; a < na; ++a) {
        const size_t first = ccol.size();
        for (int m = ptr[a]; m < ptr[a + 1]; ++m) {
          const int i = mem[m];
          for (int t = rowptr[i]; t < rowptr[i + 1]; ++t) {
            const int c = lab[colidx[t]];
            if (mark[c] != a) {
              mark[c] = a;
              ccol.push_back(c);
            }
          }
        }
        std::sort(ccol.begin() + first, ccol.end());
        crow[a + 1] = (int32_t)ccol.size();
      }
      HIPC(hipStreamSynchronize(p->stream));  // the host vectors go out of scope
      rowptr.swap(crow);
      colidx.swap(ccol);
      cur = na;
      continue;
    }
    if (L.k) {
      if (L.k < 2 || ml_tile(b, L.split) % L.k)
        return fail(DPGO_ERR_INVALID, "multilevel: aggregate size must divide the workgroup tile of its level (" +
                                          std::to_string(ml_tile(b, L.split)) + " nodes)");
    }
    if (l > 0) {
      CHK(upload_bsr(L.A, cur, cur, (int)colidx.size(), b, rowptr.data(), colidx.data(), nullptr, p->stream));
      std::vector<int32_t> srow(colidx.size());
      for (int i = 0; i < cur; ++i)
        for (int t = rowptr[i]; t < rowptr[i + 1]; ++t) srow[t] = i;
      CHK(upload(&L.slot_row, srow.data(), srow.size(), p->stream));
      HIPC(hipStreamSynchronize(p->stream));  // srow goes out of scope at the end of this block
      HIPC(hipMalloc(&L.r, tb * cur));
      if (!L.k) HIPC(hipMalloc(&L.x, tb * cur));  // dense level: its solution, read by the level above
    }
    if (l == 0 && L.k) {  // pattern of A P: the aggregates the block columns of every row fall into
      const int k = L.k;
      std::vector<int32_t> arow(cur + 1, 0), acol;
      acol.reserve(colidx.size());
      for (int i = 0; i < cur; ++i) {
        const size_t first = acol.size();
        for (int t = rowptr[i]; t < rowptr[i + 1]; ++t) acol.push_back(colidx[t] / k);
        std::sort(acol.begin() + first, acol.end());
        acol.erase(std::unique(acol.begin() + first, acol.end()), acol.end());
        arow[i + 1] = (int32_t)acol.size();
      }
      CHK(upload_bsr(L.AP, cur, (cur + k - 1) / k, (int)acol.size(), b, arow.data(), acol.data(), nullptr, p->stream));
      HIPC(hipMalloc(&L.res1, tb * cur));
    }
    if (L.k) {
      if (l > 0) HIPC(hipMalloc(&L.dinv, sizeof(double) * (size_t)cur * bb));
      HIPC(hipMalloc(&L.Pb, sizeof(double) * (size_t)cur * bb));
      HIPC(hipMalloc(&L.x1, tb * cur));
      HIPC(hipMalloc(&L.x, tb * cur));
      // pattern of the next level: block columns j / k of the rows of every aggregate
      const int k = L.k, nc = (cur + k - 1) / k;
      std::vector<int32_t> crow(nc + 1, 0), ccol;
      std::vector<int32_t> mark(nc, -1);
      for (int a = 0; a < nc; ++a) {
        const size_t first = ccol.size();
        for (int i = a * k; i < std::min(cur, a * k + k); ++i)
          for (int t = rowptr[i]; t < rowptr[i + 1]; ++t) {
            const int c = colidx[t] / k;
            if (mark[c] != a) {
              mark[c] = a;
              ccol.push_back(c);
            }
          }
        std::sort(ccol.begin() + first, ccol.end());
        crow[a + 1] = (int32_t)ccol.size();
      }
      rowptr.swap(crow);
      colidx.swap(ccol);
      cur = nc;
    }
  }
  const int N = cur * b;
  if (N > 16384) return fail(DPGO_ERR_INVALID, "multilevel: dense coarsest operator too large (" + std::to_string(N) +
                                                   " unknowns): use more levels / larger aggregates");
  p->ml_lda = ((N + kNB - 1) / kNB) * kNB;
  // + 8 rows: the apply kernel reads (and discards) the rows of a ghost node behind a ragged last node group
  HIPC(hipMalloc(&p->ml_dense, sizeof(double) * (size_t)p->ml_lda * (p->ml_lda + 8)));
  HIPC(hipMalloc(&p->ml_dense32, sizeof(float) * (size_t)p->ml_lda * (p->ml_lda + 8)));
  if (p->ml.size() == 2) {  // two levels: the packed lower triangle and the bookkeeping of k_dense_sym_apply
    const int nT = p->ml_lda / kNB;
    int chunk = kDenseChunk;
    if (const char* e = std::getenv("DPGO_DENSE_CHUNK")) chunk = std::max(1, std::atoi(e));  // tuning knob
    std::vector<DenseChunk> chunks;
    std::vector<int> first(nT + 1, 0);
    for (int I = 0; I < nT; ++I) {
      first[I] = (int)chunks.size();
      for (int J0 = 0; J0 <= I; J0 += chunk) chunks.push_back(DenseChunk{I, J0, std::min(chunk, I + 1 - J0), 0});
    }
    first[nT] = (int)chunks.size();
    p->ml_nchunks = (int)chunks.size();
    CHK(upload(&p->ml_chunks, chunks.data(), chunks.size(), p->stream));
    CHK(upload(&p->ml_chunk_first, first.data(), first.size(), p->stream));
    HIPC(hipMalloc(&p->ml_packed, sizeof(double) * (size_t)nT * (nT + 1) / 2 * kNB * kNB));
    HIPC(hipMalloc(&p->ml_pd, sizeof(double) * (size_t)p->ml_nchunks * kNB * p->r));
    HIPC(hipMalloc(&p->ml_pt, sizeof(double) * (size_t)nT * p->ml_lda * p->r));
    HIPC(hipStreamSynchronize(p->stream));  // the host vectors go out of scope
  }
  HIPC(hipMalloc(&p->ml_W, sizeof(double) * (size_t)p->ml_lda * kNB));
  HIPC(hipMalloc(&p->ml_Rx, sizeof(double) * (size_t)p->ml_lda * kNB));
  HIPC(hipStreamSynchronize(p->stream));
  p->ml_symbolic = true;
  return DPGO_OK;
}

int flat_grid(size_t items) {
  size_t g = (items + kBlock - 1) / kBlock;
  if (g < 1) g = 1;
  return g < (size_t)kMaxGrid ? (int)g : kMaxGrid;
}

bool gj_use_mfma() {
  if (const char* e = std::getenv("DPGO_GJ_MFMA")) return std::atoi(e) != 0;
  return true;
}

// In-place inverse of the dense SPD lda x lda array M (lda a multiple of 64); W, Rx: lda x 64 panels.
int dense_spd_inverse(hipStream_t s, double* M, int lda, double* W, double* Rx, bool mfma) {
  const int nt = lda / kNB;
  for (int kb = 0; kb < nt; ++kb) {
    hipLaunchKernelGGL(k_sweep_panel, dim3(nt), dim3(kBlock), 0, s, M, lda, kb, W, Rx);
    if (mfma)
      hipLaunchKernelGGL(k_sweep_update<true>, dim3(nt, nt), dim3(kBlock), 0, s, M, lda, kb, W, Rx);
    else
      hipLaunchKernelGGL(k_sweep_update<false>, dim3(nt, nt), dim3(kBlock), 0, s, M, lda, kb, W, Rx);
  }
  hipLaunchKernelGGL(k_sweep_finish, dim3(nt, nt), dim3(kBlock), 0, s, M, lda);
  HIPC(hipGetLastError());
  return DPGO_OK;
}

template <int D>
int ml_numeric_setup_d(dpgo_problem_s* p) {
  const int nl = (int)p->ml.size();
  long long stride = 1;
  for (int l = 0; l + 1 < nl; ++l) {
    auto& L = p->ml[l];
    auto& C = p->ml[l + 1];
    const long long span = stride * L.k;
    // (wave-parallel forms of the two setup kernels that walked an aggregate's members with ONE thread; DPGO_ML_SETUP_SERIAL=1
    // restores them)
    static const bool serial = [] { const char* e = std::getenv("DPGO_ML_SETUP_SERIAL"); return e && std::atoi(e) != 0; }();
    auto wave_grid = [](int items) { return std::max(1, std::min(kMaxGrid, (items + kWaves - 1) / kWaves)); };
    if (L.graph && !serial)
      hipLaunchKernelGGL(k_ml_build_P_tree_wave<D>, dim3(wave_grid(C.n)), dim3(kBlock), 0, p->stream, p->Q.dev(), L.agg_ptr,
                         L.agg_mem, L.parent, L.pslot, L.mem_pos, L.Pb, C.n);
    else if (L.graph)
      hipLaunchKernelGGL(k_ml_build_P_tree<D>, dim3(flat_grid(C.n)), dim3(kBlock), 0, p->stream, p->Q.dev(), L.agg_ptr,
                         L.agg_mem, L.parent, L.pslot, L.Pb, C.n);
    else
      hipLaunchKernelGGL(k_ml_build_P<D>, dim3(flat_grid(C.n)), dim3(kBlock), 0, p->stream, p->Q.dev(), p->n, (int)stride,
                         (int)span, L.Pb, C.n);
    const BsrDev A = (l == 0) ? p->Q.dev() : L.A.dev();
    const bool have_ap = l == 0 && L.AP.vals;
    if (have_ap)  // A P first: the Galerkin operator of a two-level hierarchy is its restriction
      hipLaunchKernelGGL(k_ml_build_AP<D>, dim3(flat_grid(L.n)), dim3(kBlock), 0, p->stream, p->Q.dev(), p->ml_shift, L.Pb,
                         L.agg(), L.n, L.AP.dev(), L.AP.vals);
    if (have_ap && !serial)
      hipLaunchKernelGGL(k_ml_galerkin_ap<D>, dim3(wave_grid(C.A.nnzb)), dim3(kBlock), 0, p->stream, L.AP.dev(), L.Pb,
                         L.agg(), L.agg_ptr, L.agg_mem, L.n, C.slot_row, C.A.colidx, C.A.vals, C.A.nnzb);
    else
      hipLaunchKernelGGL(k_ml_galerkin<D>, dim3(flat_grid(C.A.nnzb)), dim3(kBlock), 0, p->stream, A,
                         (l == 0) ? p->ml_shift : 0.0, L.Pb, L.agg(), L.agg_ptr, L.agg_mem, L.n, C.slot_row, C.A.colidx,
                         C.A.vals, C.A.nnzb);
    if (C.k)  // smoother of the next level (level 0 uses the handle's block-Jacobi factors)
      hipLaunchKernelGGL(k_build_dinv<D>, dim3(flat_grid(C.n)), dim3(kBlock), 0, p->stream, C.A.dev(), 0.0, C.dinv, C.n);
    stride = span;
  }
  HIPC(hipGetLastError());
  auto& Lc = p->ml.back();
  const int lda = p->ml_lda, N = Lc.n * p->b;
  HIPC(hipMemsetAsync(p->ml_dense, 0, sizeof(double) * (size_t)lda * (lda + 8), p->stream));
  hipLaunchKernelGGL(k_dense_pad_identity, dim3(1), dim3(kBlock), 0, p->stream, p->ml_dense, lda, N);
  hipLaunchKernelGGL(k_ml_dense_assemble<D>, dim3(flat_grid(Lc.A.nnzb)), dim3(kBlock), 0, p->stream, Lc.A.dev(),
                     Lc.slot_row, p->ml_dense, lda, Lc.A.nnzb);
  HIPC(hipGetLastError());
  CHK(dense_spd_inverse(p->stream, p->ml_dense, lda, p->ml_W, p->ml_Rx, gj_use_mfma()));
  if (p->ml_packed) {
    const int nT = lda / kNB;
    hipLaunchKernelGGL(k_dense_pack_lower, dim3(nT, nT), dim3(kBlock), 0, p->stream, p->ml_dense, lda, p->ml_packed);
    HIPC(hipGetLastError());
  }
  if (p->ml_coarse_bits == 32) {
    const size_t total = (size_t)lda * (lda + 8);
    hipLaunchKernelGGL(k_dense_round_f32, dim3(flat_grid(total)), dim3(kBlock), 0, p->stream, p->ml_dense, p->ml_dense32,
                       total);
    HIPC(hipGetLastError());
  }
  return DPGO_OK;
}

// Numeric setup for the CURRENT values of Q (device only; redone after every re-weighting).
int ml_numeric_setup(dpgo_problem_s* p) {
  if (!p->ml_symbolic) return fail(DPGO_ERR_STATE, "multilevel: symbolic setup missing");
  CHK(build_dinv(p, p->ml_shift));
  if (p->d == 2)
    CHK(ml_numeric_setup_d<2>(p));
  else
    CHK(ml_numeric_setup_d<3>(p));
  p->ml_ready = true;
  return DPGO_OK;
}

// The hierarchy's shape in the form ml_symbolic_setup takes it.
std::vector<int> ml_current_ks(const dpgo_problem_s* p) {
  std::vector<int> ks;
  for (size_t l = 0; l + 1 < p->ml.size(); ++l) ks.push_back(p->ml[l].graph ? -p->ml[l].k : p->ml[l].k);
  if (p->ml.size() == 2 && p->ml[0].graph && p->ml[0].merge_cap) ks.push_back(-p->ml[0].merge_cap);
  return ks;
}

// Layout of the additive preconditioner inside the one-launch solve (k_rtr_persist<..., ADD>): ONE aggregate per workgroup,
// at most kPersistMax = 256 of them.  Host only, once per block pattern:
//   1. graph aggregates of at most one 4-lane-group tile (16 poses in 3-D), plain greedy growth, while there are <= 256
//      (blocks up to ~3 500 poses: the lowest-latency layout);
//   2. else one pose per (d+1) lanes (tile = 64 poses in 3-D): graph aggregates grown to S poses, fragments merged up to
//      min(tile, 3 S / 2), with the smallest S (from ceil(n / 230) in steps of an eighth) that leaves <= 256 aggregates
//      (12 500-pose slab: S = 55, 230 aggregates; 6 250-pose grid: S = 28, 221) -- blocks up to ~14 000 poses;
//   3. without graph aggregates (DPGO_ML_GRAPH=0): index runs of one tile.
const dpgo_problem_s::AddPlan& additive_plan(dpgo_problem_s* p) {
  if (p->add_plan_known) return p->add_plan;
  p->add_plan = dpgo_problem_s::AddPlan();
  p->add_plan_known = true;
  p->add_agg = dpgo_problem_s::AggCache();
  if (p->split != 4 || (int)p->h_rowptr.size() != p->n + 1) return p->add_plan;
  const int P4 = ml_tile(p->b, 4), P1 = ml_tile(p->b, 1), n = p->n;
  static const bool graph_ok = [] { const char* e = std::getenv("DPGO_ML_GRAPH"); return !e || std::atoi(e) != 0; }();
  if (graph_ok) {
    auto& A = p->add_agg;
    auto &lab = A.lab, &ptr = A.ptr, &mem = A.mem, &parent = A.parent, &pslot = A.pslot;
    if ((long long)n <= (long long)kPersistMax * P4) {
      const int na = ml_graph_aggregates(p->h_rowptr, p->h_colidx, n, P4, lab, ptr, mem, parent, pslot);
      if (na <= kPersistMax) {
        p->add_plan = dpgo_problem_s::AddPlan{4, P4, P4, 0, na, true};
        A.S = P4, A.cap = 0;
        return p->add_plan;
      }
    }
    if ((long long)n <= (long long)kPersistMax * P1) {
      // A handle that is solved next to other handles of the device (dpgo_optimize_device_many: persist_share > 1 when the
      // plan is first asked for) aims at HALF the chip -- every aggregate is a workgroup that owns a CU for the whole solve,
      // so two such solves run side by side instead of taking turns; the product count is a weak function of the aggregate
      // size, the time of an iteration is not a function of how full the tiles are.
      const int want = (p->persist_share > 1 && (long long)n * 10 <= (long long)(kPersistMax / 2) * P1 * 8) ? kPersistMax / 2 : kPersistMax;
      for (int S = std::max(8, (n + (want * 9) / 10 - 1) / ((want * 9) / 10)); S <= P1; S += std::max(2, S / 8)) {
        const int cap = std::min(P1, S + S / 2);
        ml_graph_aggregates(p->h_rowptr, p->h_colidx, n, S, lab, ptr, mem, parent, pslot);
        const int na = ml_merge_small_aggregates(p->h_rowptr, p->h_colidx, n, S, cap, lab, ptr, mem, parent, pslot);
        if (na <= want) {
          p->add_plan = dpgo_problem_s::AddPlan{1, P1, S, cap, na, true};
          A.S = S, A.cap = cap;
          return p->add_plan;
        }
      }
      if (want < kPersistMax) {  // (no growth size reaches half the chip: the whole-chip plan)
        for (int S = std::max(8, (n + 229) / 230); S <= P1; S += std::max(2, S / 8)) {
          const int cap = std::min(P1, S + S / 2);
          ml_graph_aggregates(p->h_rowptr, p->h_colidx, n, S, lab, ptr, mem, parent, pslot);
          const int na = ml_merge_small_aggregates(p->h_rowptr, p->h_colidx, n, S, cap, lab, ptr, mem, parent, pslot);
          if (na <= kPersistMax) {
            p->add_plan = dpgo_problem_s::AddPlan{1, P1, S, cap, na, true};
            A.S = S, A.cap = cap;
            return p->add_plan;
          }
        }
      }
    }
    p->add_agg = dpgo_problem_s::AggCache();
  }
  if ((n + P4 - 1) / P4 <= kPersistMax)
    p->add_plan = dpgo_problem_s::AddPlan{4, P4, P4, 0, (n + P4 - 1) / P4, false};
  else if ((n + P1 - 1) / P1 <= kPersistMax)
    p->add_plan = dpgo_problem_s::AddPlan{1, P1, P1, 0, (n + P1 - 1) / P1, false};
  return p->add_plan;
}

// lane groups per pose of the additive layout the CURRENT two-level hierarchy fits (0: none)
int additive_split_of(const dpgo_problem_s* p) {
  if (!p->ml_symbolic || p->ml.size() != 2 || p->split != 4 || p->ml[1].n > kPersistMax) return 0;
  const auto& L = p->ml[0];
  const int P4 = ml_tile(p->b, 4), P1 = ml_tile(p->b, 1);
  const int tile = L.graph ? (L.tile_perm ? L.perm_tile : 0) : L.k;
  return tile == P4 ? 4 : (tile == P1 ? 1 : 0);
}

// Make the hierarchy match the handle's Q (lazily, like the reference's constructPreconditioner inside the first
// PreConditioner call, src/PoseGraph.cpp:582-586).
int ml_ensure(dpgo_problem_s* p, double shift, bool additive = false) {
  // the additive preconditioner needs ONE aggregate per workgroup tile of its persistent layout (two levels); a hierarchy
  // the caller set up explicitly is kept if it has that shape, the default one is replaced by the handle's plan
  // (additive_plan) and put back when the V-cycle is asked for again
  if (additive && !additive_split_of(p)) {
    const auto& plan = additive_plan(p);
    if (!plan.split) return fail(DPGO_ERR_UNSUPPORTED, "additive preconditioner: the block does not fit 256 aggregates of one workgroup tile");
    std::vector<int> ks{plan.graph ? -plan.S : plan.S};
    if (plan.graph && plan.cap) ks.push_back(-plan.cap);
    CHK(ml_symbolic_setup(p, ks, plan.graph ? plan.tile : 0));
    if (!additive_split_of(p)) return fail(DPGO_ERR_STATE, "additive preconditioner: hierarchy does not match its plan");
    p->ml_additive_layout = true;
    p->ml_user_ks = false;
  } else if (!additive && p->ml_additive_layout && !p->ml_user_ks) {
    CHK(ml_symbolic_setup(p, ml_default_ks(p->n, p->b, p->split)));
    p->ml_additive_layout = false;
  }
  // level 0 smooths with the handle's shared block-Jacobi factors: a block-Jacobi solve with another shift in between
  // has overwritten them, so they are re-derived for THIS shift even when the hierarchy itself is current (no-op otherwise)
  if (p->ml_ready && p->ml_shift == shift) return build_dinv(p, shift);
  if (!p->ml_symbolic) CHK(ml_symbolic_setup(p, ml_default_ks(p->n, p->b, p->split)));
  p->ml_shift = shift;
  return ml_numeric_setup(p);
}

// Dense level + prolongation.  Large coarsest levels: two nodes per workgroup (halves the right-hand-side loads per
// matrix byte); balanced rounds: every workgroup takes the same number of node groups (a ragged last round would leave
// most of the chip idle while the dense inverse streams).
int persist_capacity(int device);  // (two resident slots per CU; below)
int launch_coarse_prolong(dpgo_problem_s* p, const dpgo_problem_s::MlLevel& L, const dpgo_problem_s::MlLevel& C,
                          const DevState* gate, double* xc_out = nullptr) {
  const bool f32 = p->ml_coarse_bits == 32;
  // nodes per workgroup (the right-hand side is read once per workgroup): 732 nodes: 1 -> 2: 19.7 -> 17.1 us, 4: 18.1;
  // three (fp64 storage) where that brings the level down to one workgroup per CU in one round: 546 nodes: 2 -> 3:
  // 273 -> 182 workgroups, 13.2 -> 11.6 us, the 100k bench step 4.45 -> 4.33 ms
  const int cus = persist_capacity(p->device) / 2;
  int nodes = C.n >= 512 ? 2 : 1;
  if (nodes == 2 && !f32 && (C.n + 1) / 2 > cus && (C.n + 2) / 3 <= cus) nodes = 3;
  if (const char* e = std::getenv("DPGO_COARSE_NODES")) {  // tuning knob
    const int v = std::atoi(e);
    nodes = (v == 4 || v == 2 || (v == 3 && !f32)) ? v : 1;
  }
  const int groups = (C.n + nodes - 1) / nodes;
  int cap = kMaxGrid;
  if (const char* e = std::getenv("DPGO_COARSE_GRID")) cap = std::max(1, std::atoi(e));  // tuning knob
  const int rounds = (groups + cap - 1) / cap;
  const int gc = std::max(1, (groups + rounds - 1) / rounds);
  // non-temporal loads of the inverse whenever the loop's working set does not fit the Infinity Cache (kernel comment)
  int hint = p->beyond_cache();
  if (const char* e = std::getenv("DPGO_COARSE_NT")) hint = std::atoi(e) != 0;  // tuning knob
#define COARSE_LAUNCH(NODES, MT, MPTR)                                                                               \
  hipLaunchKernelGGL((k_ml_coarse_prolong<D, R, NODES, MT>), dim3(gc), dim3(kBlock), 0, p->stream, MPTR, p->ml_lda,   \
                     reinterpret_cast<const MT*>(C.r), L.x1, L.Pb, L.k, L.x, gate, L.n, C.n, xc_out, hint)
  DISPATCH(p->d, p->r, {
    if (nodes == 4 && f32)
      COARSE_LAUNCH(4, float, p->ml_dense32);
    else if (nodes == 4)
      COARSE_LAUNCH(4, double, p->ml_dense);
    else if (nodes == 3)
      COARSE_LAUNCH(3, double, p->ml_dense);
    else if (nodes == 2 && f32)
      COARSE_LAUNCH(2, float, p->ml_dense32);
    else if (nodes == 2)
      COARSE_LAUNCH(2, double, p->ml_dense);
    else if (f32)
      COARSE_LAUNCH(1, float, p->ml_dense32);
    else
      COARSE_LAUNCH(1, double, p->ml_dense);
  });
#undef COARSE_LAUNCH
  HIPC(hipGetLastError());
  return DPGO_OK;
}

// Dense level from the packed lower triangle: xc = A_c^-1 rc into C.x (two launches: partial products, fixed-order sums)
int launch_dense_sym(dpgo_problem_s* p, const dpgo_problem_s::MlLevel& C, const DevState* gate) {
  const int N = C.n * p->b, nT = p->ml_lda / kNB;
  switch (p->r) {
#define DENSE_SYM_CASE(RR)                                                                                              \
  case RR:                                                                                                              \
    hipLaunchKernelGGL((k_dense_sym_apply<RR>), dim3(p->ml_nchunks), dim3(kBlock), 0, p->stream, p->ml_packed,          \
                       p->ml_chunks, C.r, N, p->ml_lda, p->ml_pd, p->ml_pt, gate);                                       \
    hipLaunchKernelGGL((k_dense_sym_finish<RR>), dim3(nT, 4), dim3(kBlock), 0, p->stream, p->ml_pd, p->ml_pt,            \
                       p->ml_chunk_first, nT, N, p->ml_lda, C.x, gate);                                                  \
    break;
    DENSE_SYM_CASE(2)
    DENSE_SYM_CASE(3)
    DENSE_SYM_CASE(4)
    DENSE_SYM_CASE(5)
    DENSE_SYM_CASE(6)
#undef DENSE_SYM_CASE
    default:
      return fail(DPGO_ERR_UNSUPPORTED, "unsupported r");
  }
  HIPC(hipGetLastError());
  return DPGO_OK;
}

// Level-0 restriction of the cycle: rc = P^T (r - A x1) into ml[1].r (+ the residual itself for k_ml_post_ap).  Graph
// aggregates: the restriction kernel adds P_i^T res_i up over every run of same-aggregate poses inside a wave's chunk and
// writes one partial sum per run, k_ml_agg_sum adds an aggregate's partial sums up.
int launch_ml_restrict0(dpgo_problem_s* p, const double* r, const DevState* gate, int g0, bool stop_check = false) {
  auto& L = p->ml[0];
  auto& C = p->ml[1];
  // inside the tCG loop (not after its first update): tCG's residual test one kernel early (TcgStopCheck, multilevel.h);
  // the <r,r> partial sums are the ones k_tcg_update wrote, one per workgroup of ITS grid
  TcgStopCheck stop;
  if (stop_check && gate) {
    stop.state = const_cast<DevState*>(gate);
    stop.pin = p->pB();
    stop.nb = p->grid();
    stop.hflag = p->hflag;
    stop.gen = p->gen;
  }
  float* rc32 = (C.k == 0 && p->ml_coarse_bits == 32) ? reinterpret_cast<float*>(C.r) : (float*)nullptr;
  double* res_out = p->ml_use_ap() ? L.res1 : nullptr;
  const double* dnext = C.k ? C.dinv : (const double*)nullptr;
  if (p->tcg_sym) {  // level 0 reads Q: the symmetric copy when the tCG-step kernel does
    DISPATCH(p->d, p->r, hipLaunchKernelGGL((k_ml_restrict<D, R, 1, BsrSymDev>), dim3(g0), dim3(kBlock), 0, p->stream,
                                            p->sym.dev(), L.x1, r, L.Pb, p->ml_shift, L.k, C.r, rc32, dnext, p->ml_omega,
                                            C.x1, gate, L.n, res_out, L.tbuf, L.seg_info, stop));
  } else {
    DISPATCH(p->d, p->r, LAUNCH_SPLIT(p, k_ml_restrict, g0, p->Q.dev(), L.x1, r, L.Pb, p->ml_shift, L.k, C.r, rc32, dnext,
                                      p->ml_omega, C.x1, gate, L.n, res_out, L.tbuf, L.seg_info, stop));
  }
  if (L.graph)
    DISPATCH(p->d, p->r, hipLaunchKernelGGL((k_ml_agg_sum<D, R>), dim3(std::min(C.n, kMaxGrid)), dim3(kBlock), 0, p->stream,
                                            L.tbuf, L.seg_ptr, C.n, C.r, rc32, gate));
  HIPC(hipGetLastError());
  return DPGO_OK;
}

// Level-0 post-smoothing of a two-level hierarchy through A P (k_ml_post_ap).
int launch_ml_post_ap(dpgo_problem_s* p, const double* Xdev, const double* r, double* z, double* pout, const DevState* gate) {
  auto& L0 = p->ml[0];
  DISPATCH(p->d, p->r, LAUNCH_SPLIT(p, k_ml_post_ap, p->grid_post(), L0.AP.dev(), Xdev, r, L0.res1, p->ml[1].x, L0.Pb,
                                    L0.agg(), p->dinv, p->ml_omega, z, pout, gate, p->n));
  HIPC(hipGetLastError());
  return DPGO_OK;
}

// The launches of one cycle after the pre-smoothing step of level 0 (x1 = w Dinv r is in ml[0].x1):
// z = proj_X(M^-1 r); partial sums <r,r>, <z,r> into `pout` (may be NULL).  `gate`: state record for early exit.
int launch_ml_tail(dpgo_problem_s* p, const double* Xdev, const double* r, double* z, double* pout,
                   const DevState* gate, bool stop_check = false) {
  const int nl = (int)p->ml.size();
  // the dense level reads its right-hand side in the precision its inverse is stored in (same buffer)
  auto rc32_of = [&](const dpgo_problem_s::MlLevel& C) {
    return (C.k == 0 && p->ml_coarse_bits == 32) ? reinterpret_cast<float*>(C.r) : (float*)nullptr;
  };
  auto A_of = [&](int l) { return l == 0 ? p->Q.dev() : p->ml[l].A.dev(); };
  auto r_of = [&](int l) { return l == 0 ? r : (const double*)p->ml[l].r; };
  int g0 = p->grid_restrict();  // grid of the level-0 launches: restriction first, post-smoothing later
  auto grid_of = [&](const dpgo_problem_s::MlLevel& L) {
    if (&L == &p->ml[0]) return g0;
    const int P = ml_tile(p->b, L.split);
    return std::max(1, std::min(kMaxGrid, (L.n + P - 1) / P));
  };
#define ML_SPLIT_LAUNCH(L, KERNEL, ...)                                                                  \
  do {                                                                                                   \
    const int g_ = grid_of(L);                                                                           \
    if ((L).split == 4)                                                                                  \
      hipLaunchKernelGGL((KERNEL<D, R, 4>), dim3(g_), dim3(kBlock), 0, p->stream, __VA_ARGS__);          \
    else if ((L).split == 2)                                                                             \
      hipLaunchKernelGGL((KERNEL<D, R, 2>), dim3(g_), dim3(kBlock), 0, p->stream, __VA_ARGS__);          \
    else                                                                                                 \
      hipLaunchKernelGGL((KERNEL<D, R, 1>), dim3(g_), dim3(kBlock), 0, p->stream, __VA_ARGS__);          \
  } while (0)
  const bool ap = p->ml_use_ap();  // two levels: the residual after pre-smoothing is kept, the dense level hands over xc
  CHK(launch_ml_restrict0(p, r, gate, g0, stop_check));
  for (int l = 1; l + 1 < nl; ++l) {  // down
    auto& L = p->ml[l];
    auto& C = p->ml[l + 1];
    DISPATCH(p->d, p->r, ML_SPLIT_LAUNCH(L, k_ml_restrict, A_of(l), L.x1, r_of(l), L.Pb, 0.0, L.k, C.r, rc32_of(C),
                                         C.k ? C.dinv : (const double*)nullptr, p->ml_omega, C.x1, gate, L.n,
                                         (double*)nullptr, (double*)nullptr, (const int32_t*)nullptr));
  }
  {  // dense level (+ prolongation unless the level above does it itself)
    auto& L = p->ml[nl - 2];
    auto& C = p->ml[nl - 1];
    if (p->ml_use_dense_sym())
      CHK(launch_dense_sym(p, C, gate));
    else
      CHK(launch_coarse_prolong(p, L, C, gate, ap ? C.x : nullptr));
  }
  g0 = p->grid_post();
  if (ap) return launch_ml_post_ap(p, Xdev, r, z, pout, gate);
  for (int l = nl - 2; l >= 1; --l) {  // up
    auto& L = p->ml[l];
    auto& F = p->ml[l - 1];
    DISPATCH(p->d, p->r, ML_SPLIT_LAUNCH(L, k_ml_post_mid, L.A.dev(), L.x, L.r, L.dinv, p->ml_omega, F.x1, F.Pb, F.k, F.x,
                                         F.n, gate, L.n));
  }
  if (p->tcg_sym) {
    DISPATCH(p->d, p->r, hipLaunchKernelGGL((k_ml_post<D, R, 1, BsrSymDev>), dim3(g0), dim3(kBlock), 0, p->stream,
                                            p->sym.dev(), Xdev, p->ml[0].x, r, p->dinv, p->ml_omega, p->ml_shift, z, pout,
                                            gate, p->n));
  } else {
    DISPATCH(p->d, p->r, ML_SPLIT_LAUNCH(p->ml[0], k_ml_post, p->Q.dev(), Xdev, p->ml[0].x, r, p->dinv, p->ml_omega,
                                         p->ml_shift, z, pout, gate, p->n));
  }
#undef ML_SPLIT_LAUNCH
  HIPC(hipGetLastError());
  return DPGO_OK;
}

// Stand-alone application z = proj_X(M^-1 v) (QuadraticProblem::PreConditioner outside the tCG loop).
int launch_ml_apply(dpgo_problem_s* p, const double* Xdev, const double* v, double* z) {
  DISPATCH(p->d, p->r, hipLaunchKernelGGL((k_ml_presmooth<D, R>), dim3(p->grid()), dim3(kBlock), 0, p->stream, v, p->dinv,
                                          p->ml_omega, p->ml[0].x1, (const DevState*)nullptr, p->n));
  HIPC(hipGetLastError());
  return launch_ml_tail(p, Xdev, v, z, nullptr, nullptr);
}

// ---------------------------------------------------------------------------------------------------------
// One-launch solve (kernels/persist.h, k_rtr_persist): one launch runs QuadraticOptimizer::optimize whole.
//
// Residency.  Every workgroup of such a launch waits for all the others, so all of them must be resident at once.  The
// grid is therefore sized against a per-device count of resident slots shared by all handles of the process (one slot =
// one 256-thread workgroup; capacity = two per CU: every variant of the kernel is compiled for two workgroups per CU
// -- registers, LDS --, whatever else runs), reserved for the duration of the solve.  A handle that cannot reserve runs the
// multi-launch scheme.  Other processes are not covered: every in-kernel spin is bounded, a time-out poisons the state
// record (rtr_stop = kPersistPoison) and leaves the caller's iterate untouched; run_optimize then runs the solve with the
// multi-launch scheme.
constexpr int kMaxDevices = 64;
std::atomic<int> g_warnings{0};  // warnings printed to stderr so far (dpgo_warning_count)
std::atomic<int> g_persist_used[kMaxDevices];
std::atomic<int> g_persist_cap[kMaxDevices];  // 0 = not yet queried

int persist_capacity(int device) {
  int cap = g_persist_cap[device % kMaxDevices].load();
  if (cap == 0) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 1;
    cap = 2 * cus;
    g_persist_cap[device % kMaxDevices].store(cap);
  }
  return cap;
}
bool persist_reserve(dpgo_problem_s* p, int slots, int limit) {
  auto& used = g_persist_used[p->device % kMaxDevices];
  int cur = used.load();
  while (cur + slots <= limit)
    if (used.compare_exchange_weak(cur, cur + slots)) {
      p->persist_reserved = slots;
      return true;
    }
  return false;
}
void persist_release(dpgo_problem_s* p) {
  if (p->persist_reserved > 0) g_persist_used[p->device % kMaxDevices].fetch_sub(p->persist_reserved);
  p->persist_reserved = 0;
}

// Geometry of a launch: lane groups per pose (SPLIT), tiles per workgroup (MT), workgroups.  The smallest-latency layout
// whose grid fits the handle's share of the resident slots: 4 lane groups per pose (short gather chains) while the tiles
// fit, otherwise one pose per (d+1) lanes, with up to 2 tiles per workgroup.
struct PersistGeo {
  int split = 0, mt = 0, wgs = 0, slots = 0;
};
bool additive_available(dpgo_problem_s* p) {
  return p->persist && !p->persist_failed_once && additive_plan(p).split != 0;
}
// `free_slots`: what may be reserved.  Alone on the device (share = 1): the lowest-latency layout that fits (4 lane groups
// per pose while the tiles fit, then one pose per (d+1) lanes).  Sharing the device with `share` concurrently solved
// agents: the lowest-latency layout of which `share` copies fit side by side; if there is none, the most compact one
// (the solves then take turns).
PersistGeo persist_geometry(const dpgo_problem_s* p, int free_slots, int share = 1, bool additive = false) {
  if (additive) {  // fixed by the hierarchy (after ml_ensure); one workgroup per CU (the rows of the coarse inverse live in its LDS)
    const int sp = additive_split_of(p);
    if (!sp) return PersistGeo();
    PersistGeo g{sp, 1, p->ml[1].n, 0};
    g.slots = g.wgs * persist_slots_per_wg(sp, 1, true);
    if (g.wgs > kPersistMax || g.slots > free_slots) return PersistGeo();
    return g;
  }
  static const int env_split = [] { const char* e = std::getenv("DPGO_PERSIST_SPLIT"); return e ? std::atoi(e) : 0; }();
  static const int env_mt = [] { const char* e = std::getenv("DPGO_PERSIST_MT"); return e ? std::atoi(e) : 0; }();
  const int cand[4][2] = {{4, 1}, {4, 2}, {1, 1}, {1, 2}};
  PersistGeo compact;
  for (auto& c : cand) {
    if (env_split && c[0] != env_split) continue;
    if (env_mt && c[1] != env_mt) continue;
    const int P = (64 / (p->b * c[0])) * kWaves;
    const int tiles = std::max(1, (p->n + P - 1) / P);
    const int wgs = (tiles + c[1] - 1) / c[1];
    const int slots = wgs * persist_slots_per_wg(c[0], c[1]);
    if (wgs > kPersistMax || slots > free_slots) continue;
    const PersistGeo g{c[0], c[1], wgs, slots};
    if ((long long)slots * std::max(1, share) <= free_slots) return g;  // everybody fits at once
    if (compact.wgs == 0 || slots < compact.slots) compact = g;
  }
  return compact;
}

// Enqueues the persistent launch of a WHOLE solve (k_rtr_persist; no host wait).  *used = false: not launched (no geometry
// / no free slots) -- the caller runs the multi-launch scheme.
int launch_rtr_persistent(dpgo_problem_s* p, const dpgo_ropt_params* prm, const double* dinv, bool* used, bool additive) {
  *used = false;
  p->gen += 1;
  if (p->persist_stream_ordered) {
    // dpgo_optimize_device_begin: the caller enqueues this handle's solves and everything between them on ONE stream, so
    // no two of its one-launch solves are ever resident together -- nothing to reserve (a reservation could only be
    // released by the collecting call, long after the kernel has left the chip)
    const PersistGeo g = persist_geometry(p, persist_capacity(p->device), 1, additive);
    if (g.wgs <= 0) return DPGO_OK;
    p->persist_split = g.split;
    p->persist_mt = g.mt;
    p->persist_wgs = g.wgs;
    p->persist_add = additive;
  } else if (p->persist_reserved == 0) {
    // Alone on the device: what is free now, first come first served.  Sharing it with other concurrently solved agents:
    // the most compact layout, at most 4/5 of the slots in use at once (a CU that holds a persistent workgroup has no
    // registers left for anything else, and every agent's other kernels -- gradient, retraction, rho test -- need
    // somewhere to run: packing the chip full made a 16-agent sweep slower), and a solve that finds no room WAITS for
    // another one to finish (a solve is well under a millisecond) instead of taking the slow path.
    const int cap = persist_capacity(p->device);
    auto& used = g_persist_used[p->device % kMaxDevices];
    PersistGeo g;
    if (p->persist_share <= 1) {
      g = persist_geometry(p, cap - used.load(), 1, additive);
      if (g.wgs <= 0 || !persist_reserve(p, g.slots, cap)) return DPGO_OK;
    } else {
      // (the additive form's grid is fixed by its hierarchy -- one workgroup per aggregate, up to the whole chip: such solves
      // take turns)
      const int limit = additive ? cap : cap - cap / 5;
      g = persist_geometry(p, limit, p->persist_share, additive);
      if (g.wgs <= 0) return DPGO_OK;
      const auto t0 = std::chrono::steady_clock::now();
      while (!persist_reserve(p, g.slots, limit)) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.05) return DPGO_OK;
        std::this_thread::yield();
      }
    }
    p->persist_split = g.split;
    p->persist_mt = g.mt;
    p->persist_wgs = g.wgs;
    p->persist_add = additive;
  }
  // (the granules' epochs are salted per launch, so what earlier launches left in the table never matches; the table is
  // cleared before a salt can repeat -- every 2047 generations of this handle -- and at creation)
  HIPC(hipMemsetAsync(p->pctrl, 0, sizeof(PersistCtrl), p->stream));
  if (p->gen - p->gran_cleared_at >= 0x7ffu) {  // (the multi-launch scheme advances `gen` too: count, do not test bits)
    HIPC(hipMemsetAsync(p->pgran, 0, sizeof(unsigned long long) * kGranWords, p->stream));
    p->gran_cleared_at = p->gen;
  }
  const unsigned salt = ((p->gen & 0x7ffu) + 1u) << 20;  // never 0; the in-launch step counter fills the low 20 bits
  // granule sweeps of the in-kernel all-reduce: wait before the first one (a granule needs ~1 us to cross the chip and
  // the slowest of more workgroups arrives later; sweeping earlier only loads the fabric: sphere2500 11.9 -> 7.9 us per
  // iteration, 12.5k slab 15.3 -> 12.1), back off between sweeps.  DPGO_POLL_FIRST / DPGO_POLL_SLEEP override.
  static const int env_first = [] { const char* e = std::getenv("DPGO_POLL_FIRST"); return e ? std::atoi(e) : -1; }();
  static const int env_sleep = [] { const char* e = std::getenv("DPGO_POLL_SLEEP"); return e ? std::atoi(e) : -1; }();
  // (whole-solve kernel, run r4j, us per product at first = 16 / 24 / 32 / 44 / 56: sphere2500, 157 workgroups of 4 lane
  // groups per pose, 7.3 / 6.5 / 7.0 / 7.6 / 8.3; 6 250 poses, 196 workgroups of the same layout with two tiles, 10.9 / 9.9 /
  // 9.9 / 10.5 / 11.1; 12.5k slab, one pose per (d+1) lanes, 12.0 / 10.9 / 10.7 / 10.5 / 10.4)
  const int first = env_first >= 0 ? std::min(255, env_first)
                                   : (p->persist_split == 4 ? (p->persist_wgs <= 160 ? kPollFirstSleep : 30) : 44);
  const int between = env_sleep >= 0 ? std::min(255, env_sleep) : kPollSleep;
  const int poll = (first << 8) | between;

  AddDev add{};
  size_t lds = 0;
  if (additive) {
    auto& L0 = p->ml[0];
    auto& C = p->ml[1];
    add = AddDev{L0.Pb, p->ml_dense, p->ml_lda, C.n, C.r, 1.0, L0.graph ? L0.tile_perm : nullptr};
    lds = sizeof(double) * (size_t)p->b * C.n * p->b;  // (d+1) rows of the inverse
  }
  const RtrArgs ra{prm->gradnorm_tol, prm->RTR_initial_radius, 5.0 * prm->RTR_initial_radius, prm->RTR_tCG_iterations,
                   prm->RTR_iterations, prm->accept_tiny_decrease};
  const double* Glin = p->has_G ? p->G : nullptr;
  // (static + dynamic LDS of the additive instances can exceed 64 KB: the attribute is raised once per handle, layout
  // and size)
#define PERSIST_LAUNCH(SP, MT_, ADD_, LDS_)                                                                           \
  do {                                                                                                                \
    if ((LDS_) > 0 && p->persist_lds_attr != (size_t)(LDS_) * 8 + SP) {                                               \
      HIPC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rtr_persist<D, R, SP, MT_, ADD_>),                     \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS_)));                             \
      p->persist_lds_attr = (size_t)(LDS_) * 8 + SP;                                                                  \
    }                                                                                                                 \
    hipLaunchKernelGGL((k_rtr_persist<D, R, SP, MT_, ADD_>), dim3(p->persist_wgs), dim3(kBlock), LDS_, p->stream,     \
                       p->Q.dev(), p->x1, Glin, dinv, p->x2, p->eta, p->z, p->pgran, salt, p->dstate, p->pctrl, p->n, \
                       p->hflag, p->gen, poll, ra, add);                                                              \
  } while (0)
  DISPATCH(p->d, p->r, {
    if (additive && p->persist_split == 4) PERSIST_LAUNCH(4, 1, true, lds);
    else if (additive) PERSIST_LAUNCH(1, 1, true, lds);
    else if (p->persist_split == 4 && p->persist_mt == 1) PERSIST_LAUNCH(4, 1, false, 0);
    else if (p->persist_split == 4) PERSIST_LAUNCH(4, 2, false, 0);
    else if (p->persist_mt == 1) PERSIST_LAUNCH(1, 1, false, 0);
    else PERSIST_LAUNCH(1, 2, false, 0);
  });
#undef PERSIST_LAUNCH
  {  // the iterate reaches the caller's X only if the launch completed on every participant (k_persist_commit)
    const size_t count = (size_t)p->n * p->T;
    const int grid = (int)std::min<size_t>(1024, (count + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_persist_commit, dim3(grid), dim3(kBlock), 0, p->stream, p->dstate, p->pctrl, p->x2, p->x1, count);
  }
  HIPC(hipGetLastError());
  p->cur = 0;
  *used = true;
  return DPGO_OK;
}

void persist_report(dpgo_problem_s* p) {  // (hctrl has been read back with the state record)
  if (!std::getenv("DPGO_PERSIST_VERBOSE")) return;
  const double it = std::max<double>(1.0, (double)p->hctrl->ticks[4]);
  std::fprintf(stderr,
               "dpgo_hip: persistent tCG: %u workgroups (%d lane groups per pose, %d tiles each)%s, %u iterations; per "
               "iteration (us): Hessian phase %.2f, all-reduce %.2f, update phase %.2f, all-reduce %.2f\n",
               p->hctrl->members, p->persist_split, p->persist_mt, p->hctrl->error ? " TIMED OUT" : "", p->hctrl->iters,
               0.01 * (double)p->hctrl->ticks[0] / it, 0.01 * (double)p->hctrl->ticks[1] / it,
               0.01 * (double)p->hctrl->ticks[2] / it, 0.01 * (double)p->hctrl->ticks[3] / it);
}

// One ROPTLIB SolversTR::Run outer iteration: tCG + retraction + rho test.  State stays on the device; the host
// feeds tCG-step kernels just-in-time (or polls the state every `tcg_poll_interval` inner iterations).
int rtr_outer_iteration(dpgo_problem_s* p, const dpgo_ropt_params* prm, const double* dinv, Counters& cnt,
                        bool poll_at_end) {
  p->gen += 1;
  const bool add = prm->precond == DPGO_PRECOND_ADDITIVE;
  // (additive: where the persistent kernel cannot run -- no free slots, an earlier time-out -- the V-cycle on the same
  // two-level hierarchy takes over)
  const bool ml = prm->precond == DPGO_PRECOND_MULTILEVEL || add;
  p->zr_from_post = ml;
  if (add) cnt.vcycle_for_additive = true;
  // multilevel: the update kernel writes the pre-smoothing step of level 0 instead of the block-Jacobi z; the cycle's
  // last kernel produces z and the partial sums <r,r>, <z,r>
  auto update = [&](int first) -> int {
    if (ml) {
      CHK(launch_tcg_update(p, dinv, first, p->ml[0].x1, p->ml_omega));
      static const bool early_stop = [] { const char* e = std::getenv("DPGO_ML_EARLY_STOP"); return !e || std::atoi(e) != 0; }();
      return launch_ml_tail(p, p->x1, p->rr, p->z, p->pB(), p->dstate + p->cur, early_stop && !first);
    }
    return launch_tcg_update(p, dinv, first);
  };
  CHK(update(1));
  const int max_inner = prm->RTR_tCG_iterations;
  auto step = [&](int j) -> int {
    CHK(launch_tcg_hess(p, j == 0 ? 1 : 0));
    return update(0);
  };
  if (max_inner <= 0) CHK(launch_tcg_hess(p, 1));  // only finalises the tCG state (eta = 0)
  bool done = false;
  if (prm->tcg_poll_interval > 0) {
    // polling mode: enqueue `poll` iterations, then synchronise and read the state back
    const int poll = prm->tcg_poll_interval;
    int j = 0;
    while (j < max_inner) {
      const int chunk = (max_inner - j) < poll ? (max_inner - j) : poll;
      for (int c = 0; c < chunk; ++c) CHK(step(j + c));
      j += chunk;
      CHK(poll_state(p));
      if (p->hstate->tcg_done || p->hstate->rtr_stop) {
        done = true;
        break;
      }
    }
  } else {
    // just-in-time feed: stay kAhead iterations ahead of the progress word the device publishes into
    // host-coherent memory; no synchronisation, no copy, at most kAhead wasted (early-exit) iterations
    // a multilevel iteration is 5+ launches: waste fewer of them after tCG stops.  (2 is the minimum: iteration j's count
    // is published by the prologue of iteration j+1's Hessian-step kernel, so one iteration ahead never sees progress --
    // tried in round 4, the watchdog fires.)
    int kAhead = ml ? 2 : 4;
    if (const char* e = std::getenv("DPGO_TCG_AHEAD")) kAhead = std::max(2, std::atoi(e));  // tuning knob (>= 2, see above)
    int enq = 0, last_j = -1;
    auto t_progress = std::chrono::steady_clock::now();
    while (true) {
      const unsigned long long w = __atomic_load_n(p->hflag, __ATOMIC_ACQUIRE);
      int dev_j = 0;
      if ((unsigned)(w >> 32) == p->gen) {
        dev_j = (int)((w >> 8) & 0xFFFFFFu);
        if (w & 3ull) {
          done = true;
          if (w & 2ull) p->saw_rtr_stop = true;
          break;
        }
      }
      if (enq >= max_inner) break;
      if (dev_j != last_j) {  // the watchdog measures time WITHOUT progress, not time since the loop started
        last_j = dev_j;
        t_progress = std::chrono::steady_clock::now();
      }
      if (enq < dev_j + kAhead) {
        CHK(step(enq));
        enq += 1;
      } else {
        __builtin_ia32_pause();
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_progress).count() > 60.0) {
          HIPC(hipStreamSynchronize(p->stream));  // surfaces a device fault instead of spinning forever
          return fail(DPGO_ERR_HIP, "tCG progress word did not advance for 60 s");
        }
      }
    }
  }
  if (!done) CHK(launch_tcg_hess(p, 0));  // max_inner iterations enqueued, not (yet known to be) finished: last prologue
  if (p->saw_rtr_stop) return DPGO_OK;  // the previous outer iteration already met the stop test
  CHK(launch_retract(p, p->x1, p->eta, 1.0, p->x2, p->dstate + p->cur));
  CHK(launch_grad(p, p->x2, p->g2, p->S2, nullptr, p->dstate + p->cur, p->tcg_sym && outer_sym_enabled()));
  cnt.spmm += 1;
  CHK(launch_hess(p, p->x1, p->S1, p->eta, p->g1, p->Hd, p->pH(), p->dstate + p->cur, 0, p->tcg_sym && outer_sym_enabled()));
  cnt.spmm += 1;
  CHK(launch_rtr_update(p));
  if (poll_at_end) CHK(poll_state(p));
  return DPGO_OK;
}

// DPGO_PRECOND_AUTO after a solve: what the next one runs (dpgo_hip.h).  `used` = the preconditioner the solve resolved to,
// `products` = its Hessian-vector products.
//   * a block without coupling to other agents never hands back (its cheap early calls end on the trust-region boundary
//     after a few products whatever the preconditioner, and the switch back and forth cost the 100k grid 37.6 against
//     29.3 ms);
//   * a coupled block the additive one-launch solve cannot hold: hysteresis on the share of the tCG budget a solve used
//     (the V-cycle is ~3x a block-Jacobi iteration: it pays when the budget binds);
//   * a coupled block the additive solve CAN hold (<= ~14 000 poses): the cost rule.  Q is constant across RBCD sweeps, so
//     the hierarchy is paid once: when the block-Jacobi solves since the last change of Q have cost as much as one set-up
//     (kAutoSetupUnits: 2.8-3.0 ms against 9.9-10.6 us per block-Jacobi product on 6 250 / 12 500 poses), the next solve
//     runs additive on trial; it stays while its products x the additive unit cost (19 us) stay below the reference
//     block-Jacobi solve's x the block-Jacobi unit cost, and hands back otherwise (the hierarchy is kept: the next trial
//     is free but waits twice as long).
//     The unit costs are those of a solve that has the device to itself.  A handle solved NEXT TO others of the device
//     (dpgo_optimize_device_many, persist_share > 1) is charged for the part of the chip its launch blocks instead: the
//     additive form owns one CU per aggregate (up to the whole chip: such solves take turns), block-Jacobi's compact
//     layout a quarter to a half of it, so there a product costs [us] x max(resident slots / slots in use at once, 1 / share)
//     x share -- the same figure as alone whenever every concurrently solved handle fits at once.
constexpr int kAutoUnitsJacobi = 10, kAutoUnitsAdditive = 18, kAutoSetupUnits = 2800, kAutoMinProducts = 6;
int auto_units_jacobi(dpgo_problem_s* p) {
  const int share = std::max(1, p->persist_share);
  if (share == 1 || !p->persist) return kAutoUnitsJacobi;
  const int cap = persist_capacity(p->device), limit = cap - cap / 5;  // (what launch_rtr_persistent lets such solves use)
  const PersistGeo g = persist_geometry(p, limit, share, false);
  if (g.wgs <= 0) return kAutoUnitsJacobi;
  const double part = std::max((double)g.slots / limit, 1.0 / share);
  return std::max(1, (int)std::lround(kAutoUnitsJacobi * (g.mt == 2 ? 1.25 : 1.0) * part * share));
}
int auto_units_additive(dpgo_problem_s* p) {  // (after additive_available(p): the plan exists)
  const int share = std::max(1, p->persist_share);
  if (share == 1) return kAutoUnitsAdditive;
  const int cap = persist_capacity(p->device);
  const double part = std::max((double)(p->add_plan.na * persist_slots_per_wg(p->add_plan.split, 1, true)) / cap, 1.0 / share);
  return std::max(1, (int)std::lround(kAutoUnitsAdditive * part * share));
}
void auto_update(dpgo_problem_s* p, const dpgo_ropt_params* prm, int used, int products) {
  const int budget = std::max(1, prm->RTR_iterations) * std::max(1, prm->RTR_tCG_iterations);
  const bool coupled = p->has_G || p->C.nnzb > 0;
  auto& a = p->auto_cost;
  a.last_used = used;
  a.last_products = products;
  if (!coupled) {
    if (!p->auto_ml && 2 * products >= budget) p->auto_ml = true;
    return;
  }
  static const bool cost_rule = [] { const char* e = std::getenv("DPGO_AUTO_COST_RULE"); return !e || std::atoi(e) != 0; }();
  if (!p->auto_ml) {  // the solve ran block-Jacobi
    a.state = 0;
    a.uj = auto_units_jacobi(p);
    a.jac_units += (long long)kAutoUnitsJacobi * products;  // (the set-up is wall time: paid back in solo units)
    const bool binds = 2 * products >= budget;
    const bool paid = cost_rule && products >= kAutoMinProducts && a.jac_units >= ((long long)kAutoSetupUnits << a.backoff);
    if (binds || paid) {
      // (the plan -- host aggregation, once per block pattern -- is only looked for when the rule wants it)
      const bool add = cost_rule && !p->ml_user_ks && additive_available(p);
      if (add) a.ua = auto_units_additive(p);
      // a trial that cannot win is not run: even at kAutoMinProducts the additive solve would cost more than this one did
      const bool hopeless = add && !binds && (long long)a.ua * kAutoMinProducts >= (long long)a.uj * products;
      if (hopeless) {
        a.jac_units = 0;
        a.backoff = std::min(a.backoff + 1, 6);
      } else if (binds || add) {
        p->auto_ml = true;
        if (add) {
          a.state = 1;
          a.ref = products;
          a.switches += 1;
        }
      }
    }
    return;
  }
  if (a.state == 0) {  // a multilevel choice outside the cost rule (V-cycle blocks, dpgo_problem_auto_state): budget hysteresis
    if (10 * products <= budget) p->auto_ml = false;
    return;
  }
  a.ua = auto_units_additive(p);
  // on trial: strictly cheaper than the reference solve; once accepted: handed back only when 15 % dearer (the two are
  // within a few per cent of each other on interior blocks of a chain partition -- no flapping)
  const long long cost = (long long)a.ua * products * 100, ref = (long long)a.uj * a.ref * (a.state == 2 ? 115 : 100);
  if (cost < ref) {
    a.state = 2;
  } else {  // no cheaper than block-Jacobi on this block in this phase of the run: hand back, try again later
    p->auto_ml = false;
    a.state = 0;
    a.jac_units = 0;
    a.backoff = std::min(a.backoff + 1, 6);
  }
}

// phase: RUN_FULL = the whole solve, synchronously.  RUN_BEGIN = enqueue only: if the solve is a one-launch solve
// (k_rtr_persist) the call returns with the launch, its commit kernel and the read-backs in flight (p->pending.launched);
// otherwise the solve runs to completion right here.  RUN_END = collect what RUN_BEGIN left in flight (waits for the
// stream, reads the state record, falls back to the multi-launch scheme after a time-out exactly as the synchronous call).
enum { RUN_FULL = 0, RUN_BEGIN = 1, RUN_END = 2 };
int run_optimize(dpgo_problem_s* p, const dpgo_ropt_params* prm, dpgo_ropt_result* res, int phase = RUN_FULL) {
  // X is in p->x1 on entry and on exit.  src/QuadraticOptimizer.cpp:26-48.
  auto t0 = std::chrono::steady_clock::now();
  Counters cnt;
  std::memset(res, 0, sizeof(*res));
  res->tCGStatus = DPGO_TCG_MAXITER;
  const bool resume = phase == RUN_END;
  if (p->hctrl && !resume) std::memset(p->hctrl, 0, sizeof(PersistCtrl));
  struct SlotGuard {  // the resident-slot reservation of the persistent kernel lives as long as the solve
    dpgo_problem_s* p;
    bool armed;
    ~SlotGuard() {
      if (armed) persist_release(p);
    }
  } slot_guard{p, true};
  dpgo_ropt_params resolved = resume ? p->pending.resolved : *prm;  // DPGO_PRECOND_AUTO -> what this handle currently runs
  if (!resume) {
    if (prm->precond == DPGO_PRECOND_AUTO) p->auto_decide();
    if (prm->precond == DPGO_PRECOND_AUTO)  // (the multilevel choice: the additive form wherever its persistent kernel runs)
      resolved.precond = (p->auto_ml && prm->method == DPGO_METHOD_RTR)
                             ? ((additive_available(p) && !p->ml_user_ks) ? DPGO_PRECOND_ADDITIVE : DPGO_PRECOND_MULTILEVEL)
                             : DPGO_PRECOND_BLOCK_JACOBI;
  }
  const bool is_auto = resume ? p->pending.is_auto : prm->precond == DPGO_PRECOND_AUTO;
  if (resume) t0 = p->pending.t0;
  prm = &resolved;
  const double* dinv = nullptr;
  if (resume) {
    dinv = p->pending.dinv;
  } else
  if (prm->precond == DPGO_PRECOND_BLOCK_JACOBI) {
    CHK(build_dinv(p, prm->precond_shift));
    dinv = p->dinv;
  } else if (prm->precond == DPGO_PRECOND_MULTILEVEL) {
    // built lazily for the current Q, like the reference's factor (src/PoseGraph.cpp:582-586)
    CHK(ml_ensure(p, prm->precond_shift));
    dinv = p->dinv;  // the smoother's block-Jacobi factors (same shift)
  } else if (prm->precond == DPGO_PRECOND_ADDITIVE) {
    if (prm->method != DPGO_METHOD_RTR) return fail(DPGO_ERR_UNSUPPORTED, "the additive preconditioner exists inside the tCG loop only");
    if (!additive_split_of(p) && !additive_plan(p).split)
      return fail(DPGO_ERR_UNSUPPORTED, "additive preconditioner: block too large (at most 256 aggregates of one workgroup tile, " +
                                            std::to_string(ml_tile(p->b, 1)) + " poses)");
    CHK(ml_ensure(p, prm->precond_shift, /*additive=*/true));
    dinv = p->dinv;
  } else if (prm->precond != DPGO_PRECOND_NONE) {
    return fail(DPGO_ERR_INVALID, "unknown preconditioner");
  }
  if (!resume) {
    p->loop_extra_bytes = 0;
    if (prm->precond == DPGO_PRECOND_MULTILEVEL && !p->ml.empty()) {
      const size_t nd = (size_t)p->ml_lda;
      size_t bytes = nd * nd * (size_t)(p->ml_coarse_bits / 8) / (p->ml_use_dense_sym() ? 2 : 1);
      const auto& L0 = p->ml[0];
      bytes += (size_t)L0.AP.nnzb * (sizeof(double) * p->b * p->b + sizeof(int32_t)) + sizeof(double) * (size_t)p->n * p->b * p->b;
      bytes += 3 * p->vec_bytes();
      p->loop_extra_bytes = bytes;
    }
    CHK(resolve_tcg_storage(p));
  }
  // ---- blocks in the latency regime: the whole solve is ONE persistent launch (k_rtr_persist) and one read-back.  The
  // single-iteration radius-shrink mode (:80-99) and the polling mode keep the multi-launch scheme.
  const bool add = prm->precond == DPGO_PRECOND_ADDITIVE;
  // (the in-kernel all-reduce tags its granules with salt | step, the step counter in the low 20 bits: a solve whose
  // parameters allow more reductions than that -- at most 3 per tCG iteration + 4 per outer iteration + 1 -- keeps the
  // multi-launch scheme)
  const bool epochs_fit = (long long)std::max(1, prm->RTR_iterations) * (3LL * std::max(0, prm->RTR_tCG_iterations) + 4) + 1 < (1LL << 20);
  if (resume || (prm->method == DPGO_METHOD_RTR && prm->RTR_iterations != 1 && prm->tcg_poll_interval <= 0 && p->persist && epochs_fit &&
                 !p->persist_failed_once && (add || prm->precond == DPGO_PRECOND_BLOCK_JACOBI || prm->precond == DPGO_PRECOND_NONE))) {
    bool used = resume;
    if (!resume) CHK(launch_rtr_persistent(p, prm, dinv, &used, add));
    if (used) {
      if (!resume) {
        HIPC(hipMemcpyAsync(p->hctrl, p->pctrl, sizeof(PersistCtrl), hipMemcpyDeviceToHost, p->stream));
        HIPC(hipMemcpyAsync(p->hstate, p->dstate + p->cur, sizeof(DevState), hipMemcpyDeviceToHost, p->stream));
        if (phase == RUN_BEGIN) {  // everything of the solve is enqueued: the caller collects it with RUN_END
          auto& pd = p->pending;
          pd.launched = true;
          pd.resolved = resolved;
          pd.is_auto = is_auto;
          pd.dinv = dinv;
          pd.t0 = t0;
          slot_guard.armed = false;  // (the reservation is released by the collecting call)
          return DPGO_OK;
        }
      }
      HIPC(hipStreamSynchronize(p->stream));
      persist_report(p);
      // (k_persist_commit, which ran behind the solve, saw the same two words: with a poisoned record OR a raised time-out
      // flag -- some participant gave up, however late -- the caller's iterate has not been touched)
      if (p->hstate->rtr_stop != kPersistPoison && !p->hctrl->error) {
        const DevState& h = *p->hstate;
        res->fInit = h.fInit;
        res->gradNormInit = h.gnInit;
        res->fOpt = h.f1;
        res->gradNormOpt = h.ngf;
        res->tCGStatus = h.outer_iter > 0 ? h.tcg_status : DPGO_TCG_MAXITER;
        res->rtr_iterations = h.outer_iter;
        res->rtr_accepted = h.n_accept;
        res->latest_step_accepted = h.accepted_last;
        res->tcg_iterations = h.n_hess;
        res->precond_used = prm->precond;
        res->spmm_count = 1 + 2 * h.outer_iter + h.n_hess;
        if (is_auto) auto_update(p, prm, prm->precond, h.n_hess);
        res->success = 1;  // :44
        res->elapsedMs = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return DPGO_OK;
      }
      // a time-out (the launch's workgroups were not all resident: another process on the device): the caller's iterate is
      // untouched; this handle stops using the kernel and the solve runs on the multi-launch scheme
      p->persist_failed_once = true;
      persist_release(p);
      if (std::getenv("DPGO_PERSIST_VERBOSE"))
        std::fprintf(stderr, "dpgo_hip: persistent solve timed out; this handle continues with the multi-launch scheme\n");
    }
  }
  // statistics before optimisation (:28-29) -- one fused pass: f, rgrad, S
  CHK(launch_grad(p, p->x1, p->g1, p->S1, nullptr, nullptr, prm->method == DPGO_METHOD_RTR && p->tcg_sym && outer_sym_enabled()));
  cnt.spmm += 1;
  CHK(launch_rtr_begin(p, prm->gradnorm_tol, prm->RTR_initial_radius, 5.0 * prm->RTR_initial_radius,
                       prm->RTR_tCG_iterations, prm->accept_tiny_decrease));
  // The initial statistics are read back together with the final ones when the solve is fed just-in-time (one
  // synchronisation per call instead of two): an iterate that already meets the tolerance (:57-59) makes the first tCG
  // launch publish rtr_stop, which ends the loop below before anything is changed.
  const bool deferred = prm->method == DPGO_METHOD_RTR && prm->RTR_iterations != 1 && prm->tcg_poll_interval <= 0;
  if (!deferred) {
    CHK(poll_state(p));
    res->fInit = p->hstate->fInit;
    res->gradNormInit = p->hstate->gnInit;
  } else {
    p->hstate->rtr_stop = 0;
  }
  int n_hess_total = 0;
  int shrink_tries = 0;

  if (prm->method == DPGO_METHOD_RTR) {
    // trustRegion(): src/QuadraticOptimizer.cpp:50-108
    if (!p->hstate->rtr_stop) {  // :57-59 early-out
      if (prm->RTR_iterations == 1) {  // :80-99 shrink the radius until the step is accepted
        double radius = prm->RTR_initial_radius;
        int total_steps = 0;
        while (true) {
          shrink_tries += 1;
          p->hstate->Delta = radius;
          p->hstate->Delta_max = radius;
          p->hstate->outer_iter = 0;
          CHK(push_state(p));
          p->saw_rtr_stop = false;
          CHK(rtr_outer_iteration(p, prm, dinv, cnt, true));
          if (p->hstate->accepted_last) break;
          if (total_steps > 10) break;  // "Too many RTR rejections. Returning initial guess." (x1 untouched)
          radius /= 4.0;
          total_steps++;
        }
      } else {
        const bool polling = prm->tcg_poll_interval > 0;
        p->saw_rtr_stop = false;
        for (int it = 0; it < prm->RTR_iterations; ++it) {
          CHK(rtr_outer_iteration(p, prm, dinv, cnt, polling));
          if (polling ? (p->hstate->rtr_stop != 0) : p->saw_rtr_stop) break;
          const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
          if (prm->time_bound_s > 0 && el > prm->time_bound_s) break;  // Solver.TimeBound (:78)
        }
        CHK(poll_state(p));
        if (deferred) {
          res->fInit = p->hstate->fInit;
          res->gradNormInit = p->hstate->gnInit;
        }
      }
      res->tCGStatus = p->hstate->tcg_status;
      res->rtr_iterations = (prm->RTR_iterations == 1) ? shrink_tries : p->hstate->outer_iter;
      res->rtr_accepted = p->hstate->n_accept;
      res->latest_step_accepted = p->hstate->accepted_last;
      n_hess_total = p->hstate->n_hess;
    }
    res->fOpt = p->hstate->f1;
    res->gradNormOpt = p->hstate->ngf;
  } else if (prm->method == DPGO_METHOD_RGD) {
    // gradientDescent(): src/QuadraticOptimizer.cpp:110-137 (one fixed-step preconditioned step)
    const double* step = p->g1;
    if (prm->RGD_use_preconditioner) {
      if (prm->precond == DPGO_PRECOND_MULTILEVEL)
        CHK(launch_ml_apply(p, p->x1, p->g1, p->z));
      else
        CHK(launch_precond(p, p->x1, p->g1, dinv, p->z));
      step = p->z;
    }
    CHK(launch_retract(p, p->x1, step, -prm->RGD_stepsize, p->x2, nullptr));
    HIPC(hipMemcpyAsync(p->x1, p->x2, p->vec_bytes(), hipMemcpyDeviceToDevice, p->stream));
    CHK(launch_grad(p, p->x1, p->g1, p->S1, nullptr));
    cnt.spmm += 1;
    CHK(launch_rtr_begin(p, prm->gradnorm_tol, prm->RTR_initial_radius, 5.0 * prm->RTR_initial_radius,
                         prm->RTR_tCG_iterations, prm->accept_tiny_decrease));
    CHK(poll_state(p));
    res->fOpt = p->hstate->f1;
    res->gradNormOpt = p->hstate->ngf;
  } else {
    return fail(DPGO_ERR_INVALID, "unknown method");
  }
  res->tcg_iterations = n_hess_total;
  res->precond_used = (prm->precond == DPGO_PRECOND_ADDITIVE && cnt.vcycle_for_additive) ? DPGO_PRECOND_MULTILEVEL : prm->precond;
  if (is_auto && prm->method == DPGO_METHOD_RTR) auto_update(p, prm, res->precond_used, n_hess_total);
  res->spmm_count = cnt.spmm + n_hess_total;
  res->success = 1;  // :44 (set unconditionally after a solve)
  res->elapsedMs = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return DPGO_OK;
}

int check_ready(dpgo_problem_s* p) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  if (!p->Q.vals) return fail(DPGO_ERR_STATE, "quadratic matrix Q not set");
  return set_device(p);
}

int h2d(dpgo_problem_s* p, double* dst, const double* src) {
  HIPC(hipMemcpyAsync(dst, src, p->vec_bytes(), hipMemcpyHostToDevice, p->stream));
  return DPGO_OK;
}
int d2h(dpgo_problem_s* p, double* dst, const double* src) {
  HIPC(hipMemcpyAsync(dst, src, p->vec_bytes(), hipMemcpyDeviceToHost, p->stream));
  HIPC(hipStreamSynchronize(p->stream));
  return DPGO_OK;
}

}  // namespace

namespace {
int free_edges(dpgo_problem_s* p) {
  void* ptrs[] = {p->e_p1,  p->e_p2,    p->c_ptr,  p->c_edge,   p->e_R,    p->e_t,    p->e_kappa,
                  p->e_tau, p->e_w,     p->e_rsq,  p->q_base,   p->e_fixed, p->c_kind, p->e_counts,
                  p->e_role, p->e_slot, p->g_ptr,  p->g_edge,   p->g_kind, p->c_base};
  for (void* q : ptrs)
    if (q) (void)hipFree(q);
  p->e_role = p->g_kind = nullptr;
  p->e_slot = p->g_ptr = p->g_edge = nullptr;
  p->c_base = nullptr;
  p->n_shared_edges = 0;
  p->e_p1 = p->e_p2 = p->c_ptr = p->c_edge = nullptr;
  p->e_R = p->e_t = p->e_kappa = p->e_tau = p->e_w = p->e_rsq = p->q_base = nullptr;
  p->e_fixed = p->c_kind = nullptr;
  p->e_counts = nullptr;
  p->em = 0;
  return DPGO_OK;
}
int rebuild_vals(dpgo_problem_s* p, int nnzb, const int32_t* cptr, const int32_t* cedge, const uint8_t* ckind,
                 const double* base, double sign, double* out) {
  if (nnzb <= 0) return DPGO_OK;
  const int g = std::max(1, std::min(kMaxGrid, (nnzb + kBlock - 1) / kBlock));
  if (p->d == 2)
    hipLaunchKernelGGL(k_rebuild_Q<2>, dim3(g), dim3(kBlock), 0, p->stream, p->edges(), cptr, cedge, ckind, base,
                       sign, out, nnzb);
  else
    hipLaunchKernelGGL(k_rebuild_Q<3>, dim3(g), dim3(kBlock), 0, p->stream, p->edges(), cptr, cedge, ckind, base,
                       sign, out, nnzb);
  HIPC(hipGetLastError());
  return DPGO_OK;
}
int rebuild_Q_from_weights(dpgo_problem_s* p, const double* base, double sign, double* out) {
  return rebuild_vals(p, p->Q.nnzb, p->c_ptr, p->c_edge, p->c_kind, base, sign, out);
}
int rebuild_C_from_weights(dpgo_problem_s* p, const double* base, double sign, double* out) {
  if (!p->g_ptr) return DPGO_OK;
  return rebuild_vals(p, p->C.nnzb, p->g_ptr, p->g_edge, p->g_kind, base, sign, out);
}
int refresh_after_weights(dpgo_problem_s* p) {
  CHK(rebuild_Q_from_weights(p, p->q_base, 1.0, p->Q.vals));
  CHK(rebuild_C_from_weights(p, p->c_base, 1.0, p->C.vals));  // G itself is refreshed by the next update_G call
  p->ml_ready = false;
  p->auto_decided = false;
  p->sym.ready = p->tcg_sym = false;
  const double s = p->dinv_shift > 0 ? p->dinv_shift : 1e-1;
  p->dinv_shift = -1.0;  // clearQuadraticMatrix also drops the preconditioner (src/PoseGraph.cpp:352-355)
  return build_dinv(p, s);
}
}  // namespace

namespace {
// The two kernels of the tCG loop run as persistent grids: one workgroup per resident slot (occupancy x CUs).
// More workgroups than slots only add prologues (state record + partial-sum reduction) and a ragged second
// round: 100k poses, same box: caps 1024/1024 -> 70.7 us per tCG iteration, 512/768 (= the resident counts of
// k_tcg_update / k_tcg_hess at 203 / 164 VGPRs) -> 64.7 us.  The other kernels keep the family's cap.
template <class K>
int resident_blocks(K kernel, int* out) {
  int per_cu = 0;
  HIPC(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0));
  int dev = 0, cus = 0;
  HIPC(hipGetDevice(&dev));
  HIPC(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  *out = std::max(1, std::min(kPartialCap, per_cu * cus));
  return DPGO_OK;
}
int tune_launch_caps(dpgo_problem_s* p) {
  DISPATCH(p->d, p->r, {
    if constexpr (Span<D, R, 1>::kOk) {
      CHK(resident_blocks(k_tcg_update_span<D, R>, &p->cap_u));
      if (p->split == 4)
        CHK(resident_blocks(k_tcg_hess_span<D, R, 4>, &p->cap_h));
      else if (p->split == 2)
        CHK(resident_blocks(k_tcg_hess_span<D, R, 2>, &p->cap_h));
      else
        CHK(resident_blocks(k_tcg_hess_span<D, R, 1>, &p->cap_h));
      CHK(resident_blocks(k_tcg_hess_sym<D, R, 1>, &p->cap_hs));
    } else {
      CHK(resident_blocks(k_tcg_update<D, R>, &p->cap_u));
      if (p->split == 4)
        CHK(resident_blocks(k_tcg_hess<D, R, 4>, &p->cap_h));
      else if (p->split == 2)
        CHK(resident_blocks(k_tcg_hess<D, R, 2>, &p->cap_h));
      else
        CHK(resident_blocks(k_tcg_hess<D, R, 1>, &p->cap_h));
    }
  });
  DISPATCH(p->d, p->r, {
    if (p->split == 4) {
      CHK(resident_blocks(k_ml_restrict<D, R, 4, BsrDev>, &p->cap_restrict));
      CHK(resident_blocks(k_ml_post_ap<D, R, 4>, &p->cap_post));
    } else if (p->split == 2) {
      CHK(resident_blocks(k_ml_restrict<D, R, 2, BsrDev>, &p->cap_restrict));
      CHK(resident_blocks(k_ml_post_ap<D, R, 2>, &p->cap_post));
    } else {
      CHK(resident_blocks(k_ml_restrict<D, R, 1, BsrDev>, &p->cap_restrict));
      CHK(resident_blocks(k_ml_post_ap<D, R, 1>, &p->cap_post));
    }
  });
  if (const char* e = std::getenv("DPGO_GRID_ML")) {
    p->cap_restrict = p->cap_post = std::max(1, std::min(kPartialCap, std::atoi(e)));
  }
  // tuning knobs (any value up to the partial-sum capacity is valid)
  if (const char* e = std::getenv("DPGO_GRID_UPDATE")) p->cap_u = std::max(1, std::min(kPartialCap, std::atoi(e)));
  if (const char* e = std::getenv("DPGO_GRID_HESS")) p->cap_h = std::max(1, std::min(kPartialCap, std::atoi(e)));
  if (const char* e = std::getenv("DPGO_GRID_HESS_SYM")) p->cap_hs = std::max(1, std::min(kPartialCap, std::atoi(e)));
  return DPGO_OK;
}

// One-launch solve: on by size -- every block the kernel can hold (two 64-pose tiles on each of 256 workgroups: 32 768 poses
// in 3-D; measured per Hessian-vector product against the multi-launch scheme: 625 poses 6.6 / 13.7 us, sphere2500 6.5 /
// 15.4, 6 250 9.9 / 17.2, 12.5k slab 10.5 / 19.7, 25k 18.7 / 26.5).  DPGO_PERSIST_MAX_POSES lowers the limit, DPGO_PERSIST=0/1
// overrides.
int tune_persist(dpgo_problem_s* p) {
  static const int max_poses = [] { const char* e = std::getenv("DPGO_PERSIST_MAX_POSES"); return e ? std::atoi(e) : 1 << 30; }();
  const bool fits = persist_geometry(p, persist_capacity(p->device)).wgs > 0;
  bool on = fits && p->n <= max_poses;
  if (const char* e = std::getenv("DPGO_PERSIST")) on = fits && std::atoi(e) != 0;
  p->persist = on;
  return DPGO_OK;
}

}  // namespace

// =====================================================================================
extern "C" {

const char* dpgo_version(void) { return "dpgo_hip 0.1 (gfx950)"; }
const char* dpgo_last_error(void) { return g_err.c_str(); }
void dpgo_set_last_error(const char* msg) { g_err = msg ? msg : ""; }  // for the other translation units of the library

int dpgo_device_count(int* count) {
  if (!count) return fail(DPGO_ERR_INVALID, "null count");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    *count = 0;
    return fail(DPGO_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  }
  *count = c;
  return DPGO_OK;
}

void dpgo_ropt_params_default(dpgo_ropt_params* p) {
  if (!p) return;
  p->method = DPGO_METHOD_RTR;
  p->verbose = 0;
  p->gradnorm_tol = 1e-2;
  p->RGD_stepsize = 1e-3;
  p->RGD_use_preconditioner = 1;
  p->RTR_iterations = 3;
  p->RTR_tCG_iterations = 50;
  p->RTR_initial_radius = 100.0;
  p->precond = DPGO_PRECOND_AUTO;
  p->precond_shift = 1e-1;
  p->accept_tiny_decrease = 1;
  p->tcg_poll_interval = 0;
  p->time_bound_s = 5.0;
}

int dpgo_supported(int d, int r) { return supported(d, r) ? 1 : 0; }

int dpgo_problem_create(dpgo_problem_t* out, int r, int d, int n, int device) {
  if (!out) return fail(DPGO_ERR_INVALID, "null out");
  *out = nullptr;
  if (n <= 0 || r < d || d < 2 || d > 3) return fail(DPGO_ERR_INVALID, "need n > 0, r >= d, d in {2,3}");
  if (!supported(d, r)) return fail(DPGO_ERR_UNSUPPORTED, "(d, r) not compiled in");
  int cnt = 0;
  CHK(dpgo_device_count(&cnt));
  if (cnt <= 0) return fail(DPGO_ERR_HIP, "no HIP device (this library has no CPU fallback)");
  if (device < 0 || device >= cnt) return fail(DPGO_ERR_INVALID, "device index out of range");
  auto* p = new dpgo_problem_s();
  p->r = r;
  p->d = d;
  p->n = n;
  p->b = d + 1;
  p->T = p->b * r;
  p->device = device;
  // small blocks are latency-bound: spread each row over 4 lane groups
  p->split = (n < 40000) ? 4 : 1;
  if (const char* e = std::getenv("DPGO_SPLIT")) {
    const int v = std::atoi(e);
    if (v == 1 || v == 2 || v == 4) p->split = v;
  }
  int rc = [&]() -> int {
    HIPC(hipSetDevice(device));
    CHK(tune_launch_caps(p));
    HIPC(hipStreamCreateWithFlags(&p->own_stream, hipStreamNonBlocking));
    p->stream = p->own_stream;
    const size_t vb = p->vec_bytes();
    double** vecs[] = {&p->x1, &p->x2, &p->g1, &p->g2, &p->eta, &p->delta, &p->Hd, &p->rr, &p->z, &p->G, &p->G0};
    for (auto v : vecs) {
      HIPC(hipMalloc(v, vb));
      HIPC(hipMemsetAsync(*v, 0, vb, p->stream));
    }
    HIPC(hipMalloc(&p->S1, sizeof(double) * (size_t)n * d * d));
    HIPC(hipMalloc(&p->S2, sizeof(double) * (size_t)n * d * d));
    HIPC(hipMalloc(&p->dinv, sizeof(double) * (size_t)n * p->b * p->b));
    HIPC(hipMalloc(&p->partials, sizeof(double) * 5 * kPartialCap * kNP));
    HIPC(hipMemsetAsync(p->partials, 0, sizeof(double) * 5 * kPartialCap * kNP, p->stream));
    HIPC(hipMalloc(&p->dstate, sizeof(DevState) * 2));
    HIPC(hipHostMalloc(&p->hstate, sizeof(DevState)));
    HIPC(hipMalloc(&p->pctrl, sizeof(PersistCtrl)));
    HIPC(hipMalloc(&p->pgran, sizeof(unsigned long long) * kGranWords));
    HIPC(hipMemsetAsync(p->pgran, 0, sizeof(unsigned long long) * kGranWords, p->stream));
    HIPC(hipHostMalloc(&p->hctrl, sizeof(PersistCtrl)));
    CHK(tune_persist(p));
    HIPC(hipHostMalloc(&p->hflag, 64, hipHostMallocCoherent | hipHostMallocMapped));
    *p->hflag = 0ull;
    HIPC(hipStreamSynchronize(p->stream));
    return DPGO_OK;
  }();
  if (rc != DPGO_OK) {
    dpgo_problem_destroy(p);
    return rc;
  }
  *out = p;
  return DPGO_OK;
}

int dpgo_problem_destroy(dpgo_problem_t p) {
  if (!p) return DPGO_OK;
  (void)hipSetDevice(p->device);
  if (p->own_stream) (void)hipStreamSynchronize(p->own_stream);
  free_bsr(p->Q);
  free_bsr(p->C);
  ml_free(p);
  sym_free(p);
  free_edges(p);
  double* vecs[] = {p->x1, p->x2, p->g1, p->g2, p->eta, p->delta, p->Hd, p->rr, p->z, p->G, p->G0,
                    p->S1, p->S2, p->dinv, p->partials};
  for (auto v : vecs)
    if (v) (void)hipFree(v);
  if (p->dstate) (void)hipFree(p->dstate);
  if (p->hstate) (void)hipHostFree(p->hstate);
  if (p->hflag) (void)hipHostFree(p->hflag);
  if (p->pctrl) (void)hipFree(p->pctrl);
  if (p->pgran) (void)hipFree(p->pgran);
  if (p->hctrl) (void)hipHostFree(p->hctrl);
  if (p->own_stream) (void)hipStreamDestroy(p->own_stream);
  delete p;
  return DPGO_OK;
}

int dpgo_problem_set_stream(dpgo_problem_t p, void* hip_stream) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  CHK(set_device(p));
  HIPC(hipStreamSynchronize(p->stream));
  p->stream = (hipStream_t)hip_stream;  // NULL = the default (null) stream
  return DPGO_OK;
}

int dpgo_problem_use_own_stream(dpgo_problem_t p) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  CHK(set_device(p));
  HIPC(hipStreamSynchronize(p->stream));
  p->stream = p->own_stream;
  return DPGO_OK;
}

int dpgo_problem_dims(dpgo_problem_t p, int* r, int* d, int* n, int* nnzb) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  if (r) *r = p->r;
  if (d) *d = p->d;
  if (n) *n = p->n;
  if (nnzb) *nnzb = p->Q.nnzb;
  return DPGO_OK;
}

int dpgo_problem_set_Q_bsr(dpgo_problem_t p, int nnzb, const int32_t* rowptr, const int32_t* colidx,
                           const double* vals) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  if (!vals) return fail(DPGO_ERR_INVALID, "null vals");
  CHK(validate_bsr(p->n, p->n, nnzb, rowptr, colidx, true));
  CHK(set_device(p));
  // registered re-weightable edges index into the old pattern: drop them (the caller re-registers)
  if (p->e_w) CHK(free_edges(p));
  CHK(upload_bsr(p->Q, p->n, p->n, nnzb, p->b, rowptr, colidx, vals, p->stream));
  const bool same_pattern = (int)p->h_rowptr.size() == p->n + 1 && (int)p->h_colidx.size() == nnzb &&
                            std::equal(rowptr, rowptr + p->n + 1, p->h_rowptr.begin()) &&
                            std::equal(colidx, colidx + nnzb, p->h_colidx.begin());
  if (!same_pattern) {
    p->h_rowptr.assign(rowptr, rowptr + p->n + 1);
    p->h_colidx.assign(colidx, colidx + nnzb);
    p->add_plan_known = false;
    if (p->ml_user_ks && p->ml_symbolic) {  // keep the caller's aggregate sizes across a pattern change
      const int perm_tile = p->ml[0].perm_tile;
      CHK(ml_symbolic_setup(p, ml_current_ks(p), perm_tile));
    } else {
      ml_free(p);
    }
    sym_free(p);
  }
  p->sym.ready = p->tcg_sym = false;
  p->ml_ready = false;  // the hierarchy's values belong to the old Q: rebuilt on the device at the next use
  p->auto_decided = false;
  p->dinv_shift = -1.0;
  CHK(build_dinv(p, 1e-1));  // src/PoseGraph.cpp:603
  HIPC(hipStreamSynchronize(p->stream));
  return DPGO_OK;
}

int dpgo_problem_set_Q_csr(dpgo_problem_t p, const int32_t* outer, const int32_t* inner, const double* values) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  if (!outer || !inner || !values) return fail(DPGO_ERR_INVALID, "null CSR arrays");
  const int b = p->b, n = p->n;
  std::vector<int32_t> rowptr(n + 1, 0), colidx;
  std::vector<double> vals;
  std::vector<int> slot(n, -1);  // block column -> position in the current block row
  for (int i = 0; i < n; ++i) {
    const int first = (int)colidx.size();
    std::vector<int> cols;
    for (int rr = 0; rr < b; ++rr) {
      const int row = i * b + rr;
      if (outer[row + 1] < outer[row]) return fail(DPGO_ERR_INVALID, "CSR outer index not monotone");
      for (int t = outer[row]; t < outer[row + 1]; ++t) {
        const int c = inner[t];
        if (c < 0 || c >= n * b) return fail(DPGO_ERR_INVALID, "CSR column index out of range");
        const int j = c / b;
        if (slot[j] < 0) {
          slot[j] = 1;
          cols.push_back(j);
        }
      }
    }
    std::sort(cols.begin(), cols.end());
    for (size_t k = 0; k < cols.size(); ++k) slot[cols[k]] = first + (int)k;
    colidx.insert(colidx.end(), cols.begin(), cols.end());
    vals.resize(colidx.size() * (size_t)b * b, 0.0);
    for (int rr = 0; rr < b; ++rr) {
      const int row = i * b + rr;
      for (int t = outer[row]; t < outer[row + 1]; ++t) {
        const int c = inner[t];
        vals[(size_t)slot[c / b] * b * b + rr * b + (c % b)] += values[t];
      }
    }
    for (int j : cols) slot[j] = -1;
    rowptr[i + 1] = (int32_t)colidx.size();
  }
  return dpgo_problem_set_Q_bsr(p, (int)colidx.size(), rowptr.data(), colidx.data(), vals.data());
}

int dpgo_problem_set_reweightable_edges_ex(dpgo_problem_t p, int m, const int32_t* p1, const int32_t* p2,
                                           const uint8_t* role, const int32_t* slot_in, const double* R,
                                           const double* t, const double* kappa, const double* tau,
                                           const double* weight, const uint8_t* fixed_weight) {
  CHK(check_ready(p));
  if (m < 0 || (m > 0 && (!p1 || !p2 || !R || !t || !kappa || !tau || !weight || !fixed_weight)))
    return fail(DPGO_ERR_INVALID, "null edge arrays");
  if (role && !slot_in) return fail(DPGO_ERR_INVALID, "roles given without neighbour slots");
  const int n = p->n, d = p->d, nnzb = p->Q.nnzb;
  // host copy of the patterns to locate the blocks each edge contributes to
  std::vector<int32_t> rowptr(n + 1), colidx(nnzb);
  HIPC(hipMemcpy(rowptr.data(), p->Q.rowptr, sizeof(int32_t) * (n + 1), hipMemcpyDeviceToHost));
  HIPC(hipMemcpy(colidx.data(), p->Q.colidx, sizeof(int32_t) * nnzb, hipMemcpyDeviceToHost));
  int n_shared = 0;
  for (int e = 0; e < m; ++e)
    if (role && role[e]) ++n_shared;
  std::vector<int32_t> crow, ccol;
  const int cnnz = (n_shared > 0) ? p->C.nnzb : 0;
  if (n_shared > 0) {
    if (!p->C.rowptr) return fail(DPGO_ERR_STATE, "shared re-weightable edges need the G coupling first");
    crow.resize(n + 1);
    ccol.resize(cnnz > 0 ? cnnz : 1);
    HIPC(hipMemcpy(crow.data(), p->C.rowptr, sizeof(int32_t) * (n + 1), hipMemcpyDeviceToHost));
    if (cnnz > 0) HIPC(hipMemcpy(ccol.data(), p->C.colidx, sizeof(int32_t) * cnnz, hipMemcpyDeviceToHost));
  }
  auto find = [](const std::vector<int32_t>& rp, const std::vector<int32_t>& ci, int i, int j) -> int {
    const int32_t* b = ci.data() + rp[i];
    const int32_t* e = ci.data() + rp[i + 1];
    const int32_t* it = std::lower_bound(b, e, (int32_t)j);
    return (it != e && *it == j) ? (int)(it - ci.data()) : -1;
  };
  std::vector<std::vector<std::pair<int, uint8_t>>> lists(nnzb), glists(cnnz);
  std::vector<uint8_t> role_v(m > 0 ? m : 1, 0);
  std::vector<int32_t> slot_v(m > 0 ? m : 1, 0);
  for (int e = 0; e < m; ++e) {
    const int i = p1[e], j = p2[e];
    const int ro = role ? role[e] : 0;
    role_v[e] = (uint8_t)ro;
    if (ro == 0) {
      if (i < 0 || i >= n || j < 0 || j >= n || i == j) return fail(DPGO_ERR_INVALID, "edge endpoint out of range");
      const int sii = find(rowptr, colidx, i, i), sjj = find(rowptr, colidx, j, j), sij = find(rowptr, colidx, i, j),
                sji = find(rowptr, colidx, j, i);
      if (sii < 0 || sjj < 0 || sij < 0 || sji < 0)
        return fail(DPGO_ERR_STATE, "edge does not fit the block pattern of Q");
      lists[sii].push_back({e, 0});
      lists[sjj].push_back({e, 1});
      lists[sij].push_back({e, 2});
      lists[sji].push_back({e, 3});
    } else if (ro == 1 || ro == 2) {
      // outgoing: Q_ii += T Om T^T, C(i, slot) = -T Om; incoming: Q_jj += Om, C(j, slot) = -Om T^T
      // (PoseGraph::constructQ :462-486, constructG :533-562)
      const int mine = (ro == 1) ? i : j;
      const int sl = slot_in[e];
      if (mine < 0 || mine >= n || sl < 0 || sl >= p->C.ncols)
        return fail(DPGO_ERR_INVALID, "shared edge endpoint / neighbour slot out of range");
      slot_v[e] = sl;
      const int sd = find(rowptr, colidx, mine, mine), sc = find(crow, ccol, mine, sl);
      if (sd < 0 || sc < 0) return fail(DPGO_ERR_STATE, "shared edge does not fit the pattern of Q / the G coupling");
      lists[sd].push_back({e, (uint8_t)(ro == 1 ? 0 : 1)});
      glists[sc].push_back({e, (uint8_t)(ro == 1 ? 2 : 3)});
    } else {
      return fail(DPGO_ERR_INVALID, "edge role must be 0, 1 or 2");
    }
  }
  auto flatten = [](const std::vector<std::vector<std::pair<int, uint8_t>>>& L, std::vector<int32_t>& ptr,
                    std::vector<int32_t>& edge, std::vector<uint8_t>& kind) {
    ptr.assign(L.size() + 1, 0);
    for (size_t s = 0; s < L.size(); ++s) {
      for (auto& pr : L[s]) {
        edge.push_back(pr.first);
        kind.push_back(pr.second);
      }
      ptr[s + 1] = (int32_t)edge.size();
    }
  };
  std::vector<int32_t> cptr, cedge, gptr, gedge;
  std::vector<uint8_t> ckind, gkind;
  flatten(lists, cptr, cedge, ckind);
  CHK(free_edges(p));
  p->em = m;
  p->n_shared_edges = n_shared;
  CHK(upload(&p->e_p1, p1, (size_t)m, p->stream));
  CHK(upload(&p->e_p2, p2, (size_t)m, p->stream));
  CHK(upload(&p->e_role, role_v.data(), (size_t)m, p->stream));
  CHK(upload(&p->e_slot, slot_v.data(), (size_t)m, p->stream));
  CHK(upload(&p->e_R, R, (size_t)m * d * d, p->stream));
  CHK(upload(&p->e_t, t, (size_t)m * d, p->stream));
  CHK(upload(&p->e_kappa, kappa, (size_t)m, p->stream));
  CHK(upload(&p->e_tau, tau, (size_t)m, p->stream));
  CHK(upload(&p->e_w, weight, (size_t)m, p->stream));
  CHK(upload(&p->e_fixed, fixed_weight, (size_t)m, p->stream));
  CHK(upload(&p->c_ptr, cptr.data(), cptr.size(), p->stream));
  CHK(upload(&p->c_edge, cedge.data(), cedge.size(), p->stream));
  CHK(upload(&p->c_kind, ckind.data(), ckind.size(), p->stream));
  HIPC(hipMalloc(&p->e_rsq, sizeof(double) * (m > 0 ? m : 1)));
  HIPC(hipMalloc(&p->e_counts, sizeof(int) * 4));
  HIPC(hipMalloc(&p->q_base, sizeof(double) * (size_t)nnzb * p->b * p->b));
  // base = Q(current weights) - sum of the listed edges' contributions at those weights
  CHK(rebuild_Q_from_weights(p, p->Q.vals, -1.0, p->q_base));
  if (n_shared > 0 && cnnz > 0) {
    flatten(glists, gptr, gedge, gkind);
    CHK(upload(&p->g_ptr, gptr.data(), gptr.size(), p->stream));
    CHK(upload(&p->g_edge, gedge.data(), gedge.size(), p->stream));
    CHK(upload(&p->g_kind, gkind.data(), gkind.size(), p->stream));
    HIPC(hipMalloc(&p->c_base, sizeof(double) * (size_t)cnnz * p->b * p->b));
    CHK(rebuild_C_from_weights(p, p->C.vals, -1.0, p->c_base));
  }
  HIPC(hipStreamSynchronize(p->stream));
  return DPGO_OK;
}

int dpgo_problem_set_reweightable_edges(dpgo_problem_t p, int m, const int32_t* p1, const int32_t* p2, const double* R,
                                        const double* t, const double* kappa, const double* tau, const double* weight,
                                        const uint8_t* fixed_weight) {
  return dpgo_problem_set_reweightable_edges_ex(p, m, p1, p2, nullptr, nullptr, R, t, kappa, tau, weight,
                                                fixed_weight);
}

int dpgo_problem_gnc_reweight_device(dpgo_problem_t p, const double* X_dev, const double* nbr_tiles_dev, double mu,
                                     double barc, double w_tol, int update, int counts[3], double* max_rsq) {
  CHK(check_ready(p));
  if (!p->e_w) return fail(DPGO_ERR_STATE, "re-weightable edges not set");
  if (!X_dev) return fail(DPGO_ERR_INVALID, "null X");
  if (p->n_shared_edges > 0 && !nbr_tiles_dev) return fail(DPGO_ERR_INVALID, "shared edges need the neighbour tiles");
  if (update && !(mu > 0.0)) return fail(DPGO_ERR_INVALID, "GNC mu must be positive");
  HIPC(hipMemsetAsync(p->e_counts, 0, sizeof(int) * 4, p->stream));
  const int g = std::max(1, std::min(kMaxGrid, (p->em + kBlock - 1) / kBlock));
  DISPATCH(p->d, p->r, hipLaunchKernelGGL((k_edge_weights<D, R>), dim3(g), dim3(kBlock), 0, p->stream, p->edges(), X_dev,
                                          nbr_tiles_dev, mu, barc, w_tol, update, p->e_counts));
  HIPC(hipGetLastError());
  if (update) CHK(refresh_after_weights(p));
  int h[4] = {0, 0, 0, 0};
  HIPC(hipMemcpyAsync(h, p->e_counts, sizeof(int) * 4, hipMemcpyDeviceToHost, p->stream));
  std::vector<double> rs;
  if (max_rsq) {
    rs.resize(p->em > 0 ? p->em : 1, 0.0);
    if (p->em > 0)
      HIPC(hipMemcpyAsync(rs.data(), p->e_rsq, sizeof(double) * p->em, hipMemcpyDeviceToHost, p->stream));
  }
  HIPC(hipStreamSynchronize(p->stream));
  if (counts) {
    counts[0] = h[0];
    counts[1] = h[1];
    counts[2] = h[2];
  }
  if (max_rsq) {
    double mx = 0.0;
    for (int e = 0; e < p->em; ++e) mx = std::max(mx, rs[e]);
    *max_rsq = mx;
  }
  return DPGO_OK;
}

int dpgo_problem_gnc_reweight(dpgo_problem_t p, const double* X_host, double mu, double barc, double w_tol, int update,
                              int counts[3], double* max_rsq) {
  CHK(check_ready(p));
  if (!X_host) return fail(DPGO_ERR_INVALID, "null X");
  if (p->n_shared_edges > 0) return fail(DPGO_ERR_STATE, "shared edges need the device flavour (neighbour tiles)");
  CHK(h2d(p, p->x2, X_host));
  return dpgo_problem_gnc_reweight_device(p, p->x2, nullptr, mu, barc, w_tol, update, counts, max_rsq);
}

int dpgo_problem_set_edge_weights(dpgo_problem_t p, const double* weight_host) {
  CHK(check_ready(p));
  if (!p->e_w) return fail(DPGO_ERR_STATE, "re-weightable edges not set");
  if (!weight_host) return fail(DPGO_ERR_INVALID, "null weights");
  if (p->em > 0) HIPC(hipMemcpyAsync(p->e_w, weight_host, sizeof(double) * p->em, hipMemcpyHostToDevice, p->stream));
  CHK(refresh_after_weights(p));
  HIPC(hipStreamSynchronize(p->stream));
  return DPGO_OK;
}

int dpgo_problem_get_edge_weights(dpgo_problem_t p, double* weight_host, double* rsq_host) {
  CHK(check_ready(p));
  if (!p->e_w) return fail(DPGO_ERR_STATE, "re-weightable edges not set");
  if (weight_host && p->em > 0)
    HIPC(hipMemcpyAsync(weight_host, p->e_w, sizeof(double) * p->em, hipMemcpyDeviceToHost, p->stream));
  if (rsq_host && p->em > 0)
    HIPC(hipMemcpyAsync(rsq_host, p->e_rsq, sizeof(double) * p->em, hipMemcpyDeviceToHost, p->stream));
  HIPC(hipStreamSynchronize(p->stream));
  return DPGO_OK;
}

int dpgo_problem_update_Q_values(dpgo_problem_t p, const double* vals) {
  CHK(check_ready(p));
  if (!vals) return fail(DPGO_ERR_INVALID, "null vals");
  HIPC(hipMemcpyAsync(p->Q.vals, vals, sizeof(double) * (size_t)p->Q.nnzb * p->b * p->b, hipMemcpyHostToDevice,
                      p->stream));
  // registered re-weightable edges: the constant part of Q is whatever the new values hold beyond the listed
  // edges' contributions at the current weights
  if (p->e_w) CHK(rebuild_Q_from_weights(p, p->Q.vals, -1.0, p->q_base));
  const double s = p->dinv_shift > 0 ? p->dinv_shift : 1e-1;
  p->dinv_shift = -1.0;  // PoseGraph::clearQuadraticMatrix also drops the preconditioner (src/PoseGraph.cpp:352-355)
  p->ml_ready = false;
  p->auto_decided = false;
  p->sym.ready = p->tcg_sym = false;
  CHK(build_dinv(p, s));
  HIPC(hipStreamSynchronize(p->stream));
  return DPGO_OK;
}

int dpgo_problem_get_Q_values(dpgo_problem_t p, double* vals_host) {
  CHK(check_ready(p));
  if (!vals_host) return fail(DPGO_ERR_INVALID, "null vals");
  HIPC(hipMemcpyAsync(vals_host, p->Q.vals, sizeof(double) * (size_t)p->Q.nnzb * p->b * p->b, hipMemcpyDeviceToHost,
                      p->stream));
  HIPC(hipStreamSynchronize(p->stream));
  return DPGO_OK;
}

int dpgo_multilevel_default_ks(int n, int d, int* ks, int* nks) {
  if (n <= 0 || d < 2 || d > 3 || !nks) return fail(DPGO_ERR_INVALID, "bad arguments");
  int split = (n < 40000) ? 4 : 1;
  if (const char* e = std::getenv("DPGO_SPLIT")) {
    const int v = std::atoi(e);
    if (v == 1 || v == 2 || v == 4) split = v;
  }
  const std::vector<int> v = ml_default_ks(n, d + 1, split);
  if (ks)
    for (size_t l = 0; l < v.size() && (int)l < *nks; ++l) ks[l] = v[l];
  *nks = (int)v.size();
  return DPGO_OK;
}

int dpgo_multilevel_graph_aggregates(int n, const int32_t* rowptr, const int32_t* colidx, int max_size, int32_t* label,
                                     int32_t* parent, int* n_aggregates) {
  if (n <= 0 || !rowptr || !colidx || max_size < 2 || !label) return fail(DPGO_ERR_INVALID, "bad arguments");
  const std::vector<int32_t> rp(rowptr, rowptr + n + 1), ci(colidx, colidx + rowptr[n]);
  for (int32_t c : ci)
    if (c < 0 || c >= n) return fail(DPGO_ERR_INVALID, "block column out of range");
  std::vector<int32_t> lab, ptr, mem, par, pslot;
  const int na = ml_graph_aggregates(rp, ci, n, max_size, lab, ptr, mem, par, pslot);
  std::copy(lab.begin(), lab.end(), label);
  if (parent) std::copy(par.begin(), par.end(), parent);
  if (n_aggregates) *n_aggregates = na;
  return DPGO_OK;
}

int dpgo_multilevel_merged_aggregates(int n, const int32_t* rowptr, const int32_t* colidx, int max_size, int merge_cap,
                                      int32_t* label, int32_t* parent, int* n_aggregates) {
  if (n <= 0 || !rowptr || !colidx || max_size < 2 || merge_cap < max_size || !label) return fail(DPGO_ERR_INVALID, "bad arguments");
  const std::vector<int32_t> rp(rowptr, rowptr + n + 1), ci(colidx, colidx + rowptr[n]);
  for (int32_t c : ci)
    if (c < 0 || c >= n) return fail(DPGO_ERR_INVALID, "block column out of range");
  std::vector<int32_t> lab, ptr, mem, par, pslot;
  ml_graph_aggregates(rp, ci, n, max_size, lab, ptr, mem, par, pslot);
  const int na = ml_merge_small_aggregates(rp, ci, n, max_size, merge_cap, lab, ptr, mem, par, pslot);
  std::copy(lab.begin(), lab.end(), label);
  if (parent) std::copy(par.begin(), par.end(), parent);
  if (n_aggregates) *n_aggregates = na;
  return DPGO_OK;
}

int dpgo_problem_additive_plan(dpgo_problem_t p, int* lane_groups, int* tile, int* growth, int* merge_cap, int* aggregates,
                               int* graph) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  if ((int)p->h_rowptr.size() != p->n + 1) return fail(DPGO_ERR_STATE, "Q's block pattern is not set");
  const auto& plan = additive_plan(p);
  if (lane_groups) *lane_groups = plan.split;
  if (tile) *tile = plan.tile;
  if (growth) *growth = plan.S;
  if (merge_cap) *merge_cap = plan.cap;
  if (aggregates) *aggregates = plan.na;
  if (graph) *graph = plan.graph ? 1 : 0;
  return DPGO_OK;
}

int dpgo_problem_setup_multilevel(dpgo_problem_t p, int nks, const int* ks, double omega, double shift) {
  CHK(check_ready(p));
  if (nks < 0 || nks > 8 || (nks > 0 && !ks) || !(omega > 0.0) || !(shift >= 0.0))
    return fail(DPGO_ERR_INVALID, "bad multilevel arguments");
  std::vector<int> v = nks > 0 ? std::vector<int>(ks, ks + nks) : ml_default_ks(p->n, p->b, p->split);
  const bool same = p->ml_symbolic && ml_current_ks(p) == v;
  if (!same) {
    // graph aggregates with merged fragments that fit a workgroup tile of the one-launch solve also get that layout's
    // (aggregate, slot) table, so that an explicit hierarchy of this shape serves precond = additive as well
    int perm_tile = 0;
    if (v.size() == 2 && v[0] < 0 && v[1] < 0 && p->split == 4)
      perm_tile = -v[1] <= ml_tile(p->b, 4) ? ml_tile(p->b, 4) : (-v[1] <= ml_tile(p->b, 1) ? ml_tile(p->b, 1) : 0);
    CHK(ml_symbolic_setup(p, v, perm_tile));
  }
  p->ml_user_ks = nks > 0;
  p->ml_additive_layout = false;
  p->ml_omega = omega;
  p->ml_shift = shift;
  CHK(ml_numeric_setup(p));
  HIPC(hipStreamSynchronize(p->stream));
  return DPGO_OK;
}

int dpgo_problem_multilevel_path(dpgo_problem_t p, int* flags) {
  if (!p || !flags) return fail(DPGO_ERR_INVALID, "null handle / pointer");
  if (!p->ml_symbolic) return fail(DPGO_ERR_STATE, "multilevel hierarchy not set up");
  *flags = (p->ml_use_ap() ? DPGO_ML_PATH_AP : 0) | (p->ml_use_dense_sym() ? DPGO_ML_PATH_PACKED_DENSE : 0);
  return DPGO_OK;
}

int dpgo_problem_multilevel_coarse_bits(dpgo_problem_t p, int* bits) {
  if (!p || !bits) return fail(DPGO_ERR_INVALID, "null handle / pointer");
  if (*bits < 0) {
    *bits = p->ml_coarse_bits;
    return DPGO_OK;
  }
  if (*bits != 32 && *bits != 64) return fail(DPGO_ERR_INVALID, "the coarsest inverse is stored in 32 or 64 bits");
  if (*bits != p->ml_coarse_bits) {
    p->ml_coarse_bits = *bits;
    p->ml_ready = false;  // the stored inverse is rebuilt at the next use
  }
  return DPGO_OK;
}

int dpgo_problem_multilevel_info(dpgo_problem_t p, int* nlevels, int* sizes, int* ks, int* nnzb) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  if (!p->ml_symbolic) return fail(DPGO_ERR_STATE, "multilevel hierarchy not set up");
  const int cap = nlevels ? *nlevels : 0;
  for (int l = 0; l < (int)p->ml.size() && l < cap; ++l) {
    if (sizes) sizes[l] = p->ml[l].n;
    if (ks) ks[l] = p->ml[l].graph ? -p->ml[l].k : p->ml[l].k;  // negative: graph aggregates of at most that many poses
    // (graph aggregates whose fragments were merged: the LAST level's entry, otherwise 0, carries -merge bound)
    if (ks && l > 0 && l + 1 == (int)p->ml.size() && p->ml[0].graph && p->ml[0].merge_cap) ks[l] = -p->ml[0].merge_cap;
    if (nnzb) nnzb[l] = (l == 0) ? p->Q.nnzb : p->ml[l].A.nnzb;
  }
  if (nlevels) *nlevels = (int)p->ml.size();
  return DPGO_OK;
}

int dpgo_problem_multilevel_get(dpgo_problem_t p, int level, int what, void* out_host) {
  CHK(check_ready(p));
  if (!p->ml_ready) return fail(DPGO_ERR_STATE, "multilevel hierarchy not built");
  if (!out_host || level < 0 || level >= (int)p->ml.size()) return fail(DPGO_ERR_INVALID, "bad level / null pointer");
  auto& L = p->ml[level];
  const int bb = p->b * p->b;
  const void* src = nullptr;
  size_t bytes = 0;
  switch (what) {
    case DPGO_ML_P_BLOCKS:
      src = L.Pb, bytes = sizeof(double) * (size_t)L.n * bb;
      break;
    case DPGO_ML_A_ROWPTR:
      src = L.A.rowptr, bytes = sizeof(int32_t) * ((size_t)L.n + 1);
      break;
    case DPGO_ML_A_COLIDX:
      src = L.A.colidx, bytes = sizeof(int32_t) * (size_t)L.A.nnzb;
      break;
    case DPGO_ML_A_VALUES:
      src = L.A.vals, bytes = sizeof(double) * (size_t)L.A.nnzb * bb;
      break;
    case DPGO_ML_AGG_LABELS:
      src = L.graph ? L.lab : nullptr, bytes = sizeof(int32_t) * (size_t)L.n;
      break;
    case DPGO_ML_AP_NNZB: {
      if (!L.AP.vals) return fail(DPGO_ERR_INVALID, "this level does not hold that item");
      *static_cast<int32_t*>(out_host) = L.AP.nnzb;
      return DPGO_OK;
    }
    case DPGO_ML_RESTRICT_PARTIALS: {
      if (!L.graph) return fail(DPGO_ERR_INVALID, "this level does not hold that item");
      *static_cast<int32_t*>(out_host) = L.nseg;
      return DPGO_OK;
    }
    case DPGO_ML_DENSE_INVERSE: {
      if (level + 1 != (int)p->ml.size()) return fail(DPGO_ERR_INVALID, "the dense inverse belongs to the last level");
      const int N = L.n * p->b;  // the N x N corner of the padded lda x lda array
      HIPC(hipMemcpy2DAsync(out_host, sizeof(double) * N, p->ml_dense, sizeof(double) * p->ml_lda, sizeof(double) * N, N,
                            hipMemcpyDeviceToHost, p->stream));
      HIPC(hipStreamSynchronize(p->stream));
      return DPGO_OK;
    }
    default:
      return fail(DPGO_ERR_INVALID, "unknown item");
  }
  if (!src) return fail(DPGO_ERR_INVALID, "this level does not hold that item");
  HIPC(hipMemcpyAsync(out_host, src, bytes, hipMemcpyDeviceToHost, p->stream));
  HIPC(hipStreamSynchronize(p->stream));
  return DPGO_OK;
}

int dpgo_problem_auto_state(dpgo_problem_t p, int* use_multilevel) {
  if (!p || !use_multilevel) return fail(DPGO_ERR_INVALID, "null pointer");
  if (*use_multilevel >= 0) {
    p->auto_ml = *use_multilevel != 0;
    p->auto_decided = true;
    p->auto_cost = dpgo_problem_s::AutoCost();  // (a choice made from outside is followed by the budget hysteresis)
  } else {
    if (*use_multilevel == -2) p->auto_decided = false;  // back to the decision a fresh handle takes for this problem
    p->auto_decide();
  }
  *use_multilevel = p->auto_ml ? 1 : 0;
  return DPGO_OK;
}

int dpgo_problem_auto_info(dpgo_problem_t p, int* state, long long* jacobi_units, int* reference_products, int* switches,
                           int* backoff, int* units_jacobi, int* units_additive) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  const auto& a = p->auto_cost;
  if (units_jacobi) *units_jacobi = a.uj;
  if (units_additive) *units_additive = a.ua;
  if (state) *state = a.state;
  if (jacobi_units) *jacobi_units = a.jac_units;
  if (reference_products) *reference_products = a.ref;
  if (switches) *switches = a.switches;
  if (backoff) *backoff = a.backoff;
  return DPGO_OK;
}

int dpgo_auto_rule_constants(int* units_jacobi, int* units_additive, int* setup_units, int* min_products) {
  if (units_jacobi) *units_jacobi = kAutoUnitsJacobi;
  if (units_additive) *units_additive = kAutoUnitsAdditive;
  if (setup_units) *setup_units = kAutoSetupUnits;
  if (min_products) *min_products = kAutoMinProducts;
  return DPGO_OK;
}

int dpgo_dense_spd_inverse(int N, const double* A_host, double* Ainv_host, int device, int use_mfma) {
  if (N <= 0 || N > 16384 || !A_host || !Ainv_host) return fail(DPGO_ERR_INVALID, "bad arguments");
  int cnt = 0;
  CHK(dpgo_device_count(&cnt));
  if (cnt <= 0) return fail(DPGO_ERR_HIP, "no HIP device (this library has no CPU fallback)");
  if (device < 0 || device >= cnt) return fail(DPGO_ERR_INVALID, "device index out of range");
  HIPC(hipSetDevice(device));
  const int lda = ((N + kNB - 1) / kNB) * kNB;
  TmpDev tmp;
  double *M = nullptr, *W = nullptr, *Rx = nullptr;
  CHK(tmp.alloc(&M, sizeof(double) * (size_t)lda * lda));
  CHK(tmp.alloc(&W, sizeof(double) * (size_t)lda * kNB));
  CHK(tmp.alloc(&Rx, sizeof(double) * (size_t)lda * kNB));
  HIPC(hipMemset(M, 0, sizeof(double) * (size_t)lda * lda));
  HIPC(hipMemcpy2D(M, sizeof(double) * lda, A_host, sizeof(double) * N, sizeof(double) * N, N, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_dense_pad_identity, dim3(1), dim3(kBlock), 0, (hipStream_t) nullptr, M, lda, N);
  CHK(dense_spd_inverse(nullptr, M, lda, W, Rx, use_mfma != 0));
  HIPC(hipMemcpy2D(Ainv_host, sizeof(double) * N, M, sizeof(double) * lda, sizeof(double) * N, N, hipMemcpyDeviceToHost));
  return DPGO_OK;
}

int dpgo_problem_set_G(dpgo_problem_t p, const double* G_host) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  CHK(set_device(p));
  if (!G_host) {
    p->has_G = false;
    return DPGO_OK;
  }
  CHK(h2d(p, p->G, G_host));
  HIPC(hipStreamSynchronize(p->stream));
  p->has_G = true;
  return DPGO_OK;
}

int dpgo_problem_set_G_device(dpgo_problem_t p, const double* G_dev) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  CHK(set_device(p));
  if (!G_dev) {
    p->has_G = false;
    return DPGO_OK;
  }
  HIPC(hipMemcpyAsync(p->G, G_dev, p->vec_bytes(), hipMemcpyDeviceToDevice, p->stream));
  p->has_G = true;
  return DPGO_OK;
}

int dpgo_problem_set_G_coupling(dpgo_problem_t p, int ncols, int nnzb, const int32_t* rowptr, const int32_t* colidx,
                                const double* vals, const double* G0_host) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  if (ncols < 0) return fail(DPGO_ERR_INVALID, "ncols < 0");
  if (nnzb > 0 && !vals) return fail(DPGO_ERR_INVALID, "null vals");
  CHK(validate_bsr(p->n, ncols, nnzb, rowptr, colidx, false));
  CHK(set_device(p));
  // shared re-weightable edges index into the old coupling pattern: drop them (the caller re-registers)
  if (p->n_shared_edges > 0) CHK(free_edges(p));
  CHK(upload_bsr(p->C, p->n, ncols, nnzb, p->b, rowptr, colidx, vals, p->stream));
  if (G0_host) {
    CHK(h2d(p, p->G0, G0_host));
  } else {
    HIPC(hipMemsetAsync(p->G0, 0, p->vec_bytes(), p->stream));
  }
  HIPC(hipStreamSynchronize(p->stream));
  return DPGO_OK;
}

int dpgo_problem_update_G_from_neighbors_device(dpgo_problem_t p, const double* nbr_tiles_dev) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  if (!p->C.rowptr) return fail(DPGO_ERR_STATE, "G coupling not set");
  if (!nbr_tiles_dev && p->C.nnzb > 0) return fail(DPGO_ERR_INVALID, "null neighbour tiles");
  CHK(set_device(p));
  CHK(launch_spmm(p, p->C, nbr_tiles_dev, p->G0, p->G));
  p->has_G = true;
  return DPGO_OK;
}

// ---- QuadraticProblem methods (host pointers) ----
static int eval_common(dpgo_problem_t p, const double* X) {
  CHK(check_ready(p));
  if (!X) return fail(DPGO_ERR_INVALID, "null X");
  CHK(h2d(p, p->x2, X));
  return DPGO_OK;
}

int dpgo_problem_f(dpgo_problem_t p, const double* X, double* f) {
  if (!f) return fail(DPGO_ERR_INVALID, "null out");
  CHK(eval_common(p, X));
  CHK(launch_grad(p, p->x2, nullptr, nullptr, nullptr));
  CHK(launch_rtr_begin(p, 0.0, 1.0, 1.0, 0, 0));
  CHK(poll_state(p));
  *f = p->hstate->f1;
  return DPGO_OK;
}

int dpgo_problem_euc_grad(dpgo_problem_t p, const double* X, double* EG) {
  if (!EG) return fail(DPGO_ERR_INVALID, "null out");
  CHK(eval_common(p, X));
  CHK(launch_spmm(p, p->Q, p->x2, p->has_G ? p->G : nullptr, p->g2));
  return d2h(p, EG, p->g2);
}

int dpgo_problem_euc_hess(dpgo_problem_t p, const double* V, double* HV) {
  if (!HV) return fail(DPGO_ERR_INVALID, "null out");
  CHK(eval_common(p, V));
  CHK(launch_spmm(p, p->Q, p->x2, nullptr, p->g2));
  return d2h(p, HV, p->g2);
}

int dpgo_problem_rie_grad(dpgo_problem_t p, const double* X, double* RG) {
  if (!RG) return fail(DPGO_ERR_INVALID, "null out");
  CHK(eval_common(p, X));
  CHK(launch_grad(p, p->x2, p->g2, nullptr, nullptr));
  return d2h(p, RG, p->g2);
}

int dpgo_problem_rie_grad_norm(dpgo_problem_t p, const double* X, double* gn) {
  if (!gn) return fail(DPGO_ERR_INVALID, "null out");
  CHK(eval_common(p, X));
  CHK(launch_grad(p, p->x2, nullptr, nullptr, nullptr));
  CHK(launch_rtr_begin(p, 0.0, 1.0, 1.0, 0, 0));
  CHK(poll_state(p));
  *gn = p->hstate->ngf;
  return DPGO_OK;
}

int dpgo_problem_rie_hess(dpgo_problem_t p, const double* X, const double* V, double* HV) {
  if (!HV || !V) return fail(DPGO_ERR_INVALID, "null pointer");
  CHK(eval_common(p, X));
  CHK(h2d(p, p->eta, V));
  CHK(launch_grad(p, p->x2, nullptr, p->S2, nullptr));
  CHK(launch_hess(p, p->x2, p->S2, p->eta, nullptr, p->g2, p->pH(), nullptr, 0));
  return d2h(p, HV, p->g2);
}

int dpgo_problem_precondition(dpgo_problem_t p, int precond, double shift, const double* X, const double* V,
                              double* Z) {
  if (!Z || !V) return fail(DPGO_ERR_INVALID, "null pointer");
  CHK(eval_common(p, X));
  CHK(h2d(p, p->eta, V));
  const double* dinv = nullptr;
  if (precond == DPGO_PRECOND_AUTO) {
    p->auto_decide();
    precond = p->auto_ml ? DPGO_PRECOND_MULTILEVEL : DPGO_PRECOND_BLOCK_JACOBI;
  }
  if (precond == DPGO_PRECOND_BLOCK_JACOBI) {
    CHK(build_dinv(p, shift));
    dinv = p->dinv;
  } else if (precond == DPGO_PRECOND_MULTILEVEL) {
    CHK(ml_ensure(p, shift));
    CHK(launch_ml_apply(p, p->x2, p->eta, p->g2));
    return d2h(p, Z, p->g2);
  } else if (precond == DPGO_PRECOND_ADDITIVE) {
    return fail(DPGO_ERR_UNSUPPORTED, "the additive preconditioner exists inside the persistent tCG kernel only");
  } else if (precond != DPGO_PRECOND_NONE) {
    return fail(DPGO_ERR_INVALID, "unknown preconditioner");
  }
  CHK(launch_precond(p, p->x2, p->eta, dinv, p->g2));
  return d2h(p, Z, p->g2);
}

int dpgo_optimize(dpgo_problem_t p, const dpgo_ropt_params* params, const double* X0, double* Xopt,
                  dpgo_ropt_result* result) {
  CHK(check_ready(p));
  if (!params || !X0 || !Xopt || !result) return fail(DPGO_ERR_INVALID, "null pointer");
  CHK(h2d(p, p->x1, X0));
  CHK(run_optimize(p, params, result));
  return d2h(p, Xopt, p->x1);
}

int dpgo_optimize_device(dpgo_problem_t p, const dpgo_ropt_params* params, double* X_dev, dpgo_ropt_result* result) {
  CHK(check_ready(p));
  if (!params || !X_dev || !result) return fail(DPGO_ERR_INVALID, "null pointer");
  // the caller's buffer IS the iterate for the duration of the call (no copies in or out): accepted steps are written
  // into it by k_rtr_update, rejected ones leave it untouched
  double* own = p->x1;
  p->x1 = X_dev;
  const int rc = run_optimize(p, params, result);
  p->x1 = own;
  if (rc != DPGO_OK) return rc;
  HIPC(hipStreamSynchronize(p->stream));
  return DPGO_OK;
}

int dpgo_warning_count(void) { return g_warnings.load(); }

int dpgo_optimize_device_begin(dpgo_problem_t p, const dpgo_ropt_params* params, double* X_dev,
                               const double* nbr_tiles_dev) {
  CHK(check_ready(p));
  if (!params || !X_dev) return fail(DPGO_ERR_INVALID, "null pointer");
  if (p->pending.active) return fail(DPGO_ERR_STATE, "a solve of this handle is already in flight (dpgo_optimize_device_end)");
  if (nbr_tiles_dev) {  // PGOAgent::updateX: G from the neighbours' public poses first (same stream)
    if (!p->C.rowptr) return fail(DPGO_ERR_STATE, "G coupling not set");
    CHK(launch_spmm(p, p->C, nbr_tiles_dev, p->G0, p->G));
    p->has_G = true;
  }
  auto& pd = p->pending;
  pd = dpgo_problem_s::Pending();
  pd.own_x1 = p->x1;
  p->x1 = X_dev;
  p->persist_stream_ordered = true;
  dpgo_ropt_result tmp;
  const int rc = run_optimize(p, params, &tmp, RUN_BEGIN);
  p->persist_stream_ordered = false;
  if (rc != DPGO_OK || !pd.launched) {  // failed, or the solve is not a one-launch solve and has run to completion
    p->x1 = pd.own_x1;
    if (rc != DPGO_OK) return rc;
    pd.result = tmp;
  }
  pd.active = true;
  return DPGO_OK;
}

int dpgo_optimize_device_end(dpgo_problem_t p, dpgo_ropt_result* result) {
  if (!p || !result) return fail(DPGO_ERR_INVALID, "null pointer");
  CHK(set_device(p));
  auto& pd = p->pending;
  if (!pd.active) return fail(DPGO_ERR_STATE, "no solve in flight (dpgo_optimize_device_begin)");
  pd.active = false;
  if (!pd.launched) {
    *result = pd.result;
    return DPGO_OK;
  }
  pd.launched = false;
  const int rc = run_optimize(p, &pd.resolved, result, RUN_END);
  p->x1 = pd.own_x1;
  if (rc != DPGO_OK) return rc;
  HIPC(hipStreamSynchronize(p->stream));
  return DPGO_OK;
}

// ---- several agents of one process updated concurrently (same-colour agents of a parallel RBCD sweep) ----
namespace {
// Host threads that feed the just-in-time tCG loops of several handles at once: one worker per concurrently solved handle,
// created on first use and kept (a solve is a few milliseconds; creating threads per sweep would show).
class FeedPool {
 public:
  static FeedPool& get() {
    static FeedPool* pool = new FeedPool();  // never destroyed: workers may outlive static destructors
    return *pool;
  }
  // runs job(0..count-1): job(0) on the calling thread, the others on workers; returns when all are done
  void run(int count, const std::function<void(int)>& job) {
    std::unique_lock<std::mutex> call(call_mu_);  // one batch at a time
    {
      std::lock_guard<std::mutex> lk(mu_);
      while ((int)workers_.size() < count - 1) {
        const int id = (int)workers_.size();
        workers_.emplace_back([this, id] { loop(id); });
        workers_.back().detach();
      }
      job_ = &job;
      count_ = count;
      pending_ = count - 1;
      epoch_ += 1;
    }
    cv_.notify_all();
    job(0);
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [this] { return pending_ == 0; });
    job_ = nullptr;
  }

 private:
  void loop(int id) {
    unsigned long long seen = 0;
    while (true) {
      const std::function<void(int)>* job = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return epoch_ != seen; });
        seen = epoch_;
        if (id + 1 < count_) job = job_;
      }
      if (job) {
        (*job)(id + 1);
        std::lock_guard<std::mutex> lk(mu_);
        if (--pending_ == 0) done_.notify_all();
      }
    }
  }
  std::mutex call_mu_, mu_;
  std::condition_variable cv_, done_;
  std::vector<std::thread> workers_;
  const std::function<void(int)>* job_ = nullptr;
  int count_ = 0, pending_ = 0;
  unsigned long long epoch_ = 0;
};

// Runs body(k) for every handle on its OWN stream, ordered after `after_stream`; the handles' previous streams are
// restored afterwards.  Returns the first failure (its message becomes this thread's dpgo_last_error).
static int run_many(int count, const dpgo_problem_t* handles, void* after_stream, const std::function<int(int)>& body) {
  if (count <= 0) return DPGO_OK;
  if (!handles) return fail(DPGO_ERR_INVALID, "null handle array");
  for (int k = 0; k < count; ++k) {
    CHK(check_ready(handles[k]));
    if (handles[k]->device != handles[0]->device) return fail(DPGO_ERR_INVALID, "handles on different devices");
    for (int q = 0; q < k; ++q)
      if (handles[q] == handles[k]) return fail(DPGO_ERR_INVALID, "a handle appears twice");
  }
  {  // ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): more concurrently solved handles than
     // that still complete, but queue behind each other -- say so once instead of silently serialising
    static std::atomic<bool> warned{false};
    const char* e = std::getenv("GPU_MAX_HW_QUEUES");
    const int queues = (e && std::atoi(e) > 0) ? std::atoi(e) : 4;
    if (count > queues && !warned.exchange(true)) {
      g_warnings.fetch_add(1);
      std::fprintf(stderr,
                   "dpgo_hip: warning: %d handles are solved concurrently but GPU_MAX_HW_QUEUES is %d: their streams share %d "
                   "hardware queues and partly serialise; set GPU_MAX_HW_QUEUES >= %d in the environment before the HIP runtime "
                   "initialises (bench.py does)\n",
                   count, queues, queues, count);
    }
  }
  std::vector<hipStream_t> prev(count);
  for (int k = 0; k < count; ++k) prev[k] = handles[k]->stream;  // (all of them first: restored below whatever fails)
  hipEvent_t ev = nullptr;
  HIPC(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t e = hipEventRecord(ev, (hipStream_t)after_stream);
  for (int k = 0; k < count && e == hipSuccess; ++k) {
    if (prev[k] != handles[k]->own_stream) e = hipStreamSynchronize(prev[k]);  // earlier work of the handle itself
    handles[k]->stream = handles[k]->own_stream;
    handles[k]->persist_share = count;  // persistent tCG launches: prefer the layout with the fewest resident slots
    if (e == hipSuccess) e = hipStreamWaitEvent(handles[k]->own_stream, ev, 0);
  }
  std::vector<int> rc(count, DPGO_OK);
  std::vector<std::string> msg(count);
  if (e == hipSuccess) {
    FeedPool::get().run(count, [&](int k) {
      int r = (hipSetDevice(handles[k]->device) == hipSuccess) ? body(k) : fail(DPGO_ERR_HIP, "hipSetDevice failed");
      if (r == DPGO_OK && hipStreamSynchronize(handles[k]->stream) != hipSuccess)
        r = fail(DPGO_ERR_HIP, "hipStreamSynchronize failed");
      rc[k] = r;
      if (r != DPGO_OK) msg[k] = g_err;  // thread-local message of the worker
    });
  }
  for (int k = 0; k < count; ++k) {
    handles[k]->stream = prev[k];
    handles[k]->persist_share = 1;
  }
  (void)hipEventDestroy(ev);
  if (e != hipSuccess) return fail(DPGO_ERR_HIP, std::string("stream ordering of the concurrent update: ") + hipGetErrorString(e));
  for (int k = 0; k < count; ++k)
    if (rc[k] != DPGO_OK) return fail(rc[k], "handle " + std::to_string(k) + ": " + msg[k]);
  return DPGO_OK;
}
}  // namespace

int dpgo_optimize_device_many(int count, const dpgo_problem_t* handles, const dpgo_ropt_params* params,
                              double* const* X_dev, const double* const* nbr_tiles_dev, void* after_stream,
                              dpgo_ropt_result* results) {
  if (count <= 0) return DPGO_OK;
  if (!params || !X_dev || !results) return fail(DPGO_ERR_INVALID, "null pointer");
  for (int k = 0; k < count; ++k)
    if (!X_dev[k]) return fail(DPGO_ERR_INVALID, "null iterate");
  return run_many(count, handles, after_stream, [&](int k) -> int {
    dpgo_problem_s* p = handles[k];
    if (nbr_tiles_dev && nbr_tiles_dev[k]) {  // PGOAgent::updateX: G from the neighbours' public poses first
      if (!p->C.rowptr) return fail(DPGO_ERR_STATE, "G coupling not set");
      CHK(launch_spmm(p, p->C, nbr_tiles_dev[k], p->G0, p->G));
      p->has_G = true;
    }
    double* own = p->x1;
    p->x1 = X_dev[k];
    const int rc = run_optimize(p, params, &results[k]);
    p->x1 = own;
    return rc;
  });
}

int dpgo_problem_eval_terms_device_many(int count, const dpgo_problem_t* handles, const double* const* X_dev,
                                        const double* const* nbr_tiles_dev, void* after_stream, double* terms) {
  if (count <= 0) return DPGO_OK;
  if (!X_dev || !terms) return fail(DPGO_ERR_INVALID, "null pointer");
  for (int k = 0; k < count; ++k)
    if (!X_dev[k]) return fail(DPGO_ERR_INVALID, "null iterate");
  return run_many(count, handles, after_stream, [&](int k) -> int {
    dpgo_problem_s* p = handles[k];
    if (nbr_tiles_dev && nbr_tiles_dev[k]) {
      if (!p->C.rowptr) return fail(DPGO_ERR_STATE, "G coupling not set");
      CHK(launch_spmm(p, p->C, nbr_tiles_dev[k], p->G0, p->G));
      p->has_G = true;
    }
    CHK(launch_grad(p, X_dev[k], nullptr, nullptr, nullptr));
    CHK(launch_rtr_begin(p, 0.0, 1.0, 1.0, 0, 0));
    CHK(poll_state(p));
    terms[3 * k + 0] = p->hstate->xqx;
    terms[3 * k + 1] = p->hstate->xg;
    terms[3 * k + 2] = p->hstate->ngf * p->hstate->ngf;
    return DPGO_OK;
  });
}

int dpgo_spmm_device(dpgo_problem_t p, const double* V_dev, double* OUT_dev, int add_G) {
  CHK(check_ready(p));
  if (!V_dev || !OUT_dev) return fail(DPGO_ERR_INVALID, "null pointer");
  return launch_spmm(p, p->Q, V_dev, (add_G && p->has_G) ? p->G : nullptr, OUT_dev);
}

int dpgo_problem_eval_device(dpgo_problem_t p, const double* X_dev, double* f, double* gradnorm) {
  CHK(check_ready(p));
  if (!X_dev) return fail(DPGO_ERR_INVALID, "null pointer");
  CHK(launch_grad(p, X_dev, nullptr, nullptr, nullptr));
  CHK(launch_rtr_begin(p, 0.0, 1.0, 1.0, 0, 0));
  CHK(poll_state(p));
  if (f) *f = p->hstate->f1;
  if (gradnorm) *gradnorm = p->hstate->ngf;
  return DPGO_OK;
}

int dpgo_problem_eval_terms_device(dpgo_problem_t p, const double* X_dev, double* xqx, double* xg, double* g2) {
  CHK(check_ready(p));
  if (!X_dev) return fail(DPGO_ERR_INVALID, "null pointer");
  CHK(launch_grad(p, X_dev, nullptr, nullptr, nullptr));
  CHK(launch_rtr_begin(p, 0.0, 1.0, 1.0, 0, 0));
  CHK(poll_state(p));
  if (xqx) *xqx = p->hstate->xqx;
  if (xg) *xg = p->hstate->xg;
  if (g2) *g2 = p->hstate->ngf * p->hstate->ngf;
  return DPGO_OK;
}

#ifdef DPGO_TIMELINE
int dpgo_debug_timeline(long long* out /* [2][16] */) {
  HIPC(hipDeviceSynchronize());
  HIPC(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_timeline), sizeof(long long) * 32));
  return DPGO_OK;
}
#endif

int dpgo_problem_persistent_info(dpgo_problem_t p, int* enabled, int* workgroups, int* last_members, int* last_iterations,
                                 int* last_layout) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  if (enabled) *enabled = (p->persist && !p->persist_failed_once) ? 1 : 0;
  if (workgroups) *workgroups = p->persist_wgs;
  if (last_members) *last_members = p->hctrl ? (int)p->hctrl->members : 0;
  if (last_iterations) *last_iterations = p->hctrl ? (int)p->hctrl->iters : 0;
  if (last_layout) *last_layout = (p->hctrl && p->hctrl->members) ? p->persist_split * 16 + p->persist_mt : 0;
  return DPGO_OK;
}

int dpgo_problem_persistent_phases(dpgo_problem_t p, double us_per_iteration[4], int* iterations) {
  if (!p || !us_per_iteration) return fail(DPGO_ERR_INVALID, "null handle / pointer");
  const double it = p->hctrl ? (double)p->hctrl->ticks[4] : 0.0;
  for (int k = 0; k < 4; ++k)  // (100 MHz wall-clock ticks of participant 0, summed over the iterations after the first)
    us_per_iteration[k] = (p->hctrl && it > 0.0 && p->hctrl->members) ? 0.01 * (double)p->hctrl->ticks[k] / it : 0.0;
  if (iterations) *iterations = (p->hctrl && p->hctrl->members) ? (int)p->hctrl->iters : 0;
  return DPGO_OK;
}

int dpgo_problem_set_persistent(dpgo_problem_t p, int enable) {
  if (!p) return fail(DPGO_ERR_INVALID, "null handle");
  if (enable && persist_geometry(p, persist_capacity(p->device)).wgs <= 0)
    return fail(DPGO_ERR_UNSUPPORTED, "block too large for the persistent tCG kernel (at most 2 tiles on each of 256 workgroups)");
  p->persist = enable != 0;
  if (enable) p->persist_failed_once = false;
  return DPGO_OK;
}

int dpgo_bench_spmm(dpgo_problem_t p, int reps, int warmup, double* avg_ms) {
  CHK(check_ready(p));
  if (reps <= 0 || !avg_ms) return fail(DPGO_ERR_INVALID, "bad arguments");
  hipEvent_t e0, e1;
  HIPC(hipEventCreate(&e0));
  HIPC(hipEventCreate(&e1));
  for (int i = 0; i < warmup; ++i) CHK(launch_spmm(p, p->Q, p->x1, nullptr, p->x2));
  HIPC(hipEventRecord(e0, p->stream));
  for (int i = 0; i < reps; ++i) CHK(launch_spmm(p, p->Q, p->x1, nullptr, p->x2));
  HIPC(hipEventRecord(e1, p->stream));
  HIPC(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPC(hipEventElapsedTime(&ms, e0, e1));
  HIPC(hipEventDestroy(e0));
  HIPC(hipEventDestroy(e1));
  *avg_ms = (double)ms / reps;
  return DPGO_OK;
}

int dpgo_bench_hess_rotating(dpgo_problem_t p, int nsets, int reps, int warmup, double* avg_ms) {
  CHK(check_ready(p));
  if (nsets < 1 || nsets > 512 || reps <= 0 || !avg_ms) return fail(DPGO_ERR_INVALID, "bad arguments");
  std::memset(p->hstate, 0, sizeof(DevState));  // as dpgo_bench_hess: a state without early exits
  p->hstate->z_r = 1.0;
  p->hstate->theta = 1.0;
  p->hstate->kappa = -1.0;
  p->hstate->max_inner = 1 << 30;
  p->hstate->min_inner = 1 << 30;  // the convergence test is never evaluated, whatever the partial sums hold
  CHK(push_state(p));
  // every operand of the tCG-step kernel gets nsets private copies; the handle's pointers are swapped per launch
  const size_t vbytes = sizeof(double) * (size_t)p->Q.nnzb * p->b * p->b;
  const size_t cbytes = sizeof(int32_t) * (size_t)p->Q.nnzb;
  const size_t sbytes = sizeof(double) * (size_t)p->n * p->d * p->d;
  CHK(resolve_tcg_storage(p));
  const bool symq = p->tcg_sym;  // the kernel reads the symmetric copy: that is what rotates
  auto& SY = p->sym;
  struct Set {
    double *vals = nullptr, *x1 = nullptr, *S1 = nullptr, *z = nullptr, *delta = nullptr, *Hd = nullptr;
    int32_t* colidx = nullptr;
    double* uv = nullptr;
    int32_t *uc = nullptr, *lc = nullptr, *ls = nullptr;
  };
  std::vector<Set> sets(nsets);
  const Set orig{p->Q.vals, p->x1, p->S1, p->z, p->delta, p->Hd, p->Q.colidx, SY.uvalsT, SY.ucol, SY.lcol, SY.lslot};
  bool ok = true;
  auto dup = [&](auto** dst, const void* src, size_t bytes) {
    if (!ok) return;
    if (hipMalloc(dst, bytes) != hipSuccess) {
      ok = false;
      return;
    }
    (void)hipMemcpyAsync(*dst, src, bytes, hipMemcpyDeviceToDevice, p->stream);
  };
  for (auto& st : sets) {
    if (symq) {
      dup(&st.uv, orig.uv, sizeof(double) * (size_t)SY.nu * p->b * p->b);
      dup(&st.uc, orig.uc, sizeof(int32_t) * (size_t)SY.nu);
      dup(&st.lc, orig.lc, sizeof(int32_t) * (size_t)std::max(1, SY.nl));
      dup(&st.ls, orig.ls, sizeof(int32_t) * (size_t)std::max(1, SY.nl));
    } else {
      dup(&st.vals, orig.vals, vbytes);
      dup(&st.colidx, orig.colidx, cbytes);
    }
    dup(&st.x1, orig.x1, p->vec_bytes());
    dup(&st.S1, orig.S1, sbytes);
    dup(&st.z, orig.z, p->vec_bytes());
    dup(&st.delta, orig.delta, p->vec_bytes());
    dup(&st.Hd, orig.Hd, p->vec_bytes());
  }
  auto use = [&](const Set& st) {
    if (symq) {
      SY.uvalsT = st.uv;
      SY.ucol = st.uc;
      SY.lcol = st.lc;
      SY.lslot = st.ls;
    } else {
      p->Q.vals = st.vals;
      p->Q.colidx = st.colidx;
    }
    p->x1 = st.x1;
    p->S1 = st.S1;
    p->z = st.z;
    p->delta = st.delta;
    p->Hd = st.Hd;
  };
  int rc = ok ? DPGO_OK : fail(DPGO_ERR_HIP, "hipMalloc failed for the rotating buffer sets");
  auto launch = [&](int i) -> int {
    use(sets[i % nsets]);
    DISPATCH(p->d, p->r, LAUNCH_TCG_HESS(p, p->dstate, p->dstate + 1, 0, (unsigned long long*)nullptr, 0u));
    HIPC(hipGetLastError());
    return DPGO_OK;
  };
  float ms = 0.f;
  if (rc == DPGO_OK) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < warmup && rc == DPGO_OK; ++i) rc = launch(i);
    (void)hipEventRecord(e0, p->stream);
    for (int i = 0; i < reps && rc == DPGO_OK; ++i) rc = launch(i);
    (void)hipEventRecord(e1, p->stream);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
  use(orig);
  (void)hipStreamSynchronize(p->stream);
  for (auto& st : sets) {
    void* ptrs[] = {st.vals, st.colidx, st.x1, st.S1, st.z, st.delta, st.Hd, st.uv, st.uc, st.lc, st.ls};
    for (void* q : ptrs)
      if (q) (void)hipFree(q);
  }
  if (rc != DPGO_OK) return rc;
  *avg_ms = (double)ms / reps;
  return DPGO_OK;
}

int dpgo_problem_set_spmm_variant(dpgo_problem_t p, int variant, int* in_use) {
  CHK(check_ready(p));
  if (variant != DPGO_SPMM_AUTO && variant != DPGO_SPMM_PLAIN && variant != DPGO_SPMM_SYMMETRIC)
    return fail(DPGO_ERR_INVALID, "unknown product storage");
  p->spmm_variant = variant;
  bool usable = false;
  if (p->sym_wanted()) CHK(sym_ensure(p, &usable));
  if (in_use) *in_use = usable ? DPGO_SPMM_SYMMETRIC : DPGO_SPMM_PLAIN;
  return DPGO_OK;
}

int dpgo_problem_tcg_kernel_info(dpgo_problem_t p, int* symmetric, int* split, int* stream_nt) {
  CHK(check_ready(p));
  CHK(resolve_tcg_storage(p));
  if (symmetric) *symmetric = p->tcg_sym ? 1 : 0;
  if (split) *split = p->tcg_sym ? 1 : p->split;
  if (stream_nt) *stream_nt = (p->stream_nt && (p->tcg_sym || p->split == 1)) ? 1 : 0;
  return DPGO_OK;
}

namespace {
// rotating copies of the symmetric storage (values, column indices, references; the row pointers are shared)
int bench_spmm_sym_rotating(dpgo_problem_s* p, int nsets, int reps, int warmup, double* avg_ms, double* set_bytes) {
  const auto& S = p->sym;
  const size_t vbytes = sizeof(double) * (size_t)S.nu * p->b * p->b;
  struct Set {
    double *v = nullptr, *x = nullptr, *o = nullptr;
    int32_t *uc = nullptr, *lc = nullptr, *ls = nullptr;
  };
  std::vector<Set> sets(nsets);
  int rc = DPGO_OK;
  auto cleanup = [&]() {
    for (auto& st : sets) {
      void* ptrs[] = {st.v, st.x, st.o, st.uc, st.lc, st.ls};
      for (void* q : ptrs)
        if (q) (void)hipFree(q);
    }
  };
  for (auto& st : sets) {
    if (hipMalloc(&st.v, vbytes) != hipSuccess || hipMalloc(&st.x, p->vec_bytes()) != hipSuccess ||
        hipMalloc(&st.o, p->vec_bytes()) != hipSuccess || hipMalloc(&st.uc, sizeof(int32_t) * S.nu) != hipSuccess ||
        hipMalloc(&st.lc, sizeof(int32_t) * std::max(1, S.nl)) != hipSuccess ||
        hipMalloc(&st.ls, sizeof(int32_t) * std::max(1, S.nl)) != hipSuccess) {
      rc = fail(DPGO_ERR_HIP, "hipMalloc failed for the rotating buffer sets");
      break;
    }
    (void)hipMemcpyAsync(st.v, S.uvalsT, vbytes, hipMemcpyDeviceToDevice, p->stream);
    (void)hipMemcpyAsync(st.uc, S.ucol, sizeof(int32_t) * S.nu, hipMemcpyDeviceToDevice, p->stream);
    (void)hipMemcpyAsync(st.lc, S.lcol, sizeof(int32_t) * S.nl, hipMemcpyDeviceToDevice, p->stream);
    (void)hipMemcpyAsync(st.ls, S.lslot, sizeof(int32_t) * S.nl, hipMemcpyDeviceToDevice, p->stream);
    (void)hipMemcpyAsync(st.x, p->x1, p->vec_bytes(), hipMemcpyDeviceToDevice, p->stream);
  }
  if (rc != DPGO_OK) {
    cleanup();
    return rc;
  }
  auto launch = [&](int i) {
    const Set& st = sets[i % nsets];
    return launch_spmm_sym(p, BsrSymDev{S.urow, st.uc, st.v, S.lrow, st.lc, st.ls}, st.x, nullptr, st.o);
  };
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int i = 0; i < warmup && rc == DPGO_OK; ++i) rc = launch(i);
  (void)hipEventRecord(e0, p->stream);
  for (int i = 0; i < reps && rc == DPGO_OK; ++i) rc = launch(i);
  (void)hipEventRecord(e1, p->stream);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  cleanup();
  if (rc != DPGO_OK) return rc;
  *avg_ms = (double)ms / reps;
  if (set_bytes)
    *set_bytes = (double)(vbytes + sizeof(int32_t) * ((size_t)S.nu + 2 * (size_t)S.nl) + 2 * p->vec_bytes());
  return DPGO_OK;
}
}  // namespace

int dpgo_bench_spmm_rotating(dpgo_problem_t p, int nsets, int reps, int warmup, double* avg_ms,
                             double* set_bytes) {
  CHK(check_ready(p));
  if (nsets < 1 || nsets > 512 || reps <= 0 || !avg_ms) return fail(DPGO_ERR_INVALID, "bad arguments");
  if (p->sym_wanted()) {
    bool usable = false;
    CHK(sym_ensure(p, &usable));
    if (usable) return bench_spmm_sym_rotating(p, nsets, reps, warmup, avg_ms, set_bytes);
  }
  // nsets private copies of (Q values, block columns, X, OUT): cycling through them makes every launch read
  // data that left the 256 MB Infinity Cache ("rotate >= 3 buffer sets > 256 MB total")
  const size_t vbytes = sizeof(double) * (size_t)p->Q.nnzb * p->b * p->b;
  const size_t cbytes = sizeof(int32_t) * (size_t)p->Q.nnzb;
  std::vector<Bsr> mats(nsets);
  std::vector<double*> xs(nsets, nullptr), outs(nsets, nullptr);
  int rc = DPGO_OK;
  auto cleanup = [&]() {
    for (int k = 0; k < nsets; ++k) {
      if (mats[k].vals) (void)hipFree(mats[k].vals);
      if (mats[k].colidx) (void)hipFree(mats[k].colidx);
      if (xs[k]) (void)hipFree(xs[k]);
      if (outs[k]) (void)hipFree(outs[k]);
    }
  };
  for (int k = 0; k < nsets && rc == DPGO_OK; ++k) {
    mats[k] = p->Q;  // shares rowptr (0.4 MB)
    mats[k].vals = nullptr;
    mats[k].colidx = nullptr;
    if (hipMalloc(&mats[k].vals, vbytes) != hipSuccess || hipMalloc(&mats[k].colidx, cbytes) != hipSuccess ||
        hipMalloc(&xs[k], p->vec_bytes()) != hipSuccess || hipMalloc(&outs[k], p->vec_bytes()) != hipSuccess) {
      rc = fail(DPGO_ERR_HIP, "hipMalloc failed for the rotating buffer sets");
      break;
    }
    (void)hipMemcpyAsync(mats[k].vals, p->Q.vals, vbytes, hipMemcpyDeviceToDevice, p->stream);
    (void)hipMemcpyAsync(mats[k].colidx, p->Q.colidx, cbytes, hipMemcpyDeviceToDevice, p->stream);
    (void)hipMemcpyAsync(xs[k], p->x1, p->vec_bytes(), hipMemcpyDeviceToDevice, p->stream);
  }
  if (rc != DPGO_OK) {
    cleanup();
    return rc;
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int i = 0; i < warmup && rc == DPGO_OK; ++i) rc = launch_spmm(p, mats[i % nsets], xs[i % nsets], nullptr, outs[i % nsets]);
  (void)hipEventRecord(e0, p->stream);
  for (int i = 0; i < reps && rc == DPGO_OK; ++i) rc = launch_spmm(p, mats[i % nsets], xs[i % nsets], nullptr, outs[i % nsets]);
  (void)hipEventRecord(e1, p->stream);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  cleanup();
  if (rc != DPGO_OK) return rc;
  *avg_ms = (double)ms / reps;
  if (set_bytes) *set_bytes = (double)(vbytes + cbytes + 2 * p->vec_bytes());
  return DPGO_OK;
}

int dpgo_bench_hess(dpgo_problem_t p, int reps, int warmup, double* avg_ms) {
  CHK(check_ready(p));
  if (reps <= 0 || !avg_ms) return fail(DPGO_ERR_INVALID, "bad arguments");
  CHK(resolve_tcg_storage(p));
  // a state in which the tCG-step kernel never takes an early exit
  std::memset(p->hstate, 0, sizeof(DevState));
  p->hstate->z_r = 1.0;
  p->hstate->theta = 1.0;
  p->hstate->kappa = -1.0;  // convergence test can never fire
  p->hstate->max_inner = 1 << 30;
  p->hstate->min_inner = 1 << 30;  // the convergence test is never evaluated, whatever the partial sums hold
  CHK(push_state(p));
  auto launch = [&]() -> int {
    DISPATCH(p->d, p->r, LAUNCH_TCG_HESS(p, p->dstate, p->dstate + 1, 0, (unsigned long long*)nullptr, 0u));
    HIPC(hipGetLastError());
    return DPGO_OK;
  };
  hipEvent_t e0, e1;
  HIPC(hipEventCreate(&e0));
  HIPC(hipEventCreate(&e1));
  for (int i = 0; i < warmup; ++i) CHK(launch());
  HIPC(hipEventRecord(e0, p->stream));
  for (int i = 0; i < reps; ++i) CHK(launch());
  HIPC(hipEventRecord(e1, p->stream));
  HIPC(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPC(hipEventElapsedTime(&ms, e0, e1));
  HIPC(hipEventDestroy(e0));
  HIPC(hipEventDestroy(e1));
  *avg_ms = (double)ms / reps;
  return DPGO_OK;
}

int dpgo_bench_solve(dpgo_problem_t p, const dpgo_ropt_params* params, const double* X0_dev, int reps, int warmup,
                     double* avg_ms, double* avg_products, int* persistent) {
  CHK(check_ready(p));
  if (reps <= 0 || !params || !X0_dev || !avg_ms) return fail(DPGO_ERR_INVALID, "bad arguments");
  // every repetition solves from the same iterate (copied in outside the event pair)
  hipEvent_t e0, e1;
  HIPC(hipEventCreate(&e0));
  HIPC(hipEventCreate(&e1));
  double total = 0.0, products = 0.0;
  bool all_persistent = true;
  int rc = DPGO_OK;
  for (int i = 0; i < warmup + reps && rc == DPGO_OK; ++i) {
    rc = [&]() -> int {
      HIPC(hipMemcpyAsync(p->x1, X0_dev, p->vec_bytes(), hipMemcpyDeviceToDevice, p->stream));
      dpgo_ropt_result res;
      HIPC(hipEventRecord(e0, p->stream));
      CHK(run_optimize(p, params, &res));
      HIPC(hipEventRecord(e1, p->stream));
      HIPC(hipEventSynchronize(e1));
      float ms = 0.f;
      HIPC(hipEventElapsedTime(&ms, e0, e1));
      if (i >= warmup) {
        total += ms;
        products += res.tcg_iterations;
        all_persistent = all_persistent && p->hctrl && p->hctrl->members > 0 && !p->persist_failed_once;
      }
      return DPGO_OK;
    }();
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (rc != DPGO_OK) return rc;
  *avg_ms = total / reps;
  if (avg_products) *avg_products = products / reps;
  if (persistent) *persistent = all_persistent ? 1 : 0;
  return DPGO_OK;
}

int dpgo_bench_iteration_kernels(dpgo_problem_t p, int reps, int warmup, double out_ms[5]) {
  CHK(check_ready(p));
  if (reps <= 0 || !out_ms) return fail(DPGO_ERR_INVALID, "bad arguments");
  for (int q = 0; q < 5; ++q) out_ms[q] = 0.0;
  CHK(resolve_tcg_storage(p));
  // a state in which no kernel takes an early exit (as dpgo_bench_hess); alpha = z_r / d_Hd stays finite
  std::memset(p->hstate, 0, sizeof(DevState));
  p->hstate->z_r = 1.0;
  p->hstate->theta = 1.0;
  p->hstate->kappa = -1.0;
  p->hstate->max_inner = 1 << 30;
  p->hstate->min_inner = 1 << 30;  // the convergence test is never evaluated, whatever the partial sums hold
  p->hstate->Delta = 1e300;
  CHK(push_state(p));
  CHK(build_dinv(p, p->ml_ready ? p->ml_shift : 1e-1));
  {  // <delta, H delta> partials of a "previous k_tcg_hess": positive, so that the update kernel takes its regular path
    std::vector<double> ones((size_t)kPartialCap * kNP, 1.0);
    HIPC(hipMemcpyAsync(p->pA(), ones.data(), sizeof(double) * ones.size(), hipMemcpyHostToDevice, p->stream));
    HIPC(hipStreamSynchronize(p->stream));
  }
  hipEvent_t e0, e1;
  HIPC(hipEventCreate(&e0));
  HIPC(hipEventCreate(&e1));
  auto timed = [&](auto&& launch, double* out) -> int {
    for (int i = 0; i < warmup; ++i) CHK(launch());
    HIPC(hipEventRecord(e0, p->stream));
    for (int i = 0; i < reps; ++i) CHK(launch());
    HIPC(hipEventRecord(e1, p->stream));
    HIPC(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPC(hipEventElapsedTime(&ms, e0, e1));
    *out = (double)ms / reps;
    return DPGO_OK;
  };
  const bool ml = p->ml_ready;
  int rc = timed([&]() -> int {
    const int cur = p->cur;
    int r2 = launch_tcg_update(p, p->dinv, 0, ml ? p->ml[0].x1 : nullptr, ml ? p->ml_omega : 0.0);
    p->cur = cur;  // keep reading the pushed state
    return r2;
  }, &out_ms[0]);
  if (rc == DPGO_OK && ml) {
    const int nl = (int)p->ml.size();
    auto& L0 = p->ml[0];
    const bool ap = p->ml_use_ap();
    rc = timed([&]() -> int { return launch_ml_restrict0(p, p->rr, nullptr, p->grid_restrict()); }, &out_ms[1]);
    if (rc == DPGO_OK) rc = timed([&]() -> int {
      auto& L = p->ml[nl - 2];
      auto& Cc = p->ml[nl - 1];
      if (p->ml_use_dense_sym()) return launch_dense_sym(p, Cc, nullptr);
      return launch_coarse_prolong(p, L, Cc, nullptr, ap ? Cc.x : nullptr);
    }, &out_ms[2]);
    if (rc == DPGO_OK) rc = timed([&]() -> int {
      if (ap) {
        return launch_ml_post_ap(p, p->x1, p->rr, p->z, p->pB(), nullptr);
      } else if (p->tcg_sym) {
        DISPATCH(p->d, p->r, hipLaunchKernelGGL((k_ml_post<D, R, 1, BsrSymDev>), dim3(p->grid_post()), dim3(kBlock), 0,
                                                p->stream, p->sym.dev(), p->x1, L0.x, p->rr, p->dinv, p->ml_omega,
                                                p->ml_shift, p->z, p->pB(), (const DevState*)nullptr, p->n));
      } else {
        DISPATCH(p->d, p->r, LAUNCH_SPLIT(p, k_ml_post, p->grid_post(), p->Q.dev(), p->x1, L0.x, p->rr, p->dinv, p->ml_omega,
                                          p->ml_shift, p->z, p->pB(), (const DevState*)nullptr, p->n));
      }
      HIPC(hipGetLastError());
      return DPGO_OK;
    }, &out_ms[3]);
    if (rc == DPGO_OK)
      rc = timed([&]() -> int { return launch_ml_tail(p, p->x1, p->rr, p->z, p->pB(), nullptr); }, &out_ms[4]);
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return rc;
}

int dpgo_device_malloc(void** out, size_t bytes, int device) {
  if (!out) return fail(DPGO_ERR_INVALID, "null out");
  *out = nullptr;
  int cnt = 0;
  CHK(dpgo_device_count(&cnt));
  if (cnt <= 0) return fail(DPGO_ERR_HIP, "no HIP device (this library has no CPU fallback)");
  if (device < 0 || device >= cnt) return fail(DPGO_ERR_INVALID, "device index out of range");
  HIPC(hipSetDevice(device));
  HIPC(hipMalloc(out, bytes > 0 ? bytes : 1));
  return DPGO_OK;
}

int dpgo_device_free(void* p) {
  if (p) HIPC(hipFree(p));
  return DPGO_OK;
}

int dpgo_device_memcpy(void* dst, const void* src, size_t bytes, int kind, void* stream) {
  if (bytes == 0) return DPGO_OK;
  if (!dst || !src) return fail(DPGO_ERR_INVALID, "null pointer");
  hipMemcpyKind k;
  switch (kind) {
    case DPGO_COPY_H2D: k = hipMemcpyHostToDevice; break;
    case DPGO_COPY_D2H: k = hipMemcpyDeviceToHost; break;
    case DPGO_COPY_D2D: k = hipMemcpyDeviceToDevice; break;
    default: return fail(DPGO_ERR_INVALID, "unknown copy kind");
  }
  HIPC(hipMemcpyAsync(dst, src, bytes, k, (hipStream_t)stream));
  if (kind == DPGO_COPY_D2H) HIPC(hipStreamSynchronize((hipStream_t)stream));
  return DPGO_OK;
}

int dpgo_device_synchronize(void* stream) {
  HIPC(hipStreamSynchronize((hipStream_t)stream));
  return DPGO_OK;
}

// ---- initial guesses ----
namespace {
// Masked PCG: solve  mask A mask x = rhs  (rhs already masked) for the tiles x; A = the handle's Q.  Host-driven
// (two tiny read-backs per iteration): initialisation runs once per problem, outside the hot path.
struct InitBufs {
  double *x, *r, *z, *p, *Ap, *diag, *partial;
};
int init_dot(dpgo_problem_s* h, const double* a, const double* b, InitBufs& w, size_t total, int g, double* out) {
  hipLaunchKernelGGL(k_init_dot, dim3(g), dim3(kBlock), 0, h->stream, a, b, w.partial, total);
  HIPC(hipGetLastError());
  std::vector<double> host(g);
  HIPC(hipMemcpyAsync(host.data(), w.partial, sizeof(double) * g, hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  double s = 0.0;
  for (double v : host) s += v;
  *out = s;
  return DPGO_OK;
}
int init_pcg(dpgo_problem_s* h, InitBufs& w, const double* rhs, int mode, double tol, int max_iter, int* iters) {
  const int T = h->T, R = h->r, D = h->d;
  const size_t total = (size_t)h->n * T;
  const int g = std::max(1, std::min(kMaxGrid, (int)((total + kBlock - 1) / kBlock)));
  auto axpby = [&](double a, const double* x, double b, double* y) -> int {
    hipLaunchKernelGGL(k_init_axpby, dim3(g), dim3(kBlock), 0, h->stream, a, x, b, y, total, T, R, D, mode);
    HIPC(hipGetLastError());
    return DPGO_OK;
  };
  auto apply = [&](const double* v, double* out) -> int {  // out = mask(A v), v masked
    CHK(launch_spmm(h, h->Q, v, nullptr, out));
    return axpby(1.0, out, 0.0, out);
  };
  auto precond = [&](const double* r, double* z) -> int {
    hipLaunchKernelGGL(k_init_jacobi, dim3(g), dim3(kBlock), 0, h->stream, r, w.diag, z, total, T, R, D, mode);
    HIPC(hipGetLastError());
    return DPGO_OK;
  };
  HIPC(hipMemsetAsync(w.x, 0, sizeof(double) * total, h->stream));
  CHK(axpby(1.0, rhs, 0.0, w.r));
  CHK(precond(w.r, w.z));
  CHK(axpby(1.0, w.z, 0.0, w.p));
  double rz = 0.0, r0 = 0.0;
  CHK(init_dot(h, w.r, w.z, w, total, g, &rz));
  CHK(init_dot(h, w.r, w.r, w, total, g, &r0));
  *iters = 0;
  if (!(r0 > 0.0)) return DPGO_OK;
  double best = r0;
  for (int it = 0; it < max_iter; ++it) {
    CHK(apply(w.p, w.Ap));
    double pAp = 0.0;
    CHK(init_dot(h, w.p, w.Ap, w, total, g, &pAp));
    if (!(pAp > 0.0)) break;
    const double alpha = rz / pAp;
    CHK(axpby(alpha, w.p, 1.0, w.x));
    CHK(axpby(-alpha, w.Ap, 1.0, w.r));
    CHK(precond(w.r, w.z));
    double rz_new = 0.0, rr = 0.0;
    CHK(init_dot(h, w.r, w.z, w, total, g, &rz_new));
    CHK(init_dot(h, w.r, w.r, w, total, g, &rr));
    *iters = it + 1;
    best = std::min(best, rr);
    if (rr <= tol * tol * r0) break;
    CHK(axpby(1.0, w.z, rz_new / rz, w.p));
    rz = rz_new;
  }
  return DPGO_OK;
}
}  // namespace

int dpgo_chordal_initialization(int d, int n, int m, const int32_t* p1, const int32_t* p2, const double* R,
                                const double* t, const double* kappa, const double* tau, double tol, int max_iter,
                                double* T_host, int iters_out[2], int device) {
  if ((d != 2 && d != 3) || n <= 0 || m < 0 || !T_host || (m > 0 && (!p1 || !p2 || !R || !t || !kappa || !tau)))
    return fail(DPGO_ERR_INVALID, "bad arguments");
  if (!(tol > 0.0)) tol = 1e-13;
  if (max_iter <= 0) max_iter = (int)std::min<long long>(20ll * n + 100, 200000);
  const int b = d + 1, r = d;  // tiles [n][d+1][d]: the rank-d "lifted" problem IS the SE(d) problem
  std::vector<int32_t> zero(std::max(m, 1), 0);
  std::vector<double> ones(std::max(m, 1), 1.0), tau0(std::max(m, 1), 0.0);
  dpgo_problem_t hq[2] = {nullptr, nullptr};  // [0]: rotation-only connection Laplacian (tau = 0), [1]: Q
  struct Cleanup {
    dpgo_problem_t* h;
    ~Cleanup() {
      dpgo_problem_destroy(h[0]);
      dpgo_problem_destroy(h[1]);
    }
  } cleanup{hq};
  for (int which = 0; which < 2; ++which) {
    const double* tw = which == 0 ? tau0.data() : tau;
    int nnzb = 0;
    int rc = dpgo_build_Q_bsr(0, d, n, m, zero.data(), p1, zero.data(), p2, R, t, kappa, tw, ones.data(), 0, nullptr, 0.0,
                              0.0, &nnzb, nullptr, nullptr, nullptr);
    if (rc != DPGO_OK) return fail(rc, "chordal initialisation: measurement index out of range");
    std::vector<int32_t> rowptr(n + 1), colidx(nnzb);
    std::vector<double> vals((size_t)nnzb * b * b);
    rc = dpgo_build_Q_bsr(0, d, n, m, zero.data(), p1, zero.data(), p2, R, t, kappa, tw, ones.data(), 0, nullptr, 0.0, 0.0,
                          &nnzb, rowptr.data(), colidx.data(), vals.data());
    if (rc != DPGO_OK) return fail(rc, "chordal initialisation: could not build the data matrix");
    CHK(dpgo_problem_create(&hq[which], r, d, n, device));
    CHK(dpgo_problem_set_Q_bsr(hq[which], nnzb, rowptr.data(), colidx.data(), vals.data()));
  }
  dpgo_problem_s* hr = hq[0];
  dpgo_problem_s* ht = hq[1];
  const size_t total = (size_t)n * hr->T;
  TmpDev tmp;
  InitBufs w{};
  double *rhs = nullptr, *V = nullptr, *Tr = nullptr;
  for (double** v : {&w.x, &w.r, &w.z, &w.p, &w.Ap, &rhs, &V, &Tr}) CHK(tmp.alloc(v, sizeof(double) * total));
  CHK(tmp.alloc(&w.diag, sizeof(double) * (size_t)n * b));
  CHK(tmp.alloc(&w.partial, sizeof(double) * kMaxGrid));
  const int gflat = std::max(1, std::min(kMaxGrid, (n + kBlock - 1) / kBlock));
  const int gtot = std::max(1, std::min(kMaxGrid, (int)((total + kBlock - 1) / kBlock)));
  int it_rot = 0, it_tr = 0;
  // ---- rotations: minimise sum kappa |R_j - R_i R_ij|^2, R_0 = I.  With E0 = tile 0 = [I | 0]:  L (E0 + x) = 0 on
  // the free rows  =>  mask L mask x = -mask(L E0)
  std::vector<double> e0(hr->T, 0.0);
  for (int c = 0; c < d; ++c) e0[(size_t)c * r + c] = 1.0;
  HIPC(hipMemsetAsync(V, 0, sizeof(double) * total, hr->stream));
  HIPC(hipStreamSynchronize(hr->stream));  // (the host copy below comes from pageable memory: keep it strictly after)
  HIPC(hipMemcpyAsync(V, e0.data(), sizeof(double) * hr->T, hipMemcpyHostToDevice, hr->stream));
  HIPC(hipStreamSynchronize(hr->stream));
  CHK(launch_spmm(hr, hr->Q, V, nullptr, rhs));
  hipLaunchKernelGGL(k_init_axpby, dim3(gtot), dim3(kBlock), 0, hr->stream, -1.0, rhs, 0.0, rhs, total, hr->T, r, d, 0);
  if (d == 2)
    hipLaunchKernelGGL(k_init_diag<2>, dim3(gflat), dim3(kBlock), 0, hr->stream, hr->Q.dev(), w.diag, n);
  else
    hipLaunchKernelGGL(k_init_diag<3>, dim3(gflat), dim3(kBlock), 0, hr->stream, hr->Q.dev(), w.diag, n);
  HIPC(hipGetLastError());
  CHK(init_pcg(hr, w, rhs, 0, tol, max_iter, &it_rot));
  // V = E0 + x, then every block to SO(d) (projectToRotationGroup, src/DPGO_utils.cpp:464-478): the rounding kernel with
  // the identity as anchor
  hipLaunchKernelGGL(k_init_axpby, dim3(gtot), dim3(kBlock), 0, hr->stream, 1.0, w.x, 0.0, w.x, total, hr->T, r, d, 0);
  HIPC(hipMemcpyAsync(V, w.x, sizeof(double) * total, hipMemcpyDeviceToDevice, hr->stream));
  HIPC(hipStreamSynchronize(hr->stream));  // (as above)
  HIPC(hipMemcpyAsync(V, e0.data(), sizeof(double) * hr->T, hipMemcpyHostToDevice, hr->stream));
  HIPC(hipStreamSynchronize(hr->stream));
  CHK(dpgo_round_trajectory_device(r, d, n, V, e0.data(), Tr, hr->stream));
  HIPC(hipStreamSynchronize(hr->stream));
  // ---- translations: minimise sum tau |t_j - t_i - R_i t_ij|^2, t_0 = 0: the translation columns of Q [R | t] = 0
  CHK(launch_spmm(ht, ht->Q, Tr, nullptr, rhs));
  hipLaunchKernelGGL(k_init_axpby, dim3(gtot), dim3(kBlock), 0, ht->stream, -1.0, rhs, 0.0, rhs, total, ht->T, r, d, 1);
  if (d == 2)
    hipLaunchKernelGGL(k_init_diag<2>, dim3(gflat), dim3(kBlock), 0, ht->stream, ht->Q.dev(), w.diag, n);
  else
    hipLaunchKernelGGL(k_init_diag<3>, dim3(gflat), dim3(kBlock), 0, ht->stream, ht->Q.dev(), w.diag, n);
  HIPC(hipGetLastError());
  CHK(init_pcg(ht, w, rhs, 1, tol, max_iter, &it_tr));
  // T = [R | t]: rotation columns from Tr, translation column from the solve (pose 0: zero)
  hipLaunchKernelGGL(k_init_axpby, dim3(gtot), dim3(kBlock), 0, ht->stream, 1.0, w.x, 0.0, w.x, total, ht->T, r, d, 1);
  hipLaunchKernelGGL(k_axpby_plain, dim3(gtot), dim3(kBlock), 0, ht->stream, 1.0, w.x, 1.0, Tr, total);
  HIPC(hipGetLastError());
  HIPC(hipMemcpyAsync(T_host, Tr, sizeof(double) * total, hipMemcpyDeviceToHost, ht->stream));
  HIPC(hipStreamSynchronize(ht->stream));
  if (iters_out) {
    iters_out[0] = it_rot;
    iters_out[1] = it_tr;
  }
  return DPGO_OK;
}

int dpgo_odometry_initialization(int d, int n, int m, const int32_t* p1, const int32_t* p2, const double* R,
                                 const double* t, double* T_host) {
  if ((d != 2 && d != 3) || n <= 0 || m < 0 || !T_host || (m > 0 && (!p1 || !p2 || !R || !t)))
    return fail(DPGO_ERR_INVALID, "bad arguments");
  const int b = d + 1;
  std::vector<int> edge_of(n, -1);  // odometry edge leaving pose i (i -> i + 1)
  for (int e = 0; e < m; ++e)
    if (p1[e] >= 0 && p1[e] + 1 == p2[e] && p2[e] < n && edge_of[p1[e]] < 0) edge_of[p1[e]] = e;
  std::memset(T_host, 0, sizeof(double) * (size_t)n * b * d);
  for (int c = 0; c < d; ++c) T_host[(size_t)c * d + c] = 1.0;  // tile 0 = [I | 0]
  for (int dst = 1; dst < n; ++dst) {
    const int e = edge_of[dst - 1];
    if (e < 0) return fail(DPGO_ERR_INVALID, "odometry initialisation: no odometry edge " + std::to_string(dst - 1) +
                                                 " -> " + std::to_string(dst));  // reference: CHECK(m.p1 == src)
    const double* Ts = T_host + (size_t)(dst - 1) * b * d;  // tile [c][row]: R(row, c) at c*d + row, t(row) at d*d + row
    double* Td = T_host + (size_t)dst * b * d;
    const double* Re = R + (size_t)e * d * d;  // R[e][row][col]
    const double* te = t + (size_t)e * d;
    for (int row = 0; row < d; ++row) {
      for (int c = 0; c < d; ++c) {
        double s = 0.0;
        for (int k = 0; k < d; ++k) s += Ts[(size_t)k * d + row] * Re[k * d + c];  // (R_src R_e)(row, c)
        Td[(size_t)c * d + row] = s;
      }
      double s = Ts[(size_t)d * d + row];
      for (int k = 0; k < d; ++k) s += Ts[(size_t)k * d + row] * te[k];  // t_src + R_src t_e
      Td[(size_t)d * d + row] = s;
    }
  }
  return DPGO_OK;
}

// ---- manifold ----
namespace {
int manifold_args(int r, int d, int n, int device) {
  if (n <= 0 || r < d || d < 2 || d > 3) return fail(DPGO_ERR_INVALID, "need n > 0, r >= d, d in {2,3}");
  if (!supported(d, r)) return fail(DPGO_ERR_UNSUPPORTED, "(d, r) not compiled in");
  int cnt = 0;
  CHK(dpgo_device_count(&cnt));
  if (cnt <= 0) return fail(DPGO_ERR_HIP, "no HIP device (this library has no CPU fallback)");
  if (device < 0 || device >= cnt) return fail(DPGO_ERR_INVALID, "device index out of range");
  HIPC(hipSetDevice(device));
  return DPGO_OK;
}
int tiles_grid(int d, int n) {
  const int P = (64 / (d + 1)) * kWaves;
  int t = (n + P - 1) / P;
  if (t < 1) t = 1;
  return t < kMaxGrid ? t : kMaxGrid;
}
}  // namespace

int dpgo_manifold_project_device(int r, int d, int n, const double* M_dev, double* out_dev, void* stream) {
  return dpgo_axpby_project_device(r, d, n, 1.0, M_dev, 0.0, nullptr, 0.0, nullptr, 1, out_dev, stream);
}

int dpgo_axpby_project_device(int r, int d, int n, double a, const double* A_dev, double b, const double* B_dev,
                              double c, const double* C_dev, int project, double* out_dev, void* stream) {
  if (!A_dev || !out_dev) return fail(DPGO_ERR_INVALID, "null pointer");
  if (n <= 0) return fail(DPGO_ERR_INVALID, "n <= 0");
  DISPATCH(d, r, hipLaunchKernelGGL((k_axpby_project<D, R>), dim3(tiles_grid(d, n)), dim3(kBlock), 0,
                                    (hipStream_t)stream, a, A_dev, b, B_dev, c, C_dev, project, out_dev, n));
  HIPC(hipGetLastError());
  return DPGO_OK;
}

int dpgo_round_trajectory_device(int r, int d, int n, const double* X_dev, const double* anchor_host, double* T_dev,
                                 void* stream) {
  if (!X_dev || !T_dev) return fail(DPGO_ERR_INVALID, "null pointer");
  if (n <= 0) return fail(DPGO_ERR_INVALID, "n <= 0");
  if (!dpgo_supported(d, r)) return fail(DPGO_ERR_UNSUPPORTED, "unsupported (d, r)");
  AnchorArg an;
  std::memset(&an, 0, sizeof(an));
  an.use = anchor_host ? 1 : 0;
  if (anchor_host) std::memcpy(an.v, anchor_host, sizeof(double) * (size_t)(d + 1) * r);
  int g = (n + kBlock - 1) / kBlock;
  if (g > kMaxGrid) g = kMaxGrid;
  DISPATCH(d, r, hipLaunchKernelGGL((k_round<D, R>), dim3(g), dim3(kBlock), 0, (hipStream_t)stream, X_dev, an, T_dev, n));
  HIPC(hipGetLastError());
  return DPGO_OK;
}

int dpgo_round_trajectory(int r, int d, int n, const double* X_host, const double* anchor_host, double* T_host,
                          int device) {
  if (!X_host || !T_host) return fail(DPGO_ERR_INVALID, "null pointer");
  CHK(manifold_args(r, d, n, device));
  TmpDev tmp;
  const size_t xb = sizeof(double) * (size_t)n * (d + 1) * r, tb = sizeof(double) * (size_t)n * (d + 1) * d;
  double *X = nullptr, *T = nullptr;
  CHK(tmp.alloc(&X, xb));
  CHK(tmp.alloc(&T, tb));
  HIPC(hipMemcpy(X, X_host, xb, hipMemcpyHostToDevice));
  CHK(dpgo_round_trajectory_device(r, d, n, X, anchor_host, T, nullptr));
  HIPC(hipMemcpy(T_host, T, tb, hipMemcpyDeviceToHost));
  return DPGO_OK;
}

int dpgo_gather_tiles_device(int r, int d, const double* src_dev, const int32_t* idx_dev, int count, double* dst_dev,
                             void* stream) {
  if (count == 0) return DPGO_OK;
  if (!src_dev || !idx_dev || !dst_dev || count < 0) return fail(DPGO_ERR_INVALID, "bad arguments");
  size_t total = (size_t)count * (d + 1) * r;
  int g = (int)((total + kBlock - 1) / kBlock);
  if (g > kMaxGrid) g = kMaxGrid;
  DISPATCH(d, r, hipLaunchKernelGGL((k_gather_tiles<D, R>), dim3(g), dim3(kBlock), 0, (hipStream_t)stream, src_dev,
                                    idx_dev, count, dst_dev));
  HIPC(hipGetLastError());
  return DPGO_OK;
}

int dpgo_permute_tiles_device(int r, int d, int n, const int32_t* new_index_dev, const double* in_dev, double* out_dev,
                              int forward, void* stream) {
  if (n == 0) return DPGO_OK;
  if (!new_index_dev || !in_dev || !out_dev || n < 0 || in_dev == out_dev) return fail(DPGO_ERR_INVALID, "bad arguments");
  size_t total = (size_t)n * (d + 1) * r;
  int g = (int)((total + kBlock - 1) / kBlock);
  if (g > kMaxGrid) g = kMaxGrid;
  if (forward) {
    DISPATCH(d, r, hipLaunchKernelGGL((k_scatter_tiles<D, R>), dim3(g), dim3(kBlock), 0, (hipStream_t)stream, in_dev,
                                      new_index_dev, n, out_dev));
  } else {
    DISPATCH(d, r, hipLaunchKernelGGL((k_gather_tiles<D, R>), dim3(g), dim3(kBlock), 0, (hipStream_t)stream, in_dev,
                                      new_index_dev, n, out_dev));
  }
  HIPC(hipGetLastError());
  return DPGO_OK;
}

struct dpgo_exchange_plan_s {
  int device = 0, T = 0, nmsg = 0, total = 0;
  void *src = nullptr, *idx = nullptr, *dst = nullptr, *first = nullptr;
};

int dpgo_exchange_plan_create(dpgo_exchange_plan_t* out, int r, int d, int nmsg, const double* const* src_dev,
                              const int32_t* const* idx_dev, const int* count, double* const* dst_dev, int device) {
  if (!out || nmsg <= 0 || !src_dev || !idx_dev || !count || !dst_dev || !supported(d, r))
    return fail(DPGO_ERR_INVALID, "bad exchange plan arguments");
  *out = nullptr;
  std::vector<int32_t> first(nmsg + 1, 0);
  for (int m = 0; m < nmsg; ++m) {
    if (count[m] < 0 || (count[m] > 0 && (!src_dev[m] || !idx_dev[m] || !dst_dev[m]))) return fail(DPGO_ERR_INVALID, "bad message");
    first[m + 1] = first[m] + count[m];
  }
  HIPC(hipSetDevice(device));
  auto* pl = new dpgo_exchange_plan_s();
  pl->device = device;
  pl->T = (d + 1) * r;
  pl->nmsg = nmsg;
  pl->total = first[nmsg];
  int rc = [&]() -> int {
    HIPC(hipMalloc(&pl->src, sizeof(void*) * nmsg));
    HIPC(hipMalloc(&pl->idx, sizeof(void*) * nmsg));
    HIPC(hipMalloc(&pl->dst, sizeof(void*) * nmsg));
    HIPC(hipMalloc(&pl->first, sizeof(int32_t) * (nmsg + 1)));
    HIPC(hipMemcpy(pl->src, src_dev, sizeof(void*) * nmsg, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(pl->idx, idx_dev, sizeof(void*) * nmsg, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(pl->dst, dst_dev, sizeof(void*) * nmsg, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(pl->first, first.data(), sizeof(int32_t) * (nmsg + 1), hipMemcpyHostToDevice));
    return DPGO_OK;
  }();
  if (rc != DPGO_OK) {
    dpgo_exchange_plan_destroy(pl);
    return rc;
  }
  *out = pl;
  return DPGO_OK;
}

int dpgo_exchange_plan_run(dpgo_exchange_plan_t pl, void* stream) {
  if (!pl) return fail(DPGO_ERR_INVALID, "null exchange plan");
  if (pl->total == 0) return DPGO_OK;
  HIPC(hipSetDevice(pl->device));
  const ExchangeTable tb{(const double* const*)pl->src, (const int32_t* const*)pl->idx, (double* const*)pl->dst,
                         (const int32_t*)pl->first, pl->nmsg};
  const int g = std::max(1, std::min(kMaxGrid, (pl->total + kBlock / 4 - 1) / (kBlock / 4)));
  switch (pl->T) {
#define CASE_T(TT) case TT: hipLaunchKernelGGL((k_gather_tiles_batched<TT>), dim3(g), dim3(kBlock), 0, (hipStream_t)stream, tb); break;
    CASE_T(6) CASE_T(9) CASE_T(12) CASE_T(15) CASE_T(16) CASE_T(20) CASE_T(24)
#undef CASE_T
    default: return fail(DPGO_ERR_UNSUPPORTED, "unsupported (d, r)");
  }
  HIPC(hipGetLastError());
  return DPGO_OK;
}

int dpgo_exchange_plan_destroy(dpgo_exchange_plan_t pl) {
  if (!pl) return DPGO_OK;
  for (void* q : {pl->src, pl->idx, pl->dst, pl->first})
    if (q) (void)hipFree(q);
  delete pl;
  return DPGO_OK;
}

int dpgo_max_translation_distance_device(int r, int d, int n, const double* X_dev, const double* Xprev_dev,
                                         double* out_dev, double* out_host, void* stream) {
  if (!X_dev || !Xprev_dev || !out_dev || n <= 0) return fail(DPGO_ERR_INVALID, "bad arguments");
  hipStream_t s = (hipStream_t)stream;
  HIPC(hipMemsetAsync(out_dev, 0, sizeof(double), s));
  const int g = std::max(1, std::min(kMaxGrid, (n + kBlock - 1) / kBlock));
  DISPATCH(d, r, hipLaunchKernelGGL((k_max_translation_distance<D, R>), dim3(g), dim3(kBlock), 0, s, X_dev, Xprev_dev, n,
                                    reinterpret_cast<unsigned long long*>(out_dev)));
  HIPC(hipGetLastError());
  if (out_host) {
    HIPC(hipMemcpyAsync(out_host, out_dev, sizeof(double), hipMemcpyDeviceToHost, s));
    HIPC(hipStreamSynchronize(s));
  }
  return DPGO_OK;
}

int dpgo_manifold_project(int r, int d, int n, const double* M, double* out, int device) {
  if (!M || !out) return fail(DPGO_ERR_INVALID, "null pointer");
  CHK(manifold_args(r, d, n, device));
  TmpDev tmp;
  const size_t vb = sizeof(double) * (size_t)n * (d + 1) * r;
  double *a = nullptr, *o = nullptr;
  CHK(tmp.alloc(&a, vb));
  CHK(tmp.alloc(&o, vb));
  HIPC(hipMemcpy(a, M, vb, hipMemcpyHostToDevice));
  CHK(dpgo_manifold_project_device(r, d, n, a, o, nullptr));
  HIPC(hipMemcpy(out, o, vb, hipMemcpyDeviceToHost));
  return DPGO_OK;
}

int dpgo_manifold_tangent_project(int r, int d, int n, const double* X, const double* V, double* out, int device) {
  if (!X || !V || !out) return fail(DPGO_ERR_INVALID, "null pointer");
  CHK(manifold_args(r, d, n, device));
  TmpDev tmp;
  const size_t vb = sizeof(double) * (size_t)n * (d + 1) * r;
  double *x = nullptr, *v = nullptr, *o = nullptr;
  CHK(tmp.alloc(&x, vb));
  CHK(tmp.alloc(&v, vb));
  CHK(tmp.alloc(&o, vb));
  HIPC(hipMemcpy(x, X, vb, hipMemcpyHostToDevice));
  HIPC(hipMemcpy(v, V, vb, hipMemcpyHostToDevice));
  DISPATCH(d, r, hipLaunchKernelGGL((k_precond<D, R>), dim3(tiles_grid(d, n)), dim3(kBlock), 0, (hipStream_t) nullptr,
                                    x, v, (const double*)nullptr, o, n));
  HIPC(hipGetLastError());
  HIPC(hipMemcpy(out, o, vb, hipMemcpyDeviceToHost));
  return DPGO_OK;
}

int dpgo_manifold_retract(int r, int d, int n, const double* X, const double* eta, double scale, double* out,
                          int device) {
  if (!X || !eta || !out) return fail(DPGO_ERR_INVALID, "null pointer");
  CHK(manifold_args(r, d, n, device));
  TmpDev tmp;
  const size_t vb = sizeof(double) * (size_t)n * (d + 1) * r;
  double *x = nullptr, *v = nullptr, *o = nullptr;
  CHK(tmp.alloc(&x, vb));
  CHK(tmp.alloc(&v, vb));
  CHK(tmp.alloc(&o, vb));
  HIPC(hipMemcpy(x, X, vb, hipMemcpyHostToDevice));
  HIPC(hipMemcpy(v, eta, vb, hipMemcpyHostToDevice));
  DISPATCH(d, r, hipLaunchKernelGGL((k_retract<D, R>), dim3(tiles_grid(d, n)), dim3(kBlock), 0, (hipStream_t) nullptr,
                                    x, v, scale, o, (const DevState*)nullptr, n));
  HIPC(hipGetLastError());
  HIPC(hipMemcpy(out, o, vb, hipMemcpyDeviceToHost));
  return DPGO_OK;
}

}  // extern "C"
