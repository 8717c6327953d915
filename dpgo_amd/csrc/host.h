// host.h -- what the translation units of libdpgo_hip.so share: error reporting, the (d, r) dispatch macros, the problem
// handle (struct dpgo_problem_s: device buffers, hierarchy, solver state of one PoseGraph) and the declarations of the
// host-side helpers each unit defines.
//   problem.hip     handle lifecycle, Q / G upload, symmetric storage, QuadraticProblem evaluations (k_spmm, k_grad, k_hess ...)
//   multilevel.hip  hierarchy set-up (symbolic on the host, numeric on the device) and the V-cycle's launches
//   solve.hip       QuadraticOptimizer::optimize: tCG launches, the one-launch solve, RTR outer loop, preconditioner
//                   selection (DPGO_PRECOND_AUTO), concurrent / begin-end solves
//   agents.hip      GNC re-weighting, initial guesses, manifold operations, public-pose exchange plans
//   bench_probes.hip  kernel timing probes used by bench.py
#pragma once
#include "kernels.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <pthread.h>
#include <sched.h>
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

namespace dpgo_host {


inline thread_local std::string g_err;

inline int fail(int code, const std::string& msg) {
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

inline bool supported(int d, int r) {
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

inline int free_bsr(Bsr& m) {
  if (m.rowptr) HIPC(hipFree(m.rowptr));
  if (m.colidx) HIPC(hipFree(m.colidx));
  if (m.vals) HIPC(hipFree(m.vals));
  m = Bsr();
  return DPGO_OK;
}



// Every tuning / A-B switch of the library in ONE place: read from the environment once (first use; dpgo_options_reload
// reads again), printed by dpgo_describe_options / dpgo_problem_describe.  -1 (or 0 where noted) = not set: the size rules
// decide.  The kernel-selecting ones are exercised by tests/test_parity_gpu.py::test_kernel_selecting_switches_*.
//        field              variable                  unset  meaning
#define DPGO_OPTIONS(X)                                                                                                  \
  X(split,             "DPGO_SPLIT",              0,  "lane groups per pose of the SpMM-family kernels: 1, 2, 4 (0: by size)")       \
  X(spmm_symmetric,    "DPGO_SPMM_SYMMETRIC",    -1,  "symmetric storage of Q for blocks beyond the Infinity Cache: 0 / 1")          \
  X(stream_nt,         "DPGO_STREAM_NT",         -1,  "non-temporal single-use operands in the tCG-step kernels: 0 / 1")             \
  X(outer_sym,         "DPGO_OUTER_SYM",          1,  "outer RTR iteration (k_grad / k_hess) reads the symmetric copy when tCG does") \
  X(tile_walk,         "DPGO_TILE_WALK",          1,  "symmetric-storage kernels walk each XCD's tiles breadth-first over the tile graph (0: index order)") \
  X(iter_graph,        "DPGO_ITER_GRAPH",         0,  "steady tCG iterations replayed from an instantiated hipGraph (measured slower)") \
  X(tcg_ahead,         "DPGO_TCG_AHEAD",          0,  "iterations the just-in-time feed stays ahead (0: 2 multilevel / 4 otherwise)") \
  X(grid_update,       "DPGO_GRID_UPDATE",        0,  "launch cap of k_tcg_update (0: resident count)")                              \
  X(grid_hess,         "DPGO_GRID_HESS",          0,  "launch cap of k_tcg_hess (0: resident count)")                                \
  X(grid_hess_sym,     "DPGO_GRID_HESS_SYM",      0,  "launch cap of k_tcg_hess_sym (0: resident count)")                            \
  X(grid_retract,      "DPGO_GRID_RETRACT",       0,  "launch cap of k_retract (0: 1024)")                                           \
  X(grid_outer_sym,    "DPGO_GRID_OUTER_SYM",     0,  "launch cap of k_grad / k_hess on the symmetric storage (0: resident count)")  \
  X(grid_spmm_sym,     "DPGO_GRID_SPMM_SYM",      0,  "launch cap of k_spmm_sym (0: resident count, at most 1024)")                  \
  X(grid_ml,           "DPGO_GRID_ML",            0,  "launch cap of the level-0 restriction / post-smoothing (0: resident count)")  \
  X(persist,           "DPGO_PERSIST",           -1,  "one-launch solve (k_rtr_persist) off / on whatever the size: 0 / 1")          \
  X(persist_max_poses, "DPGO_PERSIST_MAX_POSES",  0,  "largest block the one-launch solve takes (0: every block it can hold)")       \
  X(persist_split,     "DPGO_PERSIST_SPLIT",      0,  "lane groups per pose of the one-launch solve: 1, 4 (0: by size)")             \
  X(persist_mt,        "DPGO_PERSIST_MT",         0,  "tiles per workgroup of the one-launch solve: 1, 2 (0: by size)")              \
  X(poll_first,        "DPGO_POLL_FIRST",        -1,  "s_sleep units before the first sweep of the in-kernel all-reduce")            \
  X(poll_sleep,        "DPGO_POLL_SLEEP",        -1,  "s_sleep units between sweeps of the in-kernel all-reduce")                    \
  X(poll_first_pay,    "DPGO_POLL_FIRST_PAY",    -1,  "the same before the first sweep of a reduction that carries a payload")       \
  X(persist_verbose,   "DPGO_PERSIST_VERBOSE",    0,  "per-solve phase report of the one-launch solve on stderr")                    \
  X(hess_dma,          "DPGO_HESS_DMA",           0,  "k_tcg_hess_sym's own tiles by LDS-DMA: 1 = double-buffered, 2 waves / SIMD, 4 blocks in flight; 2 = 3 waves, 2 blocks") \
  X(setup_timing,      "DPGO_SETUP_TIMING",       0,  "section times of the hierarchy's symbolic set-up on stderr")                  \
  X(setup_threads,     "DPGO_SETUP_THREADS",      0,  "host threads of the hierarchy's symbolic set-up (0: min(8, cores); 1: serial)") \
  X(setup_pin,         "DPGO_SETUP_PIN",          1,  "set-up worker threads placed in the CPU group of the thread that first used them") \
  X(auto_cost_rule,    "DPGO_AUTO_COST_RULE",     1,  "DPGO_PRECOND_AUTO on coupled blocks: cost rule (0: tCG-budget hysteresis only)") \
  X(ml_graph,          "DPGO_ML_GRAPH",           1,  "graph aggregates in the default hierarchy (0: index runs)")                   \
  X(ml_graph_size,     "DPGO_ML_GRAPH_SIZE",      0,  "growth size of the default graph aggregates (0: by size)")                    \
  X(ml_growth_chunks,  "DPGO_ML_GROWTH_CHUNKS",   0,  "index ranges the graph aggregates grow and merge in (0: 8 from 65 536 poses, else 1)") \
  X(ml_ap,             "DPGO_ML_AP",              1,  "two-level post-smoothing through A P (0: gather through Q; index runs only)") \
  X(ml_dense_sym,      "DPGO_ML_DENSE_SYM",      -1,  "dense level from the packed lower triangle on the matrix cores: 0 / 1")       \
  X(ml_early_stop,     "DPGO_ML_EARLY_STOP",      1,  "tCG's residual test in the restriction kernel, one kernel early")             \
  X(ml_operator_bits,  "DPGO_ML_OPERATOR_BITS",   0,  "level-0 operator copies of the cycle on HBM-bound blocks: 32 / 64 (0: 32)")   \
  X(ml_vector_bits,    "DPGO_ML_VECTOR_BITS",     0,  "cycle-internal vectors (pre-smoothed iterate, kept residual) beside fp32 operator copies: 32 / 64 (0: 32)") \
  X(ml_dense_bits,     "DPGO_ML_DENSE_BITS",      0,  "dense level beside fp32 cycle vectors: 32 / 64 (0: 64 -- fp32 measured neutral)") \
  X(ml_setup_serial,   "DPGO_ML_SETUP_SERIAL",    0,  "one-thread-per-aggregate set-up kernels of round 3")                          \
  X(gj_mfma,           "DPGO_GJ_MFMA",            1,  "rank-64 updates of the dense inverse on the fp64 matrix cores")               \
  X(dense_chunk,       "DPGO_DENSE_CHUNK",        0,  "tiles per workgroup of k_dense_sym_apply (0: default)")                       \
  X(coarse_nodes,      "DPGO_COARSE_NODES",       0,  "nodes per workgroup of k_ml_coarse_prolong: 1-4 (0: by size)")                \
  X(coarse_grid,       "DPGO_COARSE_GRID",        0,  "launch cap of k_ml_coarse_prolong (0: default)")                              \
  X(coarse_nt,         "DPGO_COARSE_NT",         -1,  "non-temporal loads of the dense inverse: 0 / 1")
struct Options {
#define X(field, var, unset, text) int field = unset;
  DPGO_OPTIONS(X)
#undef X
};
inline Options& options_storage() {
  static Options o;
  return o;
}
inline void options_read(Options& o) {
  o = Options();
#define X(field, var, unset, text) \
  if (const char* e_ = std::getenv(var)) o.field = (*e_ == 0) ? 1 : std::atoi(e_);
  DPGO_OPTIONS(X)
#undef X
}
inline const Options& options() {
  static const bool once = (options_read(options_storage()), true);
  (void)once;
  return options_storage();
}
inline std::string options_describe() {
  const Options& o = options();
  std::string s;
#define X(field, var, unset, text) s += std::string(var) + "=" + std::to_string(o.field) + (o.field == (unset) ? "" : " [set]") + "  # " + text + "\n";
  DPGO_OPTIONS(X)
#undef X
  return s;
}

}  // namespace dpgo_host
using namespace dpgo_host;

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
    float *Pb32 = nullptr, *AP32 = nullptr;  // fp32 copies of Pb and of A P's values (level 0, ml_operator_bits == 32)
    float *x1f = nullptr, *res1f = nullptr;  // ... and the cycle-internal vectors of level 0 in that storage: the
                                             // pre-smoothed iterate (k_tcg_update -> k_ml_restrict), the kept residual
                                             // (k_ml_restrict -> k_ml_post_ap)
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
    const int env = options().ml_dense_sym;
    if (!ml_use_ap() || ml_coarse_bits != 64 || !ml_packed || env == 0) return false;
    return env == 1 || ml_lda >= 3072;  // below, the row-streaming kernel (one launch, cache-resident inverse) is as fast
  }
  int ml_coarse_bits = 64;  // 32: opt-in (dpgo_problem_multilevel_coarse_bits)
  // Storage precision of the OPERATOR COPIES the V-cycle streams on level 0 of an HBM-bound block (symmetric storage, two
  // levels): Q's values in the restriction's residual r - A x1, the values of A P in the post-smoothing, the prolongation
  // blocks in both -- 32: fp32 copies beside the fp64 originals (the Hessian step, the set-up and every product and sum
  // stay fp64; the cycle is a preconditioner) -- and, with them, the two vectors that live INSIDE a cycle: the
  // pre-smoothed iterate x1 and the kept residual res1 (written once, read once per application).  DEFAULT since round 5 (100k poses: restriction 34.1 -> 29.2 us, post-smoothing
  // 25.7 -> 23.9 us, 152 -> 148 us per product, same product counts; 64 restores the fp64 originals).  ml_ops32_ready: the
  // copies hold the current values.
  int ml_operator_bits = 32;
  bool ml_ops32_ready = false;
  bool ml_ops32_suspend = false;  // set around a stand-alone application of the cycle (its pre-smoothing kernel writes fp64)
  bool ml_ops32_wanted() const { return ml_operator_bits == 32 && tcg_sym && split == 1 && ml_use_ap() && sym.uvalsT != nullptr; }
  bool ml_ops32_active() const { return !ml_ops32_suspend && ml_ops32_wanted() && ml_ops32_ready; }
  bool ml_vec32_active() const { return ml_ops32_active() && options().ml_vector_bits != 64; }  // (x1 / res1 in fp32 too)
  // the dense level (the inverse of the coarsest operator and the restricted residual it multiplies) streamed in fp32: by
  // request per handle (ml_coarse_bits == 32) or, with the rest of the cycle's fp32 storage, by DPGO_ML_DENSE_BITS=32 --
  // unless the level runs on the packed-triangle matrix-core kernels, which exist in fp64 only.  NOT the default: at 100k
  // poses the dense kernel itself gets faster (11.8 -> 9.3 us), the loop does not (136.0 against 135.6 us per product, three
  // interleaved pairs: the 19 MB it frees in the Infinity Cache change nothing the other kernels notice).
  bool coarse32_active() const {
    return ml_coarse_bits == 32 || (ml_vec32_active() && !ml_use_dense_sym() && options().ml_dense_bits == 32);
  }
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
    const bool off = options().ml_ap == 0;
    return ml.size() == 2 && ml[0].AP.vals != nullptr && (!off || ml[0].graph);  // (graph aggregates exist in this form only)
  }
  // symmetric copy of Q for the plain SpMM on Infinity-Cache-cold blocks (k_spmm_sym): upper blocks transposed + lower references
  struct SymQ {
    int nu = 0, nl = 0;
    int32_t *urow = nullptr, *ucol = nullptr, *usrc = nullptr, *lrow = nullptr, *lcol = nullptr, *lslot = nullptr,
            *lsrc = nullptr;
    double* uvalsT = nullptr;
    float* uvalsT32 = nullptr;  // fp32 copy of the values (the cycle's level-0 restriction, ml_operator_bits == 32)
    int32_t* tord = nullptr;    // walk over the workgroup tiles (BsrSymDevT::tord; NULL: index order)
    int* flag = nullptr;       // device: set by k_sym_check when a lower block is not the transpose of its upper one
    bool symbolic = false;     // pattern arrays belong to the current block pattern
    bool pattern_ok = false;   // the pattern is structurally symmetric
    bool ready = false;        // uvalsT holds the current values and they passed the symmetry check
    bool values_ok = false;
    BsrSymDev dev() const { return BsrSymDev{urow, ucol, uvalsT, lrow, lcol, lslot, tord}; }
    BsrSymDev32 dev32() const { return BsrSymDev32{urow, ucol, uvalsT32, lrow, lcol, lslot, tord}; }
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
    if (options().spmm_symmetric >= 0) return options().spmm_symmetric != 0;
    return beyond_cache();
  }
  // what the tCG loop streams besides Q and the pose vectors (the multilevel cycle's dense inverse, A P, prolongation):
  // set by the solve that last chose a preconditioner
  size_t loop_extra_bytes = 0;
  // non-temporal single-use operands: when the launch is fed from HBM (same size rule as the symmetric storage)
  bool want_stream_nt() const {
    return options().stream_nt >= 0 ? options().stream_nt != 0 : beyond_cache();
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
  bool pctrl_dirty = true;  // the control block has to be cleared in front of the next one-launch solve (creation, after a time-out)
  unsigned long long* pgran = nullptr;  // granule table of the in-kernel all-reduce (kGranWords 8-byte words)
  unsigned gran_cleared_at = 0;         // value of `gen` when the table was last cleared
  PersistCtrl* hctrl = nullptr;  // pinned
  double* partials = nullptr;  // 5 regions of kPartialCap*kNP
  DevState* dstate = nullptr;  // 2 slots
  DevState* hstate = nullptr;  // pinned
  unsigned long long* hflag = nullptr;  // pinned, host-coherent: device-published tCG progress word
  unsigned gen = 0;
  bool saw_rtr_stop = false;  // set from the progress word in just-in-time mode
  // One STEADY tCG iteration (j >= 1: Hessian step, update, and the V-cycle's launches when that is the preconditioner)
  // of the multi-launch scheme as an instantiated hipGraph, replayed by the just-in-time feed instead of 2-6 stream
  // launches (tools/launch_lab.hip: the boundary between two dependent kernels is 3.4 us on a stream, 1.6 us inside a
  // graph).  One per parity of the state slot the iteration starts from; `key` = hash of everything the launches read
  // from the handle (iter_graph_key, solve.hip): a graph is re-captured when it no longer matches.
  struct IterGraph {
    hipGraphExec_t exec = nullptr;
    unsigned long long key = 0;
  } iter_graph[2];
  bool capturing = false;          // launches are being recorded: they pass generation 0 (kernels/common.h, state_gen)
  bool iter_graph_failed = false;  // capture / instantiation / launch failed once: this handle keeps plain launches
  unsigned launch_gen() const { return capturing ? 0u : gen; }
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
  // k_tcg_update_span's multilevel-mode instance (no iterate, no projection: 143 VGPRs = 3 waves per SIMD) has its own cap
  int cap_u_ml = kMaxGrid;
  int grid_u(bool ml_mode) const {
    const int P = (64 / b) * kWaves;
    const int tiles = std::max(1, (n + P - 1) / P);
    return std::min(tiles, ml_mode ? cap_u_ml : cap_u);
  }
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
  // the outer iteration's kernels on the symmetric storage (k_grad, k_hess: 110-114 VGPRs = 4 waves per SIMD) have their own
  // cap -- grid_s() is sized for the tCG-step kernel's 2 waves per SIMD -- and whoever sums their partials is told the
  // grid of the launch that wrote them
  int cap_outer_sym = kMaxGrid;
  int nb_grad = 0, nb_hess = 0;
  int grid_outer_sym() const {
    const int P = (64 / b) * kWaves;
    const int tiles = std::max(1, (n + P - 1) / P);
    return std::min(tiles, cap_outer_sym);
  }
  int cap_spmm_sym = kMaxGrid;  // launch cap of k_spmm_sym (resident count of the compiled kernel)
  int grid_spmm_sym() const {
    const int P = (64 / b) * kWaves;
    const int tiles = std::max(1, (n + P - 1) / P);
    return std::min(tiles, cap_spmm_sym);
  }
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

namespace dpgo_host {

struct Counters {
  int spmm = 0;
  bool vcycle_for_additive = false;  // an outer iteration of an "additive" solve ran the V-cycle instead
};

template <class Tp>
inline int upload(Tp** dst, const Tp* src, size_t count, hipStream_t s) {
  HIPC(hipMalloc(dst, sizeof(Tp) * (count > 0 ? count : 1)));
  if (count > 0) HIPC(hipMemcpyAsync(*dst, src, sizeof(Tp) * count, hipMemcpyHostToDevice, s));
  return DPGO_OK;
}

// Host worker threads of the set-up code (the hierarchy's symbolic set-up): created once per process and kept -- in a
// process that has the HIP runtime and an ML framework loaded pthread_create costs 0.2-0.4 ms (static TLS of every loaded
// library), more than the sections it would run.  A batch = fn(0 .. count-1); whoever waits for a batch executes its items
// too, so batches may be started from inside items (nested sections) without a deadlock.
class TaskPool {
 public:
  struct Batch {
    std::function<void(int)> fn;
    int count = 0;
    std::atomic<int> next{0}, done{0};
  };
  using Job = std::shared_ptr<Batch>;
  static TaskPool& get() {
    static TaskPool* pool = new TaskPool();  // never destroyed: the workers are detached
    return *pool;
  }
  // starts fn(0 .. count-1) on the workers; the caller goes on and later calls wait()
  Job submit(int count, std::function<void(int)> fn, int max_threads) {
    Job b = std::make_shared<Batch>();
    b->fn = std::move(fn);
    b->count = count;
    const int helpers = std::max(0, std::min(count, max_threads - 1));
    if (helpers > 0) {
      {
        std::lock_guard<std::mutex> lk(mu_);
        while ((int)nworkers_ < std::min(kMaxWorkers, std::max(helpers, (int)nworkers_))) {
          const int id = (int)nworkers_;
          std::thread([this, id] {
            pin(id);
            loop();
          }).detach();
          ++nworkers_;
        }
        for (int k = 0; k < helpers; ++k) queue_.push_back(b);
      }
      cv_.notify_all();
    }
    return b;
  }
  // executes what is left of the batch, then waits for the items other threads are still running
  void wait(const Job& b) {
    if (!b) return;
    help(*b);
    while (b->done.load(std::memory_order_acquire) < b->count) std::this_thread::yield();
  }
  void run(int count, const std::function<void(int)>& fn, int max_threads) {
    if (count <= 0) return;
    if (count == 1 || max_threads <= 1) {
      for (int k = 0; k < count; ++k) fn(k);
      return;
    }
    wait(submit(count, fn, max_threads));
  }
  // creates the workers ahead of their first use, from a helper thread (the caller pays one pthread_create)
  void warm(int threads) {
    bool expected = false;
    if (threads <= 1 || !warmed_.compare_exchange_strong(expected, true)) return;
    std::thread([this, threads] { wait(submit(threads - 1, [](int) {}, threads)); }).detach();
  }

 private:
  static constexpr int kMaxWorkers = 15;
  // Workers next to the thread that created the pool (DPGO_SETUP_PIN=0: wherever the scheduler puts them): the sections
  // they run share arrays of a few MB that thread has just written -- on a multi-socket host a worker on another socket
  // (or another L3 slice) reads them across the fabric.  Heuristic: the aligned group of 8 consecutive CPU numbers around
  // the creator's CPU (one core complex on current server parts), restricted to the process's affinity mask; a worker may
  // run on any CPU of the group.  Purely a placement hint: failures are ignored.
  int home_cpu_ = -1;
  void pin(int) {
    if (!pin_enabled() || home_cpu_ < 0) return;
    cpu_set_t allowed, want;
    CPU_ZERO(&allowed);
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return;
    const int lo = home_cpu_ & ~7;
    int cnt = 0;
    for (int c = lo; c < lo + 8; ++c)
      if (c < CPU_SETSIZE && CPU_ISSET(c, &allowed)) CPU_SET(c, &want), ++cnt;
    if (cnt >= 2) (void)pthread_setaffinity_np(pthread_self(), sizeof(want), &want);
  }
  static bool pin_enabled() { return options().setup_pin != 0; }
  TaskPool() { home_cpu_ = sched_getcpu(); }
  static void help(Batch& b) {
    for (;;) {
      const int k = b.next.fetch_add(1, std::memory_order_relaxed);
      if (k >= b.count) return;
      b.fn(k);
      b.done.fetch_add(1, std::memory_order_release);
    }
  }
  void loop() {
    for (;;) {
      Job b;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return !queue_.empty(); });
        b = std::move(queue_.front());
        queue_.erase(queue_.begin());
      }
      help(*b);
    }
  }
  std::mutex mu_;
  std::condition_variable cv_;
  std::vector<Job> queue_;
  unsigned nworkers_ = 0;
  std::atomic<bool> warmed_{false};
};
// waits for a job when the scope is left, whichever way
struct JobGuard {
  TaskPool::Job job;
  ~JobGuard() { TaskPool::get().wait(job); }
};

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

template <typename T>
inline int sym_upload(T** dst, const std::vector<T>& v, hipStream_t stream) {
  HIPC(hipMalloc(dst, sizeof(T) * std::max<size_t>(1, v.size())));
  if (!v.empty()) HIPC(hipMemcpyAsync(*dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice, stream));
  return DPGO_OK;
}

struct PersistGeo {
  int split = 0, mt = 0, wgs = 0, slots = 0;
};

enum { RUN_FULL = 0, RUN_BEGIN = 1, RUN_END = 2 };  // run_optimize phases (solve.hip)

// ---- problem.hip
int set_device(dpgo_problem_s* p);
int upload_bsr(Bsr& m, int nrows, int ncols, int nnzb, int b, const int32_t* rowptr, const int32_t* colidx,
               const double* vals, hipStream_t s);
int validate_bsr(int nrows, int ncols, int nnzb, const int32_t* rowptr, const int32_t* colidx, bool need_diag);
int build_dinv(dpgo_problem_s* p, double shift);
int poll_state(dpgo_problem_s* p);
int push_state(dpgo_problem_s* p);
void sym_free(dpgo_problem_s* p);
int sym_symbolic_setup(dpgo_problem_s* p);
int sym_ensure(dpgo_problem_s* p, bool* usable);
int launch_spmm_sym(dpgo_problem_s* p, const BsrSymDev& M, const double* V, const double* Gadd, double* OUT);
int launch_spmm(dpgo_problem_s* p, const Bsr& M, const double* V, const double* Gadd, double* OUT, int nrows = -1);
bool outer_sym_enabled();
int launch_grad(dpgo_problem_s* p, const double* X, double* RG, double* S, double* EG,
                const DevState* st = nullptr, bool sym = false);
int launch_hess(dpgo_problem_s* p, const double* X, const double* S, const double* V, const double* Gdot,
                double* HV, double* partials, const DevState* st, int check_tcg, bool sym = false);
int launch_retract(dpgo_problem_s* p, const double* X, const double* eta, double scale, double* X2,
                   const DevState* st);
int launch_rtr_update(dpgo_problem_s* p);
int launch_precond(dpgo_problem_s* p, const double* X, const double* V, const double* dinv, double* Z);
int launch_rtr_begin(dpgo_problem_s* p, double tol, double Delta0, double Dmax, int max_inner, int tiny);
int check_ready(dpgo_problem_s* p);
int h2d(dpgo_problem_s* p, double* dst, const double* src);
int d2h(dpgo_problem_s* p, double* dst, const double* src);

// ---- multilevel.hip
int ml_tile(int b, int split);
int additive_tile(const dpgo_problem_s* p);
int ml_level_split(int n);
int ml_default_graph_size(int n, int b);
std::vector<int> ml_default_ks(int n, int b, int split0);
void ml_free(dpgo_problem_s* p);
int ml_graph_aggregates(const std::vector<int32_t>& rowptr, const std::vector<int32_t>& colidx, int n, int S,
                        std::vector<int32_t>& lab, std::vector<int32_t>& ptr, std::vector<int32_t>& mem,
                        std::vector<int32_t>& parent, std::vector<int32_t>& pslot);
int ml_merge_small_aggregates(const std::vector<int32_t>& rowptr, const std::vector<int32_t>& colidx, int n, int S, int cap,
                              std::vector<int32_t>& lab, std::vector<int32_t>& ptr, std::vector<int32_t>& mem,
                              std::vector<int32_t>& parent, std::vector<int32_t>& pslot);
int ml_symbolic_setup(dpgo_problem_s* p, const std::vector<int>& ks_in, int perm_tile = 0);
int setup_threads();  // host threads of the symbolic set-up (DPGO_SETUP_THREADS)
int flat_grid(size_t items);
bool gj_use_mfma();
int dense_spd_inverse(hipStream_t s, double* M, int lda, double* W, double* Rx, bool mfma);
int ml_numeric_setup(dpgo_problem_s* p);
std::vector<int> ml_current_ks(const dpgo_problem_s* p);
const dpgo_problem_s::AddPlan& additive_plan(dpgo_problem_s* p);
int additive_split_of(const dpgo_problem_s* p);
int ml_ensure(dpgo_problem_s* p, double shift, bool additive = false);
int ml_ops32_ensure(dpgo_problem_s* p);
int persist_capacity(int device);  // (two resident slots per CU; below)
int launch_coarse_prolong(dpgo_problem_s* p, const dpgo_problem_s::MlLevel& L, const dpgo_problem_s::MlLevel& C,
                          const DevState* gate, double* xc_out = nullptr);
int launch_dense_sym(dpgo_problem_s* p, const dpgo_problem_s::MlLevel& C, const DevState* gate);
int launch_ml_restrict0(dpgo_problem_s* p, const double* r, const DevState* gate, int g0, bool stop_check = false);
int launch_ml_post_ap(dpgo_problem_s* p, const double* Xdev, const double* r, double* z, double* pout, const DevState* gate);
int launch_ml_tail(dpgo_problem_s* p, const double* Xdev, const double* r, double* z, double* pout,
                   const DevState* gate, bool stop_check = false);
int launch_ml_apply(dpgo_problem_s* p, const double* Xdev, const double* v, double* z);

// ---- solve.hip
int launch_tcg_update(dpgo_problem_s* p, const double* dinv, int first, double* z_out = nullptr,
                      double ml_omega = 0.0);
int launch_tcg_hess(dpgo_problem_s* p, int first);
int launch_tcg_hess_with(dpgo_problem_s* p, const DevState* sin, DevState* sout, int first, unsigned long long* hflag,
                         unsigned gen);
int resolve_tcg_storage(dpgo_problem_s* p);
int persist_capacity(int device);
bool persist_reserve(dpgo_problem_s* p, int slots, int limit);
void persist_release(dpgo_problem_s* p);
bool additive_available(dpgo_problem_s* p);
PersistGeo persist_geometry(const dpgo_problem_s* p, int free_slots, int share = 1, bool additive = false);
int launch_rtr_persistent(dpgo_problem_s* p, const dpgo_ropt_params* prm, const double* dinv, bool* used, bool additive);
void persist_report(dpgo_problem_s* p);
int rtr_outer_iteration(dpgo_problem_s* p, const dpgo_ropt_params* prm, const double* dinv, Counters& cnt,
                        bool poll_at_end);
int auto_units_jacobi(dpgo_problem_s* p);
int auto_units_additive(dpgo_problem_s* p);
void auto_update(dpgo_problem_s* p, const dpgo_ropt_params* prm, int used, int products);
int run_optimize(dpgo_problem_s* p, const dpgo_ropt_params* prm, dpgo_ropt_result* res, int phase = RUN_FULL);
int tune_launch_caps(dpgo_problem_s* p);
int tune_persist(dpgo_problem_s* p);

// ---- agents.hip
int free_edges(dpgo_problem_s* p);
int rebuild_vals(dpgo_problem_s* p, int nnzb, const int32_t* cptr, const int32_t* cedge, const uint8_t* ckind,
                 const double* base, double sign, double* out);
int rebuild_Q_from_weights(dpgo_problem_s* p, const double* base, double sign, double* out);
int rebuild_C_from_weights(dpgo_problem_s* p, const double* base, double sign, double* out);
int refresh_after_weights(dpgo_problem_s* p);

}  // namespace dpgo_host
