/* dpgo_hip.h -- C ABI of the MI355X-native RBCD local solver (libdpgo_hip.so).
 *
 * Drop-in boundary for the per-agent local solve of mit-acl/dpgo: what
 * PGOAgent::updateX (reference src/PGOAgent.cpp:961-986) does through
 * QuadraticProblem + QuadraticOptimizer + LiftedSEManifold.  Every entry point cites
 * the reference interface it replaces.  Plain pointers and sizes only; no torch / Eigen
 * types.  The reference-side binding a maintainer would add is in INTEGRATION.md.
 *
 * Matrix layout (all dense matrices): the reference's Eigen::MatrixXd r x (d+1)n,
 * COLUMN-major (include/DPGO/manifold/Poses.h:16-21; pinned by tests/testEigenMap.cpp)
 * = n consecutive pose tiles of (d+1)*r doubles; tile i = [Y_i (r x d) | p_i (r)].
 *
 * Q layout: block-CSR with (d+1)x(d+1) blocks; vals[t] is the dense block
 * Q[i*b:(i+1)*b, j*b:(j+1)*b] stored ROW-major (scipy.sparse.bsr_matrix layout); Q is
 * symmetric, so this is also the column-major block of Q's column i.  int32 indices, as
 * Eigen::SparseMatrix<double,RowMajor> (include/DPGO/DPGO_types.h:26).
 *
 * Error convention: every function returns DPGO_OK (0) or a positive error code and
 * never aborts (the reference's glog CHECKs become DPGO_ERR_INVALID);
 * dpgo_last_error() returns a thread-local message.  There is NO CPU fallback: without
 * a HIP device every compute entry point returns DPGO_ERR_HIP.
 *
 * Threading: one solve at a time per handle (as PGOAgent::updateX holds its mutexes for
 * the whole call, src/PGOAgent.cpp:940-942); distinct handles are independent (one HIP
 * stream each, no global mutable state).
 */
#ifndef DPGO_HIP_H
#define DPGO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPGO_OK 0
#define DPGO_ERR_INVALID 1     /* bad argument / shape mismatch (reference: glog CHECK) */
#define DPGO_ERR_HIP 2         /* HIP runtime error or no device */
#define DPGO_ERR_UNSUPPORTED 3 /* (d, r) combination not compiled in */
#define DPGO_ERR_STATE 4       /* e.g. solve requested before Q was set */

/* ROptParameters::ROptMethod (include/DPGO/DPGO_types.h:47-52) */
#define DPGO_METHOD_RTR 0
#define DPGO_METHOD_RGD 1

/* Preconditioner applied by QuadraticProblem::PreConditioner's replacement.
 * The reference factorises Q + 0.1 I with CHOLMOD (src/PoseGraph.cpp:598-613); the
 * device path applies the inverse of the (d+1)x(d+1) diagonal blocks of Q + shift I
 * (same fixed point, different tCG trajectory). */
#define DPGO_PRECOND_NONE 0
#define DPGO_PRECOND_BLOCK_JACOBI 1
#define DPGO_PRECOND_MULTILEVEL 2 /* aggregation-multigrid V-cycle for Q + shift I (built on the device) */
/* DEFAULT.  A multilevel preconditioner whenever it pays, block-Jacobi otherwise, decided per handle from the solves
 * themselves (a function of their product counts only, so repeated runs reproduce; every change of Q starts over):
 *  - a block WITHOUT coupling to other agents runs multilevel from its first solve and never hands back;
 *  - a coupled block the additive one-launch solve can hold (dpgo_problem_additive_plan: up to ~14 000 poses in 3-D)
 *    follows a COST RULE, in units of a tenth of a block-Jacobi product (dpgo_auto_rule_constants): Q -- hence the
 *    hierarchy -- is constant across RBCD sweeps, so its set-up (2 800 units = 280 block-Jacobi products: 2.8-3.0 ms
 *    against 10 us) is paid once.  When the block-Jacobi solves since Q last changed have cost as much as one set-up
 *    (and the last one ran >= 6 products), or one of them used >= half its tCG budget, the next solve runs additive on
 *    trial; it stays additive while its products x 13 stay below the reference block-Jacobi solve's x 10 (round 6: 11.0 us
 *    against 9.6 us per in-kernel iteration on a 12 500-pose slab, 8.7 against 6.6 on sphere2500; 18 units before the
 *    additive iteration lost its third reduction), and hands back otherwise -- the hierarchy is kept, the next trial waits
 *    for twice the work.
 *    Handles solved next to others of one device are charged for the part of the chip their launch blocks
 *    (dpgo_problem_auto_info): there the additive form -- one CU per aggregate -- has to need ~3x fewer products;
 *  - any other coupled block: a block-Jacobi solve that used >= half of its tCG budget (RTR_iterations x
 *    RTR_tCG_iterations products) makes the next solves multilevel (the V-cycle: ~3x a block-Jacobi iteration, 4-7x fewer of
 *    them when the budget binds); a multilevel solve that needed <= a tenth of the budget hands back.
 * In multi-agent RBCD far from the optimum the trust-region boundary and the coupling, not the preconditioner, end the
 * local solves: there the trial fails and block-Jacobi stays.  dpgo_ropt_result::precond_used says
 * what a call ran, dpgo_problem_auto_info where the rule stands.  DPGO_AUTO_COST_RULE=0 in the environment: budget
 * hysteresis only (the rule of earlier versions). */
#define DPGO_PRECOND_AUTO 3
/* Additive two-level preconditioner  z = proj_X( Dinv r + P A_c^-1 P^T r )  (block-Jacobi plus the coarse-grid correction
 * of the residual; same tree prolongations and Galerkin coarse operator as the multilevel cycle; ONE aggregate per
 * workgroup of the one-launch solve: graph aggregates of at most 16 (3-D) / 20 (2-D) poses while 256 of them cover the
 * block, beyond that -- up to ~14 000 poses in 3-D -- of at most 64 / 84 poses with the growth's fragments merged,
 * dpgo_problem_additive_plan): nothing inside it applies an operator to a distributed vector, so a whole preconditioned
 * tCG iteration runs inside the persistent kernel (three in-kernel reductions).  Where the persistent kernel cannot run
 * (larger blocks: DPGO_ERR_UNSUPPORTED; no free resident slots or a time-out: silently) the solve uses the multilevel
 * V-cycle on the same hierarchy instead.  What DPGO_PRECOND_AUTO selects for such blocks when it selects a multilevel
 * preconditioner. */
#define DPGO_PRECOND_ADDITIVE 4

/* tCG termination status; replaces ROPTLIB::tCGstatusSet in ROPTResult
 * (include/DPGO/DPGO_types.h:106). */
#define DPGO_TCG_NEGCURVTURE 0
#define DPGO_TCG_EXCREGION 1
#define DPGO_TCG_LCON 2
#define DPGO_TCG_SCON 3
#define DPGO_TCG_MAXITER 4

/* Mirrors DPGO::ROptParameters (include/DPGO/DPGO_types.h:44-86); the fields after
 * RTR_initial_radius are extensions with reference-compatible defaults set by
 * dpgo_ropt_params_default(). */
typedef struct dpgo_ropt_params {
  int method;                 /* DPGO_METHOD_*            default RTR  */
  int verbose;                /*                          default 0    */
  double gradnorm_tol;        /*                          default 1e-2 */
  double RGD_stepsize;        /*                          default 1e-3 */
  int RGD_use_preconditioner; /*                          default 1    */
  int RTR_iterations;         /*                          default 3    */
  int RTR_tCG_iterations;     /*                          default 50   */
  double RTR_initial_radius;  /*                          default 100  */
  /* --- extensions --- */
  int precond;                /* DPGO_PRECOND_*           default AUTO */
  double precond_shift;       /* reference: 1e-1 (src/PoseGraph.cpp:603) */
  int accept_tiny_decrease;   /* ROPTLIB's second acceptance clause (SURVEY 8c' item 5), default 1 */
  int tcg_poll_interval;      /* 0 (default): just-in-time kernel feed driven by the progress word the
                                 device publishes into host-coherent memory (no synchronisation);
                                 k > 0: enqueue k tCG iterations, then synchronise and poll */
  double time_bound_s;        /* ROPTLIB Solver.TimeBound = 5.0 (src/QuadraticOptimizer.cpp:78) */
} dpgo_ropt_params;

/* Mirrors DPGO::ROPTResult (include/DPGO/DPGO_types.h:91-107) + counters. */
typedef struct dpgo_ropt_result {
  int success;
  double fInit, gradNormInit, fOpt, gradNormOpt, elapsedMs;
  int tCGStatus;          /* DPGO_TCG_* of the last tCG run */
  int rtr_iterations;     /* outer iterations executed */
  int rtr_accepted;       /* outer iterations accepted */
  int tcg_iterations;     /* Hessian-vector products inside tCG (all outer iterations) */
  int spmm_count;         /* Q*X block-SpMM launches in this call */
  int latest_step_accepted;
  int precond_used;       /* DPGO_PRECOND_* this call ran (what DPGO_PRECOND_AUTO resolved to) */
} dpgo_ropt_result;

typedef struct dpgo_problem_s* dpgo_problem_t;

/* ---- library ---- */
const char* dpgo_version(void);
const char* dpgo_last_error(void);
int dpgo_device_count(int* count);
void dpgo_ropt_params_default(dpgo_ropt_params* p);
/* 1 if the (d, r) pair has compiled kernels */
int dpgo_supported(int d, int r);

/* Every tuning / A-B switch the library reads from the environment (DPGO_*; read once at first use), one
 * "NAME=value  # meaning" line each ("[set]" marks the ones the environment overrides); dpgo_options_reload reads the
 * environment again (tools).  dpgo_problem_describe: what a handle currently runs -- layout, storage, one-launch solve,
 * preconditioner selection, hierarchy -- followed by the same list.  Text is truncated to capacity - 1 characters. */
int dpgo_describe_options(char* out, int capacity);
int dpgo_options_reload(void);

/* ---- problem lifecycle: replaces QuadraticProblem(shared_ptr<PoseGraph>)
 * (include/DPGO/QuadraticProblem.h:39) + the data PoseGraph caches for it
 * (Q_, G_, precon_: include/DPGO/PoseGraph.h:324-331).  The handle owns device copies of
 * Q (BSR), G, the block-Jacobi factors and all solver work vectors; create it once per
 * PoseGraph and keep it across iterations (the reference rebuilds problem+optimizer per
 * iteration, src/PGOAgent.cpp:968-969, which is free on the CPU but not on a device). */
int dpgo_problem_create(dpgo_problem_t* out, int r, int d, int n, int device);
int dpgo_problem_destroy(dpgo_problem_t h);
int dpgo_problem_describe(dpgo_problem_t h, char* out, int capacity);
/* Run this handle's work on an external HIP stream, e.g. torch's current stream, so that it is ordered
 * with the caller's own device work.  hip_stream = NULL selects the device's DEFAULT (null) stream --
 * which is what torch.cuda.current_stream().cuda_stream is unless the caller switched streams.
 * dpgo_problem_use_own_stream goes back to the handle's private non-blocking stream. */
int dpgo_problem_set_stream(dpgo_problem_t h, void* hip_stream);
int dpgo_problem_use_own_stream(dpgo_problem_t h);
int dpgo_problem_dims(dpgo_problem_t h, int* r, int* d, int* n, int* nnzb);

/* PoseGraph::quadraticMatrix() (include/DPGO/PoseGraph.h:162).  Host arrays; rowptr has
 * n+1 entries; every block row must contain its diagonal block (as the reference's QDiag
 * guarantees, src/PoseGraph.cpp:470-485).  Also (re)builds the preconditioner, like
 * PoseGraph::constructPreconditioner (src/PoseGraph.cpp:598-613). */
int dpgo_problem_set_Q_bsr(dpgo_problem_t h, int nnzb, const int32_t* rowptr, const int32_t* colidx,
                           const double* vals);
/* The same from the reference's own storage: Eigen::SparseMatrix<double, RowMajor> Q is scalar CSR
 * (outerIndexPtr / innerIndexPtr / valuePtr, int32; include/DPGO/DPGO_types.h:26) of size
 * (d+1)n x (d+1)n.  Entries are binned into (d+1)x(d+1) blocks (structural zeros of the blocks,
 * e.g. the last row of -T*Omega, are filled in); duplicates are summed. */
int dpgo_problem_set_Q_csr(dpgo_problem_t h, const int32_t* outer, const int32_t* inner, const double* values);
/* same pattern, new values (GNC re-weighting: PGOAgent.cpp:1122 clearDataMatrices path) */
int dpgo_problem_update_Q_values(dpgo_problem_t h, const double* vals);

/* ---- robust re-weighting on the device (GNC): replaces the "rebuild PoseGraph + constructQ + re-factorise"
 * loop of solveRobustPGO (src/DPGO_solver.cpp:335-412) and PGOAgent::updateMeasurementWeights
 * (src/PGOAgent.cpp:1104-1142), whose pattern-preserving part is values-only.
 * set_reweightable_edges: the private edges (both poses owned by this agent; R row-major per edge) whose
 * weights may change, with their CURRENT weights (the ones Q was built with) and fixedWeight flags
 * (RelativeSEMeasurement.h:44-47).  Call after dpgo_problem_set_Q_*.  Everything else in Q (shared-edge
 * diagonal terms, priors) is kept as a constant base. */
int dpgo_problem_set_reweightable_edges(dpgo_problem_t h, int m, const int32_t* p1, const int32_t* p2,
                                        const double* R, const double* t, const double* kappa, const double* tau,
                                        const double* weight, const uint8_t* fixed_weight);
/* General form including SHARED loop closures (distributed GNC: PGOAgent::updateMeasurementWeights re-weights
 * private and shared edges alike, src/PGOAgent.cpp:1104-1118).  role[e]: 0 private (p1, p2 both mine),
 * 1 shared outgoing (p1 mine; the other pose is neighbour-tile slot[e]), 2 shared incoming (p2 mine).
 * A shared edge contributes to Q's diagonal block of my pose and to one block of the G-coupling matrix
 * (call dpgo_problem_set_G_coupling first); both are rebuilt on the device after a weight change. */
int dpgo_problem_set_reweightable_edges_ex(dpgo_problem_t h, int m, const int32_t* p1, const int32_t* p2,
                                           const uint8_t* role, const int32_t* slot, const double* R,
                                           const double* t, const double* kappa, const double* tau,
                                           const double* weight, const uint8_t* fixed_weight);
/* Residuals rSq_e = computeMeasurementError (src/DPGO_utils.cpp:501-507) at the device iterate X_dev (shared
 * edges: other pose from nbr_tiles_dev, may be NULL if there are none); if update != 0 the non-fixed weights
 * become RobustCost::weight(sqrt(rSq)) for GNC_TLS with the given mu and barc (src/DPGO_robust.cpp:80-92) and
 * Q's values, the coupling values and the preconditioner are rebuilt on the device.
 * counts[3] = {inliers, outliers, undecided} among non-fixed edges whose source pose this agent owns
 * (w_tol as in DPGO_solver.cpp:340).  max_rsq (optional) = max residual over all edges (muInit, :358). */
int dpgo_problem_gnc_reweight_device(dpgo_problem_t h, const double* X_dev, const double* nbr_tiles_dev, double mu,
                                     double barc, double w_tol, int update, int counts[3], double* max_rsq);
/* Host-pointer flavour of dpgo_problem_gnc_reweight_device (X_host: r x (d+1)n column-major; private edges only). */
int dpgo_problem_gnc_reweight(dpgo_problem_t h, const double* X_host, double mu, double barc, double w_tol, int update,
                              int counts[3], double* max_rsq);
int dpgo_problem_set_edge_weights(dpgo_problem_t h, const double* weight_host);   /* + rebuild Q, preconditioner */
int dpgo_problem_get_edge_weights(dpgo_problem_t h, double* weight_host, double* rsq_host /* may be NULL */);

/* Q's current values (nnzb blocks, same order as set_Q_bsr) -- they change on the device under GNC re-weighting. */
int dpgo_problem_get_Q_values(dpgo_problem_t h, double* vals_host);
/* Multilevel (aggregation multigrid) preconditioner, precond = DPGO_PRECOND_MULTILEVEL (what DPGO_PRECOND_AUTO, the
 * default, switches to when the tCG budget binds): the device
 * path's stand-in for the reference's exact solve of Q + 0.1 I inside QuadraticProblem::PreConditioner
 * (src/QuadraticProblem.cpp:56-69; factor from PoseGraph::constructPreconditioner, src/PoseGraph.cpp:598-613).
 * One V(1,1) cycle: damped block-Jacobi smoothing on every level, level l+1's nodes = runs of ks[l] consecutive
 * level-l nodes, prolongation blocks = relative poses composed along the odometry chain (read off Q's own blocks),
 * Galerkin coarse operators, dense inverse of the coarsest operator -- all fp64, all built ON THE DEVICE (the block
 * patterns of the coarse operators are the only host step, once per pattern of Q).  The hierarchy is built lazily by
 * the first solve that needs it and rebuilt (values only) after Q's values changed (set_Q_*, update_Q_values, GNC
 * re-weighting), exactly where the reference drops its factor (src/PoseGraph.cpp:352-355,582-586).
 *   dpgo_problem_setup_multilevel: explicit setup.  nks = 0: default aggregates (dpgo_multilevel_default_ks);
 *     otherwise nks coarsenings with the given sizes.  A positive size k: the nodes of the next level are RUNS of k
 *     consecutive nodes, prolongation along the odometry chain (k must divide the workgroup tile of its level: 16 nodes
 *     for levels below 40 000 nodes in 3-D, 64 above; 20 / 84 in 2-D).  A single negative size -S: two levels with
 *     GRAPH aggregates of at most S poses -- grown breadth-first over Q's block pattern from seeds in index order,
 *     prolongation composed along each aggregate's breadth-first tree (compact aggregates: 40-60 % of the
 *     Hessian-vector products of index runs of the same size; the default whenever one coarsening with S <= 512
 *     reaches a dense level of about 2 500 unknowns, i.e. up to 200 000 poses in 3-D; DPGO_ML_GRAPH=0 restores runs).
 *     Two negative sizes {-S, -cap}: the same with the fragments of the growth merged up to cap poses
 *     (dpgo_multilevel_merged_aggregates; what the additive preconditioner builds for blocks beyond ~3 500 poses, and the
 *     DEFAULT of blocks of >= ~100 000 unknowns -- 25 600 poses in 3-D --: S = ceil(n (d+1) / 2 200), cap = 3 S / 2; the
 *     uniform aggregates need a quarter fewer of them for the same convergence, dpgo_multilevel_default_ks tells).
 *     omega: smoother damping (0.7); shift: the reference's 0.1.  Explicit sizes stick to the handle until the next call.
 *   dpgo_problem_multilevel_info: *nlevels in = capacity of the arrays, out = number of levels (coarsenings + 1);
 *     sizes[l] = nodes, ks[l] = aggregate size towards level l+1 (0 on the last; negative: graph aggregates of at most
 *     that many nodes -- with merged fragments the last level's entry holds -cap instead of 0), nnzb[l] = blocks of A_l.
 *   dpgo_problem_multilevel_get: copy one item of a built hierarchy to the host (tests / inspection). */
#define DPGO_ML_P_BLOCKS 0      /* level < last: n_l blocks (d+1)x(d+1), row-major                  (double) */
#define DPGO_ML_A_ROWPTR 1      /* level >= 1: n_l + 1                                              (int32)  */
#define DPGO_ML_A_COLIDX 2      /* level >= 1: nnzb_l                                               (int32)  */
#define DPGO_ML_A_VALUES 3      /* level >= 1: nnzb_l blocks, row-major                             (double) */
#define DPGO_ML_DENSE_INVERSE 4 /* last level: (n_L (d+1))^2, row-major                             (double) */
#define DPGO_ML_AGG_LABELS 5    /* level 0 with graph aggregates: aggregate of every pose, n_0         (int32)  */
#define DPGO_ML_AP_NNZB 6       /* level 0 of a two-level hierarchy: blocks of A P, one value          (int32)  */
#define DPGO_ML_RESTRICT_PARTIALS 7 /* level 0 with graph aggregates: partial sums one restriction writes -- one per run of
                                     * same-aggregate poses inside a wave's chunk of consecutive poses, one value (int32) */
int dpgo_multilevel_default_ks(int n, int d, int* ks, int* nks); /* *nks in: capacity of ks, out: count */
/* The host step of a graph hierarchy by itself (no device): aggregates of at most max_size nodes grown breadth-first over
 * the block pattern (seeds in index order, FIFO, neighbours in block-row order).  label[n] = aggregate of every node,
 * parent[n] (optional) = the node that discovered it (-1: the aggregate's root).  What dpgo_problem_setup_multilevel builds
 * for ks = {-max_size}; exposed so that the rule can be checked without a GPU. */
int dpgo_multilevel_graph_aggregates(int n, const int32_t* rowptr, const int32_t* colidx, int max_size, int32_t* label,
                                     int32_t* parent, int* n_aggregates);
/* The same followed by the merge of the growth's fragments (what ks = {-max_size, -merge_cap} builds): an aggregate of at
 * most max_size / 2 nodes joins the neighbouring aggregate it shares the most blocks with among those with room (sizes
 * add up to at most merge_cap; passes in index order until nothing changes); aggregates renumbered by smallest member,
 * breadth-first trees rebuilt from it.  Used where an aggregate is a WORKGROUP (the additive preconditioner of the
 * one-launch solve): every fragment would cost a whole workgroup of the at most 256. */
int dpgo_multilevel_merged_aggregates(int n, const int32_t* rowptr, const int32_t* colidx, int max_size, int merge_cap,
                                      int32_t* label, int32_t* parent, int* n_aggregates);
/* Layout precond = DPGO_PRECOND_ADDITIVE uses for this handle's block pattern (host only, computed once per pattern):
 * lane_groups = lane groups per pose of the one-launch kernel (4: tiles of 16 poses in 3-D / 20 in 2-D; 1: tiles of 64 / 84;
 * 0: the block does not fit 256 aggregates), tile = slots per aggregate, growth / merge_cap = the graph aggregates' sizes
 * (merge_cap 0: plain greedy growth; graph 0: index runs of `tile` poses, DPGO_ML_GRAPH=0), aggregates = workgroups. */
int dpgo_problem_additive_plan(dpgo_problem_t h, int* lane_groups, int* tile, int* growth, int* merge_cap, int* aggregates,
                               int* graph);
int dpgo_problem_setup_multilevel(dpgo_problem_t h, int nks, const int* ks, double omega, double shift);
int dpgo_problem_multilevel_info(dpgo_problem_t h, int* nlevels, int* sizes, int* ks, int* nnzb);
int dpgo_problem_multilevel_get(dpgo_problem_t h, int level, int what, void* out_host);
/* Which kernels a cycle of the current hierarchy runs (informational; same operator either way up to summation order):
 * DPGO_ML_PATH_AP = two-level hierarchy: the level-0 post-smoothing reads A P and the coarse solution (k_ml_post_ap) instead of
 * gathering a pose vector through Q; DPGO_ML_PATH_PACKED_DENSE = the dense level is applied from the packed lower triangle of
 * its (exactly symmetric) inverse on the fp64 matrix cores (k_dense_sym_apply; 64-bit storage, >= 3072 unknowns). */
#define DPGO_ML_PATH_AP 1
#define DPGO_ML_PATH_PACKED_DENSE 2
int dpgo_problem_multilevel_path(dpgo_problem_t h, int* flags);
/* Storage precision of the dense level (the inverse of the coarsest operator and the restricted residual it multiplies):
 * *bits = 64 (default) or 32 (opt-in: the cycle streams half the bytes; every product and sum stays fp64, the number of
 * Hessian-vector products to the tolerance is unchanged and the optimum does not depend on the preconditioner -- DESIGN.md
 * section 5); a negative input only queries.  DPGO_ML_DENSE_INVERSE returns the values the cycle applies. */
int dpgo_problem_multilevel_coarse_bits(dpgo_problem_t h, int* bits);
/* Storage precision of the OPERATOR COPIES the V-cycle streams on level 0 of an HBM-bound block (symmetric storage of Q,
 * two-level hierarchy; ignored elsewhere): *bits = 32 (default) or 64 -- fp32 copies of Q's values for the restriction's
 * residual r - A x1, of A P's values for the post-smoothing and of the prolongation blocks for both, beside the fp64
 * originals -- and the two vectors that live INSIDE a cycle (the pre-smoothed iterate the update kernel hands to the
 * restriction, the residual the restriction keeps for the post-smoothing) are stored in fp32 as well: the cycle streams
 * ~100 MB less per application at 100 000 poses and its gathers fetch half-size tiles.  The cycle is a preconditioner:
 * the tCG vectors, the Hessian step, the hierarchy's set-up, the smoother's factors, the dense level and every product and
 * sum stay fp64; the optimum does not depend on it, the products to the tolerance are the same (100 000-pose grid: 70
 * either way, 152 -> 138 us each); 64 (or DPGO_ML_OPERATOR_BITS=64) streams the fp64 originals (DPGO_ML_VECTOR_BITS=64: fp32
 * operator copies, fp64 vectors -- the A/B of the two halves).  The DENSE LEVEL (the inverse of the coarsest operator and
 * the restricted residual it multiplies) stays fp64 unless dpgo_problem_multilevel_coarse_bits(32) or DPGO_ML_DENSE_BITS=32
 * asks otherwise (measured neutral for the loop at 100 000 poses).
 * *active is a mask: 1 = operator copies, 2 = internal vectors, 4 = dense level streamed in fp32 by the last solve.  A negative input only queries; *active (optional) = 1 if the last
 * solve's cycle streamed the fp32 copies.  The oracle mirrors the storage (amg_operator_bits). */
int dpgo_problem_multilevel_operator_bits(dpgo_problem_t h, int* bits, int* active);
/* State of DPGO_PRECOND_AUTO on this handle: *use_multilevel in/out; 0 / 1 set it, -1 only queries, -2 returns to the
 * decision a fresh handle takes for the current problem (multilevel for a block without coupling to other agents,
 * block-Jacobi for a block of a multi-agent problem; every change of Q does the same). */
int dpgo_problem_auto_state(dpgo_problem_t h, int* use_multilevel);
/* Where the cost rule of DPGO_PRECOND_AUTO stands on this handle (any pointer may be NULL): state 0 = block-Jacobi,
 * 1 = the additive form on trial (its first solve is next or just ran), 2 = additive; jacobi_units = block-Jacobi work
 * counted since Q last changed or since the last hand-back; reference_products = products of the block-Jacobi solve the
 * additive form is measured against; switches = block-Jacobi -> additive transitions since Q last changed; backoff =
 * hand-backs so far (incl. trials not run because they could not win); units_jacobi / units_additive = the unit costs the
 * rule last charged a product of either kind on THIS handle: the constants below for a solve that has the device to
 * itself; for a handle solved next to others of the device (dpgo_optimize_device_many) scaled by the part of the chip the
 * launch blocks (the additive form owns one CU per aggregate, so such solves take turns).  dpgo_auto_rule_constants: the rule's units per block-Jacobi / additive product, the set-up cost in
 * the same units and the fewest products of a solve that can trigger a trial (host only). */
int dpgo_problem_auto_info(dpgo_problem_t h, int* state, long long* jacobi_units, int* reference_products, int* switches,
                           int* backoff, int* units_jacobi, int* units_additive);
int dpgo_auto_rule_constants(int* units_jacobi, int* units_additive, int* setup_units, int* min_products);
/* In-place blocked Gauss-Jordan inverse of a dense SPD matrix on the device (the kernel pair that inverts the coarsest
 * operator; exposed for tests).  N <= 16384, row-major host arrays; use_mfma: fp64 matrix cores for the rank-64
 * updates (v_mfma_f64_16x16x4_f64) or plain FMAs. */
int dpgo_dense_spd_inverse(int N, const double* A_host, double* Ainv_host, int device, int use_mfma);

/* PoseGraph::linearMatrix() (include/DPGO/PoseGraph.h:171): dense r x (d+1)n; NULL = zero */
int dpgo_problem_set_G(dpgo_problem_t h, const double* G_host);
int dpgo_problem_set_G_device(dpgo_problem_t h, const double* G_dev);
/* PoseGraph::constructG (src/PoseGraph.cpp:493-580) as a rectangular block-SpMM:
 * G = G0 + Xnbr * C, C = the inter-agent off-diagonal blocks of the global connection
 * Laplacian restricted to this agent's rows (n block rows, ncols neighbour-pose slots).
 * Set the pattern once, then update G on the device from the neighbour tiles buffer
 * (ncols tiles of (d+1)*r doubles, device pointer) every iteration. */
int dpgo_problem_set_G_coupling(dpgo_problem_t h, int ncols, int nnzb, const int32_t* rowptr,
                                const int32_t* colidx, const double* vals, const double* G0_host);
int dpgo_problem_update_G_from_neighbors_device(dpgo_problem_t h, const double* nbr_tiles_dev);

/* ---- QuadraticProblem methods, host-pointer flavour (drop-in) ----
 * f           : QuadraticProblem::f            (src/QuadraticProblem.cpp:29-41)
 * euc_grad    : QuadraticProblem::EucGrad      (:43-47)
 * euc_hess    : QuadraticProblem::EucHessianEta(:49-54)
 * precondition: QuadraticProblem::PreConditioner (:56-69)
 * rie_grad    : QuadraticProblem::RieGrad      (:71-79)
 * rie_grad_norm: QuadraticProblem::RieGradNorm (:81-83)
 * rie_hess    : ROPTLIB Problem::HessianEta = EucHessianEta + Stiefel::EucHvToHv + projection
 */
int dpgo_problem_f(dpgo_problem_t h, const double* X, double* f);
int dpgo_problem_euc_grad(dpgo_problem_t h, const double* X, double* EG);
int dpgo_problem_euc_hess(dpgo_problem_t h, const double* V, double* HV);
int dpgo_problem_rie_grad(dpgo_problem_t h, const double* X, double* RG);
int dpgo_problem_rie_grad_norm(dpgo_problem_t h, const double* X, double* gn);
int dpgo_problem_rie_hess(dpgo_problem_t h, const double* X, const double* V, double* HV);
int dpgo_problem_precondition(dpgo_problem_t h, int precond, double shift, const double* X,
                              const double* V, double* Z);

/* ---- QuadraticOptimizer::optimize (src/QuadraticOptimizer.cpp:26-48) ----
 * host flavour: X0 -> Xopt are host matrices (H2D + solve + D2H);
 * device flavour: X is a device matrix updated in place (nothing but scalars crosses PCIe). */
int dpgo_optimize(dpgo_problem_t h, const dpgo_ropt_params* params, const double* X0, double* Xopt,
                  dpgo_ropt_result* result);
int dpgo_optimize_device(dpgo_problem_t h, const dpgo_ropt_params* params, double* X_dev,
                         dpgo_ropt_result* result);
/* Number of warnings the library has printed to stderr so far (each kind once per process): e.g. more concurrently solved
 * handles (dpgo_optimize_device_many, dpgo_problem_eval_terms_device_many) than GPU_MAX_HW_QUEUES hardware queues. */
int dpgo_warning_count(void);
/* The same solve in two halves, for callers that enqueue a whole sweep without waiting (RBCDCluster.sweep when a process
 * hosts one agent per colour: colour c's solve, colour c+1's pack + exchange and its solve are all stream-ordered; the
 * host reads the results back at the end of the sweep instead of idling the GPU between phases).
 *   begin: G from the neighbour tile buffer (nbr_tiles_dev may be NULL), then -- if the solve is a one-launch solve
 *     (k_rtr_persist) -- the launch, its commit kernel and the read-backs are enqueued on the handle's stream and the call
 *     returns; any other solve runs to completion inside begin.  X_dev belongs to the solve until `end`.  The caller keeps
 *     everything it enqueues between begin and end on that same stream (one-launch solves of different handles must not be
 *     resident together; the resident-slot accounting of the synchronous calls is not used here).
 *   end: waits for the stream, fills `result`; after a time-out of the launch the solve is re-run with the multi-launch
 *     scheme exactly as dpgo_optimize_device does.  One solve in flight per handle (DPGO_ERR_STATE otherwise). */
int dpgo_optimize_device_begin(dpgo_problem_t h, const dpgo_ropt_params* params, double* X_dev,
                               const double* nbr_tiles_dev);
int dpgo_optimize_device_end(dpgo_problem_t h, dpgo_ropt_result* result);

/* Several agents hosted by one process / GPU updated CONCURRENTLY: the agents of one colour class of a parallel RBCD
 * sweep (examples/MultiRobotExample.cpp:170-255 updates its robots one after the other inside one process; the reference's
 * asynchronous mode runs one optimisation thread per agent, src/PGOAgent.cpp:483-520).  Per handle k, on the handle's OWN
 * stream and ordered after everything enqueued on `after_stream` so far (the public-pose exchange): G from the neighbour
 * tile buffer nbr_tiles_dev[k] (PGOAgent::updateX -> constructG; NULL entry or NULL array: G is left as it is), then
 * QuadraticOptimizer::optimize on X_dev[k] in place.  One host thread per handle feeds its kernels, so the solves overlap
 * on the device.  Returns when every stream has finished; results[k] as dpgo_optimize_device.  Handles must be distinct
 * and on one device. */
int dpgo_optimize_device_many(int count, const dpgo_problem_t* handles, const dpgo_ropt_params* params,
                              double* const* X_dev, const double* const* nbr_tiles_dev, void* after_stream,
                              dpgo_ropt_result* results);
/* The same for dpgo_problem_eval_terms_device: terms[3k..3k+2] = (sum(XQ.X), sum(X.G), |rgrad|^2) of handle k. */
int dpgo_problem_eval_terms_device_many(int count, const dpgo_problem_t* handles, const double* const* X_dev,
                                        const double* const* nbr_tiles_dev, void* after_stream, double* terms);

/* ---- device-pointer building blocks (bench / agent loop) ---- */
/* OUT = V*Q (+G if add_G): the named north-star kernel */
int dpgo_spmm_device(dpgo_problem_t h, const double* V_dev, double* OUT_dev, int add_G);
/* f, |rgrad| at a device X without copying X back */
int dpgo_problem_eval_device(dpgo_problem_t h, const double* X_dev, double* f, double* gradnorm);
/* the three sums behind f and |rgrad|: xqx = sum(XQ.X), xg = sum(X.G), g2 = |rgrad|^2
 * (f = 0.5 xqx + xg).  Lets a driver assemble the CENTRAL cost 0.5 sum_a (xqx_a + xg_a) from
 * agent-local evaluations (examples/MultiRobotExample.cpp:220-225 does it on a central problem). */
int dpgo_problem_eval_terms_device(dpgo_problem_t h, const double* X_dev, double* xqx, double* xg, double* g2);
/* Storage of Q that its products read -- the plain Q*V product (dpgo_spmm_device, dpgo_problem_euc_grad / euc_hess, the
 * initialisation's PCG) and, in a solve, the fused tCG-step kernel and the level-0 kernels of the multilevel cycle:
 * PLAIN = the block-CSR arrays as given; SYMMETRIC = upper blocks only, stored transposed, lower blocks by reference
 * (half of Q's bytes, outer-product gather: faster once Q comes from HBM instead of the 256 MB Infinity Cache, no gain for
 * cache-resident blocks -- DESIGN.md section 3); AUTO (default) = SYMMETRIC when Q plus eight pose vectors (the tCG loop's
 * working set) exceed 256 MiB, about 120 000 poses of a 3-D grid.  SYMMETRIC needs one pose per (d+1) lanes (blocks of
 * >= 40 000 poses) and Q[j,i] == Q[i,j]^T to 1e-12 relative (checked on the device whenever the values change); when either
 * does not hold the plain arrays are read.  Same results either way up to summation order.  *in_use (optional) = what the
 * next product will read. */
#define DPGO_SPMM_AUTO 0
#define DPGO_SPMM_PLAIN 1
#define DPGO_SPMM_SYMMETRIC 2
int dpgo_problem_set_spmm_variant(dpgo_problem_t h, int variant, int* in_use);
/* Which instance of the tCG-step kernel the next solve launches (multi-launch scheme): symmetric = 1 -> k_tcg_hess_sym,
 * else k_tcg_hess_span / k_tcg_hess with `split` lane groups per pose; stream_nt = 1 -> the instance whose single-use
 * operands (own-tile X, delta, H delta, S) move as non-temporal accesses.  AUTO selects the symmetric storage and the
 * non-temporal instance together when the tCG loop's working set -- Q, eight pose vectors and what the last solve's
 * preconditioner streams -- exceeds the 256 MB Infinity Cache; DPGO_SPMM_SYMMETRIC / DPGO_STREAM_NT = 0 / 1 override. */
int dpgo_problem_tcg_kernel_info(dpgo_problem_t h, int* symmetric, int* split, int* stream_nt);
/* time `reps` back-to-back SpMM launches with HIP events on the handle's stream;
 * n_buffers >= 1 rotates that many (V, OUT) buffer pairs; returns average ms per launch */
int dpgo_bench_spmm(dpgo_problem_t h, int reps, int warmup, double* avg_ms);
/* As dpgo_bench_spmm, but cycling through nsets private copies of (Q values, block columns, X, OUT) so that no
 * launch finds its operands in the 256 MB Infinity Cache; set_bytes (optional) = bytes of one set.  Times the
 * storage the handle's products read (dpgo_problem_set_spmm_variant). */
int dpgo_bench_spmm_rotating(dpgo_problem_t h, int nsets, int reps, int warmup, double* avg_ms, double* set_bytes);
/* same for the dominant kernel of a solve: the fused Q*X + Riemannian-Hessian kernel (one per tCG
 * iteration), on the solver's own buffers (iterate, cached S, search direction) */
int dpgo_bench_hess(dpgo_problem_t h, int reps, int warmup, double* avg_ms);
/* One whole local solve (QuadraticOptimizer::optimize, src/QuadraticOptimizer.cpp:26-48) per repetition, each from a copy
 * of X0_dev taken outside the event pair; HIP events on the handle's stream around the solve.  For blocks in the latency
 * regime a solve is ONE launch of k_rtr_persist (+ an 80-byte memset and a 200-byte read-back), so avg_ms is that kernel's
 * launch duration; *persistent = 1 when every timed repetition ran that way.  avg_products = Hessian-vector products per
 * solve. */
int dpgo_bench_solve(dpgo_problem_t h, const dpgo_ropt_params* params, const double* X0_dev, int reps, int warmup,
                     double* avg_ms, double* avg_products, int* persistent);
/* As dpgo_bench_hess with every operand cycling through nsets private copies (see dpgo_bench_spmm_rotating). */
int dpgo_bench_hess_rotating(dpgo_problem_t h, int nsets, int reps, int warmup, double* avg_ms);
/* Average launch time of the other kernels of one preconditioned tCG iteration, each timed as `reps` back-to-back
 * launches on the solver's buffers: out_ms[0] = k_tcg_update, then (multilevel hierarchy built; else zeros)
 * [1] = k_ml_restrict of level 0, [2] = k_ml_coarse_prolong, [3] = k_ml_post, [4] = the whole cycle tail
 * (all restrictions, dense level, all post-smoothing launches) as launched per iteration. */
int dpgo_bench_iteration_kernels(dpgo_problem_t h, int reps, int warmup, double out_ms[5]);

/* Test probe (no reference counterpart): the communication primitives of the one-launch solve alone.  `workgroups`
 * (<= the device's CUs, <= 256) run `steps` chip-wide reductions of two partial sums per thread -- in_dev
 * [workgroups][256][2]: step s reduces in[..][0] * (s + 1) and in[..][1] - s -- each carrying a payload of `pay`
 * (6, 9, 15, 20 or 24) doubles per workgroup, the per-wave parts pay_in_dev [workgroups][4][pay] (+ s) added in wave
 * order.  sums_dev [workgroups][steps][2]: what every workgroup's thread 0 holds afterwards (identical bits in all of
 * them); pay_out_dev [workgroups][steps][workgroups][pay]: participant t's payload as thread t of every workgroup
 * received it; rows_out_dev [workgroups][4][pay]: the wavefront sums of value e = in[t][0] (e + 1) + in[t][1] over the 64
 * lanes of each wave by the reduce-scatter wave_reduce_rows.  Synchronous, default stream. */
int dpgo_debug_reduction_primitives(int workgroups, int pay, int steps, const double* in_dev, const double* pay_in_dev,
                                    double* sums_dev, double* pay_out_dev, double* rows_out_dev);

/* One-launch solve (kernels/persist.h, k_rtr_persist): for blocks in the latency regime (every block the kernel can hold:
 * <= 32 768 poses in 3-D; environment DPGO_PERSIST_MAX_POSES lowers the limit) with the block-Jacobi, additive or no
 * preconditioner, dpgo_optimize* runs the WHOLE local solve -- initial statistics, every RTR iteration's tCG_TR loop,
 * retraction, trial gradient, rho test -- as ONE launch on up to 256 resident workgroups: every tCG vector of a workgroup's
 * rows stays in registers, the iterate in LDS, the only vectors exchanged are z (per tCG iteration), the trial point and the
 * step (write-through stores, agent-scope gathers), and the barrier between two phases is the all-reduce of their dot
 * products.  On by size (DPGO_PERSIST=0/1 or set_persistent override); RTR_iterations == 1, tcg_poll_interval > 0 and RGD
 * keep the multi-launch scheme.  If a launch times out (its workgroups were not all resident, e.g. because another process
 * occupies the device) the caller's iterate is untouched, the solve runs on the multi-launch scheme and the handle stops
 * using the kernel.  info: what the LAST optimize call did (last_members = 0: the multi-launch scheme ran; last_layout =
 * 16 * lane groups per pose + tiles per workgroup). */
int dpgo_problem_set_persistent(dpgo_problem_t h, int enable);
int dpgo_problem_persistent_info(dpgo_problem_t h, int* enabled, int* workgroups, int* last_members,
                                 int* last_iterations, int* last_layout);
/* The in-kernel phase split of the LAST one-launch solve, microseconds per tCG iteration on participant 0 (100 MHz wall
 * clock, iterations after the first): [0] Hessian phase, [1] the all-reduce behind it, [2] update phase (with the additive
 * form: + restriction), [3] the reduction(s) behind it (additive: coarse solve, correction, projection and two reductions);
 * all 0 when the last call ran the multi-launch scheme.  *iterations (optional) = tCG iterations of that launch. */
int dpgo_problem_persistent_phases(dpgo_problem_t h, double us_per_iteration[4], int* iterations);

/* ---- pose order for HBM-bound blocks (host only; no reference counterpart: Eigen's product has no such notion) ----
 * The block-SpMM kernels gather the tiles of a pose's graph neighbours; each XCD's workgroups sweep one contiguous eighth
 * of the poses, so what is re-used stays in that XCD's L2 only if neighbours are close in the pose ORDER.
 * dpgo_locality_order keeps the nparts contiguous chunks of the index range (boundaries at multiples of `align` poses) and,
 * inside every chunk, keeps runs of 16 consecutive poses together (odometry neighbours: a wave's poses then gather contiguous
 * memory) and orders the runs by reverse Cuthill-McKee over the graph of runs (dpgo_locality_order_runs: the run length as
 * an argument; 1 = plain reverse Cuthill-McKee of the poses, measured 6 % slower than no renumbering).  rowptr / colidx: any
 * structurally symmetric adjacency of the poses (Q's block pattern as dpgo_build_Q_bsr emits it); new_index[i] = position
 * of caller pose i.  The solver itself never renumbers: a caller that owns the graph (the agent layer, dpgo_amd/agent.py,
 * build_pose_graphs(reorder = True)) relabels its measurements with it and moves X through dpgo_permute_tiles_device on the
 * way in and out, so that public pose ids and trajectories keep the caller's numbering.  Opt-in: on the lattice workloads
 * of BASELINE.json the odometry order measured FASTER than every renumbering tried. */
int dpgo_locality_order(int n, const int32_t* rowptr, const int32_t* colidx, int nparts, int align, int32_t* new_index);
int dpgo_locality_order_runs(int n, const int32_t* rowptr, const int32_t* colidx, int nparts, int align, int run,
                             int32_t* new_index);
/* out tile new_index[i] = in tile i (forward = 1) or out tile i = in tile new_index[i] (forward = 0); r x (d+1) doubles per
 * tile, device pointers, on `stream`. */
int dpgo_permute_tiles_device(int r, int d, int n, const int32_t* new_index_dev, const double* in_dev, double* out_dev,
                              int forward, void* stream);

/* ---- initial guesses (src/DPGO_solver.cpp:220-303) ----
 * chordal: the two linear least-squares problems of chordalInitialization (rotations with pose 0 pinned to the
 *   identity, projected to SO(d); then translations) -- the reference solves them with SPQR (constructBMatrices /
 *   recoverTranslations, src/DPGO_utils.cpp:346-462); here their normal equations are solved ON THE DEVICE by
 *   Jacobi-preconditioned conjugate gradients over the block-SpMM of the hot path, the SO(d) projection by the
 *   rounding kernel.  Single-robot measurements (all poses of one robot, frames 0 .. n-1); weights are not used
 *   (as in the reference).  tol: relative residual of the CG solves (<= 0: 1e-13); max_iter <= 0: 20 n.
 *   iters_out (optional): CG iterations of the rotation and of the translation solve.
 * odometry: poses composed along the odometry chain p -> p+1 (odometryInitialization, :271-303); host only.
 * T_host: tiles [n][d+1][d] (= the reference's d x (d+1)n column-major Matrix [R_0 t_0 ... ]). */
int dpgo_chordal_initialization(int d, int n, int m, const int32_t* p1, const int32_t* p2, const double* R,
                                const double* t, const double* kappa, const double* tau, double tol, int max_iter,
                                double* T_host, int iters_out[2], int device);
int dpgo_odometry_initialization(int d, int n, int m, const int32_t* p1, const int32_t* p2, const double* R,
                                 const double* t, double* T_host);

/* ---- device memory for callers above the C ABI that keep their iterates in HBM (the C++ / Python agent mirrors keep
 * X, Y, V, XPrev and the neighbour tile buffers there; src/PGOAgent.cpp:376-432,880-936 keeps them in Eigen matrices).
 * kind: DPGO_COPY_H2D | DPGO_COPY_D2H | DPGO_COPY_D2D; the copy is enqueued on `stream` (NULL = default stream) and,
 * for D2H, completed before the call returns. */
#define DPGO_COPY_H2D 0
#define DPGO_COPY_D2H 1
#define DPGO_COPY_D2D 2
int dpgo_device_malloc(void** out, size_t bytes, int device);
int dpgo_device_free(void* p);
int dpgo_device_memcpy(void* dst, const void* src, size_t bytes, int kind, void* stream);
int dpgo_device_synchronize(void* stream);

/* ---- RCCL transport of the public-pose exchange (one process per GPU; dpgo_amd/csrc/comm.cpp) ----
 * Replaces, for agents living in different processes, the in-process pointer calls of the reference's driver
 * (examples/MultiRobotExample.cpp:183-204: getSharedPoseDict -> updateNeighborPoses) and its central reductions
 * (:220-254).  Everything is enqueued on the caller's HIP stream: the pack kernel (dpgo_gather_tiles_device) before,
 * the coupling SpMM (dpgo_problem_update_G_from_neighbors_device) after, no host wait in between.
 *   unique_id : rank 0 creates the 128-byte id (ncclGetUniqueId) and hands it to the other ranks by any means
 *   create    : collective over all ranks (ncclCommInitRank) on `device`
 *   exchange  : ONE grouped batch of ncclSend / ncclRecv of packed pose tiles (counts in doubles); messages between
 *               one pair of ranks are matched in list order
 *   allreduce : in place, op = DPGO_COMM_SUM | DPGO_COMM_MAX;  broadcast: in place from `root`
 * RCCL is bound at run time (librccl.so.1); without it these return DPGO_ERR_HIP and the rest of the library works. */
#define DPGO_COMM_ID_BYTES 128
#define DPGO_COMM_SUM 0
#define DPGO_COMM_MAX 1
typedef struct dpgo_comm_s* dpgo_comm_t;
int dpgo_comm_unique_id(char id[DPGO_COMM_ID_BYTES]);
int dpgo_comm_create(dpgo_comm_t* out, int nranks, int rank, const char id[DPGO_COMM_ID_BYTES], int device);
int dpgo_comm_destroy(dpgo_comm_t c);
int dpgo_comm_info(dpgo_comm_t c, int* nranks, int* rank);
int dpgo_comm_exchange(dpgo_comm_t c, int nsend, const int* send_peer, const double* const* send_dev,
                       const int* send_count, int nrecv, const int* recv_peer, double* const* recv_dev,
                       const int* recv_count, void* stream);
int dpgo_comm_allreduce(dpgo_comm_t c, double* buf_dev, int count, int op, void* stream);
int dpgo_comm_broadcast(dpgo_comm_t c, double* buf_dev, int count, int root, void* stream);

/* ---- manifold: LiftedSEManifold (include/DPGO/manifold/LiftedSEManifold.h:28-43) + the
 * ROPTLIB Stiefel x Euclidean product-manifold operations it configures
 * (src/manifold/LiftedSEManifold.cpp:16-24).  Host pointers; `device` selects the GPU. */
/* LiftedSEManifold::project (polar factor per pose, src/manifold/LiftedSEManifold.cpp:34-45) */
int dpgo_manifold_project(int r, int d, int n, const double* M, double* out, int device);
/* ROPTLIB ProductManifold::Projection: tangent projection of V at X */
int dpgo_manifold_tangent_project(int r, int d, int n, const double* X, const double* V, double* out,
                                  int device);
/* ROPTLIB ProductManifold::Retraction (qf on Stiefel, x+eta on Euclidean): out = R_X(scale*eta) */
int dpgo_manifold_retract(int r, int d, int n, const double* X, const double* eta, double scale,
                          double* out, int device);
/* device-pointer versions on a caller-supplied stream (NULL = default stream) */
int dpgo_manifold_project_device(int r, int d, int n, const double* M_dev, double* out_dev, void* stream);
/* Rounding to SE(d): PGOAgent::getTrajectoryInLocalFrame / getTrajectoryInGlobalFrame
 * (src/PGOAgent.cpp:718-767).  T (d x (d+1)n column-major, as the reference's Matrix) gets, per pose,
 * [ projectToRotationGroup(Ya^T Y_i) | Ya^T p_i - Ya^T pa ] (src/DPGO_utils.cpp:464-478) with the anchor
 * (Ya | pa) = anchor_host (r x (d+1) column-major, host memory: PGOAgent::setGlobalAnchor, :838-844) or, when
 * NULL, pose 0 of X (local frame). */
int dpgo_round_trajectory(int r, int d, int n, const double* X_host, const double* anchor_host, double* T_host,
                          int device);
int dpgo_round_trajectory_device(int r, int d, int n, const double* X_dev, const double* anchor_host, double* T_dev,
                                 void* stream);

/* The public-pose exchange between agents that live in ONE process (what examples/MultiRobotExample.cpp:183-204 does by
 * pointer calls: getSharedPoseDict -> updateNeighborPoses), as one launch per exchange phase: message m copies count[m]
 * pose tiles  src_dev[m][idx_dev[m][k]] -> dst_dev[m][k]  (src: the sender's iterate, dst: the receiver's neighbour tile
 * buffer at the sender's slot range).  The addresses are captured at creation and must stay valid.  (d, r) as
 * dpgo_supported. */
typedef struct dpgo_exchange_plan_s* dpgo_exchange_plan_t;
int dpgo_exchange_plan_create(dpgo_exchange_plan_t* out, int r, int d, int nmsg, const double* const* src_dev,
                              const int32_t* const* idx_dev, const int* count, double* const* dst_dev, int device);
int dpgo_exchange_plan_run(dpgo_exchange_plan_t plan, void* stream);
int dpgo_exchange_plan_destroy(dpgo_exchange_plan_t plan);
/* Device-side ordering of an exchange between PROCESSES that map each other's buffers (dpgo_amd/ipc.py, the peer-store
 * transport; no reference counterpart -- the demo driver's agents share one address space): 64-bit epoch words in device
 * memory both sides map.  write: words_dev[k] <- values[k] (system-scope release store), enqueued BEHIND the kernel that
 * produced the data.  wait: the stream stalls until every word >= its value (system-scope acquire loads), enqueued IN
 * FRONT of the kernel that consumes the data; a word that does not arrive within timeout_ms traps the waiting kernel (the
 * process fails loudly; nothing hangs).  Host arrays of device pointers / values; any n. */
int dpgo_flags_write_device(int n, unsigned long long* const* words_dev, const unsigned long long* values, void* stream);
int dpgo_flags_wait_device(int n, unsigned long long* const* words_dev, const unsigned long long* values, int timeout_ms,
                           void* stream);
/* The same wait without the trap (what dpgo_amd/ipc.py enqueues): timeout_ms = 0 waits without a limit -- a late peer (a host
 * side Q rebuild, a checkpoint, a debugger) only delays the stream, as the host barriers of round 4 did; timeout_ms > 0: a
 * word that has not arrived by then is REPORTED -- 1 + its index stored (system scope) into *err_word, which must be memory
 * the host can read without synchronising the stream (pinned host memory) -- and the kernel returns, so that the host can
 * raise an error at its next exchange instead of losing its HIP context. */
int dpgo_flags_wait_device_checked(int n, unsigned long long* const* words_dev, const unsigned long long* values,
                                   long long timeout_ms, unsigned long long* err_word, void* stream);

/* Agent status (PGOAgent::iterate, src/PGOAgent.cpp:399-420): relativeChange = LiftedPoseArray::maxTranslationDistance
 * (src/manifold/Poses.cpp:86-94) of the iterate and the previous one, max_i |p_i - p_i'| over the translation columns.
 * The result is left in *out_dev (device double: e.g. a slot of the vector a termination vote all-reduces with
 * DPGO_COMM_MAX); out_host != NULL additionally copies it back (one synchronisation). */
int dpgo_max_translation_distance_device(int r, int d, int n, const double* X_dev, const double* Xprev_dev,
                                         double* out_dev, double* out_host, void* stream);
/* out[k] = src tile idx[k]  (K11 pack for the public-pose exchange:
 * PGOAgent::getSharedPoseDict, src/PGOAgent.cpp:97-166) */
int dpgo_gather_tiles_device(int r, int d, const double* src_dev, const int32_t* idx_dev, int count,
                             double* dst_dev, void* stream);
/* out = a*A + b*B (elementwise over n tiles), then optional polar projection: the Nesterov
 * updates PGOAgent::updateY / updateV (src/PGOAgent.cpp:922-936) */
int dpgo_axpby_project_device(int r, int d, int n, double a, const double* A_dev, double b,
                              const double* B_dev, double c, const double* C_dev, int project,
                              double* out_dev, void* stream);

/* ---- host-side data-matrix construction (no GPU needed) ----
 * PoseGraph::constructQ (src/PoseGraph.cpp:381-491) with constructConnectionLaplacianSE
 * (src/DPGO_utils.cpp:272-344).  Measurements in SoA form (RelativeSEMeasurement.h:21-50):
 * R is m x d x d with R[e][row][col] (row-major per edge), t is m x d.  Edges with
 * r1 == r2 == my_id are private; others are shared (outgoing if r1 == my_id).
 * Two-call pattern: call with rowptr/colidx/vals NULL to get *nnzb_out, then with buffers. */
int dpgo_build_Q_bsr(int my_id, int d, int n, int m, const int32_t* r1, const int32_t* p1, const int32_t* r2,
                     const int32_t* p2, const double* R, const double* t, const double* kappa,
                     const double* tau, const double* weight, int n_priors, const int32_t* prior_idx,
                     double prior_kappa, double prior_tau, int* nnzb_out, int32_t* rowptr,
                     int32_t* colidx, double* vals);
/* Inter-agent coupling blocks for dpgo_problem_set_G_coupling: slot_of_edge[e] (>= 0 for
 * shared edges) is the column slot of the neighbour pose of edge e in the neighbour tile
 * buffer.  Two-call pattern as above. */
int dpgo_build_G_coupling(int my_id, int d, int n, int m, const int32_t* r1, const int32_t* p1,
                          const int32_t* r2, const int32_t* p2, const double* R, const double* t,
                          const double* kappa, const double* tau, const double* weight,
                          const int32_t* slot_of_edge, int* nnzb_out, int32_t* rowptr, int32_t* colidx,
                          double* vals);

#ifdef __cplusplus
}
#endif
#endif /* DPGO_HIP_H */
