// dpgo_hip.hpp -- header-only C++17 mirror of the reference's local-solver classes over the C ABI
// (dpgo_hip.h).  Same class / method names, argument meaning and error behaviour as
//
//   DPGO::QuadraticProblem    include/DPGO/QuadraticProblem.h:33-115
//   DPGO::QuadraticOptimizer  include/DPGO/QuadraticOptimizer.h:20-104
//   DPGO::LiftedSEManifold    include/DPGO/manifold/LiftedSEManifold.h:28-43
//   DPGO::ROptParameters / ROPTResult   include/DPGO/DPGO_types.h:44-107
//   DPGO::PoseGraph (data-matrix part)  include/DPGO/PoseGraph.h:59-69,106-194
//   DPGO::LiftedSEVariable / LiftedSEVector   include/DPGO/manifold/LiftedSEVariable.h:33-119, LiftedSEVector.h:28-47
//   DPGO::PGOAgent (hot-path subset: iterate, pose dictionaries, setX / getX)   include/DPGO/PGOAgent.h:250-548
//   solvePGO / solveRobustPGO / chordalInitialization / odometryInitialization   include/DPGO/DPGO_solver.h
//
// so that the bodies of PGOAgent::updateX (src/PGOAgent.cpp:961-991) and solvePGO (src/DPGO_solver.cpp:322-331) read
// the same against it.  UNTESTED against the reference's own headers: Eigen, ROPTLIB, glog and Boost are absent from
// this image, so the namespace is dpgo_hip:: (not DPGO::), Matrix is a minimal column-major class of this header with
// Eigen::MatrixXd's data layout (with Eigen present, Eigen::Map<const Eigen::MatrixXd>(m.data(), m.rows(), m.cols())
// and back are zero-copy views) and the ROPTLIB-typed virtuals appear as double*-based overloads; INTEGRATION.md
// shows the Eigen-typed shim a maintainer would write.  What IS tested: this header compiles with g++ -std=c++17 and
// re-runs the reference's known-answer tests and the demo schedule on the GPU (tests/cxx/test_shim.cpp).
//
// Errors: the reference aborts through glog CHECK; here a dpgo_hip::Error (std::runtime_error) is
// thrown with the C ABI's message.  Nothing in this header computes on the CPU: without a HIP
// device every compute call throws.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <map>
#include <set>
#include <memory>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "dpgo_hip.h"

namespace dpgo_hip {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error("dpgo_hip error " + std::to_string(c) + ": " + m), code(c) {}
};
inline void check(int rc) {
  if (rc != DPGO_OK) throw Error(rc, dpgo_last_error());
}

// Column-major dense matrix with Eigen::MatrixXd's memory layout.
class Matrix {
 public:
  Matrix() = default;
  Matrix(size_t rows, size_t cols) : r_(rows), c_(cols), a_(rows * cols, 0.0) {}
  static Matrix Zero(size_t rows, size_t cols) { return Matrix(rows, cols); }
  static Matrix Identity(size_t rows, size_t cols) {
    Matrix m(rows, cols);
    for (size_t i = 0; i < rows && i < cols; ++i) m(i, i) = 1.0;
    return m;
  }
  size_t rows() const { return r_; }
  size_t cols() const { return c_; }
  double* data() { return a_.data(); }
  const double* data() const { return a_.data(); }
  double& operator()(size_t i, size_t j) { return a_[j * r_ + i]; }
  double operator()(size_t i, size_t j) const { return a_[j * r_ + i]; }
  Matrix block(size_t i0, size_t j0, size_t nr, size_t nc) const {
    Matrix m(nr, nc);
    for (size_t j = 0; j < nc; ++j)
      for (size_t i = 0; i < nr; ++i) m(i, j) = (*this)(i0 + i, j0 + j);
    return m;
  }
  void setBlock(size_t i0, size_t j0, const Matrix& b) {
    for (size_t j = 0; j < b.cols(); ++j)
      for (size_t i = 0; i < b.rows(); ++i) (*this)(i0 + i, j0 + j) = b(i, j);
  }
  double norm() const {
    double s = 0;
    for (double v : a_) s += v * v;
    return std::sqrt(s);
  }

 private:
  size_t r_ = 0, c_ = 0;
  std::vector<double> a_;
};

// DPGO::RelativeSEMeasurement (include/DPGO/RelativeSEMeasurement.h:21-50); R is d x d, t is d x 1.
struct RelativeSEMeasurement {
  size_t r1 = 0, r2 = 0, p1 = 0, p2 = 0;
  Matrix R, t;
  double kappa = 0, tau = 0;
  bool fixedWeight = false;
  double weight = 1.0;
  RelativeSEMeasurement() = default;
  RelativeSEMeasurement(size_t first_robot, size_t second_robot, size_t first_pose, size_t second_pose,
                        const Matrix& relative_rotation, const Matrix& relative_translation,
                        double rotational_precision, double translational_precision)
      : r1(first_robot), r2(second_robot), p1(first_pose), p2(second_pose), R(relative_rotation),
        t(relative_translation), kappa(rotational_precision), tau(translational_precision) {}
};

// DPGO::ROptParameters (include/DPGO/DPGO_types.h:44-86)
class ROptParameters {
 public:
  enum class ROptMethod { RTR, RGD };
  ROptMethod method = ROptMethod::RTR;
  bool verbose = false;
  double gradnorm_tol = 1e-2;
  double RGD_stepsize = 1e-3;
  bool RGD_use_preconditioner = true;
  int RTR_iterations = 3;
  int RTR_tCG_iterations = 50;
  double RTR_initial_radius = 100;
  // not in the reference struct: tCG preconditioner of the device path (DPGO_PRECOND_*).  The reference always uses
  // the exact solve of Q + 0.1 I (src/QuadraticProblem.cpp:56-69); MULTILEVEL -- an aggregation-multigrid V-cycle for the
  // same matrix, built on the device -- is this library's stand-in for it, and the default (AUTO) runs it whenever the
  // tCG budget binds and the cheaper block-Jacobi while it does not (include/dpgo_hip.h).
  int precond = DPGO_PRECOND_AUTO;
  dpgo_ropt_params to_c() const {
    dpgo_ropt_params c;
    dpgo_ropt_params_default(&c);
    c.method = method == ROptMethod::RTR ? DPGO_METHOD_RTR : DPGO_METHOD_RGD;
    c.verbose = verbose;
    c.gradnorm_tol = gradnorm_tol;
    c.RGD_stepsize = RGD_stepsize;
    c.RGD_use_preconditioner = RGD_use_preconditioner;
    c.RTR_iterations = RTR_iterations;
    c.RTR_tCG_iterations = RTR_tCG_iterations;
    c.RTR_initial_radius = RTR_initial_radius;
    c.precond = precond;
    return c;
  }
};

// DPGO::ROPTResult (include/DPGO/DPGO_types.h:91-107); tCGStatus is the DPGO_TCG_* code.
struct ROPTResult {
  ROPTResult(bool suc = false, double f0 = 0, double gn0 = 0, double fStar = 0, double gnStar = 0, double ms = 0)
      : success(suc), fInit(f0), gradNormInit(gn0), fOpt(fStar), gradNormOpt(gnStar), elapsedMs(ms) {}
  bool success;
  double fInit, gradNormInit, fOpt, gradNormOpt, elapsedMs;
  int tCGStatus = DPGO_TCG_MAXITER;
  int tcgIterations = 0, rtrIterations = 0;
  int precondUsed = DPGO_PRECOND_NONE;  // what the call ran (DPGO_PRECOND_AUTO resolved)
};

// Data-matrix part of DPGO::PoseGraph: measurements -> Q (block-CSR), G (dense), with the reference's
// invalidation rules (src/PoseGraph.cpp:183-186, 345-379).
class PoseGraph {
 public:
  using PoseID = std::pair<unsigned, unsigned>;  // (robot, frame); DPGO::PoseID, DPGO_types.h:110-120
  PoseGraph(unsigned id, unsigned r, unsigned d) : id_(id), r_(r), d_(d) {
    if (r < d) throw Error(DPGO_ERR_INVALID, "CHECK(r >= d) failed");  // src/PoseGraph.cpp:19
  }
  unsigned id() const { return id_; }
  unsigned r() const { return r_; }
  unsigned d() const { return d_; }
  unsigned n() const { return n_; }

  void setMeasurements(const std::vector<RelativeSEMeasurement>& ms) {  // src/PoseGraph.cpp:61-66
    meas_.clear();
    n_ = 0;
    std::map<std::pair<PoseID, PoseID>, bool> seen;
    for (const auto& m : ms) {
      if (m.r1 != id_ && m.r2 != id_) continue;  // irrelevant edge (:68-71)
      auto key = std::make_pair(PoseID(m.r1, m.p1), PoseID(m.r2, m.p2));
      if (seen.count(key)) continue;  // duplicate (:83-88)
      seen[key] = true;
      meas_.push_back(m);
      if (m.r1 == id_) n_ = std::max<unsigned>(n_, (unsigned)m.p1 + 1);
      if (m.r2 == id_) n_ = std::max<unsigned>(n_, (unsigned)m.p2 + 1);
    }
    neighbor_poses_.clear();
    priors_.clear();
    neighbor_active_.clear();
    for (const auto& m : meas_)
      if (m.r1 != m.r2) neighbor_active_[(unsigned)(m.r1 == id_ ? m.r2 : m.r1)] = true;
    clearDataMatrices();
  }
  // Neighbour activity (src/PoseGraph.cpp:188-207, 252-264, 632-634): the shared edges with an INACTIVE neighbour are
  // skipped by constructQ / constructG (:425-430, :527-532) unless useInactiveNeighbors is set and the pose is known.
  bool hasNeighbor(unsigned robot_id) const { return neighbor_active_.count(robot_id) != 0; }
  bool isNeighborActive(unsigned neighbor_id) const {
    auto it = neighbor_active_.find(neighbor_id);
    return it != neighbor_active_.end() && it->second;
  }
  void setNeighborActive(unsigned neighbor_id, bool active) {
    if (!hasNeighbor(neighbor_id)) return;
    if (neighbor_active_[neighbor_id] != active) clearDataMatrices();
    neighbor_active_[neighbor_id] = active;
  }
  void useInactiveNeighbors(bool use = true) {
    use_inactive_neighbors_ = use;
    clearDataMatrices();
  }
  std::set<unsigned> activeNeighborIDs() const {
    std::set<unsigned> out;
    for (const auto& kv : neighbor_active_)
      if (kv.second) out.insert(kv.first);
    return out;
  }
  void setPrior(unsigned index, const Matrix& Xi) {  // :176-181
    if (index >= n_ || Xi.rows() != r_ || Xi.cols() != d_ + 1) throw Error(DPGO_ERR_INVALID, "bad prior");
    priors_[index] = Xi;
    clearDataMatrices();
  }
  void setNeighborPoses(const std::map<PoseID, Matrix>& pose_dict) {  // :183-186 (resets G only)
    neighbor_poses_ = pose_dict;
    has_G_ = false;
  }
  void clearDataMatrices() {  // :376-379
    has_Q_ = has_G_ = false;
    ++q_version_;
  }
  unsigned long qVersion() const { return q_version_; }
  const std::vector<RelativeSEMeasurement>& measurements() const { return meas_; }  // after duplicate removal
  bool hasLinearTerm() const {
    if (!priors_.empty()) return true;
    for (const auto& m : meas_)
      if (m.r1 != m.r2) return true;
    return false;
  }

  struct Bsr {
    std::vector<int32_t> rowptr, colidx;
    std::vector<double> vals;
  };
  // PoseGraph::quadraticMatrix (:345-350) as block-CSR, built by dpgo_build_Q_bsr (constructQ :381-491)
  const Bsr& quadraticMatrix() {
    if (!has_Q_) {
      Soa s = soa();
      std::vector<int32_t> pidx;
      for (const auto& kv : priors_) pidx.push_back((int32_t)kv.first);
      int nnzb = 0;
      check(dpgo_build_Q_bsr((int)id_, (int)d_, (int)n_, (int)meas_.size(), s.r1.data(), s.p1.data(), s.r2.data(),
                             s.p2.data(), s.R.data(), s.t.data(), s.kappa.data(), s.tau.data(), s.w.data(),
                             (int)pidx.size(), pidx.data(), 10000.0, 100.0, &nnzb, nullptr, nullptr, nullptr));
      const unsigned b = d_ + 1;
      Q_.rowptr.assign(n_ + 1, 0);
      Q_.colidx.assign(nnzb, 0);
      Q_.vals.assign((size_t)nnzb * b * b, 0.0);
      check(dpgo_build_Q_bsr((int)id_, (int)d_, (int)n_, (int)meas_.size(), s.r1.data(), s.p1.data(), s.r2.data(),
                             s.p2.data(), s.R.data(), s.t.data(), s.kappa.data(), s.tau.data(), s.w.data(),
                             (int)pidx.size(), pidx.data(), 10000.0, 100.0, &nnzb, Q_.rowptr.data(),
                             Q_.colidx.data(), Q_.vals.data()));
      has_Q_ = true;
    }
    return Q_;
  }
  // Sorted ids of the neighbour poses this agent's shared loop closures touch (nbr_shared_pose_ids_) = column order of
  // the device's neighbour tile buffer; and the ids of this agent's own public poses (local_shared_pose_ids_).
  std::vector<PoseID> neighborPoseIDs() const {
    std::vector<PoseID> ids;
    for (const auto& m : meas_)
      if (m.r1 != m.r2) ids.push_back(m.r1 == id_ ? PoseID((unsigned)m.r2, (unsigned)m.p2) : PoseID((unsigned)m.r1, (unsigned)m.p1));
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    return ids;
  }
  std::vector<unsigned> localSharedPoseIDs() const {
    std::vector<unsigned> ids;
    for (const auto& m : meas_)
      if (m.r1 != m.r2) ids.push_back((unsigned)(m.r1 == id_ ? m.p1 : m.p2));
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    return ids;
  }
  // Operator form of constructG (:493-580): G = G0 + Xnbr * C over the neighbour tile buffer (slot order above);
  // G0 carries the priors (:565-574).  Built by dpgo_build_G_coupling.
  struct Coupling {
    std::vector<PoseID> slots;
    Bsr C;
    Matrix G0;
  };
  Coupling couplingMatrix() {
    Coupling out;
    out.slots = neighborPoseIDs();
    std::map<PoseID, int32_t> index;
    for (size_t k = 0; k < out.slots.size(); ++k) index[out.slots[k]] = (int32_t)k;
    Soa s = soa();
    std::vector<int32_t> slot_of_edge(meas_.size(), -1);
    for (size_t e = 0; e < meas_.size(); ++e) {
      const auto& m = meas_[e];
      if (m.r1 != m.r2)
        slot_of_edge[e] = index[m.r1 == id_ ? PoseID((unsigned)m.r2, (unsigned)m.p2) : PoseID((unsigned)m.r1, (unsigned)m.p1)];
    }
    int nnzb = 0;
    check(dpgo_build_G_coupling((int)id_, (int)d_, (int)n_, (int)meas_.size(), s.r1.data(), s.p1.data(), s.r2.data(),
                                s.p2.data(), s.R.data(), s.t.data(), s.kappa.data(), s.tau.data(), s.w.data(),
                                slot_of_edge.data(), &nnzb, nullptr, nullptr, nullptr));
    const unsigned b = d_ + 1;
    out.C.rowptr.assign(n_ + 1, 0);
    out.C.colidx.assign(std::max(nnzb, 1), 0);
    out.C.vals.assign((size_t)std::max(nnzb, 1) * b * b, 0.0);
    check(dpgo_build_G_coupling((int)id_, (int)d_, (int)n_, (int)meas_.size(), s.r1.data(), s.p1.data(), s.r2.data(),
                                s.p2.data(), s.R.data(), s.t.data(), s.kappa.data(), s.tau.data(), s.w.data(),
                                slot_of_edge.data(), &nnzb, out.C.rowptr.data(), out.C.colidx.data(), out.C.vals.data()));
    out.C.colidx.resize(nnzb);
    out.C.vals.resize((size_t)nnzb * b * b);
    out.G0 = Matrix(r_, (size_t)b * n_);
    for (const auto& kv : priors_)
      for (unsigned c = 0; c < b; ++c)
        for (unsigned a = 0; a < r_; ++a) out.G0(a, (size_t)kv.first * b + c) -= kv.second(a, c) * (c < d_ ? 10000.0 : 100.0);
    return out;
  }
  // PoseGraph::linearMatrix (:359-364), constructG (:493-580); throws if an active neighbour pose is missing
  const Matrix& linearMatrix() {
    if (!has_G_) {
      const unsigned b = d_ + 1;
      G_ = Matrix(r_, (size_t)b * n_);
      for (const auto& m : meas_) {
        if (m.r1 == m.r2) continue;
        const bool outgoing = m.r1 == id_;
        const PoseID nid = outgoing ? PoseID(m.r2, m.p2) : PoseID(m.r1, m.p1);
        auto it = neighbor_poses_.find(nid);
        if (!isNeighborActive(nid.first) && (!use_inactive_neighbors_ || it == neighbor_poses_.end())) continue;  // :527-532
        if (it == neighbor_poses_.end()) throw Error(DPGO_ERR_STATE, "Missing active neighbor pose");
        const Matrix& Xn = it->second;  // r x (d+1)
        double T[4][4] = {}, om[4];
        for (unsigned p = 0; p < d_; ++p) {
          for (unsigned q = 0; q < d_; ++q) T[p][q] = m.R(p, q);
          T[p][d_] = m.t(p, 0);
          om[p] = m.weight * m.kappa;
        }
        T[d_][d_] = 1.0;
        om[d_] = m.weight * m.tau;
        const size_t col0 = (size_t)(outgoing ? m.p1 : m.p2) * b;
        for (unsigned c = 0; c < b; ++c)
          for (unsigned k = 0; k < b; ++k) {
            // outgoing: L = -Xj Om T^T (:533-537);  incoming: L = -Xi T Om (:558-562)
            const double coef = outgoing ? -om[k] * T[c][k] : -T[k][c] * om[c];
            if (coef == 0.0) continue;
            for (unsigned a = 0; a < r_; ++a) G_(a, col0 + c) += Xn(a, k) * coef;
          }
      }
      for (const auto& kv : priors_) {  // :565-574
        for (unsigned c = 0; c < b; ++c)
          for (unsigned a = 0; a < r_; ++a)
            G_(a, (size_t)kv.first * b + c) -= kv.second(a, c) * (c < d_ ? 10000.0 : 100.0);
      }
      has_G_ = true;
    }
    return G_;
  }

 private:
  struct Soa {
    std::vector<int32_t> r1, p1, r2, p2;
    std::vector<double> R, t, kappa, tau, w;
  };
  Soa soa() const {
    Soa s;
    for (const auto& m : meas_) {
      s.r1.push_back((int32_t)m.r1);
      s.p1.push_back((int32_t)m.p1);
      s.r2.push_back((int32_t)m.r2);
      s.p2.push_back((int32_t)m.p2);
      for (unsigned p = 0; p < d_; ++p)
        for (unsigned q = 0; q < d_; ++q) s.R.push_back(m.R(p, q));
      for (unsigned p = 0; p < d_; ++p) s.t.push_back(m.t(p, 0));
      s.kappa.push_back(m.kappa);
      s.tau.push_back(m.tau);
      // an edge with an inactive neighbour is out of the data matrices (unless useInactiveNeighbors and its pose is known):
      // carried as weight 0, which adds exact zeros where the reference skips the edge (Omega = w diag(kappa.., tau))
      double w = m.weight;
      if (m.r1 != m.r2) {
        const PoseID nid = m.r1 == id_ ? PoseID((unsigned)m.r2, (unsigned)m.p2) : PoseID((unsigned)m.r1, (unsigned)m.p1);
        if (!isNeighborActive(nid.first) && !(use_inactive_neighbors_ && neighbor_poses_.count(nid))) w = 0.0;
      }
      s.w.push_back(w);
    }
    return s;
  }
  unsigned id_, r_, d_, n_ = 0;
  std::vector<RelativeSEMeasurement> meas_;
  std::map<PoseID, Matrix> neighbor_poses_;
  std::map<unsigned, Matrix> priors_;
  std::map<unsigned, bool> neighbor_active_;
  bool use_inactive_neighbors_ = false;
  Bsr Q_;
  Matrix G_;
  bool has_Q_ = false, has_G_ = false;
  unsigned long q_version_ = 0;
};

// DPGO::QuadraticProblem: f(X) = 0.5 <Q, X^T X> + <X, G>.  Owns the device handle; unlike the reference
// (which rebuilds the problem every iteration, src/PGOAgent.cpp:968) keep it alive next to the PoseGraph.
class QuadraticProblem {
 public:
  // host_linear_term = false: G is not taken from PoseGraph::linearMatrix() on the host but built on the device from
  // the neighbour tile buffer (setCouplingFromPoseGraph + updateLinearMatrixFromNeighbors) -- the agent path.
  explicit QuadraticProblem(const std::shared_ptr<PoseGraph>& pose_graph, int device = 0, bool host_linear_term = true)
      : pose_graph_(pose_graph), host_G_(host_linear_term) {
    check(dpgo_problem_create(&h_, (int)pose_graph_->r(), (int)pose_graph_->d(), (int)pose_graph_->n(), device));
    refresh();
  }
  ~QuadraticProblem() { dpgo_problem_destroy(h_); }
  QuadraticProblem(const QuadraticProblem&) = delete;
  QuadraticProblem& operator=(const QuadraticProblem&) = delete;

  unsigned num_poses() const { return pose_graph_->n(); }
  unsigned dimension() const { return pose_graph_->d(); }
  unsigned relaxation_rank() const { return pose_graph_->r(); }
  dpgo_problem_t handle() const { return h_; }

  void refresh() {  // lazy-getter semantics of PoseGraph::quadraticMatrix / linearMatrix
    if (q_version_ != pose_graph_->qVersion()) {
      const auto& Q = pose_graph_->quadraticMatrix();
      check(dpgo_problem_set_Q_bsr(h_, (int)Q.colidx.size(), Q.rowptr.data(), Q.colidx.data(), Q.vals.data()));
      q_version_ = pose_graph_->qVersion();
    }
    if (!host_G_) return;
    if (pose_graph_->hasLinearTerm())
      check(dpgo_problem_set_G(h_, pose_graph_->linearMatrix().data()));
    else
      check(dpgo_problem_set_G(h_, nullptr));
  }
  // Upload the G operator once (pattern + values); returns the neighbour slot order.
  std::vector<PoseGraph::PoseID> setCouplingFromPoseGraph() {
    auto cp = pose_graph_->couplingMatrix();
    const int nnzb = (int)cp.C.colidx.size();
    check(dpgo_problem_set_G_coupling(h_, (int)cp.slots.size(), nnzb, cp.C.rowptr.data(),
                                      nnzb ? cp.C.colidx.data() : nullptr, nnzb ? cp.C.vals.data() : nullptr,
                                      cp.G0.data()));
    return cp.slots;
  }
  // constructG on the device from the neighbour tile buffer (device pointer, slot order of setCouplingFromPoseGraph)
  void updateLinearMatrixFromNeighbors(const double* nbr_tiles_dev) {
    check(dpgo_problem_update_G_from_neighbors_device(h_, nbr_tiles_dev));
  }
  // Explicit setup of the multilevel preconditioner for the CURRENT Q (the analogue of
  // PoseGraph::constructPreconditioner, src/PoseGraph.cpp:598-613).  Optional: a solve with
  // ROptParameters::precond = DPGO_PRECOND_MULTILEVEL builds / refreshes it by itself.  ks: aggregate sizes per
  // coarsening -- positive: index runs of that many nodes; a single negative entry -S: two levels with breadth-first-grown
  // graph aggregates of at most S poses (what the default builds up to 200 000 poses) --, empty = defaults.  Returns the
  // number of levels.
  int setupMultilevel(const std::vector<int>& ks = {}, double omega = 0.7, double shift = 0.1) {
    refresh();
    check(dpgo_problem_setup_multilevel(h_, (int)ks.size(), ks.empty() ? nullptr : ks.data(), omega, shift));
    int nl = 0;
    check(dpgo_problem_multilevel_info(h_, &nl, nullptr, nullptr, nullptr));
    return nl;
  }
  // Layout precond = DPGO_PRECOND_ADDITIVE (and AUTO, in the one-launch regime) uses for this block: lane groups per pose of
  // the kernel (0: the block does not fit 256 aggregates of one workgroup tile), slots per aggregate, growth size and merge
  // bound of the graph aggregates, number of aggregates = workgroups (dpgo_problem_additive_plan; host only).
  struct AdditivePlan {
    int lane_groups = 0, tile = 0, growth = 0, merge_cap = 0, aggregates = 0, graph = 0;
  };
  AdditivePlan additivePlan() {
    refresh();
    AdditivePlan a;
    check(dpgo_problem_additive_plan(h_, &a.lane_groups, &a.tile, &a.growth, &a.merge_cap, &a.aggregates, &a.graph));
    return a;
  }
  // Storage of the dense coarsest level (64 bits by default, 32 = opt-in) and of Q for its products (DPGO_SPMM_*; the
  // default picks the half-size symmetric storage for blocks that no longer fit the Infinity Cache).  See dpgo_hip.h.
  int multilevelCoarseBits(int bits = -1) {
    check(dpgo_problem_multilevel_coarse_bits(h_, &bits));
    return bits;
  }
  // Storage of the level-0 operator copies the V-cycle streams on HBM-bound blocks (32 bits by default, 64 = the fp64
  // originals; dpgo_problem_multilevel_operator_bits).  *active: the last solve's cycle streamed the fp32 copies.
  int multilevelOperatorBits(int bits = -1, bool* active = nullptr) {
    int act = 0;
    check(dpgo_problem_multilevel_operator_bits(h_, &bits, &act));
    if (active) *active = act != 0;
    return bits;
  }
  // Where the cost rule of DPGO_PRECOND_AUTO stands on this handle (dpgo_problem_auto_info): what a coupled block the
  // additive one-launch solve can hold is running and why.
  struct AutoInfo {
    int state = 0;  // 0 block-Jacobi, 1 additive on trial, 2 additive
    long long jacobi_units = 0;
    int reference_products = 0, switches = 0, backoff = 0, units_jacobi = 0, units_additive = 0;
  };
  AutoInfo autoInfo() {
    AutoInfo a;
    check(dpgo_problem_auto_info(h_, &a.state, &a.jacobi_units, &a.reference_products, &a.switches, &a.backoff,
                                 &a.units_jacobi, &a.units_additive));
    return a;
  }
  // What this handle currently runs and the library's DPGO_* switches, as text (dpgo_problem_describe).
  std::string describe() {
    std::string out(32768, '\0');
    check(dpgo_problem_describe(h_, &out[0], (int)out.size()));
    out.resize(std::strlen(out.c_str()));
    return out;
  }
  int multilevelPath() {  // DPGO_ML_PATH_* flags of the kernels a cycle of the current hierarchy runs
    int flags = 0;
    check(dpgo_problem_multilevel_path(h_, &flags));
    return flags;
  }
  int setSpmmVariant(int variant = DPGO_SPMM_AUTO) {
    refresh();
    int in_use = DPGO_SPMM_PLAIN;
    check(dpgo_problem_set_spmm_variant(h_, variant, &in_use));
    return in_use;
  }
  double f(const Matrix& Y) const {  // src/QuadraticProblem.cpp:29-35
    shape(Y);
    double out = 0;
    check(dpgo_problem_f(h_, Y.data(), &out));
    return out;
  }
  Matrix RieGrad(const Matrix& Y) const {  // :71-79
    shape(Y);
    Matrix out(Y.rows(), Y.cols());
    check(dpgo_problem_rie_grad(h_, Y.data(), out.data()));
    return out;
  }
  double RieGradNorm(const Matrix& Y) const {  // :81-83
    shape(Y);
    double out = 0;
    check(dpgo_problem_rie_grad_norm(h_, Y.data(), &out));
    return out;
  }
  // double*-based overloads of the ROPTLIB-typed virtuals (x->ObtainReadData() / ObtainWriteEntireData())
  void EucGrad(const double* x, double* g) const { check(dpgo_problem_euc_grad(h_, x, g)); }            // :43-47
  void EucHessianEta(const double*, const double* v, double* Hv) const { check(dpgo_problem_euc_hess(h_, v, Hv)); }  // :49-54
  void PreConditioner(const double* x, const double* in, double* out) const {                            // :56-69
    check(dpgo_problem_precondition(h_, DPGO_PRECOND_MULTILEVEL, 1e-1, x, in, out));
  }

 private:
  void shape(const Matrix& Y) const {  // CHECK_EQ (src/QuadraticProblem.cpp:30-31)
    if (Y.rows() != relaxation_rank() || Y.cols() != (size_t)(dimension() + 1) * num_poses())
      throw Error(DPGO_ERR_INVALID, "matrix shape mismatch");
  }
  std::shared_ptr<PoseGraph> pose_graph_;
  bool host_G_ = true;
  dpgo_problem_t h_ = nullptr;
  unsigned long q_version_ = (unsigned long)-1;
};

// DPGO::QuadraticOptimizer (src/QuadraticOptimizer.cpp)
class QuadraticOptimizer {
 public:
  QuadraticOptimizer(QuadraticProblem* p, ROptParameters params = ROptParameters()) : problem_(p), params_(params) {
    result_.success = false;
  }
  Matrix optimize(const Matrix& Y) {  // :26-48
    Matrix out(Y.rows(), Y.cols());
    dpgo_ropt_params c = params_.to_c();
    dpgo_ropt_result res;
    check(dpgo_optimize(problem_->handle(), &c, Y.data(), out.data(), &res));
    result_ = ROPTResult(res.success != 0, res.fInit, res.gradNormInit, res.fOpt, res.gradNormOpt, res.elapsedMs);
    result_.tCGStatus = res.tCGStatus;
    result_.tcgIterations = res.tcg_iterations;
    result_.precondUsed = res.precond_used;
    result_.rtrIterations = res.rtr_iterations;
    return out;
  }
  // Device-resident flavour: X_dev (r x (d+1)n doubles in HBM) is updated in place.
  ROPTResult optimizeDevice(double* X_dev) {
    dpgo_ropt_params c = params_.to_c();
    dpgo_ropt_result res;
    check(dpgo_optimize_device(problem_->handle(), &c, X_dev, &res));
    result_ = ROPTResult(res.success != 0, res.fInit, res.gradNormInit, res.fOpt, res.gradNormOpt, res.elapsedMs);
    result_.tCGStatus = res.tCGStatus;
    result_.tcgIterations = res.tcg_iterations;
    result_.precondUsed = res.precond_used;
    result_.rtrIterations = res.rtr_iterations;
    return result_;
  }
  // The same in two halves (dpgo_optimize_device_begin / _end): `begin` enqueues the solve on the handle's stream when it is a
  // one-launch solve (G is rebuilt from the neighbour tile buffer first when one is given) and returns; `end` waits and
  // returns the result.  For callers that enqueue a whole sweep -- exchange, solve, next exchange, next solve -- without a
  // host wait in between (dpgo_amd/agent.py, RBCDCluster.sweep).
  void optimizeDeviceBegin(double* X_dev, const double* nbr_tiles_dev = nullptr) {
    dpgo_ropt_params c = params_.to_c();
    check(dpgo_optimize_device_begin(problem_->handle(), &c, X_dev, nbr_tiles_dev));
  }
  ROPTResult optimizeDeviceEnd() {
    dpgo_ropt_result res;
    check(dpgo_optimize_device_end(problem_->handle(), &res));
    result_ = ROPTResult(res.success != 0, res.fInit, res.gradNormInit, res.fOpt, res.gradNormOpt, res.elapsedMs);
    result_.tCGStatus = res.tCGStatus;
    result_.tcgIterations = res.tcg_iterations;
    result_.precondUsed = res.precond_used;
    result_.rtrIterations = res.rtr_iterations;
    return result_;
  }
  void setProblem(QuadraticProblem* p) { problem_ = p; }
  void setVerbose(bool v) { params_.verbose = v; }
  void setAlgorithm(ROptParameters::ROptMethod alg) { params_.method = alg; }
  void setRGDStepsize(double s) { params_.RGD_stepsize = s; }
  void setRTRIterations(int iter) { params_.RTR_iterations = iter; }
  void setGradientNormTolerance(double tol) { params_.gradnorm_tol = tol; }
  void setRTRInitialRadius(double radius) { params_.RTR_initial_radius = radius; }
  void setRTRtCGIterations(int iter) { params_.RTR_tCG_iterations = iter; }
  ROPTResult getOptResult() const { return result_; }

 private:
  QuadraticProblem* problem_;
  ROptParameters params_;
  ROPTResult result_;
};

// DPGO::LiftedSEManifold (include/DPGO/manifold/LiftedSEManifold.h:28-43)
class LiftedSEManifold {
 public:
  LiftedSEManifold(unsigned r, unsigned d, unsigned n, int device = 0) : r_(r), d_(d), n_(n), device_(device) {}
  Matrix project(const Matrix& M) const {  // src/manifold/LiftedSEManifold.cpp:34-45
    if (M.rows() != r_ || M.cols() != (size_t)(d_ + 1) * n_) throw Error(DPGO_ERR_INVALID, "matrix shape mismatch");
    Matrix out(M.rows(), M.cols());
    check(dpgo_manifold_project((int)r_, (int)d_, (int)n_, M.data(), out.data(), device_));
    return out;
  }
  Matrix Projection(const Matrix& X, const Matrix& V) const {  // ROPTLIB ProductManifold::Projection
    Matrix out(X.rows(), X.cols());
    check(dpgo_manifold_tangent_project((int)r_, (int)d_, (int)n_, X.data(), V.data(), out.data(), device_));
    return out;
  }
  Matrix Retraction(const Matrix& X, const Matrix& eta, double scale = 1.0) const {  // ProductManifold::Retraction
    Matrix out(X.rows(), X.cols());
    check(dpgo_manifold_retract((int)r_, (int)d_, (int)n_, X.data(), eta.data(), scale, out.data(), device_));
    return out;
  }

 private:
  unsigned r_, d_, n_;
  int device_;
};

// Writable view of a block of columns of a column-major matrix (the role Eigen::Ref<Matrix> plays in the reference).
class MatrixRef {
 public:
  MatrixRef(double* p, size_t rows, size_t cols) : p_(p), r_(rows), c_(cols) {}
  size_t rows() const { return r_; }
  size_t cols() const { return c_; }
  double* data() { return p_; }
  double& operator()(size_t i, size_t j) { return p_[j * r_ + i]; }
  double operator()(size_t i, size_t j) const { return p_[j * r_ + i]; }
  MatrixRef& operator=(const Matrix& m) {
    if (m.rows() != r_ || m.cols() != c_) throw Error(DPGO_ERR_INVALID, "matrix shape mismatch");
    std::memcpy(p_, m.data(), sizeof(double) * r_ * c_);
    return *this;
  }
  operator Matrix() const {
    Matrix m(r_, c_);
    std::memcpy(m.data(), p_, sizeof(double) * r_ * c_);
    return m;
  }

 private:
  double* p_;
  size_t r_, c_;
};

// DPGO::LiftedSEVariable (include/DPGO/manifold/LiftedSEVariable.h:33-119, src/manifold/LiftedSEVariable.cpp:15-67):
// n lifted poses X = [Y1 p1 ... Yn pn], r x (d+1)n.  The reference stores them in a ROPTLIB ProductElement whose flat
// memory IS this column-major matrix (tests/testEigenMap.cpp:12-36); here the matrix is the storage, and it is also
// exactly what the device kernels read (n consecutive pose tiles) -- no conversion on either side.
class LiftedSEVariable {
 public:
  LiftedSEVariable(unsigned r, unsigned d, unsigned n) : r_(r), d_(d), n_(n), X_(r, (size_t)(d + 1) * n) {
    if (r < d) throw Error(DPGO_ERR_INVALID, "CHECK(r >= d) failed");
    for (unsigned i = 0; i < n; ++i)  // Y_i = [I_d; 0], p_i = 0 (src/manifold/LiftedSEVariable.cpp:22-27)
      for (unsigned k = 0; k < d; ++k) X_(k, (size_t)i * (d + 1) + k) = 1.0;
  }
  unsigned r() const { return r_; }
  unsigned d() const { return d_; }
  unsigned n() const { return n_; }
  Matrix getData() const { return X_; }
  void setData(const Matrix& X) {  // CHECKs of :59-61
    if (X.rows() != r_ || X.cols() != (size_t)(d_ + 1) * n_) throw Error(DPGO_ERR_INVALID, "matrix shape mismatch");
    X_ = X;
  }
  double* data() { return X_.data(); }  // what var()->ObtainWriteEntireData() returns in the reference
  const double* data() const { return X_.data(); }
  MatrixRef pose(unsigned index) { return MatrixRef(col(index, 0), r_, d_ + 1); }
  Matrix pose(unsigned index) const { return X_.block(0, (size_t)chk(index) * (d_ + 1), r_, d_ + 1); }
  MatrixRef rotation(unsigned index) { return MatrixRef(col(index, 0), r_, d_); }
  Matrix rotation(unsigned index) const { return X_.block(0, (size_t)chk(index) * (d_ + 1), r_, d_); }
  MatrixRef translation(unsigned index) { return MatrixRef(col(index, d_), r_, 1); }
  Matrix translation(unsigned index) const { return X_.block(0, (size_t)chk(index) * (d_ + 1) + d_, r_, 1); }

 private:
  unsigned chk(unsigned index) const {
    if (index >= n_) throw Error(DPGO_ERR_INVALID, "CHECK(index < n_) failed");
    return index;
  }
  double* col(unsigned index, unsigned c) { return X_.data() + ((size_t)chk(index) * (d_ + 1) + c) * r_; }
  unsigned r_, d_, n_;
  Matrix X_;
};

// DPGO::LiftedSEVector (include/DPGO/manifold/LiftedSEVector.h:28-47, src/manifold/LiftedSEVector.cpp:16-54): a tangent
// vector of the product manifold in the ambient (extrinsic) representation, same r x (d+1)n layout, zero-initialised.
class LiftedSEVector {
 public:
  LiftedSEVector(int r, int d, int n) : r_(r), d_(d), n_(n), V_((size_t)r, (size_t)(d + 1) * n) {}
  Matrix getData() const { return V_; }
  void setData(const Matrix& Y) {
    if (Y.rows() != (size_t)r_ || Y.cols() != (size_t)(d_ + 1) * n_) throw Error(DPGO_ERR_INVALID, "matrix shape mismatch");
    V_ = Y;
  }
  double* data() { return V_.data(); }  // vec()->ObtainWriteEntireData()
  const double* data() const { return V_.data(); }

 private:
  int r_, d_, n_;
  Matrix V_;
};

// ---------------------------------------------------------------------------------------------------------
// The part of DPGO::PGOAgent that drives the hot path (include/DPGO/PGOAgent.h:250-548, src/PGOAgent.cpp:376-432,
// 880-995): iterate(bool) with Nesterov acceleration and restarts, the public-pose dictionaries, setX / getX.
// X, XPrev, Y, V and the neighbour tile buffers live in HBM; updateY / updateV are one fused kernel each (linear
// combination + LiftedSEManifold::project), updateX is constructG on the device + QuadraticOptimizer::optimize on the
// device iterate.  Public poses cross the ABI as the reference's PoseDict (host maps): a few hundred small tiles.
// Status and the termination vote (PGOAgentStatus, getStatus / setNeighborStatus / shouldTerminate, src/PGOAgent.cpp:399-420,
// 846-878): relativeChange is LiftedPoseArray::maxTranslationDistance(X, XPrev) evaluated on the device.
// Not mirrored: the state machine / initialisation in a global frame / threads / logging / robust weights.
struct PGOAgentParameters {  // the fields of DPGO::PGOAgentParameters (PGOAgent.h:49-160) this path reads
  unsigned d = 3, r = 5, numRobots = 1;
  bool acceleration = false;
  unsigned restartInterval = 30;
  ROptParameters localOptimizationParams;
  unsigned maxNumIters = 500;   // PGOAgent.h:135
  double relChangeTol = 5e-3;   // PGOAgent.h:136
  PGOAgentParameters(unsigned dIn, unsigned rIn, unsigned numRobotsIn = 1, ROptParameters local = ROptParameters(),
                     bool accel = false, unsigned restart = 30, unsigned maxIters = 500, double changeTol = 5e-3)
      : d(dIn), r(rIn), numRobots(numRobotsIn), acceleration(accel), restartInterval(restart),
        localOptimizationParams(local), maxNumIters(maxIters), relChangeTol(changeTol) {}
};

enum class PGOAgentState { WAIT_FOR_DATA, WAIT_FOR_INITIALIZATION, INITIALIZED };  // PGOAgent.h:36-47

struct PGOAgentStatus {  // DPGO::PGOAgentStatus (PGOAgent.h:196-227)
  unsigned agentID = 0;
  PGOAgentState state = PGOAgentState::WAIT_FOR_DATA;
  unsigned instanceNumber = 0;
  unsigned iterationNumber = 0;
  bool readyToTerminate = false;
  double relativeChange = 0;
  explicit PGOAgentStatus(unsigned id = 0, PGOAgentState s = PGOAgentState::WAIT_FOR_DATA, unsigned instance = 0,
                          unsigned iteration = 0, bool ready_to_terminate = false, double relative_change = 0)
      : agentID(id), state(s), instanceNumber(instance), iterationNumber(iteration),
        readyToTerminate(ready_to_terminate), relativeChange(relative_change) {}
};

using PoseID = PoseGraph::PoseID;
using PoseDict = std::map<PoseID, Matrix>;  // DPGO::PoseDict (manifold/Poses.h:218): PoseID -> r x (d+1) lifted pose

class PGOAgent {
 public:
  PGOAgent(unsigned ID, const PGOAgentParameters& params, int device = 0)
      : id_(ID), prm_(params), device_(device), pg_(std::make_shared<PoseGraph>(ID, params.r, params.d)) {}
  ~PGOAgent() { release(); }
  PGOAgent(const PGOAgent&) = delete;
  PGOAgent& operator=(const PGOAgent&) = delete;

  unsigned getID() const { return id_; }
  unsigned num_poses() const { return pg_->n(); }
  unsigned dimension() const { return prm_.d; }
  unsigned relaxation_rank() const { return prm_.r; }
  unsigned iteration_number() const { return iteration_; }

  // PGOAgent::getStatus / setNeighborStatus (PGOAgent.h:300-310, src/PGOAgent.cpp:604-610) and shouldTerminate (:846-878)
  PGOAgentStatus getStatus() const { return status_; }
  void setNeighborStatus(const PGOAgentStatus& status) { team_[status.agentID] = status; }
  // PGOAgent::isRobotActive / setRobotActive (src/PGOAgent.cpp:1167-1184): for a NEIGHBOUR the shared edges with it leave
  // (or re-enter) Q and G -- the pose graph drops its data matrices and the device handle is refreshed (values only)
  bool isRobotActive(unsigned robot_id) const { return robot_id < prm_.numRobots && !inactive_.count(robot_id); }
  void setRobotActive(unsigned robot_id, bool active = true) {
    if (robot_id >= prm_.numRobots) return;
    if (active) inactive_.erase(robot_id); else inactive_.insert(robot_id);
    if (pg_ && pg_->hasNeighbor(robot_id)) {
      const unsigned long before = pg_->qVersion();
      pg_->setNeighborActive(robot_id, active);
      if (pg_->qVersion() != before && problem_) {
        problem_->refresh();                   // Q's values (same block pattern)
        problem_->setCouplingFromPoseGraph();  // the coupling blocks of G (same slot order)
      }
    }
  }
  bool shouldTerminate() {
    if (iteration_number() >= prm_.maxNumIters) return true;
    team_[id_] = status_;
    for (unsigned robot = 0; robot < prm_.numRobots; ++robot) {
      if (!isRobotActive(robot)) continue;  // :861-862
      auto it = team_.find(robot);
      if (it == team_.end()) return false;
      if (it->second.state != PGOAgentState::INITIALIZED) return false;
      if (!it->second.readyToTerminate) return false;
    }
    return true;
  }

  // PGOAgent::setMeasurements (:263): odometry, private and shared loop closures of this robot
  void setMeasurements(const std::vector<RelativeSEMeasurement>& inputOdometry,
                       const std::vector<RelativeSEMeasurement>& inputPrivateLoopClosures,
                       const std::vector<RelativeSEMeasurement>& inputSharedLoopClosures) {
    std::vector<RelativeSEMeasurement> all(inputOdometry);
    all.insert(all.end(), inputPrivateLoopClosures.begin(), inputPrivateLoopClosures.end());
    all.insert(all.end(), inputSharedLoopClosures.begin(), inputSharedLoopClosures.end());
    setMeasurements(all);
  }
  void setMeasurements(const std::vector<RelativeSEMeasurement>& all) {
    release();
    pg_->setMeasurements(all);
    const size_t T = (size_t)(prm_.d + 1) * prm_.r, n = pg_->n();
    problem_.reset(new QuadraticProblem(pg_, device_, /*host_linear_term=*/false));
    check(dpgo_problem_set_stream(problem_->handle(), nullptr));  // one stream (the default one) for everything below
    optimizer_.reset(new QuadraticOptimizer(problem_.get(), prm_.localOptimizationParams));
    slots_ = problem_->setCouplingFromPoseGraph();
    pub_ = pg_->localSharedPoseIDs();
    for (double** v : {&X_, &XPrev_, &Y_, &V_}) check(dpgo_device_malloc((void**)v, sizeof(double) * T * n, device_));
    check(dpgo_device_malloc((void**)&rel_, sizeof(double), device_));
    status_ = PGOAgentStatus(id_, PGOAgentState::INITIALIZED);
    team_.clear();
    const size_t ns = std::max<size_t>(slots_.size(), 1), np = std::max<size_t>(pub_.size(), 1);
    check(dpgo_device_malloc((void**)&nbr_, sizeof(double) * T * ns, device_));
    check(dpgo_device_malloc((void**)&nbr_aux_, sizeof(double) * T * ns, device_));
    check(dpgo_device_malloc((void**)&pack_, sizeof(double) * T * np, device_));
    check(dpgo_device_malloc((void**)&pub_idx_, sizeof(int32_t) * np, device_));
    std::vector<int32_t> idx(pub_.begin(), pub_.end());
    if (!idx.empty()) check(dpgo_device_memcpy(pub_idx_, idx.data(), sizeof(int32_t) * idx.size(), DPGO_COPY_H2D, nullptr));
    nbr_h_.assign(T * ns, 0.0);
    nbr_aux_h_.assign(T * ns, 0.0);
    // Y_i = [I; 0] like PGOAgent's default initial iterate
    setX(LiftedSEVariable(prm_.r, prm_.d, (unsigned)n).getData());
  }

  // PGOAgent::setX (:465) -- also (re)starts the acceleration state (initializeAcceleration, src/PGOAgent.cpp:899-908)
  void setX(const Matrix& Xin) {
    const size_t bytes = shape(Xin);
    check(dpgo_device_memcpy(X_, Xin.data(), bytes, DPGO_COPY_H2D, nullptr));
    for (double* v : {XPrev_, Y_, V_}) check(dpgo_device_memcpy(v, X_, bytes, DPGO_COPY_D2D, nullptr));
    gamma_ = alpha_ = 0.0;
    iteration_ = 0;
  }
  bool getX(Matrix& Mout) {  // :477
    Mout = Matrix(prm_.r, (size_t)(prm_.d + 1) * pg_->n());
    check(dpgo_device_memcpy(Mout.data(), X_, sizeof(double) * Mout.rows() * Mout.cols(), DPGO_COPY_D2H, nullptr));
    return true;
  }
  // PGOAgent::getSharedPoseDict / getAuxSharedPoseDict (:442, :454): this robot's public poses (of X, of Y)
  bool getSharedPoseDict(PoseDict& map) { return sharedDict(X_, map); }
  bool getAuxSharedPoseDict(PoseDict& map) { return sharedDict(prm_.acceleration ? Y_ : X_, map); }
  // PGOAgent::updateNeighborPoses / updateAuxNeighborPoses (:532, :538)
  void updateNeighborPoses(unsigned neighborID, const PoseDict& poseDict) { fill(neighborID, poseDict, nbr_h_, nbr_dirty_); }
  void updateAuxNeighborPoses(unsigned neighborID, const PoseDict& poseDict) { fill(neighborID, poseDict, nbr_aux_h_, aux_dirty_); }

  // PGOAgent::iterate (src/PGOAgent.cpp:376-432), INITIALIZED state
  bool iterate(bool doOptimization = true) {
    iteration_ += 1;
    const size_t bytes = sizeof(double) * (size_t)(prm_.d + 1) * prm_.r * pg_->n();
    check(dpgo_device_memcpy(XPrev_, X_, bytes, DPGO_COPY_D2D, nullptr));  // XPrev = X (:386)
    if (!prm_.acceleration) {
      const bool ok = updateX(doOptimization, false);
      if (doOptimization) updateStatus(ok);
      return ok;
    }
    const double N = (double)prm_.numRobots;
    gamma_ = (1 + std::sqrt(1 + 4 * N * N * gamma_ * gamma_)) / (2 * N);  // updateGamma (:910-914)
    alpha_ = 1 / (gamma_ * N);                                              // updateAlpha (:916-920)
    combineProject(1 - alpha_, X_, alpha_, V_, 0.0, nullptr, Y_);           // updateY (:922-928)
    const bool ok = updateX(doOptimization, true);
    combineProject(1.0, V_, gamma_, X_, -gamma_, Y_, V_);                   // updateV (:930-936)
    if ((iteration_ + 1) % prm_.restartInterval == 0) {                     // shouldRestart (:880-885)
      check(dpgo_device_memcpy(X_, XPrev_, bytes, DPGO_COPY_D2D, nullptr));  // restartNesterovAcceleration (:887-897)
      updateX(doOptimization, false);
      check(dpgo_device_memcpy(V_, X_, bytes, DPGO_COPY_D2D, nullptr));
      check(dpgo_device_memcpy(Y_, X_, bytes, DPGO_COPY_D2D, nullptr));
      gamma_ = alpha_ = 0.0;
    }
    if (doOptimization) updateStatus(ok);
    return ok;
  }
  const ROPTResult& latestResult() const { return result_; }

  // This agent's block of the central objective with the neighbour poses received last: 0.5 (<X Q, X> + <X, G>) and
  // |rgrad_a|^2 -- the agent-local gradient IS its block of the central gradient, so a driver can evaluate the central
  // cost / gradient norm / greedy selection (examples/MultiRobotExample.cpp:220-247) from these.
  void localTerms(double* half_cost, double* gradnorm_sq) {
    syncNeighbours(false);
    if (!slots_.empty()) problem_->updateLinearMatrixFromNeighbors(nbr_);
    double xqx = 0, xg = 0, g2 = 0;
    check(dpgo_problem_eval_terms_device(problem_->handle(), X_, &xqx, &xg, &g2));
    if (half_cost) *half_cost = 0.5 * (xqx + xg);
    if (gradnorm_sq) *gradnorm_sq = g2;
  }

 private:
  size_t shape(const Matrix& M) const {
    if (M.rows() != prm_.r || M.cols() != (size_t)(prm_.d + 1) * pg_->n()) throw Error(DPGO_ERR_INVALID, "matrix shape mismatch");
    return sizeof(double) * M.rows() * M.cols();
  }
  void release() {
    for (double** v : {&X_, &XPrev_, &Y_, &V_, &nbr_, &nbr_aux_, &pack_, &rel_}) {
      if (*v) dpgo_device_free(*v);
      *v = nullptr;
    }
    if (pub_idx_) dpgo_device_free(pub_idx_);
    pub_idx_ = nullptr;
    optimizer_.reset();
    problem_.reset();
  }
  void updateStatus(bool success) {  // src/PGOAgent.cpp:399-420 (L2 cost: no weight statistics)
    double rel = 0.0;
    check(dpgo_max_translation_distance_device((int)prm_.r, (int)prm_.d, (int)pg_->n(), X_, XPrev_, rel_, &rel, nullptr));
    status_ = PGOAgentStatus(id_, PGOAgentState::INITIALIZED, 0, iteration_, success && rel <= prm_.relChangeTol, rel);
  }
  bool sharedDict(const double* src, PoseDict& map) {
    map.clear();
    if (pub_.empty()) return true;
    const size_t T = (size_t)(prm_.d + 1) * prm_.r;
    check(dpgo_gather_tiles_device((int)prm_.r, (int)prm_.d, src, pub_idx_, (int)pub_.size(), pack_, nullptr));
    std::vector<double> host(T * pub_.size());
    check(dpgo_device_memcpy(host.data(), pack_, sizeof(double) * host.size(), DPGO_COPY_D2H, nullptr));
    for (size_t k = 0; k < pub_.size(); ++k) {
      Matrix M(prm_.r, prm_.d + 1);
      std::memcpy(M.data(), host.data() + k * T, sizeof(double) * T);
      map[PoseID(id_, pub_[k])] = M;
    }
    return true;
  }
  void fill(unsigned neighborID, const PoseDict& dict, std::vector<double>& host, bool& dirty) {
    const size_t T = (size_t)(prm_.d + 1) * prm_.r;
    for (size_t k = 0; k < slots_.size(); ++k) {
      if (slots_[k].first != neighborID) continue;
      auto it = dict.find(slots_[k]);
      if (it == dict.end()) continue;
      if (it->second.rows() != prm_.r || it->second.cols() != prm_.d + 1) throw Error(DPGO_ERR_INVALID, "matrix shape mismatch");
      std::memcpy(host.data() + k * T, it->second.data(), sizeof(double) * T);
      dirty = true;
    }
  }
  void syncNeighbours(bool aux) {
    bool& dirty = aux ? aux_dirty_ : nbr_dirty_;
    if (!dirty) return;
    auto& host = aux ? nbr_aux_h_ : nbr_h_;
    check(dpgo_device_memcpy(aux ? nbr_aux_ : nbr_, host.data(), sizeof(double) * host.size(), DPGO_COPY_H2D, nullptr));
    dirty = false;
  }
  void combineProject(double a, const double* A, double b, const double* B, double c, const double* Cm, double* out) {
    check(dpgo_axpby_project_device((int)prm_.r, (int)prm_.d, (int)pg_->n(), a, A, b, B, c, Cm, 1, out, nullptr));
  }
  bool updateX(bool doOptimization, bool acceleration) {  // PGOAgent::updateX (:938-995)
    const size_t bytes = sizeof(double) * (size_t)(prm_.d + 1) * prm_.r * pg_->n();
    if (!doOptimization) {
      if (acceleration) check(dpgo_device_memcpy(X_, Y_, bytes, DPGO_COPY_D2D, nullptr));
      return true;
    }
    if (!slots_.empty()) {
      syncNeighbours(acceleration);
      problem_->updateLinearMatrixFromNeighbors(acceleration ? nbr_aux_ : nbr_);
    }
    if (acceleration) check(dpgo_device_memcpy(X_, Y_, bytes, DPGO_COPY_D2D, nullptr));  // X0 = Y (:973-978)
    result_ = optimizer_->optimizeDevice(X_);
    return result_.success;
  }

  unsigned id_;
  PGOAgentParameters prm_;
  int device_;
  std::shared_ptr<PoseGraph> pg_;
  std::unique_ptr<QuadraticProblem> problem_;
  std::unique_ptr<QuadraticOptimizer> optimizer_;
  std::vector<PoseID> slots_;   // neighbour poses, slot order of the tile buffers
  std::vector<unsigned> pub_;   // my public frames
  double *X_ = nullptr, *XPrev_ = nullptr, *Y_ = nullptr, *V_ = nullptr, *nbr_ = nullptr, *nbr_aux_ = nullptr, *pack_ = nullptr;
  double* rel_ = nullptr;  // device scalar: relativeChange of the last optimising iterate
  PGOAgentStatus status_;
  std::map<unsigned, PGOAgentStatus> team_;
  std::set<unsigned> inactive_;  // robots switched off by setRobotActive(id, false)
  int32_t* pub_idx_ = nullptr;
  std::vector<double> nbr_h_, nbr_aux_h_;
  bool nbr_dirty_ = false, aux_dirty_ = false;
  double gamma_ = 0.0, alpha_ = 0.0;
  unsigned iteration_ = 0;
  ROPTResult result_;
};

// ---------------------------------------------------------------------------------------------------------
// Rounding: PGOAgent::getTrajectoryInLocalFrame / getTrajectoryInGlobalFrame (src/PGOAgent.cpp:718-767).
// X: r x (d+1)n; anchor: r x (d+1) lifted pose or nullptr (frame of pose 0).  Returns d x (d+1)n.
inline Matrix roundTrajectory(const Matrix& X, unsigned d, const Matrix* anchor = nullptr, int device = 0) {
  const unsigned r = (unsigned)X.rows();
  if (X.cols() % (d + 1) != 0) throw Error(DPGO_ERR_INVALID, "matrix shape mismatch");
  if (anchor && (anchor->rows() != r || anchor->cols() != d + 1))
    throw Error(DPGO_ERR_INVALID, "CHECK(M.rows() == relaxation_rank() && M.cols() == dimension() + 1) failed");
  const unsigned n = (unsigned)(X.cols() / (d + 1));
  Matrix T(d, (size_t)(d + 1) * n);
  check(dpgo_round_trajectory((int)r, (int)d, (int)n, X.data(), anchor ? anchor->data() : nullptr, T.data(), device));
  return T;
}

// ---------------------------------------------------------------------------------------------------------
// DPGO::RobustCostParameters / RobustCost (include/DPGO/DPGO_robust.h:20-133, src/DPGO_robust.cpp:49-134)
struct RobustCostParameters {
  enum class Type { L2, L1, TLS, Huber, GM, GNC_TLS };
  Type costType = Type::L2;
  unsigned GNCMaxNumIters = 20;
  double GNCBarc = 5.0, GNCMuStep = 1.4, GNCInitMu = 1e-4, HuberThreshold = 3.0, TLSThreshold = 10.0;
};
// chi2inv (include/DPGO/DPGO_utils.h:146-153, src/DPGO_utils.cpp:509-512: boost's chi-squared quantile; "equivalent to chi2inv
// in Matlab"): x with P(dof / 2, x / 2) = quantile, P = the regularised lower incomplete gamma function (series below a + 1,
// Lentz's continued fraction above), by bisection refined with Newton steps.
namespace detail {
inline double gammaP(double a, double x) {
  if (x <= 0) return 0.0;
  const double lg = std::lgamma(a);
  if (x < a + 1) {
    double ap = a, sum = 1.0 / a, del = sum;
    for (int k = 0; k < 1000; ++k) {
      ap += 1;
      del *= x / ap;
      sum += del;
      if (std::fabs(del) < std::fabs(sum) * 1e-17) break;
    }
    return sum * std::exp(-x + a * std::log(x) - lg);
  }
  const double tiny = 1e-300;
  double b = x + 1 - a, c = 1 / tiny, dd = 1 / b, h = dd;
  for (int k = 1; k < 1000; ++k) {
    const double an = -k * (k - a);
    b += 2;
    dd = an * dd + b;
    if (std::fabs(dd) < tiny) dd = tiny;
    c = b + an / c;
    if (std::fabs(c) < tiny) c = tiny;
    dd = 1 / dd;
    const double del = dd * c;
    h *= del;
    if (std::fabs(del - 1) < 1e-16) break;
  }
  return 1.0 - std::exp(-x + a * std::log(x) - lg) * h;
}
}  // namespace detail
inline double chi2inv(double quantile, size_t dof) {
  if (!(quantile >= 0.0) || !(quantile < 1.0) || dof == 0) throw Error(DPGO_ERR_INVALID, "chi2inv: quantile in [0, 1), dof > 0");
  if (quantile == 0.0) return 0.0;
  const double a = 0.5 * (double)dof;
  double lo = 0.0, hi = std::max(1.0, (double)dof);
  while (detail::gammaP(a, 0.5 * hi) < quantile) hi *= 2;
  for (int it = 0; it < 200 && hi - lo > 1e-15 * hi; ++it) {
    const double mid = 0.5 * (lo + hi);
    (detail::gammaP(a, 0.5 * mid) < quantile ? lo : hi) = mid;
  }
  return 0.5 * (lo + hi);
}

class RobustCost {
 public:
  explicit RobustCost(const RobustCostParameters& p) : params_(p), mu_(p.GNCInitMu) {}
  // RobustCost::computeErrorThresholdAtQuantile (include/DPGO/DPGO_robust.h:116-123): the GNC threshold barc for a 3-D
  // measurement whose squared error is chi-squared with 6 degrees of freedom
  static double computeErrorThresholdAtQuantile(double quantile, size_t dimension) {
    if (dimension != 3) throw Error(DPGO_ERR_INVALID, "CHECK_EQ(dimension, 3) failed: quantile function currently only supports 3D problem.");
    if (!(quantile > 0)) throw Error(DPGO_ERR_INVALID, "CHECK_GT(quantile, 0) failed");
    return quantile < 1 ? std::sqrt(chi2inv(quantile, 6)) : 1e5;
  }
  double weight(double r) const {  // src/DPGO_robust.cpp:54-98
    using T = RobustCostParameters::Type;
    switch (params_.costType) {
      case T::L2: return 1.0;
      case T::L1: return 1.0 / r;
      case T::Huber: return r < params_.HuberThreshold ? 1.0 : params_.HuberThreshold / r;
      case T::TLS: return r < params_.TLSThreshold ? 1.0 : 0.0;
      case T::GM: { const double a = 1 + r * r; return 1.0 / (a * a); }
      case T::GNC_TLS: {
        const double rSq = r * r, bSq = params_.GNCBarc * params_.GNCBarc;
        const double upper = (mu_ + 1) / mu_ * bSq, lower = mu_ / (mu_ + 1) * bSq;
        if (rSq >= upper) return 0.0;
        if (rSq <= lower) return 1.0;
        return std::sqrt(bSq * mu_ * (mu_ + 1) / rSq) - mu_;
      }
    }
    throw std::runtime_error("weight function for selected cost function is not implemented !");  // :95
  }
  void update() {  // :106-121 (only GNC_TLS has state)
    if (params_.costType != RobustCostParameters::Type::GNC_TLS) return;
    ++iteration_;
    if (iteration_ > params_.GNCMaxNumIters) return;
    mu_ *= params_.GNCMuStep;
  }
  double mu() const { return mu_; }

 private:
  RobustCostParameters params_;
  double mu_;
  unsigned iteration_ = 0;
};

// DPGO::solvePGO / solveRobustPGO (include/DPGO/DPGO_solver.h:100-123, src/DPGO_solver.cpp:305-412).  T0 is
// required here (the reference falls back to its SPQR-based chordal initialisation, which is outside the path).
struct solveRobustPGOParams {
  ROptParameters opt_params;
  RobustCostParameters robust_params;
  bool verbose = false;
  solveRobustPGOParams() { robust_params.costType = RobustCostParameters::Type::GNC_TLS; }
};

// chordalInitialization / odometryInitialization (src/DPGO_solver.cpp:220-303) for the poses of one robot: d x (d+1)n.
// The chordal relaxation is solved on the device (dpgo_chordal_initialization).
namespace detail {
struct EdgeSoa {
  std::vector<int32_t> p1, p2;
  std::vector<double> R, t, kappa, tau;
  unsigned d = 0, n = 0;
  explicit EdgeSoa(const std::vector<RelativeSEMeasurement>& ms) {
    if (ms.empty()) throw Error(DPGO_ERR_INVALID, "no measurements");
    d = (unsigned)ms[0].R.rows();
    for (const auto& m : ms) {
      p1.push_back((int32_t)m.p1);
      p2.push_back((int32_t)m.p2);
      for (unsigned p = 0; p < d; ++p)
        for (unsigned q = 0; q < d; ++q) R.push_back(m.R(p, q));
      for (unsigned p = 0; p < d; ++p) t.push_back(m.t(p, 0));
      kappa.push_back(m.kappa);
      tau.push_back(m.tau);
      n = std::max<unsigned>(n, (unsigned)std::max(m.p1, m.p2) + 1);
    }
  }
};
}  // namespace detail
inline Matrix chordalInitialization(const std::vector<RelativeSEMeasurement>& measurements, int device = 0) {
  detail::EdgeSoa s(measurements);
  Matrix T(s.d, (size_t)(s.d + 1) * s.n);
  check(dpgo_chordal_initialization((int)s.d, (int)s.n, (int)measurements.size(), s.p1.data(), s.p2.data(), s.R.data(),
                                    s.t.data(), s.kappa.data(), s.tau.data(), 0.0, 0, T.data(), nullptr, device));
  return T;
}
inline Matrix odometryInitialization(const std::vector<RelativeSEMeasurement>& odometry) {
  detail::EdgeSoa s(odometry);
  Matrix T(s.d, (size_t)(s.d + 1) * s.n);
  check(dpgo_odometry_initialization((int)s.d, (int)s.n, (int)odometry.size(), s.p1.data(), s.p2.data(), s.R.data(),
                                     s.t.data(), T.data()));
  return T;
}

// solvePGO (src/DPGO_solver.cpp:305-333): T0 = nullptr -> chordal initialisation, as the reference
inline Matrix solvePGO(const std::vector<RelativeSEMeasurement>& measurements, const ROptParameters& params,
                       const Matrix* T0 = nullptr, int device = 0) {
  if (measurements.empty()) throw Error(DPGO_ERR_INVALID, "no measurements");
  Matrix Tinit;
  if (!T0) {
    Tinit = chordalInitialization(measurements, device);
    T0 = &Tinit;
  }
  const unsigned d = (unsigned)measurements[0].R.rows();
  auto pg = std::make_shared<PoseGraph>(measurements[0].r1, d, d);  // robot id of the data, rank r = d (src/DPGO_solver.cpp:322-324)
  pg->setMeasurements(measurements);
  QuadraticProblem problem(pg, device);
  QuadraticOptimizer optimizer(&problem, params);
  return optimizer.optimize(*T0);
}

// GNC with truncated least squares.  One device problem serves all outer iterations: the weights are updated
// and Q's values / the preconditioner rebuilt ON THE DEVICE (the reference rebuilds a PoseGraph per iteration).
inline Matrix solveRobustPGO(std::vector<RelativeSEMeasurement>& mutable_measurements,
                             const solveRobustPGOParams& params, const Matrix* T0 = nullptr, int device = 0) {
  if (mutable_measurements.empty()) throw Error(DPGO_ERR_INVALID, "no measurements");
  Matrix Tinit;
  if (!T0) {  // src/DPGO_solver.cpp:341: chordal initialisation on the full measurement set
    Tinit = chordalInitialization(mutable_measurements, device);
    T0 = &Tinit;
  }
  if (params.robust_params.costType != RobustCostParameters::Type::GNC_TLS)
    throw Error(DPGO_ERR_INVALID, "CHECK(params.robust_params.costType == GNC_TLS) failed");  // :355
  const double w_tol = 1e-8;  // :340
  const unsigned d = (unsigned)mutable_measurements[0].R.rows();
  auto pg = std::make_shared<PoseGraph>(mutable_measurements[0].r1, d, d);
  pg->setMeasurements(mutable_measurements);
  QuadraticProblem problem(pg, device);
  QuadraticOptimizer optimizer(&problem, params.opt_params);
  const auto& ms = pg->measurements();
  const int m = (int)ms.size();
  std::vector<int32_t> p1(m), p2(m);
  std::vector<double> R((size_t)m * d * d), t((size_t)m * d), kappa(m), tau(m), w(m, 1.0);
  std::vector<uint8_t> fixed(m);
  for (int e = 0; e < m; ++e) {
    p1[e] = (int32_t)ms[e].p1;
    p2[e] = (int32_t)ms[e].p2;
    for (unsigned a = 0; a < d; ++a) {
      for (unsigned b = 0; b < d; ++b) R[((size_t)e * d + a) * d + b] = ms[e].R(a, b);
      t[(size_t)e * d + a] = ms[e].t(a, 0);
    }
    kappa[e] = ms[e].kappa;
    tau[e] = ms[e].tau;
    w[e] = ms[e].weight;
    fixed[e] = ms[e].fixedWeight ? 1 : 0;
  }
  Matrix T = optimizer.optimize(*T0);  // :342 initial estimate (current weights)
  check(dpgo_problem_set_reweightable_edges(problem.handle(), m, p1.data(), p2.data(), R.data(), t.data(), kappa.data(),
                                            tau.data(), w.data(), fixed.data()));
  std::fill(w.begin(), w.end(), 1.0);  // :346 meas.weight = 1
  check(dpgo_problem_set_edge_weights(problem.handle(), w.data()));
  int counts[3] = {0, 0, 0};
  double max_rsq = 0.0;
  check(dpgo_problem_gnc_reweight(problem.handle(), T.data(), 1.0, params.robust_params.GNCBarc, w_tol, 0, counts,
                                  &max_rsq));  // residuals only (:347-351)
  const double barcSq = params.robust_params.GNCBarc * params.robust_params.GNCBarc;
  const double muInit = barcSq / (2 * max_rsq - barcSq);  // :358
  if (muInit > 0) {  // negative: small residuals, GNC is skipped (:367)
    RobustCostParameters pg_params = params.robust_params;
    pg_params.GNCInitMu = muInit;
    RobustCost cost(pg_params);
    for (unsigned iter = 0; iter < pg_params.GNCMaxNumIters; ++iter) {
      T = optimizer.optimize(*T0);  // always restarted from T0 (:372)
      check(dpgo_problem_gnc_reweight(problem.handle(), T.data(), cost.mu(), pg_params.GNCBarc, w_tol, 1, counts,
                                      nullptr));
      if (counts[2] == 0) break;  // no undecided weight (:403)
      cost.update();
    }
  }
  T = optimizer.optimize(*T0);  // :409
  check(dpgo_problem_get_edge_weights(problem.handle(), w.data(), nullptr));
  for (auto& mm : mutable_measurements)  // duplicates (dropped by PoseGraph) keep their weight
    for (int e = 0; e < m; ++e)
      if (mm.r1 == ms[e].r1 && mm.p1 == ms[e].p1 && mm.r2 == ms[e].r2 && mm.p2 == ms[e].p2) {
        mm.weight = w[e];
        break;
      }
  return T;
}

}  // namespace dpgo_hip
