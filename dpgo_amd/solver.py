"""Host-side mirror of the reference's local-solver interface, backed by libdpgo_hip.so.

Same names, argument meaning and error behaviour as the reference classes so parity tests
read like the reference's own tests:

  PoseGraph (data-matrix part)   include/DPGO/PoseGraph.h:59-69,106-194 ; src/PoseGraph.cpp
  QuadraticProblem               include/DPGO/QuadraticProblem.h:33-115
  QuadraticOptimizer             include/DPGO/QuadraticOptimizer.h:20-104
  LiftedSEManifold               include/DPGO/manifold/LiftedSEManifold.h:28-43
  ROptParameters / ROPTResult    include/DPGO/DPGO_types.h:44-107

"Matrix" arguments are numpy arrays of shape (r, (d+1)*n) -- the reference's
Eigen::MatrixXd; they are passed to the device in column-major order (Fortran order), i.e.
n consecutive pose tiles.  There is no CPU fallback anywhere in this module.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np

from . import lib as L
from .measurements import RelativeSEMeasurements


# --------------------------------------------------------------------------- parameters
@dataclass
class ROptParameters:
    """DPGO::ROptParameters (include/DPGO/DPGO_types.h:44-86) + device-path extensions."""
    method: str = "RTR"  # ROptMethod::RTR | "RGD"
    verbose: bool = False
    gradnorm_tol: float = 1e-2
    RGD_stepsize: float = 1e-3
    RGD_use_preconditioner: bool = True
    RTR_iterations: int = 3
    RTR_tCG_iterations: int = 50
    RTR_initial_radius: float = 100.0
    # extensions
    # "auto" (default: a multilevel preconditioner when the tCG budget binds, block-Jacobi while it does not -- decided
    # per problem handle, ROPTResult.precond_used tells) | "multilevel" (aggregation-multigrid V-cycle for Q + shift I, the
    # stand-in for the reference's exact solve) | "additive" (block-Jacobi + coarse-grid correction, runs inside the
    # persistent kernel; blocks of <= 4 096 poses in 3-D) | "jacobi" (block-Jacobi) | "none"
    precond: str = "auto"
    precond_shift: float = 1e-1  # src/PoseGraph.cpp:603
    accept_tiny_decrease: bool = True
    tcg_poll_interval: int = 0  # 0 = just-in-time feed (default); k > 0 = poll every k tCG iterations
    time_bound_s: float = 5.0  # Solver.TimeBound, src/QuadraticOptimizer.cpp:78

    def to_c(self) -> L.RoptParamsC:
        c = L.RoptParamsC()
        c.method = {"RTR": L.METHOD_RTR, "RGD": L.METHOD_RGD}[self.method]
        c.verbose = int(self.verbose)
        c.gradnorm_tol = self.gradnorm_tol
        c.RGD_stepsize = self.RGD_stepsize
        c.RGD_use_preconditioner = int(self.RGD_use_preconditioner)
        c.RTR_iterations = self.RTR_iterations
        c.RTR_tCG_iterations = self.RTR_tCG_iterations
        c.RTR_initial_radius = self.RTR_initial_radius
        c.precond = {"jacobi": L.PRECOND_BLOCK_JACOBI, "none": L.PRECOND_NONE, "multilevel": L.PRECOND_MULTILEVEL,
                     "auto": L.PRECOND_AUTO, "additive": L.PRECOND_ADDITIVE}[self.precond]
        c.precond_shift = self.precond_shift
        c.accept_tiny_decrease = int(self.accept_tiny_decrease)
        c.tcg_poll_interval = self.tcg_poll_interval
        c.time_bound_s = self.time_bound_s
        return c


@dataclass
class ROPTResult:
    """DPGO::ROPTResult (include/DPGO/DPGO_types.h:91-107) + counters."""
    success: bool = False
    fInit: float = 0.0
    gradNormInit: float = 0.0
    fOpt: float = 0.0
    gradNormOpt: float = 0.0
    elapsedMs: float = 0.0
    tCGStatus: str = "MAXITER"
    rtr_iterations: int = 0
    rtr_accepted: int = 0
    tcg_iterations: int = 0
    spmm_count: int = 0
    latest_step_accepted: bool = False
    precond_used: str = ""

    @staticmethod
    def from_c(c: L.RoptResultC) -> "ROPTResult":
        return ROPTResult(bool(c.success), c.fInit, c.gradNormInit, c.fOpt, c.gradNormOpt, c.elapsedMs,
                          L.TCG_STATUS[c.tCGStatus], c.rtr_iterations, c.rtr_accepted, c.tcg_iterations,
                          c.spmm_count, bool(c.latest_step_accepted), L.PRECOND_NAMES[c.precond_used])


def _colmajor(X, r: int, N: int, what: str = "Matrix") -> np.ndarray:
    X = np.asarray(X, dtype=np.float64)
    if X.shape != (r, N):  # reference: CHECK_EQ on rows / cols (src/QuadraticProblem.cpp:30-31)
        raise ValueError("%s has shape %s, expected (%d, %d)" % (what, X.shape, r, N))
    return np.asfortranarray(X)


# --------------------------------------------------------------------------- PoseGraph
class PoseGraph:
    """Data-matrix part of DPGO::PoseGraph: owns the measurements, neighbour poses and priors of
    one agent and lazily builds Q (block-CSR) and G, with the reference's invalidation rules:
    setNeighborPoses resets G only (src/PoseGraph.cpp:183-186); clearQuadraticMatrix also drops
    the preconditioner (:352-355); setMeasurements empties everything (:61-66)."""

    def __init__(self, id: int, r: int, d: int):
        if r < d:
            raise ValueError("CHECK(r >= d) failed")  # src/PoseGraph.cpp:19
        self.id_, self.r_, self.d_, self.n_ = int(id), int(r), int(d), 0
        self.prior_kappa_, self.prior_tau_ = 10000.0, 100.0  # :17-18
        self._meas: Optional[RelativeSEMeasurements] = None
        self.neighbor_poses_: Dict[Tuple[int, int], np.ndarray] = {}
        self.priors_: Dict[int, np.ndarray] = {}
        self.neighbor_active_: Dict[int, bool] = {}  # neighbour robot -> active (src/PoseGraph.cpp:192-207)
        self.use_inactive_neighbors_ = False          # :632
        self._Q = None
        self._G = None
        self._coupling = None
        self.q_version = 0

    def id(self): return self.id_
    def r(self): return self.r_
    def d(self): return self.d_
    def n(self): return self.n_

    def setMeasurements(self, measurements: RelativeSEMeasurements) -> None:
        """PoseGraph::setMeasurements (:61-66).  Irrelevant edges are dropped with a warning in
        the reference (:68-71); duplicate (src, dst) edges are silently dropped (:83-88)."""
        m = measurements
        if m.d != self.d_:
            raise ValueError("measurement dimension %d != %d" % (m.d, self.d_))
        keep = (m.r1 == self.id_) | (m.r2 == self.id_)
        m = m.select(keep)
        seen, first = set(), []
        for e in range(len(m)):
            key = (int(m.r1[e]), int(m.p1[e]), int(m.r2[e]), int(m.p2[e]))
            if key not in seen:
                seen.add(key)
                first.append(e)
        kept_in_selected = np.array(first, dtype=np.int64)
        self.kept_index = np.nonzero(keep)[0][kept_in_selected]  # positions in the caller's array
        m = m.select(kept_in_selected)
        self._meas = m
        mine1 = m.r1 == self.id_
        mine2 = m.r2 == self.id_
        n = 0
        if mine1.any():
            n = max(n, int(m.p1[mine1].max()) + 1)
        if mine2.any():
            n = max(n, int(m.p2[mine2].max()) + 1)
        self.n_ = n
        self.neighbor_poses_ = {}
        self.priors_ = {}
        self.neighbor_active_ = {int(q): True for q in set(m.r1.tolist()) | set(m.r2.tolist()) if q != self.id_}
        self.clearDataMatrices()

    # ---- neighbour activity (src/PoseGraph.cpp:188-207, 252-303, 632-634) ----
    def hasNeighbor(self, robot_id: int) -> bool:
        return int(robot_id) in self.neighbor_active_

    def isNeighborActive(self, neighbor_id: int) -> bool:
        return self.neighbor_active_.get(int(neighbor_id), False)

    def setNeighborActive(self, neighbor_id: int, active: bool) -> None:
        """PoseGraph::setNeighborActive: the shared edges with an inactive neighbour leave Q and G (constructQ /
        constructG skip them, :418-430, :520-532) -- the data matrices are dropped when the flag changes."""
        if not self.hasNeighbor(neighbor_id):
            return
        if self.neighbor_active_[int(neighbor_id)] != bool(active):
            self.clearDataMatrices()
        self.neighbor_active_[int(neighbor_id)] = bool(active)

    def useInactiveNeighbors(self, use: bool = True) -> None:
        """PoseGraph::useInactiveNeighbors (:632-634): keep an inactive neighbour's edges whose pose is still known."""
        self.use_inactive_neighbors_ = bool(use)
        self.clearDataMatrices()

    def activeNeighborIDs(self):
        return sorted(q for q, act in self.neighbor_active_.items() if act)

    def activeNeighborPublicPoseIDs(self):
        return [pid for pid in self.neighborPoseIDs() if self.isNeighborActive(pid[0])]

    def _effective_weights(self) -> np.ndarray:
        """The measurement weights the data matrices are built with: an edge with an inactive neighbour contributes
        nothing (unless use_inactive_neighbors_ and its pose is known) -- carried as weight 0, which adds exact zeros
        where the reference skips the edge (Omega = w diag(kappa.., tau))."""
        m = self._meas
        w = m.weight
        if all(self.neighbor_active_.values()):
            return w
        w = w.copy()
        for e in range(len(m)):
            if m.r1[e] == m.r2[e]:
                continue
            nb = (int(m.r2[e]), int(m.p2[e])) if m.r1[e] == self.id_ else (int(m.r1[e]), int(m.p1[e]))
            if not self.isNeighborActive(nb[0]) and not (self.use_inactive_neighbors_ and nb in self.neighbor_poses_):
                w[e] = 0.0
        return w

    def odometry_mask(self, m: Optional[RelativeSEMeasurements] = None) -> np.ndarray:
        """Which of the measurements (default: all of this graph's) are odometry -- consecutive poses of this robot in
        the CALLER's numbering (PoseGraph::addOdometry keeps exactly those apart, src/PoseGraph.cpp:61-77).  A graph
        whose poses the agent layer renumbered for locality (build_pose_graphs(reorder=True): pose_order = internal row
        of every caller frame) tests the caller's ids, so that a loop closure landing on consecutive internal rows
        stays a loop closure."""
        m = self._meas if m is None else m
        same = m.r1 == m.r2
        order = getattr(self, "pose_order", None)
        if order is None:
            return same & (m.p1 + 1 == m.p2)
        caller = np.empty(len(order), dtype=np.int64)
        caller[np.asarray(order, dtype=np.int64)] = np.arange(len(order))
        p1 = np.where(same, caller[np.minimum(m.p1, len(order) - 1)], 0)
        p2 = np.where(same, caller[np.minimum(m.p2, len(order) - 1)], 0)
        return same & (p1 + 1 == p2)

    def measurements(self) -> RelativeSEMeasurements:
        return self._meas

    def sharedLoopClosures(self) -> RelativeSEMeasurements:
        m = self._meas
        return m.select(m.r1 != m.r2)

    def neighborPoseIDs(self):
        """Sorted (robot, frame) ids of the neighbour poses this agent needs (nbr_shared_pose_ids_)."""
        m = self.sharedLoopClosures()
        ids = set()
        for e in range(len(m)):
            if m.r1[e] == self.id_:
                ids.add((int(m.r2[e]), int(m.p2[e])))
            else:
                ids.add((int(m.r1[e]), int(m.p1[e])))
        return sorted(ids)

    def localSharedPoseIDs(self):
        m = self.sharedLoopClosures()
        ids = set()
        for e in range(len(m)):
            ids.add(int(m.p1[e]) if m.r1[e] == self.id_ else int(m.p2[e]))
        return sorted(ids)

    def setPrior(self, index: int, Xi) -> None:  # :176-181
        if not (0 <= index < self.n_):
            raise ValueError("CHECK_LT(index, n()) failed")
        Xi = np.asarray(Xi, dtype=np.float64)
        if Xi.shape != (self.r_, self.d_ + 1):
            raise ValueError("prior has shape %s, expected (%d, %d)" % (Xi.shape, self.r_, self.d_ + 1))
        self.priors_[int(index)] = Xi.copy()
        self.clearDataMatrices()

    def setNeighborPoses(self, pose_dict: Dict[Tuple[int, int], np.ndarray]) -> None:  # :183-186
        self.neighbor_poses_ = dict(pose_dict)
        self._G = None

    def clearDataMatrices(self) -> None:  # :376-379
        self._Q = None
        self._G = None
        self._coupling = None
        self.q_version += 1

    def _edge_arrays(self):
        m = self._meas
        self._w_eff = np.ascontiguousarray(self._effective_weights(), dtype=np.float64)  # (kept alive for the C call)
        return (len(m), L.ptr(m.r1), L.ptr(m.p1), L.ptr(m.r2), L.ptr(m.p2), L.ptr(m.R), L.ptr(m.t),
                L.ptr(m.kappa), L.ptr(m.tau), L.ptr(self._w_eff))

    def quadraticMatrix(self):
        """PoseGraph::quadraticMatrix (:345-350) -> (rowptr, colidx, vals[nnzb, b, b]) block-CSR,
        built by dpgo_build_Q_bsr (constructQ, :381-491).  Shared edges with an inactive neighbour
        (setNeighborActive) are left out (:425-430); the reference's missing-pose check for ACTIVE
        neighbours (:418-424) is enforced where the poses are actually consumed, in linearMatrix()."""
        if self._Q is None:
            if self._meas is None or self.n_ == 0:
                raise RuntimeError("PoseGraph has no measurements")
            lib = L.load()
            b = self.d_ + 1
            pidx = L.i32(sorted(self.priors_.keys()))
            nnzb = C.c_int(0)
            m, *arrs = self._edge_arrays()
            L.check(lib.dpgo_build_Q_bsr(self.id_, self.d_, self.n_, m, *arrs, len(pidx), L.ptr(pidx),
                                         self.prior_kappa_, self.prior_tau_, C.byref(nnzb), None, None, None))
            rowptr = np.zeros(self.n_ + 1, dtype=np.int32)
            colidx = np.zeros(nnzb.value, dtype=np.int32)
            vals = np.zeros((nnzb.value, b, b))
            L.check(lib.dpgo_build_Q_bsr(self.id_, self.d_, self.n_, m, *arrs, len(pidx), L.ptr(pidx),
                                         self.prior_kappa_, self.prior_tau_, C.byref(nnzb), L.ptr(rowptr),
                                         L.ptr(colidx), L.ptr(vals)))
            self._Q = (rowptr, colidx, vals)
        return self._Q

    def couplingMatrix(self):
        """Operator form of constructG (:493-580): returns (slots, rowptr, colidx, vals, G0) with
        G = G0 + Xnbr * C; slots = sorted neighbour pose ids = column order of the tile buffer."""
        if self._coupling is None:
            lib = L.load()
            m_all = self._meas
            slots = self.neighborPoseIDs()
            slot_index = {pid: k for k, pid in enumerate(slots)}
            slot_of_edge = np.full(len(m_all), -1, dtype=np.int32)
            for e in range(len(m_all)):
                if m_all.r1[e] != m_all.r2[e]:
                    pid = (int(m_all.r2[e]), int(m_all.p2[e])) if m_all.r1[e] == self.id_ else \
                          (int(m_all.r1[e]), int(m_all.p1[e]))
                    slot_of_edge[e] = slot_index[pid]
            b = self.d_ + 1
            nnzb = C.c_int(0)
            m, *arrs = self._edge_arrays()
            L.check(lib.dpgo_build_G_coupling(self.id_, self.d_, self.n_, m, *arrs, L.ptr(slot_of_edge),
                                              C.byref(nnzb), None, None, None))
            rowptr = np.zeros(self.n_ + 1, dtype=np.int32)
            colidx = np.zeros(max(nnzb.value, 1), dtype=np.int32)[:nnzb.value]
            vals = np.zeros((nnzb.value, b, b))
            L.check(lib.dpgo_build_G_coupling(self.id_, self.d_, self.n_, m, *arrs, L.ptr(slot_of_edge),
                                              C.byref(nnzb), L.ptr(rowptr), L.ptr(colidx) if nnzb.value else None,
                                              L.ptr(vals) if nnzb.value else None))
            G0 = np.zeros((self.r_, b * self.n_), order="F")
            om = np.array([self.prior_kappa_] * self.d_ + [self.prior_tau_])
            for idx, P in self.priors_.items():  # :565-574  L = -P Omega
                G0[:, idx * b:(idx + 1) * b] += -(P * om[None, :])
            self._coupling = (slots, rowptr, colidx, vals, G0)
        return self._coupling

    def linearMatrix(self) -> np.ndarray:
        """PoseGraph::linearMatrix (:359-364): dense r x (d+1)n.  Host evaluation of
        G = G0 + Xnbr * C (used by the host-pointer flavour; the device flavour is
        QuadraticProblem.updateLinearMatrixFromNeighbors)."""
        if self._G is None:
            slots, rowptr, colidx, vals, G0 = self.couplingMatrix()
            b, r = self.d_ + 1, self.r_
            G = G0.copy(order="F")
            for i in range(self.n_):
                for t in range(rowptr[i], rowptr[i + 1]):
                    pid = slots[colidx[t]]
                    if pid not in self.neighbor_poses_:
                        if not self.isNeighborActive(pid[0]):
                            continue  # (its coupling blocks are zero: the edge is out of the problem, :525-530)
                        raise LookupError("Missing active neighbor pose %s" % (pid,))
                    Xn = np.asarray(self.neighbor_poses_[pid], dtype=np.float64)  # r x b
                    G[:, i * b:(i + 1) * b] += Xn @ vals[t].T  # out[a,c] = sum_k X[a,k] blk[c][k]
            self._G = G
        return self._G


# --------------------------------------------------------------------------- QuadraticProblem
class QuadraticProblem:
    """DPGO::QuadraticProblem: f(X) = 0.5 <Q, X^T X> + <X, G> on (St(d,r) x R^r)^n.

    Owns the device handle (Q in block-CSR, G, block-Jacobi factors, work vectors).  The
    reference builds a new problem per iteration (src/PGOAgent.cpp:968); keep this object alive
    across iterations instead and call refresh() after the pose graph changed."""

    def __init__(self, pose_graph: PoseGraph, device: int = 0, host_linear_term: bool = True):
        """host_linear_term = False: G is not taken from PoseGraph::linearMatrix() on the host but
        built on the device from the neighbour tile buffer (setCouplingFromPoseGraph +
        updateLinearMatrixFromNeighbors)."""
        self.pose_graph_ = pose_graph
        self._host_G = bool(host_linear_term)
        self._lib = L.load()
        self._h = L._P()
        r, d, n = pose_graph.r(), pose_graph.d(), pose_graph.n()
        if n <= 0:
            raise RuntimeError("pose graph must be initialised (n > 0)")
        L.check(self._lib.dpgo_problem_create(C.byref(self._h), r, d, n, device))
        self._q_version = -1
        self._g_obj = None
        self.refresh()

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                self._lib.dpgo_problem_destroy(self._h)
                self._h = L._P()
        except Exception:
            pass

    def num_poses(self) -> int: return self.pose_graph_.n()
    def dimension(self) -> int: return self.pose_graph_.d()
    def relaxation_rank(self) -> int: return self.pose_graph_.r()

    @property
    def handle(self):
        return self._h

    def _N(self):
        return (self.dimension() + 1) * self.num_poses()

    def refresh(self) -> None:
        """Re-upload Q and/or G if the pose graph invalidated them (same rules as the reference's
        lazy getters, src/PoseGraph.cpp:345-366)."""
        pg = self.pose_graph_
        if self._q_version != pg.q_version:
            rowptr, colidx, vals = pg.quadraticMatrix()
            L.check(self._lib.dpgo_problem_set_Q_bsr(self._h, len(colidx), L.ptr(rowptr), L.ptr(colidx), L.ptr(vals)))
            self._q_version = pg.q_version
            self._g_obj = None
            # a new Q drops the device's re-weightable edge lists (they index the old values): register them again
            if getattr(self, "reweightable_index", None) is not None and not getattr(self, "_in_reregister", False):
                self._in_reregister = True
                try:
                    if self._reweight_shared:
                        self.setCouplingFromPoseGraph()
                    self.setReweightableEdges(self._reweight_shared)
                finally:
                    self._in_reregister = False
        has_shared = self._host_G and (len(pg.sharedLoopClosures()) > 0 or len(pg.priors_) > 0)
        if not self._host_G:
            return
        if not has_shared:
            if self._g_obj is not False:
                L.check(self._lib.dpgo_problem_set_G(self._h, None))
                self._g_obj = False
        else:
            G = pg.linearMatrix()
            if self._g_obj is not G:
                L.check(self._lib.dpgo_problem_set_G(self._h, L.ptr(G)))
                self._g_obj = G

    def _in(self, X, what="Y"):
        return _colmajor(X, self.relaxation_rank(), self._N(), what)

    def _out(self):
        return np.empty((self.relaxation_rank(), self._N()), order="F")

    def f(self, Y) -> float:  # src/QuadraticProblem.cpp:29-35
        Yc = self._in(Y)
        out = C.c_double(0.0)
        L.check(self._lib.dpgo_problem_f(self._h, L.ptr(Yc), C.byref(out)))
        return out.value

    def EucGrad(self, X) -> np.ndarray:  # :43-47
        Xc, o = self._in(X), self._out()
        L.check(self._lib.dpgo_problem_euc_grad(self._h, L.ptr(Xc), L.ptr(o)))
        return o

    def EucHessianEta(self, X, V) -> np.ndarray:  # :49-54  (x unused: the cost is quadratic)
        Vc, o = self._in(V, "V"), self._out()
        L.check(self._lib.dpgo_problem_euc_hess(self._h, L.ptr(Vc), L.ptr(o)))
        return o

    def RieHessianEta(self, X, V) -> np.ndarray:
        """ROPTLIB Problem::HessianEta: EucHessianEta + Stiefel::EucHvToHv + tangent projection."""
        Xc, Vc, o = self._in(X), self._in(V, "V"), self._out()
        L.check(self._lib.dpgo_problem_rie_hess(self._h, L.ptr(Xc), L.ptr(Vc), L.ptr(o)))
        return o

    def PreConditioner(self, X, inVec, precond: str = "multilevel", shift: float = 1e-1) -> np.ndarray:  # :56-69
        Xc, Vc, o = self._in(X), self._in(inVec, "inVec"), self._out()
        pc = {"jacobi": L.PRECOND_BLOCK_JACOBI, "none": L.PRECOND_NONE, "multilevel": L.PRECOND_MULTILEVEL,
              "auto": L.PRECOND_AUTO}[precond]
        L.check(self._lib.dpgo_problem_precondition(self._h, pc, shift, L.ptr(Xc), L.ptr(Vc), L.ptr(o)))
        return o

    def RieGrad(self, Y) -> np.ndarray:  # :71-79
        Yc, o = self._in(Y), self._out()
        L.check(self._lib.dpgo_problem_rie_grad(self._h, L.ptr(Yc), L.ptr(o)))
        return o

    def RieGradNorm(self, Y) -> float:  # :81-83
        Yc = self._in(Y)
        out = C.c_double(0.0)
        L.check(self._lib.dpgo_problem_rie_grad_norm(self._h, L.ptr(Yc), C.byref(out)))
        return out.value

    # ---- device-resident flavour (torch tensors or raw device addresses) ----
    def setStream(self, hip_stream: Optional[int]) -> None:
        """Order this problem's device work with an external stream (torch's current stream);
        0 / None = the default (null) stream.  useOwnStream() reverts to the private stream."""
        L.check(self._lib.dpgo_problem_set_stream(self._h, hip_stream or None))

    def useOwnStream(self) -> None:
        L.check(self._lib.dpgo_problem_use_own_stream(self._h))

    def setCouplingFromPoseGraph(self) -> list:
        """Upload the G operator once; returns the neighbour slot order (list of (robot, frame))."""
        slots, rowptr, colidx, vals, G0 = self.pose_graph_.couplingMatrix()
        L.check(self._lib.dpgo_problem_set_G_coupling(self._h, len(slots), len(colidx), L.ptr(rowptr),
                                                      L.ptr(colidx) if len(colidx) else None,
                                                      L.ptr(vals) if len(colidx) else None, L.ptr(G0)))
        return slots

    def updateLinearMatrixFromNeighbors(self, nbr_tiles_dev) -> None:
        """constructG on the device from the neighbour tile buffer (src/PoseGraph.cpp:493-580)."""
        L.check(self._lib.dpgo_problem_update_G_from_neighbors_device(self._h, L.ptr(nbr_tiles_dev)))
        self._g_obj = None

    # ---- persistent whole-chip tCG kernel (blocks in the latency regime, block-Jacobi / no preconditioner) ----
    def setPersistent(self, enable: bool = True) -> None:
        L.check(self._lib.dpgo_problem_set_persistent(self._h, int(enable)))

    def tcgKernelInfo(self) -> dict:
        """{"symmetric", "split", "stream_nt"}: the instance of the tCG-step kernel the next multi-launch solve runs."""
        v = [C.c_int(0) for _ in range(3)]
        L.check(self._lib.dpgo_problem_tcg_kernel_info(self._h, *[C.byref(x) for x in v]))
        return dict(symmetric=bool(v[0].value), split=v[1].value, stream_nt=bool(v[2].value))

    def persistentInfo(self) -> dict:
        """{"enabled", "workgroups", "last_members", "last_iterations", "last_split", "last_tiles"}: last_members = 0
        means the last optimize call ran the multi-launch scheme; last_split = lane groups per pose, last_tiles = pose
        tiles per workgroup of the last persistent launch."""
        v = [C.c_int(0) for _ in range(5)]
        L.check(self._lib.dpgo_problem_persistent_info(self._h, *[C.byref(x) for x in v]))
        out = dict(zip(("enabled", "workgroups", "last_members", "last_iterations"), (x.value for x in v[:4])))
        out["last_split"], out["last_tiles"] = v[4].value // 16, v[4].value % 16
        return out

    def persistentPhases(self) -> dict:
        """In-kernel phase split of the last one-launch solve, us per tCG iteration on participant 0
        (dpgo_problem_persistent_phases): hessian, reduce_after_hessian, update, reduce_after_update; iterations."""
        v = (C.c_double * 4)()
        it = C.c_int(0)
        L.check(self._lib.dpgo_problem_persistent_phases(self._h, v, C.byref(it)))
        return dict(hessian=v[0], reduce_after_hessian=v[1], update=v[2], reduce_after_update=v[3], iterations=it.value)

    # ---- multilevel preconditioner (built on the device; lazily by the first solve, like
    # PoseGraph::constructPreconditioner inside the first PreConditioner call, src/PoseGraph.cpp:582-586) ----
    def multilevelPath(self) -> dict:
        """Which kernels a cycle of the current hierarchy runs: {"ap": bool, "packed_dense": bool} (see dpgo_hip.h)."""
        v = C.c_int(0)
        L.check(self._lib.dpgo_problem_multilevel_path(self._h, C.byref(v)))
        return dict(ap=bool(v.value & 1), packed_dense=bool(v.value & 2))

    def multilevelCoarseBits(self, bits=None) -> int:
        """Storage precision (32 or 64 bits) of the dense inverse of the coarsest operator; an argument sets it."""
        v = C.c_int(-1 if bits is None else int(bits))
        L.check(self._lib.dpgo_problem_multilevel_coarse_bits(self._h, C.byref(v)))
        return int(v.value)

    def multilevelOperatorBits(self, bits=None) -> dict:
        """Storage precision (32 or 64 bits) of the level-0 operator copies the V-cycle streams on HBM-bound blocks
        (dpgo_problem_multilevel_operator_bits); an argument sets it.  {"bits", "active", "vectors", "dense"}: what the
        last solve's cycle streamed in fp32 -- the operator copies, its two internal vectors, the dense level."""
        v, act = C.c_int(-1 if bits is None else int(bits)), C.c_int(0)
        L.check(self._lib.dpgo_problem_multilevel_operator_bits(self._h, C.byref(v), C.byref(act)))
        return dict(bits=int(v.value), active=bool(act.value & 1), vectors=bool(act.value & 2), dense=bool(act.value & 4))

    def setupMultilevel(self, ks=None, omega: float = 0.7, shift: float = 1e-1, coarse_bits=None) -> dict:
        """Explicit setup for the current Q: ks = aggregate sizes per coarsening (None: the library's defaults; a single
        negative entry -S = two levels with graph aggregates of at most S poses, [-S, -cap] = the same with the growth's
        fragments merged up to cap poses, see dpgo_hip.h),
        coarse_bits = storage precision of the dense level (None: keep the handle's, 64 by default; 32 = opt-in).
        Returns multilevelInfo()."""
        self.refresh()
        if coarse_bits is not None:
            self.multilevelCoarseBits(coarse_bits)
        ks = L.i32(ks if ks is not None else [])
        L.check(self._lib.dpgo_problem_setup_multilevel(self._h, len(ks), L.ptr(ks) if len(ks) else None,
                                                        float(omega), float(shift)))
        return self.multilevelInfo()

    def setSpmmVariant(self, variant: str = "auto") -> str:
        """Storage the plain Q*V products of this handle read: "plain", "symmetric" (upper blocks only, transposed; for
        Infinity-Cache-cold blocks) or "auto" (by size).  Returns what the next product will read."""
        self.refresh()
        names = {"auto": 0, "plain": 1, "symmetric": 2}
        if variant not in names:
            raise ValueError("variant must be one of %s" % sorted(names))
        v = C.c_int(0)
        L.check(self._lib.dpgo_problem_set_spmm_variant(self._h, names[variant], C.byref(v)))
        return "symmetric" if v.value == 2 else "plain"

    def describe(self) -> str:
        """What this handle currently runs (layout, storage, one-launch solve, preconditioner selection, hierarchy) and the
        library's switches (dpgo_problem_describe)."""
        buf = C.create_string_buffer(32768)
        L.check(self._lib.dpgo_problem_describe(self._h, buf, len(buf)))
        return buf.value.decode("utf-8", "replace")

    def autoState(self, use_multilevel=None) -> bool:
        """What precond = "auto" currently runs on this handle (True: the multilevel cycle); a bool argument sets it,
        "reset" returns to the decision a fresh handle takes for this problem."""
        v = C.c_int(-1 if use_multilevel is None else (-2 if use_multilevel == "reset" else int(bool(use_multilevel))))
        L.check(self._lib.dpgo_problem_auto_state(self._h, C.byref(v)))
        return bool(v.value)

    def autoInfo(self) -> dict:
        """Where the cost rule of precond = "auto" stands on this handle (dpgo_problem_auto_info): state ("jacobi" /
        "trial" / "additive"), jacobi_units counted since Q last changed (or the last hand-back), reference_products,
        switches, backoff, the unit costs the rule last charged on this handle (units_jacobi / units_additive: scaled by the
        part of the chip a launch blocks when the handle is solved next to others) and the rule's constants
        (dpgo_auto_rule_constants)."""
        st, ref, sw, bo, uj, ua = (C.c_int(0) for _ in range(6))
        units = C.c_longlong(0)
        L.check(self._lib.dpgo_problem_auto_info(self._h, C.byref(st), C.byref(units), C.byref(ref), C.byref(sw), C.byref(bo),
                                                 C.byref(uj), C.byref(ua)))
        k = [C.c_int(0) for _ in range(4)]
        L.check(self._lib.dpgo_auto_rule_constants(*[C.byref(x) for x in k]))
        return dict(state=("jacobi", "trial", "additive")[st.value], jacobi_units=int(units.value),
                    reference_products=ref.value, switches=sw.value, backoff=bo.value,
                    units_jacobi=uj.value, units_additive=ua.value, units_jacobi_alone=k[0].value,
                    units_additive_alone=k[1].value, setup_units=k[2].value, min_products=k[3].value)

    def multilevelInfo(self) -> dict:
        """{"sizes": nodes per level, "ks": aggregate size per coarsening (negative: graph aggregates of at most that many
        poses), "nnzb": blocks of A_l per level}."""
        cap = 16
        nl = C.c_int(cap)
        sizes, ks, nnzb = (np.zeros(cap, dtype=np.int32) for _ in range(3))
        L.check(self._lib.dpgo_problem_multilevel_info(self._h, C.byref(nl), L.ptr(sizes), L.ptr(ks), L.ptr(nnzb)))
        n = nl.value
        # (graph aggregates with merged fragments: ks = [-S, -cap], the form setupMultilevel takes)
        kk = [int(v) for v in ks[:n - 1]] + ([int(ks[n - 1])] if n >= 2 and ks[n - 1] < 0 else [])
        return dict(sizes=[int(v) for v in sizes[:n]], ks=kk, nnzb=[int(v) for v in nnzb[:n]])

    def additivePlan(self) -> dict:
        """Layout precond = "additive" uses for this block (dpgo_problem_additive_plan; host only): lane_groups per pose of
        the one-launch kernel (0: the block does not fit), tile = slots per aggregate = workgroup, growth / merge_cap of
        the graph aggregates (ks = [-growth, -merge_cap], or [-growth] when merge_cap is 0), aggregates = workgroups."""
        v = [C.c_int(0) for _ in range(6)]
        L.check(self._lib.dpgo_problem_additive_plan(self._h, *[C.byref(x) for x in v]))
        keys = ("lane_groups", "tile", "growth", "merge_cap", "aggregates", "graph")
        out = {k: int(x.value) for k, x in zip(keys, v)}
        out["ks"] = ([-out["growth"]] + ([-out["merge_cap"]] if out["merge_cap"] else [])) if out["graph"] else [out["tile"]]
        return out

    def multilevelGet(self, level: int, what: str) -> np.ndarray:
        """Copy of one item of the built hierarchy: "P" (prolongation blocks of a level), "rowptr" / "colidx" / "A"
        (Galerkin operator of a level >= 1), "inverse" (dense inverse of the last level), "labels" (graph aggregates:
        the aggregate of every pose), "ap_nnzb" (two levels: blocks of A P, a 1-element array), "restrict_partials" (graph
        aggregates: partial sums one restriction writes, a 1-element array)."""
        info = self.multilevelInfo()
        b = self.dimension() + 1
        n_l, nz = info["sizes"][level], info["nnzb"][level]
        code, shape, dtype = {"P": (L.ML_P_BLOCKS, (n_l, b, b), np.float64),
                              "rowptr": (L.ML_A_ROWPTR, (n_l + 1,), np.int32),
                              "colidx": (L.ML_A_COLIDX, (nz,), np.int32),
                              "A": (L.ML_A_VALUES, (nz, b, b), np.float64),
                              "labels": (L.ML_AGG_LABELS, (n_l,), np.int32),
                              "ap_nnzb": (L.ML_AP_NNZB, (1,), np.int32),
                              "restrict_partials": (L.ML_RESTRICT_PARTIALS, (1,), np.int32),
                              "inverse": (L.ML_DENSE_INVERSE, (n_l * b, n_l * b), np.float64)}[what]
        out = np.zeros(shape, dtype=dtype)  # (only the requested item: "inverse" of level 0 would be (n (d+1))^2)
        L.check(self._lib.dpgo_problem_multilevel_get(self._h, int(level), code, L.ptr(out)))
        return out

    # ---- GNC re-weighting on the device (PGOAgent::updateMeasurementWeights, src/PGOAgent.cpp:1104-1142) ----
    def setReweightableEdges(self, include_shared: bool = False) -> int:
        """Register the pose graph's edges as re-weightable: private edges always, shared loop closures
        too when include_shared (needs setCouplingFromPoseGraph first).  Edge order = measurements() order
        (optionally without the shared ones).  Returns the number of registered edges.

        On the agent path (include_shared) odometry edges are never re-weighted whatever their fixedWeight flag
        says: PGOAgent::updateMeasurementWeights only walks activeLoopClosures() (src/PGOAgent.cpp:1104-1142;
        PoseGraph::addOdometry keeps odometry apart, src/PoseGraph.cpp:73-77).  The single-agent solveRobustPGO
        follows the flags alone (src/DPGO_solver.cpp:376-380)."""
        self._reweight_shared = bool(include_shared)
        pg = self.pose_graph_
        m = pg.measurements()
        shared = m.r1 != m.r2
        sel = np.arange(len(m)) if include_shared else np.nonzero(~shared)[0]
        m = m.select(sel)
        role = np.zeros(len(m), dtype=np.uint8)
        slot = np.zeros(len(m), dtype=np.int32)
        if include_shared and (m.r1 != m.r2).any():
            slot_index = {pid: k for k, pid in enumerate(pg.couplingMatrix()[0])}
            for e in np.nonzero(m.r1 != m.r2)[0]:
                if m.r1[e] == pg.id():
                    role[e], slot[e] = 1, slot_index[(int(m.r2[e]), int(m.p2[e]))]
                else:
                    role[e], slot[e] = 2, slot_index[(int(m.r1[e]), int(m.p1[e]))]
        fixed = np.asarray(m.fixedWeight, dtype=bool)
        if include_shared:
            fixed = fixed | pg.odometry_mask(m)
        # what Q and the coupling blocks were built with: the shared edges with an INACTIVE neighbour are out (weight 0,
        # PoseGraph::constructQ / constructG skip them; PGOAgent::updateMeasurementWeights only walks
        # activeLoopClosures(), src/PGOAgent.cpp:1104-1118) -- registered with that weight and never re-weighted, so
        # that the device's base Q - sum(w contrib) and every later re-weighting leave them out as well
        w_eff = np.asarray(pg._effective_weights(), dtype=np.float64)[sel]
        fixed = fixed | (w_eff != np.asarray(m.weight, dtype=np.float64))
        fixed = np.ascontiguousarray(fixed, dtype=np.uint8)
        w = np.ascontiguousarray(w_eff, dtype=np.float64)
        L.check(self._lib.dpgo_problem_set_reweightable_edges_ex(
            self._h, len(m), L.ptr(m.p1), L.ptr(m.p2), L.ptr(role), L.ptr(slot), L.ptr(m.R), L.ptr(m.t),
            L.ptr(m.kappa), L.ptr(m.tau), L.ptr(w), L.ptr(fixed)))
        self.reweightable_index = sel  # positions in pose_graph.measurements()
        return len(m)

    def gncReweightDevice(self, X_dev, nbr_tiles_dev, mu: float, barc: float, w_tol: float = 1e-8,
                          update: bool = True):
        """Residuals (and, if update, GNC-TLS weights + rebuilt Q / coupling / preconditioner values) at a
        device iterate.  Returns ((inliers, outliers, undecided), max residual^2)."""
        counts = (C.c_int * 3)()
        mx = C.c_double(0.0)
        L.check(self._lib.dpgo_problem_gnc_reweight_device(
            self._h, L.ptr(X_dev), L.ptr(nbr_tiles_dev) if nbr_tiles_dev is not None else None, float(mu),
            float(barc), float(w_tol), int(update), C.byref(counts), C.byref(mx)))
        return tuple(counts), mx.value

    def pullEdgeWeights(self) -> None:
        """Write the device's current GNC weights of the registered edges back into the pose graph's measurements (the
        host copy is what refresh() rebuilds Q from after a neighbour (de)activation); edges that are out because their
        neighbour is inactive keep the host weight they will come back with."""
        idx = getattr(self, "reweightable_index", None)
        if idx is None or len(idx) == 0:
            return
        pg = self.pose_graph_
        w, _ = self.getEdgeWeights()
        m = pg.measurements()
        live = np.asarray(pg._effective_weights())[idx] == np.asarray(m.weight)[idx]
        m.weight[idx[live]] = np.asarray(w)[live]

    def setEdgeWeights(self, w: np.ndarray) -> None:
        w = np.ascontiguousarray(w, dtype=np.float64)
        L.check(self._lib.dpgo_problem_set_edge_weights(self._h, L.ptr(w)))

    def getEdgeWeights(self):
        """(weights, squared residuals of the last gncReweightDevice) of the registered edges."""
        m = len(self.reweightable_index)
        w, rs = np.zeros(m), np.zeros(m)
        L.check(self._lib.dpgo_problem_get_edge_weights(self._h, L.ptr(w), L.ptr(rs)))
        return w, rs

    def evalDevice(self, X_dev) -> Tuple[float, float]:
        f, g = C.c_double(0.0), C.c_double(0.0)
        L.check(self._lib.dpgo_problem_eval_device(self._h, L.ptr(X_dev), C.byref(f), C.byref(g)))
        return f.value, g.value

    def evalTermsDevice(self, X_dev) -> Tuple[float, float, float]:
        """(sum(XQ.X), sum(X.G), |rgrad|^2) at a device X; f = 0.5 xqx + xg."""
        a, b, c = C.c_double(0.0), C.c_double(0.0), C.c_double(0.0)
        L.check(self._lib.dpgo_problem_eval_terms_device(self._h, L.ptr(X_dev), C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def spmmDevice(self, V_dev, OUT_dev, add_G: bool = False) -> None:
        L.check(self._lib.dpgo_spmm_device(self._h, L.ptr(V_dev), L.ptr(OUT_dev), int(add_G)))

    def benchSpmm(self, reps: int, warmup: int = 5) -> float:
        ms = C.c_double(0.0)
        L.check(self._lib.dpgo_bench_spmm(self._h, reps, warmup, C.byref(ms)))
        return ms.value


# --------------------------------------------------------------------------- QuadraticOptimizer
class QuadraticOptimizer:
    """DPGO::QuadraticOptimizer (src/QuadraticOptimizer.cpp)."""

    def __init__(self, p: QuadraticProblem, params: Optional[ROptParameters] = None):
        self.problem_ = p
        self.params_ = params or ROptParameters()
        self.result_ = ROPTResult(success=False)  # :21-23

    def setProblem(self, p): self.problem_ = p
    def setVerbose(self, v): self.params_.verbose = bool(v)
    def setAlgorithm(self, alg): self.params_.method = alg
    def setRGDStepsize(self, s): self.params_.RGD_stepsize = float(s)
    def setRTRIterations(self, it): self.params_.RTR_iterations = int(it)
    def setGradientNormTolerance(self, tol): self.params_.gradnorm_tol = float(tol)
    def setRTRInitialRadius(self, radius): self.params_.RTR_initial_radius = float(radius)
    def setRTRtCGIterations(self, it): self.params_.RTR_tCG_iterations = int(it)
    def getOptResult(self) -> ROPTResult: return self.result_

    def optimize(self, Y) -> np.ndarray:
        """Matrix optimize(const Matrix&) (:26-48): host matrix in, host matrix out."""
        p = self.problem_
        Yc = p._in(Y)
        out = p._out()
        cp, cr = self.params_.to_c(), L.RoptResultC()
        L.check(p._lib.dpgo_optimize(p._h, C.byref(cp), L.ptr(Yc), L.ptr(out), C.byref(cr)))
        self.result_ = ROPTResult.from_c(cr)
        return out

    def optimizeDevice(self, X_dev) -> ROPTResult:
        """Device-resident flavour: X_dev (torch tensor / device address) is updated in place."""
        p = self.problem_
        cp, cr = self.params_.to_c(), L.RoptResultC()
        L.check(p._lib.dpgo_optimize_device(p._h, C.byref(cp), L.ptr(X_dev), C.byref(cr)))
        self.result_ = ROPTResult.from_c(cr)
        return self.result_


    def optimizeDeviceBegin(self, X_dev, nbr_tiles_dev=None) -> None:
        """First half of optimizeDevice (C ABI dpgo_optimize_device_begin): G from the neighbour tile buffer (optional), then
        the solve is ENQUEUED on the handle's stream when it is a one-launch solve (any other solve runs to completion
        here).  X_dev belongs to the solve until optimizeDeviceEnd()."""
        p = self.problem_
        cp = self.params_.to_c()
        L.check(p._lib.dpgo_optimize_device_begin(p._h, C.byref(cp), L.ptr(X_dev),
                                                  L.ptr(nbr_tiles_dev) if nbr_tiles_dev is not None else None))

    def optimizeDeviceEnd(self) -> ROPTResult:
        """Second half: waits for the handle's stream and returns the result of the solve optimizeDeviceBegin enqueued."""
        p = self.problem_
        cr = L.RoptResultC()
        L.check(p._lib.dpgo_optimize_device_end(p._h, C.byref(cr)))
        self.result_ = ROPTResult.from_c(cr)
        return self.result_


def bench_solve(optimizer, X0_dev, reps=20, warmup=3):
    """HIP-event time of one whole local solve from X0_dev (C ABI dpgo_bench_solve): dict(ms, products, persistent)."""
    p = optimizer.problem_
    cp = optimizer.params_.to_c()
    ms, prod, pers = C.c_double(0.0), C.c_double(0.0), C.c_int(0)
    L.check(p._lib.dpgo_bench_solve(p._h, C.byref(cp), L.ptr(X0_dev), reps, warmup, C.byref(ms), C.byref(prod), C.byref(pers)))
    return dict(ms=ms.value, products=prod.value, persistent=bool(pers.value))


def _ptr_array(ptrs):
    arr = (C.c_void_p * len(ptrs))()
    for k, v in enumerate(ptrs):
        arr[k] = v
    return arr


def _addr(x):
    return L.ptr(x)


def optimize_device_many(optimizers, X_devs, nbr_tiles=None, after_stream=None):
    """QuadraticOptimizer::optimize of several agents hosted by this process, CONCURRENTLY (C ABI
    dpgo_optimize_device_many): every problem runs on its own stream, ordered after `after_stream`; nbr_tiles[k]
    (or None) is agent k's neighbour tile buffer, from which G is rebuilt first (PGOAgent::updateX).  Optimizers that do
    not all carry the same parameters are solved one after the other, each with its own.  Returns the list of ROPTResult
    (also stored in each optimizer)."""
    if not optimizers:
        return []
    lib = optimizers[0].problem_._lib
    cp = optimizers[0].params_.to_c()
    n = len(optimizers)
    # the C entry point takes ONE parameter record for all handles: optimizers configured differently are solved one after
    # the other with their own parameters instead of silently inheriting the first one's
    first = bytes(cp)
    if any(bytes(o.params_.to_c()) != first for o in optimizers[1:]):
        out = []
        for k, (o, x) in enumerate(zip(optimizers, X_devs)):
            if nbr_tiles is not None and nbr_tiles[k] is not None:
                o.problem_.updateLinearMatrixFromNeighbors(nbr_tiles[k])
            out.append(o.optimizeDevice(x))
        return out
    handles = _ptr_array([o.problem_._h.value for o in optimizers])
    xs = _ptr_array([_addr(x) for x in X_devs])
    nb = _ptr_array([_addr(t) for t in nbr_tiles]) if nbr_tiles is not None else None
    res = (L.RoptResultC * n)()
    L.check(lib.dpgo_optimize_device_many(n, handles, C.byref(cp), xs, nb, after_stream or None, res))
    out = []
    for o, cr in zip(optimizers, res):
        o.result_ = ROPTResult.from_c(cr)
        if nbr_tiles is not None:
            o.problem_._g_obj = None
        out.append(o.result_)
    return out


def eval_terms_device_many(problems, X_devs, nbr_tiles=None, after_stream=None):
    """(sum(XQ.X), sum(X.G), |rgrad|^2) of several agents in one concurrent pass (dpgo_problem_eval_terms_device_many);
    returns an [n, 3] array."""
    n = len(problems)
    out = np.zeros((n, 3))
    if n == 0:
        return out
    lib = problems[0]._lib
    handles = _ptr_array([p._h.value for p in problems])
    xs = _ptr_array([_addr(x) for x in X_devs])
    nb = _ptr_array([_addr(t) for t in nbr_tiles]) if nbr_tiles is not None else None
    L.check(lib.dpgo_problem_eval_terms_device_many(n, handles, xs, nb, after_stream or None, L.ptr(out)))
    return out


# --------------------------------------------------------------------------- LiftedSEManifold
class LiftedSEManifold:
    """DPGO::LiftedSEManifold (include/DPGO/manifold/LiftedSEManifold.h:28-43) =
    (St(d,r) x R^r)^n with ROPTLIB's Stiefel parameter set 3 (Euclidean metric, qf retraction)."""

    def __init__(self, r: int, d: int, n: int, device: int = 0):
        self.r_, self.d_, self.n_, self.device = int(r), int(d), int(n), int(device)
        self._lib = L.load()

    def _N(self): return (self.d_ + 1) * self.n_

    def project(self, M) -> np.ndarray:  # src/manifold/LiftedSEManifold.cpp:34-45
        Mc = _colmajor(M, self.r_, self._N(), "M")
        o = np.empty_like(Mc, order="F")
        L.check(self._lib.dpgo_manifold_project(self.r_, self.d_, self.n_, L.ptr(Mc), L.ptr(o), self.device))
        return o

    def Projection(self, X, V) -> np.ndarray:  # ROPTLIB ProductManifold::Projection
        Xc, Vc = _colmajor(X, self.r_, self._N(), "X"), _colmajor(V, self.r_, self._N(), "V")
        o = np.empty_like(Xc, order="F")
        L.check(self._lib.dpgo_manifold_tangent_project(self.r_, self.d_, self.n_, L.ptr(Xc), L.ptr(Vc), L.ptr(o),
                                                        self.device))
        return o

    def Retraction(self, X, eta, scale: float = 1.0) -> np.ndarray:  # ROPTLIB ProductManifold::Retraction
        Xc, Ec = _colmajor(X, self.r_, self._N(), "X"), _colmajor(eta, self.r_, self._N(), "eta")
        o = np.empty_like(Xc, order="F")
        L.check(self._lib.dpgo_manifold_retract(self.r_, self.d_, self.n_, L.ptr(Xc), L.ptr(Ec), float(scale),
                                                L.ptr(o), self.device))
        return o
