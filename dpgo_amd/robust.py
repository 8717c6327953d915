"""Robust cost functions and the GNC outer loop around the device local solver.

Mirrors (names, defaults, semantics) of the reference:
  RobustCostParameters / RobustCost   include/DPGO/DPGO_robust.h:20-133, src/DPGO_robust.cpp:49-134
  solvePGO / solveRobustPGO           src/DPGO_solver.cpp:305-412, include/DPGO/DPGO_solver.h:100-123

The reference's solveRobustPGO rebuilds a PoseGraph (Q, preconditioner) for every GNC outer iteration;
here Q's block pattern is fixed and only its VALUES are rebuilt on the device from the edge weights
(dpgo_problem_gnc_reweight_device: residual kernel K10 + value rebuild K9), X never leaves HBM.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import lib as L
from .measurements import RelativeSEMeasurements
from .solver import PoseGraph, QuadraticOptimizer, QuadraticProblem, ROptParameters


@dataclass
class RobustCostParameters:
    """include/DPGO/DPGO_robust.h:20-57"""
    costType: str = "L2"  # "L2" | "L1" | "TLS" | "Huber" | "GM" | "GNC_TLS"
    GNCMaxNumIters: int = 20
    GNCBarc: float = 5.0
    GNCMuStep: float = 1.4
    GNCInitMu: float = 1e-4
    HuberThreshold: float = 3.0
    TLSThreshold: float = 10.0


def _gamma_p(a: float, x: float) -> float:
    """Regularised lower incomplete gamma function P(a, x): series below a + 1, Lentz's continued fraction above."""
    import math
    if x <= 0:
        return 0.0
    lg = math.lgamma(a)
    if x < a + 1:
        ap, total, delta = a, 1.0 / a, 1.0 / a
        for _ in range(1000):
            ap += 1
            delta *= x / ap
            total += delta
            if abs(delta) < abs(total) * 1e-17:
                break
        return total * math.exp(-x + a * math.log(x) - lg)
    tiny = 1e-300
    b = x + 1 - a
    c, dd = 1 / tiny, 1 / b
    h = dd
    for k in range(1, 1000):
        an = -k * (k - a)
        b += 2
        dd = an * dd + b
        dd = tiny if abs(dd) < tiny else dd
        c = b + an / c
        c = tiny if abs(c) < tiny else c
        dd = 1 / dd
        delta = dd * c
        h *= delta
        if abs(delta - 1) < 1e-16:
            break
    return 1.0 - math.exp(-x + a * math.log(x) - lg) * h


def chi2inv(quantile: float, dof: int) -> float:
    """chi2inv (include/DPGO/DPGO_utils.h:146-153, src/DPGO_utils.cpp:509-512: boost's chi-squared quantile,
    "equivalent to chi2inv in Matlab"): x with P(dof / 2, x / 2) = quantile."""
    if not (0.0 <= quantile < 1.0) or dof <= 0:
        raise ValueError("chi2inv: quantile in [0, 1), dof > 0")
    if quantile == 0.0:
        return 0.0
    a = 0.5 * dof
    lo, hi = 0.0, max(1.0, float(dof))
    while _gamma_p(a, 0.5 * hi) < quantile:
        hi *= 2
    for _ in range(200):
        if hi - lo <= 1e-15 * hi:
            break
        mid = 0.5 * (lo + hi)
        if _gamma_p(a, 0.5 * mid) < quantile:
            lo = mid
        else:
            hi = mid
    return 0.5 * (lo + hi)


class RobustCost:
    """src/DPGO_robust.cpp:49-134"""

    @staticmethod
    def computeErrorThresholdAtQuantile(quantile: float, dimension: int) -> float:
        """include/DPGO/DPGO_robust.h:116-123: the GNC threshold barc for a 3-D measurement whose squared error is
        chi-squared with 6 degrees of freedom."""
        if dimension != 3:
            raise ValueError("CHECK_EQ(dimension, 3) failed: quantile function currently only supports 3D problem.")
        if not quantile > 0:
            raise ValueError("CHECK_GT(quantile, 0) failed")
        return chi2inv(quantile, 6) ** 0.5 if quantile < 1 else 1e5

    def __init__(self, params: RobustCostParameters):
        self.mParams = params
        self.mu = params.GNCInitMu
        self.mGNCIteration = 0
        self.reset()

    def weight(self, r: float) -> float:  # :54-98
        p = self.mParams
        t = p.costType
        if t == "L2":
            return 1.0
        if t == "L1":
            return 1.0 / r
        if t == "Huber":
            return 1.0 if r < p.HuberThreshold else p.HuberThreshold / r
        if t == "TLS":
            return 1.0 if r < p.TLSThreshold else 0.0
        if t == "GM":
            a = 1 + r * r
            return 1.0 / (a * a)
        if t == "GNC_TLS":  # eq. (14) of the GNC paper
            rSq, bSq, mu = r * r, p.GNCBarc * p.GNCBarc, self.mu
            upper, lower = (mu + 1) / mu * bSq, mu / (mu + 1) * bSq
            if rSq >= upper:
                return 0.0
            if rSq <= lower:
                return 1.0
            return math.sqrt(bSq * mu * (mu + 1) / rSq) - mu
        raise RuntimeError("weight function for selected cost function is not implemented !")  # :95

    def reset(self) -> None:  # :100-114
        if self.mParams.costType == "GNC_TLS":
            self.mu = self.mParams.GNCInitMu
            self.mGNCIteration = 0

    def update(self) -> None:  # :116-134
        if self.mParams.costType != "GNC_TLS":
            return
        self.mGNCIteration += 1
        if self.mGNCIteration > self.mParams.GNCMaxNumIters:
            return
        self.mu = self.mParams.GNCMuStep * self.mu


@dataclass
class solveRobustPGOParams:
    """include/DPGO/DPGO_solver.h:114-123"""
    opt_params: ROptParameters = field(default_factory=ROptParameters)
    robust_params: RobustCostParameters = field(default_factory=lambda: RobustCostParameters("GNC_TLS"))
    verbose: bool = False


class _DeviceSolve:
    """One problem handle reused across GNC iterations (pattern fixed, values re-weighted on device)."""

    def __init__(self, measurements: RelativeSEMeasurements, num_poses: int, device: int = 0):
        import torch
        self.torch = torch
        d = measurements.d
        self.d, self.n = d, num_poses
        robot_id = int(measurements.r1[0])
        self.pg = PoseGraph(robot_id, d, d)  # solvePGO: rank r = d (src/DPGO_solver.cpp:322)
        self.pg.setMeasurements(measurements)
        if self.pg.n() != num_poses:
            raise ValueError("measurements span %d poses, expected %d" % (self.pg.n(), num_poses))
        self.problem = QuadraticProblem(self.pg, device=device)
        self.problem.setStream(torch.cuda.current_stream().cuda_stream)
        self.m = self.pg.measurements()
        self.kept = self.pg.kept_index  # positions of the kept edges in the caller's array
        self.problem.setReweightableEdges()
        self.device = torch.device("cuda", device)

    def set_weights(self, w: np.ndarray) -> None:
        self.problem.setEdgeWeights(w)

    def solve(self, T0_tiles: np.ndarray, params: ROptParameters):
        X = self.torch.tensor(np.ascontiguousarray(T0_tiles), dtype=self.torch.float64, device=self.device)
        opt = QuadraticOptimizer(self.problem, params)
        opt.optimizeDevice(X)
        return X, opt.getOptResult()

    def reweight(self, X, mu: float, barc: float, w_tol: float, update: bool):
        return self.problem.gncReweightDevice(X, None, mu, barc, w_tol, update)

    def weights(self):
        return self.problem.getEdgeWeights()


def solvePGO(measurements: RelativeSEMeasurements, num_poses: int, params: Optional[ROptParameters] = None,
             T0: Optional[np.ndarray] = None, device: int = 0) -> np.ndarray:
    """solvePGO (src/DPGO_solver.cpp:305-333): rank-d problem, chordal initialisation unless T0 is given.
    Tiles in / out: T[n, d+1, d]."""
    from .initialization import chordal_initialization
    T0 = chordal_initialization(measurements, num_poses) if T0 is None else T0
    ds = _DeviceSolve(measurements, num_poses, device)
    X, _ = ds.solve(T0, params or ROptParameters())
    return X.cpu().numpy()


def solveRobustPGO(mutable_measurements: RelativeSEMeasurements, num_poses: int,
                   params: Optional[solveRobustPGOParams] = None, T0: Optional[np.ndarray] = None, device: int = 0):
    """solveRobustPGO (src/DPGO_solver.cpp:335-412), GNC with truncated least squares.  The weights of
    `mutable_measurements` are updated in place.  Returns (T tiles [n, d+1, d], info dict)."""
    from .initialization import chordal_initialization
    params = params or solveRobustPGOParams()
    if params.robust_params.costType != "GNC_TLS":
        raise ValueError("CHECK(params.robust_params.costType == GNC_TLS) failed")  # :355
    w_tol = 1e-8  # :340
    meas = mutable_measurements
    T0 = chordal_initialization(meas, num_poses) if T0 is None else T0
    ds = _DeviceSolve(meas, num_poses, device)
    X, _ = ds.solve(T0, params.opt_params)  # :342 initial estimate
    ds.set_weights(np.ones(len(ds.m)))  # :346 meas.weight = 1
    _, max_rsq = ds.reweight(X, 1.0, params.robust_params.GNCBarc, w_tol, update=False)  # residuals only (:347-351)
    barcSq = params.robust_params.GNCBarc ** 2
    muInit = barcSq / (2 * max_rsq - barcSq)  # :358
    info = dict(muInit=muInit, gnc_iterations=0, history=[])
    if muInit > 0:  # negative: small residuals, skip GNC (:367)
        cost = RobustCost(RobustCostParameters("GNC_TLS", params.robust_params.GNCMaxNumIters,
                                               params.robust_params.GNCBarc, params.robust_params.GNCMuStep, muInit))
        for it in range(params.robust_params.GNCMaxNumIters):
            X, res = ds.solve(T0, params.opt_params)  # always restarted from T0 (:372)
            (n_in, n_out, n_und), _ = ds.reweight(X, cost.mu, params.robust_params.GNCBarc, w_tol, update=True)
            info["history"].append(dict(mu=cost.mu, inliers=n_in, outliers=n_out, undecided=n_und, f=res.fOpt))
            info["gnc_iterations"] = it + 1
            if n_und == 0:  # :403
                break
            cost.update()
    X, res = ds.solve(T0, params.opt_params)  # :409
    w, _ = ds.weights()
    meas.weight[ds.kept] = w
    info["fOpt"] = res.fOpt
    return X.cpu().numpy(), info


class DistributedGNC:
    """Synchronous distributed GNC-TLS over an RBCDCluster (BASELINE configs[4]).

    Assembled from the reference's per-agent pieces -- the in-tree library never drives them itself, the
    external dpgo_ros node does:
      * PGOAgent::updateMeasurementWeights (src/PGOAgent.cpp:1104-1142): every agent re-weights ALL its
        non-fixed loop closures, private and shared, from its iterate and its neighbours' public poses
        (computeMeasurementResidual, :1048-1102); RobustCost::update (mu <- GNCMuStep * mu); the data
        matrices are refreshed (clearDataMatrices -> here: Q / coupling / preconditioner VALUES rebuilt on
        the device, pattern fixed); warm start (robustOptNumResets = 0);
      * initial mu as solveRobustPGO (src/DPGO_solver.cpp:358) from the largest residual after the first
        block of sweeps; stop when no weight is undecided (tolerance 1e-8, :340, :403).
    Both endpoints of a shared edge evaluate the same residual on the same poses and hence agree on its
    weight without a message (the reference ships weights from the owner instead; same values).
    Global scalars (max residual, the three counters) are one tiny all-reduce each per weight update."""

    def __init__(self, cluster, robust_params: Optional[RobustCostParameters] = None, inner_sweeps: int = 5,
                 agent_params=None):
        """agent_params (a dpgo_amd.agent.PGOAgentParameters): the weight updates follow the reference's own trigger,
        PGOAgent::shouldUpdateMeasurementWeights (src/PGOAgent.cpp:997-1045) -- global iterations (= colour phases)
        until every agent is readyToTerminate (relative change of its last update <= relChangeTol, 5 before the first
        weight update; converged-weight ratio >= robustOptMinConvergenceRatio) or robustOptInnerIters of them have
        passed -- instead of a fixed number of sweeps; info["inner_iterations"] lists the count of every block."""
        self.cluster = cluster
        self.params = robust_params or RobustCostParameters("GNC_TLS", GNCMaxNumIters=30)
        if self.params.costType != "GNC_TLS":
            raise ValueError("CHECK(costType == GNC_TLS) failed")
        self.inner_sweeps = int(inner_sweeps)
        self.agent_params = None
        self.iteration = 0
        self.weight_updates = 0
        self.inner_counts = []
        for agent in cluster.agents.values():
            agent.problem.setReweightableEdges(include_shared=True)
        if agent_params is not None:
            from dataclasses import replace
            self.agent_params = replace(agent_params, robust=True)
            for agent in cluster.agents.values():
                agent.enable_status(self.agent_params)
                agent.lc_weights = agent.loop_closure_weights()

    def _allreduce(self, values, op: str):
        c = self.cluster
        if c.world == 1:
            return np.asarray(values, dtype=np.float64)
        import torch
        dev = getattr(next(iter(c.agents.values())), "device", "cpu")
        if getattr(c, "comm", None) is not None:  # library-owned RCCL communicator
            t = torch.tensor(values, dtype=torch.float64, device=dev)
            c.comm.allreduce(t, 1 if op == "max" else 0, torch.cuda.current_stream().cuda_stream)
            return t.cpu().numpy()
        import torch.distributed as dist
        t = torch.tensor(values, dtype=torch.float64, device="cpu" if c.stage else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def _reweight_all(self, mu: float, update: bool):
        self.cluster.exchange(None)  # everyone's public poses are current
        counts, mx = np.zeros(3), 0.0
        for agent in self.cluster.agents.values():
            c, m = agent.problem.gncReweightDevice(agent.X, agent.nbr if agent.has_neighbours else None, mu,
                                                   self.params.GNCBarc, 1e-8, update)
            counts += c
            mx = max(mx, m)
        counts = self._allreduce(counts, "sum")
        mx = float(self._allreduce([mx], "max")[0])
        if update and self.agent_params is not None:  # PGOAgent::updateMeasurementWeights (:1119-1125)
            for agent in self.cluster.agents.values():
                agent.lc_weights = agent.loop_closure_weights()
                agent.status.readyToTerminate = False
                agent.status.relativeChange = 0.0
        return tuple(int(v) for v in counts), mx

    def _sweeps(self) -> None:
        if self.agent_params is None:
            for _ in range(self.inner_sweeps):
                self.cluster.sweep()
            return
        from .agent import should_update_measurement_weights
        c, prm = self.cluster, self.agent_params
        for agent in c.agents.values():
            agent.weight_update_count = self.weight_updates
            agent.status.iterationNumber = 0  # mTeamStatus.clear() after a weight update (:1124)
        inner, latest = 0, self.iteration
        while True:
            colour = self.iteration % c.plan.num_colours
            self.iteration += 1
            inner += 1
            c.phase(colour, self.iteration)
            team = c.team_status()
            # (the cap on the NUMBER of updates is run()'s loop bound, so the count passed here is 0)
            if should_update_measurement_weights(prm, 0, inner, latest, team, c.plan.num_agents):
                break
        self.inner_counts.append(inner)

    def run(self):
        """Returns info = {muInit, updates, history[], cost, gradnorm}; the iterates stay on the agents."""
        p = self.params
        for agent in self.cluster.agents.values():
            agent.problem.setEdgeWeights(np.ones(len(agent.problem.reweightable_index)))
        self._sweeps()
        _, max_rsq = self._reweight_all(1.0, update=False)
        barcSq = p.GNCBarc ** 2
        muInit = barcSq / (2 * max_rsq - barcSq)
        info = dict(muInit=muInit, updates=0, history=[])
        if muInit > 0:
            mu = muInit
            for it in range(p.GNCMaxNumIters):
                (n_in, n_out, n_und), _ = self._reweight_all(mu, update=True)
                info["history"].append(dict(mu=mu, inliers=n_in, outliers=n_out, undecided=n_und))
                info["updates"] = it + 1
                self.weight_updates = it + 1
                if n_und == 0:
                    break
                mu = p.GNCMuStep * mu
                self._sweeps()
        self._sweeps()
        info["inner_iterations"] = list(self.inner_counts)
        f, gn = self.cluster.central_cost_and_gradnorm()
        info["cost"], info["gradnorm"] = 2 * f, gn
        return info

    def weights(self):
        """{agent id: (positions in that agent's pose_graph.measurements(), weights)} of the local agents."""
        return {a: (ag.problem.reweightable_index, ag.problem.getEdgeWeights()[0])
                for a, ag in self.cluster.agents.items()}
