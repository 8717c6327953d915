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


class RobustCost:
    """src/DPGO_robust.cpp:49-134"""

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
        fixed = np.ascontiguousarray(self.m.fixedWeight, dtype=np.uint8)
        L.check(self.problem._lib.dpgo_problem_set_reweightable_edges(
            self.problem.handle, len(self.m), L.ptr(self.m.p1), L.ptr(self.m.p2), L.ptr(self.m.R), L.ptr(self.m.t),
            L.ptr(self.m.kappa), L.ptr(self.m.tau), L.ptr(self.m.weight), L.ptr(fixed)))
        self.device = torch.device("cuda", device)

    def set_weights(self, w: np.ndarray) -> None:
        w = np.ascontiguousarray(w, dtype=np.float64)
        L.check(self.problem._lib.dpgo_problem_set_edge_weights(self.problem.handle, L.ptr(w)))

    def solve(self, T0_tiles: np.ndarray, params: ROptParameters):
        X = self.torch.tensor(np.ascontiguousarray(T0_tiles), dtype=self.torch.float64, device=self.device)
        opt = QuadraticOptimizer(self.problem, params)
        opt.optimizeDevice(X)
        return X, opt.getOptResult()

    def reweight(self, X, mu: float, barc: float, w_tol: float, update: bool):
        counts = (C.c_int * 3)()
        mx = C.c_double(0.0)
        L.check(self.problem._lib.dpgo_problem_gnc_reweight_device(
            self.problem.handle, L.ptr(X), float(mu), float(barc), float(w_tol), int(update), C.byref(counts),
            C.byref(mx)))
        return tuple(counts), mx.value

    def weights(self):
        w = np.zeros(len(self.m))
        rs = np.zeros(len(self.m))
        L.check(self.problem._lib.dpgo_problem_get_edge_weights(self.problem.handle, L.ptr(w), L.ptr(rs)))
        return w, rs


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
