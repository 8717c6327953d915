"""Initial guesses, through the C ABI -- no SciPy, no CPU solver in the product path.

  chordal_initialization    reference src/DPGO_solver.cpp:220-269 (+ constructBMatrices / recoverTranslations,
                            src/DPGO_utils.cpp:346-462)  ->  dpgo_chordal_initialization: the two least-squares problems
                            are solved on the device (Jacobi-preconditioned CG over the block-SpMM of the hot path,
                            SO(d) projection by the rounding kernel)
  odometry_initialization   reference src/DPGO_solver.cpp:271-303  ->  dpgo_odometry_initialization (a sequential
                            composition along the chain: host code inside the library)

Results are tiles T[n, d+1, d] (tile i = [R_i^T ; t_i], i.e. the d x (d+1) pose in the reference's column-major layout).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as L
from .measurements import RelativeSEMeasurements


def _arrays(meas: RelativeSEMeasurements):
    return (L.i32(meas.p1), L.i32(meas.p2), L.f64(meas.R), L.f64(meas.t), L.f64(meas.kappa), L.f64(meas.tau))


def chordal_initialization(meas: RelativeSEMeasurements, num_poses: int, tol: float = 0.0, max_iter: int = 0,
                           device: int = 0, return_iterations: bool = False):
    """chordalInitialization for the poses of ONE robot (frames 0 .. n-1); pose 0 is the identity."""
    d, n = meas.d, int(num_poses)
    p1, p2, R, t, kappa, tau = _arrays(meas)
    T = np.zeros((n, d + 1, d))
    its = (C.c_int * 2)()
    L.check(L.load().dpgo_chordal_initialization(d, n, len(meas), L.ptr(p1), L.ptr(p2), L.ptr(R), L.ptr(t), L.ptr(kappa),
                                                 L.ptr(tau), float(tol), int(max_iter), L.ptr(T), its, int(device)))
    return (T, (its[0], its[1])) if return_iterations else T


def odometry_initialization(odometry: RelativeSEMeasurements, num_poses: int) -> np.ndarray:
    """odometryInitialization: poses composed along the edges p -> p + 1; pose 0 is the identity."""
    d, n = odometry.d, int(num_poses)
    p1, p2, R, t, _, _ = _arrays(odometry)
    T = np.zeros((n, d + 1, d))
    L.check(L.load().dpgo_odometry_initialization(d, n, len(odometry), L.ptr(p1), L.ptr(p2), L.ptr(R), L.ptr(t), L.ptr(T)))
    return T
