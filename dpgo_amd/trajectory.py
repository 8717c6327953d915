"""Rounding of a lifted iterate to SE(d) trajectories and the reference's CSV trajectory / measurement files.

Mirrors of the reference:
  getTrajectoryInLocalFrame / getTrajectoryInGlobalFrame   src/PGOAgent.cpp:718-767   (device kernel K12)
  PGOLogger::logTrajectory / loadTrajectory / logMeasurements   src/PGOLogger.cpp:18-155
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np

from . import lib as L
from .measurements import RelativeSEMeasurements


def round_trajectory(X, r: int, d: int, anchor=None, device: int = 0) -> np.ndarray:
    """X: r x (d+1)n (host numpy, column-major semantics) -> T: d x (d+1)n.
    anchor = None: local frame of pose 0 (getTrajectoryInLocalFrame); anchor = r x (d+1) lifted pose:
    global frame of that anchor (setGlobalAnchor + getTrajectoryInGlobalFrame)."""
    X = np.asfortranarray(X, dtype=np.float64)
    if X.ndim != 2 or X.shape[0] != r or X.shape[1] % (d + 1) != 0:
        raise ValueError("X has shape %s, expected (%d, (d+1) n)" % (X.shape, r))
    n = X.shape[1] // (d + 1)
    T = np.empty((d, (d + 1) * n), order="F")
    a = None
    if anchor is not None:
        a = np.asfortranarray(anchor, dtype=np.float64)
        if a.shape != (r, d + 1):
            raise ValueError("CHECK(M.rows() == relaxation_rank() && M.cols() == dimension() + 1) failed")
    L.check(L.load().dpgo_round_trajectory(r, d, n, L.ptr(X), L.ptr(a) if a is not None else None, L.ptr(T), device))
    return T


def round_trajectory_device(X_dev, anchor=None):
    """Device flavour: X_dev = torch tensor of pose tiles [n, d+1, r] on the GPU; returns a tensor of
    rounded tiles [n, d+1, d] (the d x (d+1)n matrix of the reference, column-major) on the same device."""
    import torch
    n, b, r = X_dev.shape
    d = b - 1
    T = torch.empty((n, b, d), dtype=torch.float64, device=X_dev.device)
    a = None
    if anchor is not None:
        a = np.asfortranarray(anchor, dtype=np.float64)
        if a.shape != (r, d + 1):
            raise ValueError("CHECK(M.rows() == relaxation_rank() && M.cols() == dimension() + 1) failed")
    L.check(L.load().dpgo_round_trajectory_device(r, d, n, L.ptr(X_dev), L.ptr(a) if a is not None else None,
                                                  L.ptr(T), torch.cuda.current_stream().cuda_stream or None))
    return T


# ---------------------------------------------------------------- quaternions (Eigen::Quaternion conventions)
def _rot_to_quat(R: np.ndarray):
    """Eigen::Quaterniond(Matrix3d) (Shepperd's branches as in Eigen's quaternionbase_assign_impl)."""
    t = R[0, 0] + R[1, 1] + R[2, 2]
    if t > 0:
        s = math.sqrt(t + 1.0)
        w = 0.5 * s
        s = 0.5 / s
        return ((R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s, w)
    i = 0
    if R[1, 1] > R[0, 0]:
        i = 1
    if R[2, 2] > R[i, i]:
        i = 2
    j, k = (i + 1) % 3, (i + 2) % 3
    s = math.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
    q = [0.0, 0.0, 0.0]
    q[i] = 0.5 * s
    s = 0.5 / s
    w = (R[k, j] - R[j, k]) * s
    q[j] = (R[j, i] + R[i, j]) * s
    q[k] = (R[k, i] + R[i, k]) * s
    return (q[0], q[1], q[2], w)


def _quat_to_rot(x, y, z, w) -> np.ndarray:
    nrm = math.sqrt(x * x + y * y + z * z + w * w)  # loadTrajectory normalises (src/PGOLogger.cpp:121)
    x, y, z, w = x / nrm, y / nrm, z / nrm, w / nrm
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def log_trajectory(d: int, n: int, T: np.ndarray, filename: str) -> bool:
    """PGOLogger::logTrajectory (src/PGOLogger.cpp:56-84): 3-D only (returns False for d == 2, as the
    reference silently does); header 'pose_index,qx,qy,qz,qw,tx,ty,tz'."""
    if d == 2:
        return False
    T = np.asarray(T)
    if T.shape != (d, (d + 1) * n):
        raise ValueError("CHECK_EQ(T.rows(), d) / CHECK_EQ(T.cols(), (d + 1) * n) failed")
    with open(filename, "w") as fh:
        fh.write("pose_index,qx,qy,qz,qw,tx,ty,tz\n")
        for i in range(n):
            R = T[:, i * (d + 1):i * (d + 1) + d]
            t = T[:, i * (d + 1) + d]
            qx, qy, qz, qw = _rot_to_quat(R)
            fh.write("%d,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g\n" % (i, qx, qy, qz, qw, t[0], t[1], t[2]))
    return True


def load_trajectory(filename: str) -> np.ndarray:
    """PGOLogger::loadTrajectory (src/PGOLogger.cpp:86-155): 3 x 4n matrix, poses ordered by pose_index."""
    poses = {}
    with open(filename) as fh:
        next(fh)
        for line in fh:
            tok = line.strip().split(",")
            if len(tok) < 8:
                continue
            pid = int(tok[0])
            qx, qy, qz, qw, tx, ty, tz = (float(v) for v in tok[1:8])
            Ti = np.zeros((3, 4))
            Ti[:, :3] = _quat_to_rot(qx, qy, qz, qw)
            Ti[:, 3] = (tx, ty, tz)
            poses[pid] = Ti
    n = len(poses)
    T = np.zeros((3, 4 * n), order="F")
    for i in range(n):
        T[:, 4 * i:4 * i + 4] = poses[i]
    return T


def log_measurements(meas: RelativeSEMeasurements, filename: str) -> bool:
    """PGOLogger::logMeasurements (src/PGOLogger.cpp:18-54): 3-D only; header
    'robot_src,pose_src,robot_dst,pose_dst,qx,qy,qz,qw,tx,ty,tz,kappa,tau,is_known_inlier,weight'."""
    if len(meas) == 0 or meas.d == 2:
        return False
    with open(filename, "w") as fh:
        fh.write("robot_src,pose_src,robot_dst,pose_dst,qx,qy,qz,qw,tx,ty,tz,kappa,tau,is_known_inlier,weight\n")
        for e in range(len(meas)):
            qx, qy, qz, qw = _rot_to_quat(meas.R[e])
            t = meas.t[e]
            fh.write("%d,%d,%d,%d,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%d,%.17g\n" % (
                meas.r1[e], meas.p1[e], meas.r2[e], meas.p2[e], qx, qy, qz, qw, t[0], t[1], t[2], meas.kappa[e],
                meas.tau[e], int(meas.fixedWeight[e]), meas.weight[e]))
    return True
