"""Host-side initial guesses (setup, outside the hot path).

  chordalInitialization    reference src/DPGO_solver.cpp:220-269 (+ constructBMatrices /
                           recoverTranslations, src/DPGO_utils.cpp:346-462)
  odometryInitialization   reference src/DPGO_solver.cpp:271-303

The reference solves the two sparse least-squares problems with SPQR; here the same minimisers are
obtained from the sparse normal equations (SciPy SuperLU) -- full column rank once pose 0 is pinned.
Results are returned as tiles T[n, d+1, d] (tile i = [R_i^T ; t_i], i.e. the d x (d+1) pose in the
reference's column-major layout).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from .measurements import RelativeSEMeasurements


def project_to_rotation_group(M: np.ndarray) -> np.ndarray:
    """projectToRotationGroup (src/DPGO_utils.cpp:464-478)."""
    U, _, Vt = np.linalg.svd(M)
    neg = np.linalg.det(U) * np.linalg.det(Vt) < 0
    if M.ndim == 2:
        if neg:
            U = U.copy()
            U[:, -1] *= -1
        return U @ Vt
    U = U.copy()
    U[neg, :, -1] *= -1
    return U @ Vt


def _solve_ls(A: sp.spmatrix, rhs: np.ndarray) -> np.ndarray:
    A = A.tocsc()
    return spla.splu((A.T @ A).tocsc()).solve(A.T @ rhs)


def chordal_initialization(meas: RelativeSEMeasurements, num_poses: int) -> np.ndarray:
    d, m, n = meas.d, len(meas), num_poses
    d2 = d * d
    e = np.arange(m)
    p1, p2 = meas.p1.astype(np.int64), meas.p2.astype(np.int64)
    sk, st = np.sqrt(meas.kappa), np.sqrt(meas.tau)
    # rotations: minimise sum kappa |R_j - R_i R_ij|^2 with R_0 = I        (B3, :417-433)
    rows, cols, vals = [], [], []
    for r_ in range(d):
        for c in range(d):
            for l in range(d):
                rows.append(e * d2 + d * r_ + l); cols.append(p1 * d2 + d * c + l); vals.append(-sk * meas.R[:, c, r_])
    for l in range(d2):
        rows.append(e * d2 + l); cols.append(p2 * d2 + l); vals.append(sk)
    B3 = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(m * d2, n * d2))
    cR = B3[:, :d2] @ np.eye(d).reshape(-1)
    rvec = -_solve_ls(B3[:, d2:], cR)
    R = np.empty((n, d, d))
    R[0] = np.eye(d)
    R[1:] = project_to_rotation_group(np.swapaxes(rvec.reshape(n - 1, d, d), 1, 2))  # column-major blocks
    # translations: minimise sum tau |t_j - t_i - R_i t_ij|^2 with t_0 = 0   (B1 :367-389, B2 :394-407)
    rows, cols, vals = [], [], []
    for l in range(d):
        rows += [e * d + l, e * d + l]; cols += [p1 * d + l, p2 * d + l]; vals += [-st, st]
    B1 = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(m * d, n * d))
    rows, cols, vals = [], [], []
    for k in range(d):
        for r_ in range(d):
            rows.append(e * d + r_); cols.append(p1 * d2 + d * k + r_); vals.append(-st * meas.t[:, k])
    B2 = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(m * d, n * d2))
    c = B2 @ np.swapaxes(R, 1, 2).reshape(-1)
    t = np.zeros((n, d))
    t[1:] = -_solve_ls(B1[:, d:], c).reshape(n - 1, d)
    T = np.zeros((n, d + 1, d))
    T[:, :d, :] = np.swapaxes(R, 1, 2)
    T[:, d, :] = t
    return T


def odometry_initialization(odometry: RelativeSEMeasurements, num_poses: int) -> np.ndarray:
    d = odometry.d
    T = np.zeros((num_poses, d + 1, d))
    T[0, :d, :] = np.eye(d)
    by_src = {int(odometry.p1[k]): k for k in range(len(odometry)) if odometry.p1[k] + 1 == odometry.p2[k]}
    for dst in range(1, num_poses):
        k = by_src[dst - 1]  # reference: CHECK(m.p1 == src), CHECK(m.p2 == dst)
        Rs, ts = T[dst - 1, :d, :].T, T[dst - 1, d, :]
        T[dst, :d, :] = (Rs @ odometry.R[k]).T
        T[dst, d, :] = ts + Rs @ odometry.t[k]
    return T
