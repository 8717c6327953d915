"""Synthetic 3-D grid pose graphs (BASELINE.json config 4: "synthetic 3D grid 100k poses").

Definition: nx x ny x nz lattice with unit spacing; a boustrophedon
("snake") odometry path with x fastest, then y, then z; a loop closure on every remaining
lattice-adjacent pair; ground-truth rotations uniform random; measurement noise: translation
N(0, sigma_t^2 I), rotation = axis-angle N(0, sigma_r^2 I); information matrices 1/sigma^2 I, i.e.
tau = 1/sigma_t^2 and kappa = 1/(2 sigma_r^2) by the g2o formulas (reference src/DPGO_utils.cpp:223,230).
RNG: numpy PCG64(seed).  50 x 50 x 40 gives 100 000 poses, 293 500 edges, nnzb = 687 000.
"""
from __future__ import annotations

import numpy as np

from .measurements import RelativeSEMeasurements


def _rotvec_to_R(v: np.ndarray) -> np.ndarray:
    th = np.linalg.norm(v, axis=1)
    k = v / np.maximum(th, 1e-300)[:, None]
    K = np.zeros((len(v), 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -k[:, 2], k[:, 1]
    K[:, 1, 0], K[:, 1, 2] = k[:, 2], -k[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -k[:, 1], k[:, 0]
    s, c = np.sin(th)[:, None, None], np.cos(th)[:, None, None]
    return np.eye(3)[None] + s * K + (1 - c) * (K @ K)


def _quat_to_R(q: np.ndarray) -> np.ndarray:
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((len(q), 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def snake_index(nx: int, ny: int, nz: int) -> np.ndarray:
    """idx[x, y, z] = position of lattice site (x, y, z) on the boustrophedon path."""
    idx = np.empty((nx, ny, nz), dtype=np.int64)
    k = 0
    for z in range(nz):
        ys = range(ny) if z % 2 == 0 else range(ny - 1, -1, -1)
        for yi, y in enumerate(ys):
            fwd = ((yi + z * ny) % 2 == 0)
            xs = np.arange(nx) if fwd else np.arange(nx - 1, -1, -1)
            idx[xs, y, z] = k + np.arange(nx)
            k += nx
    return idx


def synthetic_grid(nx: int, ny: int, nz: int, seed: int = 0, sigma_t: float = 0.1, sigma_r: float = 0.2):
    """Returns (measurements, num_poses, T_true) with T_true[n, 4, 3] = tiles [R^T ; t] (r = d)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = nx * ny * nz
    idx = snake_index(nx, ny, nz)
    pos = np.zeros((n, 3))
    gx, gy, gz = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    pos[idx.reshape(-1)] = np.stack([gx.reshape(-1), gy.reshape(-1), gz.reshape(-1)], axis=1)
    q = rng.standard_normal((n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    Rt = _quat_to_R(q)
    pairs = [np.stack([idx[:-1, :, :].reshape(-1), idx[1:, :, :].reshape(-1)], 1),
             np.stack([idx[:, :-1, :].reshape(-1), idx[:, 1:, :].reshape(-1)], 1),
             np.stack([idx[:, :, :-1].reshape(-1), idx[:, :, 1:].reshape(-1)], 1)]
    pairs = np.concatenate(pairs, 0)
    lo, hi = np.minimum(pairs[:, 0], pairs[:, 1]), np.maximum(pairs[:, 0], pairs[:, 1])
    order = np.lexsort((hi, lo))
    lo, hi = lo[order], hi[order]
    m = len(lo)
    Ri, Rj = Rt[lo], Rt[hi]
    Rij = np.swapaxes(Ri, 1, 2) @ Rj
    tij = (np.swapaxes(Ri, 1, 2) @ (pos[hi] - pos[lo])[:, :, None])[:, :, 0]
    Rmeas = Rij @ _rotvec_to_R(sigma_r * rng.standard_normal((m, 3)))
    tmeas = tij + sigma_t * rng.standard_normal((m, 3))
    meas = RelativeSEMeasurements(3, np.zeros(m), lo, np.zeros(m), hi, Rmeas, tmeas,
                                  np.full(m, 1.0 / (2.0 * sigma_r ** 2)), np.full(m, 1.0 / sigma_t ** 2),
                                  np.ones(m), lo + 1 == hi)
    Ttrue = np.zeros((n, 4, 3))
    Ttrue[:, :3, :] = np.swapaxes(Rt, 1, 2)
    Ttrue[:, 3, :] = pos
    return meas, n, Ttrue


def perturbed_truth(Ttrue: np.ndarray, seed: int = 2, sigma_t: float = 0.1, sigma_r: float = 0.2) -> np.ndarray:
    """Initial guess for C4: the ground truth perturbed by the measurement-noise model (chordal
    initialisation at 900k unknowns is outside the hot path)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = Ttrue.shape[0]
    R = np.swapaxes(Ttrue[:, :3, :], 1, 2) @ _rotvec_to_R(sigma_r * rng.standard_normal((n, 3)))
    T = Ttrue.copy()
    T[:, :3, :] = np.swapaxes(R, 1, 2)
    T[:, 3, :] += sigma_t * rng.standard_normal((n, 3))
    return T


def lift_tiles(T: np.ndarray, r: int) -> np.ndarray:
    """X = YLift T with YLift = [I_d; 0] (tiles [n, d+1, r]); the solve is invariant to the choice of
    YLift in St(d, r) (SURVEY 8c gauge note; reference uses fixedStiefelVariable, src/DPGO_utils.cpp:488-493)."""
    n, b, d = T.shape
    X = np.zeros((n, b, r))
    X[:, :, :d] = T
    return X


def write_g2o(path: str, meas: RelativeSEMeasurements) -> None:
    """EDGE_SE3:QUAT writer so the same reader is exercised (3-D only)."""
    assert meas.d == 3
    with open(path, "w") as fh:
        for e in range(len(meas)):
            R = meas.R[e]
            # rotation matrix -> quaternion (w, x, y, z)
            tr = np.trace(R)
            if tr > 0:
                s = 2.0 * np.sqrt(tr + 1.0); w = 0.25 * s
                x = (R[2, 1] - R[1, 2]) / s; y = (R[0, 2] - R[2, 0]) / s; z = (R[1, 0] - R[0, 1]) / s
            else:
                i = int(np.argmax(np.diag(R))); j, k = (i + 1) % 3, (i + 2) % 3
                s = 2.0 * np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k])
                v = [0.0, 0.0, 0.0]
                v[i] = 0.25 * s; v[j] = (R[j, i] + R[i, j]) / s; v[k] = (R[k, i] + R[i, k]) / s
                w = (R[k, j] - R[j, k]) / s; x, y, z = v
            it, ir = meas.tau[e], 2.0 * meas.kappa[e]
            info = [it, 0, 0, 0, 0, 0, it, 0, 0, 0, 0, it, 0, 0, 0, ir, 0, 0, ir, 0, ir]
            fh.write("EDGE_SE3:QUAT %d %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g %s\n" % (
                meas.p1[e], meas.p2[e], meas.t[e, 0], meas.t[e, 1], meas.t[e, 2], x, y, z, w,
                " ".join("%.17g" % v for v in info)))
