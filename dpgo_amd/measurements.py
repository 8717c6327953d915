"""Host-side data formats on either side of the hot path: relative SE(d) measurements,
the .g2o reader and the contiguous multi-robot partition.

Mirrors (names and semantics) of the reference:
  RelativeSEMeasurement            include/DPGO/RelativeSEMeasurement.h:21-50
  read_g2o_file                    src/DPGO_utils.cpp:113-257
  partition used by the demo       examples/MultiRobotExample.cpp:71-119
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Tuple

import numpy as np


@dataclass
class RelativeSEMeasurements:
    """Structure-of-arrays std::vector<RelativeSEMeasurement>: edge e goes from pose
    (r1[e], p1[e]) to (r2[e], p2[e]) with rotation R[e] (d x d), translation t[e],
    precisions kappa[e], tau[e], GNC weight[e] and the fixedWeight flag."""
    d: int
    r1: np.ndarray
    p1: np.ndarray
    r2: np.ndarray
    p2: np.ndarray
    R: np.ndarray
    t: np.ndarray
    kappa: np.ndarray
    tau: np.ndarray
    weight: np.ndarray
    fixedWeight: np.ndarray

    def __len__(self) -> int:
        return int(len(self.p1))

    def __post_init__(self):
        m = len(self.p1)
        self.r1 = np.ascontiguousarray(self.r1, dtype=np.int32)
        self.p1 = np.ascontiguousarray(self.p1, dtype=np.int32)
        self.r2 = np.ascontiguousarray(self.r2, dtype=np.int32)
        self.p2 = np.ascontiguousarray(self.p2, dtype=np.int32)
        self.R = np.ascontiguousarray(self.R, dtype=np.float64).reshape(m, self.d, self.d)
        self.t = np.ascontiguousarray(self.t, dtype=np.float64).reshape(m, self.d)
        self.kappa = np.ascontiguousarray(self.kappa, dtype=np.float64)
        self.tau = np.ascontiguousarray(self.tau, dtype=np.float64)
        self.weight = np.ascontiguousarray(self.weight, dtype=np.float64)
        self.fixedWeight = np.ascontiguousarray(self.fixedWeight, dtype=bool)

    def select(self, mask_or_idx) -> "RelativeSEMeasurements":
        k = mask_or_idx
        return RelativeSEMeasurements(self.d, self.r1[k], self.p1[k], self.r2[k], self.p2[k], self.R[k],
                                      self.t[k], self.kappa[k], self.tau[k], self.weight[k],
                                      self.fixedWeight[k])

    @staticmethod
    def concatenate(parts: List["RelativeSEMeasurements"]) -> "RelativeSEMeasurements":
        d = parts[0].d
        cat = lambda name: np.concatenate([getattr(p, name) for p in parts], axis=0)
        return RelativeSEMeasurements(d, cat("r1"), cat("p1"), cat("r2"), cat("p2"), cat("R"), cat("t"),
                                      cat("kappa"), cat("tau"), cat("weight"), cat("fixedWeight"))


def _quat_rot(w: float, x: float, y: float, z: float) -> np.ndarray:
    # Eigen::Quaterniond(w,x,y,z).toRotationMatrix(): the quaternion is NOT normalised
    # (src/DPGO_utils.cpp:215)
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[1.0 - (tyy + tzz), txy - twz, txz + twy],
                     [txy + twz, 1.0 - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, 1.0 - (txx + tyy)]])


def read_g2o_file(filename: str) -> Tuple[RelativeSEMeasurements, int]:
    """read_g2o_file (src/DPGO_utils.cpp:113-257): EDGE_SE2 / EDGE_SE3:QUAT lines become
    measurements with r1 = r2 = 0, weight 1, fixedWeight = (i + 1 == j); VERTEX_* lines are
    ignored; num_poses = 1 + max pose index.  Unknown tokens raise (reference: LOG(FATAL))."""
    p1, p2, Rs, ts, kap, tau = [], [], [], [], [], []
    d = 0
    with open(filename) as fh:
        for line in fh:
            tok = line.split()
            if not tok:
                continue
            kind = tok[0]
            if kind == "EDGE_SE2":
                i, j = int(tok[1]), int(tok[2])
                dx, dy, dth = float(tok[3]), float(tok[4]), float(tok[5])
                I11, I12, _I13, I22, _I23, I33 = (float(v) for v in tok[6:12])
                d = 2
                c, s = math.cos(dth), math.sin(dth)
                Rs.append(np.array([[c, -s], [s, c]]))
                ts.append(np.array([dx, dy]))
                cov = np.linalg.inv(np.array([[I11, I12], [I12, I22]]))
                tau.append(2.0 / (cov[0, 0] + cov[1, 1]))  # :174
                kap.append(I33)  # :176
            elif kind == "EDGE_SE3:QUAT":
                i, j = int(tok[1]), int(tok[2])
                dx, dy, dz, qx, qy, qz, qw = (float(v) for v in tok[3:10])
                I = [float(v) for v in tok[10:31]]
                d = 3
                Rs.append(_quat_rot(qw, qx, qy, qz))
                ts.append(np.array([dx, dy, dz]))
                tc = np.linalg.inv(np.array([[I[0], I[1], I[2]], [I[1], I[6], I[7]], [I[2], I[7], I[11]]]))
                tau.append(3.0 / np.trace(tc))  # :223
                rc = np.linalg.inv(np.array([[I[15], I[16], I[17]], [I[16], I[18], I[19]], [I[17], I[19], I[20]]]))
                kap.append(3.0 / (2.0 * np.trace(rc)))  # :230
            elif kind in ("VERTEX_SE2", "VERTEX_SE3:QUAT"):
                continue
            else:
                raise ValueError("Error: unrecognized type: %s!" % kind)
            p1.append(i)
            p2.append(j)
    m = len(p1)
    if m == 0:
        raise ValueError("no edges in %s" % filename)
    p1a, p2a = np.array(p1), np.array(p2)
    meas = RelativeSEMeasurements(d, np.zeros(m), p1a, np.zeros(m), p2a, np.array(Rs), np.array(ts),
                                  np.array(kap), np.array(tau), np.ones(m), p1a + 1 == p2a)
    return meas, int(max(p1a.max(), p2a.max())) + 1


def partition_contiguous(dataset: RelativeSEMeasurements, num_poses: int, num_robots: int):
    """The demo's partition (examples/MultiRobotExample.cpp:71-119): robot a owns the global pose
    range [a*per, (a+1)*per) (last robot takes the remainder); edges are relabelled to
    (robot, local index); a shared edge goes to BOTH endpoints' lists.
    Returns (ranges, [measurements of robot a])."""
    per = num_poses // num_robots
    if per <= 0:
        raise ValueError("More robots than total number of poses! Decrease the number of robots")
    starts = np.arange(num_robots) * per
    ends = starts + per
    ends[-1] = num_poses
    robot_of = np.minimum(np.arange(num_poses) // per, num_robots - 1)
    local = np.arange(num_poses) - starts[robot_of]
    m = len(dataset)
    # (the demo's relabelled measurements get weight 1 / fixedWeight false from the RelativeSEMeasurement
    # constructor; weights and flags of the input are carried through so GNC drivers can partition too)
    g = RelativeSEMeasurements(dataset.d, robot_of[dataset.p1], local[dataset.p1], robot_of[dataset.p2],
                               local[dataset.p2], dataset.R, dataset.t, dataset.kappa, dataset.tau,
                               dataset.weight.copy(), dataset.fixedWeight.copy())
    per_robot = [g.select((g.r1 == a) | (g.r2 == a)) for a in range(num_robots)]
    return [(int(s), int(e)) for s, e in zip(starts, ends)], per_robot


def relabel(dataset: RelativeSEMeasurements, new_index: np.ndarray) -> RelativeSEMeasurements:
    """The same measurements with pose i of a SINGLE index space (r1 = r2 = 0: a data set before partitioning) renamed
    new_index[i].  Weights and the fixedWeight flags (odometry edges, never re-weighted) travel with their edges."""
    if np.any(dataset.r1 != 0) or np.any(dataset.r2 != 0):
        raise ValueError("relabel() takes a data set with global pose indices (robot ids 0)")
    ni = np.asarray(new_index, dtype=np.int32)
    return RelativeSEMeasurements(dataset.d, dataset.r1, ni[dataset.p1], dataset.r2, ni[dataset.p2], dataset.R, dataset.t,
                                  dataset.kappa, dataset.tau, dataset.weight.copy(), dataset.fixedWeight.copy())


def locality_order(dataset: RelativeSEMeasurements, num_poses: int, num_robots: int, parts_per_robot: int = 8) -> np.ndarray:
    """Pose order for HBM-bound blocks (C ABI dpgo_locality_order, host code): every robot's contiguous block of the
    demo's partition keeps its poses; inside it, each of `parts_per_robot` contiguous chunks (the shares of the 8 XCDs in
    the kernels' tile walk, boundaries at workgroup tiles) is renumbered by reverse Cuthill-McKee over its own edges.
    Returns new_index[num_poses]: the position of every caller pose -- a permutation that maps every robot's range onto
    itself."""
    import ctypes as C
    from . import lib as L
    per = num_poses // num_robots
    starts = [a * per for a in range(num_robots)] + [num_poses]
    tile = (64 // (dataset.d + 1)) * 4  # poses per workgroup tile of the one-pose-per-(d+1)-lanes kernels
    out = np.arange(num_poses, dtype=np.int32)
    p1, p2 = dataset.p1.astype(np.int64), dataset.p2.astype(np.int64)
    for a in range(num_robots):
        s, e = starts[a], starts[a + 1]
        keep = (p1 >= s) & (p1 < e) & (p2 >= s) & (p2 < e) & (p1 != p2)
        i = np.concatenate([p1[keep], p2[keep]]) - s
        j = np.concatenate([p2[keep], p1[keep]]) - s
        order = np.lexsort((j, i))
        i, j = i[order], j[order]
        n = e - s
        rowptr = np.zeros(n + 1, dtype=np.int32)
        np.add.at(rowptr, i + 1, 1)
        rowptr = np.cumsum(rowptr).astype(np.int32)
        colidx = np.ascontiguousarray(j, dtype=np.int32)
        ni = np.zeros(n, dtype=np.int32)
        import os
        run = int(os.environ.get("DPGO_REORDER_RUN", "16"))  # (A/B knob of the benchmarks; 16 = a wave's poses)
        L.check(L.load().dpgo_locality_order_runs(n, L.ptr(rowptr), L.ptr(colidx), int(parts_per_robot), tile, run, L.ptr(ni)))
        out[s:e] = s + ni
    return out

