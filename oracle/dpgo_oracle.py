"""CPU oracle for the dpgo RBCD local solve -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product path (dpgo_amd/) never does; it fails loudly when the
HIP extension is missing.

This is a NumPy/SciPy restatement of the reference's per-agent Riemannian
block-coordinate-descent local solve.  Every function cites the reference
file:line (under /root/reference) whose arithmetic it follows.

PARITY STATUS ("parity unpinned" at trajectory level).  The reference cannot be
built in this image (Eigen3, SuiteSparse, glog, Boost and -- by git fetch at
configure time -- ROPTLIB `yuluntian/ROPTLIB@feature/cmake` are absent;
SURVEY.md section 8c).  RTR/tCG, the Stiefel projection / Hessian correction /
qf retraction live in ROPTLIB, which is not in the tree; they are restated here
from the published algorithm (Absil-Baker-Gallivan RTR; ROPTLIB SolversTR).
What pins this oracle:
  * the reference's own known-answer tests, restated in tests/test_oracle.py:
    tests/testTriangleGraph.cpp:57 (1e-4), tests/testPGO.cpp:188-189 (1e-6),
    tests/testUtils.cpp:28-54 (1e-5), tests/testEigenMap.cpp (layout);
  * literature optima of sphere2500 / torus3D / smallGrid3D (SE-Sync, DPGO
    papers) reproduced to 1e-9 relative.
Per-iteration trajectories are NOT pinned against the reference binary.

Memory layout.  The reference stores X as Eigen::MatrixXd r x (d+1)n,
column-major (include/DPGO/manifold/Poses.h:16-21).  The same bytes are viewed
here as a C-contiguous array X[n, d+1, r]: X[i, k, a] == X_ref(a, i*(d+1)+k).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

# --------------------------------------------------------------------------
# g2o reader -- src/DPGO_utils.cpp:113-257
# --------------------------------------------------------------------------


@dataclass
class Measurements:
    """SoA form of std::vector<RelativeSEMeasurement>
    (include/DPGO/RelativeSEMeasurement.h:21-50)."""
    d: int
    r1: np.ndarray  # int64 [m]
    p1: np.ndarray
    r2: np.ndarray
    p2: np.ndarray
    R: np.ndarray  # [m, d, d] (R[e][row, col])
    t: np.ndarray  # [m, d]
    kappa: np.ndarray
    tau: np.ndarray
    weight: np.ndarray
    fixed: np.ndarray  # bool

    @property
    def m(self) -> int:
        return len(self.p1)

    def subset(self, idx) -> "Measurements":
        idx = np.asarray(idx, dtype=np.int64)
        return Measurements(self.d, self.r1[idx], self.p1[idx], self.r2[idx], self.p2[idx],
                            self.R[idx], self.t[idx], self.kappa[idx], self.tau[idx],
                            self.weight[idx], self.fixed[idx])

    @staticmethod
    def empty(d: int) -> "Measurements":
        z = np.zeros(0, dtype=np.int64)
        f = np.zeros(0)
        return Measurements(d, z, z.copy(), z.copy(), z.copy(), np.zeros((0, d, d)), np.zeros((0, d)),
                            f, f.copy(), f.copy(), np.zeros(0, dtype=bool))

    @staticmethod
    def concat(parts: List["Measurements"]) -> "Measurements":
        d = parts[0].d
        cat = lambda name: np.concatenate([getattr(p, name) for p in parts], axis=0)
        return Measurements(d, cat("r1"), cat("p1"), cat("r2"), cat("p2"), cat("R"), cat("t"),
                            cat("kappa"), cat("tau"), cat("weight"), cat("fixed"))


def quat_to_rot_unnormalised(w, x, y, z):
    """Eigen::Quaterniond(w,x,y,z).toRotationMatrix() -- no normalisation
    (src/DPGO_utils.cpp:215; SURVEY 8c')."""
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[1 - (tyy + tzz), txy - twz, txz + twy],
                     [txy + twz, 1 - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, 1 - (txx + tyy)]])


def read_g2o(path: str) -> Tuple[Measurements, int]:
    """src/DPGO_utils.cpp:113-257.  Returns (measurements, num_poses)."""
    p1, p2, Rs, ts, kap, tau, fixed = [], [], [], [], [], [], []
    d = None
    with open(path) as fh:
        for line in fh:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "EDGE_SE2":  # :144-182
                i, j = int(tok[1]), int(tok[2])
                dx, dy, dth = map(float, tok[3:6])
                I11, I12, I13, I22, I23, I33 = map(float, tok[6:12])
                d = 2
                c, s = math.cos(dth), math.sin(dth)
                R = np.array([[c, -s], [s, c]])
                t = np.array([dx, dy])
                TranCov = np.array([[I11, I12], [I12, I22]])
                tau_e = 2.0 / np.trace(np.linalg.inv(TranCov))  # :174
                kap_e = I33  # :176
            elif tok[0] == "EDGE_SE3:QUAT":  # :184-236
                i, j = int(tok[1]), int(tok[2])
                dx, dy, dz, qx, qy, qz, qw = map(float, tok[3:10])
                I = list(map(float, tok[10:31]))
                (I11, I12, I13, I14, I15, I16, I22, I23, I24, I25, I26,
                 I33, I34, I35, I36, I44, I45, I46, I55, I56, I66) = I
                d = 3
                R = quat_to_rot_unnormalised(qw, qx, qy, qz)
                t = np.array([dx, dy, dz])
                TranCov = np.array([[I11, I12, I13], [I12, I22, I23], [I13, I23, I33]])
                tau_e = 3.0 / np.trace(np.linalg.inv(TranCov))  # :223
                RotCov = np.array([[I44, I45, I46], [I45, I55, I56], [I46, I56, I66]])
                kap_e = 3.0 / (2.0 * np.trace(np.linalg.inv(RotCov)))  # :230
            elif tok[0] in ("VERTEX_SE2", "VERTEX_SE3:QUAT"):  # :238-240
                continue
            else:
                raise ValueError("unrecognized g2o token %r" % tok[0])
            p1.append(i); p2.append(j); Rs.append(R); ts.append(t)
            kap.append(kap_e); tau.append(tau_e); fixed.append(i + 1 == j)
    m = len(p1)
    z = np.zeros(m, dtype=np.int64)
    meas = Measurements(d, z, np.array(p1, dtype=np.int64), z.copy(), np.array(p2, dtype=np.int64),
                        np.array(Rs), np.array(ts), np.array(kap), np.array(tau),
                        np.ones(m), np.array(fixed, dtype=bool))
    n = int(max(meas.p1.max(), meas.p2.max())) + 1  # :246-254
    return meas, n


# --------------------------------------------------------------------------
# Partition -- examples/MultiRobotExample.cpp:71-119
# --------------------------------------------------------------------------


def partition_contiguous(meas: Measurements, n: int, num_robots: int):
    """Contiguous index-range partition.  Returns (ranges, per_robot) where
    per_robot[a] = dict(odometry, private, shared) in local pose indices; a shared
    edge is pushed to BOTH endpoints' lists (MultiRobotExample.cpp:113-117)."""
    per = n // num_robots
    assert per > 0
    starts = [a * per for a in range(num_robots)]
    ends = [(a + 1) * per for a in range(num_robots)]
    ends[-1] = n
    robot_of = np.minimum(np.arange(n) // per, num_robots - 1)
    local = np.arange(n) - np.array(starts)[robot_of]
    # the demo relabels with weight 1 / fixedWeight false (RelativeSEMeasurement ctor); weights and flags of the
    # input are carried through here so that GNC drivers can partition a re-weighted graph
    g = Measurements(meas.d, robot_of[meas.p1], local[meas.p1], robot_of[meas.p2], local[meas.p2],
                     meas.R, meas.t, meas.kappa, meas.tau, meas.weight.copy(), meas.fixed.copy())
    out = []
    for a in range(num_robots):
        same = (g.r1 == a) & (g.r2 == a)
        odo = same & (g.p1 + 1 == g.p2)
        prv = same & ~odo
        shr = ((g.r1 == a) | (g.r2 == a)) & ~same
        out.append(dict(odometry=g.subset(np.nonzero(odo)[0]), private=g.subset(np.nonzero(prv)[0]),
                        shared=g.subset(np.nonzero(shr)[0])))
    return list(zip(starts, ends)), out


# --------------------------------------------------------------------------
# Connection Laplacian and the agent-local data matrices
#   src/DPGO_utils.cpp:272-344, src/PoseGraph.cpp:381-491 (Q), :493-580 (G)
# --------------------------------------------------------------------------


def _T_Omega(meas: Measurements):
    """T = [R t; 0 1], Omega = diag(w*kappa (x d), w*tau)  (DPGO_utils.cpp:307-329)."""
    m, d = meas.m, meas.d
    b = d + 1
    T = np.zeros((m, b, b))
    T[:, :d, :d] = meas.R
    T[:, :d, d] = meas.t
    T[:, d, d] = 1.0
    om = np.empty((m, b))
    om[:, :d] = (meas.weight * meas.kappa)[:, None]
    om[:, d] = meas.weight * meas.tau
    return T, om


class BSR:
    """Block-CSR of the symmetric (d+1)n x (d+1)n matrix Q.  vals[t] is the dense
    b x b block Q[i*b:(i+1)*b, j*b:(j+1)*b] stored ROW-major (scipy.sparse.bsr_matrix
    layout); colidx sorted within each block row."""

    def __init__(self, n, b, rowptr, colidx, vals):
        self.n, self.b = n, b
        self.rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        self.colidx = np.ascontiguousarray(colidx, dtype=np.int32)
        self.vals = np.ascontiguousarray(vals, dtype=np.float64)

    @property
    def nnzb(self):
        return len(self.colidx)

    def to_scipy(self):
        N = self.n * self.b
        return sp.bsr_matrix((self.vals, self.colidx, self.rowptr), shape=(N, N))

    def diag_blocks(self):
        rows = np.repeat(np.arange(self.n), np.diff(self.rowptr))
        sel = np.nonzero(rows == self.colidx)[0]
        out = np.zeros((self.n, self.b, self.b))
        out[rows[sel]] = self.vals[sel]
        return out


def bsr_from_block_triplets(n, b, bi, bj, blocks) -> BSR:
    """Sum duplicate (bi,bj) blocks, sort by (row, col)."""
    bi = np.asarray(bi, dtype=np.int64); bj = np.asarray(bj, dtype=np.int64)
    key = bi * n + bj
    order = np.argsort(key, kind="stable")
    key_s = key[order]
    uniq, first = np.unique(key_s, return_index=True)
    vals = np.add.reduceat(blocks[order], first, axis=0) if len(key_s) else np.zeros((0, b, b))
    rows = uniq // n
    cols = uniq % n
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, rows + 1, 1)
    rowptr = np.cumsum(rowptr)
    return BSR(n, b, rowptr, cols, vals)


def active_shared_edges(shared: Optional[Measurements], my_id: int, inactive=None, use_inactive: bool = False,
                        neighbor_poses=None) -> Optional[Measurements]:
    """The shared edges constructQ / constructG use (src/PoseGraph.cpp:418-430, 442-454, 520-532, 545-557): an edge with
    an INACTIVE neighbour (PoseGraph::setNeighborActive, set by PGOAgent::setRobotActive, src/PGOAgent.cpp:1173-1184) is
    skipped unless use_inactive_neighbors_ is set and its pose is available."""
    if shared is None or not shared.m or not inactive:
        return shared
    keep = []
    for e in range(shared.m):
        nb = (int(shared.r2[e]), int(shared.p2[e])) if shared.r1[e] == my_id else (int(shared.r1[e]), int(shared.p1[e]))
        if nb[0] in inactive and not (use_inactive and neighbor_poses is not None and nb in neighbor_poses):
            continue
        keep.append(e)
    return shared.subset(np.array(keep, dtype=np.int64))


def construct_Q(n: int, d: int, private: Measurements, shared: Optional[Measurements] = None,
                my_id: int = 0, priors: Optional[Dict[int, np.ndarray]] = None,
                prior_kappa: float = 10000.0, prior_tau: float = 100.0) -> BSR:
    """PoseGraph::constructQ (src/PoseGraph.cpp:381-491) with
    constructConnectionLaplacianSE (src/DPGO_utils.cpp:272-344).

    Private edge (i->j): Q_ii += T Om T^T, Q_jj += Om, Q_ij = -T Om, Q_ji = Q_ij^T.
    Shared edge, mine = source (outgoing): Q[p1,p1] += T Om T^T (:431-434);
    mine = destination (incoming): Q[p2,p2] += Om (:455-457).
    Prior on pose idx: Q[idx,idx] += diag(prior_kappa.., prior_tau) (:461-468).
    Every pose gets an (explicit) diagonal block, as the reference's QDiag does (:470-485).
    """
    b = d + 1
    bi, bj, blk = [np.arange(n)], [np.arange(n)], [np.zeros((n, b, b))]
    if private is not None and private.m:
        T, om = _T_Omega(private)
        TO = T * om[:, None, :]  # T @ diag(om)
        TOT = TO @ np.transpose(T, (0, 2, 1))
        Om = np.zeros((private.m, b, b))
        Om[:, np.arange(b), np.arange(b)] = om
        i, j = private.p1, private.p2
        bi += [i, j, i, j]
        bj += [i, j, j, i]
        blk += [TOT, Om, -TO, -np.transpose(TO, (0, 2, 1))]
    if shared is not None and shared.m:
        T, om = _T_Omega(shared)
        TO = T * om[:, None, :]
        TOT = TO @ np.transpose(T, (0, 2, 1))
        Om = np.zeros((shared.m, b, b))
        Om[:, np.arange(b), np.arange(b)] = om
        out = shared.r1 == my_id
        if out.any():
            bi.append(shared.p1[out]); bj.append(shared.p1[out]); blk.append(TOT[out])
        if (~out).any():
            bi.append(shared.p2[~out]); bj.append(shared.p2[~out]); blk.append(Om[~out])
    if priors:
        idx = np.array(sorted(priors.keys()), dtype=np.int64)
        P = np.zeros((len(idx), b, b))
        P[:, np.arange(d), np.arange(d)] = prior_kappa
        P[:, d, d] = prior_tau
        bi.append(idx); bj.append(idx); blk.append(P)
    return bsr_from_block_triplets(n, b, np.concatenate(bi), np.concatenate(bj), np.concatenate(blk, axis=0))


def construct_G(n: int, d: int, r: int, shared: Optional[Measurements], my_id: int,
                neighbor_poses: Dict[Tuple[int, int], np.ndarray],
                priors: Optional[Dict[int, np.ndarray]] = None,
                prior_kappa: float = 10000.0, prior_tau: float = 100.0) -> np.ndarray:
    """PoseGraph::constructG (src/PoseGraph.cpp:493-580).  neighbor_poses maps
    (robot, frame) -> tile [d+1, r] (the LiftedPose r x (d+1), column-major).
    Outgoing: G[:,p1] += -X_j Om T^T (:533-537); incoming: G[:,p2] += -X_i T Om (:558-562);
    prior: G[:,idx] += -P Om (:565-574).  Returns G in the [n, d+1, r] view."""
    b = d + 1
    G = np.zeros((n, b, r))
    if shared is not None and shared.m:
        T, om = _T_Omega(shared)
        for e in range(shared.m):
            if shared.r1[e] == my_id:
                Xj = neighbor_poses[(int(shared.r2[e]), int(shared.p2[e]))]  # [b, r] == (r x b)^T
                # L = -Xj Om T^T  ->  L^T = -T Om Xj^T
                G[shared.p1[e]] += -(T[e] * om[e][None, :]) @ Xj
            else:
                Xi = neighbor_poses[(int(shared.r1[e]), int(shared.p1[e]))]
                # L = -Xi T Om  ->  L^T = -Om T^T Xi^T
                G[shared.p2[e]] += -(om[e][:, None] * T[e].T) @ Xi
    if priors:
        om = np.array([prior_kappa] * d + [prior_tau])
        for idx, P in priors.items():
            G[idx] += -(om[:, None] * P)
    return G


# --------------------------------------------------------------------------
# Manifold (St(d,r) x R^r)^n -- ROPTLIB Stiefel (ChooseStieParamsSet3: Euclidean metric,
# qf retraction, extrinsic representation) x Euclidean, configured at
# src/manifold/LiftedSEManifold.cpp:16-24.  Restated from the published algorithm
# (SURVEY.md 8c' items 1-3).
# --------------------------------------------------------------------------


def sym(A):
    return 0.5 * (A + np.swapaxes(A, -1, -2))


def tangent_project(X, W, d):
    """ROPTLIB Stiefel::ExtrProjection per pose: W_rot - Y sym(Y^T W_rot); translation
    column untouched (ProductManifold::Projection).  X, W: [n, d+1, r]."""
    Y = X[:, :d, :]  # [n, d, r] == Y^T
    Wr = W[:, :d, :]
    S = sym(Y @ np.swapaxes(Wr, 1, 2))  # (Y^T W)[k,c] = sum_a Y[a,k] W[a,c]
    out = W.copy()
    out[:, :d, :] = Wr - np.swapaxes(S, 1, 2) @ Y  # (Y S)^T = S^T Y^T
    return out


def qf_retract(X, eta, d):
    """ROPTLIB Stiefel::qfRetraction: Q-factor of the thin QR of Y+eta with diag(R) > 0;
    Euclidean factor p + eta (SURVEY 8c' item 2).  Implemented as modified Gram-Schmidt,
    which equals any positive-diagonal thin QR up to round-off."""
    out = X + eta
    A = out[:, :d, :]  # rows are the columns of Y+eta
    Q = np.empty_like(A)
    for k in range(d):
        v = A[:, k, :].copy()
        for l in range(k):
            v -= np.sum(Q[:, l, :] * v, axis=1, keepdims=True) * Q[:, l, :]
        Q[:, k, :] = v / np.linalg.norm(v, axis=1, keepdims=True)
    out[:, :d, :] = Q
    return out


def polar_project(M, d):
    """LiftedSEManifold::project (src/manifold/LiftedSEManifold.cpp:34-45) ->
    projectToStiefelManifold (src/DPGO_utils.cpp:480-486): U V^T of the thin SVD of the
    r x d block; translations copied."""
    out = M.copy()
    A = np.swapaxes(M[:, :d, :], 1, 2)  # [n, r, d]
    U, _, Vt = np.linalg.svd(A, full_matrices=False)
    out[:, :d, :] = np.swapaxes(U @ Vt, 1, 2)
    return out


def project_to_rotation_group(M):
    """src/DPGO_utils.cpp:464-478."""
    U, _, Vt = np.linalg.svd(M)
    if np.linalg.det(U) * np.linalg.det(Vt) > 0:
        return U @ Vt
    U = U.copy(); U[:, -1] *= -1
    return U @ Vt


# --------------------------------------------------------------------------
# QuadraticProblem -- src/QuadraticProblem.cpp
# --------------------------------------------------------------------------


AMG_DENSE, AMG_DENSE_MAX = 3200, 6400  # unknowns of the dense coarsest operator (82 MB / 328 MB in fp64)


AMG_GRAPH_MAX, AMG_GRAPH_UNKNOWNS_PER_POSE = 512, 1600
AMG_MERGE_FROM, AMG_MERGED_UNKNOWNS_PER_POSE = 12, 2200  # mirrors kMlMergeFrom / kMlMergedUnknownsPerPose


def amg_default_graph_size(n: int, b: int) -> int:
    """Mirrors ml_default_graph_size (dpgo_amd/csrc/multilevel.hip): the largest aggregate of the default two-level
    hierarchy with GRAPH aggregates, 0 where the default is a hierarchy of index runs (more than one coarsening needed)."""
    if os.environ.get("DPGO_ML_GRAPH", "1") == "0":
        return 0
    if int(os.environ.get("DPGO_ML_GRAPH_SIZE", "0")) >= 2:  # experiments: force the size
        return int(os.environ["DPGO_ML_GRAPH_SIZE"])
    S = max(4, -(-(n * b) // AMG_GRAPH_UNKNOWNS_PER_POSE))
    return S if S <= AMG_GRAPH_MAX else 0


AMG_GROWTH_CHUNKS_FROM = 65536  # poses from which the aggregates grow and merge inside 8 index ranges (ml_growth_chunks)


def amg_growth_chunks(n: int) -> int:
    """Mirrors ml_growth_chunks (dpgo_amd/csrc/multilevel.hip): the number of contiguous index ranges
    [n c / chunks, n (c + 1) / chunks) inside which large blocks grow and merge their aggregates independently (the device
    library gives every range a host thread; DPGO_ML_GROWTH_CHUNKS overrides the count on both sides)."""
    forced = int(os.environ.get("DPGO_ML_GROWTH_CHUNKS", "0") or 0)
    if forced > 0:
        return max(1, min(forced, max(1, n // 64)))
    return 8 if n >= AMG_GROWTH_CHUNKS_FROM else 1


def _amg_grow_range(rowptr, colidx, lo: int, hi: int, S: int, lab, parent, pslot):
    """Growth inside [lo, hi) (grow_range): aggregate ids local to the range; returns (ptr, mem) of the range."""
    mem: List[int] = []
    ptr = [0]
    na = 0
    for s in range(lo, hi):
        if lab[s] >= 0:
            continue
        first = len(mem)
        lab[s] = na
        mem.append(s)
        head = first
        while head < len(mem) and len(mem) - first < S:
            u = mem[head]
            head += 1
            for t in range(rowptr[u], rowptr[u + 1]):
                if len(mem) - first >= S:
                    break
                v = colidx[t]
                if v < lo or v >= hi or lab[v] >= 0:
                    continue
                lab[v] = na
                parent[v] = u
                pslot[v] = t
                mem.append(v)
        ptr.append(len(mem))
        na += 1
    return ptr, mem


def amg_graph_aggregates(Q: "BSR", S: int, chunks: Optional[int] = None):
    """Mirrors ml_graph_aggregates: aggregates of at most S nodes grown greedily over Q's block pattern -- seeds in index
    order; a seed's aggregate takes unassigned nodes in breadth-first order (FIFO; a node's neighbours in the order of its
    block row) until it holds S -- independently inside each of amg_growth_chunks(n) contiguous index ranges (a search
    does not leave its range), the ranges' aggregates numbered one range after the other.  Returns (lab[n], ptr[na+1],
    mem[n] in discovery order, parent[n] (-1: root), pslot[n] = slot of block (parent, node))."""
    n = Q.n
    chunks = amg_growth_chunks(n) if chunks is None else max(1, min(int(chunks), max(1, n)))
    rowptr, colidx = np.asarray(Q.rowptr), np.asarray(Q.colidx)
    lab = -np.ones(n, dtype=np.int64)
    parent = -np.ones(n, dtype=np.int64)
    pslot = np.zeros(n, dtype=np.int64)
    mem: List[int] = []
    ptr = [0]
    na = 0
    for c in range(chunks):
        lo, hi = n * c // chunks, n * (c + 1) // chunks
        ptr_c, mem_c = _amg_grow_range(rowptr, colidx, lo, hi, S, lab, parent, pslot)
        lab[lo:hi] += na
        ptr.extend(len(mem) + q for q in ptr_c[1:])
        mem.extend(mem_c)
        na += len(ptr_c) - 1
    return lab, np.asarray(ptr, dtype=np.int64), np.asarray(mem, dtype=np.int64), parent, pslot


def _amg_merge_range(rowptr, colidx, lo: int, hi: int, S: int, cap: int, lab, members, new_lab, parent, pslot):
    """merge_range: `members` = the range's aggregates (lists of poses), lab = their ids local to the range at the range's
    indices.  Writes new_lab (ids local to the range), parent, pslot at the range's indices; returns (ptr, mem) of the range."""
    na = len(members)
    changed = True
    while changed:
        changed = False
        for a in range(na):
            if not members[a] or 2 * len(members[a]) > S:
                continue
            conn: Dict[int, int] = {}
            for i in members[a]:
                for t in range(rowptr[i], rowptr[i + 1]):
                    v = colidx[t]
                    if v < lo or v >= hi:
                        continue
                    c = int(lab[v])
                    if c != a:
                        conn[c] = conn.get(c, 0) + 1
            best, best_n = -1, 0
            for c in sorted(conn):
                if len(members[c]) + len(members[a]) <= cap and conn[c] > best_n:
                    best, best_n = c, conn[c]
            if best >= 0:
                members[best].extend(members[a])
                for i in members[a]:
                    lab[i] = best
                members[a] = []
                changed = True
    # renumber by smallest member, rebuild the breadth-first trees
    alive = [a for a in range(na) if members[a]]
    alive.sort(key=lambda a: min(members[a]))
    out_mem: List[int] = []
    out_ptr = [0]
    for k, a in enumerate(alive):
        # (a merged aggregate is connected by construction, so the search from its smallest member reaches everything;
        # should Q's pattern not be symmetric, the members it misses become further roots in index order)
        for root in sorted(members[a]):
            if new_lab[root] >= 0:
                continue
            head = len(out_mem)
            new_lab[root] = k
            out_mem.append(root)
            while head < len(out_mem):
                u = out_mem[head]
                head += 1
                for t in range(rowptr[u], rowptr[u + 1]):
                    v = colidx[t]
                    if v < lo or v >= hi or lab[v] != a or new_lab[v] >= 0:
                        continue
                    new_lab[v] = k
                    parent[v] = u
                    pslot[v] = t
                    out_mem.append(v)
        out_ptr.append(len(out_mem))
    return out_ptr, out_mem


def amg_merge_small_aggregates(Q: "BSR", S: int, lab, ptr, mem, cap: Optional[int] = None, chunks: Optional[int] = None):
    """Mirrors ml_merge_small_aggregates: the greedy growth leaves fragments (pockets between full aggregates); where an
    aggregate is a workgroup of the one-launch solve every fragment costs a whole workgroup.  Passes over the aggregates
    in index order until nothing changes: an aggregate of at most S / 2 nodes joins the neighbouring aggregate (one it
    shares a block of Q with) it has the most blocks in common with among those that still have room (sizes add up to at
    most `cap`; ties: the lower index).  Afterwards the aggregates are renumbered in the order of their smallest member and
    every aggregate's breadth-first tree is rebuilt from that member (neighbours in block-row order).  Like the growth,
    independently inside each of amg_growth_chunks(n) index ranges (aggregates that do not sit inside one range -- not the
    growth's output -- are merged as one range).  Returns (lab, ptr, mem, parent, pslot) like amg_graph_aggregates."""
    n = Q.n
    cap = S if cap is None else cap
    chunks = amg_growth_chunks(n) if chunks is None else max(1, min(int(chunks), max(1, n)))
    rowptr, colidx = np.asarray(Q.rowptr), np.asarray(Q.colidx)
    lab = np.array(lab, dtype=np.int64)
    na = len(ptr) - 1
    members = [[int(v) for v in mem[ptr[a]:ptr[a + 1]]] for a in range(na)]
    bound = lambda c: n * c // chunks  # noqa: E731
    # the aggregates of every range (numbered range after range by the growth)
    first_agg = [0] * (chunks + 1)
    ok = chunks > 1
    if ok:
        c = 0
        for a in range(na):
            if not members[a]:
                ok = False
                break
            lo_m, hi_m = min(members[a]), max(members[a])
            while c + 1 < chunks and lo_m >= bound(c + 1):
                c += 1
                first_agg[c] = a
            if lo_m < bound(c) or hi_m >= bound(c + 1):
                ok = False
                break
        while c + 1 < chunks:
            c += 1
            first_agg[c] = na
        first_agg[chunks] = na
    if not ok:
        chunks, first_agg = 1, [0, na]
        bound = lambda c: n * c  # noqa: E731
    new_lab = -np.ones(n, dtype=np.int64)
    parent = -np.ones(n, dtype=np.int64)
    pslot = np.zeros(n, dtype=np.int64)
    out_mem: List[int] = []
    out_ptr = [0]
    total = 0
    for c in range(chunks):
        lo, hi = bound(c), bound(c + 1)
        a0, a1 = first_agg[c], first_agg[c + 1]
        lab[lo:hi] -= a0
        ptr_c, mem_c = _amg_merge_range(rowptr, colidx, lo, hi, S, cap, lab, members[a0:a1], new_lab, parent, pslot)
        new_lab[lo:hi] += total
        out_ptr.extend(len(out_mem) + q for q in ptr_c[1:])
        out_mem.extend(mem_c)
        total += len(ptr_c) - 1
    return new_lab, np.asarray(out_ptr, dtype=np.int64), np.asarray(out_mem, dtype=np.int64), parent, pslot


def amg_tree_prolongation(Q: "BSR", d: int, mem, parent, pslot):
    """Mirrors k_ml_build_P_tree: Pb[i] = G(root of i's aggregate -> i)^T, composed along the aggregate's breadth-first
    tree.  The relative pose of a tree edge parent -> i is read off the block Q[parent, i]: a measurement parent -> i
    leaves -T Om = -[w kappa R, w tau t; 0, w tau] there (last row zero but for -w tau), a measurement i -> parent its
    transpose -(T' Om)^T (last column zero but for -w tau; T = T'^-1).  Anything else (zero weight, several measurements
    summed): the chain restarts at the identity."""
    b = d + 1
    Pb = np.zeros((Q.n, b, b))
    for i in mem:  # discovery order: a parent precedes its children
        G = np.eye(b)
        par = parent[i]
        if par >= 0:
            blk = Q.vals[pslot[i]]
            wt = -blk[d, d]
            wk = np.linalg.norm(blk[:d, 0])
            fwd, bwd = bool(np.all(blk[d, :d] == 0.0)), bool(np.all(blk[:d, d] == 0.0))
            if wt > 0 and wk > 0 and (fwd or bwd):
                T = np.eye(b)
                T[:d, :d] = -blk[:d, :d] / wk
                T[:d, d] = (-blk[:d, d] / wt) if fwd else T[:d, :d] @ (blk[d, :d] / wt)
                G = Pb[par].T @ T
        Pb[i] = G.T
    return Pb


def amg_default_ks(n: int, b: int, split0: Optional[int] = None) -> List[int]:
    """Aggregate sizes (one per coarsening) of the device's multilevel preconditioner; mirrors ml_default_ks
    (dpgo_amd/csrc/multilevel.hip).  Every k divides the workgroup tile of its level ((64 / (b split)) * 4 nodes, split = 4
    lane groups per node below 40 000 nodes, else 1); the coarsest operator is a dense inverse of at most AMG_DENSE
    unknowns, AMG_DENSE_MAX if that is what it takes to get there in one coarsening; otherwise one more level."""
    S = amg_default_graph_size(n, b)
    if S:
        # (blocks beyond ~17 600 unknowns: aggregates grown to ceil(n b / 2 200) poses, fragments merged up to 3/2 of that)
        if S >= AMG_MERGE_FROM and "DPGO_ML_GRAPH_SIZE" not in os.environ:
            Sm = -(-(n * b) // AMG_MERGED_UNKNOWNS_PER_POSE)
            if Sm + Sm // 2 <= AMG_GRAPH_MAX:
                return [-Sm, -(Sm + Sm // 2)]
        return [-S]
    lsplit = lambda m: 4 if m < 40000 else 1  # noqa: E731
    ks: List[int] = []
    cur, split = n, (split0 or lsplit(n))
    for _ in range(16):
        P = (64 // (b * split)) * 4
        pick = 0
        for limit in (AMG_DENSE, AMG_DENSE_MAX):
            for k in range(4, P + 1):
                if P % k == 0 and ((cur + k - 1) // k) * b <= limit:
                    pick = k
                    break
            if pick:
                break
        if pick:
            ks.append(pick)
            return ks
        k = max(c for c in range(2, 9) if P % c == 0)
        ks.append(k)
        cur = (cur + k - 1) // k
        split = lsplit(cur)
    return ks


def amg_chain_prolongations(Q: "BSR", d: int, ks: List[int]):
    """Prolongation blocks of every coarsening.  Level l has n_l nodes; node a stands for the run of stride_l
    consecutive poses that starts at the fine pose a * stride_l (its root), stride_0 = 1, stride_{l+1} = k_l stride_l.
    Pb[l][a] = G(root of a's parent -> root of a)^T: the homogeneous relative pose composed along the ODOMETRY chain of
    the fine graph, read off Q's own blocks: for an edge i -> i+1 the block Q_{i,i+1} is
    -T Om = -[w kappa R, w tau t; 0, w tau] (src/DPGO_utils.cpp:307-329), so w tau = -Q[d][d], t = -Q[:d, d] / (w tau),
    w kappa = |first column of Q[:d, :d]|, R = -Q[:d, :d] / (w kappa).  Missing block: the chain restarts at identity.
    With these blocks P_0 P_1 ... C reproduces, on every aggregate, the kernel vectors V_i = G_i^T C of the chain's
    Laplacian ("rigid-body" coarse modes)."""
    n, b = Q.n, d + 1
    # relative pose of every odometry link i-1 -> i (None where the chain is broken)
    link = [None] * n
    for i in range(1, n):
        blk = None
        for t in range(Q.rowptr[i - 1], Q.rowptr[i]):
            if Q.colidx[t] == i:
                blk = Q.vals[t]
        if blk is not None and -blk[d, d] > 0:
            wt = -blk[d, d]
            wk = np.linalg.norm(blk[:d, 0])
            if wk > 0:
                T = np.eye(b)
                T[:d, :d] = -blk[:d, :d] / wk
                T[:d, d] = -blk[:d, d] / wt
                link[i] = T
    out, stride, cur = [], 1, n
    for k in ks:
        span = stride * k
        Pb = np.zeros((cur, b, b))
        G = np.eye(b)
        for i in range(n):
            if i % span == 0:
                G = np.eye(b)
            else:
                G = G @ link[i] if link[i] is not None else np.eye(b)
            if i % stride == 0:
                Pb[i // stride] = G.T
        out.append(Pb)
        stride, cur = span, (cur + k - 1) // k
    return out


def amg_prolongation_blocks(Q: "BSR", d: int, k: int):
    """Two-level case: Pb[i] = G(root -> i)^T for the aggregate {root = (i // k) k, ..., root + k - 1}."""
    return amg_chain_prolongations(Q, d, [k])[0]


class QuadraticProblem:
    """f(X) = 0.5 <Q, X^T X> + <X, G>  (include/DPGO/QuadraticProblem.h:28-32).

    precond: 'exact'  -> (Q + 0.1 I)^-1 via sparse LU of the SPD matrix, the reference's
                         CHOLMOD path (src/PoseGraph.cpp:598-613, src/QuadraticProblem.cpp:56-69);
             'jacobi' -> inverse of the (d+1)x(d+1) diagonal blocks of Q + 0.1 I (what the
                         MI355X path runs; same fixed point, different trajectory);
             'none'   -> identity;
             'amg'    -> aggregation-multigrid V(1,1) cycle for Q + 0.1 I (the device's default "multilevel"
                         preconditioner): level l+1's nodes = runs of amg_k[l] consecutive level-l
                         nodes, prolongation blocks = relative poses read off Q's odometry blocks, Galerkin operators,
                         dense inverse on the coarsest level, damped block-Jacobi pre- and post-smoothing.
             'amg_additive' -> the ADDITIVE two-level combination on the same hierarchy (the device's "additive"
                         preconditioner, what its persistent kernel runs on small blocks):
                         M^-1 r = amg_add_w Dinv r + P A_c^-1 P^T r   (block-Jacobi plus the coarse-grid correction of the
                         residual itself: no operator application inside the preconditioner), amg_add_w = 1.
    All of them are followed by the tangent projection (QuadraticProblem.cpp:68)."""

    def __init__(self, Q: BSR, G: Optional[np.ndarray], r: int, d: int, precond: str = "exact",
                 shift: float = 0.1, amg_k=None, amg_omega: float = 0.7, amg_gamma: int = 1, amg_nu: int = 1,
                 amg_coarse_bits: int = 64, amg_merge: int = 0, amg_operator_bits: int = 64,
                 amg_vector_bits: Optional[int] = None):
        self.amg_k, self.amg_omega, self._amg = amg_k, amg_omega, None
        # storage precision of the level-0 OPERATOR COPIES the device's cycle streams on HBM-bound blocks (device default
        # 64; 32 = opt-in, dpgo_problem_multilevel_operator_bits): Q's values in the residual r - A x1, the values of A P
        # in the post-smoothing, the prolongation blocks in both -- two-level hierarchies only (amg_cycle)
        self.amg_operator_bits = amg_operator_bits
        # ... and of the two vectors that live inside such a cycle (pre-smoothed iterate, kept residual): the operator
        # copies' precision unless told otherwise (the device's A/B switch DPGO_ML_VECTOR_BITS)
        self.amg_vector_bits = amg_operator_bits if amg_vector_bits is None else amg_vector_bits
        self.amg_merge = amg_merge  # graph aggregates: > 0 = join the fragments of the greedy growth up to this many poses (amg_merge_small_aggregates)
        self.amg_coarse_bits = amg_coarse_bits  # storage precision of the dense level (device default: 64; 32 = opt-in)
        self.amg_gamma, self.amg_nu = amg_gamma, amg_nu  # coarse-level cycle index / smoothing sweeps (experiments)
        self.Q, self.r, self.d, self.n = Q, r, d, Q.n
        self.b = d + 1
        self.N = self.n * self.b
        self.Qs = Q.to_scipy().tocsr()
        self.G = np.zeros((self.n, self.b, r)) if G is None else G
        if precond == "amg_additive":
            self.amg_additive, self.amg_add_w = True, 1.0
        self.precond = precond
        self.shift = shift
        self._lu = None
        self._dinv = None
        self.n_spmm = 0

    # --- SpMM: (X Q)^T = Q X^T (Q symmetric) ---
    def XQ(self, X):
        self.n_spmm += 1
        return (self.Qs @ X.reshape(self.N, self.r)).reshape(X.shape)

    def f(self, X):  # QuadraticProblem.cpp:29-41
        return 0.5 * np.sum(self.XQ(X) * X) + np.sum(X * self.G)

    def euc_grad(self, X):  # :43-47
        return self.XQ(X) + self.G

    def euc_hess(self, V):  # :49-54
        return self.XQ(V)

    def rie_grad(self, X):  # :71-79
        return tangent_project(X, self.euc_grad(X), self.d)

    def rie_grad_norm(self, X):  # :81-83
        return float(np.linalg.norm(self.rie_grad(X)))

    def rie_hess(self, X, S, V):
        """ROPTLIB Stiefel::EucHvToHv, Euclidean metric (SURVEY 8c' item 3):
        proj_X( V Q - V_rot sym(Y^T EG_rot) ),  S = sym(Y^T EG_rot) cached per outer iterate."""
        H = self.XQ(V)
        d = self.d
        H[:, :d, :] -= np.swapaxes(S, 1, 2) @ V[:, :d, :]
        return tangent_project(X, H, d)

    def sym_ytg(self, X, EG):
        d = self.d
        return sym(X[:, :d, :] @ np.swapaxes(EG[:, :d, :], 1, 2))

    def dinv_blocks(self):
        if self._dinv is None:
            D = self.Q.diag_blocks() + self.shift * np.eye(self.b)[None]
            self._dinv = np.linalg.inv(D)
        return self._dinv

    def precondition(self, X, V):  # QuadraticProblem.cpp:56-69
        if self.precond == "exact":
            if self._lu is None:
                P = (self.Qs + self.shift * sp.identity(self.N, format="csr")).tocsc()
                self._lu = spla.splu(P)
            Z = self._lu.solve(V.reshape(self.N, self.r)).reshape(V.shape)
        elif self.precond == "jacobi":
            # z_i (r x b) = v_i (r x b) Dinv_i ; in the [b, r] view: Z_i = Dinv_i^T V_i = Dinv_i V_i
            Z = self.dinv_blocks() @ V
        elif self.precond == "none":
            Z = V.copy()
        elif self.precond in ("amg", "amg2", "amg_additive"):
            Z = self.amg_cycle(V)
        else:
            raise ValueError(self.precond)
        return tangent_project(X, Z, self.d)

    # --- aggregation multigrid V(1,1) cycle (device option "multilevel") ---
    def amg_setup(self):
        """Hierarchy for A_0 = Q + shift I: A_{l+1} = P_l^T A_l P_l (Galerkin), damped block-Jacobi smoother on every
        level (the diagonal blocks of A_l), dense inverse of the coarsest operator.  All fp64, except that the finished
        inverse is rounded to fp32 values when `amg_coarse_bits` = 32 (the device's opt-in storage mode; default 64)."""
        if self._amg is None:
            ks = self.amg_k
            if ks is None:
                ks = amg_default_ks(self.n, self.b)
            elif isinstance(ks, (int, np.integer)):
                ks = [int(ks)]
            ks = [int(k) for k in ks]
            # two levels, graph aggregates of at most -ks[0] poses; [-S, -cap]: fragments merged up to cap poses
            merged = len(ks) == 2 and ks[0] < 0 and ks[1] < 0
            merge_cap = -ks[1] if merged else int(self.amg_merge)
            spec = list(ks) if merged or not merge_cap else [ks[0], -merge_cap]  # (the form the device reports)
            if merged:
                ks = ks[:1]
            graph = len(ks) == 1 and ks[0] < 0
            if graph:
                lab, ptr, mem, parent, pslot = amg_graph_aggregates(self.Q, -ks[0])
                if merge_cap:
                    lab, ptr, mem, parent, pslot = amg_merge_small_aggregates(self.Q, -ks[0], lab, ptr, mem, merge_cap)
                Pbs = [amg_tree_prolongation(self.Q, self.d, mem, parent, pslot)]
            else:
                Pbs = amg_chain_prolongations(self.Q, self.d, ks)
            b = self.b
            A = (self.Qs + self.shift * sp.identity(self.N, format="csr")).tocsr()
            levels, cur = [], self.n
            for k, Pb in zip(ks, Pbs):
                agg = lab if graph else np.arange(cur) // k
                nc = int(agg.max()) + 1 if graph else (cur + k - 1) // k
                rows = (np.arange(cur)[:, None, None] * b + np.arange(b)[None, :, None]) + np.zeros((1, 1, b), dtype=np.int64)
                cols = (agg[:, None, None] * b + np.arange(b)[None, None, :]) + np.zeros((1, b, 1), dtype=np.int64)
                P = sp.csr_matrix((Pb.ravel(), (rows.ravel(), cols.ravel())), shape=(cur * b, nc * b))
                Ab = A.tobsr(blocksize=(b, b))
                Ab.sort_indices()
                rr = np.repeat(np.arange(cur), np.diff(Ab.indptr))
                D = np.zeros((cur, b, b))
                D[rr[rr == Ab.indices]] = Ab.data[rr == Ab.indices]
                levels.append(dict(k=k, n=cur, A=A, P=P, Pb=Pb, Dinv=np.linalg.inv(D)))
                A = (P.T @ A @ P).tocsr()
                cur = nc
            Ac = A.toarray()
            Ac = 0.5 * (Ac + Ac.T)
            AcInv = np.linalg.inv(Ac)
            if self.amg_coarse_bits == 32:  # the device STORES the inverse in fp32 (products stay fp64)
                AcInv = AcInv.astype(np.float32).astype(np.float64)
            self._amg = dict(ks=spec, levels=levels, Ac=Ac, AcInv=AcInv, nc=cur)
        return self._amg

    def amg_cycle(self, V):
        m = self.amg_setup()
        w, b, r = self.amg_omega, self.b, self.r
        if self.amg_operator_bits == 32 and len(m["levels"]) == 1 and not getattr(self, "amg_additive", False):
            # The device's two-level cycle in the form its kernels evaluate it (k_tcg_update, k_ml_restrict, k_ml_post_ap):
            #   x1 = w Dinv r;  res1 = r - Q x1 - shift x1;  rc = P^T res1;  xc = Ac^-1 rc;
            #   z = (x1 + P xc) + w Dinv (res1 - (A P) xc)
            # with Q, P and A P read from fp32 COPIES (each rounded from its fp64 original; A P is the fp64 product
            # rounded, not the product of the rounded factors) and the two vectors that live inside the cycle STORED in
            # fp32: the x1 the restriction reads (own rows and gathers) is the rounded one -- the post-smoothing
            # recomputes x1 from r in fp64 --, the res1 the post-smoothing reads is the rounded one -- P^T res1 uses the
            # fp64 value the restriction still holds.  Everything else -- Dinv, the dense level, every product and sum --
            # in fp64.  With fp64 storage this is the generic cycle below up to summation order.
            L = m["levels"][0]
            if "ops32" not in m:
                f32 = lambda M: M.astype(np.float32).astype(np.float64)  # noqa: E731
                Q32 = self.Qs.copy()
                Q32.data = f32(Q32.data)
                P32 = L["P"].copy()
                P32.data = f32(P32.data)
                AP32 = (L["A"] @ L["P"]).tocsr()
                AP32.data = f32(AP32.data)
                m["ops32"] = (Q32, P32, AP32)
            Q32, P32, AP32 = m["ops32"]
            smooth = lambda res: (L["Dinv"] @ res.reshape(L["n"], b, r)).reshape(res.shape)  # noqa: E731
            rhs = V.reshape(self.N, self.r)
            f32 = (lambda M: M.astype(np.float32).astype(np.float64)) if self.amg_vector_bits == 32 else (lambda M: M)  # noqa: E731
            x1 = w * smooth(rhs)
            x1s = f32(x1)
            res1 = rhs - Q32 @ x1s - self.shift * x1s
            rc = P32.T @ res1
            if self.amg_coarse_bits == 32:
                rc = rc.astype(np.float32).astype(np.float64)
            xc = m["AcInv"] @ rc
            z = (x1 + P32 @ xc) + w * smooth(f32(res1) - AP32 @ xc)
            return z.reshape(V.shape)

        def cycle(lv, rhs):
            if lv == len(m["levels"]):
                if self.amg_coarse_bits == 32:  # the dense level reads its right-hand side in its storage precision
                    rhs = rhs.astype(np.float32).astype(np.float64)
                return m["AcInv"] @ rhs
            L = m["levels"][lv]
            smooth = lambda res: (L["Dinv"] @ res.reshape(L["n"], b, r)).reshape(res.shape)  # noqa: E731
            if getattr(self, "amg_additive", False):  # additive combination (precond = "amg_additive")
                return getattr(self, "amg_add_w", w) * smooth(rhs) + L["P"] @ cycle(lv + 1, L["P"].T @ rhs)
            x = w * smooth(rhs)
            for _ in range((self.amg_nu if lv > 0 else 1) - 1):
                x = x + w * smooth(rhs - L["A"] @ x)
            for g in range(self.amg_gamma if (lv > 0 and lv + 1 < len(m["levels"])) else 1):
                x = x + L["P"] @ cycle(lv + 1, L["P"].T @ (rhs - L["A"] @ x))
            for _ in range(self.amg_nu if lv > 0 else 1):
                x = x + w * smooth(rhs - L["A"] @ x)
            return x

        return cycle(0, V.reshape(self.N, self.r)).reshape(V.shape)


# --------------------------------------------------------------------------
# QuadraticOptimizer -- src/QuadraticOptimizer.cpp  (+ ROPTLIB RTRNewton / tCG_TR)
# --------------------------------------------------------------------------

TCG_NEGCURV, TCG_EXCREGION, TCG_LCON, TCG_SCON, TCG_MAXITER = 0, 1, 2, 3, 4
TCG_NAMES = ["NEGCURVTURE", "EXCREGION", "LCON", "SCON", "MAXITER"]


@dataclass
class ROptParameters:  # include/DPGO/DPGO_types.h:44-86
    method: str = "RTR"
    verbose: bool = False
    gradnorm_tol: float = 1e-2
    RGD_stepsize: float = 1e-3
    RGD_use_preconditioner: bool = True
    RTR_iterations: int = 3
    RTR_tCG_iterations: int = 50
    RTR_initial_radius: float = 100.0


@dataclass
class ROPTResult:  # include/DPGO/DPGO_types.h:91-107
    success: bool = False
    fInit: float = 0.0
    gradNormInit: float = 0.0
    fOpt: float = 0.0
    gradNormOpt: float = 0.0
    elapsedMs: float = 0.0
    tCGStatus: int = TCG_MAXITER
    # extras (not in the reference struct): bookkeeping used by parity tests
    tcg_iters: int = 0
    outer_iters: int = 0
    trace: list = field(default_factory=list)


def dot(A, B):
    """ROPTLIB ProductManifold::Metric with Euclidean-metric factors = plain dot over all
    r(d+1)n entries."""
    return float(np.sum(A * B))


def tcg(problem: QuadraticProblem, X, g, S, Delta, max_inner, theta=1.0, kappa=0.1, min_inner=0, trace=None,
        hess_recurrence=False):
    """ROPTLIB SolversTR::tCG_TR restated, eta0 = 0 (useRand = false).
    Returns (eta, status, inner_iters, n_hess).

    hess_recurrence = False is the reference's arithmetic (H applied to delta every iteration).
    hess_recurrence = True is what the MI355X path computes: H is applied to the preconditioned
    residual z and H delta follows the direction recurrence, H delta' = beta * H delta - H z (exact in
    exact arithmetic because H is linear on the tangent space)."""
    r = g.copy()
    e_Pe = 0.0
    r_r = dot(r, r)
    norm_r0 = math.sqrt(r_r)
    z = problem.precondition(X, r)
    z_r = dot(z, r)
    d_Pd = z_r
    delta = -z
    e_Pd = 0.0
    eta = np.zeros_like(g)
    status = TCG_MAXITER
    j = 0
    n_hess = 0
    beta = 0.0
    Hd = None
    while j < max_inner:
        if hess_recurrence:
            Hz = problem.rie_hess(X, S, z)
            Hd = -Hz if j == 0 else beta * Hd - Hz
        else:
            Hd = problem.rie_hess(X, S, delta)
        n_hess += 1
        d_Hd = dot(delta, Hd)
        alpha = z_r / d_Hd if d_Hd != 0 else math.inf
        e_Pe_new = e_Pe + 2.0 * alpha * e_Pd + alpha * alpha * d_Pd
        if d_Hd <= 0 or e_Pe_new >= Delta * Delta:
            tau = (-e_Pd + math.sqrt(e_Pd * e_Pd + d_Pd * (Delta * Delta - e_Pe))) / d_Pd
            eta = eta + tau * delta
            status = TCG_NEGCURV if d_Hd < 0 else TCG_EXCREGION
            if trace is not None:
                trace.append(dict(j=j, d_Hd=d_Hd, alpha=alpha, tau=tau, status=status))
            break
        e_Pe = e_Pe_new
        eta = eta + alpha * delta
        r = r + alpha * Hd
        r_r = dot(r, r)
        norm_r = math.sqrt(r_r)
        if trace is not None:
            trace.append(dict(j=j, d_Hd=d_Hd, alpha=alpha, norm_r=norm_r))
        if j >= min_inner and norm_r <= norm_r0 * min(norm_r0 ** theta, kappa):
            status = TCG_LCON if kappa < norm_r0 ** theta else TCG_SCON
            break
        z = problem.precondition(X, r)
        zold_rold = z_r
        z_r = dot(z, r)
        beta = z_r / zold_rold
        delta = beta * delta - z
        e_Pd = beta * (e_Pd + alpha * d_Pd)
        d_Pd = z_r + beta * beta * d_Pd
        j += 1
    return eta, status, j, n_hess


class QuadraticOptimizer:
    """src/QuadraticOptimizer.cpp.  `accept_tiny_decrease` is SURVEY 8c' item 5 (newer
    ROPTLIB accepts a step whose relative decrease is positive but below sqrt(eps))."""

    def __init__(self, problem: QuadraticProblem, params: Optional[ROptParameters] = None,
                 accept_tiny_decrease: bool = True, hess_recurrence: bool = False):
        self.problem = problem
        self.params = params or ROptParameters()
        self.result = ROPTResult()
        self.accept_tiny_decrease = accept_tiny_decrease
        self.hess_recurrence = hess_recurrence

    def optimize(self, Y):  # QuadraticOptimizer.cpp:26-48
        import time
        p = self.problem
        self.result = ROPTResult()
        self.result.fInit = p.f(Y)
        self.result.gradNormInit = p.rie_grad_norm(Y)
        t0 = time.perf_counter()
        if self.params.method == "RTR":
            Yopt = self.trust_region(Y)
        else:
            Yopt = self.gradient_descent(Y)
        self.result.elapsedMs = 1e3 * (time.perf_counter() - t0)
        self.result.fOpt = p.f(Yopt)
        self.result.gradNormOpt = p.rie_grad_norm(Yopt)
        self.result.success = True
        return Yopt

    def _run_rtr(self, x1, Delta0, Delta_max, max_iter):
        """ROPTLIB SolversTR::Run restated (SURVEY 8a row a8 / 8c' item 4)."""
        p, d = self.problem, self.problem.d
        prm = self.params
        sqeps = math.sqrt(np.finfo(float).eps)
        EG = p.euc_grad(x1)
        f1 = 0.5 * np.sum((EG - p.G) * x1) + np.sum(x1 * p.G)
        S = p.sym_ytg(x1, EG)
        g1 = tangent_project(x1, EG, d)
        ngf = math.sqrt(dot(g1, g1))
        Delta = Delta0
        it = 0
        accepted_last = False
        status = TCG_MAXITER
        isstop = ngf < prm.gradnorm_tol
        while not isstop and it < max_iter:
            tr = [] if prm.verbose else None
            eta, status, inner, n_hess = tcg(p, x1, g1, S, Delta, prm.RTR_tCG_iterations, trace=tr,
                                             hess_recurrence=self.hess_recurrence)
            self.result.tcg_iters += n_hess
            x2 = qf_retract(x1, eta, d)
            f2 = p.f(x2)
            Heta = p.rie_hess(x1, S, eta)
            rho = (f1 - f2) / (-dot(eta, g1 + 0.5 * Heta))
            if rho > 0.75:
                if status in (TCG_EXCREGION, TCG_NEGCURV):
                    Delta *= 2.0
                if Delta > Delta_max:
                    Delta = Delta_max
            elif rho < 0.25:
                Delta *= 0.25
            accept = rho > 0.1 or (self.accept_tiny_decrease and
                                   abs(f1 - f2) / (abs(f1) + 1) < sqeps and f2 < f1)
            self.result.trace.append(dict(it=it, f1=f1, f2=f2, rho=rho, Delta=Delta, inner=inner,
                                          status=status, accept=accept, ngf=ngf))
            if accept:
                EG = p.euc_grad(x2)
                S = p.sym_ytg(x2, EG)
                g1 = tangent_project(x2, EG, d)
                ngf = math.sqrt(dot(g1, g1))
                x1, f1 = x2, f2
                isstop = ngf < prm.gradnorm_tol
            accepted_last = accept
            it += 1
        self.result.outer_iters += it
        self.result.tCGStatus = status
        return x1, accepted_last

    def trust_region(self, Yinit):  # QuadraticOptimizer.cpp:50-108
        prm = self.params
        gn0 = self.problem.rie_grad_norm(Yinit)
        if gn0 < prm.gradnorm_tol:  # :57-59
            return Yinit
        if prm.RTR_iterations == 1:  # :80-99
            radius = prm.RTR_initial_radius
            total = 0
            while True:
                x, accepted = self._run_rtr(Yinit.copy(), radius, radius, 1)
                if accepted:
                    return x
                if total > 10:
                    return Yinit
                radius /= 4
                total += 1
        x, _ = self._run_rtr(Yinit.copy(), prm.RTR_initial_radius, 5 * prm.RTR_initial_radius,
                             prm.RTR_iterations)  # :68-69,100
        return x

    def gradient_descent(self, Yinit):  # QuadraticOptimizer.cpp:110-137
        p, prm = self.problem, self.params
        g = p.rie_grad(Yinit)
        if prm.RGD_use_preconditioner:
            g = p.precondition(Yinit, g)
        return qf_retract(Yinit, -prm.RGD_stepsize * g, p.d)


# --------------------------------------------------------------------------
# Initialisation helpers (host-side, "next" rows of SURVEY 8f; needed by the harness)
# --------------------------------------------------------------------------


def lift(T, r):
    """X = YLift * T with YLift = [I_d; 0] (gauge note, SURVEY 8c: cost, gradnorm and the
    whole RTR trajectory are invariant under X -> O X)."""
    n, b, d = T.shape
    X = np.zeros((n, b, r))
    X[:, :, :d] = T
    return X


def odometry_initialization(odometry: Measurements, n: int):
    """src/DPGO_solver.cpp:271-303.  Returns T in the [n, d+1, d] view."""
    d = odometry.d
    T = np.zeros((n, d + 1, d))
    T[0, :d, :] = np.eye(d)
    order = {int(odometry.p1[e]): e for e in range(odometry.m)}
    for dst in range(1, n):
        e = order[dst - 1]
        assert odometry.p2[e] == dst
        Rsrc = T[dst - 1, :d, :].T
        tsrc = T[dst - 1, d, :]
        T[dst, :d, :] = (Rsrc @ odometry.R[e]).T
        T[dst, d, :] = tsrc + Rsrc @ odometry.t[e]
    return T


def chordal_initialization(meas: Measurements, n: int):
    """src/DPGO_solver.cpp:220-269 with constructBMatrices / recoverTranslations
    (src/DPGO_utils.cpp:346-462).  The reference solves the two least-squares problems by
    SPQR; here by sparse normal equations (same minimiser; full column rank once pose 0 is
    pinned)."""
    d, m = meas.d, meas.m
    d2 = d * d
    rows, cols, vals = [], [], []
    sk = np.sqrt(meas.kappa)
    # B3: rows e*d2 + d*r + l ; cols i*d2 + d*c + l ; val -sqrt(kappa) R(c, r)   (:417-433)
    for r_ in range(d):
        for c in range(d):
            for l in range(d):
                rows.append(np.arange(m) * d2 + d * r_ + l)
                cols.append(meas.p1 * d2 + d * c + l)
                vals.append(-sk * meas.R[:, c, r_])
    for l in range(d2):
        rows.append(np.arange(m) * d2 + l)
        cols.append(meas.p2 * d2 + l)
        vals.append(sk)
    B3 = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(m * d2, n * d2))
    Id_vec = np.eye(d).reshape(-1)
    cR = B3[:, :d2] @ Id_vec
    B3red = B3[:, d2:].tocsc()
    A = (B3red.T @ B3red).tocsc()
    rvec = -spla.splu(A).solve(B3red.T @ cR)
    Rch = np.zeros((n, d, d))
    Rch[0] = np.eye(d)
    # column-major d x d blocks: rvec[(i-1)*d2 + d*c + l] = R_i(l, c)
    Rm = rvec.reshape(n - 1, d, d)  # [i, c, l]
    for i in range(1, n):
        Rch[i] = project_to_rotation_group(Rm[i - 1].T)
    # translations: B1 (:367-389), B2 (:394-407)
    st = np.sqrt(meas.tau)
    rows, cols, vals = [], [], []
    for l in range(d):
        rows += [np.arange(m) * d + l, np.arange(m) * d + l]
        cols += [meas.p1 * d + l, meas.p2 * d + l]
        vals += [-st, st]
    B1 = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(m * d, n * d))
    rows, cols, vals = [], [], []
    for k in range(d):
        for r_ in range(d):
            rows.append(np.arange(m) * d + r_)
            cols.append(meas.p1 * d2 + d * k + r_)
            vals.append(-st * meas.t[:, k])
    B2 = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(m * d, n * d2))
    rv = np.swapaxes(Rch, 1, 2).reshape(-1)  # column-major vec of each R_i
    c = B2 @ rv
    B1red = B1[:, d:].tocsc()
    A = (B1red.T @ B1red).tocsc()
    tred = -spla.splu(A).solve(B1red.T @ c)
    t = np.zeros((n, d))
    t[1:] = tred.reshape(n - 1, d)
    T = np.zeros((n, d + 1, d))
    T[:, :d, :] = np.swapaxes(Rch, 1, 2)
    T[:, d, :] = t
    return T


def measurement_error(meas: Measurements, X):
    """computeMeasurementError (src/DPGO_utils.cpp:501-507) for every edge of a single-agent
    graph; X in the [n, d+1, r] view."""
    d = meas.d
    Y1 = np.swapaxes(X[meas.p1, :d, :], 1, 2)  # [m, r, d]
    Y2 = np.swapaxes(X[meas.p2, :d, :], 1, 2)
    t1, t2 = X[meas.p1, d, :], X[meas.p2, d, :]
    rot = np.sum((Y1 @ meas.R - Y2) ** 2, axis=(1, 2))
    tr = np.sum((t2 - t1 - (Y1 @ meas.t[:, :, None])[:, :, 0]) ** 2, axis=1)
    return meas.kappa * rot + meas.tau * tr


# --------------------------------------------------------------------------
# Synthetic 3-D grid (config C4 of SURVEY 8d)
# --------------------------------------------------------------------------


def _rotvec_to_R(v):
    th = np.linalg.norm(v, axis=1)
    k = v / np.maximum(th, 1e-300)[:, None]
    K = np.zeros((len(v), 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -k[:, 2], k[:, 1]
    K[:, 1, 0], K[:, 1, 2] = k[:, 2], -k[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -k[:, 1], k[:, 0]
    s, c = np.sin(th)[:, None, None], np.cos(th)[:, None, None]
    return np.eye(3)[None] + s * K + (1 - c) * (K @ K)


def synthetic_grid(nx: int, ny: int, nz: int, seed: int = 0, sigma_t: float = 0.1, sigma_r: float = 0.2):
    """Config C4: nx x ny x nz lattice, boustrophedon odometry path with x fastest,
    then y, then z; a loop closure on every remaining lattice-adjacent pair; ground-truth
    rotations uniform random; noise: translation N(0, sigma_t^2 I), rotation = axis-angle
    N(0, sigma_r^2 I); information 1/sigma^2 I  => tau = 1/sigma_t^2, kappa = 1/(2 sigma_r^2)
    by the g2o formulas (src/DPGO_utils.cpp:223,230).  RNG: numpy PCG64(seed).
    Returns (measurements, n, T_true[n, 4, 3])."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = nx * ny * nz
    idx = np.empty((nx, ny, nz), dtype=np.int64)
    k = 0
    for z in range(nz):
        ys = range(ny) if z % 2 == 0 else range(ny - 1, -1, -1)
        for yi, y in enumerate(ys):
            fwd = ((yi + z * ny) % 2 == 0)
            xs = range(nx) if fwd else range(nx - 1, -1, -1)
            for x in xs:
                idx[x, y, z] = k
                k += 1
    pos = np.zeros((n, 3))
    gx, gy, gz = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    pos[idx.reshape(-1)] = np.stack([gx.reshape(-1), gy.reshape(-1), gz.reshape(-1)], axis=1)
    q = rng.standard_normal((n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    Rt = np.stack([quat_to_rot_unnormalised(*qq) for qq in q]) if n <= 4096 else _quat_batch(q)
    pairs = []
    a = idx[:-1, :, :].reshape(-1); b_ = idx[1:, :, :].reshape(-1); pairs.append(np.stack([a, b_], 1))
    a = idx[:, :-1, :].reshape(-1); b_ = idx[:, 1:, :].reshape(-1); pairs.append(np.stack([a, b_], 1))
    a = idx[:, :, :-1].reshape(-1); b_ = idx[:, :, 1:].reshape(-1); pairs.append(np.stack([a, b_], 1))
    pairs = np.concatenate(pairs, 0)
    lo = np.minimum(pairs[:, 0], pairs[:, 1]); hi = np.maximum(pairs[:, 0], pairs[:, 1])
    order = np.lexsort((hi, lo))
    lo, hi = lo[order], hi[order]
    m = len(lo)
    Ri, Rj = Rt[lo], Rt[hi]
    Rij = np.swapaxes(Ri, 1, 2) @ Rj
    tij = (np.swapaxes(Ri, 1, 2) @ (pos[hi] - pos[lo])[:, :, None])[:, :, 0]
    Rn = _rotvec_to_R(sigma_r * rng.standard_normal((m, 3)))
    Rmeas = Rij @ Rn
    tmeas = tij + sigma_t * rng.standard_normal((m, 3))
    z64 = np.zeros(m, dtype=np.int64)
    meas = Measurements(3, z64, lo, z64.copy(), hi, Rmeas, tmeas,
                        np.full(m, 1.0 / (2.0 * sigma_r ** 2)), np.full(m, 1.0 / sigma_t ** 2),
                        np.ones(m), lo + 1 == hi)
    Ttrue = np.zeros((n, 4, 3))
    Ttrue[:, :3, :] = np.swapaxes(Rt, 1, 2)
    Ttrue[:, 3, :] = pos
    return meas, n, Ttrue


def _quat_batch(q):
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((len(q), 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def perturbed_truth(Ttrue, seed: int = 2, sigma_t: float = 0.1, sigma_r: float = 0.2):
    """Initial guess for C4: ground truth perturbed by the measurement-noise model."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = Ttrue.shape[0]
    R = np.swapaxes(Ttrue[:, :3, :], 1, 2) @ _rotvec_to_R(sigma_r * rng.standard_normal((n, 3)))
    T = Ttrue.copy()
    T[:, :3, :] = np.swapaxes(R, 1, 2)
    T[:, 3, :] += sigma_t * rng.standard_normal((n, 3))
    return T


# --------------------------------------------------------------------------
# Multi-agent RBCD driver (checker for dpgo_amd/agent.py)
# --------------------------------------------------------------------------


def rbcd_coloured(meas: Measurements, n: int, num_robots: int, r: int, X0, sweeps: int,
                  params: Optional[ROptParameters] = None, precond: str = "jacobi", hess_recurrence: bool = False,
                  amg_k=None, inactive=(), schedule=None, counts=None):
    """Two-colour (greedy-coloured) parallel RBCD of SURVEY 8e on the contiguous partition of
    examples/MultiRobotExample.cpp:71-119: in every sweep each colour class updates once; an agent's
    update is PGOAgent::updateX (src/PGOAgent.cpp:938-995): G from the neighbours' current public poses
    (constructG), then QuadraticOptimizer::optimize from its current block.  Agents of one colour are not
    adjacent, so updating them one after the other equals updating them simultaneously.
    amg_k: hierarchy of the multilevel / additive preconditioners (None: the default; a dict agent -> sizes: per agent).
    inactive: robots switched off by PGOAgent::setRobotActive(id, false) (src/PGOAgent.cpp:1173-1184): they do not
    update, and their neighbours' data matrices leave the shared edges with them out (active_shared_edges); the
    colouring is that of the full team.
    schedule: {(agent, sweep): preconditioner name} -- solves whose preconditioner differs from `precond` (the device's
    default selection switches a coupled block from block-Jacobi to the additive form in the middle of a run; the test
    tells the oracle which solve ran what, ROPTResult::precond_used).  counts: a list that receives (sweep, agent,
    Hessian-vector products) of every solve.
    Returns (X, [central 2f after each sweep], [central gradnorm after each sweep])."""
    d = meas.d
    ranges, per = partition_contiguous(meas, n, num_robots)
    agents = []
    for a in range(num_robots):
        s, e = ranges[a]
        priv = Measurements.concat([per[a]["odometry"], per[a]["private"]])
        full = per[a]["shared"]
        sh = active_shared_edges(full, a, set(inactive))
        Qa = construct_Q(e - s, d, priv, sh, my_id=a)
        need = set()
        for k in range(sh.m):
            need.add((int(sh.r2[k]), int(sh.p2[k])) if sh.r1[k] == a else (int(sh.r1[k]), int(sh.p1[k])))
        team_adj = sorted({int(full.r2[k]) if full.r1[k] == a else int(full.r1[k]) for k in range(full.m)})
        # one problem object per agent for the whole run (Q and the preconditioner belong to the PoseGraph's lifetime,
        # include/DPGO/PoseGraph.h:324-331); only G changes between solves
        agents.append(dict(Q=Qa, shared=sh, need=sorted(need), adj=team_adj,
                           prob=QuadraticProblem(Qa, None, r, d, precond=precond,
                                                 amg_k=amg_k[a] if isinstance(amg_k, dict) else amg_k)))
    colour = [-1] * num_robots
    for a in range(num_robots):
        used = {colour[q] for q in agents[a]["adj"] if colour[q] >= 0}
        c = 0
        while c in used:
            c += 1
        colour[a] = c
    central = QuadraticProblem(construct_Q(n, d, meas), None, r, d)
    X = X0.copy()
    costs, gns = [], []
    for sweep in range(sweeps):
        for c in range(max(colour) + 1):
            for a in range(num_robots):
                if colour[a] != c or a in inactive:
                    continue
                s, e = ranges[a]
                nbr = {(rob, fr): X[ranges[rob][0] + fr] for rob, fr in agents[a]["need"]}
                prob = agents[a]["prob"]
                want = (schedule or {}).get((a, sweep), precond)
                if want != precond:  # one problem object per (agent, preconditioner): its hierarchy is built once
                    alt = agents[a].setdefault("alt", {})
                    if want not in alt:
                        alt[want] = QuadraticProblem(agents[a]["Q"], None, r, d, precond=want,
                                                     amg_k=amg_k[a] if isinstance(amg_k, dict) else amg_k)
                    prob = alt[want]
                prob.G = construct_G(e - s, d, r, agents[a]["shared"], a, nbr)
                opt = QuadraticOptimizer(prob, params or ROptParameters(), hess_recurrence=hess_recurrence)
                X[s:e] = opt.optimize(X[s:e])
                if counts is not None:
                    counts.append((sweep, a, opt.result.tcg_iters))
        costs.append(2 * central.f(X))
        gns.append(central.rie_grad_norm(X))
    return X, costs, gns


class AutoCostRule:
    """Restatement of the device's default preconditioner selection (precond = "auto", include/dpgo_hip.h
    DPGO_PRECOND_AUTO; dpgo_amd/csrc: auto_update) for a COUPLED block the additive one-launch solve can hold -- the
    device's stand-in for "the factor of Q + 0.1 I is there whenever PreConditioner is called"
    (src/QuadraticProblem.cpp:56-69, src/PoseGraph.cpp:582-613).  Units: a tenth of a block-Jacobi product of a solve that
    has the device to itself (units_jacobi_alone, in which the set-up is paid back); units_jacobi / units_additive are what
    a product of either kind is charged on this handle (the same when it is solved alone; scaled by the part of the chip a
    launch blocks when it shares the device).  next() = what the next solve runs; record(products) = the solve just
    finished."""

    def __init__(self, budget=150, units_jacobi=10, units_additive=13, setup_units=2800, min_products=6,
                 units_jacobi_alone=10):
        self.budget, self.uj, self.ua, self.setup, self.minp = budget, units_jacobi, units_additive, setup_units, min_products
        self.uj0 = units_jacobi_alone
        self.ml, self.state, self.units, self.ref, self.backoff, self.switches = False, 0, 0, 0, 0, 0

    def next(self) -> str:
        return "additive" if self.ml else "jacobi"

    def record(self, products: int) -> None:
        if not self.ml:
            self.state = 0
            self.units += self.uj0 * products
            binds = 2 * products >= self.budget
            paid = products >= self.minp and self.units >= (self.setup << self.backoff)
            if binds or paid:
                if not binds and self.ua * self.minp >= self.uj * products:  # a trial that cannot win is not run
                    self.units, self.backoff = 0, min(self.backoff + 1, 6)
                    return
                self.ml, self.state, self.ref = True, 1, products
                self.switches += 1
            return
        if self.ua * products * 100 < self.uj * self.ref * (115 if self.state == 2 else 100):
            self.state = 2
        else:
            self.ml, self.state, self.units = False, 0, 0
            self.backoff = min(self.backoff + 1, 6)


# --------------------------------------------------------------------------
# The reference demo's schedule: greedy block selection + Nesterov acceleration with restarts
#   examples/MultiRobotExample.cpp:170-255, src/PGOAgent.cpp:376-432 (iterate), :880-995
# --------------------------------------------------------------------------


class OracleAgent:
    """The part of PGOAgent the demo exercises: X, XPrev, Y, V, gamma, alpha, iteration counter, the
    neighbour pose caches (plain and auxiliary) and iterate(doOptimization)."""

    def __init__(self, agent_id, num_robots, Q, shared, r, d, X0, precond, params, acceleration=True,
                 restart_interval=30, hess_recurrence=False):
        self.id, self.N, self.Q, self.shared, self.r, self.d = agent_id, num_robots, Q, shared, r, d
        self.precond, self.params, self.acc = precond, params or ROptParameters(), acceleration
        self.restart_interval = restart_interval  # PGOAgentParameters::restartInterval (PGOAgent.h:118)
        self.hess_recurrence = hess_recurrence
        self.X = X0.copy()
        self.n = X0.shape[0]
        need = set()
        for k in range(shared.m):
            need.add((int(shared.r2[k]), int(shared.p2[k])) if shared.r1[k] == agent_id else
                     (int(shared.r1[k]), int(shared.p1[k])))
        self.need = sorted(need)
        self.neighbors = sorted({rob for rob, _ in self.need})
        self.nbr, self.nbr_aux = {}, {}
        self.iteration = 0
        self.tcg_total = 0
        # initializeAcceleration (PGOAgent.cpp:899-908)
        self.XPrev, self.V, self.Y = self.X.copy(), self.X.copy(), self.X.copy()
        self.gamma = self.alpha = 0.0

    def _update_x(self, do_opt, acceleration):  # PGOAgent::updateX (:938-995)
        if not do_opt:
            if acceleration:
                self.X = self.Y.copy()
            return
        nbr = self.nbr_aux if acceleration else self.nbr
        G = construct_G(self.n, self.d, self.r, self.shared, self.id, nbr)
        prob = QuadraticProblem(self.Q, G, self.r, self.d, precond=self.precond)
        opt = QuadraticOptimizer(prob, self.params, hess_recurrence=self.hess_recurrence)
        X0 = self.Y if acceleration else self.X
        self.X = opt.optimize(X0.copy())
        self.tcg_total += opt.result.tcg_iters

    def iterate(self, do_opt):  # PGOAgent::iterate (:376-432)
        self.iteration += 1
        self.XPrev = self.X.copy()
        if self.acc:
            N = self.N
            self.gamma = (1 + math.sqrt(1 + 4 * N ** 2 * self.gamma ** 2)) / (2 * N)  # updateGamma (:910-914)
            self.alpha = 1 / (self.gamma * N)  # updateAlpha (:916-920)
            self.Y = polar_project((1 - self.alpha) * self.X + self.alpha * self.V, self.d)  # updateY (:922-928)
            self._update_x(do_opt, True)
            self.V = polar_project(self.V + self.gamma * (self.X - self.Y), self.d)  # updateV (:930-936)
            if (self.iteration + 1) % self.restart_interval == 0:  # shouldRestart (:880-885)
                self.X = self.XPrev.copy()  # restartNesterovAcceleration (:887-897)
                self._update_x(do_opt, False)
                self.V, self.Y = self.X.copy(), self.X.copy()
                self.gamma = self.alpha = 0.0
        else:
            self._update_x(do_opt, False)


def multi_robot_example(meas: Measurements, n: int, num_robots: int, r: int, X0, max_iters: int = 1000,
                        acceleration: bool = True, precond: str = "exact", params: Optional[ROptParameters] = None,
                        gradnorm_stop: float = 0.1, hess_recurrence: bool = False):
    """examples/MultiRobotExample.cpp:170-255: per iteration every non-selected agent iterate(false), the selected
    agent pulls the others' public poses (and aux poses when accelerated) and iterate(true); central cost 2f and
    gradnorm are evaluated; stop when gradnorm < 0.1; next agent = argmax block gradnorm.
    Returns dict(X, iterations, cost (2f), gradnorm, tcg_total, selected (list), trace)."""
    d = meas.d
    ranges, per = partition_contiguous(meas, n, num_robots)
    agents = []
    for a in range(num_robots):
        s, e = ranges[a]
        priv = Measurements.concat([per[a]["odometry"], per[a]["private"]])
        Qa = construct_Q(e - s, d, priv, per[a]["shared"], my_id=a)
        agents.append(OracleAgent(a, num_robots, Qa, per[a]["shared"], r, d, X0[s:e], precond, params,
                                  acceleration=acceleration, hess_recurrence=hess_recurrence))
    central = QuadraticProblem(construct_Q(n, d, meas), None, r, d)
    selected, order, trace = 0, [], []
    X = X0.copy()
    for it in range(max_iters):
        sel = agents[selected]
        for ag in agents:
            if ag.id != selected:
                ag.iterate(False)
        for rob, fr in sel.need:  # :183-204
            sel.nbr[(rob, fr)] = agents[rob].X[fr].copy()
            if acceleration:
                sel.nbr_aux[(rob, fr)] = agents[rob].Y[fr].copy()
        sel.iterate(True)
        for ag in agents:
            s, e = ranges[ag.id]
            X[s:e] = ag.X
        RG = central.rie_grad(X)
        gn = float(np.linalg.norm(RG))
        cost = 2 * central.f(X)
        order.append(selected)
        trace.append((cost, gn))
        if gn < gradnorm_stop:  # :229
            break
        if sel.neighbors:  # :234-247
            selected = int(np.argmax([np.linalg.norm(RG[s:e]) for s, e in ranges]))
    return dict(X=X, iterations=len(order), cost=trace[-1][0], gradnorm=trace[-1][1],
                tcg_total=sum(a.tcg_total for a in agents), selected=order, trace=trace)


# --------------------------------------------------------------------------
# Robust PGO (GNC-TLS) -- src/DPGO_robust.cpp:54-134, src/DPGO_solver.cpp:335-412
# --------------------------------------------------------------------------


def chi2inv(quantile: float, dof: int) -> float:
    """chi2inv (src/DPGO_utils.cpp:509-512: boost::math::quantile of the chi-squared distribution; "equivalent to chi2inv
    in Matlab", include/DPGO/DPGO_utils.h:146-153)."""
    from scipy.stats import chi2
    return float(chi2.ppf(quantile, dof))


def error_threshold_at_quantile(quantile: float, dimension: int) -> float:
    """RobustCost::computeErrorThresholdAtQuantile (include/DPGO/DPGO_robust.h:116-123)."""
    assert dimension == 3 and quantile > 0
    return math.sqrt(chi2inv(quantile, 6)) if quantile < 1 else 1e5


def gnc_tls_weight(r, mu, barc):
    """RobustCost::weight for GNC_TLS (src/DPGO_robust.cpp:80-92), vectorised."""
    r = np.asarray(r, dtype=np.float64)
    rSq, bSq = r * r, barc * barc
    upper, lower = (mu + 1) / mu * bSq, mu / (mu + 1) * bSq
    with np.errstate(divide="ignore", invalid="ignore"):
        mid = np.sqrt(bSq * mu * (mu + 1) / rSq) - mu
    return np.where(rSq >= upper, 0.0, np.where(rSq <= lower, 1.0, mid))


def solve_robust_pgo(meas: Measurements, n: int, T0, opt_params: Optional[ROptParameters] = None, barc: float = 5.0,
                     mu_step: float = 1.4, max_iters: int = 20, precond: str = "exact", hess_recurrence: bool = False):
    """solveRobustPGO restated (src/DPGO_solver.cpp:335-412): rank r = d; every solve restarts from T0;
    weights of non-fixed edges from GNC-TLS; stop when no weight is undecided.  meas.weight is updated in
    place.  Returns (T, info)."""
    d = meas.d
    prm = opt_params or ROptParameters()
    w_tol = 1e-8

    def solve():
        Q = construct_Q(n, d, meas)
        opt = QuadraticOptimizer(QuadraticProblem(Q, None, d, d, precond=precond), prm, hess_recurrence=hess_recurrence)
        T = opt.optimize(T0.copy())
        return T, opt.result

    T, _ = solve()
    meas.weight[:] = 1.0
    rsq = measurement_error(meas, T)
    muInit = barc * barc / (2 * rsq.max() - barc * barc)
    info = dict(muInit=muInit, gnc_iterations=0, history=[])
    if muInit > 0:
        mu = muInit
        for it in range(max_iters):
            T, res = solve()
            rsq = measurement_error(meas, T)
            w = gnc_tls_weight(np.sqrt(rsq), mu, barc)
            meas.weight[~meas.fixed] = w[~meas.fixed]
            nf = meas.weight[~meas.fixed]
            n_out = int((nf < w_tol).sum()); n_in = int((nf > 1 - w_tol).sum()); n_und = len(nf) - n_in - n_out
            info["history"].append(dict(mu=mu, inliers=n_in, outliers=n_out, undecided=n_und, f=res.fOpt))
            info["gnc_iterations"] = it + 1
            if n_und == 0:
                break
            mu = mu_step * mu  # RobustCost::update (the GNCMaxNumIters guard cannot fire inside this loop)
    T, res = solve()
    info["fOpt"] = res.fOpt
    return T, info


# --------------------------------------------------------------------------
# Multi-agent GNC (checker for dpgo_amd.robust.DistributedGNC)
# --------------------------------------------------------------------------


def multi_agent_gnc(meas: Measurements, n: int, num_robots: int, r: int, X0, inner_sweeps: int = 5, barc: float = 5.0,
                    mu_step: float = 1.4, max_updates: int = 30, params: Optional[ROptParameters] = None,
                    precond: str = "jacobi", hess_recurrence: bool = False, agent_params=None):
    """Synchronous distributed GNC-TLS assembled from the reference's per-agent pieces (the in-tree library never
    calls them itself; the external dpgo_ros driver does):
      * PGOAgent::updateMeasurementWeights (src/PGOAgent.cpp:1104-1142): every agent re-weights ALL its non-fixed
        loop closures, private and shared, from its own iterate and its neighbours' public poses
        (computeMeasurementResidual, :1048-1102), w = RobustCost::weight(residual), then mu <- mu_step * mu
        (RobustCost::update) and the data matrices are rebuilt (clearDataMatrices); warm start (robustOptNumResets = 0);
      * initial mu as in solveRobustPGO (src/DPGO_solver.cpp:358) from the largest residual of the first solve;
      * between weight updates: `inner_sweeps` coloured RBCD sweeps (robustOptInnerIters analogue) -- or, with
        agent_params (an AgentParameters), the reference's own trigger, PGOAgent::shouldUpdateMeasurementWeights
        (src/PGOAgent.cpp:997-1045): global iterations (= colour phases) until every agent is readyToTerminate
        (relative change of its last update <= relChangeTol, 5 before the first weight update; converged-weight ratio
        >= robustOptMinConvergenceRatio) or robustOptInnerIters of them have passed; info["inner_iterations"] lists
        the count of every block;
      * stop when no weight is undecided (tolerance 1e-8, DPGO_solver.cpp:340) or after max_updates.
    Both endpoints of a shared edge compute the same residual from the same poses, hence the same weight.
    `meas` holds GLOBAL indices; its weight array is updated in place.  Returns (X, info)."""
    d = meas.d
    prm = params or ROptParameters()
    w_tol = 1e-8
    central_n = n

    inner_counts = []
    state = dict(weight_updates=0, iteration=0)

    def sweeps(X, k):
        if agent_params is None:
            for _ in range(k):
                X, _, _ = rbcd_coloured(meas, central_n, num_robots, r, X, 1, prm, precond, hess_recurrence)
            return X
        ap = agent_params
        ranges, per = partition_contiguous(meas, central_n, num_robots)
        ags = []
        for a in range(num_robots):
            s_, e_ = ranges[a]
            priv = Measurements.concat([per[a]["odometry"], per[a]["private"]])
            sh = per[a]["shared"]
            need = sorted({(int(sh.r2[q]), int(sh.p2[q])) if sh.r1[q] == a else (int(sh.r1[q]), int(sh.p1[q]))
                           for q in range(sh.m)})
            lcw = np.concatenate([per[a]["private"].weight, sh.weight])
            ags.append(dict(shared=sh, need=need, adj=sorted({rob for rob, _ in need}), lcw=lcw,
                            prob=QuadraticProblem(construct_Q(e_ - s_, d, priv, sh, my_id=a), None, r, d, precond=precond)))
        colour = [-1] * num_robots
        for a in range(num_robots):
            used = {colour[q] for q in ags[a]["adj"] if colour[q] >= 0}
            colour[a] = min(c for c in range(num_robots + 1) if c not in used)
        ncol = max(colour) + 1
        team, inner, latest = {}, 0, state["iteration"]
        while True:
            c = state["iteration"] % ncol
            state["iteration"] += 1
            inner += 1
            for a in range(num_robots):
                if colour[a] != c:
                    continue
                s_, e_ = ranges[a]
                nbr = {(rob, fr): X[ranges[rob][0] + fr] for rob, fr in ags[a]["need"]}
                ags[a]["prob"].G = construct_G(e_ - s_, d, r, ags[a]["shared"], a, nbr)
                opt = QuadraticOptimizer(ags[a]["prob"], prm, hess_recurrence=hess_recurrence)
                XPrev = X[s_:e_].copy()
                X[s_:e_] = opt.optimize(X[s_:e_])
                team[a] = local_status(a, state["iteration"], X[s_:e_], XPrev, opt.result.success, ap,
                                       state["weight_updates"], ags[a]["lcw"])
            # (the cap on the NUMBER of updates is the caller's loop bound, so the count passed here is 0)
            if should_update_weights(AgentParameters(**{**ap.__dict__, "robust": True}), 0, inner, latest, team, num_robots):
                break
        inner_counts.append(inner)
        return X

    meas.weight[:] = 1.0
    X = sweeps(X0.copy(), inner_sweeps)
    rsq = measurement_error(meas, X)
    muInit = barc * barc / (2 * rsq.max() - barc * barc)
    info = dict(muInit=muInit, updates=0, history=[])
    if muInit > 0:
        mu = muInit
        for it in range(max_updates):
            rsq = measurement_error(meas, X)
            w = gnc_tls_weight(np.sqrt(rsq), mu, barc)
            meas.weight[~meas.fixed] = w[~meas.fixed]
            nf = meas.weight[~meas.fixed]
            n_out = int((nf < w_tol).sum()); n_in = int((nf > 1 - w_tol).sum()); n_und = len(nf) - n_in - n_out
            info["history"].append(dict(mu=mu, inliers=n_in, outliers=n_out, undecided=n_und))
            info["updates"] = it + 1
            state["weight_updates"] = it + 1
            if n_und == 0:
                break
            mu = mu_step * mu
            X = sweeps(X, inner_sweeps)
    X = sweeps(X, inner_sweeps)
    info["inner_iterations"] = inner_counts
    central = QuadraticProblem(construct_Q(n, d, meas), None, r, d)
    info["cost"] = 2 * central.f(X)
    info["gradnorm"] = central.rie_grad_norm(X)
    return X, info


# --------------------------------------------------------------------------
# Agent status, termination vote, weight-update trigger -- src/PGOAgent.cpp:399-420, 846-878, 997-1045
# --------------------------------------------------------------------------


@dataclass
class AgentParameters:
    """The fields of PGOAgentParameters these rules read, with the reference's defaults
    (include/DPGO/PGOAgent.h:121-137)."""
    robust: bool = False  # robustCostParams.costType != L2
    robustOptNumWeightUpdates: int = 10
    robustOptInnerIters: int = 30
    robustOptMinConvergenceRatio: float = 0.8
    maxNumIters: int = 500
    relChangeTol: float = 5e-3


@dataclass
class AgentStatus:
    """PGOAgentStatus (include/DPGO/PGOAgent.h:196-227); state: only INITIALIZED agents run the hot path."""
    agentID: int = 0
    state: str = "WAIT_FOR_DATA"
    instanceNumber: int = 0
    iterationNumber: int = 0
    readyToTerminate: bool = False
    relativeChange: float = 0.0


def max_translation_distance(X, Xprev):
    """LiftedPoseArray::maxTranslationDistance (src/manifold/Poses.cpp:86-94); tiles [n, d+1, r], the translation
    is the last column of a pose."""
    return float(np.sqrt(((X[:, -1, :] - Xprev[:, -1, :]) ** 2).sum(axis=1)).max())


def local_status(agent_id: int, iteration: int, X, XPrev, success: bool, prm: AgentParameters,
                 weight_update_count: int = 0, lc_weights=None) -> AgentStatus:
    """The status block of PGOAgent::iterate after an optimising step (src/PGOAgent.cpp:399-420).  lc_weights: weights
    of the agent's loop closures, private and shared (PoseGraph::statistics, src/PoseGraph.cpp:305-340: a loop closure
    counts as converged when its weight is exactly 1 or 0; with no loop closure the ratio is 0/0 and the comparison
    is false, as in the C++)."""
    rel = max_translation_distance(X, XPrev)
    ready = bool(success)
    tol = prm.relChangeTol
    if prm.robust and weight_update_count == 0:  # loose threshold during the initial inner iterations (:411-415)
        tol = 5.0
    if rel > tol:
        ready = False
    w = np.asarray(lc_weights if lc_weights is not None else [], dtype=np.float64)
    if len(w) > 0:
        ratio = float(((w == 1).sum() + (w == 0).sum()) / len(w))
        if ratio < prm.robustOptMinConvergenceRatio:
            ready = False
    return AgentStatus(agent_id, "INITIALIZED", 0, iteration, ready, rel)


def should_terminate(iteration: int, prm: AgentParameters, weight_update_count: int, team: Dict[int, AgentStatus],
                     num_robots: int, inactive=()) -> bool:
    """PGOAgent::shouldTerminate (src/PGOAgent.cpp:846-878); robots in `inactive` (PGOAgent::setRobotActive(id, false),
    :1173-1184) have no vote (:861-862)."""
    if iteration >= prm.maxNumIters:
        return True
    if prm.robust and weight_update_count < prm.robustOptNumWeightUpdates:
        return False
    for rob in range(num_robots):
        if rob in inactive:
            continue
        st = team.get(rob)
        if st is None or st.state != "INITIALIZED" or not st.readyToTerminate:
            return False
    return True


def should_update_weights(prm: AgentParameters, weight_update_count: int, inner_iter: int,
                          latest_update_iteration: int, team: Dict[int, AgentStatus], num_robots: int, inactive=()) -> bool:
    """PGOAgent::shouldUpdateMeasurementWeights (src/PGOAgent.cpp:997-1045); inactive robots are skipped (:1016-1017)."""
    if not prm.robust:
        return False
    if weight_update_count >= prm.robustOptNumWeightUpdates:
        return False
    if inner_iter >= prm.robustOptInnerIters:
        return True
    for rob in range(num_robots):
        if rob in inactive:
            continue
        st = team.get(rob)
        if st is None or st.iterationNumber < latest_update_iteration or st.state != "INITIALIZED" \
                or not st.readyToTerminate:
            return False
    return True


def rbcd_coloured_until_terminated(meas: Measurements, n: int, num_robots: int, r: int, X0,
                                   agent_params: Optional[AgentParameters] = None,
                                   params: Optional[ROptParameters] = None, precond: str = "jacobi",
                                   hess_recurrence: bool = False, max_phases: int = 10000):
    """The coloured schedule of rbcd_coloured driven by the reference's OWN stopping rule instead of a sweep count:
    one global iteration = one colour phase (the agents of the colour call iterate(true), the others iterate(false):
    their iteration number advances, their status does not, src/PGOAgent.cpp:376-420); after every phase the team
    status is shared and every agent evaluates shouldTerminate (:846-878) -- all of them see the same statuses, so
    they stop together.  Returns (X, dict(iterations, statuses, relative_changes per phase))."""
    d = meas.d
    ap = agent_params or AgentParameters()
    ranges, per = partition_contiguous(meas, n, num_robots)
    agents = []
    for a in range(num_robots):
        s, e = ranges[a]
        priv = Measurements.concat([per[a]["odometry"], per[a]["private"]])
        Qa = construct_Q(e - s, d, priv, per[a]["shared"], my_id=a)
        sh = per[a]["shared"]
        need = sorted({(int(sh.r2[k]), int(sh.p2[k])) if sh.r1[k] == a else (int(sh.r1[k]), int(sh.p1[k]))
                       for k in range(sh.m)})
        agents.append(dict(shared=sh, need=need, adj=sorted({rob for rob, _ in need}),
                           prob=QuadraticProblem(Qa, None, r, d, precond=precond)))
    colour = [-1] * num_robots
    for a in range(num_robots):
        used = {colour[q] for q in agents[a]["adj"] if colour[q] >= 0}
        colour[a] = min(c for c in range(num_robots + 1) if c not in used)
    ncol = max(colour) + 1
    X = X0.copy()
    team: Dict[int, AgentStatus] = {}
    iteration, trace = 0, []
    while iteration < max_phases:
        c = iteration % ncol
        iteration += 1
        for a in range(num_robots):
            if colour[a] != c:
                continue
            s, e = ranges[a]
            nbr = {(rob, fr): X[ranges[rob][0] + fr] for rob, fr in agents[a]["need"]}
            prob = agents[a]["prob"]
            prob.G = construct_G(e - s, d, r, agents[a]["shared"], a, nbr)
            opt = QuadraticOptimizer(prob, params or ROptParameters(), hess_recurrence=hess_recurrence)
            XPrev = X[s:e].copy()
            X[s:e] = opt.optimize(X[s:e])
            team[a] = local_status(a, iteration, X[s:e], XPrev, opt.result.success, ap)
        trace.append({a: st.relativeChange for a, st in team.items()})
        if should_terminate(iteration, ap, 0, team, num_robots):
            break
    return X, dict(iterations=iteration, statuses=team, relative_changes=trace)


# --------------------------------------------------------------------------
# Rounding
# --------------------------------------------------------------------------


def project_to_rotation_group(M):
    """projectToRotationGroup (src/DPGO_utils.cpp:464-478): U V^T from the full SVD, last column of U negated
    when det(U) det(V) < 0."""
    U, _, Vt = np.linalg.svd(M)
    if np.linalg.det(U) * np.linalg.det(Vt) > 0:
        return U @ Vt
    U = U.copy()
    U[:, -1] *= -1
    return U @ Vt


def round_trajectory(X, d, anchor=None):
    """PGOAgent::getTrajectoryInLocalFrame (anchor None: pose 0) / getTrajectoryInGlobalFrame
    (src/PGOAgent.cpp:718-767).  X: tiles [n, d+1, r]; anchor: tile [d+1, r].  Returns tiles [n, d+1, d]
    (rows 0..d-1 = the columns of the rotation, row d = translation), i.e. the d x (d+1)n matrix column-major."""
    n = X.shape[0]
    A = X[0] if anchor is None else anchor
    Ya = A[:d].T  # r x d
    t0 = Ya.T @ A[d]
    T = np.zeros((n, d + 1, d))
    for i in range(n):
        M = Ya.T @ X[i, :d].T  # d x d
        T[i, :d] = project_to_rotation_group(M).T
        T[i, d] = Ya.T @ X[i, d] - t0
    return T
