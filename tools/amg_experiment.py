"""Experiment (CPU, NumPy/SciPy): how good is an aggregation multigrid preconditioner for the connection Laplacian?

Aggregates = runs of k consecutive poses (the odometry chain); prolongation block (i, a) = homogeneous relative
pose from the aggregate's root to pose i composed from the ODOMETRY measurements, i.e. the exact kernel of the
chain's own Laplacian restricted to the aggregate ("rigid-body" coarse modes).  The coarse operator P^T (Q + s I) P
is again block-sparse with (d+1) x (d+1) blocks, so the hierarchy recurses with the same block-SpMM.  The V-cycle
(damped block-Jacobi smoothing, dense solve on the coarsest level) replaces the block-Jacobi solve inside
QuadraticProblem::PreConditioner; tCG Hessian-vector products are counted per RBCD iteration.

usage: python tools/amg_experiment.py [k] [levels]
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dpgo_oracle as O  # noqa: E402


def chain_prolongations(meas, n, d, k, levels):
    """[P_0, P_1, ...]: P_l maps level l+1 (aggregates of k^(l+1) consecutive poses, rooted at their first pose) to
    level l (aggregates of k^l poses; level 0 = poses).  Block (a, a // k) = G(parent root -> root of a)^T, the
    homogeneous relative pose composed from the ODOMETRY measurements: P_0 P_1 ... C reproduces, on every
    aggregate, the kernel vectors V_i = G_i^T C of the chain's own Laplacian ("rigid-body" coarse modes)."""
    b = d + 1
    odo = {int(meas.p1[e]): e for e in range(meas.m) if meas.p1[e] + 1 == meas.p2[e] and meas.r1[e] == meas.r2[e]}
    # absolute chain poses (identity restarts where the odometry is broken)
    A = [np.eye(b)]
    for i in range(1, n):
        if (i - 1) in odo:
            e = odo[i - 1]
            T = np.eye(b)
            T[:d, :d] = meas.R[e]
            T[:d, d] = meas.t[e]
            A.append(A[-1] @ T)
        else:
            A.append(np.eye(b))
    Ps, cur_n, stride = [], n, 1
    for lv in range(levels - 1):
        nc = (cur_n + k - 1) // k
        rows, cols, vals = [], [], []
        for a in range(cur_n):
            root_a = a * stride
            root_p = (a // k) * stride * k
            G = np.linalg.solve(A[root_p], A[root_a])  # relative pose parent root -> root of a
            blk = G.T
            for p in range(b):
                for q in range(b):
                    rows.append(a * b + p)
                    cols.append((a // k) * b + q)
                    vals.append(blk[p, q])
        Ps.append(sp.csr_matrix((vals, (rows, cols)), shape=(cur_n * b, nc * b)))
        cur_n, stride = nc, stride * k
        if cur_n <= 48:
            break
    return Ps


class Level:
    def __init__(self, A, b, omega=0.7):
        self.A = A.tocsr()
        self.b = b
        n = A.shape[0] // b
        D = np.zeros((n, b, b))
        Ab = self.A.tobsr(blocksize=(b, b))
        for i in range(n):
            for t in range(Ab.indptr[i], Ab.indptr[i + 1]):
                if Ab.indices[t] == i:
                    D[i] = Ab.data[t]
        self.Dinv = np.linalg.inv(D)
        self.omega = omega
        self.P = None
        self.dense = None

    def smooth(self, x, rhs):
        res = rhs - self.A @ x
        n = res.shape[0] // self.b
        corr = (self.Dinv @ res.reshape(n, self.b, -1)).reshape(res.shape)
        return x + self.omega * corr


class AMG:
    def __init__(self, A, meas, n, d, k, levels, coarse="dense"):
        self.levels = [Level(A, d + 1)]
        for P in chain_prolongations(meas, n, d, k, levels):
            L = self.levels[-1]
            L.P = P
            self.levels.append(Level((P.T @ L.A @ P).tocsr(), d + 1))
        last = self.levels[-1]
        self.coarse = coarse
        if coarse == "dense":
            last.dense = np.linalg.inv(last.A.toarray())
        self.cost = sum(L.A.nnz for L in self.levels) / self.levels[0].A.nnz
        self.sizes = [L.A.shape[0] // (d + 1) for L in self.levels]

    def vcycle(self, rhs, lv=0):
        L = self.levels[lv]
        if lv == len(self.levels) - 1:
            if L.dense is not None:
                return L.dense @ rhs
            x = np.zeros_like(rhs)
            for _ in range(4):  # a few smoothing steps instead of a coarse solve
                x = L.smooth(x, rhs)
            return x
        x = L.smooth(np.zeros_like(rhs), rhs)
        rc = L.P.T @ (rhs - L.A @ x)
        x = x + L.P @ self.vcycle(rc, lv + 1)
        return L.smooth(x, rhs)


class AMGProblem(O.QuadraticProblem):
    def __init__(self, Q, r, d, meas, n, k, levels):
        super().__init__(Q, None, r, d, precond="jacobi")
        A = (self.Qs + self.shift * sp.identity(self.N, format="csr")).tocsr()
        self.amg = AMG(A, meas, n, d, k, levels)

    def precondition(self, X, V):
        Z = self.amg.vcycle(V.reshape(self.N, self.r)).reshape(V.shape)
        return O.tangent_project(X, Z, self.d)


def run(name, meas, n, X0, r, iters, k, levels):
    d = meas.d
    Q = O.construct_Q(n, d, meas)
    for label in ("jacobi", "amg", "exact"):
        t = time.time()
        if label == "amg":
            P = AMGProblem(Q, r, d, meas, n, k, levels)
            extra = " (level sizes %s, operator complexity %.2f)" % (P.amg.sizes, P.amg.cost)
        else:
            P = O.QuadraticProblem(Q, None, r, d, precond=label)
            extra = ""
        X = X0.copy()
        out = []
        for _ in range(iters):
            opt = O.QuadraticOptimizer(P, O.ROptParameters())
            X = opt.optimize(X)
            out.append((opt.result.tcg_iters, float("%.3g" % opt.result.gradNormOpt)))
        print("%-10s %-7s tCG products %4d %s%s  %.1fs" % (name, label, sum(a for a, _ in out), out, extra,
                                                           time.time() - t))


if __name__ == "__main__":
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    levels = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    om, n, Tt = O.synthetic_grid(25, 25, 10, seed=0)
    run("grid6250", om, n, O.lift(O.perturbed_truth(Tt, seed=2), 5), 5, 4, k, levels)
    meas, n = O.read_g2o(os.path.join(os.path.dirname(__file__), "..", "data", "sphere2500.g2o"))
    run("sphere2500", meas, n, O.lift(O.chordal_initialization(meas, n), 5), 5, 3, k, levels)
    meas, n = O.read_g2o(os.path.join(os.path.dirname(__file__), "..", "data", "torus3D.g2o"))
    run("torus3D", meas, n, O.lift(O.chordal_initialization(meas, n), 5), 5, 3, k, levels)
