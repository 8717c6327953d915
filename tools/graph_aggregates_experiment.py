"""Design experiment (CPU, NumPy/SciPy; not part of the product path): how much does the SHAPE of the aggregates of the
two-level preconditioner matter?  Hessian-vector products until |rgrad| < 1e-2 (QuadraticOptimizer::optimize called
repeatedly from the benchmark's initial iterate, reference default parameters) for
  runs<k>      index runs of k consecutive poses (round 2's hierarchy),
  graph<S>     breadth-first-grown graph aggregates of at most S poses (the device default; amg_graph_aggregates),
  cube<a>x<b>x<c>   lattice cubes -- only possible because this script knows the grid's geometry; the yardstick.
DESIGN.md section 5 quotes this table.

usage: python tools/graph_aggregates_experiment.py 25x25x16 runs64,graph64,cube4x4x4,graph25 [additive]
"""
import os
import sys
import time
from collections import deque

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dpgo_oracle as O  # noqa: E402


def cube_hierarchy(op, meas, n, d, pos, dims, cube):
    """Installs a one-coarsening hierarchy with lattice-cube aggregates in the oracle problem `op` (prolongation along a
    breadth-first tree of each cube, relative poses from the measurements)."""
    b = d + 1
    nx, ny, _ = dims
    cx, cy, cz = cube
    lab3 = (pos[:, 0] // cx) + (-(-nx // cx)) * ((pos[:, 1] // cy) + (-(-ny // cy)) * (pos[:, 2] // cz))
    _, lab = np.unique(lab3, return_inverse=True)
    adj = [[] for _ in range(n)]
    for e in range(len(meas.p1)):
        i, j = int(meas.p1[e]), int(meas.p2[e])
        T = np.eye(b)
        T[:d, :d], T[:d, d] = meas.R[e], meas.t[e]
        adj[i].append((j, T))
        adj[j].append((i, np.linalg.inv(T)))
    Pb, seen = np.zeros((n, b, b)), np.zeros(n, bool)
    for s in range(n):
        if seen[s]:
            continue
        seen[s], Pb[s] = True, np.eye(b)
        q = deque([(s, np.eye(b))])
        while q:
            u, G = q.popleft()
            for v, T in adj[u]:
                if not seen[v] and lab[v] == lab[u]:
                    seen[v] = True
                    Pb[v] = (G @ T).T
                    q.append((v, G @ T))
    na = int(lab.max()) + 1
    A = (op.Qs + op.shift * sp.identity(op.N, format="csr")).tocsr()
    rows = (np.arange(n)[:, None, None] * b + np.arange(b)[None, :, None]) + np.zeros((1, 1, b), dtype=np.int64)
    cols = (lab[:, None, None] * b + np.arange(b)[None, None, :]) + np.zeros((1, b, 1), dtype=np.int64)
    P = sp.csr_matrix((Pb.ravel(), (rows.ravel(), cols.ravel())), shape=(n * b, na * b))
    Ab = A.tobsr(blocksize=(b, b))
    Ab.sort_indices()
    rr = np.repeat(np.arange(n), np.diff(Ab.indptr))
    D = np.zeros((n, b, b))
    D[rr[rr == Ab.indices]] = Ab.data[rr == Ab.indices]
    Ac = (P.T @ A @ P).toarray()
    Ac = 0.5 * (Ac + Ac.T)
    op._amg = dict(ks=[0], levels=[dict(k=0, n=n, A=A, P=P, Pb=Pb, Dinv=np.linalg.inv(D))], Ac=Ac,
                   AcInv=np.linalg.inv(Ac), nc=na)


def main():
    dims = [int(v) for v in sys.argv[1].split("x")]
    modes = sys.argv[2].split(",")
    pc = "amg_additive" if len(sys.argv) > 3 and sys.argv[3] == "additive" else "amg"
    meas, n, Ttrue = O.synthetic_grid(*dims, seed=0)
    r, d = 5, 3
    X0 = O.lift(O.perturbed_truth(Ttrue, seed=2), r)
    Q = O.construct_Q(n, d, meas)
    pos = Ttrue[:, 3, :].round().astype(int)
    print("grid %s: %d poses, %s" % (sys.argv[1], n, "additive form" if pc == "amg_additive" else "V(1,1) cycle"))
    for m in modes:
        t0 = time.time()
        if m.startswith("runs"):
            op = O.QuadraticProblem(Q, None, r, d, precond=pc, amg_k=[int(m[4:])])
        elif m.startswith("graph"):
            op = O.QuadraticProblem(Q, None, r, d, precond=pc, amg_k=[-int(m[5:])])
        elif m.startswith("cube"):
            op = O.QuadraticProblem(Q, None, r, d, precond=pc, amg_k=[4])
            cube_hierarchy(op, meas, n, d, pos, dims, [int(v) for v in m[4:].split("x")])
        else:
            raise SystemExit("unknown mode %r" % m)
        opt = O.QuadraticOptimizer(op, O.ROptParameters())
        X, total, rows = X0.copy(), 0, []
        for _ in range(12):
            X = opt.optimize(X)
            total += opt.result.tcg_iters
            rows.append((opt.result.tcg_iters, float("%.3g" % opt.result.gradNormOpt)))
            if opt.result.gradNormOpt < 1e-2:
                break
        print("  %-14s aggregates %5d  products %4d  %s  (%.0f s)" % (m, op.amg_setup()["nc"], total, rows, time.time() - t0),
              flush=True)


if __name__ == "__main__":
    main()
