"""Experiment (CPU oracle): tCG products per RBCD iteration for the multilevel preconditioner's hierarchy choices
(aggregate sizes per level) against block-Jacobi and the reference's exact (Q + 0.1 I)^-1.

usage: python tools/amg_levels.py <case> <ks,...;ks,...> [iters] [exact]
  case: grid6250 | slab12500 | grid100k | sphere | torus | kitti | small
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dpgo_oracle as O  # noqa: E402

GAMMA = int(os.environ.get("AMG_GAMMA", "1"))
NU = int(os.environ.get("AMG_NU", "1"))
DATA = os.path.join(os.path.dirname(__file__), "..", "data")


def case(name):
    if name.startswith("grid") or name.startswith("slab"):
        dims = {"grid6250": (25, 25, 10), "slab12500": (50, 50, 5), "grid100k": (50, 50, 40),
                "grid25k": (50, 50, 10)}[name]
        meas, n, Tt = O.synthetic_grid(*dims, seed=0)
        return meas, n, O.lift(O.perturbed_truth(Tt, seed=2), 5)
    f = {"sphere": "sphere2500.g2o", "torus": "torus3D.g2o", "kitti": "kitti_00.g2o", "small": "smallGrid3D.g2o"}[name]
    meas, n = O.read_g2o(os.path.join(DATA, f))
    return meas, n, O.lift(O.chordal_initialization(meas, n), 5)


def run(name, configs, iters, with_exact):
    meas, n, X0 = case(name)
    d = meas.d
    Q = O.construct_Q(n, d, meas)
    labels = [("jacobi", None)] + [("amg", ks) for ks in configs] + ([("exact", None)] if with_exact else [])
    for label, ks in labels:
        t = time.time()
        P = O.QuadraticProblem(Q, None, 5, d, precond=label, amg_k=ks, amg_gamma=GAMMA, amg_nu=NU)
        extra = ""
        if label == "amg":
            m = P.amg_setup()
            sizes = [L["n"] for L in m["levels"]] + [m["nc"]]
            cost = sum(L["A"].nnz for L in m["levels"]) / m["levels"][0]["A"].nnz
            extra = " ks=%s sizes=%s opcx=%.2f dense=%d" % (m["ks"], sizes, cost, m["nc"] * (d + 1))
        X = X0.copy()
        out = []
        for _ in range(iters):
            opt = O.QuadraticOptimizer(P, O.ROptParameters())
            X = opt.optimize(X)
            out.append((opt.result.tcg_iters, float("%.3g" % opt.result.gradNormOpt)))
        # products until |rgrad| < 1e-2
        tot, hit = 0, None
        for a, g in out:
            tot += a
            if g < 1e-2 and hit is None:
                hit = tot
        print("%-10s %-7s products %4d to1e-2=%s %s%s  %.1fs" % (name, label, sum(a for a, _ in out), hit, out, extra,
                                                                 time.time() - t), flush=True)


if __name__ == "__main__":
    name = sys.argv[1]
    configs = [[int(v) for v in c.split(",")] for c in sys.argv[2].split(";")] if len(sys.argv) > 2 and sys.argv[2] else []
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    run(name, configs, iters, len(sys.argv) > 4)
