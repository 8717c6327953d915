"""Design experiment (CPU, NumPy/SciPy; not part of the product path): aggregates for the additive two-level
preconditioner of the one-launch solve.  There an aggregate is a WORKGROUP (all of its poses live in one workgroup's
registers / LDS), so the number of aggregates is bounded by the 256 resident workgroups and fragments of the greedy growth
waste whole workgroups.  Compares, for a size bound S,
  greedy      amg_graph_aggregates (seeds in index order, breadth-first growth),
  merged      the same followed by amg_merge_small_aggregates (fragments joined to the neighbour they touch most, up
              to `cap` poses -- the workgroup's tile),
by the number of aggregates and by Hessian-vector products until |rgrad| < 1e-2.

usage: python tools/balanced_aggregates_experiment.py 50x50x5|sphere|torus 16,32,55:64 [vcycle]  (S or S:cap; vcycle: the
       V(1,1) cycle instead of the additive form)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dpgo_oracle as O  # noqa: E402


def main():
    r, d = 5, 3
    name = sys.argv[1]
    if "x" in name:
        dims = [int(v) for v in name.split("x")]
        meas, n, Ttrue = O.synthetic_grid(*dims, seed=0)
        X0 = O.lift(O.perturbed_truth(Ttrue, seed=2), r)
    else:
        f = {"sphere": "sphere2500.g2o", "torus": "torus3D.g2o"}[name]
        meas, n = O.read_g2o(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", f))
        X0 = O.lift(O.chordal_initialization(meas, n), r)
    Q = O.construct_Q(n, d, meas)
    for spec in sys.argv[2].split(","):  # S or S:cap
        S, cap = (int(v) for v in (spec.split(":") if ":" in spec else (spec, "0")))
        for mode in (("merged",) if cap else ("greedy",)):
            t0 = time.time()
            pc = "amg" if len(sys.argv) > 3 and sys.argv[3] == "vcycle" else "amg_additive"
            op = O.QuadraticProblem(Q, None, r, d, precond=pc, amg_k=[-S], amg_merge=cap)
            opt = O.QuadraticOptimizer(op, O.ROptParameters())
            X, total, rows = X0.copy(), 0, []
            for _ in range(12):
                X = opt.optimize(X)
                total += opt.result.tcg_iters
                rows.append((opt.result.tcg_iters, float("%.3g" % opt.result.gradNormOpt)))
                if opt.result.gradNormOpt < 1e-2:
                    break
            print("  S=%3d cap=%3d %-7s aggregates %5d  products %4d  %s  (%.0f s)" % (S, cap, mode, op.amg_setup()["nc"], total, rows,
                                                                              time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
