"""Design experiment (CPU, oracle; not part of the product path): over-correction of the coarse-grid step of the
unsmoothed-aggregation preconditioner.  Plain (tentative) prolongations under-estimate the coarse correction; the classical
remedy scales it, x <- x1 + gamma P xc, i.e. the dense level stores gamma A_c^-1 -- still SPD, free per iteration.
Hessian-vector products until |rgrad| < 1e-2 (QuadraticOptimizer::optimize repeatedly from the benchmark's initial
iterate, reference default parameters), default hierarchy of the device.

usage: python tools/r6/overcorrection_experiment.py <grid NXxNYxNZ | sphere2500 | torus3D> 1.0,1.25,1.5,1.75,2.0 [additive] [omega]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import dpgo_oracle as O  # noqa: E402


def main():
    what = sys.argv[1]
    gammas = [float(v) for v in sys.argv[2].split(",")]
    pc = "amg_additive" if len(sys.argv) > 3 and sys.argv[3] == "additive" else "amg"
    omega = float(sys.argv[4]) if len(sys.argv) > 4 else 0.7
    r, d = 5, 3
    if "x" in what:
        meas, n, Ttrue = O.synthetic_grid(*[int(v) for v in what.split("x")], seed=0)
        X0 = O.lift(O.perturbed_truth(Ttrue, seed=2), r)
    else:
        meas, n = O.read_g2o(os.path.join(os.path.dirname(__file__), "..", "..", "data", what + ".g2o"))
        X0 = O.lift(O.chordal_initialization(meas, n), r)
    Q = O.construct_Q(n, d, meas)
    print("%s: %d poses, %s, omega %.2f" % (what, n, pc, omega))
    for g in gammas:
        t0 = time.time()
        op = O.QuadraticProblem(Q, None, r, d, precond=pc, amg_omega=omega)
        m = op.amg_setup()
        m["AcInv"] = g * m["AcInv"]
        opt = O.QuadraticOptimizer(op, O.ROptParameters())
        X, total, rows = X0.copy(), 0, []
        for _ in range(12):
            X = opt.optimize(X)
            total += opt.result.tcg_iters
            rows.append((opt.result.tcg_iters, float("%.3g" % opt.result.gradNormOpt)))
            if opt.result.gradNormOpt < 1e-2:
                break
        print("  gamma %.2f  aggregates %5d  products %4d  %s  (%.0f s)" % (g, m["nc"], total, rows, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
