"""diagnostic: kitti_00 r = 3 with a forced persistent layout: counts vs oracle, run-to-run bitwise determinism."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import conftest, dpgo_oracle as O, test_parity_gpu as t
import dpgo_amd
r = int(sys.argv[1]) if len(sys.argv) > 1 else 3
name = sys.argv[2] if len(sys.argv) > 2 else "kitti_00"
om, n, d, Q, pg, prob = t.build_single_agent(O, name, r)
prob.setPersistent(True)
X0 = O.lift(O.chordal_initialization(om, n), r)
op = O.QuadraticProblem(Q, None, r, d, precond="jacobi")
oo = O.QuadraticOptimizer(op, O.ROptParameters(), hess_recurrence=True)
Xo = oo.optimize(X0)
outs = []
for rep in range(1):
    go = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond="jacobi"))
    Xg = conftest.matrix_to_tiles(go.optimize(conftest.tiles_to_matrix(X0)), d)
    rg = go.getOptResult()
    outs.append(Xg.copy())
    print(name, r, "layout", (prob.persistentInfo()["last_split"], prob.persistentInfo()["last_tiles"]), "counts", (rg.tcg_iterations, rg.rtr_iterations, rg.tCGStatus), "oracle",
          (oo.result.tcg_iters, oo.result.outer_iters, O.TCG_NAMES[oo.result.tCGStatus]), "relerr %.3e" % t.relerr(Xg, Xo),
          "f %.12e vs %.12e" % (rg.fOpt, oo.result.fOpt), "bitwise==rep0:", bool(np.array_equal(outs[0], Xg)), flush=True)
