"""Adds `grid40x40x25_exact` to tests/golden/golden_scalars.json: the synthetic 40 x 40 x 25 grid (40 000 poses, the
generator of BASELINE configs[3] at a size whose exact sparse factor is still affordable ONCE) solved by the CPU oracle in
the REFERENCE configuration -- RTR with the exact (Q + 0.1 I)^-1 preconditioner (SciPy SuperLU for CHOLMOD,
src/QuadraticProblem.cpp:56-69) -- from the perturbed-truth iterate to |rgrad| < 1e-4.

Run in the build container:  python tests/golden/make_golden_grid40k.py     (about 10 minutes: the factorisation of the
160 000-unknown 3-D operator takes most of it; that is why the number is a committed fixture and not computed by the test)
Consumed by tests/test_parity_gpu.py::test_mixed_precision_default_reaches_the_reference_cost."""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dpgo_oracle as O  # noqa: E402

NX, NY, NZ, R = 40, 40, 25, 5
meas, n, Ttrue = O.synthetic_grid(NX, NY, NZ, seed=0)
X0 = O.lift(O.perturbed_truth(Ttrue, seed=2), R)
Q = O.construct_Q(n, 3, meas)
prob = O.QuadraticProblem(Q, None, R, 3, precond="exact")
t0 = time.time()
opt = O.QuadraticOptimizer(prob, O.ROptParameters(gradnorm_tol=1e-4, RTR_iterations=60, RTR_tCG_iterations=500))
opt.optimize(X0)
res = opt.result
entry = dict(grid=[NX, NY, NZ], n=n, edges=int(meas.m), r=R, seed_graph=0, seed_iterate=2, precond="exact",
             gradnorm_tol=1e-4, fInit=res.fInit, fOpt=res.fOpt, gradNormInit=res.gradNormInit, gradNormOpt=res.gradNormOpt,
             tcg_iters=res.tcg_iters, outer_iters=res.outer_iters, seconds=time.time() - t0)
path = os.path.join(HERE, "golden_scalars.json")
out = json.load(open(path))
out["grid40x40x25_exact"] = entry
with open(path, "w") as fh:
    json.dump(out, fh, indent=1)
print(entry)
