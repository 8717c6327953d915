"""Generates tests/golden/golden_scalars.json and smallGrid3D_vectors.npz from the CPU oracle.

Run in the build container:  python tests/golden/make_golden.py
The reference itself is C++ that cannot be built or imported here, so these vectors come
from the oracle (which is pinned by the reference's known-answer tests and the literature optima,
tests/test_oracle.py); they guard the oracle and the HIP path against drift.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import dpgo_oracle as O  # noqa: E402

SEED = 11
out = {"datasets": {}}
for name in ("tinyGrid3D", "smallGrid3D", "sphere2500", "torus3D", "kitti_00"):
    om, n = O.read_g2o(os.path.join(ROOT, "data", name + ".g2o"))
    Q = O.construct_Q(n, om.d, om)
    X = O.polar_project(np.random.default_rng(SEED).standard_normal((n, om.d + 1, 5)), om.d)
    p = O.QuadraticProblem(Q, None, 5, om.d)
    out["datasets"][name] = dict(n=n, m=om.m, d=om.d, kappa_sum=float(om.kappa.sum()), tau_sum=float(om.tau.sum()),
                                 nnzb=int(Q.nnzb), Q_sum=float(Q.vals.sum()), Q_abs_sum=float(np.abs(Q.vals).sum()),
                                 seed=SEED, f_random=float(p.f(X)), gradnorm_random=float(p.rie_grad_norm(X)))

om, n = O.read_g2o(os.path.join(ROOT, "data", "smallGrid3D.g2o"))
Q = O.construct_Q(n, 3, om)
p = O.QuadraticProblem(Q, None, 5, 3, precond="jacobi")
rng = np.random.default_rng(SEED + 1)
M = rng.uniform(-1, 1, (n, 4, 5))
X = O.polar_project(rng.standard_normal((n, 4, 5)), 3)
V = O.tangent_project(X, rng.standard_normal((n, 4, 5)), 3)
eta = 0.2 * V
S = p.sym_ytg(X, p.euc_grad(X))
# the device's multilevel preconditioner (default hierarchy, 64-bit dense level): one application and one optimize
pm = O.QuadraticProblem(Q, None, 5, 3, precond="amg")
np.savez_compressed(os.path.join(HERE, "smallGrid3D_vectors.npz"), X=X, V=V, eta=eta, M=M, XQ=p.XQ(X),
                    rgrad=p.rie_grad(X), rhess=p.rie_hess(X, S, V), precond_jacobi=p.precondition(X, V),
                    precond_multilevel=pm.precondition(X, V),
                    retract=O.qf_retract(X, eta, 3), polar=O.polar_project(M, 3))
optm = O.QuadraticOptimizer(pm, O.ROptParameters(), hess_recurrence=True)  # the arithmetic the device runs
optm.optimize(O.lift(O.chordal_initialization(om, n), 5))
out["smallGrid3D_rtr_trace_multilevel"] = dict(
    ks=pm.amg_setup()["ks"], tcg_iters=optm.result.tcg_iters, outer_iters=optm.result.outer_iters,
    fInit=optm.result.fInit, fOpt=optm.result.fOpt, gradNormOpt=optm.result.gradNormOpt)
opt = O.QuadraticOptimizer(p, O.ROptParameters())
opt.optimize(O.lift(O.chordal_initialization(om, n), 5))
out["smallGrid3D_rtr_trace_jacobi"] = dict(
    tcg_iters=opt.result.tcg_iters, outer_iters=opt.result.outer_iters, fInit=opt.result.fInit,
    fOpt=opt.result.fOpt, gradNormOpt=opt.result.gradNormOpt,
    trace=[dict(inner=t["inner"], status=t["status"], accept=bool(t["accept"]), rho=t["rho"], f2=t["f2"],
                Delta=t["Delta"]) for t in opt.result.trace])
# BASELINE configs[2] in the REFERENCE configuration (exact (Q_a + 0.1 I)^-1 preconditioner, RTR 3 x <= 50): torus3D cut
# into 8 agents, 120 two-colour sweeps from the chordal initialisation (SURVEY 8e's probe: 2f = 24227.072 after sweep 119)
om, n = O.read_g2o(os.path.join(ROOT, "data", "torus3D.g2o"))
_, costs, gns = O.rbcd_coloured(om, n, 8, 5, O.lift(O.chordal_initialization(om, n), 5), 120, precond="exact")
out["torus3D_8_agents_120_sweeps_exact"] = dict(
    cost_2f={str(k): costs[k] for k in (0, 9, 49, 99, 119)}, gradnorm_after=gns[119], optimum_2f=24227.0455583823)
with open(os.path.join(HERE, "golden_scalars.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print("wrote golden vectors")
