"""GPU probe: setup time of the multilevel hierarchy (Gauss-Jordan rank-64 updates on FMAs vs fp64 matrix cores) and
RBCD-iteration cost / convergence with the multilevel and the block-Jacobi preconditioner.
usage: python tools/ml_probe.py [sphere|slab|grid100k|torus ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: E402
import dpgo_amd  # noqa: E402
import dpgo_oracle as O  # noqa: E402  (workload generation only)
from dpgo_amd.measurements import RelativeSEMeasurements  # noqa: E402


def workload(name):
    if name in ("slab", "grid100k", "grid25k", "grid6250", "grid625"):
        dims = {"slab": (50, 50, 5), "grid100k": (50, 50, 40), "grid25k": (50, 50, 10), "grid6250": (25, 25, 10),
                "grid625": (25, 5, 5)}[name]
        om, n, Tt = O.synthetic_grid(*dims, seed=0)
        X0 = O.lift(O.perturbed_truth(Tt, seed=2), 5)
    else:
        f = {"sphere": "sphere2500.g2o", "torus": "torus3D.g2o", "kitti": "kitti_00.g2o"}[name]
        om, n = O.read_g2o(os.path.join(ROOT, "data", f))
        X0 = O.lift(O.chordal_initialization(om, n), 5)
    c = np.copy
    pm = RelativeSEMeasurements(om.d, c(om.r1), c(om.p1), c(om.r2), c(om.p2), c(om.R), c(om.t), c(om.kappa), c(om.tau),
                                c(om.weight), c(om.fixed))
    return pm, n, X0


for name in (sys.argv[1:] or ["sphere", "slab", "grid100k"]):
    pm, n, X0 = workload(name)
    pg = dpgo_amd.PoseGraph(0, 5, pm.d)
    pg.setMeasurements(pm)
    prob = dpgo_amd.QuadraticProblem(pg)
    for mf in (("0", "1") if os.environ.get("PROBE_SETUP") else ()):
        os.environ["DPGO_GJ_MFMA"] = mf
        dpgo_amd.lib.load().dpgo_options_reload()  # (the library reads its switches once)
        ts = []
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            info = prob.setupMultilevel()
            ts.append(time.perf_counter() - t0)
        print("%-9s setup mfma=%s: %s ms  %s" % (name, mf, ["%.2f" % (1e3 * t) for t in ts], info), flush=True)
    only = os.environ.get("PROBE_ONLY")
    for pc in ("multilevel", "additive", "jacobi", "jacobi+persistent"):
        if only and pc not in only.split(","):
            continue
        try:
            prob.setPersistent(pc.endswith("persistent") or pc == "additive")
        except dpgo_amd.DpgoError as exc:
            print("%-9s %-17s not available: %s" % (name, pc, exc), flush=True)
            continue
        opt = dpgo_amd.QuadraticOptimizer(prob, dpgo_amd.ROptParameters(precond=pc.split("+")[0]))
        Xd = torch.tensor(X0, device="cuda", dtype=torch.float64)
        # the hierarchy is a once-per-Q cost (the reference factors inside its first solve, src/PoseGraph.cpp:582-586):
        # built and timed by itself, so that "ms" below is the solves alone
        setup_ms = 0.0
        if pc in ("multilevel", "additive") and not os.environ.get("PROBE_LAZY_SETUP"):
            try:
                ks = prob.additivePlan()["ks"] if pc == "additive" else None
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                prob.setupMultilevel(ks)
                setup_ms = 1e3 * (time.perf_counter() - t0)
            except dpgo_amd.DpgoError as exc:
                print("%-9s %-17s not available: %s" % (name, pc, exc), flush=True)
                continue
        rows, tot, ms = [], 0, 0.0
        for it in range(8):
            try:
                res = opt.optimizeDevice(Xd)
            except dpgo_amd.DpgoError as exc:  # e.g. "additive" beyond 256 aggregates
                print("%-9s %-17s not available: %s" % (name, pc, exc), flush=True)
                rows = None
                break
            rows.append((res.tcg_iterations, float("%.3g" % res.gradNormOpt), float("%.2f" % res.elapsedMs)))
            tot += res.tcg_iterations
            ms += res.elapsedMs
            if res.gradNormOpt < 1e-2:
                break
        if rows is None:
            continue
        print("%-9s %-17s products %4d  %.2f ms  (%.1f us/product) + hierarchy once per Q %.2f ms  %s %s" % (
            name, pc, tot, ms, 1e3 * ms / max(tot, 1), setup_ms, rows,
            prob.persistentInfo() if ("persist" in pc or pc == "additive") else ""), flush=True)
