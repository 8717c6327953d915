"""What a FRESH handle pays before its first tCG iteration at 100k poses: handle creation + Q upload, the symmetric
storage's set-up (host pattern + device values), the block-Jacobi factors, the hierarchy; then the first optimize() call
end to end.  usage: python tools/r6/first_solve_timing.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dpgo_amd
import bench
from dpgo_amd.agent import build_pose_graphs
meas, n, X0, desc = bench.make_workload("grid100k", 5)
ranges, graphs = build_pose_graphs(meas, n, 1, 5)
graphs[0].quadraticMatrix()  # (Q assembled on the host once: not part of what is timed below)
for rep in range(3):
    def lap(what, t0):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        print("  %-46s %7.2f ms" % (what, 1e3 * (t1 - t0)), flush=True)
        return t1
    print("fresh handle %d" % rep)
    t = time.perf_counter()
    pr = dpgo_amd.QuadraticProblem(graphs[0], host_linear_term=False)
    pr.refresh()
    t = lap("handle + Q uploaded", t)
    pr.setSpmmVariant("symmetric")
    X = torch.tensor(X0, device="cuda", dtype=torch.float64)
    t = lap("iterate uploaded", t)
    out = torch.empty_like(X)
    pr.spmmDevice(X, out)
    t = lap("first Q X product (symmetric storage set-up)", t)
    pr.setupMultilevel()
    t = lap("hierarchy (first set-up)", t)
    opt = dpgo_amd.QuadraticOptimizer(pr, dpgo_amd.ROptParameters(precond="multilevel"))
    res = opt.optimizeDevice(X)
    t = lap("first optimize(): %d products" % res.tcg_iterations, t)
    res = opt.optimizeDevice(X)
    t = lap("second optimize(): %d products" % res.tcg_iterations, t)
    del opt, pr
