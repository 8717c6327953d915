"""Where the once-per-pattern cost of the multilevel hierarchy goes: DPGO_SETUP_TIMING=1 section report of the symbolic
set-up + wall time of first / values-only set-ups on a fresh handle.  usage: python tools/setup_timing.py [workload]"""
import os, sys, time
os.environ["DPGO_SETUP_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dpgo_amd
from dpgo_amd.agent import build_pose_graphs
import bench
w = sys.argv[1] if len(sys.argv) > 1 else "grid100k"
meas, n, X0, desc = bench.make_workload(w, 5)
ranges, graphs = build_pose_graphs(meas, n, 1, 5)
for rep in range(2):
    pr = dpgo_amd.QuadraticProblem(graphs[0], host_linear_term=False)
    ts = []
    for _ in range(2):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        pr.setupMultilevel(None)
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t1))
    print("handle %d: first set-up %.2f ms, values only %.2f ms" % (rep, ts[0], ts[1]), flush=True)
    del pr
