"""As aggregate_scan.py, plus the benchmark's fixed-work step (5 settle iterations, then one optimize() from the settled
iterate) for every hierarchy: products and time of the step.  usage: python tools/r6/aggregate_scan2.py [S ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dpgo_amd
import bench
from dpgo_amd.agent import DeviceAgent, ExchangePlan, build_pose_graphs
meas, n, X0, desc = bench.make_workload("grid100k", 5)
ranges, graphs = build_pose_graphs(meas, n, 1, 5)
sizes = [int(v) for v in sys.argv[1:]] or [170, 182, 190, 200, 210, 220]
for S in sizes:
    cap = S + S // 2
    ag = DeviceAgent(graphs, ExchangePlan(graphs), 0, X0, dpgo_amd.ROptParameters(precond="multilevel"))
    info = ag.problem.setupMultilevel([-S, -cap])
    ag.update()
    ag.set_iterate(X0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    products = 0
    for _ in range(12):
        res = ag.update()
        products += res.tcg_iterations
        if res.gradNormOpt < 1e-2:
            break
    torch.cuda.synchronize()
    el = 1e3 * (time.perf_counter() - t0)
    # the benchmark's step: settle 5, snapshot, time 10 restores + updates
    ag.set_iterate(X0)
    works, states = [], []
    for _ in range(6):
        states.append(ag.X.clone())
        works.append(ag.update().tcg_iterations)
    k = max(i for i in range(len(works)) if works[i] >= 0.5 * works[0])
    ag.X.copy_(states[k])
    ag.snapshot()
    for _ in range(2):
        ag.restore(); ag.update()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tcg = 0
    for _ in range(10):
        ag.restore()
        tcg += ag.update().tcg_iterations
    torch.cuda.synchronize()
    st = 1e3 * (time.perf_counter() - t0) / 10
    print("S %3d cap %3d: %4d aggregates | to tolerance %3d products %.2f ms | step (settle %d) %.1f products %.3f ms = %.1f it/s, %.1f us / product" % (
        S, cap, info["sizes"][-1], products, el, k, tcg / 10, st, 1e3 / st, 1e3 * st / (tcg / 10)), flush=True)
    del ag
