"""Products to tolerance of the 100k grid for growth sizes S / merge bounds and range counts of the aggregates' growth:
what a hierarchy costs (dense level size, products) as a function of the aggregation rule.
usage: python tools/r6/aggregate_scan.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dpgo_amd
import bench
from dpgo_amd.agent import DeviceAgent, ExchangePlan, build_pose_graphs
lib = dpgo_amd.lib.load()
meas, n, X0, desc = bench.make_workload("grid100k", 5)
ranges, graphs = build_pose_graphs(meas, n, 1, 5)
for chunks in (1, 8):
    os.environ["DPGO_ML_GROWTH_CHUNKS"] = str(chunks)
    dpgo_amd.lib.check(lib.dpgo_options_reload())
    for S, cap in ((182, 273), (160, 240), (200, 300), (220, 330), (182, 364), (150, 300), (240, 360)):
        ag = DeviceAgent(graphs, ExchangePlan(graphs), 0, X0, dpgo_amd.ROptParameters(precond="multilevel"))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        info = ag.problem.setupMultilevel([-S, -cap])
        torch.cuda.synchronize()
        t_setup = 1e3 * (time.perf_counter() - t0)
        ag.update()  # warm-up
        ag.set_iterate(X0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        products, calls = 0, 0
        for _ in range(12):
            res = ag.update()
            products += res.tcg_iterations
            calls += 1
            if res.gradNormOpt < 1e-2:
                break
        torch.cuda.synchronize()
        el = 1e3 * (time.perf_counter() - t0)
        print("ranges %d S %3d cap %3d: %4d aggregates, set-up %.2f ms, %3d products in %d calls, %.2f ms (%.1f us / product), sum %.2f ms" % (
            chunks, S, cap, info["sizes"][-1], t_setup, products, calls, el, 1e3 * el / products, el + t_setup), flush=True)
        del ag
