"""Per-step trace of the bench workload (tcg iterations, cost, gradnorm) for the library in DPGO_LIB."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dpgo_amd
from dpgo_amd import synthetic
from dpgo_amd.agent import DeviceAgent, ExchangePlan, RBCDCluster, build_pose_graphs
w = sys.argv[1] if len(sys.argv) > 1 else "50x50x40"
nx, ny, nz = (int(v) for v in w.split("x"))
meas, n, Tt = synthetic.synthetic_grid(nx, ny, nz, seed=0)
X0 = synthetic.lift_tiles(synthetic.perturbed_truth(Tt, seed=2), 5)
ranges, graphs = build_pose_graphs(meas, n, 1, 5)
plan = ExchangePlan(graphs)
ag = DeviceAgent(graphs, plan, 0, X0, dpgo_amd.ROptParameters())
for k in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    r = ag.update()
    print("step %2d tcg %3d outer %d acc %d status %-9s f %.10e -> %.10e gn %.6e -> %.6e" % (
        k, r.tcg_iterations, r.rtr_iterations, r.rtr_accepted, r.tCGStatus, r.fInit, r.fOpt, r.gradNormInit, r.gradNormOpt))
if len(sys.argv) > 3:
    print("== fixed-state repeats")
    ag.X.copy_(torch.tensor(X0, device="cuda"))
    for k in range(5):
        ag.update()
    ag.snapshot()
    for k in range(8):
        ag.restore()
        r = ag.update()
        print("rep %d tcg %3d outer %d acc %d status %-9s f %.12e -> %.12e gn %.6e ms %.3f" % (k, r.tcg_iterations, r.rtr_iterations, r.rtr_accepted, r.tCGStatus, r.fInit, r.fOpt, r.gradNormOpt, r.elapsedMs))
