#!/usr/bin/env python3
"""Single-robot example (reference: examples/SingleRobotExample.cpp): one agent owns the whole graph; chordal
initialisation, repeated local solves on the device until the gradient norm is below 0.1, rounding, optional CSV.

  python examples/single_robot_example.py data/sphere2500.g2o [--robust] [--out traj.csv]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("g2o")
    ap.add_argument("--rank", type=int, default=5)
    ap.add_argument("--iterations", type=int, default=100)
    ap.add_argument("--robust", action="store_true", help="GNC-TLS re-weighting (solveRobustPGO, rank d)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import numpy as np
    import torch
    import dpgo_amd
    from dpgo_amd.initialization import chordal_initialization
    from dpgo_amd.synthetic import lift_tiles
    from dpgo_amd.trajectory import log_trajectory, round_trajectory_device

    meas, n = dpgo_amd.read_g2o_file(args.g2o)
    d = meas.d
    print("Loaded %d poses, %d measurements from %s" % (n, len(meas), args.g2o))
    if args.robust:
        from dpgo_amd.robust import solveRobustPGO
        T, info = solveRobustPGO(meas, n)
        rejected = int((meas.weight < 1e-8).sum())
        print("GNC: %d outer iterations, %d edges rejected, cost = %.6g" % (info["gnc_iterations"], rejected,
                                                                        2 * info["fOpt"]))
        X = torch.tensor(T, dtype=torch.float64, device="cuda")
    else:
        r = args.rank
        pg = dpgo_amd.PoseGraph(0, r, d)
        pg.setMeasurements(meas)
        problem = dpgo_amd.QuadraticProblem(pg)
        problem.setStream(torch.cuda.current_stream().cuda_stream)
        optimizer = dpgo_amd.QuadraticOptimizer(problem, dpgo_amd.ROptParameters())
        X = torch.tensor(lift_tiles(chordal_initialization(meas, n), r), dtype=torch.float64, device="cuda")
        for it in range(args.iterations):
            res = optimizer.optimizeDevice(X)
            print("Iter = %d | cost = %.8g | gradnorm = %.5g | tCG = %d" % (it, 2 * res.fOpt, res.gradNormOpt,
                                                                            res.tcg_iterations))
            if res.gradNormOpt < 0.1:
                break
    if args.out:
        T = round_trajectory_device(X).cpu().numpy()
        Tm = np.asfortranarray(np.ascontiguousarray(T).reshape(-1, d).T)
        if log_trajectory(d, n, Tm, args.out):
            print("wrote %s" % args.out)


if __name__ == "__main__":
    main()
