#!/usr/bin/env python3
"""Multi-robot pose graph optimization example on one MI355X -- the counterpart of the reference's
examples/MultiRobotExample.cpp: contiguous partition into N robots (:71-119), greedy block selection with Nesterov
acceleration (:170-255), one line per iteration in the reference's format, then the rounded trajectory of every robot in
the frame of robot 0's first pose (the global anchor, :249-254) is written as CSV (PGOLogger::logTrajectory format).

  python examples/multi_robot_example.py 5 data/smallGrid3D.g2o [--out-dir /tmp/traj] [--no-acceleration]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("num_robots", type=int)
    ap.add_argument("g2o")
    ap.add_argument("--rank", type=int, default=5)
    ap.add_argument("--iterations", type=int, default=1000)
    ap.add_argument("--no-acceleration", action="store_true")
    ap.add_argument("--out-dir", default=None)
    args = ap.parse_args()
    print("Multi-robot pose graph optimization example. ")
    if args.num_robots <= 0:
        raise SystemExit("Number of robots must be positive!")
    print("Simulating %d robots." % args.num_robots)
    import numpy as np
    import dpgo_amd
    from dpgo_amd.agent import DeviceAgent, ExchangePlan, RBCDCluster, build_pose_graphs
    from dpgo_amd.initialization import chordal_initialization
    from dpgo_amd.synthetic import lift_tiles
    from dpgo_amd.trajectory import log_trajectory

    meas, n = dpgo_amd.read_g2o_file(args.g2o)
    print("Loaded dataset from file %s." % args.g2o)
    r, d = args.rank, meas.d
    X0 = lift_tiles(chordal_initialization(meas, n), r)  # PGOAgent's default: chordal initialisation
    ranges, graphs = build_pose_graphs(meas, n, args.num_robots, r)
    plan = ExchangePlan(graphs)
    agents = {a: DeviceAgent(graphs, plan, a, X0[ranges[a][0]:ranges[a][1]], dpgo_amd.ROptParameters())
              for a in range(args.num_robots)}
    if not args.no_acceleration:
        for ag in agents.values():
            ag.enable_acceleration(args.num_robots)
    cluster = RBCDCluster(plan, agents)
    print("Running %d iterations..." % args.iterations)
    out = cluster.run_greedy(max_iters=args.iterations, gradnorm_stop=0.1)
    for it, (rob, (cost, gn)) in enumerate(zip(out["selected"], out["trace"])):
        print("Iter = %d | robot = %d | cost = %.5g | gradnorm = %.5g" % (it, rob, cost, gn))
    if args.out_dir:
        os.makedirs(args.out_dir, exist_ok=True)
        for a, T in cluster.trajectories_in_global_frame().items():
            Tm = np.ascontiguousarray(T.cpu().numpy()).reshape(-1, d).T  # tiles [n, d+1, d] -> d x (d+1)n
            if log_trajectory(d, graphs[a].n(), np.asfortranarray(Tm), os.path.join(args.out_dir, "robot%d.csv" % a)):
                print("wrote %s" % os.path.join(args.out_dir, "robot%d.csv" % a))


if __name__ == "__main__":
    main()
