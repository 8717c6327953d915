import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from oracle import dpgo_oracle as oracle
from conftest import DATA, to_product_measurements
from test_parity_gpu import _inject_outliers
import dpgo_amd
from dpgo_amd.agent import DeviceAgent, ExchangePlan, RBCDCluster, build_pose_graphs
from dpgo_amd.robust import DistributedGNC, RobustCostParameters
r, robots, k, sweeps = 5, 3, 10, 2
om, n = oracle.read_g2o(os.path.join(DATA, "smallGrid3D.g2o"))
allm = _inject_outliers(oracle, om, n, k, seed=7)
X0 = oracle.lift(oracle.chordal_initialization(om, n), r)
ref_meas = _inject_outliers(oracle, om, n, k, seed=7)
Xref, info_o = oracle.multi_agent_gnc(ref_meas, n, robots, r, X0, inner_sweeps=sweeps, barc=5.0, mu_step=1.4, max_updates=40, hess_recurrence=True)
pm = to_product_measurements(allm)
ranges, graphs = build_pose_graphs(pm, n, robots, r)
plan = ExchangePlan(graphs)
agents = {a: DeviceAgent(graphs, plan, a, X0[ranges[a][0]:ranges[a][1]], dpgo_amd.ROptParameters()) for a in range(robots)}
cluster = RBCDCluster(plan, agents)
gnc = DistributedGNC(cluster, RobustCostParameters("GNC_TLS", GNCMaxNumIters=40, GNCBarc=5.0, GNCMuStep=1.4), inner_sweeps=sweeps)
info = gnc.run()
print("muInit", info["muInit"], info_o["muInit"])
for i in range(max(len(info["history"]), len(info_o["history"]))):
    print(i, info["history"][i] if i < len(info["history"]) else None, info_o["history"][i] if i < len(info_o["history"]) else None)
print(info["cost"], info_o["cost"])
print("edges per agent", [len(g.measurements()) for g in graphs], "total", ref_meas.m)
