"""Worker of test_two_processes_exchange_through_mapped_buffers_on_one_gpu (tests/test_parity_gpu.py): launched twice by
torch.distributed.run (gloo), both ranks on device 0.  4 agents, 2 per rank (one of each colour); the public-pose exchange
travels by the peer-store transport (dpgo_amd/ipc.py: the receivers' neighbour buffers mapped through hipIpc handles, the
senders' pack kernel writing straight into them).  Rank 0 then repeats the run in ONE process (device copies) and compares
bit for bit.  usage: ipc_worker.py <smallGrid3D | grid:NXxNYxNZ> <sweeps>"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    import torch
    import torch.distributed as dist
    import dpgo_amd
    import dpgo_oracle as O  # workload generation only
    from dpgo_amd.agent import DeviceAgent, ExchangePlan, RBCDCluster, build_pose_graphs
    from dpgo_amd.measurements import RelativeSEMeasurements
    name, sweeps = sys.argv[1], int(sys.argv[2])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    r, robots, apr = 5, 2 * world, 2
    if name.startswith("grid:"):
        om, n, Tt = O.synthetic_grid(*[int(v) for v in name[5:].split("x")], seed=0)
        X0 = O.lift(O.perturbed_truth(Tt, seed=2), r)
    else:
        om, n = O.read_g2o(os.path.join(ROOT, "data", name + ".g2o"))
        X0 = O.lift(O.chordal_initialization(om, n), r)
    c = np.copy
    pm = RelativeSEMeasurements(om.d, c(om.r1), c(om.p1), c(om.r2), c(om.p2), c(om.R), c(om.t), c(om.kappa), c(om.tau),
                                c(om.weight), c(om.fixed))
    ranges, graphs = build_pose_graphs(pm, n, robots, r)
    plan = ExchangePlan(graphs)
    prm = dpgo_amd.ROptParameters(precond="jacobi")
    mine = list(range(rank * apr, (rank + 1) * apr))
    agents = {a: DeviceAgent(graphs, plan, a, X0[ranges[a][0]:ranges[a][1]], prm) for a in mine}
    cluster = RBCDCluster(plan, agents, rank, world, agents_per_rank=apr)
    cluster.enable_peer_store()
    costs = []
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ex_ms = 0.0
    _exchange = cluster.exchange

    def timed(*a, **k):
        nonlocal ex_ms
        ev0.record()
        _exchange(*a, **k)
        ev1.record()
        ev1.synchronize()
        ex_ms += ev0.elapsed_time(ev1)
    cluster.exchange = timed
    for _ in range(sweeps):
        cluster.sweep()
        costs.append(cluster.central_cost_and_gradnorm())
    cluster.exchange = _exchange
    parts = [None] * world
    dist.all_gather_object(parts, {a: agents[a].X.cpu().numpy() for a in mine})
    its = [None] * world
    dist.all_gather_object(its, {a: (agents[a].last_result.tcg_iterations, agents[a].last_result.rtr_iterations) for a in mine})
    cluster.peer_store.close()
    del cluster, agents
    torch.cuda.synchronize()
    dist.barrier()
    if rank == 0:
        X_mp = {a: x for d_ in parts for a, x in d_.items()}
        it_mp = {a: v for d_ in its for a, v in d_.items()}
        ranges, graphs = build_pose_graphs(pm, n, robots, r)
        plan = ExchangePlan(graphs)
        ref_agents = {a: DeviceAgent(graphs, plan, a, X0[ranges[a][0]:ranges[a][1]], prm) for a in range(robots)}
        ref = RBCDCluster(plan, ref_agents)
        ref.concurrent = False  # one solve at a time, as each process of the two-process run issues them
        ref_costs = []
        for _ in range(sweeps):
            ref.sweep()
            ref_costs.append(ref.central_cost_and_gradnorm())
        same = all(np.array_equal(X_mp[a], ref_agents[a].X.cpu().numpy()) for a in range(robots))
        worst = max(float(np.abs(X_mp[a] - ref_agents[a].X.cpu().numpy()).max()) for a in range(robots))
        its_ref = {a: (ref_agents[a].last_result.tcg_iterations, ref_agents[a].last_result.rtr_iterations) for a in range(robots)}
        print("IPC_RESULT bit_identical=%d worst_abs_diff=%.3e costs_equal=%d iterations_equal=%d cost=%r decrease=%d "
              "exchange_ms_per_sweep=%.4f" % (int(same), worst, int(costs == ref_costs), int(it_mp == its_ref), costs[-1][0],
                                              int(costs[-1][0] < costs[0][0] or sweeps == 1), ex_ms / sweeps), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
