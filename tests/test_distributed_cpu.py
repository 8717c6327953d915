"""N > 1 path on CPU: the exchange plan, the grouped isend/irecv schedule, the colour sweep and the
central-cost reduction of dpgo_amd.agent, driven over torch.distributed (gloo, world_size 2,
127.0.0.1).  The per-agent local solve is done by the CPU oracle here (no GPU in this container);
the orchestration code under test is the product's.  The result must equal the single-process
oracle driver (oracle.rbcd_coloured)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, to_product_measurements


class HostAgent:
    """CPU stand-in for DeviceAgent with the same interface (id, pack, recv_view, update, local_terms)."""

    def __init__(self, O, plan, my_id, om_local, X0_tiles, r, d):
        import torch
        self.O, self.plan, self.id, self.r, self.d = O, plan, my_id, r, d
        self.X = torch.tensor(np.ascontiguousarray(X0_tiles))
        self.nbr = torch.zeros((max(len(plan.slots[my_id]), 1), d + 1, r), dtype=torch.float64)
        priv = O.Measurements.concat([om_local["odometry"], om_local["private"]])
        self.shared = om_local["shared"]
        self.Q = O.construct_Q(X0_tiles.shape[0], d, priv, self.shared, my_id=my_id)
        self.device = "cpu"

    def pack(self, q, aux=False):
        import torch
        return self.X[torch.tensor(self.plan.send_frames[self.id][q], dtype=torch.long)].contiguous()

    def recv_view(self, q, aux=False):
        lo, hi = self.plan.recv_range[self.id][q]
        return self.nbr[lo:hi]

    def _problem(self):
        nbr = {pid: self.nbr[k].numpy() for k, pid in enumerate(self.plan.slots[self.id])}
        G = self.O.construct_G(self.X.shape[0], self.d, self.r, self.shared, self.id, nbr)
        return self.O.QuadraticProblem(self.Q, G, self.r, self.d, precond="jacobi")

    def update(self):
        opt = self.O.QuadraticOptimizer(self._problem(), self.O.ROptParameters())
        self.X.copy_(__import__("torch").tensor(opt.optimize(self.X.numpy().copy())))

    def local_terms(self):
        p = self._problem()
        X = self.X.numpy()
        xqx = float(np.sum(p.XQ(X) * X))
        xg = float(np.sum(X * p.G))
        return xqx, xg, p.rie_grad_norm(X) ** 2


def _worker(rank, world, port, sweeps, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import dpgo_oracle as O
    from dpgo_amd.agent import ExchangePlan, RBCDCluster, build_pose_graphs
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        om, n, Ttrue = O.synthetic_grid(6, 5, 4, seed=3)
        r, d = 5, 3
        X0 = O.lift(O.perturbed_truth(Ttrue, seed=4), r)
        ranges, graphs = build_pose_graphs(to_product_measurements(om), n, world, r)
        oranges, per = O.partition_contiguous(om, n, world)
        assert ranges == oranges
        plan = ExchangePlan(graphs)
        assert plan.num_colours == 2 and plan.adj == [[1], [0]]
        s, e = ranges[rank]
        agent = HostAgent(O, plan, rank, per[rank], X0[s:e], r, d)
        cluster = RBCDCluster(plan, {rank: agent}, rank, world)
        f0, g0 = cluster.central_cost_and_gradnorm()
        trace = [(2 * f0, g0)]
        for _ in range(sweeps):
            cluster.sweep()
            f, g = cluster.central_cost_and_gradnorm()
            trace.append((2 * f, g))
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), X=agent.X.numpy(), trace=np.array(trace), s=s, e=e)
    finally:
        dist.destroy_process_group()


def _from_conftest():
    pass


def test_two_rank_gloo_rbcd_matches_single_process_oracle(oracle, tmp_path):
    import torch.multiprocessing as mp
    O = oracle
    sweeps = 3
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(2, port, sweeps, str(tmp_path)), nprocs=2, join=True)
    om, n, Ttrue = O.synthetic_grid(6, 5, 4, seed=3)
    X0 = O.lift(O.perturbed_truth(Ttrue, seed=4), 5)
    central = O.QuadraticProblem(O.construct_Q(n, 3, om), None, 5, 3)
    Xref, costs, gns = O.rbcd_coloured(om, n, 2, 5, X0, sweeps)
    X = np.zeros_like(Xref)
    traces = []
    for rank in range(2):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        X[int(z["s"]):int(z["e"])] = z["X"]
        traces.append(z["trace"])
    assert np.allclose(traces[0], traces[1], rtol=1e-13)  # both ranks see the same central numbers
    assert np.abs(X - Xref).max() <= 1e-12
    assert abs(traces[0][0, 0] - 2 * central.f(X0)) <= 1e-10 * abs(2 * central.f(X0))
    assert abs(traces[0][0, 1] - central.rie_grad_norm(X0)) <= 1e-10 * central.rie_grad_norm(X0)
    for k in range(sweeps):
        assert abs(traces[0][k + 1, 0] - costs[k]) <= 1e-10 * abs(costs[k])
        assert abs(traces[0][k + 1, 1] - gns[k]) <= 1e-8 * gns[k]
    assert costs[-1] < costs[0] < 2 * central.f(X0)


def test_exchange_plan_structure(oracle):
    """Slots are sorted (robot, frame); per-neighbour ranges are contiguous; send lists mirror the
    receivers' slot order; message list is symmetric; colours are proper."""
    from dpgo_amd.agent import ExchangePlan, build_pose_graphs
    import dpgo_amd
    pm, n = dpgo_amd.read_g2o_file(os.path.join(ROOT, "data", "torus3D.g2o"))
    for k in (2, 3, 8):
        ranges, graphs = build_pose_graphs(pm, n, k, 5)
        plan = ExchangePlan(graphs)
        for a in range(k):
            assert plan.slots[a] == sorted(plan.slots[a])
            for q, (lo, hi) in plan.recv_range[a].items():
                assert all(rob == q for rob, _ in plan.slots[a][lo:hi])
                assert plan.send_frames[q][a] == [fr for _, fr in plan.slots[a][lo:hi]]
                assert a in plan.adj[q] and plan.colour[a] != plan.colour[q]
        msgs = plan.messages(None)
        assert sorted(msgs) == sorted((q, a) for a, q in msgs)
        per_colour = sum((plan.messages(c) for c in range(plan.num_colours)), [])
        assert sorted(per_colour) == sorted(msgs)
    assert ExchangePlan(build_pose_graphs(pm, n, 8, 5)[1]).num_colours == 2  # ring of 8 agents
