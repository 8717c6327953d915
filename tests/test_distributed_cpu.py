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


def _status_mixin():
    from dpgo_amd.agent import AgentStatusMixin
    return AgentStatusMixin


class HostAgent(_status_mixin()):
    """CPU stand-in for DeviceAgent with the same interface (id, pack, recv_view, update, local_terms, status)."""

    def __init__(self, O, plan, my_id, om_local, X0_tiles, r, d):
        import torch
        self.torch = torch
        self.O, self.plan, self.id, self.r, self.d = O, plan, my_id, r, d
        self.X = torch.tensor(np.ascontiguousarray(X0_tiles))
        self.b = d + 1
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
        self.last_result = opt.result

    def measure_relative_change(self, stream=None):
        self.rel_dev[0] = self.O.max_translation_distance(self.X.numpy(), self.XPrev.numpy())

    def getTrajectoryInGlobalFrame(self, anchor):
        import torch
        return torch.tensor(self.O.round_trajectory(self.X.numpy(), self.d, np.ascontiguousarray(np.asarray(anchor).T)))

    def local_terms(self):
        p = self._problem()
        X = self.X.numpy()
        xqx = float(np.sum(p.XQ(X) * X))
        xg = float(np.sum(X * p.G))
        return xqx, xg, p.rie_grad_norm(X) ** 2


def _worker(rank, world, port, sweeps, out_dir, apr=1):
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
        num_agents = world * apr
        ranges, graphs = build_pose_graphs(to_product_measurements(om), n, num_agents, r)
        oranges, per = O.partition_contiguous(om, n, num_agents)
        assert ranges == oranges
        plan = ExchangePlan(graphs)
        assert plan.num_colours == 2
        if apr == 1:
            assert plan.adj == [[1], [0]]
        mine = list(range(rank * apr, (rank + 1) * apr))  # consecutive agents share a rank (RBCDCluster.owner)
        local = {a: HostAgent(O, plan, a, per[a], X0[ranges[a][0]:ranges[a][1]], r, d) for a in mine}
        cluster = RBCDCluster(plan, local, rank, world, agents_per_rank=apr)
        assert all(cluster.owner(a) == a // apr for a in range(num_agents))
        f0, g0 = cluster.central_cost_and_gradnorm()
        trace = [(2 * f0, g0)]
        for _ in range(sweeps):
            cluster.sweep()
            f, g = cluster.central_cost_and_gradnorm()
            trace.append((2 * f, g))
        s, e = ranges[mine[0]][0], ranges[mine[-1]][1]
        X = np.concatenate([local[a].X.numpy() for a in mine], axis=0)
        # rounding in the frame of agent 0's first pose: the anchor is broadcast from its owner
        anchor = cluster.global_anchor()
        traj = cluster.trajectories_in_global_frame()
        T = np.concatenate([traj[a].numpy() for a in mine], axis=0)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), X=X, trace=np.array(trace), s=s, e=e, anchor=anchor, T=T)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("apr", [1, 2])
def test_two_rank_gloo_rbcd_matches_single_process_oracle(oracle, tmp_path, apr):
    """apr = agents per rank: 1 (one agent per process) and 2 (bench.py's layout for N > 1 GPUs: two consecutive
    agents of different colours per process; local pairs exchange by copies, remote pairs by grouped p2p)."""
    import torch.multiprocessing as mp
    O = oracle
    sweeps = 3
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(2, port, sweeps, str(tmp_path), apr), nprocs=2, join=True)
    om, n, Ttrue = O.synthetic_grid(6, 5, 4, seed=3)
    X0 = O.lift(O.perturbed_truth(Ttrue, seed=4), 5)
    central = O.QuadraticProblem(O.construct_Q(n, 3, om), None, 5, 3)
    Xref, costs, gns = O.rbcd_coloured(om, n, 2 * apr, 5, X0, sweeps)
    X = np.zeros_like(Xref)
    traces = []
    for rank in range(2):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        X[int(z["s"]):int(z["e"])] = z["X"]
        traces.append(z["trace"])
    assert np.allclose(traces[0], traces[1], rtol=1e-13)  # both ranks see the same central numbers
    z0, z1 = (np.load(os.path.join(str(tmp_path), "rank%d.npz" % k)) for k in range(2))
    assert np.array_equal(z0["anchor"], z1["anchor"]) and np.array_equal(z0["anchor"].T, z0["X"][0])
    Tall = np.concatenate([z0["T"], z1["T"]], axis=0)  # PGOAgent::getTrajectoryInGlobalFrame on every agent
    assert np.abs(Tall - O.round_trajectory(X, 3, X[0])).max() <= 1e-12
    assert np.abs(Tall[0, :3] - np.eye(3)).max() < 1e-12 and np.abs(Tall[0, 3]).max() < 1e-12
    assert np.abs(X - Xref).max() <= 1e-12
    assert abs(traces[0][0, 0] - 2 * central.f(X0)) <= 1e-10 * abs(2 * central.f(X0))
    assert abs(traces[0][0, 1] - central.rie_grad_norm(X0)) <= 1e-10 * central.rie_grad_norm(X0)
    for k in range(sweeps):
        assert abs(traces[0][k + 1, 0] - costs[k]) <= 1e-10 * abs(costs[k])
        assert abs(traces[0][k + 1, 1] - gns[k]) <= 1e-8 * gns[k]
    assert costs[-1] < costs[0] < 2 * central.f(X0)


def _status_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import dpgo_oracle as O
    from dpgo_amd.agent import ExchangePlan, PGOAgentParameters, RBCDCluster, build_pose_graphs
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        om, n, Ttrue = O.synthetic_grid(6, 5, 4, seed=3)
        r, d, apr = 5, 3, 2
        X0 = O.lift(O.perturbed_truth(Ttrue, seed=4), r)
        ranges, graphs = build_pose_graphs(to_product_measurements(om), n, world * apr, r)
        _, per = O.partition_contiguous(om, n, world * apr)
        plan = ExchangePlan(graphs)
        mine = list(range(rank * apr, (rank + 1) * apr))
        local = {a: HostAgent(O, plan, a, per[a], X0[ranges[a][0]:ranges[a][1]], r, d) for a in mine}
        cluster = RBCDCluster(plan, local, rank, world, agents_per_rank=apr)
        out = cluster.run_until_terminated(PGOAgentParameters(relChangeTol=2e-2, maxNumIters=60))
        X = np.concatenate([local[a].X.numpy() for a in mine], axis=0)
        np.savez(os.path.join(out_dir, "status%d.npz" % rank), X=X, s=ranges[mine[0]][0], e=ranges[mine[-1]][1],
                 iterations=out["iterations"],
                 ready=np.array([out["statuses"][a].readyToTerminate for a in range(world * apr)]),
                 rel=np.array([out["statuses"][a].relativeChange for a in range(world * apr)]),
                 its=np.array([out["statuses"][a].iterationNumber for a in range(world * apr)]))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_termination_vote_matches_oracle(oracle, tmp_path):
    """PGOAgent status + shouldTerminate (src/PGOAgent.cpp:399-420, 846-878) over two processes: every agent's
    relativeChange / readyToTerminate travels to every rank (one small all-reduce per global iteration), all ranks
    stop at the same iteration, and iteration count, statuses and iterate equal the single-process oracle driver's."""
    import torch.multiprocessing as mp
    O = oracle
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_status_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    om, n, Ttrue = O.synthetic_grid(6, 5, 4, seed=3)
    X0 = O.lift(O.perturbed_truth(Ttrue, seed=4), 5)
    Xref, info = O.rbcd_coloured_until_terminated(om, n, 4, 5, X0, O.AgentParameters(relChangeTol=2e-2, maxNumIters=60))
    z = [np.load(os.path.join(str(tmp_path), "status%d.npz" % k)) for k in range(2)]
    assert int(z[0]["iterations"]) == int(z[1]["iterations"]) == info["iterations"] < 60
    X = np.zeros_like(Xref)
    for k in range(2):
        X[int(z[k]["s"]):int(z[k]["e"])] = z[k]["X"]
        assert z[k]["ready"].all()
        assert [int(v) for v in z[k]["its"]] == [info["statuses"][a].iterationNumber for a in range(4)]
        assert np.allclose(z[k]["rel"], [info["statuses"][a].relativeChange for a in range(4)], rtol=1e-9, atol=1e-14)
    assert np.abs(X - Xref).max() <= 1e-12


def test_status_rules_follow_the_reference(oracle):
    """The status / vote rules restated from src/PGOAgent.cpp:399-420, 846-878, 997-1045 on hand-made cases, oracle
    and product (dpgo_amd.agent) side by side."""
    from dpgo_amd import agent as A
    O = oracle
    X = np.zeros((3, 4, 5))
    Xp = X.copy()
    Xp[1, 3] = [3.0, 4.0, 0.0, 0.0, 0.0]   # translation of pose 1 moved by 5
    Xp[2, 0] = 7.0                          # a rotation column does not count (Poses.cpp:86-94)
    assert O.max_translation_distance(X, Xp) == 5.0
    prm = O.AgentParameters()
    assert O.local_status(0, 3, X, X, True, prm).readyToTerminate
    assert not O.local_status(0, 3, X, Xp, True, prm).readyToTerminate            # 5 > 5e-3
    assert not O.local_status(0, 3, X, X, False, prm).readyToTerminate            # failed solve
    rob = O.AgentParameters(robust=True)
    assert O.local_status(0, 3, X, Xp, True, rob, 0).readyToTerminate             # loose threshold 5 before update 1
    assert not O.local_status(0, 3, X, Xp, True, rob, 1).readyToTerminate
    assert not O.local_status(0, 3, X, X, True, rob, 1, [1, 0, 0.5, 0.5]).readyToTerminate   # ratio 0.5 < 0.8
    assert O.local_status(0, 3, X, X, True, rob, 1, [1, 0, 1, 1, 0.5]).readyToTerminate       # ratio 0.8
    for mod, St, Prm, term, upd in ((O, O.AgentStatus, O.AgentParameters, O.should_terminate, O.should_update_weights),
                                    (A, A.PGOAgentStatus, A.PGOAgentParameters, A.should_terminate,
                                     A.should_update_measurement_weights)):
        ready = {a: St(a, "INITIALIZED", 0, 4, True, 1e-4) for a in range(3)}
        assert term(4, Prm(), 0, ready, 3)
        assert not term(4, Prm(), 0, {a: ready[a] for a in range(2)}, 3)           # a status is missing
        assert term(500, Prm(), 0, {}, 3)                                          # maxNumIters
        notyet = dict(ready)
        notyet[1] = St(1, "INITIALIZED", 0, 4, False, 1.0)
        assert not term(4, Prm(), 0, notyet, 3)
        assert not term(4, Prm(robust=True), 3, ready, 3) and term(4, Prm(robust=True), 10, ready, 3)
        assert not upd(Prm(), 0, 99, 0, ready, 3)                                  # L2: never
        assert upd(Prm(robust=True), 0, 30, 0, {}, 3)                              # inner iterations exhausted
        assert upd(Prm(robust=True), 0, 1, 4, ready, 3) and not upd(Prm(robust=True), 0, 1, 5, ready, 3)  # outdated
        assert not upd(Prm(robust=True), 10, 30, 0, ready, 3) and not upd(Prm(robust=True), 0, 1, 0, notyet, 3)


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


# ---------------------------------------------------------------------------------------------
# Distributed GNC over two gloo ranks (driver: dpgo_amd.robust.DistributedGNC; per-agent work: oracle)
# ---------------------------------------------------------------------------------------------
class HostGncProblem:
    """CPU stand-in for the GNC methods of dpgo_amd.QuadraticProblem, backed by the oracle's formulas."""

    def __init__(self, agent):
        self.a = agent
        self.reweightable_index = None

    def _all(self):
        a = self.a
        return a.O.Measurements.concat([a.odo, a.priv, a.shared])

    def setReweightableEdges(self, include_shared=False):
        assert include_shared
        self.reweightable_index = np.arange(self._all().m)
        return len(self.reweightable_index)

    def setEdgeWeights(self, w):
        a = self.a
        k1, k2 = a.odo.m, a.odo.m + a.priv.m
        a.odo.weight[:], a.priv.weight[:], a.shared.weight[:] = w[:k1], w[k1:k2], w[k2:]

    def getEdgeWeights(self):
        m = self._all()
        return m.weight.copy(), np.zeros(m.m)

    def gncReweightDevice(self, X, nbr, mu, barc, w_tol=1e-8, update=True):
        a, O = self.a, self.a.O
        m = self._all()
        d = m.d
        Xn, slot = X.numpy(), {pid: k for k, pid in enumerate(a.plan.slots[a.id])}
        rsq = np.zeros(m.m)
        for e in range(m.m):
            mine1, mine2 = m.r1[e] == a.id, m.r2[e] == a.id
            xi = Xn[m.p1[e]] if mine1 else nbr[slot[(int(m.r1[e]), int(m.p1[e]))]].numpy()
            xj = Xn[m.p2[e]] if mine2 else nbr[slot[(int(m.r2[e]), int(m.p2[e]))]].numpy()
            Yi, Yj = xi[:d].T, xj[:d].T
            rsq[e] = m.kappa[e] * np.sum((Yi @ m.R[e] - Yj) ** 2) + m.tau[e] * np.sum((xj[d] - xi[d] - Yi @ m.t[e]) ** 2)
        w = m.weight.copy()
        if update:
            nf = ~m.fixed
            w[nf] = O.gnc_tls_weight(np.sqrt(rsq[nf]), mu, barc)
            self.setEdgeWeights(w)
        counted = (~m.fixed) & (m.r1 == a.id)  # a shared edge is counted by the owner of its source pose
        wc = w[counted]
        n_out, n_in = int((wc < w_tol).sum()), int((wc > 1 - w_tol).sum())
        return (n_in, n_out, len(wc) - n_in - n_out), float(rsq.max())


class HostGncAgent(HostAgent):
    def __init__(self, O, plan, my_id, om_local, X0_tiles, r, d):
        super().__init__(O, plan, my_id, om_local, X0_tiles, r, d)
        self.odo, self.priv = om_local["odometry"], om_local["private"]
        self.has_neighbours = True
        self.problem = HostGncProblem(self)

    def _problem(self):  # Q is rebuilt from the current weights (PoseGraph::clearDataMatrices after a weight update)
        O = self.O
        self.Q = O.construct_Q(self.X.shape[0], self.d, O.Measurements.concat([self.odo, self.priv]), self.shared,
                               my_id=self.id)
        return super()._problem()


def _gnc_case(O):
    om, n, Ttrue = O.synthetic_grid(5, 4, 3, seed=5)
    rng = np.random.default_rng(9)
    k, d = 6, 3
    taken = set(zip(om.p1.tolist(), om.p2.tolist()))
    p1, p2 = [], []
    while len(p1) < k:
        a = int(rng.integers(0, n))
        b = int((a + rng.integers(3, n - 3)) % n)
        if (a, b) not in taken:
            taken.add((a, b))
            p1.append(a)
            p2.append(b)
    Rs = np.stack([np.linalg.qr(rng.standard_normal((d, d)))[0] for _ in range(k)])
    for q in range(k):
        if np.linalg.det(Rs[q]) < 0:
            Rs[q][:, 0] *= -1
    z = np.zeros(k, dtype=np.int64)
    out = O.Measurements(d, z, np.array(p1), z.copy(), np.array(p2), Rs, rng.uniform(-5, 5, (k, d)),
                         np.full(k, np.median(om.kappa)), np.full(k, np.median(om.tau)), np.ones(k),
                         np.zeros(k, dtype=bool))
    return O.Measurements.concat([om, out]), n, O.lift(O.perturbed_truth(Ttrue, seed=6), 5), om.m


def _gnc_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import dpgo_oracle as O
    from dpgo_amd.agent import ExchangePlan, RBCDCluster, build_pose_graphs
    from dpgo_amd.robust import DistributedGNC, RobustCostParameters
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        allm, n, X0, _ = _gnc_case(O)
        ranges, graphs = build_pose_graphs(to_product_measurements(allm), n, world, 5)
        _, per = O.partition_contiguous(allm, n, world)
        plan = ExchangePlan(graphs)
        s, e = ranges[rank]
        agent = HostGncAgent(O, plan, rank, per[rank], X0[s:e], 5, 3)
        cluster = RBCDCluster(plan, {rank: agent}, rank, world)
        gnc = DistributedGNC(cluster, RobustCostParameters("GNC_TLS", GNCMaxNumIters=40, GNCBarc=5.0, GNCMuStep=1.4),
                             inner_sweeps=2)
        info = gnc.run()
        hist = np.array([[h["mu"], h["inliers"], h["outliers"], h["undecided"]] for h in info["history"]])
        np.savez(os.path.join(out_dir, "gnc%d.npz" % rank), X=agent.X.numpy(), hist=hist, s=s, e=e,
                 muInit=info["muInit"], cost=info["cost"], updates=info["updates"])
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_distributed_gnc_matches_oracle(oracle, tmp_path):
    """The N > 1 path of DistributedGNC (global max residual and classification counters by all-reduce, shared
    edges counted once, every rank taking the same decisions) against oracle.multi_agent_gnc."""
    import torch.multiprocessing as mp
    O = oracle
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_gnc_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    allm, n, X0, m_clean = _gnc_case(O)
    Xref, info_o = O.multi_agent_gnc(allm, n, 2, 5, X0, inner_sweeps=2, barc=5.0, mu_step=1.4, max_updates=40)
    z = [np.load(os.path.join(str(tmp_path), "gnc%d.npz" % k)) for k in range(2)]
    assert np.array_equal(z[0]["hist"], z[1]["hist"])  # both ranks took identical decisions
    assert int(z[0]["updates"]) == info_o["updates"]
    assert abs(float(z[0]["muInit"]) - info_o["muInit"]) <= 1e-12 * info_o["muInit"]
    ho = np.array([[h["mu"], h["inliers"], h["outliers"], h["undecided"]] for h in info_o["history"]])
    assert np.allclose(z[0]["hist"], ho, rtol=1e-12)
    X = np.zeros_like(Xref)
    for k in range(2):
        X[int(z[k]["s"]):int(z[k]["e"])] = z[k]["X"]
    assert np.abs(X - Xref).max() <= 1e-9
    assert abs(float(z[0]["cost"]) - info_o["cost"]) <= 1e-9 * info_o["cost"]
    assert info_o["history"][-1]["undecided"] == 0 and np.all(allm.weight[m_clean:] < 1e-8)
