"""Device-resident counterpart of the part of PGOAgent that calls the hot path.

What the reference does per RBCD iteration (src/PGOAgent.cpp:376-432, 938-995):
  neighbours' public poses -> PoseGraph::setNeighborPoses -> constructG -> QuadraticProblem +
  QuadraticOptimizer::optimize(X0) -> X.
Here X, the neighbour tile buffer, G and Q stay in HBM; one agent maps to one GPU / one process
and the public-pose exchange (examples/MultiRobotExample.cpp:183-204 does it by pointer calls
inside one process) is carried by RCCL point-to-point send/recv (torch.distributed, backend "nccl").

The schedule for multi-GPU runs is the two-colour parallel RBCD of SURVEY 8e: agents of one colour
update simultaneously (iterate(true)), the others keep their iterate -- expressible with the
reference's unmodified iterate(bool) API.  "One RBCD iteration" = one sweep in which every agent
has updated once.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np

from . import lib as L
from .measurements import RelativeSEMeasurements, partition_contiguous
from .solver import PoseGraph, QuadraticOptimizer, QuadraticProblem, ROptParameters, ROPTResult


def build_pose_graphs(dataset: RelativeSEMeasurements, num_poses: int, num_robots: int, r: int):
    """Partition like examples/MultiRobotExample.cpp:71-146 and build one PoseGraph per robot."""
    ranges, per_robot = partition_contiguous(dataset, num_poses, num_robots)
    graphs = []
    for a in range(num_robots):
        pg = PoseGraph(a, r, dataset.d)
        pg.setMeasurements(per_robot[a])
        graphs.append(pg)
    return ranges, graphs


def greedy_colouring(graphs: Sequence[PoseGraph]) -> List[int]:
    """Colour the agent graph (agents adjacent iff they share a loop closure) greedily in id order;
    chain / even-ring partitions get two colours."""
    nbrs = [sorted({rob for rob, _ in g.neighborPoseIDs()}) for g in graphs]
    colour = [-1] * len(graphs)
    for a in range(len(graphs)):
        used = {colour[q] for q in nbrs[a] if colour[q] >= 0}
        c = 0
        while c in used:
            c += 1
        colour[a] = c
    return colour


class DeviceAgent:
    """One agent on one GPU: device-resident X, neighbour tile buffer and problem handle."""

    def __init__(self, graphs: Sequence[PoseGraph], my_id: int, X0_tiles: np.ndarray,
                 params: Optional[ROptParameters] = None, device: int = 0):
        import torch
        self.torch = torch
        self.id = my_id
        self.pg = graphs[my_id]
        self.r, self.d, self.n = self.pg.r(), self.pg.d(), self.pg.n()
        self.b = self.d + 1
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        if X0_tiles.shape != (self.n, self.b, self.r):
            raise ValueError("X0 tiles have shape %s, expected %s" % (X0_tiles.shape, (self.n, self.b, self.r)))
        self.problem = QuadraticProblem(self.pg, device=device, host_linear_term=False)
        self.problem.setStream(torch.cuda.current_stream().cuda_stream)
        self.optimizer = QuadraticOptimizer(self.problem, params or ROptParameters())
        self.X = torch.tensor(np.ascontiguousarray(X0_tiles), dtype=torch.float64, device=self.device)
        # neighbour slots: sorted (robot, frame) -> contiguous range per neighbour robot, frame order
        self.slots = self.problem.setCouplingFromPoseGraph() if self.has_neighbours else []
        self.nbr = torch.zeros((max(len(self.slots), 1), self.b, self.r), dtype=torch.float64, device=self.device)
        self.recv_range: Dict[int, tuple] = {}
        for k, (rob, _fr) in enumerate(self.slots):
            lo, hi = self.recv_range.get(rob, (k, k))
            self.recv_range[rob] = (min(lo, k), k + 1)
        # what each neighbour needs from me, in ITS slot order
        self.send_idx: Dict[int, "torch.Tensor"] = {}
        for q, g in enumerate(graphs):
            if q == my_id:
                continue
            frames = [fr for rob, fr in g.neighborPoseIDs() if rob == my_id]
            if frames:
                self.send_idx[q] = torch.tensor(frames, dtype=torch.int32, device=self.device)
        self.send_buf = {q: torch.empty((len(ix), self.b, self.r), dtype=torch.float64, device=self.device)
                         for q, ix in self.send_idx.items()}
        self.last_result: Optional[ROPTResult] = None

    @property
    def has_neighbours(self) -> bool:
        return len(self.pg.sharedLoopClosures()) > 0

    @property
    def neighbours(self) -> List[int]:
        return sorted(self.recv_range.keys())

    # ---- K11: pack / unpack of public poses (PGOAgent::getSharedPoseDict / updateNeighborPoses) ----
    def pack(self, q: int):
        ix, buf = self.send_idx[q], self.send_buf[q]
        L.check(self.problem._lib.dpgo_gather_tiles_device(self.r, self.d, L.ptr(self.X), L.ptr(ix), len(ix),
                                                           L.ptr(buf),
                                                           self.torch.cuda.current_stream().cuda_stream))
        return buf

    def recv_view(self, q: int):
        lo, hi = self.recv_range[q]
        return self.nbr[lo:hi]

    # ---- the hot path ----
    def update(self) -> ROPTResult:
        """PGOAgent::updateX(doOptimization = true): G from the neighbour buffer, then optimize."""
        if self.has_neighbours:
            self.problem.updateLinearMatrixFromNeighbors(self.nbr)
        self.last_result = self.optimizer.optimizeDevice(self.X)
        return self.last_result

    def local_terms(self):
        """(sum X_a Q_a . X_a, sum X_a . G_a, |rgrad_a|^2) with the current neighbour buffer: the
        central cost is 0.5 * sum_a (xqx_a + xg_a) and the central gradnorm^2 is sum_a |rgrad_a|^2
        (SURVEY 8c': the agent-local gradient is the agent's block of the central gradient)."""
        if self.has_neighbours:
            self.problem.updateLinearMatrixFromNeighbors(self.nbr)
        return self.problem.evalTermsDevice(self.X)


class RBCDCluster:
    """N agents, one per rank, exchanging public poses over torch.distributed (RCCL)."""

    def __init__(self, agent: DeviceAgent, graphs: Sequence[PoseGraph], rank: int, world: int):
        self.agent, self.rank, self.world = agent, rank, world
        self.colour = greedy_colouring(graphs)
        self.num_colours = max(self.colour) + 1
        self.adj = [sorted({rob for rob, _ in g.neighborPoseIDs()}) for g in graphs]

    def exchange(self, receivers: Optional[int] = None) -> None:
        """Public-pose exchange.  receivers = colour whose agents need fresh neighbour poses
        (None = everyone).  One grouped batch of isend/irecv per call (ncclGroupStart/End)."""
        if self.world == 1:
            return
        import torch.distributed as dist
        a = self.agent
        ops = []
        for q in self.adj[self.rank]:
            if receivers is None or self.colour[q] == receivers:
                ops.append(dist.P2POp(dist.isend, a.pack(q), q))
            if receivers is None or self.colour[self.rank] == receivers:
                ops.append(dist.P2POp(dist.irecv, a.recv_view(q), q))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()

    def sweep(self) -> None:
        """One RBCD iteration: every colour class updates once (parallel within a class)."""
        for c in range(self.num_colours):
            self.exchange(receivers=c)
            if self.colour[self.rank] == c:
                self.agent.update()

    def central_cost_and_gradnorm(self):
        import torch
        import torch.distributed as dist
        self.exchange(None)
        xqx, xg, g2 = self.agent.local_terms()
        t = torch.tensor([0.5 * (xqx + xg), g2], dtype=torch.float64, device=self.agent.device)
        if self.world > 1:
            dist.all_reduce(t)
        return float(t[0].item()), float(t[1].item()) ** 0.5
