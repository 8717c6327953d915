"""Device-resident counterpart of the part of PGOAgent that calls the hot path, and the
public-pose exchange between agents.

What the reference does per RBCD iteration (src/PGOAgent.cpp:376-432, 938-995):
  neighbours' public poses -> PoseGraph::setNeighborPoses -> constructG -> QuadraticProblem +
  QuadraticOptimizer::optimize(X0) -> X.
Here X, the neighbour tile buffer, G and Q stay in HBM; one agent maps to one GPU / one process
and the public-pose exchange (examples/MultiRobotExample.cpp:183-204 does it by pointer calls
inside one process; the wire unit is the LiftedPose, r x (d+1) doubles keyed by PoseID) is carried
by RCCL point-to-point send/recv issued by the solver library itself on the solver's stream
(dpgo_amd/comm.py -> C ABI dpgo_comm_exchange: one grouped ncclSend / ncclRecv batch per exchange, no host
wait; torch.distributed is the rendezvous and the fallback transport).  The same classes also run N agents inside one process on one GPU (device-to-device
copies instead of RCCL) -- used by the single-GPU parity tests.

Schedule for parallel runs: the two-colour RBCD of SURVEY 8e -- agents of one colour update
simultaneously (PGOAgent::iterate(true)), the others keep their iterate -- expressible with the
reference's unmodified iterate(bool) API.  "One RBCD iteration" = one sweep in which every agent has
updated once.
"""
from __future__ import annotations

from dataclasses import dataclass, replace
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import lib as L
from .measurements import RelativeSEMeasurements, partition_contiguous
from .solver import (PoseGraph, QuadraticOptimizer, QuadraticProblem, ROptParameters, ROPTResult,
                     eval_terms_device_many, optimize_device_many)


@dataclass
class PGOAgentParameters:
    """The fields of DPGO::PGOAgentParameters that the status / termination rules read, with the reference's
    defaults (include/DPGO/PGOAgent.h:121-137)."""
    robust: bool = False  # robustCostParams.costType != L2
    robustOptNumWeightUpdates: int = 10
    robustOptInnerIters: int = 30
    robustOptMinConvergenceRatio: float = 0.8
    maxNumIters: int = 500
    relChangeTol: float = 5e-3


@dataclass
class PGOAgentStatus:
    """DPGO::PGOAgentStatus (include/DPGO/PGOAgent.h:196-227)."""
    agentID: int = 0
    state: str = "WAIT_FOR_DATA"
    instanceNumber: int = 0
    iterationNumber: int = 0
    readyToTerminate: bool = False
    relativeChange: float = 0.0


def should_terminate(iteration: int, prm: PGOAgentParameters, weight_update_count: int,
                     team: Dict[int, PGOAgentStatus], num_robots: int, inactive=()) -> bool:
    """PGOAgent::shouldTerminate (src/PGOAgent.cpp:846-878); robots in `inactive` (PGOAgent::setRobotActive(id, false),
    :1173-1184) have no vote (:861-862)."""
    if iteration >= prm.maxNumIters:
        return True
    if prm.robust and weight_update_count < prm.robustOptNumWeightUpdates:
        return False
    for rob in range(num_robots):
        if rob in inactive:
            continue
        st = team.get(rob)
        if st is None or st.state != "INITIALIZED" or not st.readyToTerminate:
            return False
    return True


def should_update_measurement_weights(prm: PGOAgentParameters, weight_update_count: int, inner_iter: int,
                                      latest_update_iteration: int, team: Dict[int, PGOAgentStatus],
                                      num_robots: int, inactive=()) -> bool:
    """PGOAgent::shouldUpdateMeasurementWeights (src/PGOAgent.cpp:997-1045); inactive robots are skipped (:1016-1017)."""
    if not prm.robust:
        return False
    if weight_update_count >= prm.robustOptNumWeightUpdates:
        return False
    if inner_iter >= prm.robustOptInnerIters:
        return True
    for rob in range(num_robots):
        if rob in inactive:
            continue
        st = team.get(rob)
        if st is None or st.iterationNumber < latest_update_iteration or st.state != "INITIALIZED" \
                or not st.readyToTerminate:
            return False
    return True


def build_pose_graphs(dataset: RelativeSEMeasurements, num_poses: int, num_robots: int, r: int, reorder=False):
    """Partition like examples/MultiRobotExample.cpp:71-146 and build one PoseGraph per robot.

    reorder (False; True, or the environment's DPGO_REORDER=1 with the default): the agent layer renumbers the poses
    INSIDE every robot's block for locality (measurements.locality_order) before the graphs are built -- every id that
    crosses the agent's boundary keeps the caller's numbering: DeviceAgent takes X0 and returns iterates / trajectories in
    the caller's order (pose_order of the graph = internal row of every caller frame), the exchange plan is built from
    the renumbered graphs on every agent alike, so public poses meet their users whatever they are called inside.
    OFF by default: measured on the 100k-pose lattice (round 4, tools/reorder_ab.sh, rotating operands) the odometry
    ("snake") order is already the best of those tried -- k_tcg_hess_sym 41.9 us as given, 45.8 / 44.7 / 42.8 us with
    reverse Cuthill-McKee over runs of 16 / 64 / 250 consecutive poses inside each XCD's eighth, 45.2 us pose by pose
    (DESIGN.md section 3)."""
    order = None
    import os
    if reorder is False and os.environ.get("DPGO_REORDER") == "1":  # A/B switch for the benchmarks
        reorder = True
    if reorder is True:
        from .measurements import locality_order, relabel
        order = locality_order(dataset, num_poses, num_robots)
        dataset = relabel(dataset, order)
    ranges, per_robot = partition_contiguous(dataset, num_poses, num_robots)
    graphs = []
    for a in range(num_robots):
        pg = PoseGraph(a, r, dataset.d)
        pg.setMeasurements(per_robot[a])
        pg.pose_order = None if order is None else (order[ranges[a][0]:ranges[a][1]] - ranges[a][0]).astype(np.int64)
        graphs.append(pg)
    return ranges, graphs


class ExchangePlan:
    """Static description of the public-pose exchange, computed once from the pose graphs (host
    only; the reference recomputes PoseDicts every iteration, src/PGOAgent.cpp:97-166).

    slots[a]      : sorted (robot, frame) ids agent a needs = column order of its neighbour tile buffer
    recv_range[a] : {q: (lo, hi)} -- the slots filled by neighbour q are one contiguous range
    send_frames[a]: {q: [frames]} -- a's own poses that q needs, in q's slot order
    colour[a]     : greedy colouring of the agent graph in id order (chain / even ring: 2 colours)
    """

    def __init__(self, graphs: Sequence[PoseGraph]):
        self.num_agents = len(graphs)
        self.slots: List[List[Tuple[int, int]]] = [g.neighborPoseIDs() for g in graphs]
        self.recv_range: List[Dict[int, Tuple[int, int]]] = []
        self.send_frames: List[Dict[int, List[int]]] = [dict() for _ in graphs]
        for a, sl in enumerate(self.slots):
            rr: Dict[int, Tuple[int, int]] = {}
            for k, (rob, _fr) in enumerate(sl):
                lo, _hi = rr.get(rob, (k, k))
                rr[rob] = (lo, k + 1)
            self.recv_range.append(rr)
            for q in rr:
                lo, hi = rr[q]
                self.send_frames[q][a] = [fr for _rob, fr in sl[lo:hi]]
        self.adj: List[List[int]] = [sorted(rr.keys()) for rr in self.recv_range]
        self.colour = [-1] * self.num_agents
        for a in range(self.num_agents):
            used = {self.colour[q] for q in self.adj[a] if self.colour[q] >= 0}
            c = 0
            while c in used:
                c += 1
            self.colour[a] = c
        self.num_colours = (max(self.colour) + 1) if self.num_agents else 0

    def messages_to(self, q: int) -> List[Tuple[int, int]]:
        """(sender, q) pairs: what the demo's selected robot pulls (MultiRobotExample.cpp:183-204)."""
        return [(a, q) for a in self.adj[q]]

    def messages(self, receivers: Optional[int] = None) -> List[Tuple[int, int]]:
        """(sender, receiver) pairs of one exchange; receivers = colour class that needs fresh
        neighbour poses (None = everyone).  Deterministic order on every rank."""
        out = []
        for q in range(self.num_agents):
            if receivers is None or self.colour[q] == receivers:
                for a in self.adj[q]:
                    out.append((a, q))
        return out


class AgentStatusMixin:
    """The host half of PGOAgent's status block (src/PGOAgent.cpp:399-420), shared by DeviceAgent and the CPU stand-in
    of the gloo tests.  Needs: id, X, torch; provides enable_status / finish_status; measure_relative_change is the
    agent's own (a device kernel for DeviceAgent)."""

    def enable_status(self, agent_params: Optional[PGOAgentParameters] = None) -> None:
        """Track PGOAgentStatus: relativeChange = maxTranslationDistance(X, XPrev) is evaluated after every optimising
        update (on the device: dpgo_max_translation_distance_device) and left in a one-element tensor; the cluster
        reads the scalars of all its agents back together."""
        self.agent_params = agent_params or PGOAgentParameters()
        if not hasattr(self, "XPrev"):
            self.XPrev = self.X.clone()
        self.rel_dev = self.torch.zeros(1, dtype=self.torch.float64, device=self.X.device)
        self.status = PGOAgentStatus(self.id, "INITIALIZED")
        self.weight_update_count = 0
        self.lc_weights = None  # weights of this agent's loop closures (None: never re-weighted = all 1)
        self.track_status = True

    def finish_status(self, iteration: int, relative_change: float, success: bool) -> PGOAgentStatus:
        """The status block (src/PGOAgent.cpp:401-420) once relativeChange is known."""
        prm = self.agent_params
        ready = bool(success)
        tol = prm.relChangeTol
        if prm.robust and self.weight_update_count == 0:  # loose threshold before the first weight update (:411-415)
            tol = 5.0
        if relative_change > tol:
            ready = False
        w = self.lc_weights
        if w is not None and len(w) > 0:
            ratio = float(((w == 1).sum() + (w == 0).sum()) / len(w))
            if ratio < prm.robustOptMinConvergenceRatio:
                ready = False
        self.status = PGOAgentStatus(self.id, "INITIALIZED", 0, int(iteration), ready, float(relative_change))
        return self.status


class DeviceAgent(AgentStatusMixin):
    """One agent on one GPU: device-resident X, neighbour tile buffer and problem handle.

    With enable_acceleration() it also carries the Nesterov state of PGOAgent (XPrev, Y, V, gamma, alpha,
    restart every `restartInterval` iterations; src/PGOAgent.cpp:880-936) on the device; updateY / updateV
    are one fused kernel each (linear combination + polar projection, dpgo_axpby_project_device)."""

    # buffers whose device ADDRESSES the cluster's cached exchange plans hold: re-binding any of them (a second
    # enable_acceleration(), a caller assigning agent.X) advances buffer_generation, which the plan cache is keyed on --
    # a stale plan would gather from / scatter into memory the allocator may already have handed to another tensor
    _PLAN_BUFFERS = frozenset(("X", "Y", "nbr", "nbr_aux", "send_idx", "send_buf", "send_buf_aux"))

    def __setattr__(self, name, value):
        if name in DeviceAgent._PLAN_BUFFERS:
            object.__setattr__(self, "buffer_generation", self.__dict__.get("buffer_generation", 0) + 1)
        object.__setattr__(self, name, value)

    def __init__(self, graphs: Sequence[PoseGraph], plan: ExchangePlan, my_id: int, X0_tiles: np.ndarray,
                 params: Optional[ROptParameters] = None, device: int = 0):
        import torch
        self.torch = torch
        self.id = my_id
        self.pg = graphs[my_id]
        self.plan = plan
        self.r, self.d, self.n = self.pg.r(), self.pg.d(), self.pg.n()
        self.b = self.d + 1
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        if X0_tiles.shape != (self.n, self.b, self.r):
            raise ValueError("X0 tiles have shape %s, expected %s" % (X0_tiles.shape, (self.n, self.b, self.r)))
        # internal row of every caller frame (build_pose_graphs(reorder=...)); None: the caller's order is kept
        self.pose_order = getattr(self.pg, "pose_order", None)
        if self.pose_order is not None:
            Xi = np.empty_like(np.ascontiguousarray(X0_tiles))
            Xi[self.pose_order] = X0_tiles
            X0_tiles = Xi
        self.problem = QuadraticProblem(self.pg, device=device, host_linear_term=False)
        self.stream_id = torch.cuda.current_stream().cuda_stream  # the stream the handle's work is enqueued on
        self.problem.setStream(self.stream_id)
        self.optimizer = QuadraticOptimizer(self.problem, params or ROptParameters())
        self.X = torch.tensor(np.ascontiguousarray(X0_tiles), dtype=torch.float64, device=self.device)
        self.has_neighbours = len(plan.slots[my_id]) > 0
        if self.has_neighbours:
            slots = self.problem.setCouplingFromPoseGraph()
            assert slots == plan.slots[my_id]
        self.nbr = torch.zeros((max(len(plan.slots[my_id]), 1), self.b, self.r), dtype=torch.float64,
                               device=self.device)
        self.send_idx = {q: torch.tensor(fr, dtype=torch.int32, device=self.device)
                         for q, fr in plan.send_frames[my_id].items()}
        self.send_buf = {q: torch.empty((len(ix), self.b, self.r), dtype=torch.float64, device=self.device)
                         for q, ix in self.send_idx.items()}
        self.last_result: Optional[ROPTResult] = None
        self.acceleration = False
        self.iteration = 0
        self.tcg_total = 0

    def enable_acceleration(self, num_robots: int, restart_interval: int = 30) -> None:
        """PGOAgent::initializeAcceleration (src/PGOAgent.cpp:899-908); restartInterval default 30
        (include/DPGO/PGOAgent.h:118)."""
        self.acceleration = True
        self.num_robots, self.restart_interval = int(num_robots), int(restart_interval)
        self.XPrev, self.Y, self.V = self.X.clone(), self.X.clone(), self.X.clone()
        self.gamma = self.alpha = 0.0
        self.nbr_aux = self.torch.zeros_like(self.nbr)
        self.send_buf_aux = {q: self.torch.empty_like(b) for q, b in self.send_buf.items()}

    # ---- status (PGOAgent::iterate, src/PGOAgent.cpp:399-420): host half in AgentStatusMixin ----
    def loop_closure_weights(self) -> np.ndarray:
        """Current weights of the agent's loop closures, private and shared (what PoseGraph::statistics walks,
        src/PoseGraph.cpp:305-340); needs setReweightableEdges(include_shared=True)."""
        m = self.pg.measurements()
        idx = self.problem.reweightable_index
        w, _ = self.problem.getEdgeWeights()
        lc = ~self.pg.odometry_mask()[idx]
        return np.asarray(w)[lc]

    def measure_relative_change(self, stream=None) -> None:
        """Enqueue relativeChange of (X, XPrev) into this agent's device scalar."""
        L.check(self.problem._lib.dpgo_max_translation_distance_device(
            self.r, self.d, self.n, L.ptr(self.X), L.ptr(self.XPrev), L.ptr(self.rel_dev), None,
            stream if stream is not None else (self.torch.cuda.current_stream().cuda_stream or None)))

    # ---- K11: pack / unpack of public poses (PGOAgent::getSharedPoseDict / getAuxSharedPoseDict) ----
    def pack(self, q: int, aux: bool = False):
        ix = self.send_idx[q]
        buf = self.send_buf_aux[q] if aux else self.send_buf[q]
        src = self.Y if aux else self.X
        L.check(self.problem._lib.dpgo_gather_tiles_device(self.r, self.d, L.ptr(src), L.ptr(ix), len(ix),
                                                           L.ptr(buf),
                                                           self.torch.cuda.current_stream().cuda_stream or None))
        return buf

    def recv_view(self, q: int, aux: bool = False):
        lo, hi = self.plan.recv_range[self.id][q]
        return (self.nbr_aux if aux else self.nbr)[lo:hi]

    def _combine_project(self, a, A, b, B, c, Cm, out) -> None:
        """out = polar(a*A + b*B + c*C) per pose: LiftedSEManifold::project of a linear combination."""
        L.check(self.problem._lib.dpgo_axpby_project_device(
            self.r, self.d, self.n, float(a), L.ptr(A), float(b), L.ptr(B), float(c), L.ptr(Cm), 1, L.ptr(out),
            self.torch.cuda.current_stream().cuda_stream or None))

    def _update_x(self, do_opt: bool, acceleration: bool) -> None:
        """PGOAgent::updateX (src/PGOAgent.cpp:938-995)."""
        if not do_opt:
            if acceleration:
                self.X.copy_(self.Y)
            return
        if self.has_neighbours:
            self.problem.updateLinearMatrixFromNeighbors(self.nbr_aux if acceleration else self.nbr)
        if acceleration:
            self.X.copy_(self.Y)  # X0 = Y (:973-978)
        self.last_result = self.optimizer.optimizeDevice(self.X)
        self.tcg_total += self.last_result.tcg_iterations

    def iterate(self, do_opt: bool = True) -> None:
        """PGOAgent::iterate(bool) (src/PGOAgent.cpp:376-432), INITIALIZED state."""
        self.iteration += 1
        if not self.acceleration:
            self._update_x(do_opt, False)
            return
        import math
        self.XPrev.copy_(self.X)
        N = self.num_robots
        self.gamma = (1 + math.sqrt(1 + 4 * N ** 2 * self.gamma ** 2)) / (2 * N)  # updateGamma (:910-914)
        self.alpha = 1 / (self.gamma * N)  # updateAlpha (:916-920)
        self._combine_project(1 - self.alpha, self.X, self.alpha, self.V, 0.0, None, self.Y)  # updateY (:922-928)
        self._update_x(do_opt, True)
        self._combine_project(1.0, self.V, self.gamma, self.X, -self.gamma, self.Y, self.V)  # updateV (:930-936)
        if (self.iteration + 1) % self.restart_interval == 0:  # shouldRestart (:880-885)
            self.X.copy_(self.XPrev)  # restartNesterovAcceleration (:887-897)
            self._update_x(do_opt, False)
            self.V.copy_(self.X)
            self.Y.copy_(self.X)
            self.gamma = self.alpha = 0.0

    # ---- rounding (PGOAgent::getTrajectoryInLocalFrame / InGlobalFrame, src/PGOAgent.cpp:718-767) ----
    def getTrajectoryInLocalFrame(self):
        """Rounded tiles [n, d+1, d] on the device, in the frame of this agent's pose 0."""
        from .trajectory import round_trajectory_device
        anchor = None
        if self.pose_order is not None:  # "this agent's pose 0" is the CALLER's frame 0, wherever it is kept inside
            anchor = np.ascontiguousarray(self.X[int(self.pose_order[0])].cpu().numpy().T)
        return self.in_caller_order(round_trajectory_device(self.X, anchor))

    def getTrajectoryInGlobalFrame(self, anchor):
        """anchor: r x (d+1) lifted pose (PGOAgent::setGlobalAnchor); rounded tiles [n, d+1, d] on the device."""
        from .trajectory import round_trajectory_device
        return self.in_caller_order(round_trajectory_device(self.X, anchor))

    def in_caller_order(self, tiles):
        """Pose tiles [n, ...] held in the agent's internal row order -> the caller's frame order (identity unless the
        agent layer renumbered the block, build_pose_graphs(reorder=...)); torch tensor or array."""
        if self.pose_order is None:
            return tiles
        if isinstance(tiles, np.ndarray):
            return tiles[self.pose_order]
        return tiles[self.torch.as_tensor(self.pose_order, device=tiles.device)]

    def set_iterate(self, X_tiles) -> None:
        """Overwrite the iterate with tiles [n, d+1, r] given in the CALLER's frame order (host array or tensor)."""
        t = self.torch.as_tensor(np.ascontiguousarray(X_tiles) if isinstance(X_tiles, np.ndarray) else X_tiles,
                                 dtype=self.torch.float64, device=self.device)
        if self.pose_order is None:
            self.X.copy_(t)
        else:
            self.X[self.torch.as_tensor(self.pose_order, device=self.device)] = t

    def iterate_in_caller_order(self):
        """The current iterate X as tiles [n, d+1, r] in the caller's frame order (device tensor)."""
        return self.in_caller_order(self.X)

    def snapshot(self) -> None:
        """Remember the current iterate (benchmarks restore it so that every timed step does the same work)."""
        self._snap = self.X.clone()

    def restore(self) -> None:
        self.X.copy_(self._snap)

    # ---- the hot path ----
    def update(self) -> ROPTResult:
        """PGOAgent::updateX(doOptimization = true): G from the neighbour buffer, then optimize."""
        if self.has_neighbours:
            self.problem.updateLinearMatrixFromNeighbors(self.nbr)
        self.last_result = self.optimizer.optimizeDevice(self.X)
        return self.last_result

    def setRobotActive(self, robot_id: int, active: bool = True) -> None:
        """PGOAgent::setRobotActive (src/PGOAgent.cpp:1173-1184): the team's activity flags; for a NEIGHBOUR the shared
        edges with it leave (or re-enter) this agent's Q and G -- PoseGraph::setNeighborActive drops the data matrices,
        here: Q's values and the coupling blocks are rebuilt and re-uploaded (same block pattern: the hierarchy of the
        preconditioner is refreshed lazily, values only)."""
        self.__dict__.setdefault("team_inactive", set())
        (self.team_inactive.discard if active else self.team_inactive.add)(int(robot_id))
        if self.pg.hasNeighbor(robot_id) and self.pg.isNeighborActive(robot_id) != bool(active):
            # robust mode: the GNC weights live on the device; the host copy refresh() rebuilds Q from must carry them
            # (with the flags as they were: the edges that are out right now keep the weight they come back with)
            self.problem.pullEdgeWeights()
            before = self.pg.q_version
            self.pg.setNeighborActive(robot_id, active)
            if self.pg.q_version != before:
                # refresh() uploads Q and re-registers the re-weightable edges (effective weights, inactive ones fixed); with
                # shared edges registered it has uploaded the coupling blocks as well -- uploading them again would drop
                # the edge lists that index them
                self.problem.refresh()
                shared_registered = (getattr(self.problem, "reweightable_index", None) is not None
                                     and getattr(self.problem, "_reweight_shared", False))
                if self.has_neighbours and not shared_registered:
                    self.problem.setCouplingFromPoseGraph()

    def isRobotActive(self, robot_id: int) -> bool:
        return int(robot_id) not in self.__dict__.get("team_inactive", set())

    def update_begin(self) -> None:
        """update() in two halves (C ABI dpgo_optimize_device_begin / _end): everything of the update is enqueued on the
        agent's stream and the host moves on; update_end() collects the result."""
        self.optimizer.optimizeDeviceBegin(self.X, self.nbr if self.has_neighbours else None)

    def update_end(self) -> ROPTResult:
        self.last_result = self.optimizer.optimizeDeviceEnd()
        return self.last_result

    def local_terms(self):
        """(sum X_a Q_a . X_a, sum X_a . G_a, |rgrad_a|^2) with the current neighbour buffer: the
        central cost is 0.5 * sum_a (xqx_a + xg_a) and the central gradnorm^2 is sum_a |rgrad_a|^2
        (SURVEY 8c': the agent-local gradient is the agent's block of the central gradient)."""
        if self.has_neighbours:
            self.problem.updateLinearMatrixFromNeighbors(self.nbr)
        return self.problem.evalTermsDevice(self.X)


class RBCDCluster:
    """The agents hosted by THIS process plus the transport to the others.

    local_agents: {agent id: agent}; an "agent" only needs .id, .pack(q), .recv_view(q), .update(),
    .local_terms().  owner(a) = rank hosting agent a: everything on rank 0 in single-process mode,
    agents [k*apr, (k+1)*apr) on rank k under torch.distributed (one process per GPU)."""

    def __init__(self, plan: ExchangePlan, local_agents: Dict[int, object], rank: int = 0, world: int = 1,
                 stage_through_host: Optional[bool] = None, agents_per_rank: int = 1, comm=None,
                 loopback: bool = False):
        """comm: a dpgo_amd.comm.DeviceComm -- the exchange and the reductions are then issued by the solver library
        itself (RCCL on the solver's stream, no host wait); None: torch.distributed carries them (gloo on CPU test
        boxes, or "nccl" without the library-owned communicator).  loopback (with comm): pairs of agents hosted by
        this process also travel through the communicator (self send / recv) instead of device copies -- runs the
        whole transport on a single GPU."""
        self.plan, self.agents, self.rank, self.world = plan, local_agents, rank, world
        self.agents_per_rank = int(agents_per_rank)
        self.comm = comm
        self.loopback = bool(loopback) and comm is not None
        # same-colour local agents are solved concurrently (sweep); DPGO_SEQUENTIAL_SWEEP=1: one after the other (A/B)
        import os
        self.concurrent = os.environ.get("DPGO_SEQUENTIAL_SWEEP", "0") != "1"
        if world > 1 and comm is None:
            import torch.distributed as dist
            if stage_through_host is None:
                stage_through_host = dist.get_backend() == "gloo"
        self.stage = bool(stage_through_host) and comm is None
        self.peer_store = None  # dpgo_amd.ipc.IpcPeerStore (enable_peer_store): processes of one node, mapped buffers

    def set_robot_active(self, robot_id: int, active: bool = True) -> None:
        """The driver's part of PGOAgent::setRobotActive: every local agent is told (its neighbours drop / restore the
        shared edges), the inactive agent stops updating, and the termination / weight-update votes skip it
        (src/PGOAgent.cpp:861-862, 1016-1017)."""
        self.__dict__.setdefault("inactive", set())
        (self.inactive.discard if active else self.inactive.add)(int(robot_id))
        for ag in self.agents.values():
            if hasattr(ag, "setRobotActive"):
                ag.setRobotActive(robot_id, active)

    def _active_ids(self, c: int):
        off = self.__dict__.get("inactive", ())
        return [a for a in self.agents if self.plan.colour[a] == c and a not in off]

    def enable_peer_store(self) -> None:
        """Carry the public-pose exchange by the peer-store transport (dpgo_amd/ipc.py): collective over the default
        process group (gloo suffices).  Reductions and the anchor broadcast keep their transport."""
        from .ipc import IpcPeerStore
        self.peer_store = IpcPeerStore(self)
        self.__dict__.pop("_so_sig", None)

    def owner(self, agent_id: int) -> int:
        """Rank hosting an agent: consecutive ids share a rank (agents_per_rank = 2 puts one agent of
        each colour of a chain / ring partition on every GPU, so no GPU idles during a colour phase)."""
        return agent_id // self.agents_per_rank if self.world > 1 else 0

    def exchange(self, receivers: Optional[int] = None, messages=None, aux: bool = False) -> None:
        """Public-pose exchange: every (sender a -> receiver q) message of ExchangePlan.messages (or the
        given list); aux = the auxiliary sequence Y (PGOAgent::getAuxSharedPoseDict).  Local pairs are
        device copies; remote pairs form ONE grouped batch of isend/irecv (ncclGroupStart/End under
        RCCL), so no ordering between ranks can deadlock."""
        msgs = messages if messages is not None else self.plan.messages(receivers)
        if getattr(self, "peer_store", None) is not None:  # senders write straight into the receivers' mapped buffers
            self.peer_store.exchange(msgs, receivers if messages is None else tuple(msgs), bool(aux))
            return
        if self._exchange_batched(msgs, (receivers if messages is None else tuple(msgs), bool(aux)), aux):
            return
        ops, staged = [], []
        sends, recvs = [], []
        dist = None
        # the library-owned communicator: all outgoing messages of this process are packed by ONE launch
        packed = self.comm is not None and self._pack_batched(msgs, (receivers if messages is None else tuple(msgs), bool(aux)), aux)
        for a, q in msgs:
            a_here, q_here = a in self.agents, q in self.agents
            if a_here and q_here and not self.loopback:
                self.agents[q].recv_view(a, aux).copy_(self.agents[a].pack(q, aux))
            elif self.comm is not None:
                if a_here:
                    ag = self.agents[a]
                    sends.append((self.owner(q), (ag.send_buf_aux[q] if aux else ag.send_buf[q]) if packed
                                  else ag.pack(q, aux)))
                if q_here:
                    recvs.append((self.owner(a), self.agents[q].recv_view(a, aux)))
            elif a_here:
                import torch.distributed as dist
                buf = self.agents[a].pack(q, aux)
                ops.append(dist.P2POp(dist.isend, buf.cpu() if self.stage else buf, self.owner(q)))
            elif q_here:
                import torch.distributed as dist
                view = self.agents[q].recv_view(a, aux)
                if self.stage:
                    tmp = view.cpu()
                    staged.append((view, tmp))
                    ops.append(dist.P2POp(dist.irecv, tmp, self.owner(a)))
                else:
                    ops.append(dist.P2POp(dist.irecv, view, self.owner(a)))
        if sends or recvs:  # library-owned RCCL communicator: one grouped batch on the solver's stream, no host wait
            any_agent = next(iter(self.agents.values()))
            self.comm.exchange(sends, recvs, any_agent.torch.cuda.current_stream().cuda_stream)
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            for view, tmp in staged:
                view.copy_(tmp)

    def _exchange_batched(self, msgs, key, aux: bool) -> bool:
        """All messages between device agents of THIS process (no communicator in between): ONE launch for the whole
        exchange phase (C ABI dpgo_exchange_plan_*; the plan -- iterate / index / neighbour-buffer addresses per message
        -- is built once per phase and cached).  Returns False when the exchange needs the general path."""
        if self.loopback or not msgs or not all(a in self.agents and q in self.agents for a, q in msgs):
            return False
        first = self.agents[msgs[0][0]]
        if not hasattr(first, "send_idx") or not hasattr(first, "problem"):
            return False  # (CPU stand-ins of the gloo tests)
        import ctypes as C
        plan = self._cached_plan(key)
        if plan is None:
            src, idx, cnt, dst = [], [], [], []
            for a, q in msgs:
                sa, rq = self.agents[a], self.agents[q]
                src.append(L.ptr(sa.Y if aux else sa.X))
                idx.append(L.ptr(sa.send_idx[q]))
                cnt.append(len(sa.send_idx[q]))
                dst.append(L.ptr(rq.recv_view(a, aux)))
            n = len(msgs)
            h = L._P()
            L.check(first.problem._lib.dpgo_exchange_plan_create(
                C.byref(h), first.r, first.d, n, (C.c_void_p * n)(*src), (C.c_void_p * n)(*idx), (C.c_int * n)(*cnt),
                (C.c_void_p * n)(*dst), first.device.index or 0))
            plan = self._store_plan(key, h)
        L.check(first.problem._lib.dpgo_exchange_plan_run(plan, first.torch.cuda.current_stream().cuda_stream or None))
        return True

    def _buffer_generation(self) -> int:
        return sum(getattr(ag, "buffer_generation", 0) for ag in self.agents.values())

    def _cached_plan(self, key):
        """The cached exchange plan for `key`, or None -- also when any local agent re-bound one of the buffers the plans
        address since the plans were built (every plan is dropped then: they hold raw device addresses)."""
        cache = self.__dict__.setdefault("_xplans", {})
        gen = self._buffer_generation()
        if self.__dict__.get("_xplans_gen") != gen:
            self.invalidate_plans()
            self._xplans_gen = gen
        return cache.get(key)

    def _store_plan(self, key, handle):
        self._xplans[key] = handle
        return handle

    def invalidate_plans(self) -> None:
        """Destroy the cached exchange plans (they are rebuilt from the agents' current buffers at the next exchange)."""
        cache = self.__dict__.setdefault("_xplans", {})
        if cache:
            lib = next(iter(self.agents.values())).problem._lib
            for h in cache.values():
                lib.dpgo_exchange_plan_destroy(h)
            cache.clear()

    def _pack_batched(self, msgs, key, aux: bool) -> bool:
        """K11 for every message this process SENDS through the communicator, as one launch (same plan machinery as
        _exchange_batched, the destinations are the senders' send buffers)."""
        out = [(a, q) for a, q in msgs if a in self.agents and (self.loopback or q not in self.agents)]
        if not out or not hasattr(self.agents[out[0][0]], "send_idx"):
            return False
        import ctypes as C
        first = self.agents[out[0][0]]
        plan = self._cached_plan(("pack",) + key)
        if plan is None:
            n = len(out)
            src = [L.ptr(self.agents[a].Y if aux else self.agents[a].X) for a, q in out]
            idx = [L.ptr(self.agents[a].send_idx[q]) for a, q in out]
            cnt = [len(self.agents[a].send_idx[q]) for a, q in out]
            dst = [L.ptr(self.agents[a].send_buf_aux[q] if aux else self.agents[a].send_buf[q]) for a, q in out]
            h = L._P()
            L.check(first.problem._lib.dpgo_exchange_plan_create(
                C.byref(h), first.r, first.d, n, (C.c_void_p * n)(*src), (C.c_void_p * n)(*idx), (C.c_int * n)(*cnt),
                (C.c_void_p * n)(*dst), first.device.index or 0))
            plan = self._store_plan(("pack",) + key, h)
        L.check(first.problem._lib.dpgo_exchange_plan_run(plan, first.torch.cuda.current_stream().cuda_stream or None))
        return True

    def __del__(self):
        try:
            self.invalidate_plans()
        except Exception:
            pass

    def _allreduce_host(self, values: np.ndarray) -> np.ndarray:
        """Sum of a small host array over the ranks (cost / gradient-norm terms).  Loop-back mode sends it through the
        1-rank communicator as well (the identity), so that the whole N > 1 data path runs on one GPU."""
        if self.world == 1 and not self.loopback:
            return values
        import torch
        any_agent = next(iter(self.agents.values()))
        dev = getattr(any_agent, "device", "cpu")
        if self.comm is not None:
            t = torch.tensor(values, dtype=torch.float64, device=dev)
            self.comm.allreduce(t, 0, torch.cuda.current_stream().cuda_stream)
            return t.cpu().numpy()
        import torch.distributed as dist
        t = torch.tensor(values, dtype=torch.float64, device="cpu" if self.stage else dev)
        dist.all_reduce(t)
        return t.cpu().numpy()

    def _main_stream(self):
        any_agent = next(iter(self.agents.values()))
        return any_agent.torch.cuda.current_stream().cuda_stream

    def phase(self, c: int, iteration: Optional[int] = None) -> None:
        """One colour phase = one global iteration of the coloured schedule: exchange, the agents of colour c update
        (concurrently when this process hosts several).  With status tracking (DeviceAgent.enable_status) and an
        iteration number, XPrev is saved before and the agents' relative changes are measured on the device after the
        solves, read back together (one small copy per phase) and turned into PGOAgentStatus records."""
        self.exchange(receivers=c)
        ids = self._active_ids(c)
        tracked = [a for a in ids if getattr(self.agents[a], "track_status", False)] if iteration is not None else []
        for a in tracked:
            self.agents[a].XPrev.copy_(self.agents[a].X)
        if self.concurrent and len(ids) > 1 and all(hasattr(self.agents[a], "optimizer") for a in ids):
            ags = [self.agents[a] for a in ids]
            res = optimize_device_many([g.optimizer for g in ags], [g.X for g in ags],
                                       [g.nbr if g.has_neighbours else None for g in ags], self._main_stream())
            for g, r_ in zip(ags, res):
                g.last_result = r_
        else:
            for a in ids:
                self.agents[a].update()
        if tracked:
            torch = self.agents[tracked[0]].torch
            for a in tracked:
                self.agents[a].measure_relative_change()
            rel = torch.cat([self.agents[a].rel_dev for a in tracked]).cpu().numpy()  # ONE read-back per phase
            for a, v in zip(tracked, rel):
                ag = self.agents[a]
                ag.finish_status(iteration, float(v), bool(getattr(getattr(ag, "last_result", None), "success", True)))

    def team_status(self) -> Dict[int, PGOAgentStatus]:
        """Statuses of ALL agents, identical on every rank (the reference's agents publish theirs to the team,
        PGOAgent::setNeighborStatus): one all-reduce of a [num_agents, 4] array, every row owned by one rank."""
        rows = np.zeros((self.plan.num_agents, 4))
        for a, ag in self.agents.items():
            st = getattr(ag, "status", None)
            if st is not None and st.iterationNumber > 0:
                rows[a] = [1.0, st.iterationNumber, 1.0 if st.readyToTerminate else 0.0, st.relativeChange]
        rows = self._allreduce_host(rows)
        return {a: PGOAgentStatus(a, "INITIALIZED", 0, int(rows[a, 1]), bool(rows[a, 2] > 0.5), float(rows[a, 3]))
                for a in range(self.plan.num_agents) if rows[a, 0] > 0.5}

    def run_until_terminated(self, agent_params: Optional[PGOAgentParameters] = None, max_phases: int = 10000):
        """The coloured schedule driven by the reference's own stopping rule (PGOAgent::shouldTerminate,
        src/PGOAgent.cpp:846-878): one global iteration = one colour phase; after every phase the team status is
        shared and every agent evaluates the vote on the same statuses, so all ranks stop together.  Returns
        dict(iterations, statuses, relative_changes)."""
        prm = agent_params or PGOAgentParameters()
        for ag in self.agents.values():
            if not getattr(ag, "track_status", False):
                ag.enable_status(prm)
            ag.agent_params = prm
        iteration, trace, team = 0, [], {}
        while iteration < max_phases:
            c = iteration % self.plan.num_colours
            iteration += 1
            self.phase(c, iteration)
            team = self.team_status()
            trace.append({a: st.relativeChange for a, st in team.items()})
            if should_terminate(iteration, prm, 0, team, self.plan.num_agents, self.__dict__.get("inactive", ())):
                break
        return dict(iterations=iteration, statuses=team, relative_changes=trace)

    def sweep(self) -> None:
        """One RBCD iteration: every colour class updates once.  The agents of a colour hosted by THIS process are
        updated concurrently (C ABI dpgo_optimize_device_many: each on its own stream behind the exchange, one feeding
        host thread each), so a GPU that hosts several same-colour agents overlaps their latency-bound solves instead of
        running them one after the other."""
        if self._stream_ordered_sweep():
            # ONE agent per colour on this process (the multi-GPU deployment: two agents per GPU): the whole sweep --
            # colour c's exchange, its solve, colour c+1's pack + exchange, ... -- is enqueued on one stream without a host
            # wait in between (a one-launch solve needs no host feed); the results are read back at the end of the sweep
            # instead of leaving the GPU idle for a host round trip per phase
            begun = []
            try:
                for c in range(self.plan.num_colours):
                    self.exchange(receivers=c)
                    for a in self._active_ids(c):
                        self.agents[a].update_begin()
                        begun.append(a)
            except Exception:
                for a in begun:  # never leave a solve in flight behind an error: the handles would refuse the next begin
                    try:
                        self.agents[a].update_end()
                    except Exception:
                        pass
                raise
            fell_back = None
            for a in begun:
                ag = self.agents[a]
                was_on = ag.problem.persistentInfo()["enabled"]
                ag.update_end()
                if fell_back is None and was_on and not ag.problem.persistentInfo()["enabled"]:
                    fell_back = a  # its one-launch solve timed out and was re-run with the multi-launch scheme in update_end
            if fell_back is not None:
                # ... AFTER the later colours had exchanged and solved against its old poses: for those colours this sweep
                # was a Jacobi-style update, without RBCD's descent guarantee.  Say so, and redo them in order.
                import warnings
                warnings.warn("dpgo_amd: agent %d's one-launch solve fell back inside a stream-ordered sweep; the colours "
                              "behind it are exchanged and solved again" % fell_back)
                for c in range(self.plan.colour[fell_back] + 1, self.plan.num_colours):
                    self.exchange(receivers=c)
                    for a in self._active_ids(c):
                        self.agents[a].update()
            return
        for c in range(self.plan.num_colours):
            self.exchange(receivers=c)
            ids = self._active_ids(c)
            if self.concurrent and len(ids) > 1 and all(hasattr(self.agents[a], "optimizer") for a in ids):
                ags = [self.agents[a] for a in ids]
                res = optimize_device_many([g.optimizer for g in ags], [g.X for g in ags],
                                           [g.nbr if g.has_neighbours else None for g in ags], self._main_stream())
                for g, r_ in zip(ags, res):
                    g.last_result = r_
            else:
                for a in ids:
                    self.agents[a].update()

    def _stream_ordered_sweep(self) -> bool:
        """A sweep can be enqueued whole (sweep()): device agents with begin / end updates, at most one local agent per
        colour, every exchange issued on the agents' stream (device copies or the library-owned communicator -- not the
        torch.distributed fallback, whose waits block the host), all agents on the stream the exchanges use.
        DPGO_ASYNC_SWEEP=0 switches it off (A/B)."""
        # (decided afresh whenever the set of agents, their streams or the current stream change: a handle on another
        # stream than the one the exchanges are enqueued on would not be ordered behind them)
        if not self.agents or not all(hasattr(ag, "update_begin") for ag in self.agents.values()):
            return False  # (CPU stand-ins of the gloo tests: nothing to enqueue)
        cur = self._main_stream()
        sig = (tuple(sorted(self.agents)), tuple(getattr(ag, "stream_id", None) for _, ag in sorted(self.agents.items())),
               cur, self.peer_store is None, self.comm is None)
        if self.__dict__.get("_so_sig") == sig:
            return self._so_sweep
        import os
        ok = os.environ.get("DPGO_ASYNC_SWEEP", "1") != "0" and self.plan.num_agents > 1
        ok = ok and all(hasattr(ag, "update_begin") and hasattr(ag, "optimizer") for ag in self.agents.values())
        per_colour: Dict[int, int] = {}
        for a in self.agents:
            per_colour[self.plan.colour[a]] = per_colour.get(self.plan.colour[a], 0) + 1
        ok = ok and all(v <= 1 for v in per_colour.values())
        ok = ok and (self.world == 1 or self.comm is not None) and self.peer_store is None
        ok = ok and all(getattr(ag, "stream_id", None) == cur for ag in self.agents.values())
        self._so_sweep, self._so_sig = bool(ok), sig
        return self._so_sweep

    def block_terms(self) -> np.ndarray:
        """[num_agents, 2] array of (0.5 (xqx + xg), |rgrad_a|^2) per agent, identical on every rank."""
        self.exchange(None)
        out = np.zeros((self.plan.num_agents, 2))
        for a, (xqx, xg, g2) in self._local_terms().items():
            out[a] = [0.5 * (xqx + xg), g2]
        return self._allreduce_host(out)

    def _local_terms(self) -> Dict[int, Tuple[float, float, float]]:
        """{agent id: (xqx, xg, |rgrad|^2)} of the local agents with the current neighbour buffers: one concurrent
        device pass and one read-back per agent (dpgo_problem_eval_terms_device_many)."""
        ids = list(self.agents)
        if self.concurrent and len(ids) > 1 and all(hasattr(self.agents[a], "problem") for a in ids):
            ags = [self.agents[a] for a in ids]
            t = eval_terms_device_many([g.problem for g in ags], [g.X for g in ags],
                                       [g.nbr if g.has_neighbours else None for g in ags], self._main_stream())
            return {a: tuple(t[k]) for k, a in enumerate(ids)}
        return {a: self.agents[a].local_terms() for a in ids}

    def run_greedy(self, max_iters: int = 1000, gradnorm_stop: float = 0.1):
        """The reference demo's schedule (examples/MultiRobotExample.cpp:170-255): every non-selected agent
        iterate(false); the selected agent pulls its neighbours' public (and auxiliary) poses and
        iterate(true); stop when the central gradnorm < 0.1; next = argmax of the block gradnorms.
        The central gradient block of agent a equals a's local gradient (SURVEY 8c'), so the selection is
        computed from agent-local evaluations + one tiny all-reduce instead of a central problem."""
        selected, order, trace = 0, [], []
        for _ in range(max_iters):
            for a, agent in self.agents.items():
                if a != selected:
                    agent.iterate(False)
            self.exchange(messages=self.plan.messages_to(selected))
            accel = any(getattr(ag, "acceleration", False) for ag in self.agents.values())
            if accel:
                self.exchange(messages=self.plan.messages_to(selected), aux=True)
            if selected in self.agents:
                self.agents[selected].iterate(True)
            terms = self.block_terms()
            cost, gn = 2.0 * terms[:, 0].sum(), float(np.sqrt(terms[:, 1].sum()))
            order.append(selected)
            trace.append((cost, gn))
            if gn < gradnorm_stop:
                break
            if self.plan.adj[selected]:
                selected = int(np.argmax(terms[:, 1]))
        return dict(iterations=len(order), cost=trace[-1][0], gradnorm=trace[-1][1], selected=order, trace=trace)

    def global_anchor(self) -> np.ndarray:
        """The lifted pose 0 of agent 0 as r x (d+1) (what the reference's drivers pass to setGlobalAnchor),
        identical on every rank."""
        any_agent = next(iter(self.agents.values()))
        b, r = any_agent.b, any_agent.r
        a = np.zeros((b, r))
        if 0 in self.agents:
            ag0 = self.agents[0]  # (the caller's pose 0, wherever the agent keeps it)
            a = ag0.X[0 if getattr(ag0, "pose_order", None) is None else int(ag0.pose_order[0])].cpu().numpy()
        if self.world > 1:
            import torch
            dev = getattr(any_agent, "device", "cpu")
            if self.comm is not None:
                t = torch.tensor(a, dtype=torch.float64, device=dev)
                self.comm.broadcast(t, self.owner(0), torch.cuda.current_stream().cuda_stream)
            else:
                import torch.distributed as dist
                t = torch.tensor(a, dtype=torch.float64, device="cpu" if self.stage else dev)
                dist.broadcast(t, src=self.owner(0))
            a = t.cpu().numpy()
        return np.ascontiguousarray(a.T)  # tile [d+1, r] -> matrix r x (d+1)

    def trajectories_in_global_frame(self) -> Dict[int, object]:
        """{agent id: rounded device tiles [n_a, d+1, d]} of the local agents in the frame of agent 0's pose 0."""
        anchor = self.global_anchor()
        return {a: ag.getTrajectoryInGlobalFrame(anchor) for a, ag in self.agents.items()}

    def central_cost_and_gradnorm(self) -> Tuple[float, float]:
        """Central cost f(X) and Riemannian gradient norm (what examples/MultiRobotExample.cpp:220-225
        evaluates on a central problem), assembled from agent-local terms + one tiny all-reduce."""
        self.exchange(None)
        acc = np.zeros(2)
        for xqx, xg, g2 in self._local_terms().values():
            acc += [0.5 * (xqx + xg), g2]
        acc = self._allreduce_host(acc)
        return float(acc[0]), float(acc[1]) ** 0.5
