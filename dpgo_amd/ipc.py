"""Peer-store transport of the public-pose exchange between PROCESSES that share a node (SURVEY section 5's alternative to
send / receive): every agent's neighbour tile buffer is mapped into the address space of the processes that send to it
(hipIpcGetMemHandle / hipIpcOpenMemHandle, through PyTorch's CUDA-IPC tensor sharing), and a sender's pack kernel -- the one
batched launch of the exchange plan, k_gather_tiles_batched -- writes the public poses STRAIGHT into the receiver's
buffer: no staging buffer, no communicator kernel, one launch per exchange phase and process.

What it needs from the launcher: a torch.distributed process group for the rendezvous (the IPC handles travel over it
once) and for the two host barriers that order an exchange against the solves around it (gloo is enough: RCCL refuses two
ranks on one device, IPC does not -- this is the transport with which the multi-process data path runs on ONE GPU).  The
processes may sit on the same device or on different devices of one node (peer access).  RCCL stays the default transport
of bench.py and of the multi-node case; `bench.py --transport ipc` prints this transport's exchange time beside it.

Reference counterpart: the in-process pointer calls of examples/MultiRobotExample.cpp:183-204 (setNeighborPoses with the
sender's PoseDict) -- a peer store is the closest device analogue of handing the neighbour a pointer."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Tuple

from . import lib as L


class IpcPeerStore:
    """Collective constructor (every rank of the default process group calls it with its RBCDCluster)."""

    def __init__(self, cluster):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.cluster = cluster
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self._remote: Dict[Tuple[int, bool], object] = {}  # (agent id, aux) -> tensor aliasing that agent's buffer
        self._plans: Dict[tuple, object] = {}
        self._shared_gen: Dict[bool, int] = {}
        self._keep: List[object] = []
        self._share(False)

    # ---- rendezvous: export every local agent's neighbour buffer, open everybody else's ----
    def _share(self, aux: bool) -> None:
        from torch.multiprocessing.reductions import reduce_tensor
        mine = {}
        self._keep = [t for t in self._keep if getattr(t, "_dpgo_aux", None) != aux]  # (re-share: drop the old exports)
        for a, ag in self.cluster.agents.items():
            t = ag.nbr_aux if aux else ag.nbr
            mine[a] = reduce_tensor(t)  # (rebuild function, arguments incl. the hipIpcMemHandle of the allocation)
            try:
                t._dpgo_aux = aux
            except Exception:
                pass
            self._keep.append(t)
        everyone = [None] * self.world
        self.dist.all_gather_object(everyone, mine)
        for r, exported in enumerate(everyone):
            if r == self.rank:
                continue
            for a, (rebuild, args) in exported.items():
                self._remote[(a, aux)] = rebuild(*args)
        self._shared_gen[aux] = self.cluster._buffer_generation()

    def _destroy(self, handle) -> None:
        next(iter(self.cluster.agents.values())).problem._lib.dpgo_exchange_plan_destroy(handle)

    def _dst(self, q: int, a: int, aux: bool):
        """The slots of agent q's neighbour buffer that agent a fills -- local tensor or the mapped remote one."""
        lo, hi = self.cluster.plan.recv_range[q][a]
        if q in self.cluster.agents:
            ag = self.cluster.agents[q]
            return (ag.nbr_aux if aux else ag.nbr)[lo:hi]
        return self._remote[(q, aux)][lo:hi]

    def exchange(self, msgs, key, aux: bool) -> None:
        """All messages of one exchange phase whose SENDER lives here, as one launch; two host barriers order it against
        the receivers' solves (before: nobody still reads the buffers; after: everything has landed)."""
        torch, dist = self.torch, self.dist
        c = self.cluster
        gen = c._buffer_generation()
        out = [(a, q) for a, q in msgs if a in c.agents]
        torch.cuda.current_stream().synchronize()
        # first (auxiliary) exchange, or an agent of SOME rank re-bound one of its buffers: export / open again.  _share is
        # collective, the buffer generation is a local count -- so the decision is agreed on: the maximum over the ranks
        # of "mine changed" rides on the reduction that is the barrier before the launch (nobody still reads the buffers)
        flag = torch.tensor([1.0 if self._shared_gen.get(aux) != gen else 0.0],
                            device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if float(flag.item()) > 0.0:
            for key_ in [k for k in self._remote if k[1] == aux]:  # (stale mappings of this kind of buffer)
                del self._remote[key_]
            self._share(aux)
            self._plan_gen = None  # remote addresses may have changed whatever the local generation says
        if self.__dict__.get("_plan_gen") != gen:  # the plans hold raw device addresses
            for h in self._plans.values():
                self._destroy(h)
            self._plans.clear()
            self._plan_gen = gen
        if out:
            first = c.agents[out[0][0]]
            pkey = (key, aux)
            plan = self._plans.get(pkey)
            if plan is None:
                n = len(out)
                src = [L.ptr(c.agents[a].Y if aux else c.agents[a].X) for a, q in out]
                idx = [L.ptr(c.agents[a].send_idx[q]) for a, q in out]
                cnt = [len(c.agents[a].send_idx[q]) for a, q in out]
                dst = [L.ptr(self._dst(q, a, aux)) for a, q in out]
                h = L._P()
                L.check(first.problem._lib.dpgo_exchange_plan_create(
                    C.byref(h), first.r, first.d, n, (C.c_void_p * n)(*src), (C.c_void_p * n)(*idx), (C.c_int * n)(*cnt),
                    (C.c_void_p * n)(*dst), first.device.index or 0))
                plan = self._plans[pkey] = h
            L.check(first.problem._lib.dpgo_exchange_plan_run(plan, torch.cuda.current_stream().cuda_stream or None))
            torch.cuda.current_stream().synchronize()
        dist.barrier()

    def close(self) -> None:
        for h in self._plans.values():
            self._destroy(h)
        self._plans.clear()
        self._remote.clear()
