"""Peer-store transport of the public-pose exchange between PROCESSES that share a node (SURVEY section 5's alternative to
send / receive): every agent's neighbour tile buffer is mapped into the address space of the processes that send to it
(hipIpcGetMemHandle / hipIpcOpenMemHandle, through PyTorch's CUDA-IPC tensor sharing), and a sender's pack kernel -- the one
batched launch of the exchange plan, k_gather_tiles_batched -- writes the public poses STRAIGHT into the receiver's
buffer: no staging buffer, no communicator kernel, one launch per exchange phase and process.

Ordering (round 5): ON THE DEVICE.  Every process also shares a small array of 64-bit epoch words; per message a -> q there
is a `ready` word in q's process (the tiles of epoch e have landed) and an `ack` word in a's process (q has consumed epoch
e).  An exchange enqueues, on the stream everything else of the sweep is enqueued on and without a host wait:
    acknowledge what arrived earlier  ->  wait for the acks of what is about to be overwritten  ->  pack kernel (writes into
    the receivers' buffers)  ->  publish `ready`  ->  wait for the `ready` words of the incoming messages
(C ABI dpgo_flags_write_device / dpgo_flags_wait_device_checked: system-scope atomics; the data itself is ordered by kernel
boundaries on either side).  A wait has NO time limit by default -- a late peer (a host-side Q rebuild, a checkpoint, a
debugger, a long local solve) delays the stream exactly as round 4's host barriers delayed the host; DPGO_IPC_WAIT_TIMEOUT_MS
(or the constructor's wait_timeout_ms) bounds it, and a word that has not arrived by then is reported through an error word
in pinned host memory: the NEXT exchange() / check() / close() of that process raises RuntimeError (what was read in between
is stale); nothing traps, the HIP context survives.  Any kernel that reads agent.nbr / nbr_aux must be enqueued BEFORE the
next exchange() call (which acknowledges the epoch and so lets the sender overwrite the buffer), on the stream exchange() runs
on -- readers on an agent's own stream are ordered in front of the acknowledgement by an event.  Round 4 ordered an exchange with two HOST
barriers (2.0 ms of an 8.97 ms sweep at 4 x 25 000 poses); DPGO_IPC_HOST_BARRIERS=1 keeps that scheme for A/B runs.

What it needs from the launcher: a torch.distributed process group for the rendezvous (the IPC handles travel over it once;
gloo is enough: RCCL refuses two ranks on one device, IPC does not -- this is the transport with which the multi-process
data path runs on ONE GPU).  The processes may sit on the same device or on different devices of one node (peer access).
RCCL stays the default transport of bench.py and of the multi-node case; `bench.py --transport ipc` prints this transport's
exchange time beside it.

Reference counterpart: the in-process pointer calls of examples/MultiRobotExample.cpp:183-204 (setNeighborPoses with the
sender's PoseDict) -- a peer store is the closest device analogue of handing the neighbour a pointer."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Tuple

from . import lib as L


class IpcPeerStore:
    """Collective constructor (every rank of the default process group calls it with its RBCDCluster)."""

    WAIT_TIMEOUT_MS = 0  # default limit of a device-side wait, milliseconds; 0 = none (DPGO_IPC_WAIT_TIMEOUT_MS overrides)

    def __init__(self, cluster, wait_timeout_ms=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.cluster = cluster
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device_ordered = os.environ.get("DPGO_IPC_HOST_BARRIERS", "0") != "1"
        if wait_timeout_ms is None:
            wait_timeout_ms = int(os.environ.get("DPGO_IPC_WAIT_TIMEOUT_MS", str(self.WAIT_TIMEOUT_MS)))
        if wait_timeout_ms < 0:
            raise ValueError("wait_timeout_ms >= 0 (0 = no limit)")
        self.wait_timeout_ms = int(wait_timeout_ms)
        # the word a timed-out wait reports into: pinned host memory (device-visible; read here without a stream sync)
        self._err = torch.zeros(1, dtype=torch.int64).pin_memory()
        self._remote: Dict[Tuple[int, bool], object] = {}  # (agent id, aux) -> tensor aliasing that agent's buffer
        self._plans: Dict[tuple, object] = {}
        self._shared_gen: Dict[bool, int] = {}
        self._keep: List[object] = []
        # ordering words: [aux][ready / ack][message], message = index into the plan's full message list; every process
        # holds the whole (tiny) array, a word is WRITTEN by exactly one peer and READ by its owner
        self._msgs = list(cluster.plan.messages(None))
        self._mid = {m: k for k, m in enumerate(self._msgs)}
        nm = max(1, len(self._msgs))
        # (a rank that hosts no agent still takes part in the collective rendezvous below)
        dev = (next(iter(cluster.agents.values())).device if cluster.agents
               else torch.device("cuda", torch.cuda.current_device()))
        self._words = torch.zeros(2 * 2 * nm, dtype=torch.int64, device=dev)
        self._nm = nm
        self._epoch: Dict[Tuple[int, int, bool], int] = {}     # (a, q, aux) -> epochs sent / expected so far
        self._acked: Dict[Tuple[int, int, bool], int] = {}     # (a, q, aux) -> last epoch this process acknowledged
        self._peer_words: Dict[int, object] = {}               # rank -> tensor aliasing that rank's words
        self._share_words()
        self._share(False)

    # ---- rendezvous: export every local agent's neighbour buffer, open everybody else's ----
    def _share_words(self) -> None:
        from torch.multiprocessing.reductions import reduce_tensor
        everyone = [None] * self.world
        self.dist.all_gather_object(everyone, reduce_tensor(self._words))
        for r, (rebuild, args) in enumerate(everyone):
            if r != self.rank:
                self._peer_words[r] = rebuild(*args)

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
        L.load().dpgo_exchange_plan_destroy(handle)

    def check(self) -> None:
        """Raises if a device-side wait of this process timed out (wait_timeout_ms > 0 only).  Reads pinned host memory: no
        stream synchronisation."""
        code = int(self._err[0].item())
        if code:
            raise RuntimeError("dpgo_amd.ipc: a device-side wait for a peer's ordering word timed out after %d ms (word %d "
                               "of its batch); the neighbour tiles read since then are stale.  Raise "
                               "DPGO_IPC_WAIT_TIMEOUT_MS (0 = no limit) or use DPGO_IPC_HOST_BARRIERS=1"
                               % (self.wait_timeout_ms, code - 1))

    def _wait(self, lib, entries, stream) -> None:
        if not entries:
            return
        n = len(entries)
        ptrs = (C.c_void_p * n)(*[p for p, _ in entries])
        vals = (C.c_ulonglong * n)(*[v for _, v in entries])
        L.check(lib.dpgo_flags_wait_device_checked(n, ptrs, vals, self.wait_timeout_ms,
                                                   C.c_void_p(self._err.data_ptr()) if self.wait_timeout_ms > 0 else None,
                                                   stream))

    def _dst(self, q: int, a: int, aux: bool):
        """The slots of agent q's neighbour buffer that agent a fills -- local tensor or the mapped remote one."""
        lo, hi = self.cluster.plan.recv_range[q][a]
        if q in self.cluster.agents:
            ag = self.cluster.agents[q]
            return (ag.nbr_aux if aux else ag.nbr)[lo:hi]
        return self._remote[(q, aux)][lo:hi]

    # ---- ordering words ----
    def _word(self, owner_rank: int, aux: bool, which: int, msg) -> int:
        """Device address (in THIS process's address space) of a word that lives in `owner_rank`'s array."""
        t = self._words if owner_rank == self.rank else self._peer_words[owner_rank]
        return t.data_ptr() + 8 * ((2 * int(aux) + which) * self._nm + self._mid[msg])

    def _flags(self, fn, entries, stream, *extra) -> None:
        if not entries:
            return
        n = len(entries)
        ptrs = (C.c_void_p * n)(*[p for p, _ in entries])
        vals = (C.c_ulonglong * n)(*[v for _, v in entries])
        L.check(fn(n, ptrs, vals, *extra, stream))

    def _plan(self, out, key, aux: bool):
        c = self.cluster
        gen = c._buffer_generation()
        if self.__dict__.get("_plan_gen") != gen:  # the plans hold raw device addresses
            for h in self._plans.values():
                self._destroy(h)
            self._plans.clear()
            self._plan_gen = gen
        pkey = (key, aux)
        plan = self._plans.get(pkey)
        if plan is None:
            first = c.agents[out[0][0]]
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
        return plan

    def exchange(self, msgs, key, aux: bool) -> None:
        """All messages of one exchange phase whose SENDER lives here, as one launch, ordered against the receivers' solves
        on the device (see the module text); DPGO_IPC_HOST_BARRIERS=1: by two host barriers as in round 4."""
        if not self.device_ordered:
            return self._exchange_host_barriers(msgs, key, aux)
        torch = self.torch
        c = self.cluster
        gen = c._buffer_generation()
        if aux not in self._shared_gen:
            # the first exchange of this kind of buffer: every rank reaches it at the same point of the same driver code
            self._share(aux)
        elif self._shared_gen[aux] != gen:
            # re-sharing is collective and this scheme has no host rendezvous per exchange to agree on it
            raise RuntimeError("dpgo_amd.ipc: an agent re-bound a buffer the peers have mapped; call "
                               "cluster.peer_store.reshare() on EVERY rank before the next exchange")
        self.check()
        unknown = [m for m in msgs if m not in self._mid]
        if unknown:
            raise ValueError("dpgo_amd.ipc: exchange(messages=...) entries outside the exchange plan: %r" % (unknown[:4],))
        lib = L.load()
        cur = torch.cuda.current_stream()
        stream = cur.cuda_stream or None
        # readers of the neighbour buffers enqueued on an agent's OWN stream (it differs from this one only in hand-written
        # drivers) must be ahead of the acknowledgement below: this stream waits for an event recorded on theirs
        for ag in c.agents.values():
            sid = getattr(ag, "stream_id", None)
            if sid is not None and sid != cur.cuda_stream:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.ExternalStream(sid, device=ag.device) if sid else torch.cuda.default_stream(ag.device))
                cur.wait_event(ev)
        own = c.owner
        out = [(a, q) for a, q in msgs if a in c.agents]
        remote_out = [(a, q) for a, q in out if q not in c.agents]
        remote_in = [(a, q) for a, q in msgs if q in c.agents and a not in c.agents]
        # 1. acknowledge everything that arrived in earlier exchanges: every kernel that read it is ahead on this stream
        acks = []
        for (a, q, x), e in self._epoch.items():
            if q in c.agents and a not in c.agents and self._acked.get((a, q, x), 0) < e:
                acks.append((self._word(own(a), x, 1, (a, q)), e))
                self._acked[(a, q, x)] = e
        self._flags(lib.dpgo_flags_write_device, acks, stream)
        # 2. the receivers have consumed what this exchange overwrites
        self._wait(lib, [(self._word(self.rank, aux, 1, m), self._epoch.get((m[0], m[1], aux), 0))
                         for m in remote_out if self._epoch.get((m[0], m[1], aux), 0) > 0], stream)
        # 3. the tiles
        if out:
            L.check(lib.dpgo_exchange_plan_run(self._plan(out, key, aux), stream))
        # 4. publish, 5. wait for what this process receives
        for m in remote_out + remote_in:
            k = (m[0], m[1], aux)
            self._epoch[k] = self._epoch.get(k, 0) + 1
        self._flags(lib.dpgo_flags_write_device, [(self._word(own(m[1]), aux, 0, m), self._epoch[(m[0], m[1], aux)])
                                                  for m in remote_out], stream)
        self._wait(lib, [(self._word(self.rank, aux, 0, m), self._epoch[(m[0], m[1], aux)]) for m in remote_in], stream)

    def reshare(self) -> None:
        """Collective: export / open every buffer again after an agent re-bound one (agent.X = ..., a second
        enable_acceleration()).  All in-flight work is drained first."""
        self.torch.cuda.synchronize()
        self.dist.barrier()
        self._remote.clear()
        for aux in list(self._shared_gen):
            self._share(aux)
        self._plan_gen = None

    def _exchange_host_barriers(self, msgs, key, aux: bool) -> None:
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
        if out:
            first = c.agents[out[0][0]]
            L.check(first.problem._lib.dpgo_exchange_plan_run(self._plan(out, key, aux),
                                                             torch.cuda.current_stream().cuda_stream or None))
            torch.cuda.current_stream().synchronize()
        dist.barrier()

    def close(self) -> None:
        self.torch.cuda.synchronize()
        try:
            self.check()
        finally:
            self._close()

    def _close(self) -> None:
        for h in self._plans.values():
            self._destroy(h)
        self._plans.clear()
        self._remote.clear()
        self._peer_words.clear()
