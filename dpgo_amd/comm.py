"""RCCL transport of the public-pose exchange: Python face of the C ABI dpgo_comm_* (include/dpgo_hip.h).

One process per GPU; rank 0 creates the RCCL unique id and the other ranks receive it through whatever channel the
launcher provides (here: torch.distributed's rendezvous).  After that the data path -- grouped ncclSend / ncclRecv of
packed pose tiles, tiny all-reduces, the anchor broadcast -- is issued by the solver library itself on the solver's HIP
stream; torch.distributed is only the rendezvous and the fallback transport."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import lib as L

ID_BYTES = 128
SUM, MAX = 0, 1


def unique_id() -> bytes:
    buf = C.create_string_buffer(ID_BYTES)
    L.check(L.load().dpgo_comm_unique_id(buf))
    return buf.raw


class DeviceComm:
    """dpgo_comm_t: an RCCL communicator owned by the solver library."""

    def __init__(self, nranks: int, rank: int, uid: bytes, device: int = 0):
        if len(uid) != ID_BYTES:
            raise ValueError("RCCL unique id must be %d bytes" % ID_BYTES)
        self._lib = L.load()
        self._h = L._P()
        self.nranks, self.rank, self.device = int(nranks), int(rank), int(device)
        L.check(self._lib.dpgo_comm_create(C.byref(self._h), self.nranks, self.rank, uid, self.device))

    @staticmethod
    def from_torch_distributed(device: int) -> "DeviceComm":
        """Collective: rank 0's unique id travels over the initialised torch.distributed process group."""
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return DeviceComm(world, rank, box[0], device)

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.dpgo_comm_destroy(self._h)
            self._h = L._P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def exchange(self, sends: Sequence[Tuple[int, object]], recvs: Sequence[Tuple[int, object]],
                 stream: Optional[int] = None) -> None:
        """One grouped batch: sends / recvs = [(peer rank, contiguous float64 device tensor)], enqueued on `stream`
        (None = the default stream)."""
        def pack(msgs):
            n = len(msgs)
            peers = (C.c_int * max(n, 1))(*[int(p) for p, _ in msgs])
            ptrs = (C.c_void_p * max(n, 1))(*[L.ptr(t) for _, t in msgs])
            counts = (C.c_int * max(n, 1))(*[int(t.numel()) for _, t in msgs])
            return n, peers, ptrs, counts
        for _, t in list(sends) + list(recvs):
            if not t.is_contiguous():
                raise ValueError("exchange buffers must be contiguous")
        ns, sp, sb, sc = pack(sends)
        nr, rp, rb, rcnt = pack(recvs)
        L.check(self._lib.dpgo_comm_exchange(self._h, ns, sp, sb, sc, nr, rp, rb, rcnt, stream or None))

    def allreduce(self, t, op: int = SUM, stream: Optional[int] = None) -> None:
        """In place on a contiguous float64 device tensor."""
        L.check(self._lib.dpgo_comm_allreduce(self._h, L.ptr(t), int(t.numel()), int(op), stream or None))

    def broadcast(self, t, root: int, stream: Optional[int] = None) -> None:
        L.check(self._lib.dpgo_comm_broadcast(self._h, L.ptr(t), int(t.numel()), int(root), stream or None))
