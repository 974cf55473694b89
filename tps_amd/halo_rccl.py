"""The native RCCL halo exchange (``include/tpsrhs_rccl.h``, ``tps_amd/csrc/halo_rccl.cpp``) as the ``halo``
argument of :class:`tps_amd.rhs_operator.RHSoperator`.

Nothing of the exchange runs in Python: the operator receives the C function pointers ``tpsrhs_rccl_halo`` /
``tpsrhs_rccl_reduce`` and the communicator context, so the two calls per ``Mult`` go C -> C
(``ncclGroupStart .. ncclSend / ncclRecv .. ncclGroupEnd`` on the operator's communication stream).  Python only
bootstraps the communicator: rank 0 draws the ``ncclUniqueId``, ``torch.distributed`` -- already initialised by
the launcher -- broadcasts its 128 bytes (the role ``MPI_Bcast`` has in a TPS integration).
"""
from __future__ import annotations

import ctypes as C
import os

import torch
import torch.distributed as dist

from . import capi

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libtpsrhs_rccl.so")
ID_BYTES = 128
EXPORTED_SYMBOLS = ["tpsrhs_rccl_unique_id", "tpsrhs_rccl_create", "tpsrhs_rccl_destroy", "tpsrhs_rccl_halo",
                    "tpsrhs_rccl_reduce", "tpsrhs_rccl_stats", "tpsrhs_rccl_set_skip", "tpsrhs_rccl_last_error",
                    "tpsrhs_rccl_comm_info"]
_LIB = None


def load():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise capi.LibraryMissing(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(LIB_PATH)  # torch (imported above) has loaded librccl.so.1 already: one RCCL per process
        lib.tpsrhs_rccl_last_error.restype = C.c_char_p
        lib.tpsrhs_rccl_unique_id.argtypes = [C.c_void_p]
        lib.tpsrhs_rccl_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        lib.tpsrhs_rccl_destroy.argtypes = [C.c_void_p]
        lib.tpsrhs_rccl_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int)]
        lib.tpsrhs_rccl_set_skip.argtypes = [C.c_void_p, C.c_int]
        _LIB = lib
    return _LIB


class RcclHalo:
    """One RCCL communicator over the ranks of the default ``torch.distributed`` group; ``device`` = this rank's GPU."""

    backend = "rccl (native: ncclSend/ncclRecv groups from libtpsrhs_rccl.so)"

    def __init__(self, device: int):
        if not dist.is_initialized():
            raise RuntimeError("RcclHalo: torch.distributed is not initialised (it carries the ncclUniqueId)")
        lib = load()
        rank, world = dist.get_rank(), dist.get_world_size()
        ident = (C.c_ubyte * ID_BYTES)()
        if rank == 0 and lib.tpsrhs_rccl_unique_id(ident) != 0:
            raise RuntimeError("tpsrhs_rccl_unique_id: " + lib.tpsrhs_rccl_last_error().decode())
        box = [bytes(ident) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        ident = (C.c_ubyte * ID_BYTES).from_buffer_copy(box[0])
        ctx = C.c_void_p()
        if lib.tpsrhs_rccl_create(ident, world, rank, int(device), C.byref(ctx)) != 0:
            raise RuntimeError("tpsrhs_rccl_create: " + lib.tpsrhs_rccl_last_error().decode())
        self._lib, self.ctx = lib, ctx
        self.world = world
        # what RHSoperator puts into tpsrhs_runtime: C function pointers, no Python frame on the data path
        self.c_halo = capi.HALO_FN(C.cast(lib.tpsrhs_rccl_halo, C.c_void_p).value)
        self.c_reduce = capi.REDUCE_FN(C.cast(lib.tpsrhs_rccl_reduce, C.c_void_p).value)

    @property
    def skip(self):
        return self._skip

    @skip.setter
    def skip(self, on):
        self._skip = bool(on)
        self._lib.tpsrhs_rccl_set_skip(self.ctx, 1 if on else 0)

    _skip = False

    def stats(self):
        calls, sent, peers = C.c_int64(0), C.c_int64(0), C.c_int(0)
        self._lib.tpsrhs_rccl_stats(self.ctx, C.byref(calls), C.byref(sent), C.byref(peers))
        n, sep = C.c_int(0), C.c_int(0)
        self._lib.tpsrhs_rccl_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        if self._lib.tpsrhs_rccl_comm_info(self.ctx, C.byref(n), C.byref(sep)) != 0:
            raise RuntimeError("tpsrhs_rccl_comm_info: " + self._lib.tpsrhs_rccl_last_error().decode())
        return {"halo_calls": calls.value, "bytes_sent": sent.value, "peers_seen": peers.value, "nranks": n.value,
                "reduce_comm": "own communicator (ncclCommSplit)" if sep.value else "shared with the exchange"}

    def close(self):
        if self.ctx:
            self._lib.tpsrhs_rccl_destroy(self.ctx)
            self.ctx = C.c_void_p()
