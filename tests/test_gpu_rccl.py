"""libtpsrhs_rccl.so on a real RCCL communicator.  The one-GPU test box allows a communicator of ONE rank: enough
to run the library's own code -- unique id, ncclCommInitRank, a grouped ncclSend / ncclRecv on a side stream
(a rank may exchange with itself), ncclAllReduce, the counters -- through the same function pointers the operator
calls.  The partitioned operator over RCCL needs one GPU per rank: tests/test_gpu_multirank.py (skipped below that)."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_one_rank_communicator_moves_segments_and_reduces():
    import torch
    import torch.distributed as dist

    from tps_amd.halo_rccl import RcclHalo

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
    try:
        halo = RcclHalo(0)
        n = 4096
        send = torch.arange(n, dtype=torch.float64, device="cuda") * 0.5 + 1.0
        recv = torch.zeros(n, dtype=torch.float64, device="cuda")
        ranks = (C.c_int * 2)(0, 0)
        soff = (C.c_int64 * 3)(0, 1000, n)   # two segments, both to / from rank 0
        roff = (C.c_int64 * 3)(0, 1000, n)
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        st = halo.c_halo(halo.ctx, 0, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), 2, ranks, soff, roff,
                         C.c_void_p(side.cuda_stream))
        assert st == 0
        side.synchronize()
        assert torch.equal(recv, send)
        v = torch.tensor([3.0, -1.5, 7.25], dtype=torch.float64, device="cuda")
        for op in (0, 1):  # SUM, MIN over one rank: identity
            assert halo.c_reduce(halo.ctx, C.c_void_p(v.data_ptr()), 3, op, C.c_void_p(side.cuda_stream)) == 0
        side.synchronize()
        assert v.cpu().tolist() == [3.0, -1.5, 7.25]
        s = halo.stats()
        assert {k: s[k] for k in ("halo_calls", "bytes_sent", "peers_seen", "nranks")} == {
            "halo_calls": 1, "bytes_sent": 8 * n, "peers_seen": 1, "nranks": 1}  # nranks = ncclCommCount
        # default: the reductions share the exchange's communicator (the second one is opt-in, TPSRHS_RCCL_SPLIT=1)
        assert s["reduce_comm"].startswith("shared")
        halo.skip = True
        recv.zero_()
        assert halo.c_halo(halo.ctx, 1, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), 2, ranks, soff, roff,
                           C.c_void_p(side.cuda_stream)) == 0
        side.synchronize()
        assert float(recv.abs().max()) == 0.0 and halo.stats()["halo_calls"] == 1
        halo.close()
        # opt-in: a communicator of their own for the reductions, agreed on by every rank (here: one) before create returns
        os.environ["TPSRHS_RCCL_SPLIT"] = "1"
        try:
            halo2 = RcclHalo(0)
            v2 = torch.tensor([1.0, 2.0], dtype=torch.float64, device="cuda")
            assert halo2.c_reduce(halo2.ctx, C.c_void_p(v2.data_ptr()), 2, 0, C.c_void_p(side.cuda_stream)) == 0
            side.synchronize()
            assert v2.cpu().tolist() == [1.0, 2.0]
            assert halo2.stats()["reduce_comm"].startswith("own communicator")
            halo2.close()
        finally:
            del os.environ["TPSRHS_RCCL_SPLIT"]
    finally:
        dist.destroy_process_group()
