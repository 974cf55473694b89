"""Stream ordering of the Python halo / reduce hooks (tps_amd/halo.py) against the operator's streams.

Root cause of the intermittent 3-rank failure of round 2 (``test_ranks_match_serial_oracle[3-dry_air_nr]``, 4.7e-4 in
the ``tpsrhs_advance`` leg): the operator of the tests runs on torch's current stream, i.e. the legacy NULL stream,
and hands that pointer (0) to ``tpsrhs_reduce_fn``; the hook wrapped it as ``torch.cuda.ExternalStream(0)``, which is
NOT the default stream -- with a zero pointer torch takes a fresh non-blocking stream from its pool.  The hook's
device -> host copy of the boundary-patch sums (and of dt) therefore was not ordered after ``k_bc_mean`` /
``k_step_end``: it usually won the race because the host is slow, and lost it now and then with three processes
time-slicing one GPU -- the ranks then summed stale or partial values and advanced the boundary state with a
slightly wrong patch mean.  These tests fail deterministically on the old hook."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_null_stream_pointer_maps_to_the_default_stream():
    import torch

    from tps_amd.halo import HaloExchange

    h = HaloExchange(device=torch.device("cuda", 0))
    assert h._torch_stream(0).cuda_stream == 0
    assert h._torch_stream(None).cuda_stream == 0
    side = torch.cuda.Stream()
    assert h._torch_stream(side.cuda_stream).cuda_stream == side.cuda_stream
    # what the old code did: a zero pointer does not wrap the NULL stream
    assert torch.cuda.ExternalStream(0).cuda_stream != 0


@pytest.mark.parametrize("use_null_stream", [True, False])
def test_reduce_hook_is_ordered_after_pending_kernels(monkeypatch, use_null_stream):
    """A long kernel, then the producer of the value, then the hook -- all on the operator's stream: the hook must see
    the produced value (the role of k_bc_mean / k_step_end before tpsrhs_reduce_fn, operator.hpp nr_update)."""
    import torch
    import torch.distributed as dist

    from tps_amd.halo import HaloExchange

    dev = torch.device("cuda", 0)
    seen = []
    monkeypatch.setattr(dist, "all_reduce", lambda t, op=None, group=None: seen.append(t.clone()))
    h = HaloExchange(device=dev)
    stream = torch.cuda.default_stream(dev) if use_null_stream else torch.cuda.Stream(dev)
    v = torch.zeros(14, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        torch.cuda._sleep(400_000_000)  # ~0.2 s of device time: the host reaches the hook long before it ends
        v.fill_(7.0)
    assert h.reduce_callback(None, v.data_ptr(), v.numel(), 0, stream.cuda_stream or None) == 0
    torch.cuda.synchronize()
    assert len(seen) == 1 and np.all(seen[0].numpy() == 7.0), seen
    assert np.all(v.cpu().numpy() == 7.0)


def test_halo_hook_is_ordered_on_the_null_stream(monkeypatch):
    """the exchange hook with a NULL stream pointer (an operator without a second stream would pass it)"""
    import ctypes as C

    import torch
    import torch.distributed as dist

    from tps_amd.halo import HaloExchange

    dev = torch.device("cuda", 0)
    sent = []

    def fake_batch(ops):
        sent.extend(op.tensor.clone() for op in ops if op.op is dist.isend)
        return []

    monkeypatch.setattr(dist, "batch_isend_irecv", fake_batch)
    from types import SimpleNamespace

    monkeypatch.setattr(dist, "P2POp", lambda op, tensor, peer, group=None: SimpleNamespace(op=op, tensor=tensor))
    h = HaloExchange(device=dev)
    send = torch.zeros(32, dtype=torch.float64, device=dev)
    recv = torch.zeros(32, dtype=torch.float64, device=dev)
    off = (C.c_int64 * 2)(0, 32)
    ranks = (C.c_int * 1)(0)
    torch.cuda.synchronize()
    torch.cuda._sleep(400_000_000)
    send.fill_(3.0)
    assert h.callback(None, 0, send.data_ptr(), recv.data_ptr(), 1, ranks, off, off, None) == 0
    torch.cuda.synchronize()
    assert len(sent) == 1 and np.all(sent[0].numpy() == 3.0)
