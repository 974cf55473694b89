"""Operator lifetime: tpsrhs_destroy returns every device allocation (RHSoperator::~RHSoperator frees the
operator-owned temporaries, src/rhs_operator.cpp:324-341), for every optional subsystem."""
import numpy as np
import pytest

from tps_amd import capi, cases

pytestmark = pytest.mark.gpu


def _exercise(kind, side):
    import torch
    from tps_amd.rhs_operator import RHSoperator

    if kind == "plasma":
        c = cases.argon_cyl3d(4, 12, 3, 2, True, capi.ARGON_MINIMAL, "arrhenius", capi.VISC_ISOTH)
        U = c.state(seed=1, amp=0.01)
    else:
        c = cases.cyl3d(4, 12, 3, 3, capi.NS, capi.VISC_ISOTH)
        c.bcs[1] = capi.make_bc(2, capi.OUTLET, capi.SUB_P_NR, [101000.0, 0, 0, 0, 0.0, 0.0, 1.0, 0.0])
        U = c.state(seed=1)
    with torch.cuda.stream(side):
        op = RHSoperator(c.mesh, c.disc, c.physics, c.bcs, stream=side)
        x = torch.tensor(np.ascontiguousarray(U).ravel(), dtype=torch.float64, device=op.device)
        y = torch.empty_like(x)
        op.setForcing(capi.make_forcing(pressure_gradient=(1.0, 0.0, 0.0)))
        op.setJouleHeating(torch.zeros(op.NDofs, dtype=torch.float64, device=op.device))
        op.Mult(x, y, want_max_char_speed=True)
        op.enable_kernel_timing(True)
        op.Mult(x, y)
        op.kernel_times()
        op.enable_kernel_timing(False)
        op.advance(x, 0.0, 1e-9, 4, True)  # RK buffers, control block, captured step graph
        side.synchronize()
        op.close()
    del x, y


@pytest.mark.parametrize("kind", ["dry_air_nr", "plasma"])
def test_destroy_returns_the_device_memory(kind):
    import torch

    # one stream for all cycles: the HIP runtime gives every new hardware queue its own scratch arena (hundreds of
    # MB as soon as a kernel uses scratch), which is the runtime's to keep
    side = torch.cuda.Stream()
    for _ in range(2):  # first uses: one-time allocations (code objects, constant tables, queue scratch)
        _exercise(kind, side)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(5):
        _exercise(kind, side)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free1, _ = torch.cuda.mem_get_info()
    print("free before / after five create-destroy cycles:", free0, free1)
    assert free0 - free1 < 8 << 20  # nothing of an operator's footprint (tens of MB here) stays behind
