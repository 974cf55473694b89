"""Every device allocation poisoned with NaN (TPSRHS_POISON=1, operator.hpp dev_alloc): a kernel that reads memory no
kernel has written turns the residual into NaN instead of depending on what hipMalloc returned.  Representative
cases of every kernel family, the stateful boundary conditions over consecutive calls and the device time loop."""
import numpy as np
import pytest

from parity_util import RHS_RTOL, hip_mult, oracle_mult, rel_maxnorm
from tps_amd import capi, cases, meshgen
from tps_amd.rhs_operator import node_coordinates

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _poison(monkeypatch):
    monkeypatch.setenv("TPSRHS_POISON", "1")


def _check(mesh, disc, ph, bcs, U, tol):
    got = hip_mult(mesh, disc, ph, bcs, U)
    for k in ("y", "Up", "gradUp"):
        assert np.all(np.isfinite(got[k])), k
    ref = oracle_mult(mesh, disc, ph, bcs, U)
    assert rel_maxnorm(got["y"], ref["y"]).max() < tol


def test_poisoned_dry_air_cylinder():
    c = cases.cyl3d(4, 12, 3, 3, capi.NS, capi.VISC_ISOTH)
    c.physics.dry_air.visc_mult = 2000.0
    _check(c.mesh, c.disc, c.physics, c.bcs, c.state(seed=3), RHS_RTOL)


def test_poisoned_partial_blocks_p1_p2():
    """orders whose blocks hold several elements, with element counts that leave the last block partly empty"""
    for order, dims in ((1, (5, 8, 4)), (2, (3, 9, 3))):  # 160 elements in blocks of 3, 81 in blocks of 2
        c = cases.cyl3d(*dims, order, capi.NS, capi.VISC_ADIAB)
        c.physics.dry_air.visc_mult = 2000.0
        _check(c.mesh, c.disc, c.physics, c.bcs, c.state(seed=4), RHS_RTOL)


def test_poisoned_gll_pair():
    mesh = meshgen.scramble_orientations(meshgen.box_hex(3, 3, 3, lengths=(1.0, 0.8, 1.2), warp=0.1), 3)
    disc = capi.Disc(3, 1, 1, 0, 0)
    ph = capi.dry_air_physics(capi.NS, visc_mult=500.0)
    _check(mesh, disc, ph, [], cases.dry_air_state(node_coordinates(mesh, 3, 1), seed=6), 5 * RHS_RTOL)


def test_poisoned_plasma_and_axisymmetric():
    c = cases.argon_cyl3d(4, 12, 3, 3)
    _check(c.mesh, c.disc, c.physics, c.bcs, c.state(seed=2, amp=0.01), 5 * RHS_RTOL)
    c = cases.argon_axisym(6, 9, 3, True, capi.CONSTANT, "tabulated", True, capi.VISC_ISOTH)
    _check(c.mesh, c.disc, c.physics, c.bcs, c.state(seed=5, amp=0.01), 5 * RHS_RTOL)


def test_poisoned_nonreflecting_sequence_and_time_loop():
    import torch
    from oracle_lib import Oracle

    from tps_amd.rhs_operator import RHSoperator

    c = cases.cyl3d(4, 12, 3, 2, capi.NS, capi.VISC_ISOTH)
    c.physics.dry_air.visc_mult = 2000.0
    c.bcs[1] = capi.make_bc(2, capi.OUTLET, capi.SUB_P_NR, [101000.0, 0, 0, 0, 0.0, 0.0, 1.0, 0.0])
    U = c.state(seed=8)
    o = Oracle(c.mesh, c.disc, c.physics, c.bcs)
    op = RHSoperator(c.mesh, c.disc, c.physics, c.bcs)
    x = torch.tensor(U.ravel(), dtype=torch.float64, device=op.device)
    y = torch.empty_like(x)
    o.set_dt(3e-4)
    op.setDt(3e-4)
    for _ in range(3):
        yr = o.mult(U)
        op.Mult(x, y)
    assert rel_maxnorm(y.cpu().numpy().reshape(U.shape), yr).max() < RHS_RTOL
    ref = o.advance(U, 0.0, 2e-5, 4, False, 0.1, 0.05)
    t_end, dt_next, bad = op.advance(x, 0.0, 2e-5, 4, False, 0.1, 0.05)
    assert bad == 0 and np.all(np.isfinite(x.cpu().numpy()))
    assert rel_maxnorm(x.cpu().numpy().reshape(U.shape), ref[0]).max() < 1e-12
    op.close()
