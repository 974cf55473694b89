"""Mixed-out sponge zone (SpongeZoneSolution::MIXEDOUT, src/forcing_terms.cpp:713-743 with
DryAir::computeConservedStateFromConvectiveFlux, src/equation_of_state.cpp:414-444): HIP path vs the CPU oracle."""
import numpy as np
import pytest

from parity_util import RHS_RTOL, rel_maxnorm
from tps_amd import capi, cases, meshgen

pytestmark = pytest.mark.gpu


def _zone(dim, **kw):
    # the zone lies between point_init and point0, its normal points from point0 back to point_init
    # (sigma > 0 where -n.(x - point_init) > 0 and n.(x - point0) > 0, src/forcing_terms.cpp:556-570)
    z = dict(type=capi.SPONGE_PLANAR, solution_type=capi.SPONGE_MIXEDOUT, normal=(-1.0, 0.0, 0.0)[:dim],
             point0=(1.9, 0.0, 0.0)[:dim], point_init=(1.0, 0.0, 0.0)[:dim], tol=0.06, mult_factor=2.0, target_U=[])
    z.update(kw)
    return z


def _both(case, U, forcing):
    import torch
    from oracle_lib import Oracle
    from tps_amd.rhs_operator import RHSoperator

    o = Oracle(case.mesh, case.disc, case.physics, case.bcs)
    y0 = o.mult(U)
    o.set_forcing(forcing)
    y_ref = o.mult(U)
    op = RHSoperator(case.mesh, case.disc, case.physics, case.bcs)
    x = torch.tensor(np.ascontiguousarray(U).ravel(), dtype=torch.float64, device=op.device)
    y = torch.empty_like(x)
    op.setForcing(forcing)
    op.Mult(x, y)
    got = y.cpu().numpy().reshape(U.shape)
    op.close()
    return y0, y_ref, got


@pytest.mark.parametrize("dim,order", [(3, 2), (2, 3), (3, 1)])
def test_mixed_out_planar(dim, order):
    if dim == 3:
        mesh = meshgen.box_hex(8, 3, 3, lengths=(2.0, 1.0, 0.5), warp=0.0)
    else:
        mesh = meshgen.box_quad(8, 4, lengths=(2.0, 1.0), warp=0.0)
    c = cases.Case("mixedout", mesh, capi.Disc(order, 0, 0, 0, 0), capi.dry_air_physics(capi.NS), [])
    U = c.state(seed=4, amp=0.1)
    forcing = capi.make_forcing(sponge_zones=[_zone(dim)])
    y0, y_ref, got = _both(c, U, forcing)
    changed = np.abs(y_ref - y0).max(axis=1)
    print("sponge contribution", changed, "rel err", rel_maxnorm(got, y_ref))
    assert changed.min() > 0.0
    assert rel_maxnorm(got, y_ref).max() < RHS_RTOL
    scale = np.abs(y_ref).max(axis=1, keepdims=True)
    assert (np.abs((got - y0) - (y_ref - y0)) / scale).max() < RHS_RTOL


def test_mixed_out_of_a_uniform_stream_is_that_stream():
    """the mixed-out state of a uniform flow is the flow itself: the sponge adds nothing (a property, no oracle)"""
    import torch
    from tps_amd.rhs_operator import RHSoperator

    mesh = meshgen.box_hex(8, 3, 3, lengths=(2.0, 1.0, 0.5))
    c = cases.Case("mixedout_uniform", mesh, capi.Disc(2, 0, 0, 0, 0), capi.dry_air_physics(capi.NS), [])
    U = c.state(seed=1, amp=0.0)
    op = RHSoperator(c.mesh, c.disc, c.physics, c.bcs)
    x = torch.tensor(U.ravel(), dtype=torch.float64, device=op.device)
    y = torch.empty_like(x)
    op.setForcing(capi.make_forcing(sponge_zones=[_zone(3, mult_factor=50.0)]))
    op.Mult(x, y)
    r = y.cpu().numpy().reshape(U.shape)
    op.close()
    scale = np.abs(U).max(axis=1) * 340.0  # |U| c / L with L = 1
    scale[1:4] = scale[1:4].max()           # the cross-stream momenta are zero
    print("residual / scale", np.abs(r).max(axis=1) / scale)
    assert (np.abs(r).max(axis=1) / scale).max() < 1e-12


def test_mixed_out_argument_checks():
    from tps_amd.rhs_operator import RHSoperator
    mesh = meshgen.box_hex(4, 3, 3, lengths=(2.0, 1.0, 0.5))
    op = RHSoperator(mesh, capi.Disc(1, 0, 0, 0, 0), capi.dry_air_physics(capi.NS), [])
    with pytest.raises(Exception, match="tol"):
        op.setForcing(capi.make_forcing(sponge_zones=[_zone(3, tol=0.0)]))
    with pytest.raises(Exception, match="no node"):
        op.setForcing(capi.make_forcing(sponge_zones=[_zone(3, point_init=(1.013, 0, 0), tol=1e-6)]))
    op.close()
    pl = RHSoperator(mesh, capi.Disc(1, 0, 0, 0, 0), capi.argon_ternary_physics(), [])
    with pytest.raises(Exception, match="dry air"):
        pl.setForcing(capi.make_forcing(sponge_zones=[_zone(3)]))
    pl.close()
