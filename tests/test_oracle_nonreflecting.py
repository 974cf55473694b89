"""Known answers for the oracle's restatement of the non-reflecting inlet / outlet conditions
(src/inletBC.cpp:576-727, src/outletBC.cpp:573-1027)."""
import numpy as np
import pytest

from oracle_lib import Oracle
from tps_amd import capi, cases, meshgen
from tps_amd.rhs_operator import node_coordinates

GAMMA, RG = 1.4, 287.058
RHO, VEL, P = 1.2, (30.0, 0.0, 0.0), 101300.0


def _channel(order, inlet, outlet, dim=3):
    """x-channel, periodic across: patch 1 = x-min, 2 = x-max."""
    if dim == 3:
        attrs = {(0, 0): 1, (0, 1): 2}
        mesh = meshgen.box_hex(4, 3, 3, lengths=(1.0, 0.6, 0.6), periodic=(False, True, True), bdr_attr=attrs)
    else:
        attrs = {(0, 0): 1, (0, 1): 2}
        mesh = meshgen.box_quad(4, 3, lengths=(1.0, 0.6), periodic=(False, True), bdr_attr=attrs)
    return mesh, capi.Disc(order, 0, 0, 0, 0), capi.dry_air_physics(capi.NS, visc_mult=50.0), [inlet, outlet]


def _uniform(X, dim, rho=RHO, vel=VEL, p=P):
    U = np.zeros((dim + 2, X.shape[1]))
    U[0] = rho
    for d in range(dim):
        U[1 + d] = rho * vel[d]
    U[dim + 1] = p / (GAMMA - 1) + 0.5 * rho * sum(v * v for v in vel[:dim])
    return U


def _nr(cat, typ, data, tangent=(0.0, 1.0, 0.0), area=0.0):
    d = list(data) + [0.0] * (4 - len(data)) + list(tangent) + [area]
    return capi.make_bc(1 if cat == capi.INLET else 2, cat, typ, d)


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("outlet_type", [capi.SUB_P_NR, capi.SUB_MF_NR, capi.SUB_MF_NR_PW])
def test_matched_free_stream_is_a_fixed_point(dim, outlet_type):
    """Targets equal to the uniform flow: every characteristic amplitude vanishes, the boundary state stays at
    the flow state, Mult returns zero -- call after call."""
    area = 0.36 if dim == 3 else 0.6
    data = [P] if outlet_type == capi.SUB_P_NR else [RHO * VEL[0] * area]
    inlet = _nr(capi.INLET, capi.SUB_DENS_VEL_NR, [RHO, *VEL])
    outlet = _nr(capi.OUTLET, outlet_type, data, area=area)
    mesh, disc, ph, bcs = _channel(2, inlet, outlet, dim)
    o = Oracle(mesh, disc, ph, bcs)
    o.set_dt(1.0e-5)
    U = _uniform(node_coordinates(mesh, 2), dim)
    for _ in range(3):
        y = o.mult(U)
        assert np.abs(y).max() < 1e-6 * P
    bu, mean = o.boundary_state(2)
    np.testing.assert_allclose(mean[0], RHO, rtol=1e-12)
    np.testing.assert_allclose(mean[dim + 1], P / (RHO * RG), rtol=1e-12)
    np.testing.assert_allclose(bu, np.broadcast_to(U[:, 0], bu.shape), rtol=1e-10, atol=1e-9)


def test_pressure_outlet_relaxes_towards_its_target():
    """Uniform flow with p above the outlet target: L1 = sigma (p_mean - p_target) > 0 is the only non-zero
    amplitude, so d(rho)/dt = -L1 / (2 c^2), d(rho E)/dt has the sign of -L1: the boundary pressure falls; the
    closed form of one update (src/outletBC.cpp:640-724)."""
    p_target, dt, ref_len = 0.9 * P, 2.0e-5, 0.5
    inlet = capi.make_bc(1, capi.INLET, capi.SUB_DENS_VEL, [RHO, *VEL])
    outlet = _nr(capi.OUTLET, capi.SUB_P_NR, [p_target])
    mesh, disc, ph, bcs = _channel(1, inlet, outlet, 3)
    disc.ref_length = ref_len
    o = Oracle(mesh, disc, ph, bcs)
    o.set_dt(dt)
    U = _uniform(node_coordinates(mesh, 1), 3)
    o.mult(U)
    bu, mean = o.boundary_state(2)
    c = np.sqrt(GAMMA * P / RHO)
    L1 = (c / ref_len) * (P - p_target)
    d1, d2, d5 = 0.5 * L1 / c**2, -0.5 * L1 / (RHO * c), 0.5 * L1
    un = VEL[0]  # outward normal = +x
    expect = U[:, 0].copy()
    expect[0] -= dt * d1
    expect[1] -= dt * (un * d1 + RHO * d2)
    expect[4] -= dt * (RHO * un * d2 + 0.5 * un * un * d1 + d5 / (GAMMA - 1))
    np.testing.assert_allclose(bu, np.broadcast_to(expect, bu.shape), rtol=1e-10)
    p_new = (GAMMA - 1) * (bu[:, 4] - 0.5 * (bu[:, 1] ** 2 + bu[:, 2] ** 2 + bu[:, 3] ** 2) / bu[:, 0])
    assert np.all(p_new < P)
    # the state keeps moving in the following calls (every Mult advances it, src/outletBC.cpp:712-724)
    o.mult(U)
    bu2, _ = o.boundary_state(2)
    assert np.all(bu2[:, 0] < bu[:, 0])


def test_non_reflecting_needs_perfect_gas():
    c = cases.argon_cyl3d(3, 8, 3, 1, False, capi.CONSTANT, "arrhenius", capi.VISC_ISOTH)
    c.bcs[1] = _nr(capi.OUTLET, capi.SUB_P_NR, [101300.0])
    with pytest.raises(RuntimeError):
        Oracle(c.mesh, c.disc, c.physics, c.bcs)
