"""MixingLengthTransport (src/mixing_length_transport.cpp:44-131; [flow] useMixingLength of test/inputs/plasma.ini:48 and
pipe.axisym.mix.ini:19): the algebraic eddy viscosity rho l^2 |S|, l = min(0.41 d, l_max), on top of the molecular
transport.  Oracle against the closed form (CPU), HIP path against the oracle (GPU)."""
import numpy as np
import pytest

from parity_util import RHS_RTOL, rel_maxnorm
from tps_amd import capi, cases
from tps_amd.rhs_operator import node_coordinates


def _pipe(order=3, wall=capi.VISC_ISOTH):
    c = cases.dry_air_axisym(5, 7, order, capi.NS, wall, r_in=0.0)
    c.physics.dry_air.visc_mult = 50.0
    return c


def _wall_distance(case, r_out=0.05):
    X = node_coordinates(case.mesh, case.disc.order)
    return np.ascontiguousarray(r_out - X[0])  # distance to the outer wall of the tube


def test_oracle_eddy_viscosity_closed_form():
    """axisymmetric shear du_z/dr = s at distance d from a wall: tau_zr = (mu + rho l^2 |S|) s with l = min(0.41 d, l_max)
    and |S|^2 = 2 S_ij S_ij incl. the hoop strain u_r / r; the eddy conductivity is mu_t (kappa / mu) Pr_ratio"""
    import ctypes as C

    from oracle_lib import Oracle, _dp, _p, lib

    c = _pipe(2)
    o = Oracle(c.mesh, c.disc, c.physics, c.bcs)
    L = lib()
    L.tpsoracle_point_viscous_flux_dist.argtypes = [C.c_void_p, _dp, _dp, C.c_double, C.c_double, _dp]
    neq = 5
    U = np.array([1.2, 1.2 * 3.0, 1.2 * 40.0, 0.0, 101300.0 / 0.4 + 0.6 * (9.0 + 1600.0)])
    s, dT, radius = 800.0, 2.0e4, 0.02
    g = np.zeros(neq * 2)
    g[2 + 0 * neq] = s   # d u_z / d r
    g[4 + 0 * neq] = dT  # d T / d r

    def flux(d):
        out = np.zeros(neq * 2)
        L.tpsoracle_point_viscous_flux_dist(o.h, _p(U), _p(g), radius, d, _p(out))
        return out

    def heat(f):  # the conduction part of the radial energy flux: minus the work of the stresses
        return f[4 + 0 * neq] - f[1 + 0 * neq] * U[1] / U[0] - f[2 + 0 * neq] * U[2] / U[0]

    f0 = flux(0.004)
    mu, kappa = f0[2 + 0 * neq] / s, heat(f0) / dT
    dist = np.zeros(o.ndofs)  # the grid function enters Mult only; the point routine takes d directly
    for lmax, prt, bulk, d in ((1.0, 1.0, 0.0, 0.004), (1.0e-3, 0.9, 0.0, 0.004), (1.0, 0.7, 2.0, 0.01)):
        o.set_mixing_length(dist, lmax, prt, 1.0, bulk)
        f = flux(d)
        l = min(0.41 * d, lmax)
        S = np.sqrt(2 * (2 * (0.5 * s) ** 2) + 2 * (U[1] / U[0] / radius) ** 2)
        mut = U[0] * l * l * S
        assert abs(f[2 + 0 * neq] - (mu + mut) * s) < 1e-12 * abs(f[2 + 0 * neq])
        assert abs(heat(f) - (kappa + mut * (kappa / mu) * prt) * dT) < 1e-10 * abs(heat(f))
    o.set_mixing_length(None)
    assert np.array_equal(flux(0.004), f0)


def _both(case, U, dist, **prm):
    import torch
    from oracle_lib import Oracle
    from tps_amd.rhs_operator import RHSoperator

    o = Oracle(case.mesh, case.disc, case.physics, case.bcs)
    y0 = o.mult(U)
    o.set_mixing_length(dist, **prm)
    y_ref = o.mult(U)
    op = RHSoperator(case.mesh, case.disc, case.physics, case.bcs)
    x = torch.tensor(np.ascontiguousarray(U).ravel(), dtype=torch.float64, device=op.device)
    y = torch.empty_like(x)
    d = torch.tensor(dist, dtype=torch.float64, device=op.device)
    op.setMixingLength(d, **prm)
    op.Mult(x, y)
    got = y.cpu().numpy().reshape(U.shape)
    op.setMixingLength(None)
    op.Mult(x, y)
    got0 = y.cpu().numpy().reshape(U.shape)
    op.close()
    return y0, y_ref, got, got0


@pytest.mark.gpu
@pytest.mark.parametrize("order,wall", [(3, capi.VISC_ISOTH), (2, capi.VISC_ADIAB), (1, capi.INV)])
def test_dry_air_pipe(order, wall):
    """the setting of test/inputs/pipe.axisym.mix.ini: axisymmetric dry air in a tube, distance to its wall"""
    c = _pipe(order, wall)
    U = c.state(seed=5, amp=0.05)
    y0, y_ref, got, got0 = _both(c, U, _wall_distance(c), max_mixing_length=0.004, pr_ratio=0.9, bulk_multiplier=0.5)
    change = np.abs(y_ref - y0).max(axis=1) / np.abs(y_ref).max(axis=1)
    print("change by the model", change, "rel err", rel_maxnorm(got, y_ref))
    assert change[1:].max() > 1e-4
    assert rel_maxnorm(got, y_ref).max() < RHS_RTOL
    assert rel_maxnorm(got0, y0).max() < RHS_RTOL  # ... and off again


@pytest.mark.gpu
@pytest.mark.parametrize("fluid,order", [("torch6", 3), ("ternary_axi", 2), ("ternary_planar", 3)])
def test_plasma(fluid, order):
    """test/inputs/plasma.ini:48: the six-species two-temperature torch mixture, axisymmetric, with the mixing-length
    model; and the ternary mixture, axisymmetric and planar"""
    from tps_amd import meshgen

    if fluid == "torch6":
        ph = capi.argon_six_species_physics(capi.NS, capi.ARGON_MIXTURE, True, True, radiation=True)
    else:
        ph = capi.argon_ternary_physics(capi.NS, fluid == "ternary_axi", capi.ARGON_MINIMAL, "arrhenius")
    if fluid == "ternary_planar":
        mesh = meshgen.box_quad(5, 4, lengths=(0.2, 0.1), warp=0.08)
        c = cases.Case("planar", mesh, capi.Disc(order, 0, 0, 0, 0), ph, [])
        X = node_coordinates(mesh, order)
        dist = 0.02 + 0.01 * np.sin(2 * np.pi * X[0] / 0.2) * np.cos(2 * np.pi * X[1] / 0.1)
        U = cases.plasma_state(X, ph, nvel=2, seed=3, amp=0.01)
    else:
        c = cases.argon_axisym(5, 7, order, physics=ph, r_in=0.0)
        dist = _wall_distance(c)
        U = c.state(seed=3, amp=0.01)
    y0, y_ref, got, got0 = _both(c, U, dist, max_mixing_length=0.004, pr_ratio=0.9, bulk_multiplier=0.0)
    change = np.abs(y_ref - y0).max(axis=1) / np.abs(y_ref).max(axis=1)
    print("change by the model", change, "rel err", rel_maxnorm(got, y_ref))
    assert change[1:4].max() > 1e-5
    assert rel_maxnorm(got, y_ref).max() < 5 * RHS_RTOL
    assert rel_maxnorm(got0, y0).max() < 5 * RHS_RTOL


@pytest.mark.gpu
def test_unsupported_configurations():
    import torch
    from tps_amd.rhs_operator import RHSoperator

    c = cases.cyl3d(3, 8, 3, 2)
    op = RHSoperator(c.mesh, c.disc, c.physics, c.bcs)
    d = torch.zeros(op.NDofs, dtype=torch.float64, device=op.device)
    with pytest.raises(Exception, match="2-D"):
        op.setMixingLength(d, 0.01)
    op.close()
