"""The non-collocated Gauss-Lobatto pair (basisType 1, integrationRule 1: the reference's defaults,
src/M2ulPhyS.cpp:2671-2672) on the HIP path: the one number the reference holds for this path -- the error
windows of test/gradient.test -- reproduced by the kernels themselves, and HIP-vs-oracle parity of the full
operator (dense element mass matrices, volume and face terms at quadrature points that are not the nodes)."""
import numpy as np
import pytest

from parity_util import RHS_RTOL, hip_mult, oracle_mult, rel_maxnorm
from tps_amd import capi, cases, meshgen
from tps_amd.rhs_operator import RHSoperator, node_coordinates

pytestmark = pytest.mark.gpu
GLL = (1, 1)


def _compare(mesh, disc, ph, bcs, U, tol=RHS_RTOL):
    ref = oracle_mult(mesh, disc, ph, bcs, U)
    got = hip_mult(mesh, disc, ph, bcs, U)
    e_up = rel_maxnorm(got["Up"], ref["Up"])
    e_y = rel_maxnorm(got["y"], ref["y"])
    scale = np.abs(ref["gradUp"]).max()
    e_g = np.abs(got["gradUp"] - ref["gradUp"]).max() / scale
    print("rel err Up", e_up.max(), "gradUp", e_g, "y", e_y)
    assert e_up.max() < 1e-13
    assert e_g < tol
    assert e_y.max() < tol
    assert abs(got["max_char_speed"] - ref["max_char_speed"]) < 1e-12 * ref["max_char_speed"]


def test_gradient_test_error_windows_on_hip():
    """test/gradient.test:30-46, test/test_gradient.cpp:61-87,159-162,228-238: a random sine field on the 160x160
    periodic quad mesh of side 5, p = 2, Gauss-Lobatto basis and rule; the relative L2 error of d(rho)/dx must lie
    in [2.295e-4, 2.305e-4] and of d(rho)/dy in [5.74e-5, 5.75e-5] ("empirically observed" 2.2988e-4 / 5.7471e-5).
    The gradient is computed by k_traces + k_gradient (tpsrhs_update_gradients); the oracle only supplies the L2
    norm, as test_gradient.cpp's ComputeLpError does for the reference."""
    import torch
    from oracle_lib import Oracle

    m = meshgen.box_quad(160, 160, lengths=(5.0, 5.0))
    disc = capi.Disc(2, 1, 1, 0, 0)
    ph = capi.dry_air_physics(capi.NS)
    o = Oracle(m, disc, ph)
    X = o.node_coords()
    assert np.abs(X - node_coordinates(m, 2, 1)).max() < 1e-13
    rng = np.random.default_rng(2024)
    L, kx = np.array([5.0, 5.0]), np.array([2, 1])
    Up = np.zeros((o.neq, o.ndofs))
    ex = np.zeros((2, o.neq, o.ndofs))
    for eq in range(o.neq):
        up0 = 1.0 + rng.random()
        Up[eq] = up0
        for d in range(2):
            dup, off = 0.1 * up0 / 2 * rng.random(), rng.random()
            Up[eq] += dup * np.sin(2 * np.pi * kx[d] * (X[d] / L[d] - off))
            ex[d, eq] = dup * 2 * np.pi * kx[d] / L[d] * np.cos(2 * np.pi * kx[d] * (X[d] / L[d] - off))
    # the conserved state whose primitives are this field (rho, u, v, T): GetConservativesFromPrimitives
    Rg, gam = ph.dry_air.gas_constant, ph.dry_air.specific_heat_ratio
    U = np.zeros_like(Up)
    U[0] = Up[0]
    U[1], U[2] = Up[0] * Up[1], Up[0] * Up[2]
    U[3] = Up[0] * Rg * Up[3] / (gam - 1.0) + 0.5 * Up[0] * (Up[1] ** 2 + Up[2] ** 2)
    op = RHSoperator(m, disc, ph, [])
    x = torch.tensor(U.ravel(), dtype=torch.float64, device=op.device)
    op.updateGradients(x)
    g = op.getGradients().cpu().numpy()  # [d][eq][n]
    assert np.abs(op.getPrimitives().cpu().numpy() - Up).max() < 1e-13 * np.abs(Up).max()
    op.close()
    e0 = o.l2_norm(g[0, 0], ex[0, 0]) / o.l2_norm(g[0, 0])
    e1 = o.l2_norm(g[1, 0], ex[1, 0]) / o.l2_norm(g[1, 0])
    print("gradient.test on the HIP path:", e0, e1)
    assert 2.295e-4 < e0 < 2.305e-4, e0
    assert 5.74e-5 < e1 < 5.75e-5, e1
    assert abs(e0 - 2.2988e-4) < 1e-8 and abs(e1 - 5.7471e-5) < 1e-9


def test_gradient_exact_for_polynomials_gll():
    import torch

    m = meshgen.scramble_orientations(meshgen.box_hex(3, 3, 3, periodic=(False, False, False)), 3)
    bcs = [capi.make_bc(a, capi.WALL, capi.INV) for a in range(1, 7)]
    disc = capi.Disc(2, 1, 1, 0, 0)
    ph = capi.dry_air_physics(capi.NS)
    X = node_coordinates(m, 2, 1)
    Up = np.zeros((5, X.shape[1]))
    Up[0] = 1 + X[0] ** 2 + X[1] * X[2]
    Up[1] = X[0] * X[1]
    Up[2] = X[2] ** 2
    Up[3] = 1.0
    Up[4] = 300.0 + X[0] + 2 * X[1] + 3 * X[2]
    Rg, gam = ph.dry_air.gas_constant, ph.dry_air.specific_heat_ratio
    U = np.zeros_like(Up)
    U[0] = Up[0]
    for d in range(3):
        U[1 + d] = Up[0] * Up[1 + d]
    U[4] = Up[0] * Rg * Up[4] / (gam - 1.0) + 0.5 * Up[0] * (Up[1] ** 2 + Up[2] ** 2 + Up[3] ** 2)
    op = RHSoperator(m, disc, ph, bcs)
    x = torch.tensor(U.ravel(), dtype=torch.float64, device=op.device)
    op.updateGradients(x)
    g = op.getGradients().cpu().numpy()
    op.close()
    assert np.abs(g[0, 0] - 2 * X[0]).max() < 1e-11
    assert np.abs(g[1, 0] - X[2]).max() < 1e-11
    assert np.abs(g[2, 0] - X[1]).max() < 1e-11
    assert np.abs(g[:, 3]).max() < 1e-11
    assert np.abs(g[2, 4] - 3).max() < 1e-10


@pytest.mark.parametrize("order", [1, 2, 3])
def test_periodic_box_quad_gll(order):
    mesh = meshgen.scramble_orientations(meshgen.box_quad(7, 5, lengths=(1.0, 0.7), warp=0.1), 2)
    disc = capi.Disc(order, 1, 1, 0, 0)
    ph = capi.dry_air_physics(capi.NS, visc_mult=300.0)
    U = cases.dry_air_state(node_coordinates(mesh, order, 1), seed=9)
    _compare(mesh, disc, ph, [], U)


@pytest.mark.parametrize("order", [1, 2, 3])
def test_periodic_box_hex_gll(order):
    mesh = meshgen.scramble_orientations(meshgen.box_hex(4, 3, 5, lengths=(1.0, 0.8, 1.2), warp=0.12), 11 + order)
    disc = capi.Disc(order, 1, 1, 0, 0)
    ph = capi.dry_air_physics(capi.NS, visc_mult=500.0, bulk_visc_mult=2.0)
    U = cases.dry_air_state(node_coordinates(mesh, order, 1), seed=3 + order)
    _compare(mesh, disc, ph, [], U)


@pytest.mark.parametrize("order,eq,wall", [(1, capi.EULER, capi.INV), (2, capi.NS, capi.VISC_ADIAB),
                                            (3, capi.NS, capi.VISC_ISOTH)])
def test_cylinder_gll(order, eq, wall):
    c = cases.cyl3d(5, 12, 4, order, eq, wall)
    c.mesh = meshgen.scramble_orientations(c.mesh, 5)
    c.disc = capi.Disc(order, 1, 1, 0, 0)
    c.physics.dry_air.visc_mult = 2000.0
    U = cases.dry_air_state(node_coordinates(c.mesh, order, 1), seed=77)
    _compare(c.mesh, c.disc, c.physics, c.bcs, U)


def test_use_bc_in_grad_gll():
    c = cases.cyl3d(4, 12, 3, 2, capi.NS, capi.VISC_ISOTH)
    c.disc = capi.Disc(2, 1, 1, 0, 1)
    c.physics.dry_air.visc_mult = 2000.0
    U = cases.dry_air_state(node_coordinates(c.mesh, 2, 1), seed=5)
    _compare(c.mesh, c.disc, c.physics, c.bcs, U)


@pytest.mark.parametrize("dim,order,two_t", [(3, 2, False), (2, 3, True), (3, 3, False)])
def test_argon_ternary_gll(dim, order, two_t):
    """basisType = integrationRule = 1 of test/inputs/argonMinimal.ini:9-10 with its physics: the argon ternary
    plasma with the argon-minimal collision-integral transport and the two Arrhenius reactions"""
    ph = capi.argon_ternary_physics(capi.NS, two_t, capi.ARGON_MINIMAL, "arrhenius")
    if dim == 3:
        c = cases.argon_cyl3d(4, 12, 3, order, physics=ph)
        mesh, bcs = c.mesh, c.bcs
    else:
        mesh = meshgen.scramble_orientations(meshgen.box_quad(6, 5, lengths=(1.0, 0.7), warp=0.08), 4)
        bcs = []
    disc = capi.Disc(order, 1, 1, 0, 0)
    U = cases.plasma_state(node_coordinates(mesh, order, 1), ph, nvel=dim, seed=6, amp=0.01)
    _compare(mesh, disc, ph, bcs, U, tol=5e-11)


@pytest.mark.parametrize("levels,ambi,dim,order,transport,two_t", [
    (1, True, 2, 2, capi.ARGON_MIXTURE, True), (2, False, 3, 3, capi.CONSTANT, True), (3, False, 3, 2, capi.ARGON_MIXTURE, True),
    (3, True, 2, 3, capi.CONSTANT, False), (4, False, 2, 3, capi.ARGON_MIXTURE, True), (5, True, 3, 1, capi.CONSTANT, True),
    (5, False, 3, 2, capi.CONSTANT, True),
])
def test_argon_mixtures_gll(levels, ambi, dim, order, transport, two_t):
    """the Gauss-Lobatto pair for every species count (4 ... 8 species; SURVEY 8a row a16 / the reference accepts any
    basis / rule pair with any mixture, src/M2ulPhyS.cpp:557-572)"""
    ph = capi.argon_levels_physics(levels, ambi, capi.NS, transport, two_t, True)
    if dim == 3:
        c = cases.argon_cyl3d(3, 8, 3, order, physics=ph)
        mesh, bcs = c.mesh, c.bcs
    else:
        mesh = meshgen.scramble_orientations(meshgen.box_quad(5, 4, lengths=(1.0, 0.7), warp=0.08), 4)
        bcs = []
    disc = capi.Disc(order, 1, 1, 0, 0)
    U = cases.plasma_state(node_coordinates(mesh, order, 1), ph, nvel=dim, seed=6, amp=0.005 if order == 1 else 0.01)
    _compare(mesh, disc, ph, bcs, U, tol=1e-10)


def test_mixed_pairs_are_refused():
    with pytest.raises(Exception) as ei:
        hip_mult(meshgen.box_quad(3, 3), capi.Disc(2, 0, 1, 0, 0), capi.dry_air_physics(capi.NS), [],
                 cases.dry_air_state(node_coordinates(meshgen.box_quad(3, 3), 2), seed=1))
    assert "UNSUPPORTED" in str(ei.value)
