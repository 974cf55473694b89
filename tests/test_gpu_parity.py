"""HIP kernels vs the CPU oracle on identical seeded inputs (through the C ABI)."""
import numpy as np
import pytest

from parity_util import RHS_RTOL, hip_mult, oracle_mult, rel_maxnorm
from tps_amd import capi, cases, meshgen
from tps_amd.rhs_operator import node_coordinates

pytestmark = pytest.mark.gpu


def _compare(mesh, disc, ph, bcs, U, tol=RHS_RTOL):
    ref = oracle_mult(mesh, disc, ph, bcs, U)
    got = hip_mult(mesh, disc, ph, bcs, U)
    e_up = rel_maxnorm(got["Up"], ref["Up"])
    e_g = rel_maxnorm(got["gradUp"].reshape(-1, U.shape[1]), ref["gradUp"].reshape(-1, U.shape[1]))
    e_y = rel_maxnorm(got["y"], ref["y"])
    print("rel err Up", e_up.max(), "gradUp", e_g.max(), "y", e_y)
    assert e_up.max() < 1e-13
    # gradient rows that are identically ~0 (e.g. a uniform field) are compared absolutely
    scale = np.abs(ref["gradUp"]).max()
    assert np.abs(got["gradUp"] - ref["gradUp"]).max() < tol * scale
    assert e_y.max() < tol
    assert abs(got["max_char_speed"] - ref["max_char_speed"]) < 1e-12 * ref["max_char_speed"]


@pytest.mark.parametrize("order", [1, 2, 3])
def test_periodic_box_hex(order):
    mesh = meshgen.scramble_orientations(meshgen.box_hex(4, 3, 5, lengths=(1.0, 0.8, 1.2), warp=0.12), 11 + order)
    disc = capi.Disc(order, 0, 0, 0, 0)
    ph = capi.dry_air_physics(capi.NS, visc_mult=500.0, bulk_visc_mult=2.0)
    U = cases.dry_air_state(node_coordinates(mesh, order), seed=3 + order)
    _compare(mesh, disc, ph, [], U)


@pytest.mark.parametrize("order,eq,wall", [(1, capi.EULER, capi.INV), (2, capi.NS, capi.VISC_ADIAB),
                                            (3, capi.NS, capi.VISC_ISOTH), (3, capi.NS, capi.INV)])
def test_cylinder(order, eq, wall):
    c = cases.cyl3d(5, 12, 4, order, eq, wall)
    c.mesh = meshgen.scramble_orientations(c.mesh, 5)
    c.physics.dry_air.visc_mult = 2000.0
    U = c.state(seed=77)
    _compare(c.mesh, c.disc, c.physics, c.bcs, U)


@pytest.mark.parametrize("order", [1, 2, 3, 4])
def test_periodic_box_quad(order):
    mesh = meshgen.scramble_orientations(meshgen.box_quad(7, 5, lengths=(1.0, 0.7), warp=0.1), 2)
    disc = capi.Disc(order, 0, 0, 0, 0)
    ph = capi.dry_air_physics(capi.NS, visc_mult=300.0)
    U = cases.dry_air_state(node_coordinates(mesh, order), seed=9)
    _compare(mesh, disc, ph, [], U)


def test_use_bc_in_grad():
    c = cases.cyl3d(4, 12, 3, 2, capi.NS, capi.VISC_ISOTH)
    c.disc.use_bc_in_grad = 1
    c.physics.dry_air.visc_mult = 2000.0
    _compare(c.mesh, c.disc, c.physics, c.bcs, c.state(seed=5))
