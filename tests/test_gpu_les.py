"""Sub-grid scale models and the planar viscous sponge of the reference's Fluxes (src/fluxes.cpp:223-246, 513-688;
SURVEY.md 8f rank 4) on the HIP path vs the CPU oracle, through the C ABI."""
import numpy as np
import pytest

from parity_util import RHS_RTOL, hip_mult, oracle_mult, rel_maxnorm
from tps_amd import capi, cases, meshgen

pytestmark = pytest.mark.gpu


def _sheared(case, seed, amp=0.1):
    """10 % waves of every primitive: on these coarse meshes the eddy viscosity (rho (C delta)^2 |S|) then changes
    the residual by 1e-4 .. 1e-2 (larger amplitudes extrapolate to negative pressures at the faces)"""
    return case.state(seed=seed, amp=amp)


def _compare(c, U, tol=RHS_RTOL):
    ref = oracle_mult(c.mesh, c.disc, c.physics, c.bcs, U)
    got = hip_mult(c.mesh, c.disc, c.physics, c.bcs, U)
    e = rel_maxnorm(got["y"], ref["y"])
    print("rel err y", e)
    assert e.max() < tol
    return ref, got


def _plain(c, U):
    ph = capi.dry_air_physics(c.physics.eq_system, c.physics.dry_air.visc_mult, c.physics.dry_air.bulk_visc_mult)
    return oracle_mult(c.mesh, c.disc, ph, c.bcs, U)["y"]


@pytest.mark.parametrize("model,order", [(capi.SGS_SMAGORINSKY, 1), (capi.SGS_SMAGORINSKY, 3), (capi.SGS_SIGMA, 2),
                                         (capi.SGS_SIGMA, 3)])
def test_sgs_periodic_box(model, order):
    """the setting of test/inputs/input.sgsSmag.ini (dry air, periodic box, Gauss-Legendre pair), warped elements"""
    mesh = meshgen.box_hex(4, 3, 3, lengths=(1.0, 1.0, 0.4), warp=0.12)
    ph = capi.dry_air_physics(capi.NS, bulk_visc_mult=0.6)
    ph.sgs.model_type = model
    c = cases.Case("sgs_box", mesh, capi.Disc(order, 0, 0, 0, 0), ph, [])
    U = _sheared(c, 31, amp=0.1)
    ref, _ = _compare(c, U)
    # the model is active: the residual differs from the one without it
    d = np.abs(ref["y"] - _plain(c, U)).max(axis=1) / np.abs(ref["y"]).max(axis=1)
    print("relative change by the model", d)
    assert d[1:].max() > 1e-4


def test_sgs_floor_and_constant():
    mesh = meshgen.box_hex(3, 3, 3, lengths=(0.5, 0.5, 0.5), warp=0.1)
    ph = capi.dry_air_physics(capi.NS)
    ph.sgs.model_type, ph.sgs.model_const, ph.sgs.model_floor = capi.SGS_SMAGORINSKY, 0.2, 0.01
    c = cases.Case("sgs_floor", mesh, capi.Disc(2, 0, 0, 0, 0), ph, [])
    _compare(c, _sheared(c, 5, amp=0.1))


@pytest.mark.parametrize("wall", [capi.VISC_ISOTH, capi.VISC_ADIAB, capi.INV])
def test_sgs_cylinder_walls(wall):
    """the boundary routines take the same eddy viscosity (ComputeBdrViscousFluxes, src/fluxes.cpp:387-396)"""
    c = cases.cyl3d(4, 12, 3, 2, capi.NS, wall)
    c.physics.sgs.model_type = capi.SGS_SIGMA if wall == capi.VISC_ADIAB else capi.SGS_SMAGORINSKY
    c.physics.dry_air.visc_mult = 50.0
    _compare(c, c.state(seed=8, amp=0.1))


@pytest.mark.parametrize("dim,order", [(3, 3), (2, 3), (2, 2)])
def test_viscous_sponge(dim, order):
    """[viscosityMultiplierFunction]: tanh ramp of the viscosity along a plane normal; the normal given here has
    length 2.06 and is normalised by the library and the oracle like the reference's host constructor does"""
    if dim == 3:
        c = cases.cyl3d(4, 12, 3, order, capi.NS, capi.VISC_ISOTH)
    else:
        # not periodic: across a periodic face the two sides sit at different images of the point, and each side of
        # the HIP path weighs its own viscous trace at its own position (the reference: both at element 1's)
        attrs = {(0, 0): 1, (0, 1): 2, (1, 0): 3, (1, 1): 3}
        mesh = meshgen.box_quad(6, 5, lengths=(2.0, 1.0), periodic=(False, False), bdr_attr=attrs, warp=0.1)
        bcs = [capi.make_bc(1, capi.INLET, capi.SUB_DENS_VEL, [1.2, 20.0, 0.0, 0.0]),
               capi.make_bc(2, capi.OUTLET, capi.SUB_P, [101300.0]), capi.make_bc(3, capi.WALL, capi.VISC_ISOTH, [300.0])]
        c = cases.Case("sponge2d", mesh, capi.Disc(order, 0, 0, 0, 0), capi.dry_air_physics(capi.NS), bcs)
    c.physics.dry_air.visc_mult = 300.0
    vs = c.physics.visc_sponge
    vs.enabled, vs.width, vs.ratio = 1, 1.5 if dim == 3 else 0.3, 25.0
    for d, (n, p) in enumerate(zip((2.0, 0.5, 0.0), (1.0, -0.5, 0.0) if dim == 3 else (0.9, 0.2, 0.0))):
        vs.normal[d], vs.point[d] = n, p
    U = c.state(seed=3, amp=0.1)
    ref, _ = _compare(c, U)
    d = np.abs(ref["y"] - _plain(c, U)).max(axis=1) / np.abs(ref["y"]).max(axis=1)
    print("relative change by the sponge", d)
    assert d[1:].max() > 1e-4


def test_sgs_and_sponge_together_p1():
    mesh = meshgen.box_hex(5, 4, 4, lengths=(1.0, 1.0, 0.5), warp=0.1)
    ph = capi.dry_air_physics(capi.NS)
    ph.sgs.model_type = capi.SGS_SIGMA
    ph.visc_sponge.enabled, ph.visc_sponge.width, ph.visc_sponge.ratio = 1, 0.2, 8.0
    ph.visc_sponge.normal[0], ph.visc_sponge.point[0] = 1.0, 0.6
    # walls in x (the direction the sponge varies along), periodic in y and z
    mesh = meshgen.box_hex(5, 4, 4, lengths=(1.0, 1.0, 0.5), periodic=(False, True, True), warp=0.1,
                           bdr_attr={(d, s): 3 for d in range(3) for s in (0, 1)})
    c = cases.Case("les_p1", mesh, capi.Disc(1, 0, 0, 0, 0), ph, [capi.make_bc(3, capi.WALL, capi.VISC_ADIAB)])
    _compare(c, _sheared(c, 77, amp=0.1))


def test_caller_supplied_element_size():
    """tpsrhs_mesh::elem_size (the adapter passes mfem::Mesh::GetElementSize(e, 1)) replaces the library's own"""
    mesh = meshgen.box_hex(3, 3, 3, lengths=(1.0, 1.0, 0.4), warp=0.1)
    ph = capi.dry_air_physics(capi.NS)
    ph.sgs.model_type = capi.SGS_SMAGORINSKY
    c = cases.Case("elsize", mesh, capi.Disc(2, 0, 0, 0, 0), ph, [])
    U = _sheared(c, 2, amp=0.1)
    y0 = hip_mult(c.mesh, c.disc, c.physics, c.bcs, U, want_grad=False)["y"]
    mesh.elem_size = np.full(mesh.num_elements, 0.05)
    ref, got = _compare(c, U)
    assert rel_maxnorm(got["y"], y0).max() > 1e-6


def test_unsupported_combinations():
    from tps_amd.rhs_operator import RHSoperator
    ph = capi.dry_air_physics(capi.NS)
    ph.sgs.model_type = capi.SGS_SMAGORINSKY
    mesh = meshgen.box_quad(3, 3)
    with pytest.raises(Exception, match="dim == 3"):
        RHSoperator(mesh, capi.Disc(2, 0, 0, 0, 0), ph, [])
    with pytest.raises(Exception, match="Gauss-Legendre"):
        RHSoperator(meshgen.box_hex(3, 3, 3), capi.Disc(2, 1, 1, 0, 0), ph, [])
    pl = capi.argon_ternary_physics()
    pl.visc_sponge.enabled, pl.visc_sponge.width = 1, 1.0
    with pytest.raises(Exception, match="planar 2-D and the axisymmetric"):  # mixtures: the 2-D kernels (tests/test_lte.py)
        RHSoperator(meshgen.box_hex(3, 3, 3), capi.Disc(2, 0, 0, 0, 0), pl, [])
