"""Inlets with the velocity given relative to the inlet face (``subsonicFaceBasedX/Y/Z`` = ``SUB_DENS_VEL_FACE_X/Y/Z``,
``src/inletBC.cpp:453-464, 758-864``): the oracle against the closed form of a face whose frame is
known, the HIP path against the oracle on the cylinder (the inlet patch is the upstream half of the outer ring: every
face has another normal)."""
import numpy as np
import pytest

from tps_amd import capi, cases, meshgen


def test_oracle_face_inlet_closed_form():
    from oracle_lib import Oracle

    mesh = meshgen.box_hex(3, 3, 3, periodic=(False, True, True), bdr_attr={(0, 0): 1, (0, 1): 2})
    ph = capi.dry_air_physics(capi.NS)
    rho_in, Un, Ut = 1.3, 25.0, 7.0
    bcs = [capi.make_bc(1, capi.INLET, capi.SUB_DENS_VEL_FACE_Z, [rho_in, Un, Ut, 0.0]), capi.make_bc(2, capi.OUTLET, capi.SUB_P, [101300.0])]
    o = Oracle(mesh, capi.Disc(1, 0, 0, 0, 0), ph, bcs)
    gam = ph.dry_air.specific_heat_ratio
    U = np.array([1.2, 1.2 * 20.0, 1.2 * 3.0, -1.2 * 2.0, 101300.0 / (gam - 1) + 0.6 * (400 + 9 + 4)])
    nor = np.array([-2.0, 0.0, 0.0])  # outward normal of the x = 0 face, not normalised
    # frame: inward unit normal (1, 0, 0) is already orthogonal to z; tangent1 = (n1 t2 - n2 t1, -(n0 t2 - n2 t0), ...) = (0, -1, 0)
    # M = rows (1,0,0), (0,-1,0), (0,0,1): momentum of the prescribed state = rho (Un, -Ut, 0)
    mom = rho_in * np.array([Un, -Ut, 0.0])
    ghost = np.zeros(5)
    ghost[0] = rho_in
    ghost[1:4] = 2.0 * mom - U[1:4]
    p_in = (gam - 1) * (U[4] - 0.5 * (U[1:4] ** 2).sum() / U[0])
    ghost[4] = p_in / (gam - 1) + 0.5 * (ghost[1:4] ** 2).sum() / ghost[0]
    want = o.lf(U, ghost, nor)
    got = o.bdr_flux(1, nor, U, np.zeros(15))
    assert np.abs(got - want).max() < 1e-12 * np.abs(want).max()
    # a normal with a z component: the frame normal loses it (and its unit length), the z momentum comes from neither Un nor Ut
    nor2 = np.array([-1.0, 0.0, 0.5])
    got2 = o.bdr_flux(1, nor2, U, np.zeros(15))
    un = -nor2 / np.linalg.norm(nor2)
    un[2] = 0.0
    t2 = np.array([0.0, 0.0, 1.0])
    t1 = np.array([un[1] * t2[2] - un[2] * t2[1], -(un[0] * t2[2] - un[2] * t2[0]), un[0] * t2[1] - un[1] * t2[0]])
    mom2 = np.linalg.solve(np.array([un, t1, t2]), rho_in * np.array([Un, Ut, 0.0]))
    ghost2 = ghost.copy()
    ghost2[1:4] = 2.0 * mom2 - U[1:4]
    ghost2[4] = p_in / (gam - 1) + 0.5 * (ghost2[1:4] ** 2).sum() / ghost2[0]
    want2 = o.lf(U, ghost2, nor2)
    assert np.abs(got2 - want2).max() < 1e-12 * np.abs(want2).max()


@pytest.mark.gpu
@pytest.mark.parametrize("fluid,axis,order", [("dry", capi.SUB_DENS_VEL_FACE_Z, 3), ("dry", capi.SUB_DENS_VEL_FACE_Y, 2),
                                              ("argon", capi.SUB_DENS_VEL_FACE_Z, 2), ("argon6", capi.SUB_DENS_VEL_FACE_X, 1)])
def test_face_inlet_hip_vs_oracle(fluid, axis, order):
    from parity_util import RHS_RTOL, hip_mult, oracle_mult, rel_maxnorm

    if fluid == "dry":
        c = cases.cyl3d(4, 12, 3, order, capi.NS, capi.VISC_ISOTH)
        c.physics.dry_air.visc_mult = 1000.0
        c.bcs[0] = capi.make_bc(1, capi.INLET, axis, [1.2, 18.0, 4.0, 0.0])
        U, tol = c.state(seed=12), RHS_RTOL
    else:
        ph = (capi.argon_ternary_physics(capi.NS, True, capi.CONSTANT, "arrhenius") if fluid == "argon"
              else capi.argon_six_species_physics(capi.NS, capi.ARGON_MIXTURE, True, True))
        c = cases.argon_cyl3d(4, 12, 3, order, physics=ph)
        data = list(cases.argon_inlet_state(ph, 3))
        data[1], data[2], data[3] = 18.0, 4.0, 0.0
        c.bcs[0] = capi.make_bc(1, capi.INLET, axis, data)
        amp = 0.005 if order == 1 else 0.01
        U, tol = c.state(seed=12, amp=amp), RHS_RTOL * 0.05 / amp
    ref = oracle_mult(c.mesh, c.disc, c.physics, c.bcs, U)
    got = hip_mult(c.mesh, c.disc, c.physics, c.bcs, U)
    # the condition is active: a different residual than with the Cartesian inlet of the same numbers
    c.bcs[0].type = capi.SUB_DENS_VEL
    plain = oracle_mult(c.mesh, c.disc, c.physics, c.bcs, U)
    assert rel_maxnorm(ref["y"], plain["y"]).max() > 1e-6
    err = rel_maxnorm(got["y"], ref["y"])
    print("face inlet", fluid, axis, order, err)
    assert err.max() < tol


@pytest.mark.gpu
def test_face_inlet_refused_in_two_dimensions():
    from tps_amd.rhs_operator import RHSoperator

    mesh = meshgen.box_quad(4, 3, periodic=(False, True), bdr_attr={(0, 0): 1, (0, 1): 2})
    bcs = [capi.make_bc(1, capi.INLET, capi.SUB_DENS_VEL_FACE_X, [1.2, 10.0, 0.0, 0.0]), capi.make_bc(2, capi.OUTLET, capi.SUB_P, [101300.0])]
    with pytest.raises(Exception, match="3-D"):
        RHSoperator(mesh, capi.Disc(2, 0, 0, 0, 0), capi.dry_air_physics(capi.NS), bcs)
