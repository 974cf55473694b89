"""The table gas, ``fluid = lte_table`` with one-dimensional tables (``LteMixture`` / ``LteTransport``,
``src/lte_mixture.cpp``, ``src/lte_transport_properties.cpp``), and the viscous sponge of the 2-D
kernels with the heavy interface.

* the oracle against the numbers of the reference's own unit test, ``test/test_lte_mixture.cpp:20-150`` (on the tables
  that test reads, ``tests/golden/tables/lte_tables.npz``);
* the HIP path against the oracle: point closures on the device, ``Mult`` on axisymmetric tubes with every wall type, the
  radiation sink, the mixing-length model and the viscous sponge together (the combination of the reference's
  ``test/inputs/plasma.lte1d.ini``), a few RK4 steps;
* the viscous sponge for the other 2-D heavy families: axisymmetric dry air, the ternary mixture planar and axisymmetric
  (``test/inputs/argon.plasma.lte2noneq.ini:104``), the six-species torch mixture.
"""
import ctypes as C

import numpy as np
import pytest

from parity_util import RHS_RTOL, hip_mult, oracle_mult, rel_maxnorm
from tps_amd import capi, cases, meshgen
from tps_amd.rhs_operator import node_coordinates


def _oracle(density="rho0p005", radiation=False):
    from oracle_lib import Oracle

    mesh = meshgen.box_hex(3, 3, 3)
    ph = capi.lte_physics(capi.NS, density, radiation)
    return Oracle(mesh, capi.Disc(1, 0, 0, 0, 0), ph, []), ph


# ---- test/test_lte_mixture.cpp -----------------------------------------------------------------------------------------
def test_oracle_pressure_from_primitives():
    """checkPressureFromPrimitives (:20-45): p = rho R(T) T with R = 208.1321372 at (0.005, 2000 K) and 217.82066155 at
    (0.255, 12000 K)"""
    for density, rho, T, R in (("rho0p005", 0.005, 2000.0, 208.1321372), ("rho0p255", 0.255, 12000.0, 217.82066155)):
        o, _ = _oracle(density)
        U = o.cons(np.array([rho, 0.0, 0.0, 0.0, T]))
        assert o.pressure(U) == pytest.approx(rho * R * T, rel=1e-12)


def test_oracle_temperature_round_trips():
    """checkTemperature (:47-118): T(U) with U built from evaluateInternalEnergy(T, rho), without and with kinetic energy"""
    for density, rho, T in (("rho0p005", 0.005, 2000.0), ("rho0p255", 0.255, 12000.0)):
        o, _ = _oracle(density)
        U = o.cons(np.array([rho, 0.0, 0.0, 0.0, T]))
        assert o.prim(U)[4] == pytest.approx(T, rel=1e-13)
    o, _ = _oracle("rho0p255")
    rho, T = 0.255, 12000.0
    e = o.cons(np.array([rho, 0.0, 0.0, 0.0, T]))[4] / rho
    mom = np.array([0.5, 10.0, -0.5])
    U = np.array([rho, *mom, rho * e + 0.5 * (mom ** 2).sum() / rho])
    assert o.prim(U)[4] == pytest.approx(T, rel=1e-13)
    # temperatures between the table's nodes (a Newton step or two on the piecewise-linear e(T))
    for T in (333.0, 4567.8, 15999.0):
        assert o.prim(o.cons(np.array([rho, 3.0, -2.0, 1.0, T])))[4] == pytest.approx(T, rel=1e-12)


def test_oracle_viscosity_known_answer():
    """checkTransport (:120-150): mu(350 K) = 2.0656339881365003e-05 on test/inputs/air_simple_transport_table.dat"""
    o, _ = _oracle()
    U = o.cons(np.array([1.225, 0.0, 0.0, 0.0, 350.0]))
    buf, _ = o.flux_transport(U, np.zeros(15))
    assert buf[0] == pytest.approx(2.0656339881365003e-05, rel=1e-12)
    assert buf[1] == 0.0 and buf[3] == 0.0  # no bulk viscosity, no separate electron conductivity


def test_oracle_temperature_from_density_pressure():
    """checkTemperature, second half (:96-116): the pressure outlet's modifyEnergyForPressure inverts p = rho R(T) T"""
    from oracle_lib import Oracle

    mesh = meshgen.box_hex(3, 3, 3, periodic=(False, True, True), bdr_attr={(0, 0): 1, (0, 1): 2})
    ph = capi.lte_physics(capi.EULER, "rho0p255")
    rho, T = 0.255, 12000.0
    o0, _ = _oracle("rho0p255")
    Uin = o0.cons(np.array([rho, 4.0, 1.0, 0.0, 9000.0]))
    p_out = o0.pressure(o0.cons(np.array([rho, 0.0, 0.0, 0.0, T])))
    bcs = [capi.make_bc(1, capi.WALL, capi.INV), capi.make_bc(2, capi.OUTLET, capi.SUB_P, [p_out])]
    o = Oracle(mesh, capi.Disc(1, 0, 0, 0, 0), ph, bcs)
    nor = np.array([1.0, 0.0, 0.0])
    ghost = Uin.copy()  # interior density and momentum, the energy of the prescribed pressure
    ghost[4] = o0.cons(np.array([rho, 0.0, 0.0, 0.0, T]))[4] + 0.5 * (Uin[1:4] ** 2).sum() / rho
    want = o.lf(Uin, ghost, nor)
    got = o.bdr_flux(2, nor, Uin, np.zeros(15))
    assert np.abs(got - want).max() < 1e-11 * np.abs(want).max()


# ---- HIP vs oracle ---------------------------------------------------------------------------------------------------
def _compare(c, U, tol=RHS_RTOL, distance=None, ml=None):
    import torch
    from oracle_lib import Oracle
    from tps_amd.rhs_operator import RHSoperator

    o = Oracle(c.mesh, c.disc, c.physics, c.bcs)
    op = RHSoperator(c.mesh, c.disc, c.physics, c.bcs)
    if distance is not None:
        o.set_mixing_length(distance, **ml)
        d_dev = torch.tensor(distance, dtype=torch.float64, device=op.device)
        op.setMixingLength(d_dev, **ml)
    ref = o.mult(U)
    x = torch.tensor(np.ascontiguousarray(U).ravel(), dtype=torch.float64, device=op.device)
    y = torch.empty_like(x)
    op.Mult(x, y, want_max_char_speed=True)
    torch.cuda.synchronize()
    got = y.cpu().numpy().reshape(U.shape)
    e_up = rel_maxnorm(op.getPrimitives().cpu().numpy(), o.primitives())
    g_ref = o.gradients()
    e_g = np.abs(op.getGradients().cpu().numpy() - g_ref).max() / np.abs(g_ref).max()
    e_y = rel_maxnorm(got, ref)
    print("rel err Up", e_up.max(), "gradUp", e_g, "y", e_y)
    assert e_up.max() < 1e-13
    assert e_g < tol
    assert e_y.max() < tol
    assert abs(op.max_char_speed - o.max_char_speed) < 1e-12 * o.max_char_speed
    op.close()
    return ref, got


def _wall_distance(c):
    X = node_coordinates(c.mesh, c.disc.order)
    return np.ascontiguousarray(X[0].max() - X[0])


@pytest.mark.gpu
def test_hip_point_closures():
    """the device's Newton inversions and table look-ups against the oracle's, and against the reference's numbers"""
    import torch
    from tps_amd.rhs_operator import RHSoperator

    c = cases.lte_axisym(3, 3, 1, density="rho0p255")
    o, _ = _oracle("rho0p255")
    op = RHSoperator(c.mesh, c.disc, c.physics, c.bcs)
    lib = capi.load()
    rng = np.random.default_rng(7)
    n = 257
    prim = np.stack([0.255 * (1 + 0.3 * rng.uniform(-1, 1, n)), rng.uniform(-30, 30, n), rng.uniform(-30, 30, n),
                     rng.uniform(-30, 30, n), rng.uniform(150.0, 19000.0, n)])
    prim[:, 0] = [0.255, 0.0, 0.0, 0.0, 12000.0]  # the reference's spot check
    states = np.stack([o.cons(np.ascontiguousarray(prim[:, i])) for i in range(n)], axis=1)
    xd = torch.tensor(np.ascontiguousarray(states), device="cuda")
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    for quantity, ref_fn in ((1, o.pressure), (3, o.max_char_speed_point)):
        st = lib.tpsrhs_eval_pointwise(op._h, quantity, n, C.c_void_p(xd.data_ptr()), C.c_void_p(out.data_ptr()))
        assert st == 0, lib.tpsrhs_last_error().decode()
        ref = np.array([ref_fn(np.ascontiguousarray(states[:, i])) for i in range(n)])
        assert out.cpu().numpy() == pytest.approx(ref, rel=1e-13)
        if quantity == 1:
            assert out.cpu().numpy()[0] == pytest.approx(0.255 * 217.82066155 * 12000.0, rel=1e-12)
    pr = torch.empty(5 * n, dtype=torch.float64, device="cuda")
    assert lib.tpsrhs_eval_pointwise(op._h, 0, n, C.c_void_p(xd.data_ptr()), C.c_void_p(pr.data_ptr())) == 0
    assert pr.cpu().numpy().reshape(5, n) == pytest.approx(prim, rel=1e-12)
    op.close()


@pytest.mark.gpu
@pytest.mark.parametrize("order,eq,wall,r_in,warp,bcgrad", [
    (3, capi.NS, capi.VISC_ISOTH, 0.0, 0.0, 0), (2, capi.NS, capi.VISC_ADIAB, 0.01, 0.06, 0),
    (1, capi.EULER, capi.INV, 0.02, 0.06, 0), (4, capi.NS, capi.INV, 0.0, 0.0, 0),
    (3, capi.NS, capi.VISC_ISOTH, 0.0, 0.05, 1),  # useBCinGrad = true, as test/inputs/plasma.lte1d.ini:117
])
def test_lte_axisymmetric(order, eq, wall, r_in, warp, bcgrad):
    c = cases.lte_axisym(6, 9, order, eq, wall, r_in=r_in, warp=warp, radiation=(order == 3))
    c.disc.use_bc_in_grad = bcgrad
    _compare(c, c.state(seed=3 + order))


@pytest.mark.gpu
def test_lte_torch_combination():
    """test/inputs/plasma.lte1d.ini: table gas, axisymmetric, mixing-length model, viscosity multiplier function, isothermal
    and inviscid walls, useBCinGrad"""
    c = cases.lte_axisym(6, 9, 3, capi.NS, capi.VISC_ISOTH, radiation=True)
    c.disc.use_bc_in_grad = 1
    vs = c.physics.visc_sponge
    vs.enabled, vs.width, vs.ratio = 1, 0.02, 20.0  # :52-58: normal '0 1 0', width 0.02, viscosityRatio 20
    vs.normal[1], vs.point[1] = 1.0, 0.15
    U = c.state(seed=21)
    ml = dict(max_mixing_length=0.01, pr_ratio=0.0, bulk_multiplier=0.0)  # :41-44
    ref, _ = _compare(c, U, distance=_wall_distance(c), ml=ml)
    c.physics.visc_sponge.enabled = 0
    plain = oracle_mult(c.mesh, c.disc, c.physics, c.bcs, U)["y"]
    change = np.abs(ref - plain).max(axis=1) / np.abs(ref).max(axis=1)
    print("change by the mixing-length model and the sponge", change)
    assert change[1:].max() > 1e-4


@pytest.mark.gpu
def test_lte_rk4_steps():
    import torch
    from oracle_lib import Oracle
    from tps_amd.rhs_operator import RHSoperator

    c = cases.lte_axisym(5, 7, 2, capi.NS, capi.VISC_ISOTH)
    U = c.state(seed=9)
    o = Oracle(c.mesh, c.disc, c.physics, c.bcs)
    op = RHSoperator(c.mesh, c.disc, c.physics, c.bcs)
    x = torch.tensor(np.ascontiguousarray(U).ravel(), dtype=torch.float64, device=op.device)
    ref = np.ascontiguousarray(U).copy()
    dt, t = 2.0e-8, 0.0
    for _ in range(3):
        ref, t, _, _ = o.rk4_step(ref, t, dt)
    op.advance(x, 0.0, dt, 3, constant_dt=True)
    torch.cuda.synchronize()
    err = rel_maxnorm(x.cpu().numpy().reshape(U.shape), ref.reshape(U.shape))
    print("rel err after 3 RK4 steps", err)
    assert err.max() < 1e-13
    op.close()


# ---- the viscous sponge of the other 2-D heavy families -----------------------------------------------------------------
def _sponge(ph, axis, point, width, ratio):
    vs = ph.visc_sponge
    vs.enabled, vs.width, vs.ratio = 1, width, ratio
    vs.normal[axis], vs.normal[1 - axis] = 2.0, 0.3  # not a unit normal: normalised by the library and the oracle
    vs.point[axis], vs.point[1 - axis] = point, 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("fluid,order", [("dry_axi", 3), ("ternary_axi", 2), ("ternary_planar", 3), ("ternary_axi_2t_wall", 3),
                                         ("torch6", 2)])
def test_viscous_sponge_2d_heavy(fluid, order):
    if fluid == "dry_axi":
        c = cases.dry_air_axisym(6, 9, order, capi.NS, capi.VISC_ISOTH)
        c.physics.dry_air.visc_mult = 200.0
        c.physics.dry_air.bulk_visc_mult = 1.5
        U, tol = c.state(seed=4), RHS_RTOL
    elif fluid == "ternary_planar":
        # not periodic: each side of a face weighs its viscous trace at its own position (see tests/test_gpu_les.py)
        attrs = {(0, 0): 1, (0, 1): 2, (1, 0): 3, (1, 1): 3}
        mesh = meshgen.box_quad(6, 5, lengths=(0.05, 0.25), periodic=(False, False), bdr_attr=attrs, warp=0.08)
        ph = capi.argon_ternary_physics(capi.NS, False, capi.ARGON_MINIMAL, "arrhenius")
        bcs = [capi.make_bc(1, capi.INLET, capi.SUB_DENS_VEL, cases.argon_inlet_state(ph, 2)),
               capi.make_bc(2, capi.OUTLET, capi.SUB_P, [101300.0]), capi.make_bc(3, capi.WALL, capi.VISC_ISOTH, [3000.0])]
        c = cases.Case("sponge_planar", mesh, capi.Disc(order, 0, 0, 0, 0), ph, bcs)
        U, tol = cases.plasma_state(node_coordinates(mesh, order), ph, nvel=2, seed=3, amp=0.01), 5 * RHS_RTOL
    else:
        if fluid == "torch6":
            ph = capi.argon_six_species_physics(capi.NS, capi.ARGON_MIXTURE, True, True, radiation=True)
        else:
            ph = capi.argon_ternary_physics(capi.NS, fluid != "ternary_axi", capi.ARGON_MINIMAL, "arrhenius")
        c = cases.argon_axisym(6, 9, order, physics=ph, r_in=0.0)
        if fluid == "ternary_axi_2t_wall":  # prescribed wall fluxes: the sponge scales the velocities BEFORE the prescription
            c.bcs[2] = capi.make_bc(3, capi.WALL, capi.VISC_GNRL, [3000.0, 0.0, capi.ISOTH, capi.SHTH])
        U, tol = c.state(seed=3, amp=0.01), 5 * RHS_RTOL
    # stronger molecular transport, so that the sponge's share of the residual is far above the tolerance
    # (the helper leaves the ill-conditioned third-order electron conductivity at its physical size, see its comment)
    if c.physics.working_fluid == capi.USER_DEFINED:
        from test_gpu_parity import _boost_transport

        _boost_transport(c.physics, 30.0)
    plain = oracle_mult(c.mesh, c.disc, c.physics, c.bcs, U)["y"]
    _sponge(c.physics, 1, 0.12, 0.04, 20.0)
    ref, _ = _compare(c, U, tol=tol)
    change = np.abs(ref - plain).max(axis=1) / np.abs(ref).max(axis=1)
    print("relative change by the sponge", change)
    assert change[1:].max() > 1e-4


@pytest.mark.gpu
def test_lte_unsupported():
    from tps_amd.rhs_operator import RHSoperator

    ph = capi.lte_physics()
    with pytest.raises(Exception, match="axisymmetric"):
        RHSoperator(meshgen.box_hex(3, 3, 3), capi.Disc(2, 0, 0, 0, 0), ph, [])
    c = cases.lte_axisym(3, 3, 2)
    c.physics.sgs.model_type = capi.SGS_SMAGORINSKY
    with pytest.raises(Exception, match="3-D"):
        RHSoperator(c.mesh, c.disc, c.physics, c.bcs)
    pl = capi.argon_ternary_physics()
    pl.visc_sponge.enabled, pl.visc_sponge.width = 1, 1.0
    pl.visc_sponge.normal[0] = 1.0
    with pytest.raises(Exception, match="planar 2-D and the axisymmetric"):
        RHSoperator(meshgen.box_hex(3, 3, 3), capi.Disc(2, 0, 0, 0, 0), pl, [])


def test_oracle_sponge_scales_the_right_coefficients_of_a_mixture():
    """Fluxes::ComputeViscousFluxes with the viscosity-multiplier function on (src/fluxes.cpp:232-246): mu, mu_b - 2/3 mu, k_h
    and the diffusion velocities of the ACTIVE species carry the weight w; k_e and the other species' velocities do not.
    Closed-form consequences for one point of a two-temperature ternary mixture (constant transport, ambipolar: one active
    species): the momentum rows and the active species row scale with w exactly; the electron-energy row's conduction part
    does not."""
    from oracle_lib import Oracle

    mesh = meshgen.box_quad(3, 3, lengths=(1.0, 1.0))
    ph0 = capi.argon_ternary_physics(capi.NS, True, capi.CONSTANT, None)
    ph1 = capi.argon_ternary_physics(capi.NS, True, capi.CONSTANT, None)
    vs = ph1.visc_sponge
    vs.enabled, vs.width, vs.ratio = 1, 0.2, 11.0
    vs.normal[0], vs.normal[1], vs.point[0], vs.point[1] = 0.0, 3.0, 0.0, 0.4  # normalised to (0, 1)
    o0, o1 = Oracle(mesh, capi.Disc(1, 0, 0, 0, 0), ph0, []), Oracle(mesh, capi.Disc(1, 0, 0, 0, 0), ph1, [])
    X = np.array([[0.3], [0.55]])
    U = cases.plasma_state(X, ph0, nvel=2, seed=3, amp=0.0)[:, 0]
    neq = U.size  # rho, rho u, rho v, rho E, rho Y_ion, rho e_e
    rng = np.random.default_rng(5)
    g = rng.uniform(-1.0, 1.0, neq * 2) * np.tile(np.maximum(np.abs(o0.prim(U)), 1.0), 2) * 5.0
    x = (0.3, 0.55)
    w = 1.0 + 10.0 * 0.5 * (np.tanh((0.55 - 0.4) / 0.2 - 2.0) + 1.0)
    a, b = o1.viscous_flux_at(U, g, x, 0.1), o0.viscous_flux_at(U, g, x, 0.1)
    for d in range(2):
        rows = [1 + d * neq, 2 + d * neq, 4 + d * neq]  # momentum, active species
        assert np.abs(a[rows] - w * b[rows]).max() < 1e-12 * np.abs(a[rows]).max()
    # electron energy row: k_e grad T_e - h_e V_e with k_e and (ambipolar) V_e unweighted ... through the ambipolar field V_e
    # depends on the weighted ion velocity, so only the pure-conduction case is a closed form: zero species gradients
    g2 = np.zeros_like(g)
    g2[neq - 1], g2[neq - 1 + neq] = 700.0, -300.0  # grad T_e only
    a2, b2 = o1.viscous_flux_at(U, g2, x, 0.1), o0.viscous_flux_at(U, g2, x, 0.1)
    assert np.abs(a2[[neq - 1, 2 * neq - 1]] - b2[[neq - 1, 2 * neq - 1]]).max() < 1e-12 * np.abs(b2).max()
    assert np.abs(b2[[neq - 1, 2 * neq - 1]]).min() > 0.0


def _sigma_table(ph):
    """(the sigma column of the reference's air_simple_transport_table.dat is zero: an argon-like ramp instead, SYNTHETIC,
    zero below 5000 K so that the floor of 1 S/m is exercised)"""
    Ts = np.linspace(300.0, 20000.0, 80)
    sig = 0.35 * np.maximum(Ts - 5000.0, 0.0) ** 1.1
    ph.lte.electric_conductivity_table = capi.make_table(Ts, sig, keep=ph._keep)
    return Ts, sig


def test_oracle_plasma_conductivity_closed_form():
    """SourceTerm's side output for the table gas (src/source_term.cpp:196, src/lte_transport_properties.cpp:109-126):
    sigma(T) of the table, not below 1 S/m -- the transport table of the reference's unit test has sigma = 0 at low T"""
    from oracle_lib import Oracle

    c = cases.lte_axisym(4, 5, 2)
    Ts, sig = _sigma_table(c.physics)
    U = c.state(seed=2)
    o = Oracle(c.mesh, c.disc, c.physics, c.bcs)
    o.mult(U)
    T = o.primitives()[4]
    want = np.maximum(np.interp(T, Ts, sig), 1.0)
    got = o.plasma_conductivity(U)
    assert got == pytest.approx(want, rel=1e-13)
    assert (want > 1.0).any() and (want == 1.0).any()


@pytest.mark.gpu
@pytest.mark.parametrize("fluid", ["lte", "ternary_minimal", "ternary_constant_2t", "six_ambipolar_mixture", "ternary_no_reactions"])
def test_hip_plasma_conductivity(fluid):
    """SourceTerm's plasma_conductivity_ (src/source_term.cpp:125-199): the table gas, and the mixtures the reference stores it
    for -- ambipolar ones (test/inputs/argon.plasma.lte2noneq.ini: six species, ambipolar, constant transport) and
    mixtures without reactions"""
    import torch
    from oracle_lib import Oracle
    from tps_amd.rhs_operator import RHSoperator

    if fluid == "lte":
        c = cases.lte_axisym(5, 6, 3)
        _sigma_table(c.physics)
        U = c.state(seed=8)
    else:
        if fluid == "ternary_minimal":
            ph = capi.argon_ternary_physics(capi.NS, False, capi.ARGON_MINIMAL, "arrhenius")
        elif fluid == "ternary_constant_2t":
            ph = capi.argon_ternary_physics(capi.NS, True, capi.CONSTANT, "arrhenius")
        elif fluid == "six_ambipolar_mixture":
            ph = capi.argon_levels_physics(3, True, capi.NS, capi.ARGON_MIXTURE, False, True)
        else:
            ph = capi.argon_ternary_physics(capi.NS, False, capi.ARGON_MINIMAL, None, ambipolar=False)
        ph.gas_transport.multiply = 1  # the mobility multiplier enters (src/gas_transport.cpp:725-736)
        ph.gas_transport.mobil_mult, ph.gas_transport.diff_mult = 1.7, 0.6
        c = cases.argon_axisym(5, 6, 2, physics=ph, r_in=0.0)
        U = c.state(seed=8, amp=0.02)
    o = Oracle(c.mesh, c.disc, c.physics, c.bcs)
    want = o.plasma_conductivity(U)
    op = RHSoperator(c.mesh, c.disc, c.physics, c.bcs)
    x = torch.tensor(np.ascontiguousarray(U).ravel(), dtype=torch.float64, device=op.device)
    got = op.getPlasmaConductivity(x).cpu().numpy()
    print(fluid, "sigma", want.min(), want.max(), "rel err", np.abs(got - want).max() / np.abs(want).max())
    assert want.max() > 0.0
    # (collision integrals through the device's exp / log, 2-4 ulp each, and a Curtiss-Hirschfelder sum: 3.6e-12 ... 4.9e-12
    #  measured in round 3; the bound is four times that, not two orders of magnitude)
    assert got == pytest.approx(want, rel=2e-11)
    op.close()


@pytest.mark.gpu
def test_plasma_conductivity_refusals():
    import torch
    from tps_amd.rhs_operator import RHSoperator

    d = cases.dry_air_axisym(3, 3, 2)
    op = RHSoperator(d.mesh, d.disc, d.physics, d.bcs)
    x = torch.tensor(np.ascontiguousarray(d.state()).ravel(), dtype=torch.float64, device=op.device)
    with pytest.raises(Exception, match="no SourceTerm"):
        op.getPlasmaConductivity(x)
    op.close()
    ph = capi.argon_ternary_physics(capi.NS, False, capi.ARGON_MINIMAL, "arrhenius", ambipolar=False)
    c = cases.argon_axisym(3, 3, 2, physics=ph, r_in=0.0)
    op = RHSoperator(c.mesh, c.disc, c.physics, c.bcs)
    x = torch.tensor(np.ascontiguousarray(c.state(amp=0.01)).ravel(), dtype=torch.float64, device=op.device)
    with pytest.raises(Exception, match="not ambipolar"):
        op.getPlasmaConductivity(x)
    op.close()
