"""CPU checks of the oracle's plasma closures (PerfectMixture, argon transport, chemistry, sources).

The reference's own regression data for these paths are git-LFS pointers, so the
restatement is pinned by (i) the two collision-integral known answers SURVEY.md records from the
reference's collision_integrals.cpp, (ii) identities the reference's own tests assert
(test/test_perfect_mixture.cpp style round trips, mole/mass fraction sums, zero net diffusive mass
flux, zero ambipolar current) and (iii) independent numpy restatements of closed-form formulas.
"""
import numpy as np
import pytest

from oracle_lib import Oracle, collision_integral
from tps_amd import capi, cases, meshgen

R = capi.UNIVERSALGASCONSTANT


def tiny_oracle(two_t=False, transport=capi.ARGON_MINIMAL, reactions="arrhenius", dim=3, radiation=False):
    mesh = (meshgen.box_hex(1, 1, 1, periodic=(False,) * 3) if dim == 3 else
            meshgen.box_quad(1, 1, periodic=(False,) * 2))
    ph = capi.argon_ternary_physics(capi.NS, two_t, transport, reactions, radiation=radiation)
    nattr = 6 if dim == 3 else 4
    bcs = [capi.make_bc(a + 1, capi.WALL, capi.INV) for a in range(nattr)]
    return Oracle(mesh, capi.Disc(1, 0, 0, 0, 0), ph, bcs), ph


def sample_states(ph, nvel, n=64, seed=3):
    X = np.random.default_rng(seed).uniform(0, 1, size=(nvel if nvel < 3 else 3, n))
    return cases.plasma_state(X, ph, nvel=nvel, seed=seed).T.copy()


def test_collision_integral_known_answers():
    # SURVEY.md section 7 ("What does compile"): values of the reference's collision_integrals.cpp
    assert collision_integral("att11", 5.0) == pytest.approx(0.038997464294197495, rel=1e-14)
    assert collision_integral("eAr11", 1.0e4) == pytest.approx(3.9634368412470003e-20, rel=1e-14)
    # closed forms (src/collision_integrals.cpp:124-135)
    assert collision_integral("ArAr22", 8000.0) == pytest.approx(1.7e-18 * 8000.0 ** -0.25, rel=1e-15)
    assert collision_integral("ArAr1P11", 8000.0) == pytest.approx(4.574321e-18 * 8000.0 ** -0.1805, rel=1e-15)
    # the charged fits decay monotonically in the nondimensional temperature
    for name in ("att11", "att12", "att13", "att14", "att15", "rep22", "rep23", "rep24"):
        v = [collision_integral(name, t) for t in (0.5, 2.0, 8.0, 32.0)]
        assert all(a > b > 0 for a, b in zip(v, v[1:])), name
    # Devoto ordering of the e-Ar integrals at 1 eV
    q = [collision_integral(f"eAr1{r}", 11604.0) for r in range(1, 6)]
    assert all(x > 0 for x in q)


@pytest.mark.parametrize("two_t", [False, True])
def test_prim_cons_round_trip_and_pressure(two_t):
    o, ph = tiny_oracle(two_t)
    for U in sample_states(ph, 3):
        Up = o.prim(U)
        U2 = o.cons(Up)
        assert np.allclose(U2, U, rtol=1e-13, atol=0)
        # p = R (n_h T_h + n_e T_e) with n_e = n_i (ambipolar), independent restatement
        mw = [ph.mixture.gas_params[sp] for sp in range(3)]
        ni = Up[5]
        nB = (U[0] - ni * mw[0] - ni * mw[1]) / mw[2]
        Te = Up[6] if two_t else Up[4]
        assert o.pressure(U) == pytest.approx(R * ((ni + nB) * Up[4] + ni * Te), rel=1e-13)
        # speed of sound uses the heavies' heat ratio 5/3 (all cv = 1.5 R)
        c = o.max_char_speed_point(U) - np.linalg.norm(U[1:4]) / U[0]
        assert c == pytest.approx(np.sqrt(5.0 / 3.0 * o.pressure(U) / U[0]), rel=1e-12)


@pytest.mark.parametrize("transport", [capi.ARGON_MINIMAL, capi.CONSTANT])
@pytest.mark.parametrize("two_t", [False, True])
def test_diffusion_velocities_conserve_mass_and_charge(transport, two_t):
    o, ph = tiny_oracle(two_t, transport)
    rng = np.random.default_rng(5)
    neq = o.neq
    mw = np.array([ph.mixture.gas_params[sp] for sp in range(3)])
    for U in sample_states(ph, 3, n=16):
        Up = o.prim(U)
        g = rng.normal(size=(3, neq)) * np.abs(Up)[None, :] * 2.0  # gradUp[eq + d*neq]
        buf, V = o.flux_transport(U, g.ravel())
        V = V[:9].reshape(3, 3)  # [d][sp]
        ni = Up[5]
        n = np.array([ni, ni, (U[0] - ni * (mw[0] + mw[1])) / mw[2]])
        Y = n * mw / U[0]
        assert np.abs(V @ Y).max() < 1e-12 * np.abs(V).max()           # sum_sp Y V = 0
        q = np.array([1.0, -1.0, 0.0])
        assert np.abs(V @ (q * n)).max() < 1e-9 * np.abs(V * n).max()  # ambipolar: no net current
        assert buf[0] > 0 and buf[2] > 0 and buf[3] >= 0


def test_argon_viscosity_formula():
    """neutral-dominated limit: mu -> 5/16 sqrt(pi m kB T) / Q22(T) (src/gas_transport.cpp:261-262)."""
    o, ph = tiny_oracle()
    X = np.zeros((3, 1))
    mw = [ph.mixture.gas_params[sp] for sp in range(3)]
    T, p = 5000.0, 101300.0
    alpha = 1e-12
    nh = p / (R * T)
    ni = alpha * nh
    rho = ni * (mw[0] + mw[1]) + (nh - ni) * mw[2]
    U = cases.plasma_conserved(ph, 3, np.array([rho]), [np.zeros(1)] * 3, np.array([T]), [np.array([ni])])[:, 0]
    buf, _ = o.flux_transport(U, np.zeros(3 * o.neq))
    kB = R / 6.0221409e23
    mu = 5.0 / 16.0 * np.sqrt(np.pi * mw[2] / 6.0221409e23 * kB * T) / (1.7e-18 * T ** -0.25)
    assert buf[0] == pytest.approx(mu, rel=1e-9)
    assert buf[2] == pytest.approx(mu * 15.0 / 4.0 * kB / (mw[2] / 6.0221409e23), rel=1e-9)
    assert buf[1] == 0.0


def test_arrhenius_source_against_numpy():
    o, ph = tiny_oracle(two_t=True)
    ch = ph.chemistry
    mw = [ph.mixture.gas_params[sp] for sp in range(3)]
    for U in sample_states(ph, 3, n=16, seed=9):
        Up = o.prim(U)
        g = np.zeros(3 * o.neq)
        src = o.source(U, Up, g)
        Th, Te, ni = Up[4], Up[6], Up[5]
        nB = (U[0] - ni * (mw[0] + mw[1])) / mw[2]
        n = np.array([ni, ni, nB])
        w = []
        for r in range(2):
            re_ = np.array([ch.reactant_stoich[sp + 3 * r] for sp in range(3)])
            A, b, E = (ch.rate_params[k + 3 * r] for k in range(3))
            T = max(Te, 2000.0)  # both reactions involve electrons
            w.append(A * T ** b * np.exp(-E / R / T) * np.prod(n ** re_))
        assert src[5] == pytest.approx((w[0] - w[1]) * mw[0], rel=1e-11)
        assert src[0] == 0 and np.all(src[1:4] == 0)
        assert src[4] == 0  # no radiation
        # electron energy: -sum dH_r w_r - elastic exchange (grad = 0)
        _, mt, _, _ = o.source_transport(U, Up, g)
        el = sum(1.5 * R * (Te - Th) * 2 * mw[1] * mw[sp] / (mw[sp] + mw[1]) ** 2 * ni * mt[sp] for sp in (0, 2))
        expect = -(ch.reaction_energies[0] * w[0] + ch.reaction_energies[1] * w[1]) - el
        assert src[6] == pytest.approx(expect, rel=1e-10)


def test_reaction_model_variants():
    oa, pa = tiny_oracle(reactions="arrhenius")
    ot, pt = tiny_oracle(reactions="tabulated_loglog")
    ob, pb = tiny_oracle(reactions="balance")
    oh, _ = tiny_oracle(reactions="hoffertlien")
    orad, _ = tiny_oracle(radiation=True)
    for U in sample_states(pa, 3, n=8, seed=11):
        Up = oa.prim(U)
        g = np.zeros(3 * oa.neq)
        sa, st = oa.source(U, Up, g), ot.source(U, Up, g)
        # a 257-point log-log table of the same law differs by interpolation error only
        assert st[5] == pytest.approx(sa[5], rel=2e-3, abs=1e-12 * abs(sa[5]))
        sb = ob.source(U, Up, g)
        assert np.isfinite(sb).all() and sb[5] != sa[5]
        assert np.isfinite(oh.source(U, Up, g)).all()
        sr = orad.source(U, Up, g)
        T = Up[4]
        tab = capi.reference_table("nec_sample_0")  # the reference's net-emission table, linear axes
        nec = np.interp(T, tab[:, 0], tab[:, 1])
        assert sr[4] == pytest.approx(-4.0 * np.pi * nec, rel=1e-12)
        assert sr[5] == sa[5]


def test_species_clamp_offset_is_the_references():
    """SourceTerm clamps equation 3+2+sp (hard-coded nvel = 3, src/source_term.cpp:129): with nvel = 2
    that index is past the active species of a ternary mixture, so a negative ion density is NOT
    clamped there; with nvel = 3 it is."""
    o3, ph = tiny_oracle()
    U = sample_states(ph, 3, n=1)[0]
    Up = o3.prim(U)
    U[5] = -abs(U[5])
    Up[5] = -abs(Up[5])
    s = o3.source(U, Up, np.zeros(3 * o3.neq))
    assert np.isfinite(s).all() and s[5] == 0.0  # n_i = n_e = 0 after the clamp: no reaction progress


@pytest.mark.parametrize("two_t", [False, True])
def test_mixture_transport_against_the_ternary_special_case(two_t):
    """GasMixtureTransport with the argon pair table evaluates the same integrals as GasMinimalTransport;
    the two differ only in the Debye length: sum_sp Z^2 n_sp / T_e (src/gas_transport.cpp:185-204) versus
    n_e/T_e + n_i/T_h (:226-229), i.e. they agree for a single temperature."""
    omin, ph = tiny_oracle(two_t, capi.ARGON_MINIMAL)
    omix, _ = tiny_oracle(two_t, capi.ARGON_MIXTURE)
    rng = np.random.default_rng(2)
    for U in sample_states(ph, 3, n=8, seed=4):
        Up = omin.prim(U)
        g = rng.normal(size=3 * omin.neq) * np.tile(np.abs(Up), 3)
        b0, V0 = omin.flux_transport(U, g)
        b1, V1 = omix.flux_transport(U, g)
        s0 = omin.source_transport(U, Up, g)
        s1 = omix.source_transport(U, Up, g)
        if not two_t:
            assert np.allclose(b1, b0, rtol=1e-13) and np.allclose(V1, V0, rtol=1e-12, atol=1e-14 * np.abs(V0).max())
            assert np.allclose(s1[1], s0[1], rtol=1e-13) and s1[0] == pytest.approx(s0[0], rel=1e-13)
        else:
            assert np.isfinite(b1).all() and b1[0] > 0
            assert not np.allclose(b1, b0, rtol=1e-6)  # the Debye lengths differ
            assert np.allclose(b1, b0, rtol=0.5)       # ... through a logarithm only


@pytest.mark.parametrize("two_t", [False, True])
def test_sheath_wall_flux_balances(two_t):
    """viscous_general wall with the sheath condition (src/equation_of_state.cpp:1909-1942): ions leave at
    the Bohm speed, electrons and neutrals carry the balancing charge and mass, so the boundary flux has
    no net mass flux and returns ions as neutrals (fully catalytic)."""
    mesh = meshgen.box_hex(1, 1, 1, periodic=(False,) * 3)
    ph = capi.argon_ternary_physics(capi.NS, two_t, capi.CONSTANT, None)
    bcs = [capi.make_bc(a + 1, capi.WALL, capi.VISC_GNRL, [3000.0, 8000.0, capi.ISOTH, capi.SHTH]) for a in range(6)]
    o = Oracle(mesh, capi.Disc(1, 0, 0, 0, 0), ph, bcs)
    mw = [ph.mixture.gas_params[sp] for sp in range(3)]
    for U in sample_states(ph, 3, n=6, seed=12):
        nor = np.array([0.3, -0.2, 0.9])
        g = np.zeros(3 * o.neq)
        f = o.bdr_flux(1, nor, U, g)
        # wall state: no slip, T_h = 3000 K, species densities of the interior state
        Up = o.prim(U)
        Upw = Up.copy()
        Upw[1:4] = 0.0
        Upw[4] = 3000.0
        Uw = o.cons(Upw)
        lf = o.lf(U, Uw, nor)
        visc = f - lf  # -1/2 (wall viscous flux + interior viscous flux (= 0 for zero gradients))
        nm = np.linalg.norm(nor)
        ni = Upw[5]
        Te = Upw[6] if two_t else 3000.0
        VB = np.sqrt((3000.0 + Te) * R / mw[0])
        # species equation: -1/2 * (-rho_i V_B |n|)
        assert visc[5] == pytest.approx(0.5 * ni * mw[0] * VB * nm, rel=1e-12)
        assert abs(visc[0]) == 0.0 and np.abs(visc[1:4]).max() < 1e-9 * abs(lf[1:4]).max()


@pytest.mark.parametrize("kind", ["ternary_2T", "six_species", "six_species_axisym"])
def test_boundary_viscous_flux_identity_for_mixtures(kind):
    """reference test/test_boundary_flux.cpp:88-166 (PerfectMixture + ConstantTransport, optionally
    axisymmetric): with nothing prescribed, ComputeBdrViscousFluxes == ComputeViscousFluxes . n, rel <= 5e-13
    per equation (the density row is skipped there too)."""
    axisym = kind.endswith("axisym")
    if kind == "ternary_2T":
        ph = capi.argon_ternary_physics(capi.NS, True, capi.CONSTANT, None)
    else:
        ph = capi.argon_six_species_physics(capi.NS, capi.CONSTANT, True, False)
    if axisym:
        mesh = meshgen.box_quad(1, 1, periodic=(False, False), origin=(0.1, 0.0))
        bcs = [capi.make_bc(a + 1, capi.WALL, capi.INV) for a in range(4)]
    else:
        mesh = meshgen.box_hex(1, 1, 1, periodic=(False,) * 3)
        bcs = [capi.make_bc(a + 1, capi.WALL, capi.INV) for a in range(6)]
    o = Oracle(mesh, capi.Disc(1, 0, 0, 1 if axisym else 0, 0), ph, bcs)
    dim = 2 if axisym else 3
    rng = np.random.default_rng(8)
    X = rng.uniform(0, 1, size=(dim, 12))
    states = cases.plasma_state(X, ph, nvel=3, seed=21).T
    for U in states:
        g = rng.uniform(-10.0, 10.0, size=(dim, o.neq)) * np.abs(o.prim(U))[None, :] * 1e-2
        n = np.zeros(3)
        n[:dim] = rng.standard_normal(dim)
        n /= np.linalg.norm(n)
        radius = 0.37 if axisym else -1.0
        fv = o.viscous_flux(U, g.ravel(), radius)
        ref = (fv[:dim] * n[:dim, None]).sum(axis=0)
        got = o.bdr_viscous_flux(U, g.ravel(), n, radius=radius)
        assert np.abs(got[1:] - ref[1:]).max() < 5e-13 * np.abs(ref[1:]).max()


def test_inadmissible_state_raises_instead_of_aborting():
    """The reference exits on a negative background density (src/equation_of_state.cpp:643-650); the oracle throws --
    also from inside its threaded loops -- and the binding turns that into a Python exception."""
    import pytest
    from oracle_lib import Oracle
    from tps_amd import capi, cases

    c = cases.argon_cyl3d(3, 8, 3, 1, False, capi.CONSTANT, "arrhenius", capi.VISC_ISOTH)
    U = c.state(seed=1, amp=0.005)
    U[5, 10:40] = 50.0 * U[0, 10:40]  # more ion mass than total mass
    o = Oracle(c.mesh, c.disc, c.physics, c.bcs)
    o.set_threads(4) if hasattr(o, "set_threads") else None
    with pytest.raises(RuntimeError, match="[Nn]egative"):
        o.mult(U)
