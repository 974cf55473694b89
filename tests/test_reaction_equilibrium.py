"""The reference's `test/reaction.test` ("Test correct equilibrium is achieved", `test/inputs/input.reaction.ini`):
a uniform argon gas ionises through `Ar <=> Ar.+1 + E` (Arrhenius A = 1e-10, b = 4, detailed balance with
K_eq = 1e-10 T^4) until the backward rate balances the forward one; no transport, periodic square, order 2,
Gauss-Lobatto pair, 3000 RK4 steps of 2e-4 s.  The reference compares with a stored solution (a git-LFS pointer
here); the state is uniform, so every node integrates the same ODE and the answer is known: the 0-D system below,
advanced with the same RK4 (src/chemistry.cpp:160-240, src/reaction.cpp:41-53, src/source_term.cpp:107-251), and
the equilibrium it approaches, n_i n_e / n_Ar = K_eq(T) with the total energy conserved."""
import numpy as np
import pytest

from tps_amd import capi, meshgen

R = capi.UNIVERSALGASCONSTANT
M_AR, M_E = 2.896439e-2, 1.0e-7
MW = np.array([M_AR - M_E, M_E, M_AR])  # Ar.+1, E, Ar (mixture order)
CV = 2.49996 * R
E_FORM = 1.0e4
DT, STEPS = 2.0e-4, 3000


def _physics():
    ph = capi.argon_ternary_physics(capi.NS, False, capi.CONSTANT, None, ambipolar=False)
    mx = ph.mixture
    for sp in range(3):
        mx.gas_params[sp + capi.SPECIES_MW * 3] = MW[sp]
        mx.gas_params[sp + capi.FORMATION_ENERGY * 3] = E_FORM if sp == 0 else 0.0
        mx.molar_cv[sp] = 2.49996
    ct = ph.constant_transport
    ct.viscosity = ct.bulk_viscosity = ct.thermal_conductivity = ct.electron_thermal_conductivity = 0.0
    for sp in range(3):
        ct.diffusivity[sp] = ct.mt_freq[sp] = 0.0
    ch = ph.chemistry
    ch.num_reactions, ch.minimum_temperature = 1, 0.0
    ch.reaction_energies[0], ch.detailed_balance[0], ch.reaction_models[0] = 1.0e4, 1, capi.ARRHENIUS
    for sp, (re_, pr) in enumerate(((0, 1), (0, 1), (1, 0))):  # Ar -> Ar.+1 + E
        ch.reactant_stoich[sp], ch.product_stoich[sp] = re_, pr
    for k, v in enumerate((1.0e-10, 4.0, 0.0)):
        ch.rate_params[k] = v
        ch.equilibrium_constant_params[k] = v
    return ph


def _initial():
    rho, p = 1.2, 101300.0
    n_ar = rho / M_AR
    T0 = p / (R * n_ar)
    return np.array([rho, 0.0, 0.0, n_ar * CV * T0, 0.0, 0.0])  # rho, rho u, rho v, rho E, rho Y_ion, rho Y_e


def _rhs0d(u):
    rho, rhoE, n_i, n_e = u[0], u[3], u[4] / MW[0], u[5] / MW[1]
    n_ar = (rho - u[4] - u[5]) / MW[2]
    T = (rhoE - n_i * E_FORM) / (CV * (n_i + n_e + n_ar))
    kf = 1.0e-10 * T ** 4
    keq = 1.0e-10 * T ** 4
    q = kf * (n_ar - n_i * n_e / keq)
    return np.array([0.0, 0.0, 0.0, 0.0, MW[0] * q, MW[1] * q]), T, (n_i, n_e, n_ar)


def _ode_reference():
    u = _initial()
    for _ in range(STEPS):  # MFEM's RK4Solver::Step
        k1 = _rhs0d(u)[0]
        k2 = _rhs0d(u + 0.5 * DT * k1)[0]
        k3 = _rhs0d(u + 0.5 * DT * k2)[0]
        k4 = _rhs0d(u + DT * k3)[0]
        u = u + DT / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)
    return u


def test_the_ode_approaches_the_mass_action_equilibrium():
    """the answer itself: after 0.6 s the gas is within 2 % of n_i n_e / n_Ar = K_eq(T), 11 % ionised, 50 K colder"""
    u = _ode_reference()
    _, T, (n_i, n_e, n_ar) = _rhs0d(u)
    print("T", T, "ionisation degree", n_i / (n_i + n_ar), "n_i n_e / n_Ar / K_eq", n_i * n_e / n_ar / (1e-10 * T ** 4))
    assert abs(n_i - n_e) < 1e-12 * n_i  # quasi-neutral by stoichiometry
    assert 0.05 < n_i / (n_i + n_ar) < 0.2 and 200.0 < T < 290.0
    assert abs(n_i * n_e / n_ar / (1e-10 * T ** 4) - 1.0) < 0.05


def test_oracle_time_loop_follows_the_ode():
    from oracle_lib import Oracle

    ph = _physics()
    mesh = meshgen.box_quad(3, 3, lengths=(2.0, 2.0), origin=(-1.0, -1.0))
    o = Oracle(mesh, capi.Disc(2, 1, 1, 0, 0), ph, [])
    U = np.repeat(_initial()[:, None], o.ndofs, axis=1)
    xa, t, _, bad = o.advance(U, 0.0, DT, STEPS, True)
    ref = _ode_reference()
    err = np.abs(xa - ref[:, None]).max(axis=1) / np.maximum(np.abs(ref), 1e-300)
    print("oracle vs ODE", err)
    assert bad == 0 and abs(t - DT * STEPS) < 1e-12
    assert err[[0, 3, 4, 5]].max() < 1e-9 and np.abs(xa[1:3]).max() < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("pair", [(1, 1), (0, 0)])
def test_device_time_loop_reaches_the_equilibrium(pair):
    """the HIP path, through tpsrhs_advance: 3000 RK4 steps = 12 000 Mult calls on the reference's setting (order 2,
    Gauss-Lobatto pair) and on the collocated pair"""
    import torch
    from tps_amd.rhs_operator import RHSoperator

    ph = _physics()
    # MFEM's periodic-square.mesh: 3 x 3 quads on [-1, 1]^2 (with the input's dt = 2e-4 an 8 x 8 mesh is beyond the
    # explicit stability limit of the acoustic modes and blows up -- in the oracle too)
    mesh = meshgen.box_quad(3, 3, lengths=(2.0, 2.0), origin=(-1.0, -1.0))
    op = RHSoperator(mesh, capi.Disc(2, pair[0], pair[1], 0, 0), ph, [])
    U = np.repeat(_initial()[:, None], op.NDofs, axis=1)
    x = torch.tensor(U.ravel(), dtype=torch.float64, device=op.device)
    t, _, bad = op.advance(x, 0.0, DT, STEPS, True)
    xa = x.cpu().numpy().reshape(U.shape)
    op.close()
    ref = _ode_reference()
    err = np.abs(xa - ref[:, None]).max(axis=1) / np.maximum(np.abs(ref), 1e-300)
    print("HIP vs ODE", err)
    assert bad == 0 and abs(t - DT * STEPS) < 1e-12
    assert err[[0, 3, 4, 5]].max() < 1e-9 and np.abs(xa[1:3]).max() < 1e-9
    _, T, (n_i, n_e, n_ar) = _rhs0d(xa[:, 0])
    assert abs(n_i * n_e / n_ar / (1e-10 * T ** 4) - 1.0) < 0.05
