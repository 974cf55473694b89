"""The known answer of test/test_speed_of_sound.cpp:37-102: a five-species "air" mixture (CO2, Ar, O2, E, N2 of
test/inputs/perfectGas.air.ini, mole fractions 4.07e-4 / 9.34e-3 / 0.20946 / 0 / rest, rho = 1.2041 kg/m3,
T = 293.15 K) has the speed of sound of dry air, sqrt(1.4 R / 28.964e-3 T), to 1e-4 -- for the CPU oracle's
PerfectMixture and, through the C ABI, for the device closure of the HIP kernels."""
import ctypes as C

import numpy as np
import pytest

from tps_amd import capi, meshgen

R = capi.UNIVERSALGASCONSTANT
AIR_MW = 28.964e-3
RHO, TH = 1.2041, 293.15


def air_primitive():
    X = np.array([0.0407e-2, 0.934e-2, 20.946e-2, 0.0, 0.0])
    X[4] = 1.0 - X[:4].sum()
    ntot = RHO / AIR_MW
    prim = np.zeros(9)  # rho, u, v, w, T, n_CO2, n_Ar, n_O2, n_E
    prim[0], prim[4] = RHO, TH
    prim[5:9] = ntot * X[:4]
    return prim, X, ntot


def dry_air_sound():
    return np.sqrt(1.4 * R / AIR_MW * TH)


def oracle_case():
    from oracle_lib import Oracle

    ph = capi.air_five_species_physics()
    mesh = meshgen.box_hex(3, 3, 3)  # periodic: the closures are point-wise, the mesh only carries the operator
    return Oracle(mesh, capi.Disc(1, 0, 0, 0, 0), ph, []), ph, mesh


def test_oracle_air_speed_of_sound():
    o, ph, _ = oracle_case()
    prim, X, ntot = air_primitive()
    mw = np.array([ph.mixture.gas_params[sp + capi.SPECIES_MW * 5] for sp in range(5)])
    assert (ntot * X * mw).sum() == pytest.approx(RHO, rel=2e-4)  # "Input rho ~ Computed rho" of the reference
    U = o.cons(prim)
    sound = o.max_char_speed_point(U)  # zero velocity: |u| + c = c
    assert abs(dry_air_sound() - sound) / dry_air_sound() < 1e-4  # test/test_speed_of_sound.cpp:84-91
    # primitive branch of ComputeSpeedOfSound (src/equation_of_state.cpp:1405-1420) restated: same number
    cv = np.array([ph.mixture.molar_cv[sp] for sp in range(5)]) * R
    n = ntot * X
    nB = (prim[0] - (n[:4] * mw[:4]).sum()) / mw[4]
    n_all = np.array([n[0], n[1], n[2], n[3], nB])
    p = R * TH * n_all.sum()
    heavy = [0, 1, 2, 4]
    gamma = 1.0 + n_all[heavy].sum() * R / (n_all[heavy] * cv[heavy]).sum()
    assert gamma == pytest.approx(1.4, rel=5e-4)  # "DryAir gamma: 1.4 ~ Computed gamma" (printed, not gated, by the reference)
    assert np.sqrt(gamma * p / RHO) == pytest.approx(sound, rel=1e-14)
    assert o.prim(U) == pytest.approx(prim, rel=1e-14, abs=1e-300)


@pytest.mark.gpu
def test_hip_air_speed_of_sound():
    import torch

    from tps_amd.rhs_operator import RHSoperator

    o, ph, mesh = oracle_case()
    prim, _, _ = air_primitive()
    U = o.cons(prim)
    op = RHSoperator(mesh, capi.Disc(1, 0, 0, 0, 0), ph, [])
    lib = capi.load()
    # a few states: the reference's, and the same gas moving / hotter
    states = np.stack([U, o.cons(prim + np.array([0, 30.0, -5.0, 2.0, 0, 0, 0, 0, 0])),
                       o.cons(prim * np.array([1, 1, 1, 1, 3.0, 1, 1, 1, 1]))], axis=1)  # [neq][n]
    n = states.shape[1]
    xd = torch.tensor(np.ascontiguousarray(states), device="cuda")
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    for quantity, ref_fn in ((2, None), (3, o.max_char_speed_point), (1, o.pressure)):
        st = lib.tpsrhs_eval_pointwise(op._h, quantity, n, C.c_void_p(xd.data_ptr()), C.c_void_p(out.data_ptr()))
        assert st == 0, lib.tpsrhs_last_error().decode()
        got = out.cpu().numpy()
        if quantity == 2:
            assert abs(dry_air_sound() - got[0]) / dry_air_sound() < 1e-4  # the reference's gate
            assert got[0] == pytest.approx(o.max_char_speed_point(U), rel=1e-14)
        else:
            ref = np.array([ref_fn(np.ascontiguousarray(states[:, i])) for i in range(n)])
            assert got == pytest.approx(ref, rel=1e-14)
    pr = torch.empty(9 * n, dtype=torch.float64, device="cuda")
    assert lib.tpsrhs_eval_pointwise(op._h, 0, n, C.c_void_p(xd.data_ptr()), C.c_void_p(pr.data_ptr())) == 0
    got = pr.cpu().numpy().reshape(9, n)
    for i in range(n):
        assert got[:, i] == pytest.approx(o.prim(np.ascontiguousarray(states[:, i])), rel=1e-14, abs=1e-300)
    op.close()
