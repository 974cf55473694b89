"""LinearTable on the reference's own tabulated data (test/test_table.cpp, test/tabulated.test).

The fixture tests/golden/tables/reference_tables.npz holds, bit for bit, the 14 rate-coefficient tables of
test/inputs/rate-coefficients/*.h5 and test/inputs/rad-data/nec_sample.0.h5 (generator next to it).  The checks
restate testTableInterpolator1D (test/test_table.cpp:16-121): findInterval brackets random abscissae of a
jittered 57-point grid, clamps below / above the range, and evaluation at a tabulated abscissa reproduces the
tabulated value to 1e-13 -- for the CPU oracle (not gpu) and for the HIP table_eval through the C ABI (gpu).
"""
import ctypes as C

import numpy as np
import pytest

from tps_amd import capi

RATE_TABLES = ["3BdyRecomb_4p", "3BdyRecomb_Ground", "3BdyRecomb_Metastable", "3BdyRecomb_Resonant", "DeExcitation_4p",
               "DeExcitation_Metastable", "DeExcitation_Resonant", "Excitation_4p", "Excitation_Metastable",
               "Excitation_Resonant", "Ionization", "StepIonization_4p", "StepIonization_Metastable",
               "StepIonization_Resonant"]
THRESHOLD = 1e-13  # scalarErrorThreshold, test/test_table.cpp:10


def hip_table_eval(table, x):
    import torch

    lib = capi.load()
    xd = torch.tensor(np.ascontiguousarray(x, dtype=np.float64), device="cuda")
    fd = torch.empty_like(xd)
    st = lib.tpsrhs_table_eval(C.byref(table), xd.numel(), C.c_void_p(xd.data_ptr()), C.c_void_p(fd.data_ptr()))
    assert st == 0, lib.tpsrhs_last_error().decode()
    return fd.cpu().numpy()


def test_fixture_is_the_reference_data():
    """shapes and a few values as h5dump prints them (test/inputs/rate-coefficients/Ionization.h5, rows 0, 1, 23)"""
    t = capi.reference_table("Ionization")
    assert t.shape == (500, 2)
    assert t[0, 0] == pytest.approx(299.963, rel=2e-6) and t[0, 1] == pytest.approx(1.80907e-256, rel=2e-6)
    assert t[1, 0] == pytest.approx(415.819, rel=2e-6) and t[1, 1] == pytest.approx(1.19728e-182, rel=2e-6)
    assert t[23, 0] == pytest.approx(2964.65, rel=2e-6) and t[23, 1] == pytest.approx(5.27665e-18, rel=2e-6)
    for name in RATE_TABLES:
        assert capi.reference_table(name).shape == (500, 2)
    assert capi.reference_table("nec_sample_0").shape == (60, 2)


def _interval_scale(t):
    """The reference's check is |f_ref - f| / |f_ref| < 1e-13 on its log-log table.  On linear axes LinearTable
    stores a_k = f_k - b_k x_k (src/table.cpp:39-50), so f(x_k) = a_k + b_k x_k carries a rounding error of the
    size of the interval's larger end value; the rate tables span 260 decades (k_f(300 K) = 1.8e-256), where that
    is all of f_k.  The error is therefore measured against max(|f_k|, |f_k+1|) -- identical to the reference's
    measure wherever neighbouring values are of one magnitude."""
    f = np.abs(t[:, 1])
    nxt = np.concatenate([f[1:], f[-1:]])
    return np.maximum(f, nxt) + 1e-300


def _find_interval_cases(rng):
    n = 57
    L = 5.0 * rng.uniform(0.2, 1.0)
    dx = L / (n - 1)
    x = np.arange(n) * dx + 0.4 * dx * (2.0 * rng.uniform(size=n) - 1.0)
    f = rng.uniform(size=n)
    keep = []
    return x, f, capi.make_table(x, f, False, False, keep), keep


def test_oracle_find_interval():
    from oracle_lib import table_eval

    rng = np.random.default_rng(3)
    x, f, tab, keep = _find_interval_cases(rng)
    xe = x[0] + (x[-1] - x[0]) * rng.uniform(size=100)
    _, idx = table_eval(tab, xe)
    assert np.all(idx >= 0) and np.all(idx <= len(x) - 2)
    assert np.all(x[idx] <= xe) and np.all(xe <= x[idx + 1])
    _, i0 = table_eval(tab, [x[0]])
    assert x[i0[0]] <= x[0] <= x[i0[0] + 1]
    _, lo = table_eval(tab, [x[0] - 0.1 * (x[-1] - x[0])])
    _, hi = table_eval(tab, [x[-1] + 0.1 * (x[-1] - x[0])])
    assert lo[0] == 0 and hi[0] == len(x) - 2  # test/test_table.cpp:63-74


@pytest.mark.parametrize("name", RATE_TABLES + ["nec_sample_0"])
def test_oracle_exact_interpolation(name):
    from oracle_lib import table_eval

    t = capi.reference_table(name)
    keep = []
    tab = capi.make_table(t[:, 0], t[:, 1], False, False, keep)
    f, _ = table_eval(tab, t[:, 0])
    assert np.all(np.abs(t[:, 1] - f) <= THRESHOLD * _interval_scale(t))  # test/test_table.cpp:111-118
    # between the abscissae: plain linear interpolation
    xm = 0.5 * (t[:-1, 0] + t[1:, 0])
    fm, _ = table_eval(tab, xm)
    ref = np.interp(xm, t[:, 0], t[:, 1])
    assert np.allclose(fm, ref, rtol=1e-12, atol=1e-300)


@pytest.mark.gpu
@pytest.mark.parametrize("name", RATE_TABLES + ["nec_sample_0"])
def test_hip_exact_interpolation(name):
    from oracle_lib import table_eval

    t = capi.reference_table(name)
    keep = []
    tab = capi.make_table(t[:, 0], t[:, 1], False, False, keep)
    f = hip_table_eval(tab, t[:, 0])
    assert np.all(np.abs(t[:, 1] - f) <= THRESHOLD * _interval_scale(t))
    # off the abscissae, inside and outside the range: the same numbers as the oracle
    rng = np.random.default_rng(11)
    xe = np.concatenate([rng.uniform(t[0, 0], t[-1, 0], 4000), [t[0, 0] - 50.0, t[-1, 0] + 1.0e4]])
    fo, _ = table_eval(tab, xe)
    fh = hip_table_eval(tab, xe)
    assert np.allclose(fh, fo, rtol=1e-13, atol=1e-300)


@pytest.mark.gpu
def test_hip_log_axes_match_oracle():
    """x_log / f_log of [reactions/reaction1] in test/inputs/input.tabulated_reaction.ini:100-103 (its table is a
    git-LFS pointer here): the logarithmic branches of LinearTable, HIP (own exp / log) against the oracle (libm)"""
    from oracle_lib import table_eval

    t = capi.reference_table("StepIonization_Metastable")
    for xl, fl in ((True, True), (True, False), (False, True)):
        keep = []
        tab = capi.make_table(t[:, 0], t[:, 1], xl, fl, keep)
        rng = np.random.default_rng(5)
        xe = np.concatenate([t[:, 0], rng.uniform(t[0, 0], t[-1, 0], 2000)])
        fo, idx = table_eval(tab, xe)
        fh = hip_table_eval(tab, xe)
        # linear f axis: a_k + b_k x cancels down from the interval's larger end value (see _interval_scale)
        scale = np.abs(fo) if fl else np.maximum(np.abs(fo), _interval_scale(t)[idx])
        assert np.all(np.abs(fh - fo) <= 2e-13 * scale), (xl, fl, (np.abs(fh - fo) / scale).max())
