"""Accuracy of the device's own elementary functions (tps_amd/csrc/fastmath.hpp) against numpy's libm, in units
in the last place: the closures of the plasma physics call them ~40 times per point where the reference calls
pow / exp / log, so their error is part of every parity statement."""
import ctypes as C

import numpy as np
import pytest

from tps_amd import capi

pytestmark = pytest.mark.gpu
EXP, EXP_UNCHECKED, LOG, LOG_POS, RCP, SQRT, RSQRT = range(7)


def dev(fn, x):
    import torch

    lib = capi.load()
    xd = torch.tensor(np.ascontiguousarray(x, dtype=np.float64), device="cuda")
    yd = torch.empty_like(xd)
    assert lib.tpsrhs_math_eval(fn, xd.numel(), C.c_void_p(xd.data_ptr()), C.c_void_p(yd.data_ptr())) == 0
    return yd.cpu().numpy()


def ulps(got, ref):
    return np.abs(got - ref) / np.spacing(np.abs(ref))


def test_exp():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-700, 700, 200000), rng.uniform(-2, 2, 200000), rng.uniform(-40, 40, 100000),
                        [0.0, -0.0, 1.0, -1.0, 709.0, -708.0]])
    for fn in (EXP, EXP_UNCHECKED):
        assert ulps(dev(fn, x), np.exp(x)).max() < 2.0
    with np.errstate(over="ignore"):
        sp = np.array([np.inf, -np.inf, 1.0e4, -1.0e4, -745.0, -746.0, np.nan, 710.0])
        g = dev(EXP, sp)
    assert g[0] == np.inf and g[1] == 0.0 and g[2] == np.inf and g[3] == 0.0 and np.isnan(g[6]) and g[7] == np.inf
    assert g[5] == 0.0 and 0.0 < g[4] < 1e-320  # exp(-745) is the smallest denormal: v_ldexp_f64 rounds into that range
    assert np.isnan(dev(EXP_UNCHECKED, np.array([np.nan]))[0])  # a NaN stays a NaN without the range check too


def test_log():
    rng = np.random.default_rng(2)
    x = np.concatenate([10.0 ** rng.uniform(-300, 300, 200000), rng.uniform(0.5, 2.0, 200000), 1.0 + rng.uniform(-1e-3, 1e-3, 100000),
                        [1.0, 2.0, 0.5, np.e, 5e-324, 1e-310, 1.7e308]])
    ref = np.log(x)
    for fn in (LOG, LOG_POS):
        got = dev(fn, x)
        nz = ref != 0.0
        assert ulps(got[nz], ref[nz]).max() < 4.0
        assert np.all(got[~nz] == 0.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        g = dev(LOG, np.array([0.0, -1.0, np.inf, np.nan, -np.inf]))
    assert g[0] == -np.inf and np.isnan(g[1]) and g[2] == np.inf and np.isnan(g[3]) and np.isnan(g[4])


def test_pow_through_exp_log():
    """pow(x, c) of the collision fits as exp(c log x): the composed error stays within a few ulp of libm's pow"""
    rng = np.random.default_rng(3)
    x = 10.0 ** rng.uniform(-3, 6, 200000)
    for c in (1.0472, 0.9148, 1.2435, 0.8264):
        got = dev(EXP_UNCHECKED, c * dev(LOG_POS, x))
        assert (np.abs(got / np.power(x, c) - 1.0)).max() < 4e-15


def test_rcp_sqrt():
    rng = np.random.default_rng(4)
    x = 10.0 ** rng.uniform(-150, 150, 300000)
    assert ulps(dev(RCP, x), 1.0 / x).max() < 2.0
    assert ulps(dev(SQRT, x), np.sqrt(x)).max() < 2.0
    assert ulps(dev(RSQRT, x), 1.0 / np.sqrt(x)).max() < 3.0
