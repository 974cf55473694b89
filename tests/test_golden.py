"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py with the oracle).

CPU: the oracle reproduces its own committed output (drift check of the checker) and the state generators
are deterministic.  GPU: the HIP kernels are compared with the committed vectors directly -- this parity
check needs neither the oracle library nor anything outside the repository."""
import importlib.util
import os

import numpy as np
import pytest

from parity_util import RHS_RTOL, rel_maxnorm

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
make_golden = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(make_golden)


def _load(name):
    return np.load(os.path.join(HERE, "golden", name + ".npz"))


def _tol(name):
    return 5 * RHS_RTOL if name.startswith("argon") else RHS_RTOL  # 1 % plasma perturbations (see test_gpu_parity._tol)


@pytest.mark.parametrize("name", make_golden.NAMES)
def test_oracle_reproduces_its_golden_vectors(name):
    mesh, disc, ph, bcs, state = make_golden.golden_case(name)
    gold = _load(name)
    U = state()
    assert np.array_equal(U, gold["U"]), "the seeded state generator changed"
    r = make_golden.oracle_run(name, U)
    assert rel_maxnorm(r["y"], gold["y"]).max() < 1e-12
    assert np.abs(r["gradUp"] - gold["gradUp"]).max() <= 1e-12 * np.abs(gold["gradUp"]).max()
    assert r["max_char_speed"] == pytest.approx(float(gold["max_char_speed"]), rel=1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("name", make_golden.NAMES)
def test_hip_matches_golden_vectors(name):
    import torch
    from tps_amd.rhs_operator import RHSoperator

    mesh, disc, ph, bcs, _ = make_golden.golden_case(name)
    ex = make_golden.golden_extras(name) or {}
    gold = _load(name)
    op = RHSoperator(mesh, disc, ph, bcs)
    op.setDt(ex.get("dt", 0.0))
    op.setForcing(ex.get("forcing"))
    x = torch.tensor(np.ascontiguousarray(gold["U"]).ravel(), dtype=torch.float64, device=op.device)
    y = torch.empty_like(x)
    for _ in range(ex.get("ncalls", 1)):
        op.Mult(x, y, want_max_char_speed=True)
    got = {"y": y.cpu().numpy().reshape(gold["U"].shape), "gradUp": op.getGradients().cpu().numpy(),
           "max_char_speed": op.max_char_speed}
    op.close()
    err = rel_maxnorm(got["y"], gold["y"])
    print(name, "rel err per equation", err)
    assert err.max() < _tol(name)
    assert np.abs(got["gradUp"] - gold["gradUp"]).max() < _tol(name) * np.abs(gold["gradUp"]).max()
    assert got["max_char_speed"] == pytest.approx(float(gold["max_char_speed"]), rel=1e-12)
