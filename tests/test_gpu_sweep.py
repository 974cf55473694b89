"""The EXHAUSTIVE parity sweep over the plasma kernel instantiations as part of the GPU suite (round 3): every
(geometry, species count 3 ... 8, ambipolar or not, one or two temperatures, transport model, polynomial order, basis /
rule pair) the library builds -- 1008 instantiations of the three sweeps -- runs one small case against the oracle
(``tools/sweep_instantiations.py``: one worker process per family, so that a faulting kernel takes only its family with it).

Why it is a test and not only a tool: twice in round 3 ONE instantiation (3-D, Gauss-Lobatto pair, p = 1, seven species
with an electron equation, one temperature, mixture transport) returned a deterministically wrong viscous term after an
unrelated change elsewhere, while the sampled parity cases and the randomised sweeps stayed green.
A wrong instantiation now fails the suite."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("geo,expected", [("3d", 384), ("2d", 384), ("axi", 240)])
def test_every_plasma_instantiation_matches_the_oracle(geo, expected):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep_instantiations.py"), geo], capture_output=True, text=True,
                       timeout=1500)
    out = r.stdout
    bad = [l for l in out.splitlines() if "WRONG" in l or "EXCEPTION" in l or "skipped" in l]
    died = [l for l in out.splitlines() if l.startswith("==") and not l.rstrip().split("exit code ")[1].startswith("0")]
    ok = sum(1 for l in out.splitlines() if ": ok " in l)
    print(out[-1500:])
    assert not bad, bad[:5]
    assert not died, died
    assert ok == expected, (ok, expected)


def test_every_single_fluid_flavour_matches_the_oracle():
    """Round 4: the same for the single-fluid kernels -- dry air plain (2-D / 3-D, p = 1 ... 5, both basis pairs), with
    non-reflecting patches, with a sub-grid scale model + the viscous sponge, axisymmetric dry air and the table gas
    (p = 1 ... 4): 16 + 10 + 10 + 4 + 4 = 44 cases, two consecutive Mult calls each."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep_instantiations.py"), "single"], capture_output=True, text=True,
                       timeout=1500)
    out = r.stdout
    bad = [l for l in out.splitlines() if "WRONG" in l or "EXCEPTION" in l or "skipped" in l]
    died = [l for l in out.splitlines() if l.startswith("==") and not l.rstrip().split("exit code ")[1].split()[0] == "0"]
    ok = sum(1 for l in out.splitlines() if ": ok " in l)
    print(out[-2500:])
    assert not bad, bad[:5]
    assert not died, died
    assert ok == 44, ok
