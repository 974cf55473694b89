"""Register-spill hygiene of the shipped kernels (tools/spill_lib.py reads the code-object metadata of every kernel in
tps_amd/csrc/_obj): the set of kernels that spill VGPRs or use scratch must equal the committed allow-list
(tests/golden/spill_allowlist.txt), so that a new spiller fails the CPU suite and the build, and none of the kernels a
bench workload or a BASELINE.json configuration launches may be on it.

Why: round 2 met two heavily spilling instantiations with deterministically wrong results (DESIGN.md section 5,
"Spilled instantiations"); correctness of a spilling kernel rests on the parity suite and the randomised sweeps only.
Regenerate the list with `python tools/spill_report.py --write-allowlist` after a deliberate change."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
ALLOW = os.path.join(ROOT, "tests", "golden", "spill_allowlist.txt")

# (Cfg, physics) of every kernel the bench workloads and the BASELINE.json configurations launch
BENCH_INSTANTIATIONS = {
    "argon_p3 (the metric's workload)": "Cfg<3,3,0>,PlasmaPhys<3,3,3,true,false,1>",
    "cfg3 = configs[2]": "Cfg<3,2,0>,PlasmaPhys<3,3,3,true,false,1>",
    "cfg2 / cfg4 = configs[1], configs[3]": "Cfg<3,3,0>,DryAirPhys<3,false,false>",
    "cfg1 = configs[0]": "Cfg<3,1,0>,DryAirPhys<3,false,false>",
    "cfg5 = configs[4] (argon mixture transport)": "Cfg<2,3,0>,PlasmaPhys<2,3,3,true,true,2>",
    "cfg5_const (constant transport)": "Cfg<2,3,0>,PlasmaPhys<2,3,3,true,true,0>",
    # (torch6_mix -- the torch mixture with the argon mixture transport, a secondary line of bench.py -- is NOT here: its k_flux
    #  keeps 26 spilled registers at two waves per SIMD, like the other 2-D sweeps of four to six species)
    "torch6": "Cfg<2,3,0>,PlasmaPhys<2,3,6,false,true,0>",
    "gll_dry (Gauss-Lobatto pair, dry air p=3)": "Cfg<3,3,1>,DryAirPhys<3,false,false>",
    "lte_torch (table gas, axisymmetric)": "Cfg<2,3,0>,GasAxiPhys<true>",
}
# Waves per SIMD of the two hot sweeps of those workloads, by the register file (512 VGPRs per lane and SIMD on gfx950:
# floor(512 / allocated)): the budget each kernel was tuned to.  A change that pushes one of them over its edge halves
# or thirds its occupancy without a spill or any other sign -- round 3 met it: two more live registers in the six-species
# k_gradient, 256 -> 258, one wave instead of two, torch6 k_gradient 0.49 -> 0.67 ms.
BENCH_WAVES_PER_SIMD = {
    # (round 4: the lean gradient sweep of the 3-D ternary families is sized for THREE waves -- registers and 12 KB of LDS)
    "Cfg<3,3,0>,PlasmaPhys<3,3,3,true,false,1>": {"k_gradient": 3, "k_flux": 2},
    "Cfg<3,2,0>,PlasmaPhys<3,3,3,true,false,1>": {"k_gradient": 3, "k_flux": 2},
    # (round 4: the dry-air gradient sweep issues its neighbour records pair by pair: 128 registers = FOUR waves, no spill)
    "Cfg<3,3,0>,DryAirPhys<3,false,false>": {"k_gradient": 4, "k_flux": 3},
    "Cfg<2,3,0>,PlasmaPhys<2,3,3,true,true,0>": {"k_gradient": 2, "k_flux": 2},
    "Cfg<2,3,0>,PlasmaPhys<2,3,6,false,true,0>": {"k_gradient": 2, "k_flux": 2},
}


@pytest.fixture(scope="module")
def census():
    import spill_lib

    if not os.path.isdir(spill_lib.OBJ) or not any(f.endswith(".o") for f in os.listdir(spill_lib.OBJ)):
        pytest.skip("no object files (tps_amd/csrc/_obj): the census runs where the library was built")
    objs = [os.path.join(spill_lib.OBJ, f) for f in os.listdir(spill_lib.OBJ) if f.endswith(".o")]
    before = {o: os.stat(o).st_mtime_ns for o in objs}
    out = spill_lib.census()
    # the census must READ the objects: round 4 found llvm-objcopy rewriting its input in place (one file name = input AND
    # output), which bumped every object's mtime past the headers' and made a stale build look fresh to build()
    touched = [o for o in objs if os.stat(o).st_mtime_ns != before[o]]
    assert not touched, f"the spill census modified {len(touched)} object files"
    return out


def spillers(census):
    return sorted({k["kernel"] for k in census if k["vgpr_spill"] > 0 or k["scratch"] > 0})


def test_spilling_kernels_equal_the_allow_list(census):
    allowed = [l.strip() for l in open(ALLOW) if l.strip() and not l.startswith("#")]
    got = spillers(census)
    new = sorted(set(got) - set(allowed))
    gone = sorted(set(allowed) - set(got))
    assert not new, f"{len(new)} kernels spill VGPRs / use scratch and are not on the allow-list, e.g. {new[:5]}"
    assert not gone, f"{len(gone)} kernels of the allow-list no longer spill: regenerate it, e.g. {gone[:5]}"


def test_bench_and_baseline_kernels_do_not_spill(census):
    bad = []
    for what, inst in BENCH_INSTANTIATIONS.items():
        ks = [k for k in census if ("<" + inst + ">") in k["kernel"]]
        assert any(k["kernel"].startswith("k_flux<") for k in ks), f"no kernel of {what} ({inst}) in the build"
        for k in ks:
            if k["vgpr_spill"] > 0:
                bad.append((what, k["kernel"], "vgpr_spill", k["vgpr_spill"]))
            elif k["scratch"] > 0:
                # a private segment in the metadata without a single scratch instruction (frame slots of SGPR spills that
                # all ended up in VGPR lanes): tolerated; any scratch access is not
                import kernel_isa

                n = sum(kernel_isa.summary(kernel_isa.mix(body))["scratch"]
                        for _, body in kernel_isa.kernels(os.path.join(ROOT, "tps_amd", "csrc", "_obj", k["unit"]), "")
                        if _norm(_) == k["kernel"])
                if n > 0:
                    bad.append((what, k["kernel"], "scratch instructions", n))
    assert not bad, bad


def test_bench_kernels_keep_their_occupancy(census):
    bad = []
    for inst, want in BENCH_WAVES_PER_SIMD.items():
        for kern, waves in want.items():
            ks = [k for k in census if k["kernel"] == f"{kern}<{inst}>"]
            assert len(ks) == 1, (kern, inst, len(ks))
            regs = ks[0]["vgpr"] + ks[0]["agpr"]
            if 512 // max(regs, 1) < waves:
                bad.append((kern, inst, f"{regs} registers: {512 // regs} wave(s) per SIMD, tuned for {waves}"))
    assert not bad, bad


def _norm(demangled):
    import re

    d = re.sub(r"\(.*", "", demangled).replace("tpsrhs::", "").replace("void ", "")
    return re.sub(r"\s+", "", d)
