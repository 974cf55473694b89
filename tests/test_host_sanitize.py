"""The DEVICE point physics of the plasma kernels under host sanitizers (round-2 advisor / round-3 review: "run the same closures
compiled for the host under ASan / UBSan"): tests/host_physics/ compiles tps_amd/csrc/{fastmath, physics_dryair,
physics_plasma, plasma_params_host}.hpp with g++ through a stand-in <hip/hip_runtime.h> and runs the state closure, transport,
nodal flux, sources, viscous traces (every wall type, both passes), Riemann fluxes and ghost states of 13 instantiations --
among them the seven-species kernel of DESIGN.md section 5 -- under AddressSanitizer + UndefinedBehaviorSanitizer, and once
more under clang's MemorySanitizer with every output checked for initialisation.  GPU sanitizers do not exist on the
MI355X pool; this is the check that the SOURCE of the point physics has no out-of-bounds index, no signed overflow, no
uninitialised read."""
import os
import subprocess
import sys

import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_physics")


@pytest.fixture(scope="module")
def built():
    r = subprocess.run(["make", "-C", HERE, "_build/libhostphys.so", "_build/hostphys_msan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.timeout(900)
def test_point_physics_is_clean_under_asan_ubsan_and_msan(built, tmp_path):
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    dump = str(tmp_path / "cases.bin")
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0", HOSTPHYS_DUMP=dump)
    r = subprocess.run([sys.executable, os.path.join(HERE, "run_cases.py")], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "ALL CLEAN" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
    m = subprocess.run([os.path.join(HERE, "_build", "hostphys_msan"), dump], capture_output=True, text=True)
    assert m.returncode == 0 and "MSAN CLEAN: 26 cases" in m.stdout, m.stdout[-2000:] + m.stderr[-4000:]
