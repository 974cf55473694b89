"""TEST INFRASTRUCTURE: drives tests/host_physics/_build/libhostphys.so (the device point physics compiled for the host with
ASan + UBSan) on the states of small cases.  Started by tests/test_host_sanitize.py with libasan preloaded; a sanitizer
report aborts the process (non-zero exit)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from tps_amd import capi, cases  # noqa: E402

lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libhostphys.so"))
lib.hostphys_run.restype = C.c_int
dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
lib.hostphys_run.argtypes = [C.c_int] * 5 + [C.POINTER(capi.Disc), C.POINTER(capi.Physics), C.c_int, C.POINTER(capi.BC), C.c_long, dp, dp,
                             dp, dp, dp, dp]


def run(tag, geometry, nsp, ambi, two_t, tr, wall_types, third_order=False, n=96, euler=False):
    eq = capi.EULER if euler else capi.NS
    if nsp == 3:
        ph = capi.argon_ternary_physics(eq, two_t, tr, "arrhenius", ambipolar=ambi, third_order_ke=third_order and tr != capi.CONSTANT)
    else:
        ph = capi.argon_levels_physics(nsp - 3, ambi, eq, tr, two_t, True, third_order_ke=third_order and tr != capi.CONSTANT)
    dim = 3 if geometry == 3 else 2
    nvel = 3 if geometry in (1, 3) else 2
    rng = np.random.default_rng(7 + nsp)
    X = rng.uniform(0.1, 1.0, size=(dim, n))
    U = np.ascontiguousarray(cases.plasma_state(X, ph, nvel=nvel, seed=5, amp=0.05, vel0=(20.0, 3.0, 1.0)[:nvel]))
    neq = U.shape[0]
    # a primitive gradient of plausible size: every row a few per cent of its primitive per unit length, both signs
    G = np.zeros((dim * neq, n))
    scale = np.array([1.0] + [20.0] * nvel + [5000.0] + [1.0] * (neq - nvel - 2))
    dens = np.abs(U[nvel + 2:nvel + 2 + (neq - nvel - 2 - (1 if two_t else 0))]).max(axis=1) if neq > nvel + 2 else []
    for d in range(dim):
        G[d * neq:(d + 1) * neq] = rng.normal(size=(neq, n)) * scale[:, None] * 0.3
    N = np.ascontiguousarray(rng.normal(size=(n, dim)) * 0.01)
    inlet = cases.argon_inlet_state(ph, nvel)
    bcs = [capi.make_bc(1, capi.INLET, capi.SUB_DENS_VEL, inlet), capi.make_bc(2, capi.OUTLET, capi.SUB_P, [101300.0])]
    for k, w in enumerate(wall_types):
        if w == "GNRL_SHEATH":
            bcs.append(capi.make_bc(3 + k, capi.WALL, capi.VISC_GNRL, [3000.0, 9000.0, capi.ISOTH, capi.SHTH if two_t else capi.ISOTH]))
        elif w == "GNRL_ADIAB":
            bcs.append(capi.make_bc(3 + k, capi.WALL, capi.VISC_GNRL, [3000.0, 9000.0, capi.ADIAB, capi.ADIAB]))
        else:
            bcs.append(capi.make_bc(3 + k, capi.WALL, w, [3000.0]))
    arr = (capi.BC * len(bcs))(*bcs)
    F, S, FN = np.zeros((dim * neq, n)), np.zeros((neq, n)), np.zeros((neq, n))
    disc = capi.Disc(1, 0, 0, 1 if geometry == 1 else 0, 0)
    for use_bc_in_grad in (0, 1):
        disc.use_bc_in_grad = use_bc_in_grad
        if os.environ.get("HOSTPHYS_DUMP"):  # the same case for the stand-alone MemorySanitizer driver (harness.cpp, HOSTPHYS_MAIN)
            with open(os.environ["HOSTPHYS_DUMP"], "ab") as fh:
                fh.write(np.array([geometry, nsp, int(ambi), int(two_t), {capi.CONSTANT: 0, capi.ARGON_MINIMAL: 1, capi.ARGON_MIXTURE: 2}[tr],
                                   len(bcs), n, neq], dtype=np.int64).tobytes())
                fh.write(bytes(disc) + bytes(ph) + bytes(arr) + U.tobytes() + np.ascontiguousarray(G).tobytes() + N.tobytes())
        rc = lib.hostphys_run(geometry, nsp, int(ambi), int(two_t), {capi.CONSTANT: 0, capi.ARGON_MINIMAL: 1, capi.ARGON_MIXTURE: 2}[tr], C.byref(disc),
                              C.byref(ph), len(bcs), arr, n, U, np.ascontiguousarray(G), N, F, S, FN)
        assert rc == 0, f"{tag}: hostphys_run returned {rc}"
    print(f"host physics {tag}: {n} states x {len(bcs)} boundary attributes, neq {neq}: clean, |F| max {np.abs(F).max():.3e}", flush=True)


WALLS = [capi.VISC_ISOTH, capi.VISC_ADIAB, capi.INV, "GNRL_SHEATH", "GNRL_ADIAB"]
run("3d ternary ambipolar 1T minimal (the metric's)", 3, 3, True, False, capi.ARGON_MINIMAL, WALLS, third_order=True)
run("3d ternary ambipolar 1T minimal first-order", 3, 3, True, False, capi.ARGON_MINIMAL, WALLS)
run("3d ternary 2T minimal", 3, 3, False, True, capi.ARGON_MINIMAL, WALLS, third_order=True)
run("3d ternary ambipolar 2T mixture", 3, 3, True, True, capi.ARGON_MIXTURE, WALLS, third_order=True)
run("3d seven species 1T mixture (DESIGN section 5)", 3, 7, False, False, capi.ARGON_MIXTURE, WALLS)
run("3d seven species 1T mixture, Euler", 3, 7, False, False, capi.ARGON_MIXTURE, WALLS, euler=True)
run("3d four species ambipolar 2T mixture", 3, 4, True, True, capi.ARGON_MIXTURE, WALLS)
run("3d six species 2T mixture", 3, 6, False, True, capi.ARGON_MIXTURE, WALLS, third_order=True)
run("3d eight species ambipolar 2T constant", 3, 8, True, True, capi.CONSTANT, WALLS)
run("2d ternary ambipolar 2T constant", 2, 3, True, True, capi.CONSTANT, WALLS + [capi.SLIP])
run("2d five species 1T mixture", 2, 5, False, False, capi.ARGON_MIXTURE, WALLS)
run("axisymmetric ternary ambipolar 2T minimal (cfg5)", 1, 3, True, True, capi.ARGON_MINIMAL, WALLS)
run("axisymmetric six species 2T mixture (torch6)", 1, 6, False, True, capi.ARGON_MIXTURE, WALLS)
print("ALL CLEAN")
