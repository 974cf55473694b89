"""Randomised parity sweep (run on a GPU box): random mesh kind, order, physics variant, boundary types and
state seed; HIP vs oracle per case.  Not part of the test suite: a one-off robustness check.
    python tools/fuzz_parity.py [ncases] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

from parity_util import RHS_RTOL, hip_mult, oracle_mult  # noqa: E402
from tps_amd import capi, cases, meshgen  # noqa: E402
from tps_amd.rhs_operator import node_coordinates  # noqa: E402

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
t0 = time.time()
for it in range(ncases):
    fluid = rng.choice(["dry", "argon3", "argon3n", "argon6", "argon4a", "argon5a", "argon5", "argon7", "argon7a", "argon8",
                        "argon8a", "argon4", "argon6a"], p=[0.2, 0.18, 0.08, 0.12, 0.06, 0.06, 0.06, 0.05, 0.04, 0.05, 0.04, 0.03, 0.03])
    geo = rng.choice(["cyl3d", "box3d", "box2d", "axisym"])
    # p = 1 .. 5 for every fluid since round 3 (dry air axisymmetric: 1 .. 4)
    pmax = (6 if geo != "axisym" else 5) if fluid == "dry" else 6
    order = int(rng.integers(1, pmax))
    eq = capi.NS if rng.random() < 0.85 else capi.EULER
    wall = int(rng.choice([capi.INV, capi.SLIP, capi.VISC_ADIAB, capi.VISC_ISOTH]))
    seed = int(rng.integers(1, 1000))
    amp = 0.05 if fluid == "dry" else (0.005 if order == 1 else 0.01)
    # rounding grows with the order (the residual is a difference quotient over a p-times finer node spacing):
    # (p / 3)^2 above p = 3 -- the sweep of round 2 saw 1.2 x the p <= 3 tolerance at p = 5 in 2 of 4 000 cases
    tol = RHS_RTOL * 0.05 / amp * max(1.0, (order / 3.0) ** 2)
    desc = f"{fluid} {geo} p={order} eq={eq} wall={wall} seed={seed}"
    if fluid == "dry":
        ph = capi.dry_air_physics(eq, visc_mult=float(rng.choice([1.0, 50.0, 1000.0])), bulk_visc_mult=float(rng.random()))
        if eq == capi.NS and geo in ("cyl3d", "box3d", "box2d") and rng.random() < 0.3:  # the LES flavour of the kernels
            if geo != "box2d":
                ph.sgs.model_type = int(rng.choice([capi.SGS_NONE, capi.SGS_SMAGORINSKY, capi.SGS_SIGMA]))
                ph.sgs.model_floor = float(rng.choice([0.0, 0.002]))
            if geo == "cyl3d" and rng.random() < 0.5:  # (not on the fully periodic boxes: DESIGN.md section 1)
                vs = ph.visc_sponge
                vs.enabled, vs.width, vs.ratio = 1, float(rng.uniform(0.5, 3.0)), float(rng.uniform(0.5, 20.0))
                vs.normal[0], vs.normal[1], vs.point[0], vs.point[1] = rng.normal(), rng.normal(), rng.normal(), rng.normal()
            desc += f" sgs={ph.sgs.model_type} sponge={ph.visc_sponge.enabled}"
            if ph.sgs.model_type == capi.SGS_SIGMA:
                # the eigenvalue route of the sigma model (acos of a ratio that approaches +-1 for nearly two-dimensional
                # strain) amplifies the rounding of the gradient: "more sensitive to perturbations in g", src/fluxes.cpp:
                # 584-587; 2 of 5 000 cases of the round-2 sweep reached 1.6e-11
                tol *= 5
    else:
        two_t = bool(rng.random() < 0.5)
        if fluid not in ("argon3", "argon3n"):
            levels, ambi = {"argon6": (3, False), "argon4a": (1, True), "argon5a": (2, True), "argon5": (2, False),
                            "argon7": (4, False), "argon7a": (4, True), "argon8": (5, False), "argon8a": (5, True),
                            "argon4": (1, False), "argon6a": (3, True)}[fluid]
            trs = [capi.CONSTANT] if levels == 5 else [capi.CONSTANT, capi.ARGON_MIXTURE]  # mixture transport: <= 7 species
            ph = capi.argon_levels_physics(levels, ambi, eq, int(rng.choice(trs)), two_t,
                                           bool(rng.random() < 0.7), radiation=bool(rng.random() < 0.5),
                                           third_order_ke=False)
        else:
            tr = int(rng.choice([capi.CONSTANT, capi.ARGON_MINIMAL, capi.ARGON_MIXTURE]))
            ph = capi.argon_ternary_physics(eq, two_t, tr, rng.choice(["arrhenius", "tabulated", "balance", None]),
                                            third_order_ke=False, radiation=bool(rng.random() < 0.5),
                                            ambipolar=(fluid == "argon3"))
        ph.gas_transport.multiply = 1
        for k in range(4):
            ph.gas_transport.flux_trns_multiplier[k] = 30.0
        ph.gas_transport.diff_mult = ph.gas_transport.mobil_mult = 30.0
        desc += f" 2T={two_t} tr={ph.transport_model}"
    try:
        if geo == "cyl3d":
            nr, nt, nz = int(rng.integers(3, 6)), int(rng.integers(8, 14)), int(rng.integers(3, 5))
            if fluid == "dry":
                c = cases.cyl3d(nr, nt, nz, order, eq, wall)
                c.physics = ph
            else:
                c = cases.argon_cyl3d(nr, nt, nz, order, physics=ph, wall_type=wall)
            mesh, disc, bcs = meshgen.scramble_orientations(c.mesh, seed), c.disc, c.bcs
        elif geo == "axisym":
            if fluid == "dry":
                c = cases.dry_air_axisym(int(rng.integers(3, 7)), int(rng.integers(3, 8)), order, eq, wall, r_in=0.01 * rng.integers(0, 3), warp=0.05 * rng.integers(0, 2))
                c.physics = ph
            else:
                c = cases.argon_axisym(int(rng.integers(3, 7)), int(rng.integers(3, 8)), order, physics=ph, wall_type=wall, r_in=0.01 * rng.integers(0, 3))
            mesh, disc, bcs = c.mesh, c.disc, c.bcs
        else:
            dim = 3 if geo == "box3d" else 2
            n = [int(rng.integers(3, 6)) for _ in range(dim)]
            mesh = (meshgen.box_hex(*n, warp=0.1) if dim == 3 else meshgen.box_quad(*n, warp=0.1))
            mesh = meshgen.scramble_orientations(mesh, seed)
            disc, bcs = capi.Disc(order, 0, 0, 0, 0), []
            c = None
        # the non-collocated Gauss-Lobatto pair: dry air and the ternary mixtures, planar 2-D / 3-D, p <= 3, no LES flavour
        gll = (geo != "axisym" and order <= 3 and rng.random() < 0.2  # (every species count since round 3)
               and not (fluid == "dry" and (ph.sgs.model_type or ph.visc_sponge.enabled)))
        if gll:
            disc = capi.Disc(order, 1, 1, 0, 0)
            desc += " GLL"
        X = node_coordinates(mesh, order, 1 if gll else 0)
        if fluid == "dry":
            U = cases.dry_air_state(X, seed=seed, nvel=3 if disc.axisymmetric else None,
                                    vel0=(1.0, 20.0, 3.0) if disc.axisymmetric else (20.0, 0.0, 0.0))
        else:
            U = cases.plasma_state(X, ph, nvel=3 if (disc.axisymmetric or mesh.dim == 3) else 2, seed=seed, amp=amp,
                                   vel0=(1.0, 20.0, 3.0) if disc.axisymmetric else (20.0, 0.0, 0.0))
        ref = oracle_mult(mesh, disc, ph, bcs, U)
        if not np.isfinite(ref["y"]).all():
            print(f"[{it}] {desc}: oracle not finite (inadmissible input), skipped")
            continue
        got = hip_mult(mesh, disc, ph, bcs, U)
        scale = np.abs(ref["y"]).reshape(U.shape[0], -1).max(axis=1)
        if U.shape[0] >= 4:
            nv = 3 if (disc.axisymmetric or mesh.dim == 3) else 2
            scale[1:1 + nv] = scale[1:1 + nv].max()
        err = (np.abs(got["y"] - ref["y"]).reshape(U.shape[0], -1).max(axis=1) / np.maximum(scale, 1e-300)).max()
        gerr = np.abs(got["gradUp"] - ref["gradUp"]).max() / np.abs(ref["gradUp"]).max()
        worst = max(worst, err / tol)
        flag = "" if (err < tol and gerr < tol) else "   <<<<<< FAIL"
        print(f"[{it}] {desc}: y {err:.2e} grad {gerr:.2e} (tol {tol:.0e}){flag}", flush=True)
    except Exception as exc:  # noqa: BLE001
        print(f"[{it}] {desc}: EXCEPTION {type(exc).__name__}: {exc}", flush=True)
print(f"worst err/tol = {worst:.3f}; {time.time() - t0:.0f} s")
