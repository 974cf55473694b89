"""EXHAUSTIVE parity sweep over the plasma kernel instantiations (run on a GPU box): every (geometry, species count,
ambipolar, two-temperature, transport model, polynomial order, basis / rule pair) the library builds, one small case
each, against the oracle.  Complements the randomised tools/fuzz_parity.py: round 3 met an instantiation with a wrong
species residual in some lanes (and one with a memory fault) that no test had touched.

    python tools/sweep_instantiations.py <geo: 3d|2d|axi> [nsp ...]      # driver: one worker process per (nsp, ambipolar)
A worker that dies (GPU fault) takes only its own cases with it; the driver reports it and goes on.
SWEEP_SPONGE=1 (geometry axi): every case with the viscosity-multiplier function on."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(geo, nsp, ambi):
    import numpy as np

    from parity_util import RHS_RTOL, hip_mult, oracle_mult
    from tps_amd import capi, cases, meshgen
    from tps_amd.rhs_operator import node_coordinates

    levels = nsp - 3
    nbad = ncase = 0
    for two_t in (False, True):
        trs = [capi.CONSTANT] + ([capi.ARGON_MIXTURE] if nsp <= 7 else []) + ([capi.ARGON_MINIMAL] if nsp == 3 else [])
        for tr in trs:
            for order in (1, 2, 3, 4, 5):
                for nc in ((0, 1) if (geo != "axi" and order <= 3) else (0,)):
                    if nsp == 3:
                        ph = capi.argon_ternary_physics(capi.NS, two_t, tr, "arrhenius", ambipolar=ambi, third_order_ke=(tr != capi.CONSTANT))
                    else:
                        ph = capi.argon_levels_physics(levels, ambi, capi.NS, tr, two_t, True, third_order_ke=False)
                    ph.gas_transport.multiply = 1
                    for k in range(4):
                        ph.gas_transport.flux_trns_multiplier[k] = 30.0
                    if ph.gas_transport.third_order_k_electron:  # ill-conditioned in double precision (tests/test_gpu_parity.py
                        ph.gas_transport.flux_trns_multiplier[3] = 1.0  # ::_boost_transport): not amplified
                    ph.gas_transport.diff_mult = ph.gas_transport.mobil_mult = 30.0
                    small = order >= 4
                    if geo == "3d":
                        c = cases.argon_cyl3d(2, 8, 3, order, physics=ph, wall_type=capi.VISC_ISOTH)
                    elif geo == "axi":
                        c = cases.argon_axisym(3 if small else 5, 3 if small else 6, order, physics=ph, r_in=0.0)
                    else:
                        c = cases.Case("box2d", meshgen.scramble_orientations(meshgen.box_quad(4, 3 if small else 5, lengths=(0.2, 0.1), warp=0.08), 7),
                                       capi.Disc(order, 0, 0, 0, 0), ph, [])
                    if geo == "axi" and os.environ.get("SWEEP_SPONGE"):  # the viscous sponge of the 2-D heavy kernels, in every instantiation
                        vs = ph.visc_sponge
                        vs.enabled, vs.width, vs.ratio = 1, 0.04, 15.0
                        vs.normal[0], vs.normal[1], vs.point[0], vs.point[1] = 0.3, 1.0, 0.02, 0.12
                    disc = capi.Disc(order, nc, nc, 1 if geo == "axi" else 0, 0)
                    amp = 0.005 if order == 1 else 0.01
                    nvel = 2 if geo == "2d" else 3
                    tag = f"{geo} nsp={nsp} ambi={int(ambi)} 2T={int(two_t)} tr={tr} p={order} nc={nc}"
                    try:
                        for attempt in range(3):  # a state the oracle accepts (coarse meshes extrapolate to negative densities)
                            U = cases.plasma_state(node_coordinates(c.mesh, order, nc), ph, nvel=nvel, seed=11, amp=amp,
                                                   vel0=(1.0, 20.0, 3.0) if geo == "axi" else (20.0, 0.0, 0.0))
                            ref = oracle_mult(c.mesh, disc, ph, c.bcs, U)
                            if np.isfinite(ref["y"]).all():
                                break
                            amp *= 0.3
                        tag += f" neq={U.shape[0]}"
                        if not np.isfinite(ref["y"]).all():
                            print(f"{tag}: oracle not finite, skipped", flush=True)
                            continue
                        got = hip_mult(c.mesh, disc, ph, c.bcs, U)
                    except Exception as exc:  # noqa: BLE001
                        print(f"{tag}: EXCEPTION {type(exc).__name__}: {str(exc)[:150]}", flush=True)
                        continue
                    sc = np.abs(ref["y"]).reshape(U.shape[0], -1).max(axis=1)
                    sc[1:1 + nvel] = sc[1:1 + nvel].max()
                    err = (np.abs(got["y"] - ref["y"]).reshape(U.shape[0], -1).max(axis=1) / np.maximum(sc, 1e-300))
                    tol = RHS_RTOL * 0.05 / amp * max(1.0, (order / 3.0) ** 2) * (2.0 if nc else 1.0)
                    ncase += 1
                    if not (err.max() < tol):
                        nbad += 1
                        print(f"{tag}: WRONG err per eq {np.array2string(err, precision=1)} (tol {tol:.0e})", flush=True)
                    else:
                        print(f"{tag}: ok {err.max():.1e}", flush=True)
    print(f"worker {geo} nsp={nsp} ambi={int(ambi)}: {ncase} cases, {nbad} wrong", flush=True)
    return 1 if nbad else 0


if __name__ == "__main__":
    if sys.argv[1] == "--worker":
        sys.exit(worker(sys.argv[2], int(sys.argv[3]), bool(int(sys.argv[4]))))
    geo = sys.argv[1]
    nsps = [int(a) for a in sys.argv[2:]] or [3, 4, 5, 6, 7, 8]
    t0 = time.time()
    summary = []
    for nsp in nsps:
        for ambi in (1, 0):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", geo, str(nsp), str(ambi)])
            summary.append((nsp, ambi, r.returncode))
            print(f"== {geo} nsp={nsp} ambi={ambi}: worker exit code {r.returncode}   [{time.time() - t0:.0f} s]", flush=True)
    print("SUMMARY (exit 0 = all within tolerance, 1 = wrong results, other = the worker died):", summary)
