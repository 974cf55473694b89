"""EXHAUSTIVE parity sweep over the plasma kernel instantiations (run on a GPU box): every (geometry, species count,
ambipolar, two-temperature, transport model, polynomial order, basis / rule pair) the library builds, one small case
each, against the oracle.  Complements the randomised tools/fuzz_parity.py: round 3 met an instantiation with a wrong
species residual in some lanes (and one with a memory fault) that no test had touched.

    python tools/sweep_instantiations.py <geo: 3d|2d|axi> [nsp ...]      # driver: one worker process per (nsp, ambipolar)
    python tools/sweep_instantiations.py single [flavour ...]            # the single-fluid flavours: dry / dry_nr / dry_les / dry_axi / lte
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


def worker_single(flavour):
    """The single-fluid kernel flavours (round 4): dry air plain / with non-reflecting patches / with a sub-grid scale model
    and the viscous sponge, in 2-D and 3-D at every order and (plain) on both basis pairs; axisymmetric dry air and the
    table gas at p = 1 ... 4: one small case per (flavour, dimension, order, pair), two consecutive Mult calls (the
    non-reflecting boundary state advances between them)."""
    import numpy as np
    import torch

    from oracle_lib import Oracle
    from parity_util import RHS_RTOL, rel_maxnorm
    from tps_amd import capi, cases, meshgen
    from tps_amd.rhs_operator import RHSoperator, node_coordinates

    def nr(attr, cat, typ, data, tangent, area=0.0):
        return capi.make_bc(attr, cat, typ, list(data) + [0.0] * (4 - len(data)) + list(tangent) + [area])

    todo = []
    if flavour in ("dry", "dry_nr", "dry_les"):
        for dim in (2, 3):
            for order in (1, 2, 3, 4, 5):
                for nc in ((0, 1) if (flavour == "dry" and order <= 3) else (0,)):
                    todo.append((dim, order, nc))
    else:
        todo = [(2, order, 0) for order in (1, 2, 3, 4)]
    nbad = ncase = 0
    for dim, order, nc in todo:
        small = order >= 4
        tag = f"{flavour} dim={dim} p={order} nc={nc}"
        try:
            if flavour == "dry_axi":
                c = cases.dry_air_axisym(3 if small else 5, 3 if small else 6, order, r_in=0.0)
                c.physics.dry_air.visc_mult = 300.0
                mesh, disc, ph, bcs, U = c.mesh, c.disc, c.physics, c.bcs, c.state(seed=11, amp=0.02)
            elif flavour == "lte":
                c = cases.lte_axisym(3 if small else 5, 3 if small else 6, order, r_in=0.0, radiation=True)
                mesh, disc, ph, bcs, U = c.mesh, c.disc, c.physics, c.bcs, c.state(seed=11, amp=0.02)
            else:
                ph = capi.dry_air_physics(capi.NS, visc_mult=300.0, bulk_visc_mult=0.4)
                if dim == 3:
                    c = cases.cyl3d(2, 8, 3, order, capi.NS, capi.VISC_ISOTH)
                    mesh, bcs = meshgen.scramble_orientations(c.mesh, 9), c.bcs
                    tangent = (0.0, 0.0, 1.0)
                else:
                    attrs = {(0, 0): 1, (0, 1): 2, (1, 0): 3, (1, 1): 3}
                    mesh = meshgen.scramble_orientations(meshgen.box_quad(4, 3 if small else 4, lengths=(1.0, 0.7), periodic=(False, False),
                                                                          bdr_attr=attrs, warp=0.06), 5)
                    bcs = [capi.make_bc(1, capi.INLET, capi.SUB_DENS_VEL, [1.2, 20.0, 0.0, 0.0]), capi.make_bc(2, capi.OUTLET, capi.SUB_P, [101300.0]),
                           capi.make_bc(3, capi.WALL, capi.VISC_ADIAB)]
                    tangent = (0.0, 1.0, 0.0)
                disc = capi.Disc(order, nc, nc, 0, 0)
                disc.ref_length = 0.7
                if flavour == "dry_nr":
                    bcs = list(bcs)
                    bcs[0] = nr(1, capi.INLET, capi.SUB_DENS_VEL_NR, [1.2, 20.0, 0.0, 0.0], tangent)
                    bcs[1] = nr(2, capi.OUTLET, capi.SUB_MF_NR if order % 2 else capi.SUB_P_NR, [1.2 * 20.0 * 0.7] if order % 2 else [101250.0], tangent, area=0.7)
                if flavour == "dry_les":
                    if dim == 3:
                        ph.sgs.model_type = capi.SGS_SIGMA if order % 2 else capi.SGS_SMAGORINSKY
                    vs = ph.visc_sponge
                    vs.enabled, vs.width, vs.ratio = 1, 0.3, 12.0
                    vs.normal[0], vs.normal[1], vs.point[0], vs.point[1] = 1.0, 0.4, 0.5, 0.2
                U = cases.dry_air_state(node_coordinates(mesh, order, nc), seed=6, amp=0.02 if flavour != "dry_les" else 0.05, nvel=dim)
            o = Oracle(mesh, disc, ph, bcs)
            op = RHSoperator(mesh, disc, ph, bcs)
            if flavour == "dry_nr":
                o.set_dt(1.0e-4)
                op.setDt(1.0e-4)
            err = 0.0
            for call in range(2):
                Uc = U * (1.0 + 1e-3 * call)
                ref = o.mult(Uc)
                x = torch.tensor(np.ascontiguousarray(Uc).ravel(), dtype=torch.float64, device=op.device)
                y = torch.empty_like(x)
                op.Mult(x, y)
                err = max(err, rel_maxnorm(y.cpu().numpy().reshape(U.shape), ref).max())
            op.close()
        except Exception as exc:  # noqa: BLE001
            print(f"{tag}: EXCEPTION {type(exc).__name__}: {str(exc)[:200]}", flush=True)
            continue
        # (2 % perturbations; the sigma model's eigenvalue route amplifies gradient rounding, src/fluxes.cpp:584-587)
        tol = RHS_RTOL * 2.5 * max(1.0, (order / 3.0) ** 2) * (2.0 if nc else 1.0) * (5.0 if flavour == "dry_les" else 1.0)
        ncase += 1
        if not (err < tol):
            nbad += 1
            print(f"{tag}: WRONG err {err:.2e} (tol {tol:.0e})", flush=True)
        else:
            print(f"{tag}: ok {err:.1e}", flush=True)
    print(f"worker {flavour}: {ncase} cases, {nbad} wrong", flush=True)
    return 1 if nbad else 0


SINGLE_FLAVOURS = ("dry", "dry_nr", "dry_les", "dry_axi", "lte")

if __name__ == "__main__":
    if sys.argv[1] == "--worker":
        sys.exit(worker(sys.argv[2], int(sys.argv[3]), bool(int(sys.argv[4]))))
    if sys.argv[1] == "--worker-single":
        sys.exit(worker_single(sys.argv[2]))
    if sys.argv[1] == "single":  # the single-fluid flavours: one worker process each
        t0 = time.time()
        summary = []
        for fl in sys.argv[2:] or SINGLE_FLAVOURS:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker-single", fl])
            summary.append((fl, r.returncode))
            print(f"== single {fl}: worker exit code {r.returncode}   [{time.time() - t0:.0f} s]", flush=True)
        print("SUMMARY (exit 0 = all within tolerance, 1 = wrong results, other = the worker died):", summary)
        sys.exit(0)
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
