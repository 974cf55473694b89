"""Probe (GPU): one case of tools/sweep_instantiations.py in detail -- which outputs (Up, gradUp, y), which elements /
nodes / equations differ from the oracle, is it deterministic, does NaN-poisoning change it.
    python tools/probe_sweep_case.py <geo> <nsp> <ambi> <two_t> <tr> <order> <nc>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from parity_util import hip_mult, oracle_mult
from tps_amd import capi, cases, meshgen
from tps_amd.rhs_operator import node_coordinates

geo, nsp, ambi, two_t, tr, order, nc = sys.argv[1], int(sys.argv[2]), bool(int(sys.argv[3])), bool(int(sys.argv[4])), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])


def physics(eq=capi.NS):
    if nsp == 3:
        ph = capi.argon_ternary_physics(eq, two_t, tr, "arrhenius", ambipolar=ambi, third_order_ke=(tr != capi.CONSTANT))
    else:
        ph = capi.argon_levels_physics(nsp - 3, ambi, eq, tr, two_t, True, third_order_ke=False)
    ph.gas_transport.multiply = 1
    for k in range(4):
        ph.gas_transport.flux_trns_multiplier[k] = 30.0
    if ph.gas_transport.third_order_k_electron:
        ph.gas_transport.flux_trns_multiplier[3] = 1.0
    ph.gas_transport.diff_mult = ph.gas_transport.mobil_mult = 30.0
    return ph


def run(tag, ph, mesh, bcs, axisym=False):
    disc = capi.Disc(order, nc, nc, 1 if axisym else 0, 0)
    nvel = 3 if (axisym or mesh.dim == 3) else 2
    U = cases.plasma_state(node_coordinates(mesh, order, nc), ph, nvel=nvel, seed=11, amp=0.005 if order == 1 else 0.01,
                           vel0=(1.0, 20.0, 3.0) if axisym else (20.0, 0.0, 0.0))
    ref = oracle_mult(mesh, disc, ph, bcs, U)
    npe = (order + 1) ** mesh.dim
    for rep in range(2):
        got = hip_mult(mesh, disc, ph, bcs, U)
        if os.environ.get("PROBE_DUMP") and rep == 0:  # for comparisons between two builds
            np.save(os.path.join(ROOT, "gpurun_out", f"probe_{os.environ['PROBE_DUMP']}_{tag.replace(' ', '_')}.npy"), got["y"])
        sc = np.abs(ref["y"]).reshape(U.shape[0], -1).max(axis=1)
        err = np.abs(got["y"] - ref["y"]).reshape(U.shape[0], -1).max(axis=1) / sc
        uerr = np.abs(got["Up"] - ref["Up"]).max(axis=1) / np.abs(ref["Up"]).max(axis=1)
        g_ref = ref["gradUp"].reshape(-1, U.shape[1])
        g_got = got["gradUp"].reshape(-1, U.shape[1])
        gerr = np.abs(g_got - g_ref).max() / np.abs(g_ref).max()
        bad = np.nonzero((np.abs(got["y"] - ref["y"]) / sc[:, None]).max(axis=0) > 1e-9)[0]
        gbad = np.nonzero((np.abs(g_got - g_ref) / np.abs(g_ref).max()).max(axis=0) > 1e-9)[0]
        print(f"{tag} run {rep}: neq={U.shape[0]} y err per eq {np.array2string(err, precision=1)}\n    Up {uerr.max():.1e} gradUp {gerr:.1e} "
              f"(bad grad nodes {gbad.size}, elements {sorted(set((gbad // npe).tolist()))[:16]})\n    bad y nodes {bad.size}/{U.shape[1]} "
              f"elements {sorted(set((bad // npe).tolist()))[:24]} local nodes {sorted(set((bad % npe).tolist()))}", flush=True)


if geo == "3d" and os.environ.get("PROBE_KNOBS"):
    # which transport coefficient carries the wrong term?  multipliers of [plasma_models/transport]: viscosity, bulk viscosity,
    # heavy conductivity, electron conductivity; diffusivities; mobilities
    box = meshgen.scramble_orientations(meshgen.box_hex(3, 4, 3, warp=0.1), 254)
    for name, fl, dm, mm in (("all 30", (30, 30, 30, 30), 30, 30), ("no diffusion", (30, 30, 30, 30), 0, 0), ("only diffusion", (0, 0, 0, 0), 30, 30),
                             ("only viscosity", (30, 0, 0, 0), 0, 0), ("only k_h", (0, 0, 30, 0), 0, 0), ("only k_e", (0, 0, 0, 30), 0, 0),
                             ("diffusion, no mobility", (0, 0, 0, 0), 30, 0), ("all 1", (1, 1, 1, 1), 1, 1)):
        ph = physics()
        for k in range(4):
            ph.gas_transport.flux_trns_multiplier[k] = float(fl[k])
        ph.gas_transport.diff_mult, ph.gas_transport.mobil_mult = float(dm), float(mm)
        run("box " + name, ph, box, [])
    sys.exit(0)
if geo == "3d":
    c = cases.argon_cyl3d(2, 8, 3, order, physics=physics(), wall_type=capi.VISC_ISOTH)
    run("cylinder NS", c.physics, c.mesh, c.bcs)
    run("cylinder EULER", physics(capi.EULER), c.mesh, c.bcs)
    box = meshgen.scramble_orientations(meshgen.box_hex(3, 4, 3, warp=0.1), 254)
    run("periodic box NS", physics(), box, [])
    os.environ["TPSRHS_POISON"] = "1"
    run("cylinder NS poisoned", c.physics, c.mesh, c.bcs)
