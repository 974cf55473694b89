"""Probe (GPU): the 3-D p = 1 Gauss-Lobatto kernels of the seven-species mixture (the failing pattern of round 3's sweep):
which ingredient makes the residual wrong?  Never run the instantiation that faulted (4 species, 2-T, mixture)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from parity_util import hip_mult, oracle_mult
from tps_amd import capi, cases, meshgen
from tps_amd.rhs_operator import node_coordinates
mesh = meshgen.scramble_orientations(meshgen.box_hex(3, 4, 3, warp=0.1), 254)
def run(tag, levels, ambi, two_t, tr, eq, order, gll):
    ph = capi.argon_levels_physics(levels, ambi, eq, tr, two_t, True, third_order_ke=False)
    disc = capi.Disc(order, gll, gll, 0, 0)
    U = cases.plasma_state(node_coordinates(mesh, order, gll), ph, nvel=3, seed=254, amp=0.005)
    ref = oracle_mult(mesh, disc, ph, [], U)
    got = hip_mult(mesh, disc, ph, [], U)
    sc = np.abs(ref["y"]).reshape(U.shape[0], -1).max(axis=1)
    err = np.abs(got["y"] - ref["y"]).reshape(U.shape[0], -1).max(axis=1) / sc
    gerr = np.abs(got["gradUp"] - ref["gradUp"]).max() / np.abs(ref["gradUp"]).max()
    bad_nodes = np.nonzero((np.abs(got["y"] - ref["y"]) / sc[:, None]).max(axis=0) > 1e-9)[0]
    print(f"{tag}: neq={U.shape[0]} y err per eq {np.array2string(err, precision=1)} grad {gerr:.1e} bad nodes {bad_nodes.size}/{U.shape[1]} "
          f"elements {sorted(set((bad_nodes // (order + 1) ** 3).tolist()))[:12]}", flush=True)
M, K = capi.ARGON_MIXTURE, capi.CONSTANT
run("A  nsp7 1T mixture NS    GLL p1", 4, False, False, M, capi.NS, 1, 1)
run("B  nsp7 1T mixture EULER GLL p1", 4, False, False, M, capi.EULER, 1, 1)
run("C  nsp7 1T constant NS   GLL p1", 4, False, False, K, capi.NS, 1, 1)
run("D  nsp7 1T mixture NS    GL  p1", 4, False, False, M, capi.NS, 1, 0)
run("E  nsp7 1T mixture NS    GLL p2", 4, False, False, M, capi.NS, 2, 1)
os.environ["TPSRHS_POISON"] = "1"
run("A' poisoned", 4, False, False, M, capi.NS, 1, 1)
