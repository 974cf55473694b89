"""Probe: 3-D Gauss-Lobatto p = 1 (4 elements per wave) for every mixture family -- which instantiations are wrong?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from parity_util import hip_mult, oracle_mult
from tps_amd import capi, cases, meshgen
from tps_amd.rhs_operator import node_coordinates
order = int(sys.argv[1]) if len(sys.argv) > 1 else 1
gll = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mesh = meshgen.scramble_orientations(meshgen.box_hex(3, 4, 3, warp=0.1), 254)
for levels in range(0, 6):
    for ambi in (True, False):
        for two_t in (False, True):
            for tr in ((capi.CONSTANT, capi.ARGON_MIXTURE) if levels < 5 else (capi.CONSTANT,)):
                ph = capi.argon_levels_physics(levels, ambi, capi.NS, tr, two_t, True, third_order_ke=False)
                disc = capi.Disc(order, gll, gll, 0, 0)
                U = cases.plasma_state(node_coordinates(mesh, order, gll), ph, nvel=3, seed=254, amp=0.005)
                ref = oracle_mult(mesh, disc, ph, [], U)
                got = hip_mult(mesh, disc, ph, [], U)
                sc = np.abs(ref["y"]).reshape(U.shape[0], -1).max(axis=1)
                err = np.abs(got["y"] - ref["y"]).reshape(U.shape[0], -1).max(axis=1) / sc
                print(f"nsp={3+levels} ambi={ambi} 2T={two_t} tr={tr} neq={U.shape[0]}: max err {err.max():.2e}", "WRONG rows " + str(np.nonzero(err > 1e-9)[0].tolist()) if err.max() > 1e-9 else "", flush=True)
