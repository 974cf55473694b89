"""Regenerates the fixtures of this directory with the CPU oracle (run from the repository root:
``python tests/golden/make_golden.py``).  Each ``.npz`` holds the inputs that are not rebuilt from code
(the state) and the oracle's outputs (y, gradUp, max_char_speed) of one small case; the case itself is
rebuilt by ``golden_case(name)`` below, so that the HIP kernels can be compared with these vectors without
the oracle library, and the oracle against its own earlier output (drift check).

The reference itself cannot produce vectors here (no MFEM, LFS pointers), so these pin the
restatement, not pecos/tps; what pins the restatement to the reference is in tests/test_oracle_pins.py and
tests/test_oracle_plasma.py."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from tps_amd import capi, cases, meshgen  # noqa: E402
from tps_amd.rhs_operator import node_coordinates  # noqa: E402

NAMES = ["cyl3d_ns_p2", "box2d_euler_p3", "argon_2T_cyl3d_p1", "argon_axisym_p2", "dryair_axisym_p3"]


def golden_case(name):
    """-> mesh, disc, physics, bcs, state-builder"""
    if name == "cyl3d_ns_p2":
        c = cases.cyl3d(3, 8, 3, 2, capi.NS, capi.VISC_ISOTH)
        c.physics.dry_air.visc_mult = 500.0
        return c.mesh, c.disc, c.physics, c.bcs, lambda: c.state(seed=11)
    if name == "box2d_euler_p3":
        mesh = meshgen.scramble_orientations(meshgen.box_quad(4, 3, warp=0.1), 3)
        return (mesh, capi.Disc(3, 0, 0, 0, 0), capi.dry_air_physics(capi.EULER), [],
                lambda: cases.dry_air_state(node_coordinates(mesh, 3), seed=12))
    if name == "argon_2T_cyl3d_p1":
        c = cases.argon_cyl3d(4, 8, 3, 1, True, capi.ARGON_MIXTURE, "arrhenius", capi.VISC_GNRL)
        c.bcs[2] = capi.make_bc(3, capi.WALL, capi.VISC_GNRL, [3000.0, 9000.0, capi.ISOTH, capi.SHTH])
        return c.mesh, c.disc, c.physics, c.bcs, lambda: c.state(seed=13, amp=0.005)
    if name == "argon_axisym_p2":
        c = cases.argon_axisym(4, 6, 2, True, capi.CONSTANT, "tabulated", True, capi.VISC_ISOTH, r_in=0.0)
        return c.mesh, c.disc, c.physics, c.bcs, lambda: c.state(seed=14, amp=0.01)
    if name == "dryair_axisym_p3":
        c = cases.dry_air_axisym(4, 5, 3, capi.NS, capi.VISC_ADIAB, r_in=0.01, warp=0.05)
        c.physics.dry_air.visc_mult = 200.0
        return c.mesh, c.disc, c.physics, c.bcs, lambda: c.state(seed=15)
    raise KeyError(name)


if __name__ == "__main__":
    from parity_util import oracle_mult

    for name in NAMES:
        mesh, disc, ph, bcs, state = golden_case(name)
        U = state()
        r = oracle_mult(mesh, disc, ph, bcs, U)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), U=U, y=r["y"], gradUp=r["gradUp"],
                            max_char_speed=np.float64(r["max_char_speed"]))
        print(name, U.shape, "finite:", bool(np.isfinite(r["y"]).all()))
