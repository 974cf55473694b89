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

NAMES = ["cyl3d_ns_p2", "box2d_euler_p3", "argon_2T_cyl3d_p1", "argon_axisym_p2", "dryair_axisym_p3",
         "cyl3d_widened_p2", "argon5_box2d_p2"]


def golden_extras(name):
    """-> None, or what the case adds to a plain Mult: dt of the non-reflecting conditions, forcing terms, and the
    number of consecutive Mult calls (the vectors are those of the LAST call: the boundary state has moved)"""
    if name == "cyl3d_widened_p2":
        tgt = [1.15, 1.15 * 18.0, 1.15 * 1.0, -1.15 * 0.5, 100900.0 / 0.4 + 0.5 * 1.15 * (18.0 ** 2 + 1.0 + 0.25)]
        forcing = capi.make_forcing(
            pressure_gradient=(3.0, -1.5, 0.7),
            heat_sources=[dict(value=7.5e4, radius=1.3, point1=(2.0, 0.1, -0.1), point2=(2.1, 0.0, 2.3))],
            sponge_zones=[dict(type=capi.SPONGE_PLANAR, normal=(-2.0, -0.2, 0.0), point0=(9.7, 0.0, 0.0),
                               point_init=(5.2, 0.0, 0.0), mult_factor=0.6, target_U=tgt)])
        return {"dt": 3.0e-4, "forcing": forcing, "ncalls": 3}
    return None


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
    if name == "cyl3d_widened_p2":  # non-reflecting outlet + inlet, forcing terms: the rows of SURVEY 8f
        c = cases.cyl3d(3, 8, 3, 2, capi.NS, capi.VISC_ADIAB)
        c.physics.dry_air.visc_mult = 500.0
        c.disc.ref_length = 2.5
        c.bcs[0] = capi.make_bc(1, capi.INLET, capi.SUB_DENS_VEL_NR, [1.21, 19.0, 0.5, -0.3, 0.0, 0.0, 1.0])
        c.bcs[1] = capi.make_bc(2, capi.OUTLET, capi.SUB_MF_NR, [720.0, 0, 0, 0, 0.0, 0.0, 1.0, 30.0])
        return c.mesh, c.disc, c.physics, c.bcs, lambda: c.state(seed=16)
    if name == "argon5_box2d_p2":  # five species with an electron equation, two temperatures, mixture transport
        ph = capi.argon_levels_physics(2, False, capi.NS, capi.ARGON_MIXTURE, True, True, third_order_ke=False)
        mesh = meshgen.scramble_orientations(meshgen.box_quad(4, 3, lengths=(1.0, 0.7), warp=0.08), 4)
        return (mesh, capi.Disc(2, 0, 0, 0, 0), ph, [],
                lambda: cases.plasma_state(node_coordinates(mesh, 2), ph, nvel=2, seed=17, amp=0.01))
    raise KeyError(name)


def oracle_run(name, U):
    """the oracle on a golden case, with its extras"""
    from oracle_lib import Oracle

    mesh, disc, ph, bcs, _ = golden_case(name)
    ex = golden_extras(name) or {}
    o = Oracle(mesh, disc, ph, bcs)
    o.set_dt(ex.get("dt", 0.0))
    o.set_forcing(ex.get("forcing"))
    for _ in range(ex.get("ncalls", 1)):
        y = o.mult(U)
    return {"y": y, "gradUp": o.gradients(), "max_char_speed": o.max_char_speed}


if __name__ == "__main__":
    for name in NAMES:
        mesh, disc, ph, bcs, state = golden_case(name)
        U = state()
        r = oracle_run(name, U)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), U=U, y=r["y"], gradUp=r["gradUp"],
                            max_char_speed=np.float64(r["max_char_speed"]))
        print(name, U.shape, "finite:", bool(np.isfinite(r["y"]).all()))
