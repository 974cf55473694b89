"""The BASELINE.json configurations as synthetic inputs.

A case = mesh + discretisation + physics parameter blocks + boundary conditions + a smooth seeded
state.  The free stream and patches follow ``test/inputs/input.4iters.cyl.ini:24-46`` of the
reference: rho = 1.2, u = (20, 0, 0), p = 101300; patch 1 inlet (SUB_DENS_VEL), patch 2 outlet
(SUB_P), patch 3 cylinder wall.
"""
from __future__ import annotations

import dataclasses

import numpy as np

from . import capi, meshgen
from .rhs_operator import node_coordinates


@dataclasses.dataclass
class Case:
    name: str
    mesh: meshgen.HostMesh
    disc: capi.Disc
    physics: capi.Physics
    bcs: list
    description: str = ""

    def state(self, seed=12345, amp=0.05, coords=None):
        X = node_coordinates(self.mesh, self.disc.order) if coords is None else coords
        return dry_air_state(X, seed=seed, amp=amp)


def dry_air_state(X, seed=12345, amp=0.05, rho0=1.2, vel0=(20.0, 0.0, 0.0), p0=101300.0, gamma=1.4):
    """Conserved state (neq, NDofs): free stream with `amp` smooth sinusoidal perturbations of every
    primitive (seeded phases / wave vectors, functions of the physical coordinates only)."""
    rng = np.random.Generator(np.random.MT19937(seed))
    dim = X.shape[0]
    L = np.maximum(X.max(axis=1) - X.min(axis=1), 1e-12)

    def wave():
        k = rng.integers(1, 4, size=dim) * 2.0 * np.pi / L
        ph = rng.uniform(0.0, 2.0 * np.pi)
        return np.sin(np.tensordot(k, X, axes=(0, 0)) + ph)

    rho = rho0 * (1.0 + amp * wave())
    vref = max(abs(v) for v in vel0[:dim]) or 1.0
    vel = [vel0[d] + amp * vref * wave() for d in range(dim)]
    p = p0 * (1.0 + amp * wave())
    U = np.zeros((dim + 2, X.shape[1]))
    U[0] = rho
    ke = 0.0
    for d in range(dim):
        U[1 + d] = rho * vel[d]
        ke = ke + 0.5 * rho * vel[d] ** 2
    U[dim + 1] = p / (gamma - 1.0) + ke
    return U


def cylinder_bcs(wall_type=capi.VISC_ISOTH, t_wall=300.0, dim=3):
    inlet = capi.make_bc(1, capi.INLET, capi.SUB_DENS_VEL, [1.2, 20.0, 0.0, 0.0])
    outlet = capi.make_bc(2, capi.OUTLET, capi.SUB_P, [101300.0])
    wall = capi.make_bc(3, capi.WALL, wall_type, [t_wall])
    return [inlet, outlet, wall]


def cyl3d(nr, ntheta, nz, order, eq_system=capi.NS, wall_type=None, name=None):
    mesh = meshgen.ogrid_cylinder(nr, ntheta, nz)
    if wall_type is None:
        wall_type = capi.INV if eq_system == capi.EULER else capi.VISC_ISOTH
    return Case(name or f"cyl3d_{nr}x{ntheta}x{nz}_p{order}", mesh, capi.Disc(order, 0, 0, 0, 0),
                capi.dry_air_physics(eq_system), cylinder_bcs(wall_type),
                "O-grid cylinder, dry air, " + ("Euler" if eq_system == capi.EULER else "Navier-Stokes"))


def config(i: int) -> Case:
    """BASELINE.json ``configs[i-1]`` (1-based)."""
    if i == 1:
        return cyl3d(10, 24, 8, 1, capi.EULER, name="cfg1_cyl3d_euler_p1")
    if i == 2:
        return cyl3d(28, 112, 16, 3, capi.NS, name="cfg2_cyl3d_ns_p3")
    if i == 4:
        return cyl3d(56, 224, 32, 3, capi.NS, name="cfg4_cyl3d_ns_p3_8gpu")
    raise NotImplementedError(f"configuration {i} is not built yet")
