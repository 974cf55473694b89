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
        if self.physics.working_fluid == capi.LTE_FLUID:
            return lte_state(X, self.physics, seed=seed, amp=amp)
        if self.physics.working_fluid == capi.USER_DEFINED:
            if self.disc.axisymmetric:
                return plasma_state(X, self.physics, nvel=3, seed=seed, amp=amp, vel0=(1.0, 20.0, 3.0))
            return plasma_state(X, self.physics, nvel=X.shape[0], seed=seed, amp=amp)
        if self.disc.axisymmetric:
            return dry_air_state(X, seed=seed, amp=amp, vel0=(1.0, 20.0, 3.0), nvel=3)
        return dry_air_state(X, seed=seed, amp=amp)


def dry_air_state(X, seed=12345, amp=0.05, rho0=1.2, vel0=(20.0, 0.0, 0.0), p0=101300.0, gamma=1.4, nvel=None):
    """Conserved state (neq, NDofs): free stream with `amp` smooth sinusoidal perturbations of every
    primitive (seeded phases / wave vectors, functions of the physical coordinates only)."""
    rng = np.random.Generator(np.random.MT19937(seed))
    dim = X.shape[0]
    L = np.maximum(X.max(axis=1) - X.min(axis=1), 1e-12)

    def wave():
        k = rng.integers(1, 4, size=dim) * 2.0 * np.pi / L
        ph = rng.uniform(0.0, 2.0 * np.pi)
        return np.sin(np.tensordot(k, X, axes=(0, 0)) + ph)

    nvel = nvel or dim
    rho = rho0 * (1.0 + amp * wave())
    vref = max(abs(v) for v in vel0[:nvel]) or 1.0
    vel = [vel0[d] + amp * vref * wave() for d in range(nvel)]
    p = p0 * (1.0 + amp * wave())
    U = np.zeros((nvel + 2, X.shape[1]))
    U[0] = rho
    ke = 0.0
    for d in range(nvel):
        U[1 + d] = rho * vel[d]
        ke = ke + 0.5 * rho * vel[d] ** 2
    U[nvel + 1] = p / (gamma - 1.0) + ke
    return U


def _waves(X, seed, kmax=3):
    rng = np.random.Generator(np.random.MT19937(seed))
    dim = X.shape[0]
    L = np.maximum(X.max(axis=1) - X.min(axis=1), 1e-12)

    def wave():
        k = rng.integers(1, kmax + 1, size=dim) * 2.0 * np.pi / L
        ph = rng.uniform(0.0, 2.0 * np.pi)
        return np.sin(np.tensordot(k, X, axes=(0, 0)) + ph)

    return wave


def plasma_conserved(physics, nvel, rho, vel, Th, n_active, Te=None):
    """PerfectMixture::GetConservativesFromPrimitives (src/equation_of_state.cpp:744-783) on arrays, for
    the ambipolar mixtures of this package: builds INPUT states only (the kernels and the oracle each
    have their own closures)."""
    mx = physics.mixture
    nsp = mx.num_species
    nact = nsp - 2 if mx.ambipolar else nsp - 1
    R = capi.UNIVERSALGASCONSTANT
    mw = [mx.gas_params[sp + capi.SPECIES_MW * nsp] for sp in range(nsp)]
    q = [mx.gas_params[sp + capi.SPECIES_CHARGES * nsp] for sp in range(nsp)]
    ef = [mx.gas_params[sp + capi.FORMATION_ENERGY * nsp] for sp in range(nsp)]
    cv = [mx.molar_cv[sp] * R for sp in range(nsp)]
    two_t = bool(mx.two_temperature)
    neq = nvel + 2 + nact + (1 if two_t else 0)
    U = np.zeros((neq,) + np.shape(rho))
    U[0] = rho
    ke = 0.0
    for d in range(nvel):
        U[1 + d] = rho * vel[d]
        ke = ke + 0.5 * rho * vel[d] ** 2
    rhoB = np.array(rho, dtype=float)
    ne = 0.0
    for sp in range(nact):
        U[nvel + 2 + sp] = n_active[sp] * mw[sp]
        rhoB = rhoB - n_active[sp] * mw[sp]
        ne = ne + q[sp] * n_active[sp]
    if mx.ambipolar:
        rhoB = rhoB - ne * mw[nsp - 2]
    else:
        ne = n_active[nsp - 2]
    nB = rhoB / mw[nsp - 1]
    assert np.all(nB > 0)
    ch = nB * cv[nsp - 1]
    for sp in range(nact):
        if sp != nsp - 2:
            ch = ch + n_active[sp] * cv[sp]
    e = ke + ch * Th
    Te = Th if Te is None else Te
    ee = ne * cv[nsp - 2] * Te
    e = e + ee
    if two_t:
        U[neq - 1] = ee
    for sp in range(nsp - 2):
        e = e + n_active[sp] * ef[sp]
    U[nvel + 1] = e
    return U


def _active_densities(physics, nh, ni, wave=None):
    """number densities of the active species in mixture order for the argon mixtures of capi: the ion
    first, then (six-species mixture) the excited neutrals at small smooth fractions, then -- when the
    mixture is not ambipolar -- the electrons (quasi-neutral up to a smooth 1 % offset)."""
    mx = physics.mixture
    nsp = mx.num_species
    w = wave if wave is not None else (lambda: 0.0)
    n = [ni]
    for k in range(nsp - 3):  # excited levels
        n.append(nh * 10.0 ** (-5.5 - 0.5 * k) * (1.0 + 0.3 * w()))
    if not mx.ambipolar:
        n.append(ni * (1.0 + 0.01 * w()))
    return n


def plasma_state(X, physics, nvel, seed=12345, amp=0.05, p0=101300.0, vel0=(20.0, 0.0, 0.0)):
    """Smooth argon-plasma state: T_h in [3000, 12000] K, ionisation degree in
    [1e-6, 1e-2] (log-uniform waves), pressure p0(1 + amp wave), velocity free stream + amp waves,
    T_e = T_h (1 + 0.3 + 0.2 wave) for two-temperature mixtures."""
    wave = _waves(X, seed, kmax=1)  # species densities must stay positive under interpolation
    R = capi.UNIVERSALGASCONSTANT
    mx = physics.mixture
    nsp = mx.num_species
    mw = [mx.gas_params[sp + capi.SPECIES_MW * nsp] for sp in range(nsp)]
    scale = min(1.0, amp / 0.05)
    Th = 7500.0 + 4500.0 * scale * wave()
    alpha = 10.0 ** (-4.0 + 2.0 * scale * wave())
    p = p0 * (1.0 + amp * wave())
    Te = Th * (1.3 + 0.2 * wave()) if mx.two_temperature else None
    vel = [vel0[d] + amp * 20.0 * wave() for d in range(nvel)]
    # p = R (n_h T_h + n_e T_e), n_e = n_i = alpha n_h'  with n_h = n_i + n_B
    TeE = Th if Te is None else Te
    nh = p / (R * (Th + alpha * TeE))
    ni = alpha * nh
    if nsp == 3 and mx.ambipolar:
        nact = [ni]
    else:
        sw = wave if amp > 0 else None
        nact = _active_densities(physics, nh, ni, sw)
    ne = ni if mx.ambipolar else nact[nsp - 2]
    nheavy_active = sum(nact[: nsp - 2])
    rho = sum(nact[sp] * mw[sp] for sp in range(nsp - 2)) + ne * mw[nsp - 2] + (nh - nheavy_active) * mw[nsp - 1]
    return plasma_conserved(physics, nvel, rho, vel, Th, nact, Te)


def argon_inlet_state(physics, nvel, T=6000.0, alpha=1.0e-4, p0=101300.0, vel0=(20.0, 0.0, 0.0)):
    """rho, u, v, w, rho Y_active of a uniform argon free stream: the inlet data of SUB_DENS_VEL."""
    R = capi.UNIVERSALGASCONSTANT
    mx = physics.mixture
    nsp = mx.num_species
    mw = [mx.gas_params[sp + capi.SPECIES_MW * nsp] for sp in range(nsp)]
    nh = p0 / (R * T * (1.0 + alpha))
    ni = alpha * nh
    nact = [ni] if (nsp == 3 and mx.ambipolar) else _active_densities(physics, nh, ni)
    ne = ni if mx.ambipolar else nact[nsp - 2]
    rho = sum(nact[sp] * mw[sp] for sp in range(nsp - 2)) + ne * mw[nsp - 2] + (nh - sum(nact[: nsp - 2])) * mw[nsp - 1]
    nactive = nsp - 2 if mx.ambipolar else nsp - 1
    return [rho, vel0[0], vel0[1], vel0[2]] + [nact[sp] * mw[sp] for sp in range(nactive)]


def plasma_cylinder_bcs(physics, wall_type=capi.VISC_ISOTH, t_wall=3000.0):
    inlet = capi.make_bc(1, capi.INLET, capi.SUB_DENS_VEL, argon_inlet_state(physics, 3))
    outlet = capi.make_bc(2, capi.OUTLET, capi.SUB_P, [101300.0])
    wall = capi.make_bc(3, capi.WALL, wall_type, [t_wall])
    return [inlet, outlet, wall]


def argon_cyl3d(nr, ntheta, nz, order, two_temperature=False, transport=capi.ARGON_MINIMAL, reactions="arrhenius",
                wall_type=capi.VISC_ISOTH, eq_system=capi.NS, radiation=False, name=None, physics=None):
    """O-grid cylinder in an argon plasma stream (ambipolar ternary mixture unless `physics` is given)."""
    mesh = meshgen.ogrid_cylinder(nr, ntheta, nz)
    ph = physics or capi.argon_ternary_physics(eq_system, two_temperature, transport, reactions, radiation=radiation)
    return Case(name or f"argon_cyl3d_{nr}x{ntheta}x{nz}_p{order}", mesh, capi.Disc(order, 0, 0, 0, 0), ph,
                plasma_cylinder_bcs(ph, wall_type), "O-grid cylinder, argon ternary plasma")


def argon_axisym(nr, nz, order, two_temperature=True, transport=capi.CONSTANT, reactions="arrhenius", radiation=True,
                 wall_type=capi.VISC_ISOTH, r_in=0.0, r_out=0.05, length=0.25, warp=0.0, eq_system=capi.NS, name=None,
                 physics=None):
    """Axisymmetric (r, z) tube in an argon plasma: patch 1 inlet at z = 0 (axial flow with swirl), 2 outlet at
    z = L, 3 outer wall, 4 the axis / inner boundary (inviscid wall).  The shape of BASELINE.json configs[4]."""
    attrs = {(0, 0): 4, (0, 1): 3, (1, 0): 1, (1, 1): 2}
    mesh = meshgen.box_quad(nr, nz, lengths=(r_out - r_in, length), periodic=(False, False), bdr_attr=attrs, warp=warp,
                            origin=(r_in, 0.0))
    ph = physics or capi.argon_ternary_physics(eq_system, two_temperature, transport, reactions, radiation=radiation)
    inlet = argon_inlet_state(ph, 3, vel0=(0.0, 20.0, 2.0))
    bcs = [capi.make_bc(1, capi.INLET, capi.SUB_DENS_VEL, inlet), capi.make_bc(2, capi.OUTLET, capi.SUB_P, [101300.0]),
           capi.make_bc(3, capi.WALL, wall_type, [3000.0]), capi.make_bc(4, capi.WALL, capi.INV)]
    return Case(name or f"argon_axisym_{nr}x{nz}_p{order}", mesh, capi.Disc(order, 0, 0, 1, 0), ph, bcs,
                "axisymmetric argon ternary plasma")


def dry_air_axisym(nr, nz, order, eq_system=capi.NS, wall_type=capi.VISC_ISOTH, r_in=0.0, r_out=0.05, length=0.25,
                   warp=0.0, name=None):
    """Axisymmetric pipe in dry air (the shape of the reference's pipe.axisym inputs): patch 1 inlet at z = 0,
    2 outlet, 3 outer wall, 4 axis (inviscid wall)."""
    attrs = {(0, 0): 4, (0, 1): 3, (1, 0): 1, (1, 1): 2}
    mesh = meshgen.box_quad(nr, nz, lengths=(r_out - r_in, length), periodic=(False, False), bdr_attr=attrs, warp=warp,
                            origin=(r_in, 0.0))
    bcs = [capi.make_bc(1, capi.INLET, capi.SUB_DENS_VEL, [1.2, 0.0, 20.0, 2.0]),
           capi.make_bc(2, capi.OUTLET, capi.SUB_P, [101300.0]), capi.make_bc(3, capi.WALL, wall_type, [300.0]),
           capi.make_bc(4, capi.WALL, capi.INV)]
    return Case(name or f"dry_air_axisym_{nr}x{nz}_p{order}", mesh, capi.Disc(order, 0, 0, 1, 0),
                capi.dry_air_physics(eq_system), bcs, "axisymmetric dry air")


def lte_state(X, physics, seed=12345, amp=0.05, rho0=0.255, T0=7000.0, dT=4000.0, vel0=(1.0, 20.0, 3.0)):
    """Conserved state (5, NDofs) of the table gas in the axisymmetric formulation: smooth density, velocity (r, z, theta)
    and a temperature field T0 + dT * wave crossing many table intervals; rho e from the energy table (an INPUT state:
    the kernels and the oracle each invert it themselves)."""
    wave = _waves(X, seed)
    t = physics.lte.energy_table
    Tt = np.ctypeslib.as_array(t.x_data, shape=(t.n_data,))
    et = np.ctypeslib.as_array(t.f_data, shape=(t.n_data,))
    rho = rho0 * (1.0 + amp * wave())
    vref = max(abs(v) for v in vel0) or 1.0
    vel = [v + amp * vref * wave() for v in vel0]
    T = T0 + dT * wave()
    U = np.zeros((5, X.shape[1]))
    U[0] = rho
    ke = 0.0
    for d in range(3):
        U[1 + d] = rho * vel[d]
        ke = ke + 0.5 * rho * vel[d] ** 2
    U[4] = rho * np.interp(T, Tt, et) + ke
    return U


def lte_axisym(nr, nz, order, eq_system=capi.NS, wall_type=capi.VISC_ISOTH, r_in=0.0, r_out=0.05, length=0.25, warp=0.0,
               radiation=False, density="rho0p255", name=None):
    """Axisymmetric tube in the table gas (fluid = lte_table, the shape of the reference's test/inputs/plasma.lte1d.ini):
    patch 1 inlet at z = 0 (density and velocity), 2 pressure outlet, 3 outer wall, 4 axis (inviscid wall)."""
    attrs = {(0, 0): 4, (0, 1): 3, (1, 0): 1, (1, 1): 2}
    mesh = meshgen.box_quad(nr, nz, lengths=(r_out - r_in, length), periodic=(False, False), bdr_attr=attrs, warp=warp,
                            origin=(r_in, 0.0))
    bcs = [capi.make_bc(1, capi.INLET, capi.SUB_DENS_VEL, [0.26, 0.0, 20.0, 2.0]),
           capi.make_bc(2, capi.OUTLET, capi.SUB_P, [3.6e5]), capi.make_bc(3, capi.WALL, wall_type, [3000.0]),
           capi.make_bc(4, capi.WALL, capi.INV)]
    return Case(name or f"lte_axisym_{nr}x{nz}_p{order}", mesh, capi.Disc(order, 0, 0, 1, 0),
                capi.lte_physics(eq_system, density, radiation), bcs, "axisymmetric table gas (LTE)")


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
    if i == 3:
        return argon_cyl3d(28, 112, 16, 2, name="cfg3_argon_minimal_p2")
    if i == 5:
        return argon_axisym(400, 500, 3, name="cfg5_plasma_axisym_p3")
    if i == 4:
        return cyl3d(56, 224, 32, 3, capi.NS, name="cfg4_cyl3d_ns_p3_8gpu")
    raise NotImplementedError(f"configuration {i} is not built yet")
