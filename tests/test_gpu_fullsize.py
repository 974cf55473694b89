"""BASELINE.json's full sizes on the GPU, checked through size-independent properties.

The oracle needs minutes for 50 176 hexes, so the full-size runs are pinned to it indirectly:
  * spanwise invariance -- the O-grid is periodic and extruded in z; for a state that does not depend
    on z every one of the 16 element layers must reproduce, node for node, the result of a 3-layer
    mesh of the same cross-section, and THAT mesh is compared with the oracle;
  * free-stream preservation -- on a warped periodic box of the same node count a uniform state has a
    zero residual (the discrete metric identities hold for trilinear hexes at these quadrature orders),
    which exercises every face record, orientation code and geometry recomputation at scale.
"""
import numpy as np
import pytest

from parity_util import RHS_RTOL, hip_mult, oracle_mult, rel_maxnorm
from tps_amd import capi, cases, meshgen
from tps_amd.rhs_operator import node_coordinates

pytestmark = pytest.mark.gpu

NR, NTHETA, NZ = 28, 112, 16  # the mesh of BASELINE.json configs[1] and [2]


def _extruded_case(kind, order, nz):
    mesh = meshgen.ogrid_cylinder(NR, NTHETA, nz, span=2.0 * nz / NZ)
    disc = capi.Disc(order, 0, 0, 0, 0)
    X = node_coordinates(mesh, order)
    X[2] = 0.0  # the state is a function of (x, y) only
    if kind == "dry_air":
        ph = capi.dry_air_physics(capi.NS)
        bcs = cases.cylinder_bcs(capi.VISC_ISOTH, 300.0)
        U = cases.dry_air_state(X, seed=12345)
    else:
        ph = capi.argon_ternary_physics(capi.NS, False, capi.ARGON_MINIMAL, "arrhenius")
        bcs = cases.plasma_cylinder_bcs(ph, capi.VISC_ISOTH, 3000.0)
        U = cases.plasma_state(X, ph, nvel=3, seed=12345, amp=0.05)
    return mesh, disc, ph, bcs, U


@pytest.mark.parametrize("kind,order", [("dry_air", 3), ("argon", 3), ("argon", 2)])
def test_full_size_layers_reproduce_the_oracle_checked_slab(kind, order):
    """cfg2 (dry air p=3), the headline workload (argon p=3) and cfg3 (argon p=2) at 50 176 hexes."""
    mesh, disc, ph, bcs, U = _extruded_case(kind, order, NZ)
    full = hip_mult(mesh, disc, ph, bcs, U, want_grad=False)
    smesh, sdisc, sph, sbcs, sU = _extruded_case(kind, order, 3)
    small = hip_mult(smesh, sdisc, sph, sbcs, sU, want_grad=False)
    ref = oracle_mult(smesh, sdisc, sph, sbcs, sU)
    # the spanwise momentum residual vanishes by symmetry up to the rounding of the pressure terms, so
    # the three momentum rows are measured against their common scale
    scale = np.abs(ref["y"]).max(axis=1, keepdims=True)
    scale[1:4] = scale[1:4].max()
    err = (np.abs(small["y"] - ref["y"]) / scale).max(axis=1)
    print("3-layer slab vs oracle:", err)
    assert err.max() < RHS_RTOL
    per_layer = NR * NTHETA * (order + 1) ** 3
    assert (np.abs(sU[:, :per_layer] - U[:, :per_layer]) <= 1e-13 * np.abs(U).max(axis=1, keepdims=True)).all()
    y0 = small["y"][:, :per_layer]
    for k in range(NZ):
        yk = full["y"][:, k * per_layer:(k + 1) * per_layer]
        # same arithmetic in the same order; only the vertex z-coordinates differ by the layer offset
        assert (np.abs(yk - y0) <= RHS_RTOL * scale).all(), k
    assert abs(full["max_char_speed"] - ref["max_char_speed"]) < 1e-12 * ref["max_char_speed"]


@pytest.mark.parametrize("kind", ["dry_air", "argon"])
def test_free_stream_preservation_at_full_size(kind):
    n = 37  # 37^3 = 50 653 hexes, p = 3: 3.24 M nodes
    mesh = meshgen.scramble_orientations(meshgen.box_hex(n, n, n, lengths=(1.0, 0.8, 1.2), warp=0.08), 3)
    disc = capi.Disc(3, 0, 0, 0, 0)
    X = node_coordinates(mesh, 3)
    if kind == "dry_air":
        ph = capi.dry_air_physics(capi.NS)
        U = cases.dry_air_state(X, seed=1, amp=0.0)
    else:
        ph = capi.argon_ternary_physics(capi.NS, False, capi.ARGON_MINIMAL, reactions=None)  # no sources
        U = cases.plasma_state(X, ph, nvel=3, seed=1, amp=0.0)
    got = hip_mult(mesh, disc, ph, [], U, want_grad=True)
    assert np.isfinite(got["y"]).all()
    # the residual of a uniform state is rounding noise of the flux divergence: |F| / h * eps
    h = 1.0 / n / 4
    flux_scale = np.abs(U).max(axis=1) * 20.0 + 101300.0
    bound = 1e-12 * flux_scale / h
    resid = np.abs(got["y"]).max(axis=1)
    print("residual of the uniform state per equation:", resid, "bound", bound)
    assert (resid <= bound).all()
    assert np.abs(got["gradUp"]).max() <= 1e-10 * np.abs(got["Up"]).max() / h


# ---- BASELINE.json configs[4] (cfg5 of bench.py) and torch6 at their full size: 400 x 500 axisymmetric quads, p = 3 ----
def _bench_axisym(wname, nz, length):
    """mesh, disc, physics, bcs and a state that depends on r only, built with bench.py's own workload definition"""
    import bench

    order, ph, make_bcs, make_state, _, _ = bench.workload(wname)
    if ph.visc_sponge.enabled:  # lte_torch ramps the viscosity along z: turned to act along r here, where the state varies
        ph.visc_sponge.normal[0], ph.visc_sponge.normal[1] = 1.0, 0.0
        ph.visc_sponge.point[0], ph.visc_sponge.point[1] = 0.03, 0.0
    mesh = meshgen.annulus_quad(400, nz, r_in=0.0, r_out=0.05, length=length)
    disc = capi.Disc(order, 0, 0, 1, 0)
    X = node_coordinates(mesh, order)
    X[1] = 0.0  # no dependence on z
    return mesh, disc, ph, make_bcs(ph), make_state(X, ph), order


@pytest.mark.parametrize("wname", ["cfg5", "cfg5_const", "torch6", "torch6_mix", "lte_torch"])
def test_full_size_axisymmetric_layers_reproduce_the_oracle_checked_strip(wname):
    """Axial-translation invariance.  The tube is extruded along z between the inlet (z = 0) and the outlet (z = L);
    for a state that depends on r only, every layer of elements whose stencil does not reach those two patches -- the
    element, its neighbours and their neighbours (BR1 gradient): layers 2 ... 497 -- must reproduce, node for node, the
    middle layer of a 7-layer strip with the same cells, and the two layers at either end the strip's end layers.
    The strip is compared with the oracle."""
    NZF, NZS, L = 500, 7, 0.25
    mesh, disc, ph, bcs, U, order = _bench_axisym(wname, NZF, L)
    full = hip_mult(mesh, disc, ph, bcs, U, want_grad=False)
    smesh, sdisc, sph, sbcs, sU, _ = _bench_axisym(wname, NZS, L * NZS / NZF)
    small = hip_mult(smesh, sdisc, sph, sbcs, sU, want_grad=False)
    ref = oracle_mult(smesh, sdisc, sph, sbcs, sU)
    scale = np.abs(ref["y"]).max(axis=1, keepdims=True)
    err = (np.abs(small["y"] - ref["y"]) / scale).max(axis=1)
    print(wname, "7-layer strip vs oracle:", err)
    assert err.max() < 5 * RHS_RTOL  # 5 % perturbations of a plasma state: the tolerance of the plasma parity cases
    per_layer = 400 * (order + 1) ** 2
    lay = lambda y, k: y[:, k * per_layer:(k + 1) * per_layer]
    assert (np.abs(lay(sU, 0) - lay(U, 0)) <= 1e-13 * np.abs(U).max(axis=1, keepdims=True)).all()
    tol = 5 * RHS_RTOL * scale
    for k in range(NZF):
        ks = k if k < 2 else (NZS - (NZF - k) if k >= NZF - 2 else 3)
        assert (np.abs(lay(full["y"], k) - lay(small["y"], ks)) <= tol).all(), (k, ks)
    assert np.isfinite(full["y"]).all()


def test_config1_at_its_own_size_against_the_oracle():
    """BASELINE.json configs[0] itself -- cyl3d perfect-gas Euler, p = 1, 10 x 24 x 8 = 1 920 hexes, the reference's
    CPU-runnable plumbing case -- directly against the oracle (which does it in under a second): y, Up, gradUp and the
    maximum characteristic speed of one Mult."""
    c = cases.config(1)
    assert c.mesh.num_elements == 1920
    U = c.state(seed=12345)
    got = hip_mult(c.mesh, c.disc, c.physics, c.bcs, U)
    ref = oracle_mult(c.mesh, c.disc, c.physics, c.bcs, U)
    err = rel_maxnorm(got["y"], ref["y"])
    print("cfg1 at 1 920 hexes: y", err, "Up", rel_maxnorm(got["Up"], ref["Up"]).max())
    assert err.max() < RHS_RTOL
    assert rel_maxnorm(got["Up"], ref["Up"]).max() < 1e-13
    g_ref = ref["gradUp"].reshape(-1, U.shape[1])
    assert np.abs(got["gradUp"].reshape(-1, U.shape[1]) - g_ref).max() <= 1e-12 * np.abs(g_ref).max()
    assert abs(got["max_char_speed"] - ref["max_char_speed"]) <= 1e-12 * ref["max_char_speed"]
