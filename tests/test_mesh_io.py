"""The reference's own mesh files: tps_amd.mesh_io reads MFEM's native format -- quadrilaterals,
hexahedra, periodic meshes with their discontinuous order-1 ``nodes`` -- and Mult runs on them.  Fixtures (data files of the
reference's test tree, copied verbatim): test/meshes/periodic-cube.mesh (27 periodic hexes: the mesh of test/mms.euler.test),
beam-quad-o3-s5-p.mesh (45 periodic quads: test/argon_minimal.binary.test), skinny-rectangle.mesh (30 quads, four boundary
attributes)."""
import os

import numpy as np
import pytest

from parity_util import RHS_RTOL, hip_mult, oracle_mult, rel_maxnorm
from tps_amd import capi, cases, mesh_io
from tps_amd.rhs_operator import node_coordinates

MESHES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "meshes")


def _volumes(m):
    """element volumes from the corner coordinates (MFEM vertex order), by the order-1 Gauss rule"""
    g = np.array([-1.0, 1.0]) / np.sqrt(3.0) * 0.5 + 0.5
    vol = np.zeros(m.num_elements)
    lex = [0, 1, 3, 2] if m.dim == 2 else [0, 1, 3, 2, 4, 5, 7, 6]
    X = m.elem_coords[:, lex, :].reshape((m.num_elements,) + (2,) * m.dim + (m.dim,))  # [e][(z)][y][x][d]
    pts = np.stack(np.meshgrid(*([g] * m.dim), indexing="ij"), axis=-1).reshape(-1, m.dim)
    for xi in pts:
        J = np.zeros((m.num_elements, m.dim, m.dim))
        for a in range(m.dim):  # derivative along reference axis a (a = 0: x, the LAST array axis)
            w = X
            for b in range(m.dim - 1, -1, -1):  # array axes 1 .. dim hold reference axes dim-1 .. 0
                ax = 1 + (m.dim - 1 - b)
                lo, hi = np.take(w, 0, axis=ax), np.take(w, 1, axis=ax)
                w = np.expand_dims((hi - lo) if b == a else (lo * (1 - xi[b]) + hi * xi[b]), ax)
            J[:, :, a] = w.reshape(m.num_elements, m.dim)
        vol += np.linalg.det(J) / len(pts)
    return vol


@pytest.mark.parametrize("name,dim,ne,nbdr,volume", [("periodic-cube", 3, 27, 0, 8.0), ("beam-quad-o3-s5-p", 2, 45, 0, 5.0),
                                                     ("skinny-rectangle", 2, 30, 62, 0.3)])
def test_reads_the_reference_meshes(name, dim, ne, nbdr, volume):
    m = mesh_io.read_mfem_mesh(os.path.join(MESHES, name + ".mesh"))
    assert (m.dim, m.num_elements, m.bdr_vertices.shape[0]) == (dim, ne, nbdr)
    v = _volumes(m)
    assert (v > 0).all() and abs(v.sum() - volume) < 1e-5 * volume  # (the files print six digits)
    fn, fo, _, _ = capi.face_tables(m, [capi.make_bc(a, capi.WALL, capi.INV) for a in sorted(set(m.bdr_attributes.tolist()))])
    assert (fn >= 0).sum() == 2 * dim * ne - nbdr  # every face has a neighbour or a boundary attribute


def test_refuses_what_is_out_of_scope(tmp_path):
    tri = tmp_path / "tri.mesh"
    tri.write_text("MFEM mesh v1.0\n\ndimension\n2\n\nelements\n1\n1 2 0 1 2\n\nboundary\n0\n\nvertices\n3\n2\n0 0\n1 0\n0 1\n")
    with pytest.raises(ValueError, match="quadrilaterals / hexahedra"):
        mesh_io.read_mfem_mesh(str(tri))
    with pytest.raises(ValueError, match="MFEM mesh v1.0"):
        (tmp_path / "x.msh").write_text("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n")
        mesh_io.read_mfem_mesh(str(tmp_path / "x.msh"))


def test_oracle_free_stream_on_the_periodic_cube():
    """a uniform state has a zero residual on test/meshes/periodic-cube.mesh"""
    m = mesh_io.read_mfem_mesh(os.path.join(MESHES, "periodic-cube.mesh"))
    disc = capi.Disc(2, 0, 0, 0, 0)
    X = node_coordinates(m, 2)
    U = cases.dry_air_state(X, seed=1, amp=0.0)
    y = oracle_mult(m, disc, capi.dry_air_physics(capi.NS), [], U)["y"]
    assert np.abs(y).max() < 1e-9 * 101300.0


@pytest.mark.gpu
@pytest.mark.parametrize("name,fluid,order,nc", [("periodic-cube", "dry_air", 3, 0), ("periodic-cube", "argon", 2, 0),
                                                 ("periodic-cube", "dry_air", 2, 1), ("beam-quad-o3-s5-p", "dry_air", 3, 0),
                                                 ("beam-quad-o3-s5-p", "argon", 2, 1), ("skinny-rectangle", "dry_air", 2, 0)])
def test_hip_matches_the_oracle_on_the_reference_meshes(name, fluid, order, nc):
    m = mesh_io.read_mfem_mesh(os.path.join(MESHES, name + ".mesh"))
    disc = capi.Disc(order, nc, nc, 0, 0)
    X = node_coordinates(m, order, nc)
    if fluid == "dry_air":
        ph = capi.dry_air_physics(capi.NS)
        ph.dry_air.visc_mult = 1000.0
        U = cases.dry_air_state(X, seed=3, amp=0.02, nvel=m.dim)
        bcs = [] if name != "skinny-rectangle" else [
            capi.make_bc(1, capi.WALL, capi.VISC_ISOTH, [300.0]), capi.make_bc(3, capi.WALL, capi.VISC_ADIAB),
            capi.make_bc(2, capi.OUTLET, capi.SUB_P, [101300.0]), capi.make_bc(4, capi.INLET, capi.SUB_DENS_VEL, [1.2, 20.0, 0.0, 0.0])]
    else:
        ph = capi.argon_ternary_physics(capi.NS, False, capi.ARGON_MINIMAL, "arrhenius")
        U = cases.plasma_state(X, ph, nvel=m.dim, seed=3, amp=0.01)
        bcs = []
    got = hip_mult(m, disc, ph, bcs, U)
    ref = oracle_mult(m, disc, ph, bcs, U)
    err = rel_maxnorm(got["y"], ref["y"])
    print(name, fluid, order, nc, err)
    assert err.max() < (RHS_RTOL * 0.05 / 0.02 if fluid == "dry_air" else 5 * RHS_RTOL)
    # free-stream preservation on the same file, on the device
    U0 = (cases.dry_air_state(X, seed=3, amp=0.0, nvel=m.dim) if fluid == "dry_air" else cases.plasma_state(X, ph, nvel=m.dim, seed=3, amp=0.0))
    if not bcs and fluid == "dry_air":  # (a uniform reacting plasma has its chemistry sources: not zero)
        y0 = hip_mult(m, disc, ph, bcs, U0, want_grad=False)["y"]
        assert np.abs(y0).max() < 1e-9 * np.abs(ref["y"]).max()
