"""Synthetic meshes standing in for the reference's git-LFS cylinder meshes.

Everything is produced in the layout ``tpsrhs_mesh`` takes (``include/tpsrhs.h``): topological
vertex ids per element in MFEM vertex order, per-element vertex coordinates (so periodic meshes
keep their geometry, as ``mfem::Mesh::MakePeriodic`` does), boundary faces with attributes.

* :func:`box_hex` / :func:`box_quad` -- (periodic) Cartesian blocks; ``test/meshes/periodic-cube.mesh``
  of the reference is ``box_hex(3, 3, 3, periodic=(True, True, True))`` on [0, 2*pi]^3 up to numbering.
* :func:`ogrid_cylinder` -- the O-grid "cylinder in a box" of the cyl3d configurations: patches
  1 = inlet (upstream half of the outer ring), 2 = outlet, 3 = cylinder wall, mirroring
  ``test/inputs/input.4iters.cyl.ini:31-46``.
* :func:`partition` -- contiguous element blocks with shared-face lists for the multi-rank path.
"""
from __future__ import annotations

import dataclasses
import itertools

import numpy as np

# MFEM hexahedron / quadrilateral vertex -> (a, b, c) corner bits
_HEX_CORNERS = np.array(
    [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)], dtype=np.int64
)
_QUAD_CORNERS = np.array([(0, 0), (1, 0), (1, 1), (0, 1)], dtype=np.int64)


@dataclasses.dataclass
class HostMesh:
    dim: int
    num_vertices: int
    elem_vertices: np.ndarray  # (ne, 2^dim) int32, MFEM vertex order
    elem_coords: np.ndarray  # (ne, 2^dim, dim) float64
    bdr_vertices: np.ndarray  # (nbf, 2^(dim-1)) int32
    bdr_attributes: np.ndarray  # (nbf,) int32
    shared_vertices: np.ndarray | None = None  # (nsf, 2^(dim-1)) local ids, sorted by global id
    shared_neighbor_rank: np.ndarray | None = None
    global_elements: np.ndarray | None = None  # (ne,) ids in the unpartitioned mesh
    elem_size: np.ndarray | None = None  # (ne,) optional mfem::Mesh::GetElementSize(e, 1) (tpsrhs_mesh::elem_size)

    @property
    def num_elements(self) -> int:
        return int(self.elem_vertices.shape[0])


def _rotations_hex():
    """The 24 proper rotations of the reference cube as permutations of the MFEM vertex list."""
    rots = []
    for perm in itertools.permutations(range(3)):
        for signs in itertools.product((1, -1), repeat=3):
            m = np.zeros((3, 3), dtype=np.int64)
            for r in range(3):
                m[r, perm[r]] = signs[r]
            if round(np.linalg.det(m)) != 1:
                continue
            # new local vertex v sits where old corner (R^-1 applied) was
            c = 2 * _HEX_CORNERS - 1  # centred coordinates
            src = (c @ m.T + 1) // 2
            idx = [int(np.where((_HEX_CORNERS == s).all(axis=1))[0][0]) for s in src]
            rots.append(np.array(idx, dtype=np.int64))
    return rots


def _rotations_quad():
    return [np.roll(np.arange(4), -k) for k in range(4)]


def scramble_orientations(mesh: HostMesh, seed: int) -> HostMesh:
    """Re-label the local vertices of every element by a random proper rotation, so that every
    face-to-face orientation occurs.  Geometry and topology are unchanged."""
    rng = np.random.default_rng(seed)
    rots = _rotations_hex() if mesh.dim == 3 else _rotations_quad()
    ev = mesh.elem_vertices.copy()
    ex = mesh.elem_coords.copy()
    pick = rng.integers(0, len(rots), size=mesh.num_elements)
    for e in range(mesh.num_elements):
        r = rots[pick[e]]
        ev[e] = mesh.elem_vertices[e][r]
        ex[e] = mesh.elem_coords[e][r]
    return dataclasses.replace(mesh, elem_vertices=ev, elem_coords=ex)


def _structured(dim, n, vertex_xyz, periodic, bdr_attr):
    """Generic structured block.  ``n`` cells per direction, ``vertex_xyz(i,j,k)`` -> coordinates of
    structured vertex (vectorised over index arrays), ``periodic`` per direction,
    ``bdr_attr[(d, side)]`` -> attribute or callable(face-centre xyz) -> attribute."""
    n = list(n)
    nvd = [n[d] if periodic[d] else n[d] + 1 for d in range(dim)]
    for d in range(dim):
        if periodic[d] and n[d] < 3:
            raise ValueError("periodic directions need at least 3 cells")
    corners = _HEX_CORNERS if dim == 3 else _QUAD_CORNERS

    def vid(idx):
        idx = [np.mod(idx[d], nvd[d]) if periodic[d] else idx[d] for d in range(dim)]
        v = idx[0]
        mul = nvd[0]
        for d in range(1, dim):
            v = v + idx[d] * mul
            mul *= nvd[d]
        return v

    grids = np.meshgrid(*[np.arange(n[d]) for d in range(dim)], indexing="ij")
    # element index: first direction fastest
    order = np.argsort(sum(grids[d].ravel() * int(np.prod(n[:d])) for d in range(dim)), kind="stable")
    cell = [grids[d].ravel()[order] for d in range(dim)]
    ne = cell[0].size
    nc = 1 << dim
    ev = np.zeros((ne, nc), dtype=np.int32)
    ex = np.zeros((ne, nc, dim), dtype=np.float64)
    for v in range(nc):
        idx = [cell[d] + corners[v][d] for d in range(dim)]
        ev[:, v] = vid(idx)
        ex[:, v, :] = vertex_xyz(*idx)
    # boundary faces
    bv, ba = [], []
    nfv = 1 << (dim - 1)
    for d in range(dim):
        if periodic[d]:
            continue
        for side in (0, 1):
            sel = np.where(cell[d] == (n[d] - 1 if side else 0))[0]
            fc = [v for v in range(nc) if corners[v][d] == side]
            if dim == 2:  # keep the edge's two vertices
                pass
            verts = ev[sel][:, fc]
            centre = ex[sel][:, fc, :].mean(axis=1)
            attr = bdr_attr.get((d, side))
            if attr is None:  # left open: the caller turns these faces into shared faces
                continue
            a = attr(centre) if callable(attr) else np.full(sel.size, attr)
            bv.append(verts)
            ba.append(np.asarray(a, dtype=np.int32))
    if bv:
        bv = np.concatenate(bv).astype(np.int32)
        ba = np.concatenate(ba).astype(np.int32)
    else:
        bv = np.zeros((0, nfv), dtype=np.int32)
        ba = np.zeros((0,), dtype=np.int32)
    return HostMesh(dim, int(np.prod(nvd)), ev, ex, bv, ba)


def box_hex(nx, ny, nz, lengths=(1.0, 1.0, 1.0), periodic=(True, True, True), bdr_attr=None, warp=0.0):
    """Cartesian block of hexahedra.  ``warp`` > 0 displaces the vertices smoothly (still periodic)
    so that elements become genuinely trilinear (non-constant Jacobians)."""
    L = np.asarray(lengths, dtype=np.float64)
    n = (nx, ny, nz)

    def xyz(i, j, k):
        x = np.stack([i * L[0] / nx, j * L[1] / ny, k * L[2] / nz], axis=-1).astype(np.float64)
        if warp:
            s = 2 * np.pi * x / L
            h = L / np.array(n)
            x = x + warp * h * np.stack(
                [np.sin(s[..., 1]) * np.cos(s[..., 2]), np.sin(s[..., 2]) * np.cos(s[..., 0]),
                 np.sin(s[..., 0]) * np.cos(s[..., 1])], axis=-1)
        return x

    if bdr_attr is None:
        bdr_attr = {(d, s): 1 + 2 * d + s for d in range(3) for s in (0, 1)}
    return _structured(3, n, xyz, periodic, bdr_attr)


def box_quad(nx, ny, lengths=(1.0, 1.0), periodic=(True, True), bdr_attr=None, warp=0.0, origin=(0.0, 0.0)):
    L = np.asarray(lengths, dtype=np.float64)
    o = np.asarray(origin, dtype=np.float64)
    n = (nx, ny)

    def xyz(i, j):
        x = np.stack([i * L[0] / nx, j * L[1] / ny], axis=-1).astype(np.float64)
        if warp:
            s = 2 * np.pi * x / L
            h = L / np.array(n)
            x = x + warp * h * np.stack([np.sin(s[..., 1]), np.sin(s[..., 0])], axis=-1)
        return x + o

    if bdr_attr is None:
        bdr_attr = {(d, s): 1 + 2 * d + s for d in range(2) for s in (0, 1)}
    return _structured(2, n, xyz, periodic, bdr_attr)


def ogrid_cylinder(nr, ntheta, nz, r_in=0.5, r_out=10.0, span=2.0, stretch=1.05):
    """O-grid around a cylinder: radial x azimuthal x spanwise hexes, geometric radial stretching,
    periodic in theta and z.  Attributes: 1 inlet (x<0 half of the outer ring), 2 outlet, 3 wall."""
    if abs(stretch - 1.0) < 1e-14:
        r = np.linspace(r_in, r_out, nr + 1)
    else:
        dr0 = (r_out - r_in) * (stretch - 1.0) / (stretch**nr - 1.0)
        r = r_in + dr0 * (stretch ** np.arange(nr + 1) - 1.0) / (stretch - 1.0)
        r[-1] = r_out

    def xyz(i, j, k):
        th = 2.0 * np.pi * j / ntheta
        return np.stack([r[i] * np.cos(th), r[i] * np.sin(th), span * k / nz], axis=-1)

    def outer(centre):
        return np.where(centre[:, 0] < 0.0, 1, 2)

    return _structured(3, (nr, ntheta, nz), xyz, (False, True, True), {(0, 0): 3, (0, 1): outer})


def ogrid_cylinder_slab(nr, ntheta, nz_local, rank, nparts, r_in=0.5, r_out=10.0, span_local=2.0, stretch=1.05):
    """Rank ``rank``'s spanwise slab of an O-grid with ``nparts * nz_local`` layers (periodic in z):
    the structured equivalent of :func:`partition` for the weak-scaling bench, built without ever
    forming the global mesh.  Every rank owns ``nr*ntheta*nz_local`` elements and shares its two
    z-faces with ranks ``rank-1`` and ``rank+1`` (mod ``nparts``)."""
    if nparts == 1:
        return ogrid_cylinder(nr, ntheta, nz_local, r_in, r_out, span_local, stretch)
    if abs(stretch - 1.0) < 1e-14:
        r = np.linspace(r_in, r_out, nr + 1)
    else:
        dr0 = (r_out - r_in) * (stretch - 1.0) / (stretch**nr - 1.0)
        r = r_in + dr0 * (stretch ** np.arange(nr + 1) - 1.0) / (stretch - 1.0)
        r[-1] = r_out
    k0 = rank * nz_local
    nzg = nparts * nz_local

    def xyz(i, j, k):
        th = 2.0 * np.pi * j / ntheta
        return np.stack([r[i] * np.cos(th), r[i] * np.sin(th), span_local * (k + k0) / nz_local], axis=-1)

    def outer(centre):
        return np.where(centre[:, 0] < 0.0, 1, 2)

    m = _structured(3, (nr, ntheta, nz_local), xyz, (False, True, False), {(0, 0): 3, (0, 1): outer})
    # local vertex id = i + (nr+1)*(j + ntheta*k); global id uses the global (periodic) layer index
    nvi, nvj = nr + 1, ntheta
    i, j = np.meshgrid(np.arange(nr), np.arange(ntheta), indexing="ij")
    i, j = i.ravel(), j.ravel()
    sv, sr, keys = [], [], []
    for kl, nbr in ((0, (rank - 1) % nparts), (nz_local, (rank + 1) % nparts)):
        kg = (k0 + kl) % nzg
        ci = np.stack([i, i + 1, i + 1, i], axis=1)
        cj = np.stack([j, j, (j + 1) % nvj, (j + 1) % nvj], axis=1)
        loc = ci + nvi * (cj + nvj * kl)
        glo = ci + nvi * (cj + nvj * kg)
        order = np.argsort(glo, axis=1)
        loc = np.take_along_axis(loc, order, axis=1)
        glo = np.take_along_axis(glo, order, axis=1)
        sv.append(loc)
        keys.append(glo)
        sr.append(np.full(loc.shape[0], nbr))
    sv, keys, sr = np.concatenate(sv), np.concatenate(keys), np.concatenate(sr)
    # group by neighbour rank (ascending), then by the sorted global key: identical on both sides
    srt = np.lexsort((keys[:, 3], keys[:, 2], keys[:, 1], keys[:, 0], sr))
    m.shared_vertices = sv[srt].astype(np.int32)
    m.shared_neighbor_rank = sr[srt].astype(np.int32)
    return m


def annulus_quad(nr, nz, r_in=0.0, r_out=1.0, length=2.0, bdr_attr=None):
    """2-D (r, z) block for the axisymmetric configuration (x = r)."""
    if bdr_attr is None:
        bdr_attr = {(0, 0): 4, (0, 1): 3, (1, 0): 1, (1, 1): 2}

    def xyz(i, j):
        return np.stack([r_in + (r_out - r_in) * i / nr, length * j / nz], axis=-1).astype(np.float64)

    return _structured(2, (nr, nz), xyz, (False, False), bdr_attr)


def annulus_quad_slab(nr, nz_local, rank, nparts, r_in=0.0, r_out=1.0, length_local=2.0, bdr_attr=None):
    """Rank ``rank``'s axial slab of an (r, z) block with ``nparts * nz_local`` cells along z (not periodic): the
    structured equivalent of :func:`partition` for the weak-scaling runs of the axisymmetric workloads, built
    without forming the global mesh.  Patch 1 (z = 0) belongs to rank 0, patch 2 (z = L) to the last rank; the
    interfaces in between are shared with ranks ``rank - 1`` and ``rank + 1``."""
    if bdr_attr is None:
        bdr_attr = {(0, 0): 4, (0, 1): 3, (1, 0): 1, (1, 1): 2}
    if nparts == 1:
        return annulus_quad(nr, nz_local, r_in, r_out, length_local, bdr_attr)
    k0 = rank * nz_local
    attrs = dict(bdr_attr)
    if rank > 0:
        attrs.pop((1, 0))
    if rank < nparts - 1:
        attrs.pop((1, 1))

    def xyz(i, j):
        return np.stack([r_in + (r_out - r_in) * i / nr, length_local * (j + k0) / nz_local], axis=-1).astype(np.float64)

    m = _structured(2, (nr, nz_local), xyz, (False, False), attrs)
    nvi = nr + 1  # local vertex id = i + (nr + 1) * j; the global id uses the global layer index
    i = np.arange(nr)
    sv, sr, keys = [], [], []
    for jl, nbr in ((0, rank - 1), (nz_local, rank + 1)):
        if nbr < 0 or nbr >= nparts:
            continue
        ci = np.stack([i, i + 1], axis=1)
        loc = ci + nvi * jl
        glo = ci + nvi * (k0 + jl)
        sv.append(loc)  # already sorted by global id along each edge
        keys.append(glo)
        sr.append(np.full(nr, nbr))
    sv, keys, sr = np.concatenate(sv), np.concatenate(keys), np.concatenate(sr)
    srt = np.lexsort((keys[:, 1], keys[:, 0], sr))  # by neighbour rank, then by the global key: identical on both sides
    m.shared_vertices = sv[srt].astype(np.int32)
    m.shared_neighbor_rank = sr[srt].astype(np.int32)
    return m


# --------------------------------------------------------------------------------------------
def partition(mesh: HostMesh, nparts: int, owner: np.ndarray | None = None):
    """Split ``mesh`` into ``nparts`` sub-meshes (contiguous element blocks unless ``owner`` is
    given).  Returns a list of :class:`HostMesh` with ``shared_*`` and ``global_elements`` filled:
    the role of ``Mesh::GeneratePartitioning`` + ``ParMesh`` (src/M2ulPhyS.cpp:333,362)."""
    ne = mesh.num_elements
    dim = mesh.dim
    nfv = 1 << (dim - 1)
    if owner is None:
        owner = (np.arange(ne) * nparts) // ne
    owner = np.asarray(owner)
    corners = _HEX_CORNERS if dim == 3 else _QUAD_CORNERS
    # all local faces of all elements, keyed by sorted global vertex ids
    faces = {}
    for d in range(dim):
        for side in (0, 1):
            fc = [v for v in range(1 << dim) if corners[v][d] == side]
            keys = np.sort(mesh.elem_vertices[:, fc], axis=1)
            for e in range(ne):
                faces.setdefault(tuple(keys[e]), []).append(e)
    shared = [dict() for _ in range(nparts)]  # rank -> nbr -> list of keys
    for key, els in faces.items():
        if len(els) == 2 and owner[els[0]] != owner[els[1]]:
            a, b = owner[els[0]], owner[els[1]]
            shared[a].setdefault(b, []).append(key)
            shared[b].setdefault(a, []).append(key)
    bkeys = {tuple(np.sort(v)): i for i, v in enumerate(mesh.bdr_vertices)}
    parts = []
    for r in range(nparts):
        els = np.where(owner == r)[0]
        gv = np.unique(mesh.elem_vertices[els])
        g2l = {int(g): i for i, g in enumerate(gv)}
        ev = np.vectorize(g2l.get)(mesh.elem_vertices[els]).astype(np.int32)
        ex = mesh.elem_coords[els]
        # boundary faces of this part
        sel = []
        elset = set(int(e) for e in els)
        for key, idx in bkeys.items():
            fe = faces.get(key)
            if fe and len(fe) == 1 and fe[0] in elset:
                sel.append(idx)
        sel = np.array(sorted(sel), dtype=np.int64)
        if sel.size:
            bv = np.vectorize(g2l.get)(mesh.bdr_vertices[sel]).astype(np.int32)
            ba = mesh.bdr_attributes[sel]
        else:
            bv = np.zeros((0, nfv), dtype=np.int32)
            ba = np.zeros((0,), dtype=np.int32)
        sv, sr = [], []
        for nbr in sorted(shared[r]):
            for key in sorted(shared[r][nbr]):  # same order on both sides: sorted global keys
                sv.append([g2l[int(g)] for g in key])
                sr.append(nbr)
        sv = np.array(sv, dtype=np.int32).reshape(-1, nfv)
        sr = np.array(sr, dtype=np.int32)
        parts.append(HostMesh(dim, len(gv), ev, ex, bv, ba, sv, sr, els.astype(np.int64)))
    return parts
