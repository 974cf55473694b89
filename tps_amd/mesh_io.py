"""Reader of MFEM's native mesh format (``MFEM mesh v1.0``) -- the files the reference's inputs name (``mesh = meshes/*.mesh``,
read by ``mfem::Mesh(filename)`` in src/M2ulPhyS.cpp:296-330): quadrilateral and hexahedral elements, straight-sided
geometry, periodic meshes in MFEM's own representation (identified vertex ids + a discontinuous ``nodes`` grid function of
order 1 that keeps every element's own corner coordinates).  Returns the :class:`tps_amd.meshgen.HostMesh` that
``tpsrhs_create`` takes (elements in MFEM vertex order, per-element corner coordinates, boundary faces by vertex ids).

What the reference supports and this reader refuses (``ValueError``): simplices (the library's kernels are tensor-product),
curved geometry (``nodes`` of order > 1), NURBS / non-conforming sections, the Gmsh ``.msh`` files (git-LFS pointers in the
reference checkout: none is available to test against).

Boundary elements of a periodic mesh: MFEM keeps the faces of the periodic planes in the ``boundary`` section; a face that
two elements share is an INTERIOR face for the DG operators (``Mesh::GetBdrFaceTransformations`` returns NULL there, so the
reference's boundary integrators skip it): they are dropped here."""
import numpy as np

from .meshgen import HostMesh

_SQUARE, _CUBE = 3, 5
# MFEM's nodal ordering of an order-1 L2 (discontinuous) tensor element is lexicographic (x fastest); its vertex order is
# the counter-clockwise one (mesh/geom.cpp): lexicographic node -> MFEM vertex
_LEX_TO_MFEM = {2: [0, 1, 3, 2], 3: [0, 1, 3, 2, 4, 5, 7, 6]}
# local faces of MFEM's quadrilateral (edges) and hexahedron (mesh/geom.cpp: Geometry::Constants<...>::FaceVert / Edges)
_FACES = {2: [(0, 1), (1, 2), (2, 3), (3, 0)],
          3: [(3, 2, 1, 0), (0, 1, 5, 4), (1, 2, 6, 5), (2, 3, 7, 6), (3, 0, 4, 7), (4, 5, 6, 7)]}


def _tokens(path):
    with open(path) as fh:
        for line in fh:
            line = line.split("#", 1)[0].strip()
            if line:
                yield line


def read_mfem_mesh(path) -> HostMesh:
    lines = list(_tokens(path))
    if not lines or not lines[0].startswith("MFEM mesh v1.0"):
        raise ValueError(f"{path}: not an 'MFEM mesh v1.0' file (NURBS / non-conforming / v1.2 formats are not read)")
    pos = {name: i for i, name in enumerate(lines) if name in ("dimension", "elements", "boundary", "vertices", "nodes")}
    for need in ("dimension", "elements", "boundary", "vertices"):
        if need not in pos:
            raise ValueError(f"{path}: section '{need}' missing")
    dim = int(lines[pos["dimension"] + 1])
    if dim not in (2, 3):
        raise ValueError(f"{path}: dimension {dim}")
    nv_el, geom = 1 << dim, (_SQUARE if dim == 2 else _CUBE)

    def section(name, width):
        n = int(lines[pos[name] + 1])
        rows = [lines[pos[name] + 2 + k].split() for k in range(n)]
        for r in rows:
            if int(r[1]) != (geom if width == nv_el else (1 if dim == 2 else _SQUARE)):
                raise ValueError(f"{path}: geometry type {r[1]} in '{name}': only quadrilaterals / hexahedra are built "
                                 "(tensor-product kernels; the reference's simplex meshes are out of scope)")
        a = np.array([[int(x) for x in r] for r in rows], dtype=np.int64).reshape(n, 2 + width)
        return a[:, 0].astype(np.int32), a[:, 2:].astype(np.int32)

    _, ev = section("elements", nv_el)
    battr, bv = section("boundary", nv_el // 2)
    nvert = int(lines[pos["vertices"] + 1])
    ne = ev.shape[0]
    if "nodes" in pos:  # a grid function carries the geometry (periodic meshes: discontinuous, order 1)
        i = pos["nodes"] + 1
        hdr = {}
        while ":" in lines[i] or lines[i] == "FiniteElementSpace":
            if ":" in lines[i]:
                k, v = lines[i].split(":", 1)
                hdr[k.strip()] = v.strip()
            i += 1
        fec = hdr.get("FiniteElementCollection", "")
        if not (fec.startswith("L2_") and fec.endswith(f"{dim}D_P1")):
            raise ValueError(f"{path}: nodes in '{fec}': only straight-sided geometry (discontinuous order-1 nodes) is read; "
                             "curved elements are out of scope")
        if int(hdr.get("VDim", dim)) != dim:
            raise ValueError(f"{path}: nodes VDim {hdr.get('VDim')}")
        if int(hdr.get("Ordering", 1)) == 1:  # byVDIM: x y (z) per node
            vals = np.array(" ".join(lines[i:i + ne * nv_el]).split(), dtype=np.float64).reshape(ne, nv_el, dim)
        else:  # byNODES: all x, then all y, ...
            flat = np.array(" ".join(lines[i:]).split(), dtype=np.float64)[:ne * nv_el * dim]
            vals = flat.reshape(dim, ne, nv_el).transpose(1, 2, 0)
        coords = np.empty_like(vals)
        coords[:, _LEX_TO_MFEM[dim], :] = vals
    else:  # plain vertex coordinates
        vd = int(lines[pos["vertices"] + 2])
        if vd != dim:
            raise ValueError(f"{path}: vertices of dimension {vd} in a mesh of dimension {dim}")
        xyz = np.array(" ".join(lines[pos["vertices"] + 3:pos["vertices"] + 3 + nvert]).split(), dtype=np.float64).reshape(nvert, dim)
        coords = xyz[ev]
    # Mesh::CheckElementOrientation(fix_it = true) [third party: MFEM mesh/mesh.cpp], which mfem::Mesh's file constructor runs:
    # a quadrilateral listed clockwise (negative Jacobian at its first vertex) has its vertices 1 and 3 swapped;
    # an inverted hexahedron is an error there and here
    d1, d2 = coords[:, 1] - coords[:, 0], coords[:, nv_el // 2 + 1 if dim == 2 else 3] - coords[:, 0]
    if dim == 2:
        inv = d1[:, 0] * d2[:, 1] - d1[:, 1] * d2[:, 0] < 0.0
        ev[inv] = ev[inv][:, [0, 3, 2, 1]]
        coords[inv] = coords[inv][:, [0, 3, 2, 1]]
    else:
        d3 = coords[:, 4] - coords[:, 0]
        if (np.einsum("ei,ei->e", np.cross(d1, d2), d3) < 0.0).any():
            raise ValueError(f"{path}: inverted hexahedra")
    # boundary elements on faces that two elements share are interior faces (periodic planes)
    count = {}
    for e in range(ne):
        for f in _FACES[dim]:
            key = tuple(sorted(int(ev[e, k]) for k in f))
            count[key] = count.get(key, 0) + 1
    keep = []
    for b in range(bv.shape[0]):
        c = count.get(tuple(sorted(int(x) for x in bv[b])), 0)
        if c == 0:
            raise ValueError(f"{path}: boundary element {b} is not a face of any element")
        if c == 1:
            keep.append(b)
    keep = np.array(keep, dtype=np.int64)
    return HostMesh(dim, nvert, np.ascontiguousarray(ev), np.ascontiguousarray(coords),
                    np.ascontiguousarray(bv[keep]).reshape(len(keep), nv_el // 2), np.ascontiguousarray(battr[keep]))


def refine_uniform(mesh: HostMesh, levels: int = 1) -> HostMesh:
    """``[flow] refinement_levels`` (src/M2ulPhyS.cpp:353-356: ``serial_mesh->UniformRefinement()`` per level): every
    quadrilateral / hexahedron is split into 4 / 8 children at the midpoints of its edges, faces and cell.  Topology as
    in MFEM: one new vertex per edge, per face and per element, shared by the elements that share the edge / face (new ids
    are keyed by the parent vertex ids they average, so periodic identifications carry over); geometry from the parent's
    OWN corner coordinates (multi-linear), which keeps periodic images apart.  Needs at least three elements around a
    periodic direction (two vertices then name one edge only), like every mesh the library accepts."""
    for _ in range(int(levels)):
        mesh = _refine_once(mesh)
    return mesh


def _refine_once(mesh: HostMesh) -> HostMesh:
    dim = mesh.dim
    nv_el = 1 << dim
    from .meshgen import _HEX_CORNERS, _QUAD_CORNERS

    corners = _QUAD_CORNERS if dim == 2 else _HEX_CORNERS  # MFEM vertex -> corner bits
    ids = {}

    def vid(key):
        k = tuple(sorted(key))
        if k not in ids:
            ids[k] = len(ids)
        return ids[k]

    for v in range(mesh.num_vertices):  # old vertices keep their ids
        ids[(v,)] = v
    ev, ec = mesh.elem_vertices, mesh.elem_coords
    ne = ev.shape[0]
    new_ev = np.empty((ne * nv_el, nv_el), dtype=np.int32)
    new_ec = np.empty((ne * nv_el, nv_el, dim), dtype=np.float64)
    # points of the 3^dim lattice of an element: lattice index per axis 0, 1 (midpoint), 2
    lattice = np.array(np.meshgrid(*([np.arange(3)] * dim), indexing="ij")).reshape(dim, -1).T
    for e in range(ne):
        pid, pxyz = {}, {}
        for lat in lattice:
            # parent corners this point averages: per axis 0 -> bit 0, 2 -> bit 1, 1 -> both
            sel = [c for c in range(nv_el) if all(lat[a] == 1 or corners[c][a] == lat[a] // 2 for a in range(dim))]
            key = [int(ev[e, c]) for c in sel]
            # an edge / face midpoint of a periodic ring: keyed by its vertex ids; the cell centre by the element
            pid[tuple(lat)] = vid(key) if len(sel) < nv_el else vid(key + [-(e + 1)])
            pxyz[tuple(lat)] = ec[e, sel].mean(axis=0)
        for child in range(nv_el):
            off = corners[child]  # the child that touches parent corner `child`
            for c in range(nv_el):
                lat = tuple(int(off[a] + corners[c][a]) for a in range(dim))
                new_ev[e * nv_el + child, c] = pid[lat]
                new_ec[e * nv_el + child, c] = pxyz[lat]
    nb = mesh.bdr_vertices.shape[0]
    nvf = nv_el // 2
    fc = _QUAD_CORNERS if dim == 3 else np.array([[0], [1]])
    new_bv = np.empty((nb * nvf, nvf), dtype=np.int32)
    for b in range(nb):
        bvs = [int(x) for x in mesh.bdr_vertices[b]]
        flat = np.array(np.meshgrid(*([np.arange(3)] * (dim - 1)), indexing="ij")).reshape(dim - 1, -1).T
        pid = {}
        for lat in flat:
            sel = [c for c in range(nvf) if all(lat[a] == 1 or fc[c][a] == lat[a] // 2 for a in range(dim - 1))]
            pid[tuple(lat)] = vid([bvs[c] for c in sel])
        for child in range(nvf):
            for c in range(nvf):
                lat = tuple(int(fc[child][a] + fc[c][a]) for a in range(dim - 1))
                new_bv[b * nvf + child, c] = pid[lat]
    return HostMesh(dim, len(ids), new_ev, new_ec, new_bv, np.repeat(mesh.bdr_attributes, nvf).astype(np.int32))
