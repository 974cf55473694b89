"""numpy images of the index conventions of tps_amd/csrc (topology.hpp, kernels.hpp) used by the
CPU-side tests of the face tables and of the halo packing."""
import numpy as np


def gl_nodes(n):
    x, _ = np.polynomial.legendre.leggauss(n)
    return 0.5 * (x + 1.0)


def permute(dim, o, n, ia, ib=0):
    """kernels.hpp permute<DIM>: my tangential indices -> neighbour's flat index."""
    fa, fb = (o >> 1) & 1, (o >> 2) & 1
    if dim == 2:
        return n - 1 - ia if fa else ia
    if not (o & 1):
        ja = n - 1 - ia if fa else ia
        jb = n - 1 - ib if fb else ib
    else:
        ja = n - 1 - ib if fa else ib
        jb = n - 1 - ia if fb else ia
    return ja + n * jb


_LEX_OF_MFEM = {2: [0, 1, 3, 2], 3: [0, 1, 3, 2, 4, 5, 7, 6]}


def face_point_coords(mesh, e, f, pts):
    """physical coordinates of the tensor grid `pts` (1-D points in [0,1]) on local face f = 2d+s of
    element e, indexed [ia + n*ib] as the kernels do (tangential axes in increasing order)."""
    dim = mesh.dim
    d, s = f >> 1, f & 1
    V = np.zeros((1 << dim, dim))
    for v in range(1 << dim):
        V[_LEX_OF_MFEM[dim][v]] = mesh.elem_coords[e, v]
    n = len(pts)
    if dim == 2:
        a = 1 - d
        out = np.zeros((n, dim))
        for ia in range(n):
            xi = np.zeros(2)
            xi[d], xi[a] = s, pts[ia]
            out[ia] = _multilinear(V, xi)
        return out
    a, b = (1 if d == 0 else 0), (1 if d == 2 else 2)
    out = np.zeros((n * n, dim))
    for ib in range(n):
        for ia in range(n):
            xi = np.zeros(3)
            xi[d], xi[a], xi[b] = s, pts[ia], pts[ib]
            out[ia + n * ib] = _multilinear(V, xi)
    return out


def _multilinear(V, xi):
    dim = len(xi)
    x = np.zeros(dim)
    for c in range(1 << dim):
        w = 1.0
        for k in range(dim):
            w *= xi[k] if (c >> k) & 1 else 1.0 - xi[k]
        x += w * V[c]
    return x
