"""The C ABI without a GPU: the library loads, exports every symbol the header declares, refuses
to run without a device, and derives consistent face tables (host logic)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from face_util import face_point_coords, gl_nodes, permute
from tps_amd import capi, cases, meshgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    hdr = open(os.path.join(ROOT, "include", "tpsrhs.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|const char \*)\s*(tpsrhs_[a-z0-9_]+)\s*\(", hdr, flags=re.M))
    assert declared == set(capi.EXPORTED_SYMBOLS), declared ^ set(capi.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.tpsrhs_version()
    assert lib.tpsrhs_status_string(2) == b"TPSRHS_ERR_UNSUPPORTED"


def test_rccl_library_exports_every_declared_symbol():
    """libtpsrhs_rccl.so (the native RCCL implementation of the halo / reduce hooks) loads and exports what
    include/tpsrhs_rccl.h declares; no communicator is created without GPUs"""
    from tps_amd import halo_rccl

    lib = halo_rccl.load()
    hdr = open(os.path.join(ROOT, "include", "tpsrhs_rccl.h")).read()
    declared = set(re.findall(r"^(?:int|const char \*)\s*(tpsrhs_rccl_[a-z0-9_]+)\s*\(", hdr, flags=re.M))
    assert declared == set(halo_rccl.EXPORTED_SYMBOLS), declared ^ set(halo_rccl.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    # the function pointers the operator receives have the C types of tpsrhs_halo_fn / tpsrhs_reduce_fn
    assert C.cast(lib.tpsrhs_rccl_halo, C.c_void_p).value and C.cast(lib.tpsrhs_rccl_reduce, C.c_void_p).value


def test_struct_sizes_match_header():
    """ctypes mirror vs sizeof() of the C structs (compiled on the fly with gcc)."""
    import subprocess
    import tempfile

    names = ["tpsrhs_mesh", "tpsrhs_disc", "tpsrhs_dry_air", "tpsrhs_perfect_mixture", "tpsrhs_constant_transport",
             "tpsrhs_gas_transport", "tpsrhs_table", "tpsrhs_chemistry", "tpsrhs_radiation", "tpsrhs_physics",
             "tpsrhs_bc", "tpsrhs_runtime", "tpsrhs_heat_source", "tpsrhs_sponge_zone", "tpsrhs_forcing", "tpsrhs_sgs",
             "tpsrhs_visc_sponge", "tpsrhs_mixing_length"]
    mirror = [capi.Mesh, capi.Disc, capi.DryAir, capi.PerfectMixture, capi.ConstantTransport, capi.GasTransport,
              capi.Table, capi.Chemistry, capi.Radiation, capi.Physics, capi.BC, capi.Runtime, capi.HeatSource,
              capi.SpongeZone, capi.Forcing, capi.Sgs, capi.ViscSponge, capi.MixingLength]
    src = '#include <stdio.h>\n#include "tpsrhs.h"\nint main(){' + "".join(
        f'printf("%zu\\n", sizeof({n}));' for n in names) + "return 0;}"
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "s.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(td, "s"),
                        os.path.join(td, "s.c")], check=True)
        sizes = [int(x) for x in subprocess.run([os.path.join(td, "s")], capture_output=True, text=True,
                                                check=True).stdout.split()]
    assert sizes == [C.sizeof(m) for m in mirror]


def test_create_without_gpu_fails_loudly():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = capi.load()
    c = cases.cyl3d(3, 8, 3, 1, capi.EULER)
    ma = capi.MeshArgs(c.mesh)
    bcs = (capi.BC * len(c.bcs))(*c.bcs)
    h = C.c_void_p()
    st = lib.tpsrhs_create(C.byref(ma.c), C.byref(c.disc), C.byref(c.physics), len(c.bcs), bcs, None, C.byref(h))
    assert st == 5 and not h.value  # TPSRHS_ERR_NO_DEVICE: no CPU fallback
    assert b"no HIP device" in lib.tpsrhs_last_error()


def test_unsupported_configurations_are_refused():
    lib = capi.load()
    c = cases.cyl3d(3, 8, 3, 1, capi.EULER)
    ma = capi.MeshArgs(c.mesh)
    bcs = (capi.BC * len(c.bcs))(*c.bcs)
    h = C.c_void_p()
    for basis, rule in ((0, 1), (1, 0)):  # built pairs: Gauss-Legendre (0, 0) and Gauss-Lobatto (1, 1)
        disc = capi.Disc(2, basis, rule, 0, 0)
        st = lib.tpsrhs_create(C.byref(ma.c), C.byref(disc), C.byref(c.physics), len(c.bcs), bcs, None, C.byref(h))
        assert st == 2
    disc = capi.Disc(2, 1, 1, 1, 0)  # the Gauss-Lobatto pair is planar / 3-D
    st = lib.tpsrhs_create(C.byref(ma.c), C.byref(disc), C.byref(c.physics), len(c.bcs), bcs, None, C.byref(h))
    assert st in (1, 2)
    bad = capi.make_bc(3, capi.WALL, capi.VISC_GNRL)
    bcs2 = (capi.BC * 3)(c.bcs[0], c.bcs[1], bad)
    st = lib.tpsrhs_create(C.byref(ma.c), C.byref(c.disc), C.byref(c.physics), 3, bcs2, None, C.byref(h))
    assert st == 2
    with pytest.raises(RuntimeError, match="no boundary condition"):
        capi.face_tables(c.mesh, c.bcs[:2])


@pytest.mark.parametrize("dim", [2, 3])
def test_face_tables_match_geometry(dim):
    """neighbour slots are mutual and the orientation code maps my face points onto the
    neighbour's: checked with physical coordinates on a warped, orientation-scrambled mesh."""
    if dim == 3:
        mesh = meshgen.scramble_orientations(meshgen.box_hex(3, 4, 3, warp=0.1, periodic=(True, False, True)), 5)
        bcs = [capi.make_bc(3, capi.WALL, capi.INV), capi.make_bc(4, capi.WALL, capi.INV)]
    else:
        mesh = meshgen.scramble_orientations(meshgen.box_quad(4, 5, warp=0.1, periodic=(False, True)), 5)
        bcs = [capi.make_bc(1, capi.WALL, capi.INV), capi.make_bc(2, capi.WALL, capi.INV)]
    fn, fo, _, _ = capi.face_tables(mesh, bcs)
    nlf = 2 * dim
    n = 4
    pts = gl_nodes(n)
    L = np.array([1.0, 1.0, 1.0])[:dim]
    nb_bdr = 0
    for e in range(mesh.num_elements):
        for f in range(nlf):
            nb = fn[e, f]
            if nb < 0:
                nb_bdr += 1
                assert 1 <= -nb <= len(bcs)
                continue
            e2, f2 = divmod(int(nb), nlf)
            assert fn[e2, f2] == e * nlf + f
            mine = face_point_coords(mesh, e, f, pts)
            theirs = face_point_coords(mesh, e2, f2, pts)
            for ib in range(n if dim == 3 else 1):
                for ia in range(n):
                    k = permute(dim, int(fo[e, f]), n, ia, ib)
                    d = mine[ia + n * ib] - theirs[k]
                    d -= np.round(d / L) * L  # periodic images
                    assert np.abs(d).max() < 1e-12
    assert nb_bdr == len(mesh.bdr_attributes)
