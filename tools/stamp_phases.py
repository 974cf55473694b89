"""Where a block of the reacting k_gradient spends its cycles (diagnostic build, GPU box):
    tools/build_variant.sh stamp "-DTPSRHS_STAMP=1 -mllvm -disable-machine-licm" plasma_3d_n3a
    TPSRHS_FAMILY_PATH=$PWD/tps_amd/csrc/_ab/stamp python tools/stamp_phases.py
Prints the share of each phase in the summed s_memtime differences (shares, not lengths: the stamps fence
overlaps the production kernel has)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from tps_amd import capi, cases, meshgen  # noqa: E402
from tps_amd.rhs_operator import RHSoperator, node_coordinates  # noqa: E402

sys.argv = [sys.argv[0]]
import bench  # noqa: E402

order, physics, make_bcs, make_state, description, _ = bench.workload(os.environ.get("STAMP_WORKLOAD", "argon_p3"))  # or cfg3
mesh = meshgen.ogrid_cylinder_slab(28, 112, 16, 0, 1)
nc = 1 if os.environ.get("STAMP_WORKLOAD", "").startswith("gll") else 0  # the Gauss-Lobatto pair
disc = capi.Disc(order, nc, nc, 0, 0)
X = node_coordinates(mesh, order, nc)
U = make_state(X, physics)
op = RHSoperator(mesh, disc, physics, make_bcs(physics), device=0)
x = torch.tensor(U.ravel(), dtype=torch.float64, device=op.device)
y = torch.empty_like(x)
# the diagnostic build: a plasma family (TPSRHS_FAMILY_PATH) or, for the dry-air workloads, the core library (TPSRHS_LIB)
lib = (C.CDLL(os.environ["TPSRHS_LIB"]) if os.environ.get("STAMP_WORKLOAD", "argon_p3") in ("cfg2", "gll_dry")
       else C.CDLL(os.path.join(os.environ["TPSRHS_FAMILY_PATH"].split(":")[0], "libtpsrhs_plasma_3d_n3a.so")))
import numpy as np  # noqa: E402

nblocks = mesh.num_elements // (2 if order == 2 else 1)  # elements per block: one p = 3 hex, two p = 2 hexes
buf = np.zeros((nblocks, 16), dtype=np.uint32)
for _ in range(3):
    op.Mult(x, y)
torch.cuda.synchronize()
assert lib.tpsrhs_debug_stamps(buf.ctypes.data_as(C.c_void_p), nblocks) == 0
names = ["tables+vertices", "loads+prim", "volume gradient", "jump phase", "gradUp store", "visc: state interp",
         "visc: closure (coeffs)", "visc: gradient interp", "visc: flux + TB store"]
if os.environ.get("STAMP_KERNEL") == "flux":  # built with -DTPSRHS_STAMP=2
    names = ["tables+vertices", "loads of U, gradUp -> LDS", "nodal physics (closure, flux, source)", "volume term",
             "face term, direction 0", "face term, direction 1", "face term, direction 2", "store y"]
mean = buf.astype(np.float64).mean(axis=0)
tot = mean.sum()
for i, nm in enumerate(names):
    print(f"{nm:28s} {mean[i]:10.0f} cycles/block  {100.0 * mean[i] / tot:5.1f} %   (median {np.median(buf[:, i]):8.0f})")
print(f"{'sum':28s} {tot:10.0f} cycles/block")
