"""Probe (GPU): the nodal flux F_c - F_v and the point sources that k_flux of ONE plasma instantiation leaves in LDS, per
block and node, from a diagnostic build of its translation unit (-DTPSRHS_DUMPF=1, kernels.hpp), next to y and the oracle's y.
    tools/build_variant.sh dumpO3 "-DTPSRHS_DUMPF=1" plasma_3d_n7
    TPSRHS_FAMILY_PATH=$PWD/tps_amd/csrc/_ab/dumpO3 python tools/probe_dumpf.py <tag> 3d 7 0 0 2 1 1
writes gpurun_out/dumpf_<tag>.npz (F [blocks][rows][nodes], y, y_ref)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from parity_util import oracle_mult  # noqa: E402

from tps_amd import capi, cases, meshgen  # noqa: E402
from tps_amd.rhs_operator import RHSoperator, node_coordinates  # noqa: E402

tag, geo, nsp, ambi, two_t, tr, order, nc = sys.argv[1], sys.argv[2], int(sys.argv[3]), bool(int(sys.argv[4])), bool(int(sys.argv[5])), int(
    sys.argv[6]), int(sys.argv[7]), int(sys.argv[8])
assert geo == "3d"
ph = capi.argon_levels_physics(nsp - 3, ambi, capi.NS, tr, two_t, True, third_order_ke=False) if nsp > 3 else capi.argon_ternary_physics(
    capi.NS, two_t, tr, "arrhenius", ambipolar=ambi, third_order_ke=(tr != capi.CONSTANT))
ph.gas_transport.multiply = 1
for k in range(4):
    ph.gas_transport.flux_trns_multiplier[k] = 30.0
ph.gas_transport.diff_mult = ph.gas_transport.mobil_mult = 30.0
mesh = meshgen.scramble_orientations(meshgen.box_hex(3, 4, 3, warp=0.1), 254)
disc = capi.Disc(order, nc, nc, 0, 0)
U = cases.plasma_state(node_coordinates(mesh, order, nc), ph, nvel=3, seed=11, amp=0.005 if order == 1 else 0.01, vel0=(20.0, 0.0, 0.0))
ref = oracle_mult(mesh, disc, ph, [], U)
op = RHSoperator(mesh, disc, ph, [])
x = torch.tensor(np.ascontiguousarray(U).ravel(), dtype=torch.float64, device=op.device)
y = torch.empty_like(x)
op.Mult(x, y)
torch.cuda.synchronize()
neq = U.shape[0]
npe = (order + 1) ** 3
epb = {1: 2, 2: 2, 3: 1}[order] if nc else {1: 3, 2: 2, 3: 1}[order]
nblocks = (mesh.num_elements + epb - 1) // epb
rows = neq * 3 + neq
buf = np.zeros((nblocks, rows, epb * npe))
lib = C.CDLL(os.path.join(os.environ["TPSRHS_FAMILY_PATH"].split(":")[0], f"libtpsrhs_plasma_3d_n{nsp}{'a' if ambi else ''}.so"))
assert lib.tpsrhs_debug_dumpf(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
got = y.cpu().numpy().reshape(U.shape)
sc = np.abs(ref["y"]).reshape(neq, -1).max(axis=1)
print(tag, "y err per eq", np.array2string(np.abs(got - ref["y"]).reshape(neq, -1).max(axis=1) / sc, precision=1), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", f"dumpf_{tag}.npz"), F=buf, y=got, y_ref=ref["y"], U=U)
