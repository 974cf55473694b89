"""The device time loop alone (tpsrhs_advance on the bench workload), for `rocprofv3 --kernel-trace --stats`:
    rocprofv3 --kernel-trace --stats -d gpurun_out/adv -- python3 tools/advance_trace.py [workload] [steps]
TPSRHS_GRAPH=0 gives per-kernel rows (a replayed hipGraph shows up as its kernels too)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from tps_amd import capi, meshgen  # noqa: E402
from tps_amd.rhs_operator import RHSoperator, node_coordinates  # noqa: E402

wname = sys.argv[1] if len(sys.argv) > 1 else "argon_p3"
nst = int(sys.argv[2]) if len(sys.argv) > 2 else 10
sys.argv = [sys.argv[0]]
import bench  # noqa: E402

order, physics, make_bcs, make_state, _, _ = bench.workload(wname)
mesh = meshgen.ogrid_cylinder_slab(28, 112, 16, 0, 1)
disc = capi.Disc(order, 0, 0, 0, 0)
U = make_state(node_coordinates(mesh, order), physics)
op = RHSoperator(mesh, disc, physics, make_bcs(physics), device=0)
x = torch.tensor(U.ravel(), dtype=torch.float64, device=op.device)
y = torch.empty_like(x)
for _ in range(5):
    op.Mult(x, y)
op.advance(x, 0.0, 1.0e-10, 3, True)
torch.cuda.synchronize()
t0 = time.perf_counter()
op.advance(x, 0.0, 1.0e-10, nst, True)
el = (time.perf_counter() - t0) / nst
print(f"{wname}: {1e3 * el:.4f} ms per RK4 step")
