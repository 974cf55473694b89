"""Time one Mult of a full-size 3-D argon cylinder variant (run on a GPU box):
    python tools/time_case.py <two_temperature 0|1> <transport 0|1|2 = minimal|mixture|constant> [order]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tps_amd import cases  # noqa: E402
from tps_amd.rhs_operator import RHSoperator  # noqa: E402

two_t, tr = bool(int(sys.argv[1])), int(sys.argv[2])
order = int(sys.argv[3]) if len(sys.argv) > 3 else 3
c = cases.argon_cyl3d(28, 112, 16, order, two_t, tr)
op = RHSoperator(c.mesh, c.disc, c.physics, c.bcs)
x = torch.tensor(c.state(seed=1, amp=0.01).ravel(), dtype=torch.float64, device=op.device)
y = torch.empty_like(x)
for _ in range(5):
    op.Mult(x, y)
torch.cuda.synchronize()
op.enable_kernel_timing(True)
t0 = time.perf_counter()
for _ in range(30):
    op.Mult(x, y)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 30
print(f"2T={two_t} transport={tr} p={order}: {1e3 * dt:.3f} ms/Mult", {k: round(v, 3) for k, v in op.kernel_times().items()}, "finite", bool(torch.isfinite(y).all()))
