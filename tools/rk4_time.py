import sys, time
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from tps_amd import capi, cases
from tps_amd.rhs_operator import RHSoperator
c = cases.config(2)
op = RHSoperator(c.mesh, c.disc, c.physics, c.bcs)
x = torch.tensor(c.state().ravel(), dtype=torch.float64, device=op.device)
t = 0.0
dt = 1e-7
for _ in range(3): t = op.rk4_step(x, t, dt)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 20
for _ in range(n): t = op.rk4_step(x, t, dt)
torch.cuda.synchronize(); el = (time.perf_counter()-t0)/n
print("rk4 step ms", el*1e3, "finite", bool(torch.isfinite(x).all()))
