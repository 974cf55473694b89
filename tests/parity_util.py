"""Shared helpers of the parity tests: run the HIP operator and the CPU oracle on the same inputs."""
import numpy as np

# stated tolerance of the hot path: per-equation max-norm relative difference of one
# Mult between the HIP kernels and the CPU restatement.  The kernels re-order sums (sum
# factorisation, recomputed geometry), so bitwise equality is not expected in floating point.
RHS_RTOL = 1e-11


def rel_maxnorm(a, b):
    """per-row max|a-b| / max|b|"""
    a = np.asarray(a)
    b = np.asarray(b)
    num = np.abs(a - b).reshape(a.shape[0], -1).max(axis=1)
    den = np.abs(b).reshape(b.shape[0], -1).max(axis=1)
    return num / np.maximum(den, 1e-300)


def hip_mult(case_mesh, disc, physics, bcs, U, want_grad=True):
    import torch
    from tps_amd.rhs_operator import RHSoperator

    op = RHSoperator(case_mesh, disc, physics, bcs)
    x = torch.tensor(np.ascontiguousarray(U).ravel(), dtype=torch.float64, device=op.device)
    y = torch.empty_like(x)
    op.Mult(x, y, want_max_char_speed=True)
    torch.cuda.synchronize()
    out = {"y": y.cpu().numpy().reshape(U.shape), "max_char_speed": op.max_char_speed}
    if want_grad:
        out["Up"] = op.getPrimitives().cpu().numpy()
        out["gradUp"] = op.getGradients().cpu().numpy()
    op.close()
    return out


def oracle_mult(case_mesh, disc, physics, bcs, U):
    from oracle_lib import Oracle

    o = Oracle(case_mesh, disc, physics, bcs)
    y = o.mult(U)
    return {"y": y, "Up": o.primitives(), "gradUp": o.gradients(), "max_char_speed": o.max_char_speed, "oracle": o}
