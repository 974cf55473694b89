"""``test/mms.euler.test`` -- the reference's convergence check of the ASSEMBLED 3-D operator inside its time loop -- on the
oracle (CPU) and on the HIP path.

What the reference runs (``test/mms.euler.test:5-28``, ``test/inputs/mms.euler.3d.r1.ini`` / ``r2.ini``): the Euler
equations for dry air on ``test/meshes/periodic-cube.mesh`` (27 periodic hexes on [-1, 1]^3) refined once / twice, order 1,
``basisType = integrationRule = 0`` -- the collocated Gauss-Legendre pair of the metric's workloads --, RK4 with a fixed
``dt`` of 2e-5 / 1e-5 for 300 / 600 steps from the exact state, the MASA source of the manufactured solution
``euler_transient_3d`` added to the residual at the time of every RK4 stage (``src/rhs_operator.cpp:452-461``,
``src/forcing_terms.cpp:979-1011``).  ``M2ulPhyS::checkSolutionError`` (``src/masa_handler.cpp:139-152``) prints the L2
errors of the density, velocity and pressure grid functions, and the test holds the convergence RATES between the two
runs: "empirically observed" 2.1646 / 2.0385 / 2.1718 inside windows 0.01 wide (``test/mms.euler.test:39-104``).

Every parameter of the solution is set by the reference (``src/masa_handler.cpp:356-417``); its form is MASA's
(``tests/mms_util.py::euler_transient_3d``: the published ``euler_3d`` in space, the time terms of ``euler_transient_1d``),
written down once; the source is that state put through the Euler equations by sympy.  The first evaluation gave
2.1646 / 2.0385 / 2.1718 -- every digit the reference prints, for all three quantities.  Nothing was adjusted
(``tools/mms_euler_transient.py`` runs the other sine / cosine choices of the time terms for the record: they land
elsewhere).  This ties volume term, face term (Lax-Friedrichs on the reference's own hexahedral mesh, periodic
identification included), inverse mass and the RK4 stage sequence of the 3-D collocated pair to numbers the reference's
binary produced -- not only to the equations."""
import os

import numpy as np
import pytest

from mms_util import euler_transient_3d, lp_errors_box
from tps_amd import capi, mesh_io

MESH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "meshes", "periodic-cube.mesh")
# test/mms.euler.test:44-45, 66-67, 88-89 ("empirically observed rate ...") and the windows of :47-48, 69-70, 91-92
OBSERVED = (2.1646, 2.0385, 2.1718)
WINDOWS = ((2.16, 2.17), (2.03, 2.04), (2.17, 2.18))
RUNS = ((1, 300, 2e-5), (2, 600, 1e-5))  # refinement_levels, maxIters, dt_fixed of r1.ini / r2.ini


def _rk4(mult, source, X, x, steps, dt):
    """mfem::RK4Solver::Step [third party: MFEM linalg/ode.cpp] around Mult + the MASA forcing at the stage's time"""
    t = 0.0
    for _ in range(steps):
        k1 = mult(x) + source(X, t)
        k2 = mult(x + 0.5 * dt * k1) + source(X, t + 0.5 * dt)
        k3 = mult(x + 0.5 * dt * k2) + source(X, t + 0.5 * dt)
        k4 = mult(x + dt * k3) + source(X, t + dt)
        x = x + dt / 6.0 * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
        t += dt
    return x, t


def _rates(errors):
    (d1, v1, p1), (d2, v2, p2) = errors
    return tuple(np.log(b / a) / np.log(0.5) for a, b in ((d1, d2), (v1, v2), (p1, p2)))


def _check(rates):
    print("mms.euler.test rates:", ["%.5f" % r for r in rates], "reference:", OBSERVED)
    for r, (lo, hi), obs in zip(rates, WINDOWS, OBSERVED):
        assert lo < r < hi, (r, lo, hi)
        assert abs(r - obs) < 5e-5  # every digit the reference prints


def _oracle_run(level, steps, dt):
    from oracle_lib import Oracle

    m = mesh_io.refine_uniform(mesh_io.read_mfem_mesh(MESH), level)
    o = Oracle(m, capi.Disc(1, 0, 0, 0, 0), capi.dry_air_physics(capi.EULER), threads=8)
    X = o.node_coords()
    ms = euler_transient_3d()
    x, t = _rk4(o.mult, ms.source, X, ms.state(X, 0.0), steps, dt)
    return m, X, x, t, ms


def test_mms_euler_transient_rates_oracle():
    errors = []
    for level, steps, dt in RUNS:
        _, X, x, t, ms = _oracle_run(level, steps, dt)
        errors.append(lp_errors_box(X, x, ms, t, p=1))
    _check(_rates(errors))


@pytest.mark.gpu
def test_mms_euler_transient_rates_hip():
    """the same 300 + 600 RK4 steps with every Mult on the device (libtpsrhs.so through RHSoperator.Mult; the stage
    combinations in torch on the same stream); the oracle supplies the node coordinates and, for the coarse run, the state
    to compare with"""
    import torch

    from oracle_lib import Oracle
    from tps_amd.rhs_operator import RHSoperator

    ms = euler_transient_3d()
    errors = []
    for level, steps, dt in RUNS:
        m = mesh_io.refine_uniform(mesh_io.read_mfem_mesh(MESH), level)
        disc, ph = capi.Disc(1, 0, 0, 0, 0), capi.dry_air_physics(capi.EULER)
        X = Oracle(m, disc, ph).node_coords()
        op = RHSoperator(m, disc, ph, [])

        def mult(x, op=op):
            y = torch.empty_like(x)
            op.Mult(x, y)
            return y

        def source(X, t, op=op):
            return torch.tensor(ms.source(X, t).ravel(), dtype=torch.float64, device=op.device)

        # (the operator works on torch's current stream: the torch stage combinations are ordered with its kernels)
        x0 = torch.tensor(ms.state(X, 0.0).ravel(), dtype=torch.float64, device=op.device)
        x, t = _rk4(mult, source, X, x0, steps, dt)
        torch.cuda.synchronize()
        xh = x.cpu().numpy().reshape(5, -1)
        op.close()
        errors.append(lp_errors_box(X, xh, ms, t, p=1))
        if level == 1:  # HIP time loop against the oracle's, state by state
            _, _, xo, _, _ = _oracle_run(level, steps, dt)
            scale = np.abs(xo).max(axis=1, keepdims=True)
            assert (np.abs(xh - xo) / scale).max() < 1e-10
    _check(_rates(errors))
