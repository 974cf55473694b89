"""Manufactured solution of the compressible Navier-Stokes equations on a periodic box, differentiated
symbolically (sympy): the exact right-hand side  dU/dt = -div(F_c(U) - F_v(U, grad U))  of a smooth periodic
state, with the reference's closures (gamma-law gas, Sutherland viscosity, bulk viscosity multiplier, Prandtl
number: SURVEY.md appendix A.2, A.4, A.5).  Independent of the oracle and of the kernels: the order-of-accuracy
tests compare both with the PDE itself (the role of the reference's MASA-based mms.*.test, whose library is
not available here)."""
import functools

import numpy as np

GAMMA, RG, C1, S0, PR = 1.4, 287.058, 1.458e-6, 110.4, 0.71


@functools.lru_cache(maxsize=None)
def _build(dim, viscous, visc_mult, bulk_mult, lengths):
    import sympy as sp

    X = sp.symbols("x y z")[:dim]
    k = [2 * sp.pi / L for L in lengths]
    # smooth periodic primitives, bounded away from zero
    rho = 1.2 + 0.10 * sp.sin(k[0] * X[0]) * sp.cos(k[1] * X[1]) + (0.05 * sp.sin(k[2] * X[2]) if dim == 3 else 0)
    vel = [30.0 + 8.0 * sp.sin(k[0] * X[0] + 0.3) * sp.cos(k[1] * X[1]),
           -12.0 + 6.0 * sp.cos(k[0] * X[0]) * sp.sin(k[1] * X[1] + 0.5)]
    if dim == 3:
        vel[0] += 3.0 * sp.cos(k[2] * X[2])
        vel.append(7.0 + 5.0 * sp.sin(k[2] * X[2] + 0.2) * sp.cos(k[0] * X[0]))
    p = 101300.0 * (1 + 0.04 * sp.cos(k[0] * X[0] - 0.4) * sp.sin(k[1] * X[1]) + (0.02 * sp.cos(k[2] * X[2]) if dim == 3 else 0))
    T = p / (rho * RG)
    E = p / (GAMMA - 1) + rho * sum(v * v for v in vel) / 2
    U = [rho] + [rho * v for v in vel] + [E]
    F = [[U[0] * vel[d] for d in range(dim)]]
    for i in range(dim):
        F.append([U[1 + i] * vel[d] + (p if i == d else 0) for d in range(dim)])
    F.append([vel[d] * (E + p) for d in range(dim)])
    if viscous:
        mu = C1 * visc_mult * T ** sp.Rational(3, 2) / (T + S0)
        mub = bulk_mult * mu - sp.Rational(2, 3) * mu
        kap = mu * GAMMA * RG / ((GAMMA - 1) * PR)
        div = sum(sp.diff(vel[d], X[d]) for d in range(dim))
        tau = [[mu * (sp.diff(vel[i], X[j]) + sp.diff(vel[j], X[i])) + (mub * div if i == j else 0) for j in range(dim)]
               for i in range(dim)]
        for i in range(dim):
            for d in range(dim):
                F[1 + i][d] -= tau[i][d]
        for d in range(dim):
            F[dim + 1][d] -= sum(tau[i][d] * vel[i] for i in range(dim)) + kap * sp.diff(T, X[d])
    rhs = [-sum(sp.diff(F[eq][d], X[d]) for d in range(dim)) for eq in range(dim + 2)]
    return sp.lambdify(X, U, "numpy"), sp.lambdify(X, rhs, "numpy")


def manufactured(X, viscous=True, visc_mult=1.0, bulk_mult=0.0, lengths=(1.0, 1.0, 1.0)):
    """X: node coordinates (dim, N) -> (U, exact dU/dt), both (dim + 2, N)"""
    dim = X.shape[0]
    fu, fr = _build(dim, bool(viscous), float(visc_mult), float(bulk_mult), tuple(lengths[:dim]))
    bc = np.zeros(X.shape[1])
    U = np.array([np.asarray(v, dtype=np.float64) + bc for v in fu(*X)])
    R = np.array([np.asarray(v, dtype=np.float64) + bc for v in fr(*X)])
    return U, R


def observed_order(err_coarse, err_fine):
    return np.log2(np.asarray(err_coarse) / np.asarray(err_fine))
