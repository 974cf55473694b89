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


@functools.lru_cache(maxsize=None)
def _build_axisym(visc_mult, bulk_mult, lz):
    """Compressible Navier-Stokes in cylindrical coordinates (r, z) with swirl; state order of the reference's
    axisymmetric formulation: rho, rho u_r, rho u_z, rho u_theta, rho E."""
    import sympy as sp

    r, z = sp.symbols("r z", positive=True)
    k = 2 * sp.pi / lz
    rho = 1.2 + 0.10 * sp.sin(k * z) * sp.cos(9.0 * r) + 0.05 * sp.sin(14.0 * r)
    ur = 4.0 * sp.sin(11.0 * r) * sp.cos(k * z + 0.3)
    uz = 30.0 + 8.0 * sp.cos(8.0 * r) * sp.sin(k * z)
    ut = 6.0 * sp.sin(7.0 * r + 0.4) * (1 + 0.3 * sp.cos(k * z))
    p = 101300.0 * (1 + 0.04 * sp.cos(10.0 * r - 0.4) * sp.sin(k * z))
    T = p / (rho * RG)
    E = p / (GAMMA - 1) + rho * (ur * ur + uz * uz + ut * ut) / 2
    mu = C1 * visc_mult * T ** sp.Rational(3, 2) / (T + S0)
    mub = bulk_mult * mu - sp.Rational(2, 3) * mu
    kap = mu * GAMMA * RG / ((GAMMA - 1) * PR)
    div = sp.diff(r * ur, r) / r + sp.diff(uz, z)
    trr = 2 * mu * sp.diff(ur, r) + mub * div
    tzz = 2 * mu * sp.diff(uz, z) + mub * div
    ttt = 2 * mu * ur / r + mub * div
    trz = mu * (sp.diff(ur, z) + sp.diff(uz, r))
    trt = mu * (sp.diff(ut, r) - ut / r)
    tzt = mu * sp.diff(ut, z)

    def dv(fr, fz):  # divergence of an (r, z) flux in cylindrical coordinates
        return sp.diff(r * fr, r) / r + sp.diff(fz, z)

    rhs = [-dv(rho * ur, rho * uz),
           -dv(rho * ur * ur + p - trr, rho * ur * uz - trz) + (p - ttt + rho * ut * ut) / r,
           -dv(rho * ur * uz - trz, rho * uz * uz + p - tzz),
           -dv(rho * ur * ut - trt, rho * uz * ut - tzt) - (rho * ur * ut - trt) / r,
           -dv(ur * (E + p) - (trr * ur + trz * uz + trt * ut) - kap * sp.diff(T, r),
               uz * (E + p) - (trz * ur + tzz * uz + tzt * ut) - kap * sp.diff(T, z))]
    U = [rho, rho * ur, rho * uz, rho * ut, E]
    return sp.lambdify((r, z), U, "numpy"), sp.lambdify((r, z), rhs, "numpy")


def manufactured_axisym(X, visc_mult=1.0, bulk_mult=0.0, lz=1.0):
    """X: (r, z) node coordinates (2, N) -> (U, exact dU/dt), both (5, N)"""
    fu, fr = _build_axisym(float(visc_mult), float(bulk_mult), float(lz))
    bc = np.zeros(X.shape[1])
    U = np.array([np.asarray(v, dtype=np.float64) + bc for v in fu(X[0], X[1])])
    R = np.array([np.asarray(v, dtype=np.float64) + bc for v in fr(X[0], X[1])])
    return U, R


@functools.lru_cache(maxsize=None)
def _build_ternary(dim, lengths):
    """Inviscid, frozen, single-temperature ambipolar ternary plasma (Ar.+1, E, Ar): PerfectMixture closure of
    SURVEY.md appendix A.3 -- n_e = n_i, p = R (n_i + n_e + n_B) T, rho E = sum n c_v T + n_i E_f + rho |u|^2 / 2 --
    state rho, rho u, rho E, rho Y_ion."""
    import sympy as sp

    R_U = 8.3144598
    m_ar, m_e, e_form, cv = 39.948e-3, 5.4858e-7, 1520571.3883, 1.5 * R_U
    m_i = m_ar - m_e
    X = sp.symbols("x y z")[:dim]
    k = [2 * sp.pi / L for L in lengths]
    T = 8000.0 * (1 + 0.10 * sp.sin(k[0] * X[0]) * sp.cos(k[1] * X[1]))
    nB = 1.4 * (1 + 0.08 * sp.cos(k[0] * X[0] + 0.2) * sp.sin(k[1] * X[1]) + (0.04 * sp.sin(k[2] * X[2]) if dim == 3 else 0))
    ni = 2.0e-3 * (1 + 0.30 * sp.sin(k[0] * X[0] - 0.5) * sp.sin(k[1] * X[1] + 0.1))
    vel = [300.0 + 60.0 * sp.sin(k[0] * X[0] + 0.3) * sp.cos(k[1] * X[1]), -100.0 + 40.0 * sp.cos(k[0] * X[0]) * sp.sin(k[1] * X[1])]
    if dim == 3:
        vel.append(50.0 + 30.0 * sp.sin(k[2] * X[2] + 0.2) * sp.cos(k[0] * X[0]))
    ne = ni
    rho = m_i * ni + m_e * ne + m_ar * nB
    p = R_U * (ni + ne + nB) * T
    E = cv * (ni + ne + nB) * T + e_form * ni + rho * sum(v * v for v in vel) / 2
    U = [rho] + [rho * v for v in vel] + [E, m_i * ni]
    rhs = [-sum(sp.diff(U[0] * vel[d], X[d]) for d in range(dim))]
    for i in range(dim):
        rhs.append(-sum(sp.diff(U[1 + i] * vel[d] + (p if i == d else 0), X[d]) for d in range(dim)))
    rhs.append(-sum(sp.diff(vel[d] * (E + p), X[d]) for d in range(dim)))
    rhs.append(-sum(sp.diff(m_i * ni * vel[d], X[d]) for d in range(dim)))
    return sp.lambdify(X, U, "numpy"), sp.lambdify(X, rhs, "numpy")


def manufactured_ternary(X, lengths=(1.0, 1.0, 1.0)):
    dim = X.shape[0]
    fu, fr = _build_ternary(dim, tuple(lengths[:dim]))
    bc = np.zeros(X.shape[1])
    U = np.array([np.asarray(v, dtype=np.float64) + bc for v in fu(*X)])
    R = np.array([np.asarray(v, dtype=np.float64) + bc for v in fr(*X)])
    return U, R
