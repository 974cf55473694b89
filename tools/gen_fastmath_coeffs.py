"""Coefficients of tps_amd/csrc/fastmath.hpp (fexp, flog), derived here with mpmath and checked:
Chebyshev-node interpolants on the reduced intervals, rounded to double, max error evaluated in 60-digit
arithmetic on a dense grid.  Run: python tools/gen_fastmath_coeffs.py"""
import mpmath as mp

mp.mp.dps = 60


def cheb_interp_monomial(f, a, b, deg):
    """monomial coefficients (mp) of the degree-`deg` interpolant of f at the Chebyshev nodes of [a, b]"""
    n = deg + 1
    xs = [(a + b) / 2 + (b - a) / 2 * mp.cos(mp.pi * (2 * k + 1) / (2 * n)) for k in range(n)]
    A = mp.matrix(n, n)
    rhs = mp.matrix(n, 1)
    for i, x in enumerate(xs):
        for j in range(n):
            A[i, j] = x ** j
        rhs[i] = f(x)
    return list(mp.lu_solve(A, rhs))


def horner(c, x):
    r = mp.mpf(0)
    for ck in reversed(c):
        r = r * x + ck
    return r


def report(name, c):
    print(f"// {name}")
    for k, ck in enumerate(c):
        print(f"  c{k} = {float(ck)!r}")


# ---- exp(r) on [-ln2/2, ln2/2], degree 11
a = mp.log(2) / 2
ce = cheb_interp_monomial(mp.exp, -a, a, 11)
ce_d = [mp.mpf(float(c)) for c in ce]
err = max(abs(horner(ce_d, x) / mp.exp(x) - 1) for x in mp.linspace(-a, a, 4001))
report("exp degree 11", ce)
print("exp: max relative error of the rounded polynomial", mp.nstr(err, 5), "=", mp.nstr(err / mp.mpf(2) ** -53, 4), "x 2^-53")

# ---- log: log(m) = 2 s + s R(z), s = (m-1)/(m+1), z = s^2, m in [sqrt(1/2), sqrt(2)); R(z) = z h(z)
smax = (mp.sqrt(2) - 1) / (mp.sqrt(2) + 1)
zmax = smax ** 2


def h(z):
    if z == 0:
        return mp.mpf(2) / 3
    s = mp.sqrt(z)
    return (mp.log((1 + s) / (1 - s)) - 2 * s) / (s * z)


cl = cheb_interp_monomial(h, mp.mpf(0), zmax, 6)
cl_d = [mp.mpf(float(c)) for c in cl]
worst = mp.mpf(0)
for s in mp.linspace(-smax, smax, 4001):
    if s == 0:
        continue
    z = s * s
    approx = 2 * s + s * z * horner(cl_d, z)
    exact = mp.log((1 + s) / (1 - s))
    worst = max(worst, abs(approx / exact - 1))
report("log: h(z), 7 coefficients (Lg1..Lg7)", cl)
print("log: max relative error of the rounded series", mp.nstr(worst, 5), "=", mp.nstr(worst / mp.mpf(2) ** -53, 4), "x 2^-53")
print("ln2_hi/lo (hi has 21 trailing zero bits: k*hi exact for |k| < 2^21)")
import struct
ln2 = mp.log(2)
hi = struct.unpack("<d", struct.pack("<Q", struct.unpack("<Q", struct.pack("<d", float(ln2)))[0] & ~((1 << 21) - 1)))[0]
lo = float(ln2 - mp.mpf(hi))
print(f"  ln2_hi = {hi!r}\n  ln2_lo = {lo!r}\n  log2e = {float(1 / ln2)!r}")
