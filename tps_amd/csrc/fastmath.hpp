// exp / log / sqrt / reciprocal of the point physics, written for FP64 VALU throughput on gfx950.
//
// The collision-integral fits of the argon transport (src/collision_integrals.cpp:53-201, called from
// src/gas_transport.cpp:206-489) cost 16 pow() = 16 exp + 16 log per point in the reference.  The
// device library's exp / log are 37 / 93 FP64 instructions (correctly rounded, every special case
// through extended-precision paths); at 214 transport evaluations per p = 3 hex that is where the
// reacting Mult spends its time.  The versions here are 23 / 30-36 instructions, accurate to < 2 ulp (exp) and < 4 ulp (log)
// (tests/test_gpu_fastmath.py), keep IEEE semantics for the special values the physics can produce
// (NaN propagates -- Check_NAN of the time loop depends on it --, log(0) = -inf, log(x < 0) = NaN,
// exp(+-inf)), and have no branches, so that independent evaluations interleave.
//
// Coefficients: tools/gen_fastmath_coeffs.py (mpmath; polynomial errors 0.15 and 0.04 x 2^-53).
#ifndef TPSRHS_FASTMATH_HPP_
#define TPSRHS_FASTMATH_HPP_

#include <hip/hip_runtime.h>

namespace tpsrhs {

// 1/x and sqrt(x): hardware seed (about 2^-23) plus two Newton / Goldschmidt steps, < 2 ulp.  Positive,
// normal arguments (densities, temperatures, squared lengths).
__device__ inline double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
__device__ inline double fast_sqrt(double x) {
  // x < 0 -> NaN (rsq of a negative number), as sqrt: a non-physical state (negative pressure or temperature) must
  // reach the reference's a-posteriori Check_NAN (src/M2ulPhyS.cpp:2463), not become a zero wave speed
  if (x == 0.0) return 0.0;
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  return fma(fma(-g, g, x), h, g);
}
// 1/sqrt(x), same scheme
__device__ inline double fast_rsqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  h = fma(h, r, h);
  g = fma(g, r, g);
  r = fma(-h, g, 0.5);
  h = fma(h, r, h);
  return h + h;
}

namespace fm {
constexpr double kLog2e = 1.4426950408889634;
constexpr double kLn2Hi = 0.6931471803691238;      // 21 trailing zero bits: k * hi is exact for |k| < 2^21
constexpr double kLn2Lo = 1.9082149292705877e-10;
}  // namespace fm

// exp(x).  x = k ln2 + r, |r| <= ln2/2; degree-11 polynomial; 2^k by v_ldexp_f64 (which rounds into the
// denormal range and saturates to 0 / inf by itself).
// CHECKED = false: for arguments known to be finite and of moderate size (|x| < 1000, e.g. a small multiple of the
// logarithm of a finite positive number); 0 / inf saturation is left to v_ldexp_f64, a NaN still comes out as a NaN.
template <bool CHECKED = true>
__device__ inline double fexp(double x) {
  const double k = __builtin_rint(x * fm::kLog2e);
  double r = fma(k, -fm::kLn2Hi, x);
  r = fma(k, -fm::kLn2Lo, r);
  double p = 2.5110037605963777e-08;
  p = fma(p, r, 2.763263963904103e-07);
  p = fma(p, r, 2.755724091857897e-06);
  p = fma(p, r, 2.4801485482328494e-05);
  p = fma(p, r, 0.00019841269890047113);
  p = fma(p, r, 0.0013888888952314775);
  p = fma(p, r, 0.008333333333319601);
  p = fma(p, r, 0.0416666666664881);
  p = fma(p, r, 0.1666666666666668);
  p = fma(p, r, 0.5000000000000019);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  double e = __builtin_amdgcn_ldexp(p, static_cast<int>(k));
  if (!CHECKED) return e;
  // beyond +-1000 (the reduction above is exact far past that) the result is 0 or inf anyway; a NaN
  // fails the comparison and falls through as the NaN the polynomial made of it
  if (!(fabs(x) < 1000.0)) e = (x < 0.0) ? 0.0 : x * __builtin_inf();
  return e;
}

// log(x) for finite x > 0 (denormals included: v_frexp_* normalise them).  m in [sqrt(1/2), sqrt(2)),
// s = (m - 1) / (m + 1), log m = 2 s + s z h(z) with z = s^2 (the classical atanh series, 7 terms).
__device__ inline double flog_pos(double x) {
  double m = __builtin_amdgcn_frexp_mant(x);  // [1/2, 1)
  int e = __builtin_amdgcn_frexp_exp(x);
  const bool low = m < 0.70710678118654752;
  m = __builtin_amdgcn_ldexp(m, low ? 1 : 0);
  e -= low ? 1 : 0;
  const double f = m - 1.0, d = m + 1.0;
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  double s = f * r;
  s = fma(fma(-d, s, f), r, s);
  const double z = s * s;
  double h = 0.14616449685043406;
  h = fma(h, z, 0.15331721600556042);
  h = fma(h, z, 0.18182889125261723);
  h = fma(h, z, 0.2222221113479508);
  h = fma(h, z, 0.28571428625975487);
  h = fma(h, z, 0.39999999999899505);
  h = fma(h, z, 0.666666666666667);
  const double lm = s * fma(h, z, 2.0);
  const double dk = static_cast<double>(e);
  return fma(dk, fm::kLn2Hi, fma(dk, fm::kLn2Lo, lm));
}
// N independent evaluations in lock step: the same operations per element as fexp<false> / flog_pos (bit-identical
// results), written so that the N dependent chains are interleaved in program order.  A dependent FP64 instruction
// issues ~8 cycles after its producer, an independent one after 4 (tools/microbench/fp64_issue.hip): one Horner chain
// alone keeps a SIMD half idle whenever the other resident wave is not in a VALU phase, and the heavy kernels run two
// waves per SIMD.  (Round 2 evaluated the collision fits one after the other: 11 + 7 dependent FMAs per exp / log.)
// the scheduler otherwise un-interleaves the rows again to save registers (seen in the ISA): nothing crosses a row
#define TPSRHS_ROW_BARRIER() __builtin_amdgcn_sched_barrier(0)
template <int N>
__device__ inline void fexp_n(const double (&x)[N], double (&e)[N]) {
  double k[N], r[N], p[N];
#pragma unroll
  for (int i = 0; i < N; i++) k[i] = __builtin_rint(x[i] * fm::kLog2e);
#pragma unroll
  for (int i = 0; i < N; i++) r[i] = fma(k[i], -fm::kLn2Hi, x[i]);
  TPSRHS_ROW_BARRIER();
#pragma unroll
  for (int i = 0; i < N; i++) r[i] = fma(k[i], -fm::kLn2Lo, r[i]);
  TPSRHS_ROW_BARRIER();
  constexpr double c[12] = {2.5110037605963777e-08, 2.763263963904103e-07, 2.755724091857897e-06, 2.4801485482328494e-05,
                            0.00019841269890047113, 0.0013888888952314775, 0.008333333333319601, 0.0416666666664881,
                            0.1666666666666668,     0.5000000000000019,    1.0,                  1.0};
#pragma unroll
  for (int i = 0; i < N; i++) p[i] = c[0];
#pragma unroll
  for (int j = 1; j < 12; j++) {
#pragma unroll
    for (int i = 0; i < N; i++) p[i] = fma(p[i], r[i], c[j]);
    TPSRHS_ROW_BARRIER();
  }
#pragma unroll
  for (int i = 0; i < N; i++) e[i] = __builtin_amdgcn_ldexp(p[i], static_cast<int>(k[i]));
}
template <int N>
__device__ inline void flog_pos_n(const double (&x)[N], double (&l)[N]) {
  double m[N], f[N], d[N], r[N], s[N], z[N], h[N];
  int e[N];
#pragma unroll
  for (int i = 0; i < N; i++) {
    m[i] = __builtin_amdgcn_frexp_mant(x[i]);
    e[i] = __builtin_amdgcn_frexp_exp(x[i]);
    const bool low = m[i] < 0.70710678118654752;
    m[i] = __builtin_amdgcn_ldexp(m[i], low ? 1 : 0);
    e[i] -= low ? 1 : 0;
    f[i] = m[i] - 1.0;
    d[i] = m[i] + 1.0;
  }
#pragma unroll
  for (int i = 0; i < N; i++) r[i] = __builtin_amdgcn_rcp(d[i]);
#pragma unroll
  for (int t = 0; t < 2; t++) {
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = fma(fma(-d[i], r[i], 1.0), r[i], r[i]);
    TPSRHS_ROW_BARRIER();
  }
#pragma unroll
  for (int i = 0; i < N; i++) s[i] = f[i] * r[i];
  TPSRHS_ROW_BARRIER();
#pragma unroll
  for (int i = 0; i < N; i++) s[i] = fma(fma(-d[i], s[i], f[i]), r[i], s[i]);
  TPSRHS_ROW_BARRIER();
#pragma unroll
  for (int i = 0; i < N; i++) z[i] = s[i] * s[i];
  TPSRHS_ROW_BARRIER();
  constexpr double c[7] = {0.14616449685043406, 0.15331721600556042, 0.18182889125261723, 0.2222221113479508,
                           0.28571428625975487, 0.39999999999899505, 0.666666666666667};
#pragma unroll
  for (int i = 0; i < N; i++) h[i] = c[0];
#pragma unroll
  for (int j = 1; j < 7; j++) {
#pragma unroll
    for (int i = 0; i < N; i++) h[i] = fma(h[i], z[i], c[j]);
    TPSRHS_ROW_BARRIER();
  }
#pragma unroll
  for (int i = 0; i < N; i++) {
    const double lm = s[i] * fma(h[i], z[i], 2.0);
    const double dk = static_cast<double>(e[i]);
    l[i] = fma(dk, fm::kLn2Hi, fma(dk, fm::kLn2Lo, lm));
  }
}

// log(x), any x: log(0) = -inf, log(x < 0) = NaN, log(inf) = inf, NaN -> NaN.  The special values come
// from the single-precision hardware logarithm of the same argument (identical IEEE special cases).
__device__ inline double flog(double x) {
  const double l = flog_pos(x);
  // class bits: 0x080 +denormal, 0x100 +normal
  const bool regular = __builtin_amdgcn_class(x, 0x180);
  return regular ? l : static_cast<double>(__builtin_amdgcn_logf(static_cast<float>(x)));
}

}  // namespace tpsrhs
#endif
