// TEST INFRASTRUCTURE: a stand-in for <hip/hip_runtime.h> that lets g++ compile the DEVICE point physics of
// tps_amd/csrc (fastmath.hpp, physics_dryair.hpp, physics_plasma.hpp -- plain templated C++ behind __device__) for the
// HOST, so that AddressSanitizer / UndefinedBehaviorSanitizer can run it (tests/test_host_sanitize.py).  GPU sanitizers
// are not available on the MI355X pool; the point physics has no GPU-only semantics besides the builtins mapped here.
#ifndef TPSRHS_HOST_SHIM_HIP_RUNTIME_H_
#define TPSRHS_HOST_SHIM_HIP_RUNTIME_H_
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>

#define __device__
#define __host__
#define __global__
#define __constant__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__ __restrict

struct HostShimIdx {
  unsigned x = 0, y = 0, z = 0;
};
static HostShimIdx threadIdx, blockIdx;

using std::max;
using std::min;

// v_rcp_f64 / v_rsq_f64 seeds: the hardware's are good to ~2^-23; the refinement steps of fastmath.hpp then behave as on
// the device (a float-accurate seed, two Newton steps)
static inline double hostshim_rcp_wide(double x) {  // arguments outside the float range: scale through frexp
  int e;
  const double m = std::frexp(x, &e);
  return std::ldexp(static_cast<double>(1.0f / static_cast<float>(m)), -e);
}
#define __builtin_amdgcn_rcp(x) hostshim_rcp_wide(x)
static inline double hostshim_rsq(double x) {
  if (!(x > 0.0)) return (x == 0.0) ? INFINITY : NAN;
  int e;
  double m = std::frexp(x, &e);
  if (e & 1) {
    m *= 2.0;
    e -= 1;
  }
  return std::ldexp(static_cast<double>(1.0f / std::sqrt(static_cast<float>(m))), -e / 2);
}
#define __builtin_amdgcn_rsq(x) hostshim_rsq(x)
#define __builtin_amdgcn_ldexp(x, k) std::ldexp((x), (k))
static inline double hostshim_frexp_mant(double x) {
  int e;
  return std::frexp(x, &e);
}
static inline int hostshim_frexp_exp(double x) {
  int e;
  std::frexp(x, &e);
  return e;
}
#define __builtin_amdgcn_frexp_mant(x) hostshim_frexp_mant(x)
#define __builtin_amdgcn_frexp_exp(x) hostshim_frexp_exp(x)
// v_cmp_class_f64 with mask 0x180: +denormal | +normal
static inline bool hostshim_class(double x, int mask) {
  (void)mask;
  return x > 0.0 && std::isfinite(x);
}
#define __builtin_amdgcn_class(x, m) hostshim_class((x), (m))
#define __builtin_amdgcn_logf(x) std::log2((x))  // v_log_f32 is log2; only its special values are used
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_readfirstlane(x) (x)
#endif
