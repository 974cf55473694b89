// development probe: the state-only transport closure of the metric's physics alone in a kernel -- how many registers does
// it need by itself at 2 / 3 / 4 waves per SIMD?   hipcc --offload-arch=gfx950 -O3 -std=c++17 -c _probe/closure.hip
#include "../kernels.hpp"
#include "../physics_plasma.hpp"
using namespace tpsrhs;
typedef PlasmaPhys<3, 3, 3, true, false, TRANSPORT_ARGON_MINIMAL> PH;
template <int W>
__global__ __launch_bounds__(64, W) void k_closure(PH::KArg k, const double *__restrict__ U, double *__restrict__ out, int n) {
  PH::PRef p = PH::pref(k);
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  double u[PH::NEQ];
#pragma unroll
  for (int eq = 0; eq < PH::NEQ; eq++) u[eq] = U[eq * n + i];
  PH::ViscCoef c;
  PH::visc_point_coeffs(p, u, true, c);
  const double *v = reinterpret_cast<const double *>(&c);
#pragma unroll
  for (int j = 0; j < static_cast<int>(sizeof(c) / sizeof(double)); j++) out[j * n + i] = v[j];
}
template __global__ void k_closure<1>(PH::KArg, const double *, double *, int);
template __global__ void k_closure<2>(PH::KArg, const double *, double *, int);
template __global__ void k_closure<3>(PH::KArg, const double *, double *, int);
template __global__ void k_closure<4>(PH::KArg, const double *, double *, int);
