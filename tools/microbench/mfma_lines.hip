// The MFMA question of the north star ("MFMA only for the dense element-local basis x DOF contractions"), measured:
// the 1-D operator stack of a p = 3 hex line stage -- differentiation (4 rows), end-point values (2), interpolation
// to the 5 face-quadrature abscissae (5): 11 functionals of the 4 nodal values of a line -- applied to the 16
// x-lines of F nodal fields resident in LDS,
//   (a) as the kernels do it: one lane per line, coefficients as scalar operands, 44 v_fma_f64 per lane;
//   (b) with v_mfma_f64_16x16x4_f64: A = the stack padded to 16 rows x K = 4, B = the 4 x 16 line values of a field,
//       one MFMA per field (2048 flops issued for 1408 useful ones).
// Same LDS-resident data, same outputs (checked against each other), time per field from s_memtime, all CUs busy
// with W waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 mfma_lines.hip -o mfma_lines
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int NFLD = 16, NROW = 11;
__constant__ double c_S[16 * 4];  // [row][i], rows 11..15 zero

template <int MODE>
__global__ __launch_bounds__(64) void k_lines(const double *__restrict__ in, double *__restrict__ out, long long *cyc,
                                              int iters) {
  __shared__ double sU[NFLD * 64];        // [field][k][j][i]
  __shared__ double sO[NFLD * 16 * 16];   // [field][row][line]
  const int tid = threadIdx.x;
  for (int f = 0; f < NFLD; f++) sU[f * 64 + tid] = in[(blockIdx.x % 64) * NFLD * 64 + f * 64 + tid];
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {  // line stage: lane = (field % 4, line); four fields per round
      const int line = tid & 15, fsub = tid >> 4;
#pragma unroll 1
      for (int f0 = 0; f0 < NFLD; f0 += 4) {
        const double *u = &sU[(f0 + fsub) * 64 + 4 * line];
        const double u0 = u[0], u1 = u[1], u2 = u[2], u3 = u[3];
        double *o = &sO[(f0 + fsub) * 256 + line];
#pragma unroll
        for (int r = 0; r < NROW; r++)
          o[r * 16] = c_S[r * 4 + 0] * u0 + c_S[r * 4 + 1] * u1 + c_S[r * 4 + 2] * u2 + c_S[r * 4 + 3] * u3;
      }
    } else {  // MFMA: lane l supplies A[l%16][l/16] and B[l/16][l%16], receives D[l/16 + 4 r][l%16]
      const int col = tid & 15, kk = tid >> 4;
      const double a = c_S[col * 4 + kk];
#pragma unroll 1
      for (int f = 0; f < NFLD; f++) {
        const double b = sU[f * 64 + 4 * col + kk];
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        double *o = &sO[f * 256 + kk * 16 + col];  // register r holds row kk + 4 r (tools/microbench/mfma_probe.hip)
        o[0] = acc[0];
        o[64] = acc[1];
        if (kk + 8 < NROW) o[128] = acc[2];  // rows >= 11 are padding
        if (kk + 12 < NROW) o[192] = acc[3];
      }
    }
    __syncthreads();
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  if (blockIdx.x < 64)
    for (int f = 0; f < NFLD; f++)
      for (int r = 0; r < NROW; r++)
        if (tid < 16) out[((blockIdx.x * NFLD + f) * NROW + r) * 16 + tid] = sO[f * 256 + r * 16 + tid];
}

int main() {
  // the p = 3 stack on Gauss-Legendre nodes: D (4x4), b0, b1, B (5x4)
  double x[4] = {0.0694318442029737, 0.3300094782075719, 0.6699905217924281, 0.9305681557970263};
  double xq[5] = {0.0469100770306680, 0.2307653449471585, 0.5, 0.7692346550528415, 0.9530899229693320};
  auto lag = [&](int a, double t) { double v = 1; for (int j = 0; j < 4; j++) if (j != a) v *= (t - x[j]) / (x[a] - x[j]); return v; };
  auto dlag = [&](int a, double t) { double s = 0; for (int i = 0; i < 4; i++) { if (i == a) continue; double v = 1 / (x[a] - x[i]);
      for (int j = 0; j < 4; j++) if (j != a && j != i) v *= (t - x[j]) / (x[a] - x[j]); s += v; } return s; };
  std::vector<double> S(64, 0.0);
  for (int a = 0; a < 4; a++) {
    for (int i = 0; i < 4; i++) S[i * 4 + a] = dlag(a, x[i]);
    S[4 * 4 + a] = lag(a, 0.0);
    S[5 * 4 + a] = lag(a, 1.0);
    for (int q = 0; q < 5; q++) S[(6 + q) * 4 + a] = lag(a, xq[q]);
  }
  hipMemcpyToSymbol(HIP_SYMBOL(c_S), S.data(), sizeof(double) * 64);
  std::vector<double> hin(64 * NFLD * 64);
  for (size_t i = 0; i < hin.size(); i++) hin[i] = std::sin(0.37 * i) + 0.01 * (i % 17);
  double *in, *out[2];
  long long *cyc;
  const int maxblocks = 256 * 4 * 4;
  hipMalloc(&in, hin.size() * 8);
  hipMemcpy(in, hin.data(), hin.size() * 8, hipMemcpyHostToDevice);
  for (auto &o : out) hipMalloc(&o, 64 * NFLD * NROW * 16 * 8);
  hipMalloc(&cyc, maxblocks * sizeof(long long));
  const int iters = 2000;
  double per_field[2][3];
  for (int wps = 1, wi = 0; wps <= 4; wps *= 2, wi++) {
    const int blocks = 256 * 4 * wps;
    for (int mode = 0; mode < 2; mode++) {
      for (int rep = 0; rep < 2; rep++) {
        if (mode == 0)
          hipLaunchKernelGGL(k_lines<0>, dim3(blocks), dim3(64), 0, 0, in, out[0], cyc, iters);
        else
          hipLaunchKernelGGL(k_lines<1>, dim3(blocks), dim3(64), 0, 0, in, out[1], cyc, iters);
        hipDeviceSynchronize();
      }
      std::vector<long long> h(blocks);
      hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
      double avg = 0;
      for (long long v : h) avg += double(v);
      avg /= blocks;
      per_field[mode][wi] = avg / iters / NFLD;
      printf("%-28s waves/SIMD %d : %7.1f cycles per field per wave, %6.1f per SIMD\n",
             mode == 0 ? "line stage (v_fma_f64)" : "v_mfma_f64_16x16x4_f64", wps, per_field[mode][wi], per_field[mode][wi] / wps);
    }
  }
  std::vector<double> a(64 * NFLD * NROW * 16), b(a.size());
  hipMemcpy(a.data(), out[0], a.size() * 8, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), out[1], b.size() * 8, hipMemcpyDeviceToHost);
  double err = 0, mx = 0;
  for (size_t i = 0; i < a.size(); i++) { err = std::fmax(err, std::fabs(a[i] - b[i])); mx = std::fmax(mx, std::fabs(a[i])); }
  printf("max |line stage - MFMA| = %.3e (max |value| %.3e)\n", err, mx);
  printf("MFMA / line-stage time per field: %.2f (1 wave/SIMD), %.2f (2), %.2f (4)\n", per_field[1][0] / per_field[0][0],
         per_field[1][1] / per_field[0][1], per_field[1][2] / per_field[0][2]);
  return err < 1e-12 * mx ? 0 : 1;
}
