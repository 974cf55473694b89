// layout probe of v_mfma_f64_16x16x4_f64: A[i][k] = 100 i + k, B[k][j] = (k == K0) * (j + 1): which (lane, reg) holds D[i][j]
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(double *out, int amode, int bmode) {
  const int l = threadIdx.x;
  // candidate operand layouts: mode 0: row/col = l % 16, k = l / 16 ; mode 1: row/col = l / 4, k = l % 4
  const int ai = amode ? l / 4 : l % 16, ak = amode ? l % 4 : l / 16;
  const int bj = bmode ? l / 4 : l % 16, bk = bmode ? l % 4 : l / 16;
  const double a = 1000.0 * ai + 10.0 * ak;     // A[i][k]
  const double b = (bk == 0 ? 1.0 : 0.0) * (bj + 1) + (bk == 1 ? 0.001 : 0.0) * (bj + 1);  // B[0][j] = j+1, B[1][j] = (j+1)/1000
  v4d acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; r++) out[l * 4 + r] = acc[r];
}
int main() {
  double *d; hipMalloc(&d, 256 * 8);
  for (int am = 0; am < 2; am++) for (int bm = 0; bm < 2; bm++) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, am, bm); hipDeviceSynchronize();
    double h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // D[i][j] = A[i][0] B[0][j] + A[i][1] B[1][j] = 1000 i (j+1) + (1000 i + 10)(j+1)/1000
    int ok0 = 1, ok1 = 1;
    for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
      const int j = l % 16;
      const int i0 = 4 * (l / 16) + r, i1 = (l / 16) + 4 * r;
      const double e0 = 1000.0 * i0 * (j + 1) + (1000.0 * i0 + 10.0) * (j + 1) / 1000.0;
      const double e1 = 1000.0 * i1 * (j + 1) + (1000.0 * i1 + 10.0) * (j + 1) / 1000.0;
      if (fabs(h[l * 4 + r] - e0) > 1e-9) ok0 = 0;
      if (fabs(h[l * 4 + r] - e1) > 1e-9) ok1 = 0;
    }
    printf("A mode %d B mode %d: D[4(l/16)+r][l%%16] %s, D[(l/16)+4r][l%%16] %s ; lane1 regs: %g %g %g %g ; lane16: %g %g\n", am, bm,
           ok0 ? "YES" : "no", ok1 ? "YES" : "no", h[4], h[5], h[6], h[7], h[64], h[65]);
  }
  return 0;
}
