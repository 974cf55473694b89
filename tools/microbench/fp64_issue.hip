// FP64 VALU issue / latency facts of gfx950 that the plasma point physics is priced with
// (DESIGN.md, section 5): cycles per wave-instruction for dependent and independent chains, at one and
// two waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 fp64_issue.hip -o fp64_issue ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ __launch_bounds__(64) void k(double *out, long long *cyc, int iters) {
  double a = out[threadIdx.x], b = a + 1.0, c = a + 2.0, d = a + 3.0, m = 1.0000001, n = 1e-9;
  int e = 1;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    if (KIND == 0) { REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(n));) }
    if (KIND == 1) { REP64(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(n));) }
    if (KIND == 2) { REP64(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(m));) }
    if (KIND == 3) { REP64(asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(n));) }
    if (KIND == 4) { REP64(asm volatile("v_rcp_f64 %0, %0" : "+v"(a));) }
    if (KIND == 5) { REP64(asm volatile("v_rcp_f64 %0, %4\n v_rcp_f64 %1, %4\n v_rcp_f64 %2, %4\n v_rcp_f64 %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
    if (KIND == 6) { REP64(asm volatile("v_ldexp_f64 %0, %4, %5\n v_ldexp_f64 %1, %4, %5\n v_ldexp_f64 %2, %4, %5\n v_ldexp_f64 %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(e));) }
    if (KIND == 7) { REP64(asm volatile("v_frexp_mant_f64 %0, %4\n v_frexp_mant_f64 %1, %4\n v_frexp_mant_f64 %2, %4\n v_frexp_mant_f64 %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
    if (KIND == 8) { REP64(asm volatile("v_rsq_f64 %0, %4\n v_rsq_f64 %1, %4\n v_rsq_f64 %2, %4\n v_rsq_f64 %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
    if (KIND == 9) { REP64(asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %0, 4\n v_readlane_b32 s22, %0, 5\n v_readlane_b32 s23, %0, 6" : : "v"(e) : "s20", "s21", "s22", "s23");) }
    if (KIND == 10) { REP64(asm volatile("v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %1, %2, %0, vcc\n v_cndmask_b32 %2, %0, %1, vcc\n v_cndmask_b32 %0, %2, %1, vcc" : "+v"(e) : "v"(e + 1), "v"(e + 2) : "vcc");) }
    if (KIND == 11) { REP64(asm volatile("v_cmp_class_f64 vcc, %0, %1\n v_cmp_gt_f64 vcc, %0, %2\n v_cmp_class_f64 vcc, %0, %1\n v_cmp_gt_f64 vcc, %0, %2" : : "v"(a), "v"(e), "v"(m) : "vcc");) }
    if (KIND == 12) { int f = e, g = e, h = e, i = e; REP64(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4" : "+v"(f), "+v"(g), "+v"(h), "+v"(i) : "v"(e));) e += f + g + h + i; }
    if (KIND == 13) { REP64(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(a), "+v"(b) : "v"(m), "v"(n));) }
    if (KIND == 14) { REP64(asm volatile("v_fma_f64 %0, %0, s[20:21], %1" : "+v"(a) : "v"(n) : "s20", "s21");) }
    if (KIND == 15) { int f = e; REP64(asm volatile("v_cvt_f64_i32 %0, %5\n v_cvt_i32_f64 %4, %1\n v_cvt_f64_i32 %2, %5\n v_cvt_i32_f64 %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(f) : "v"(e));) e += f; }
    if (KIND == 16) { REP64(asm volatile("v_rndne_f64 %0, %4\n v_rndne_f64 %1, %4\n v_rndne_f64 %2, %4\n v_rndne_f64 %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
    if (KIND == 17) { REP64(asm volatile("v_mov_b64 %0, %4\n v_mov_b64 %1, %4\n v_mov_b64 %2, %4\n v_mov_b64 %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));) }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + e;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

struct Case { const char *name; int kind; int per_rep; };

int main() {
  const Case cases[] = {
      {"v_fma_f64 dependent chain", 0, 1}, {"v_fma_f64 2 chains", 13, 2}, {"v_fma_f64 4 chains", 1, 4},
      {"v_fma_f64 dep, SGPR operand", 14, 1},
      {"v_mul_f64 dependent", 2, 1}, {"v_add_f64 dependent", 3, 1}, {"v_rcp_f64 dependent", 4, 1},
      {"v_rcp_f64 independent", 5, 4}, {"v_rsq_f64 independent", 8, 4}, {"v_ldexp_f64 independent", 6, 4},
      {"v_frexp_mant_f64 independent", 7, 4}, {"v_rndne_f64 independent", 16, 4}, {"v_cvt f64<->i32", 15, 4},
      {"v_cmp_class/gt_f64", 11, 4}, {"v_cndmask_b32", 10, 4}, {"v_mov_b32", 12, 4}, {"v_mov_b64", 17, 4},
      {"v_readlane_b32", 9, 4}};
  double *out; long long *cyc;
  const int maxblocks = 256 * 4 * 4;  // the largest grid below
  hipMalloc(&out, maxblocks * 64 * sizeof(double)); hipMalloc(&cyc, maxblocks * sizeof(long long));
  hipMemset(out, 0, maxblocks * 64 * sizeof(double));
  const int iters = 200;
  for (const Case &c : cases) {
    for (int wps = 1; wps <= 4; wps *= 2) {  // waves per SIMD: blocks = 256 CUs x 4 SIMDs x wps
      const int blocks = 256 * 4 * wps;
      auto launch = [&](int kind) {
#define L(K) case K: hipLaunchKernelGGL(k<K>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters); break;
        switch (kind) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14) L(15) L(16) L(17) }
      };
      launch(c.kind); hipDeviceSynchronize();
      launch(c.kind); hipDeviceSynchronize();
      std::vector<long long> h(blocks);
      hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
      double avg = 0; for (long long v : h) avg += double(v); avg /= blocks;
      // s_memtime ticks = shader cycles; cycles per wave-instruction as seen by ONE wave
      printf("%-34s waves/SIMD %d : %7.2f cycles per instruction per wave  (SIMD issue interval %6.2f)\n", c.name, wps,
             avg / (double(iters) * 64 * c.per_rep), avg / (double(iters) * 64 * c.per_rep) / wps);
    }
  }
  return 0;
}
