// VALU issue-rate microbenchmark: independent fp32 FMA chains, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ void k(float* out, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // fma with VGPR operands only
      x0 = fmaf(x0, x1, x2); x1 = fmaf(x1, x2, x3); x2 = fmaf(x2, x3, x4); x3 = fmaf(x3, x4, x5);
      x4 = fmaf(x4, x5, x6); x5 = fmaf(x5, x6, x7); x6 = fmaf(x6, x7, x0); x7 = fmaf(x7, x0, x1);
    } else if (MODE == 1) {  // fma with scalar operands (a, b in SGPRs)
      x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
      x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
    } else {  // add / mul (VOP2)
      x0 = x0 * x1; x1 = x1 + x2; x2 = x2 * x3; x3 = x3 + x4; x4 = x4 * x5; x5 = x5 + x6; x6 = x6 * x7; x7 = x7 + x0;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int MODE> void run(const char* name, int waves_per_simd) {
  const int iters = 20000, threads = 256, blocks = 256 * waves_per_simd;  // 256 CUs x (4 SIMDs x W waves) / 4 waves per block
  float* d; hipMalloc(&d, sizeof(float) * threads * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, threads>>>(d, 10, 1.0001f, 0.5f);
  hipEventRecord(e0); k<MODE><<<blocks, threads>>>(d, iters, 1.0001f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = 8.0 * iters * waves_per_simd;  // VALU wave-instructions per SIMD
  printf("%-28s W=%d  %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles @2.4GHz), %.1f TFLOP/s\n", name, waves_per_simd, ms,
         ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4, 8.0 * iters * 2 * 64.0 * 4 * blocks / (ms * 1e-3) / 1e12 * (MODE == 2 ? 0.5 : 1));
  hipFree(d);
}
int main() {
  for (int w : {1, 2, 4, 8}) { run<0>("fma vgpr", w); run<1>("fma sgpr operands", w); run<2>("mul/add", w); }
  return 0;
}
