// How many cycles does a wave64 VALU instruction cost as a function of its VGPR source operands?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(float* out, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  const float y = 1.0f + 1e-7f * threadIdx.x, z = 1e-9f * threadIdx.x;  // per-lane "constants" living in VGPRs
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {        // 3 VGPR sources, independent chains
      x0 = fmaf(x0, y, z); x1 = fmaf(x1, y, z); x2 = fmaf(x2, y, z); x3 = fmaf(x3, y, z);
      x4 = fmaf(x4, y, z); x5 = fmaf(x5, y, z); x6 = fmaf(x6, y, z); x7 = fmaf(x7, y, z);
    } else if (MODE == 1) { // 2 VGPR sources + 1 SGPR
      x0 = fmaf(x0, y, b); x1 = fmaf(x1, y, b); x2 = fmaf(x2, y, b); x3 = fmaf(x3, y, b);
      x4 = fmaf(x4, y, b); x5 = fmaf(x5, y, b); x6 = fmaf(x6, y, b); x7 = fmaf(x7, y, b);
    } else if (MODE == 2) { // 1 VGPR source + 2 SGPR
      x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
      x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
    } else if (MODE == 3) { // mul, 2 VGPR sources
      x0 *= y; x1 *= y; x2 *= y; x3 *= y; x4 *= y; x5 *= y; x6 *= y; x7 *= y;
    } else {                // mul, 1 VGPR + SGPR
      x0 *= a; x1 *= a; x2 *= a; x3 *= a; x4 *= a; x5 *= a; x6 *= a; x7 *= a;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int MODE> void run(const char* name, int w) {
  const int iters = 20000, threads = 256, blocks = 256 * w;
  float* d; (void)hipMalloc(&d, sizeof(float) * threads * blocks);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<blocks, threads>>>(d, 10, 1.0001f, 0.5f);
  (void)hipEventRecord(e0); k<MODE><<<blocks, threads>>>(d, iters, 1.0001f, 0.5f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-26s W=%d %.3f ms -> %.2f ns per wave-instr per SIMD\n", name, w, ms, ms * 1e6 / (8.0 * iters * w));
  (void)hipFree(d);
}
int main() {
  for (int w : {2, 4, 8}) {
    run<0>("fma 3 vgpr", w); run<1>("fma 2 vgpr + sgpr", w); run<2>("fma 1 vgpr + 2 sgpr", w); run<3>("mul 2 vgpr", w); run<4>("mul vgpr*sgpr", w);
  }
}
