// Shader clock under a VALU-heavy load: s_memtime ticks (shader cycles) vs the constant 100 MHz wall clock.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out, long long* t, int iters) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  const float a = 1.0001f + threadIdx.x * 1e-9f, b = 0.5f;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
  if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}
int main() {
  const int blocks = 256 * 8, threads = 256;
  float* d; long long* t; hipMalloc(&d, sizeof(float) * threads * blocks); hipMalloc(&t, 16);
  for (int rep = 0; rep < 3; ++rep) {
    k<<<blocks, threads>>>(d, t, 200000);
    long long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("shader ticks %lld, wall ticks (100 MHz) %lld -> %.3f GHz; %.2f cycles per fma wave-instruction per SIMD (8 waves/SIMD)\n",
           h[0], h[1], h[0] / (h[1] * 10.0), (double)h[0] / (4.0 * 200000 * 8));
  }
  return 0;
}
