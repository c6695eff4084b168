// Does packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) issue at the same wave-instruction rate as scalar
// fp32 VALU on gfx950?  8 independent chains per lane, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, int iters, float a, float b) {
  f2 x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
  f2 va = {a + threadIdx.x * 1e-9f, a}, vb = {b, b + threadIdx.x * 1e-9f};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // packed fma, VGPR operands, 8 independent chains
      x0 = __builtin_elementwise_fma(x0, va, vb); x1 = __builtin_elementwise_fma(x1, va, vb);
      x2 = __builtin_elementwise_fma(x2, va, vb); x3 = __builtin_elementwise_fma(x3, va, vb);
      x4 = __builtin_elementwise_fma(x4, va, vb); x5 = __builtin_elementwise_fma(x5, va, vb);
      x6 = __builtin_elementwise_fma(x6, va, vb); x7 = __builtin_elementwise_fma(x7, va, vb);
    } else if (MODE == 1) {  // packed mul / add
      x0 = x0 * va; x1 = x1 + vb; x2 = x2 * va; x3 = x3 + vb; x4 = x4 * va; x5 = x5 + vb; x6 = x6 * va; x7 = x7 + vb;
    } else if (MODE == 2) {  // scalar fma on .x only (baseline)
      x0.x = fmaf(x0.x, va.x, vb.x); x1.x = fmaf(x1.x, va.x, vb.x); x2.x = fmaf(x2.x, va.x, vb.x); x3.x = fmaf(x3.x, va.x, vb.x);
      x4.x = fmaf(x4.x, va.x, vb.x); x5.x = fmaf(x5.x, va.x, vb.x); x6.x = fmaf(x6.x, va.x, vb.x); x7.x = fmaf(x7.x, va.x, vb.x);
    } else {  // scalar mul / add
      x0.x = x0.x * va.x; x1.x = x1.x + vb.x; x2.x = x2.x * va.x; x3.x = x3.x + vb.x; x4.x = x4.x * va.x; x5.x = x5.x + vb.x; x6.x = x6.x * va.x; x7.x = x7.x + vb.x;
    }
  }
  f2 s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
template <int MODE> void run(const char* name, int w) {
  const int iters = 20000, threads = 256, blocks = 256 * w;
  float* d; hipMalloc(&d, sizeof(float) * threads * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, threads>>>(d, 10, 1.0001f, 0.5f);
  hipEventRecord(e0); k<MODE><<<blocks, threads>>>(d, iters, 1.0001f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-22s W=%d  %.3f ms -> %.2f ns per wave-instruction per SIMD\n", name, w, ms, ms * 1e6 / (8.0 * iters * w));
  hipFree(d);
}
int main() {
  for (int w : {2, 4, 8}) { run<2>("scalar fma", w); run<0>("pk_fma", w); run<3>("scalar mul/add", w); run<1>("pk_mul/pk_add", w); }
  return 0;
}
