// The reference's per-operator array functions in their own shapes (float64, channel-last, materialised tensors):
//   derivatives.jacobian / hessian / hessian_robust            (derivatives.py:7-107)
//   image_optimisation.steepest_descent_images / independent_vector(_robust) / rhop / robust_error_function
//                                                               (image_optimisation.py:16-143, 159-200)
// These are NOT the hot path -- the iteration kernel never materialises J, DIJ or rho -- they exist so that code written
// against the reference's operator functions (its unit tests, test/test_derivatives.py, and the profiling notebook) runs
// against this library unchanged.  Plain one-thread-per-element / grid-stride kernels in float64, like the NumPy code.
#pragma once
#include <hip/hip_runtime.h>

#include "ica_math.hpp"

namespace ica {

constexpr int kRefBlock = 256;
constexpr int kRefMaxN = 8;                               // parameters (last dimension of DIJ)
constexpr int kRefAcc = kRefMaxN * (kRefMaxN + 1) / 2 + kRefMaxN;  // upper triangle of H + b
constexpr int kRefBlocks = 512;                           // partial sums of the reduction

__device__ __forceinline__ double ref_finite_or_zero(double v) { return (fabs(v) <= 1.7976931348623157e308) ? v : 0.0; }

// J[y][x][2n]: the x row of the 2 x n Jacobian in [0, n), the y row in [n, 2n)  (derivatives.py:26-68)
__global__ void ica_ref_jacobian_kernel(int tt, int n, int nx, int ny, double* __restrict__ J) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)nx * ny) return;
  const double x = (double)(i % nx), y = (double)(i / nx);
  double* q = J + i * 2 * n;
  for (int k = 0; k < 2 * n; ++k) q[k] = 0.0;
  switch (tt) {
    case ICA_TRANSLATION: q[0] = 1.0; q[3] = 1.0; break;
    case ICA_EUCLIDEAN: q[0] = 1.0; q[2] = -y; q[4] = 1.0; q[5] = x; break;
    case ICA_SIMILARITY: q[0] = 1.0; q[2] = x; q[3] = -y; q[5] = 1.0; q[6] = y; q[7] = x; break;
    case ICA_AFFINITY: q[0] = 1.0; q[2] = x; q[3] = y; q[7] = 1.0; q[10] = x; q[11] = y; break;
    case ICA_HOMOGRAPHY:
      q[0] = x; q[1] = y; q[2] = 1.0; q[6] = -x * x; q[7] = -x * y;
      q[11] = x; q[12] = y; q[13] = 1.0; q[14] = -x * y; q[15] = -y * y;
      break;
    default: break;
  }
}

// DIJ[y][x][c][k] = Ix[y][x][c] * J[y][x][k] + Iy[y][x][c] * J[y][x][k + n]   (image_optimisation.py:193-198)
__global__ void ica_ref_sd_images_kernel(size_t pixels, int nz, int n, const double* __restrict__ Ix,
                                         const double* __restrict__ Iy, const double* __restrict__ J,
                                         double* __restrict__ DIJ) {
#pragma clang fp contract(off)  // two products and a sum, rounded like NumPy's (no fused multiply-add)
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // (pixel, channel, parameter)
  if (e >= pixels * nz * n) return;
  const int k = (int)(e % n);
  const size_t pc = e / n, px = pc / nz;
  DIJ[e] = Ix[pc] * J[px * 2 * n + k] + Iy[pc] * J[px * 2 * n + k + n];
}

// t2 -> rho'(t2)  (image_optimisation.py:16-53; TRUNCATED_QUADRATIC element-wise: the reference raises on arrays there)
__global__ void ica_ref_rhop_kernel(size_t count, const double* __restrict__ t2, double lambda, int type,
                                    double* __restrict__ out) {
#pragma clang fp contract(off)
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const double l2 = lambda * lambda, t = t2[i];
  double r;
  switch (type) {
    case ICA_QUADRATIC: r = 1.0; break;
    case ICA_TRUNCATED_QUADRATIC: r = t < l2 ? 1.0 : 0.0; break;
    case ICA_GERMAN_MCCLURE: r = l2 / ((l2 + t) * (l2 + t)); break;
    case ICA_LORENTZIAN: r = 1.0 / (l2 + t); break;
    default: r = 1.0 / sqrt(t + l2); break;  // CHARBONNIER
  }
  out[i] = r;
}

// rho[y][x] = rho'(sum_c finite(DI)^2), 0 where that sum is not finite  (image_optimisation.py:56-78)
__global__ void ica_ref_robust_error_kernel(size_t pixels, int nz, const double* __restrict__ DI, double lambda, int type,
                                            double* __restrict__ rho) {
#pragma clang fp contract(off)
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pixels) return;
  double t = 0.0;
  for (int c = 0; c < nz; ++c) {
    const double d = ref_finite_or_zero(DI[i * nz + c]);
    t += d * d;
  }
  const double l2 = lambda * lambda;
  double r;
  switch (type) {
    case ICA_QUADRATIC: r = 1.0; break;
    case ICA_TRUNCATED_QUADRATIC: r = t < l2 ? 1.0 : 0.0; break;
    case ICA_GERMAN_MCCLURE: r = l2 / ((l2 + t) * (l2 + t)); break;
    case ICA_LORENTZIAN: r = 1.0 / (l2 + t); break;
    default: r = 1.0 / sqrt(t + l2); break;
  }
  rho[i] = (fabs(t) <= 1.7976931348623157e308) ? r : 0.0;
}

// H = sum_px rho * sum_c dij_c (x) dij_c, b = sum_px rho * sum_c dij_c * di_c, non-finite entries of DIJ / DI counted as
// zero (derivatives.py:73-107, image_optimisation.py:82-143).  Stage 1: grid-stride over pixels, per-thread sums, block
// reduction in LDS (fixed order), one partial [kRefAcc] per block.  Stage 2 (one block) sums the partials in block order.
template <int N>
__global__ __launch_bounds__(kRefBlock) void ica_ref_normal_partial_kernel(size_t pixels, int nz, const double* __restrict__ DIJ,
                                                                            const double* __restrict__ DI,
                                                                            const double* __restrict__ rho,
                                                                            double* __restrict__ partial) {
  constexpr int NH = N * (N + 1) / 2;
  double acc[NH + N];
#pragma unroll
  for (int i = 0; i < NH + N; ++i) acc[i] = 0.0;
  for (size_t px = (size_t)blockIdx.x * kRefBlock + threadIdx.x; px < pixels; px += (size_t)gridDim.x * kRefBlock) {
    const double w = rho ? rho[px] : 1.0;
    for (int c = 0; c < nz; ++c) {
      double d[N];
#pragma unroll
      for (int k = 0; k < N; ++k) d[k] = ref_finite_or_zero(DIJ[(px * nz + c) * N + k]);
      const double di = DI ? ref_finite_or_zero(DI[px * nz + c]) : 0.0;
      int t = 0;
#pragma unroll
      for (int k = 0; k < N; ++k)
#pragma unroll
        for (int m = k; m < N; ++m) acc[t++] += w * (d[k] * d[m]);
#pragma unroll
      for (int k = 0; k < N; ++k) acc[NH + k] += w * (d[k] * di);
    }
  }
  __shared__ double red[kRefBlock];
  for (int i = 0; i < NH + N; ++i) {
    red[threadIdx.x] = acc[i];
    __syncthreads();
    for (int s = kRefBlock / 2; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) partial[(size_t)blockIdx.x * kRefAcc + i] = red[0];
    __syncthreads();
  }
}

template <int N>
__global__ void ica_ref_normal_final_kernel(int blocks, const double* __restrict__ partial, double* __restrict__ H,
                                            double* __restrict__ b) {
  constexpr int NH = N * (N + 1) / 2;
  const int i = threadIdx.x;
  if (i >= NH + N) return;
  double v = 0.0;
  for (int bl = 0; bl < blocks; ++bl) v += partial[(size_t)bl * kRefAcc + i];
  if (i < NH) {
    if (!H) return;
    int k = 0, rem = i;
    while (rem >= N - k) { rem -= N - k; ++k; }
    const int m = k + rem;
    H[k * N + m] = v;
    H[m * N + k] = v;
  } else if (b) {
    b[i - NH] = v;
  }
}

}  // namespace ica
