// Small fp64 algebra shared by the kernels: transform parametrisations, the compositional update,
// the <=8x8 LU inverse and the robust weights.  Reference citations are relative to /root/reference.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/ica_hip.h"

namespace ica {

constexpr int kMaxN = ICA_MAX_PARAMS;                    // 8
constexpr int kTri = kMaxN * (kMaxN + 1) / 2;            // 36 upper-triangular Hessian terms
constexpr int kAccStride = kMaxN + kTri;                 // 44 doubles per partial: [b | triu(H)]

__host__ __device__ inline int nparams_of(int tt) {
  // transformation.py:16-32
  switch (tt) {
    case ICA_TRANSLATION: return 2;
    case ICA_EUCLIDEAN: return 3;
    case ICA_SIMILARITY: return 4;
    case ICA_AFFINITY: return 6;
    case ICA_HOMOGRAPHY: return 8;
    default: return -1;
  }
}

__host__ __device__ inline int tri_index(int k, int m, int n) {  // k <= m < n, row-major upper triangle
  return k * n - k * (k - 1) / 2 + (m - k);
}

// M - I for the warp, from p.  transformation.py:188-236 (params2matrix) and the identity shortcut of
// bicubic_interpolation.py:173-175 (all |p_i| < 1e-10 -> identity).  a = {m00-1, m01, m02, m10, m11-1,
// m12, m20, m21}; m22 == 1 for every parametrisation.  cos(t)-1 is formed as -2 sin^2(t/2) so that the
// displacement form used by the kernels keeps full precision for small angles.
__host__ __device__ inline void params_to_displacement(const double* p, int tt, double a[8]) {
  const int n = nparams_of(tt);
  bool ident = true;
  for (int i = 0; i < n; ++i) ident = ident && (fabs(p[i]) < 1e-10);
  for (int i = 0; i < 8; ++i) a[i] = 0.0;
  if (ident) return;
  switch (tt) {
    case ICA_TRANSLATION: a[2] = p[0]; a[5] = p[1]; break;
    case ICA_EUCLIDEAN: {
      const double s = sin(p[2]), h = sin(0.5 * p[2]);
      const double cm1 = -2.0 * h * h;
      a[0] = cm1; a[1] = -s; a[2] = p[0]; a[3] = s; a[4] = cm1; a[5] = p[1];
    } break;
    case ICA_SIMILARITY:
      a[0] = p[2]; a[1] = -p[3]; a[2] = p[0]; a[3] = p[3]; a[4] = p[2]; a[5] = p[1]; break;
    case ICA_AFFINITY:
      a[0] = p[2]; a[1] = p[3]; a[2] = p[0]; a[3] = p[4]; a[4] = p[5]; a[5] = p[1]; break;
    case ICA_HOMOGRAPHY:
      for (int i = 0; i < 8; ++i) a[i] = p[i];
      break;
    default: break;
  }
}

// Full 3x3 matrix (row-major) of transformation.py:188-236, no identity shortcut.
__host__ __device__ inline void params_to_matrix(const double* p, int tt, double m[9]) {
  m[0] = 1; m[1] = 0; m[2] = 0; m[3] = 0; m[4] = 1; m[5] = 0; m[6] = 0; m[7] = 0; m[8] = 1;
  switch (tt) {
    case ICA_TRANSLATION: m[2] = p[0]; m[5] = p[1]; break;
    case ICA_EUCLIDEAN: {
      const double c = cos(p[2]), s = sin(p[2]);
      m[0] = c; m[1] = -s; m[2] = p[0]; m[3] = s; m[4] = c; m[5] = p[1];
    } break;
    case ICA_SIMILARITY:
      m[0] = 1 + p[2]; m[1] = -p[3]; m[2] = p[0]; m[3] = p[3]; m[4] = 1 + p[2]; m[5] = p[1]; break;
    case ICA_AFFINITY:
      m[0] = 1 + p[2]; m[1] = p[3]; m[2] = p[0]; m[3] = p[4]; m[4] = 1 + p[5]; m[5] = p[1]; break;
    case ICA_HOMOGRAPHY:
      m[0] = 1 + p[0]; m[1] = p[1]; m[2] = p[2]; m[3] = p[3]; m[4] = 1 + p[4]; m[5] = p[5];
      m[6] = p[6]; m[7] = p[7];
      break;
    default: break;
  }
}

__host__ __device__ inline void adjugate3(const double* a, double* c) {
  c[0] = a[4] * a[8] - a[5] * a[7];
  c[1] = a[2] * a[7] - a[1] * a[8];
  c[2] = a[1] * a[5] - a[2] * a[4];
  c[3] = a[5] * a[6] - a[3] * a[8];
  c[4] = a[0] * a[8] - a[2] * a[6];
  c[5] = a[2] * a[3] - a[0] * a[5];
  c[6] = a[3] * a[7] - a[4] * a[6];
  c[7] = a[1] * a[6] - a[0] * a[7];
  c[8] = a[0] * a[4] - a[1] * a[3];
}

__host__ __device__ inline void matmul3(const double* a, const double* b, double* c) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) c[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}

// p <- params(M(p) . M(dp)^-1), in place (transformation.py:35-141).  The reference spells the
// composition out in scalars; composing 3x3 matrices (M(p).adj(M(dp)) / normaliser) is the same
// function to rounding.  Unless `exact`, the two terms in which the reference's scalar formulas differ
// from the exact composition are re-applied: AFFINITY p[1] (+ (d*d*ep - c*d*ep)/det,
// transformation.py:106) and HOMOGRAPHY p[4] (- (-a + c*g)/det, :136).  The update is skipped when
// det*det <= 1e-10 (:79, :98, :131).
__host__ __device__ inline void update_transform(double* p, const double* dp, int tt, bool exact) {
  if (tt == ICA_TRANSLATION) {
    p[0] -= dp[0];
    p[1] -= dp[1];
    return;
  }
  double P[9], A[9], adj[9], m[9];
  params_to_matrix(p, tt, P);
  params_to_matrix(dp, tt, A);
  adjugate3(A, adj);
  matmul3(P, adj, m);
  switch (tt) {
    case ICA_EUCLIDEAN:
      p[0] = m[2]; p[1] = m[5]; p[2] = atan2(m[3], m[0]);
      break;
    case ICA_SIMILARITY: {
      const double det = A[0] * A[0] + A[3] * A[3];
      if (det * det > 1e-10) {
        p[0] = m[2] / det; p[1] = m[5] / det; p[2] = m[0] / det - 1.0; p[3] = m[3] / det;
      }
    } break;
    case ICA_AFFINITY: {
      const double det = A[0] * A[4] - A[1] * A[3];
      if (det * det > 1e-10) {
        const double ep = p[5], c = dp[0], d = dp[4];
        p[0] = m[2] / det; p[1] = m[5] / det; p[2] = m[0] / det - 1.0; p[3] = m[1] / det;
        p[4] = m[3] / det; p[5] = m[4] / det - 1.0;
        if (!exact) p[1] += (d * d * ep - c * d * ep) / det;
      }
    } break;
    case ICA_HOMOGRAPHY: {
      const double det = -m[8];  // == the reference's `det` (transformation.py:130)
      if (det * det > 1e-10) {
        const double a = dp[0], c = dp[2], g = dp[6];
        const double s = 1.0 / m[8];
        p[0] = m[0] * s - 1.0; p[1] = m[1] * s; p[2] = m[2] * s; p[3] = m[3] * s;
        p[4] = m[4] * s - 1.0; p[5] = m[5] * s; p[6] = m[6] * s; p[7] = m[7] * s;
        if (!exact) p[4] -= (-a + c * g) / det;
      }
    } break;
    default: break;
  }
}

// zoom.py:62-125
__host__ __device__ inline void zoom_in_parameters(double* p, int tt, int w, int h, int ww, int hh) {
  const double fx = (double)ww / (double)w, fy = (double)hh / (double)h;
  const double nu = fx > fy ? fx : fy;
  if (tt == ICA_HOMOGRAPHY) {
    p[2] *= nu; p[5] *= nu; p[6] /= nu; p[7] /= nu;
  } else {
    p[0] *= nu; p[1] *= nu;
  }
}

// inv(H) the way np.linalg.inv does it (derivatives.py:110-130): LU with partial pivoting (dgetrf),
// then solve against the identity (dgesv).  Returns false -- and Hinv = 0 -- when a pivot is exactly
// zero, which is when LAPACK reports the matrix singular and the reference falls back to zeros.
// H and Hinv are n x n, row stride `ld`.
// `ws` is a caller-provided workspace of kLuWork doubles (LDS in the kernels: private arrays with run-time
// indices would live in scratch memory).
constexpr int kLuWork = kMaxN * kMaxN + 2 * kMaxN;

// LU with partial pivoting of the n x n matrix H (row stride ld) into a[n][kMaxN] (unit-lower L below the
// diagonal, U on and above it) and the pivot rows pivd[kMaxN] (stored as doubles: exact small integers).
// False when a pivot is exactly zero (LAPACK's "singular").
__host__ __device__ inline bool lu_factor(int n, const double* H, int ld, double* a, double* pivd) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) a[i * kMaxN + j] = H[i * ld + j];
  bool ok = true;
  for (int k = 0; k < n; ++k) {
    int pr = k;
    double best = fabs(a[k * kMaxN + k]);
    for (int i = k + 1; i < n; ++i) {
      const double v = fabs(a[i * kMaxN + k]);
      if (v > best) { best = v; pr = i; }
    }
    pivd[k] = (double)pr;
    if (pr != k)
      for (int j = 0; j < n; ++j) { const double t = a[k * kMaxN + j]; a[k * kMaxN + j] = a[pr * kMaxN + j]; a[pr * kMaxN + j] = t; }
    if (a[k * kMaxN + k] == 0.0) { ok = false; continue; }
    const double r = 1.0 / a[k * kMaxN + k];
    for (int i = k + 1; i < n; ++i) a[i * kMaxN + k] *= r;
    for (int i = k + 1; i < n; ++i) {
      const double l = a[i * kMaxN + k];
      for (int j = k + 1; j < n; ++j) a[i * kMaxN + j] -= l * a[k * kMaxN + j];
    }
  }
  return ok;
}

// Column c of the inverse from the factors: solve L U x = P e_c; x is a caller-provided scratch of kMaxN doubles.
// The columns are independent of each other (the kernels give one to each of n threads).
__host__ __device__ inline void lu_inverse_column(int n, const double* a, const double* pivd, int c, double* x,
                                                  double* Hinv, int ld) {
  for (int i = 0; i < n; ++i) x[i] = (i == c) ? 1.0 : 0.0;
  for (int k = 0; k < n; ++k) {
    const int pk = (int)pivd[k];
    if (pk != k) { const double t = x[k]; x[k] = x[pk]; x[pk] = t; }
  }
  for (int i = 1; i < n; ++i) {  // L y = P e_c (unit lower)
    double sum = x[i];
    for (int j = 0; j < i; ++j) sum -= a[i * kMaxN + j] * x[j];
    x[i] = sum;
  }
  for (int i = n - 1; i >= 0; --i) {  // U x = y
    double sum = x[i];
    for (int j = i + 1; j < n; ++j) sum -= a[i * kMaxN + j] * x[j];
    x[i] = sum / a[i * kMaxN + i];
  }
  for (int i = 0; i < n; ++i) Hinv[i * ld + c] = x[i];
}

__host__ __device__ inline bool lu_inverse(int n, const double* H, double* Hinv, int ld, double* ws) {
  double* a = ws;                        // [n][kMaxN]
  double* x = ws + kMaxN * kMaxN;        // [kMaxN]
  double* pivd = x + kMaxN;              // [kMaxN]
  if (!lu_factor(n, H, ld, a, pivd)) {
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) Hinv[i * ld + j] = 0.0;
    return false;
  }
  for (int c = 0; c < n; ++c) lu_inverse_column(n, a, pivd, c, x, Hinv, ld);
  return true;
}

// 1/d and 1/sqrt(d) in fp32: hardware estimate (1 ulp) + one Newton step, ~4 instructions instead of the ~10 of the
// IEEE division sequence -- these sit in the per-pixel loop of the projective and the robust kernels.
__device__ __forceinline__ float rcp_nr(float d) {
  const float r = __builtin_amdgcn_rcpf(d);
  return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
}
__device__ __forceinline__ float rsqrt_nr(float d) {
  const float r = __builtin_amdgcn_rsqf(d);
  const float h = 0.5f * d * r;
  return __builtin_fmaf(__builtin_fmaf(-h, r, 0.5f), r, r);
}

// rho'(t^2; lambda) (image_optimisation.py:17-53).  TRUNCATED_QUADRATIC raises in the reference when
// given an array (:40); the element-wise reading is implemented (documented extension).
__device__ inline float rho_prime(float t2, float lam2, int type) {
  switch (type) {
    case ICA_TRUNCATED_QUADRATIC: return t2 < lam2 ? 1.0f : 0.0f;
    case ICA_GERMAN_MCCLURE: {
      const float s = lam2 + t2;
      return lam2 * rcp_nr(s * s);
    }
    case ICA_LORENTZIAN: return rcp_nr(lam2 + t2);
    case ICA_CHARBONNIER: return rsqrt_nr(t2 + lam2);
    default: return 1.0f;
  }
}

}  // namespace ica
