// Device-side definitions shared by every kernel of the inverse-compositional hot path (gfx950, wave64): launch
// arguments, the sharded active-pair lists, the warp's displacement and Keys weights.  (The kernels themselves are in
// ica_kernels.hpp -- one translation unit -- and ica_iterate.hpp -- one translation unit per transform type.)
//
// Data layout (HBM): images float32 planar [B][C][H][W]; per-pair state float64.
// The reference materialises gradient, Jacobian and steepest-descent tensors (192 B/pixel for
// C=3, n=8).  Here nothing but the two images is ever read per iteration: gradients, Jacobian rows,
// steepest-descent values and robust weights are recomputed in registers.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "ica_math.hpp"

namespace ica {

// ---- chunk heights of the fused iteration kernel (ica_iterate.hpp), needed by the host to pick tile heights -------
#ifndef ICA_LOADER   // n > 0: kernels of one-channel float32 levels run with n LOADER waves (ica_iterate.hpp, IterCfg); measured slower
#define ICA_LOADER 0
#endif
#ifndef ICA_CH_LOADER   // image rows per chunk of those kernels
#define ICA_CH_LOADER 24
#endif
#ifndef ICA_CH     // image rows per chunk, one channel, no loader wave
#define ICA_CH 32
#endif
#ifndef ICA_CSEQ   // 1 (an experiment, measured slower): three channels staged one at a time (ica_iterate.hpp, TileCfg::seq)
#define ICA_CSEQ 0
#endif
#ifndef ICA_CH3    // image rows per chunk, three channels (three RESIDENT channels only fit the LDS as 16-row chunks)
#define ICA_CH3 (ICA_CSEQ ? 32 : 16)
#endif
#ifndef ICA_DUP    // 1 (an experiment): one-channel float32 tiles hold a second copy of the I2 footprint, shifted by one float, so that
#define ICA_DUP 0  // every pair of adjacent taps is an aligned 8-byte LDS read (ica_iterate.hpp, TileCfg::dup); 24-row chunks, 5 waves
#endif
#ifndef ICA_CH_DUP
#define ICA_CH_DUP 24
#endif
// rows per chunk of the kernel that runs a level of `channels` channels stored as uint8 (`u8`) or float32
constexpr int iter_chunk_rows(int channels, bool u8) {
  return (ICA_LOADER != 0 && channels == 1 && !u8) ? ICA_CH_LOADER
         : (ICA_DUP != 0 && channels == 1 && !u8) ? ICA_CH_DUP : channels == 1 ? ICA_CH : ICA_CH3;
}

constexpr int kBlock = 256;   // 4 waves
constexpr int kTileW = 64;    // one lane per column of a tile
constexpr int kWaves = kBlock / 64;
constexpr int kFlush = 8;     // fp32 partial sums are promoted to fp64 every kFlush pixels per thread

enum Mode { kModeL2 = 0, kModeRobust = 1, kModeHessian = 2, kModeL2First = 3 };

#ifndef ICA_FLAT_BELOW
#define ICA_FLAT_BELOW 8
#endif

struct LevelArgs {
  const float* I1;          // float32, or uint8 when `u8` (reinterpreted by the kernels)
  const float* I2;
  int u8;                   // the images of this level are uint8 in HBM (level 0 of an ICA_STORAGE_U8 problem)
  int H, W;
  int tiles_x, tiles_y, tile_h;
  int delta, nanif;
  int robust_type;
  const double* p;          // [B][8]
  const double* lambda_it;  // [B]
  const float* minmax;      // [B][2] clip range of I2 at this level
  const int* act_list;      // current active pairs (nullptr: pair index = slot)
  const int* act_count;   // kListShards shard counters (see list_total / list_entry)
  int list_cap;           // entries per shard
  int n_pairs;              // pairs covered by this launch (the bound when act_list == nullptr; picks the block map)
  int* host_slot;           // nullable, host-visible: receives *act_count (pairs entering this iteration)
  double* partials;         // [slot][tiles][kAccStride]; slot = index into the active list (= pair without a list)
  unsigned long long* clock_acc;  // nullable (profiling): [0] += shader cycles, [1] += 100 MHz ticks of one wave's lifetime in the
                                  // middle of this launch, [2..3] its start stamps (the shader clock THIS kernel runs at)
};

// ---- active-pair lists ------------------------------------------------------------------------------
// A list of the pairs that enter an iteration is kept in kListShards shards; a producer (the solve of workgroup /
// slot b) appends to shard b & 7 with an atomicAdd on THAT shard's counter.  1024 workgroups adding to one counter
// serialise on it (measured: +7.5 us per solve launch); eight counters on eight cache lines do not.  A shard never
// receives more than ceil(B / 8) entries, so the eight shards together are no larger than one flat list.
// Consumers number the entries shard after shard.
constexpr int kFlatMapBelow = ICA_FLAT_BELOW;  // launches of fewer pairs use the flat block map (ica_accumulate2.hpp)
constexpr int kListShards = 8;
constexpr int kShardStride = 32;  // ints between two shard counters: one 128-byte line each
__host__ __device__ inline int list_shard_cap(int batch) { return (batch + kListShards - 1) / kListShards; }
__device__ __forceinline__ int list_total(const int* cnt) {
  int t = 0;
#pragma unroll
  for (int s = 0; s < kListShards; ++s) t += cnt[s * kShardStride];
  return t;
}
// entry number `slot` of the list (caller has checked slot < list_total)
__device__ __forceinline__ int list_entry(const int* list, const int* cnt, int cap, int slot) {
  int shard = 0;
#pragma unroll
  for (int s = 0; s < kListShards - 1; ++s) {
    const int c = cnt[s * kShardStride];
    if (shard == s && slot >= c) { slot -= c; shard = s + 1; }
  }
  return list[shard * cap + slot];
}
__device__ __forceinline__ void list_append(int* list, int* cnt, int cap, int producer, int value) {
  const int shard = producer & (kListShards - 1);
  const int idx = atomicAdd(cnt + shard * kShardStride, 1);
  list[shard * cap + idx] = value;
}
__device__ __forceinline__ void list_clear(int* cnt) {
#pragma unroll
  for (int s = 0; s < kListShards; ++s) cnt[s * kShardStride] = 0;
}


template <int TT> struct NP;
template <> struct NP<ICA_TRANSLATION> { static constexpr int n = 2; };
template <> struct NP<ICA_EUCLIDEAN> { static constexpr int n = 3; };
template <> struct NP<ICA_SIMILARITY> { static constexpr int n = 4; };
template <> struct NP<ICA_AFFINITY> { static constexpr int n = 6; };
template <> struct NP<ICA_HOMOGRAPHY> { static constexpr int n = 8; };

// Steepest-descent row  D[k] = Ix*Jx[k] + Iy*Jy[k]  (image_optimisation.py:190-192) with the Jacobian
// rows of derivatives.py:33-68; x = column, y = row, integer pixel indices.
template <int TT>
__device__ __forceinline__ void steepest_descent(float ix, float iy, float x, float y, float* D) {
  if constexpr (TT == ICA_TRANSLATION) {
    D[0] = ix; D[1] = iy;
  } else if constexpr (TT == ICA_EUCLIDEAN) {
    D[0] = ix; D[1] = iy; D[2] = x * iy - y * ix;
  } else if constexpr (TT == ICA_SIMILARITY) {
    D[0] = ix; D[1] = iy; D[2] = x * ix + y * iy; D[3] = x * iy - y * ix;
  } else if constexpr (TT == ICA_AFFINITY) {
    D[0] = ix; D[1] = iy; D[2] = x * ix; D[3] = y * ix; D[4] = x * iy; D[5] = y * iy;
  } else {
    const float s = x * ix + y * iy;
    D[0] = x * ix; D[1] = y * ix; D[2] = ix; D[3] = x * iy; D[4] = y * iy; D[5] = iy;
    D[6] = -x * s; D[7] = -y * s;
  }
}

// Displacement (dx, dy) = W(x;p) - x in fp32.  Working on the displacement instead of the absolute
// position keeps every term small (|d| is tens of pixels at most), so fp32 resolves the sampling
// position to ~1e-6 px even at x ~ 2000.
template <bool PROJ>
__device__ __forceinline__ void displacement(const float* a, float x, float y, float& dx, float& dy) {
  float nx = fmaf(a[0], x, fmaf(a[1], y, a[2]));
  float ny = fmaf(a[3], x, fmaf(a[4], y, a[5]));
  if constexpr (PROJ) {
    const float g = fmaf(a[6], x, a[7] * y);
    const float r = rcp_nr(1.0f + g);
    nx = (nx - x * g) * r;
    ny = (ny - y * g) * r;
  }
  dx = nx;
  dy = ny;
}

__device__ __forceinline__ void keys_weights(float t, float* w) {
  // Catmull-Rom (Keys a = -1/2), the cubic of skimage's _warp_fast / bicubic_interpolation.py:39-41, in factored form
  // (s = 1 - t):  w0 = -t s^2 / 2,  w1 = s + t s (1 - 3t/2),  w3 = -t^2 s / 2,  w2 = 1 - w0 - w1 - w3 = (1 + t s / 2) - w1.
  // 10 operations instead of the 14 of the expanded polynomials, 8 of them two-operand; the weights sum to 1 by
  // construction.
  const float s = 1.0f - t;
  const float q = t * s;
  const float h = -0.5f * q;
  w[0] = h * s;
  w[3] = h * t;
  w[1] = fmaf(q, fmaf(-0.5f, t, s), s);   // 1 - 3t/2 = s - t/2: only inline constants
  w[2] = (1.0f - h) - w[1];
}

// floor() of a sampling position that fp32 cannot place: pixel (x, y) whose fp32 position is within 2^-16 of an integer
// (see the header).  The position is X = xw / ww with (xw, yw, ww) = M (x, y, 1), M = params2matrix(p) as the reference
// builds it (transformation.py:188-236; identity when all |p_i| < 1e-10, bicubic_interpolation.py:173-175), evaluated in
// float64 in the order NumPy evaluates it -- (m0 x + m1 y) + m2, no fused multiply-adds -- so that an exactly
// representable position comes out exact.  With N the integer the fp32 position is next to, floor(X) = N or N - 1
// according to the sign of E = xw - N ww (ww > 0): no float64 division is needed.  Works on the floor `ic` of the
// DISPLACEMENT along one axis (c = the pixel's coordinate on that axis) and the fraction `fc`.
// One axis at a time, loading only the parameters that axis needs (this sits in a cold branch of a loop that is short of
// registers).  ROW = 0: x axis (matrix row 0), ROW = 1: y axis.
template <int TT, int ROW>
__device__ __forceinline__ void exact_axis(const double* p, bool ident, int x, int y, int c, int& ic, float& fc) {
#pragma clang fp contract(off)
  double ma = ROW == 0 ? 1.0 : 0.0, mb = ROW == 0 ? 0.0 : 1.0, mc = 0.0, ww = 1.0;  // row (ma, mb, mc) of M
  const double xd = (double)x, yd = (double)y;
  if (!ident) {
    if constexpr (TT == ICA_TRANSLATION) mc = p[ROW];
    else if constexpr (TT == ICA_EUCLIDEAN) {
      // cos - 1 as -2 sin^2(t/2), like params_to_displacement
      const double sn = sin(p[2]), hs = sin(0.5 * p[2]), cs = 1.0 + (-2.0 * hs * hs);
      ma = ROW == 0 ? cs : sn; mb = ROW == 0 ? -sn : cs; mc = p[ROW];
    } else if constexpr (TT == ICA_SIMILARITY) { ma = ROW == 0 ? 1 + p[2] : p[3]; mb = ROW == 0 ? -p[3] : 1 + p[2]; mc = p[ROW]; }
    else if constexpr (TT == ICA_AFFINITY) { ma = ROW == 0 ? 1 + p[2] : p[4]; mb = ROW == 0 ? p[3] : 1 + p[5]; mc = p[ROW]; }
    else {
      ma = ROW == 0 ? 1 + p[0] : p[3]; mb = ROW == 0 ? p[1] : 1 + p[4]; mc = ROW == 0 ? p[2] : p[5];
      ww = (p[6] * xd + p[7] * yd) + 1.0;
    }
  }
  const int nn = ic + (fc >= 0.5f ? 1 : 0);
  const double E = ((ma * xd + mb * yd) + mc) - (double)(c + nn) * ww;
  const float e = (float)E / (float)ww;  // X - N, tiny
  ic = E >= 0.0 ? nn : nn - 1;
  fc = E >= 0.0 ? e : 1.0f + e;
}

// The same with the transform type known at run time (warp write-out kernel).
template <int ROW>
__device__ __forceinline__ void exact_axis_rt(int tt, const double* p, bool ident, int x, int y, int c, int& ic, float& fc) {
  switch (tt) {
    case ICA_TRANSLATION: exact_axis<ICA_TRANSLATION, ROW>(p, ident, x, y, c, ic, fc); break;
    case ICA_EUCLIDEAN: exact_axis<ICA_EUCLIDEAN, ROW>(p, ident, x, y, c, ic, fc); break;
    case ICA_SIMILARITY: exact_axis<ICA_SIMILARITY, ROW>(p, ident, x, y, c, ic, fc); break;
    case ICA_AFFINITY: exact_axis<ICA_AFFINITY, ROW>(p, ident, x, y, c, ic, fc); break;
    default: exact_axis<ICA_HOMOGRAPHY, ROW>(p, ident, x, y, c, ic, fc); break;
  }
}

constexpr float kNearInteger = 1.0f / 65536.0f;  // fp32 positions closer than this to an integer are re-evaluated

// Bicubic sample of C planes at (x+dx, y+dy); false (and nothing written) when any of the 16 taps is
// outside the image -- the reference's NaN (bicubic_interpolation.py:199, cval=nan).  The result is
// clipped to [lo, hi] as skimage.transform.warp(clip=True) does.  `p` / `tt`: the pair's parameters, for the float64
// re-evaluation of a position that fp32 puts within kNearInteger of an integer (exact_axis).
template <int C, class TIn>
__device__ __forceinline__ bool warp_sample(const TIn* __restrict__ img, int H, int W, int x, int y,
                                            float dx, float dy, float lo, float hi, float* out, const double* p, int tt) {
  const float flx = floorf(dx), fly = floorf(dy);
  float fx = dx - flx, fy = dy - fly;
  int ix = (int)flx, iy = (int)fly;
  const bool near_x = fminf(fx, 1.0f - fx) < kNearInteger, near_y = fminf(fy, 1.0f - fy) < kNearInteger;
  if (near_x | near_y) {
    bool ident = true;
    const int n = nparams_of(tt);
    for (int i = 0; i < n; ++i) ident = ident & (fabs(p[i]) < 1e-10);
    if (near_x) exact_axis_rt<0>(tt, p, ident, x, y, x, ix, fx);
    if (near_y) exact_axis_rt<1>(tt, p, ident, x, y, y, iy, fy);
  }
  const int x0 = x + ix, y0 = y + iy;
  const bool ok = (x0 >= 1) & (x0 <= W - 3) & (y0 >= 1) & (y0 <= H - 3);
  if (!ok) return false;
  float wx[4], wy[4];
  keys_weights(fx, wx);
  keys_weights(fy, wy);
  const size_t plane = (size_t)H * W;
  const TIn* base = img + (size_t)(y0 - 1) * W + (x0 - 1);
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const TIn* q = base + c * plane;
    float v = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const TIn* row = q + (size_t)r * W;
      const float s = wx[0] * (float)row[0] + wx[1] * (float)row[1] + wx[2] * (float)row[2] + wx[3] * (float)row[3];
      v = fmaf(wy[r], s, v);
    }
    out[c] = fminf(fmaxf(v, lo), hi);
  }
  return true;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;  // valid in lane 0
}

}  // namespace ica
