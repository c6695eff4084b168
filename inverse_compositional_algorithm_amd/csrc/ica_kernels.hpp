// HIP kernels of the inverse-compositional hot path other than the fused iteration kernel (ica_iterate.hpp): the
// direct-gather cross-check kernel, per-pair solve / update, level bookkeeping, warp write-out, transform_image,
// min/max, the pyramid passes and the per-operator test hooks.  Included by ONE translation unit (ica_api.hip).
#pragma once
#include "ica_common.hpp"

namespace ica {

// The fused per-iteration pass (ref :111-117 / :227-244 and the setup Hessian :96-102):
//   MODE kModeL2      b = sum DIJ^T DI                      over F & V(p)
//   MODE kModeRobust  b and H = sum rho' DIJ^T DIJ           (rho'(0) where the warp is outside)
//   MODE kModeHessian H = sum DIJ^T DIJ over F               (L2 setup, no warp)
//   MODE kModeL2First b and H with weights 1                 (first L2 iteration of a level; accumulate2 only)
// F = pixels outside the NaN frame, V(p) = pixels whose 4x4 footprint is inside I2.
// One workgroup = one 64 x tile_h tile of one pair; lane = column, the 4 waves interleave rows.
template <int TT, int MODE, int C>
__global__ __launch_bounds__(kBlock) void ica_accumulate_kernel(LevelArgs a) {
  constexpr int N = NP<TT>::n;
  constexpr int NT = N * (N + 1) / 2;
  constexpr bool kB = (MODE != kModeHessian);
  constexpr bool kH = (MODE != kModeL2);
  constexpr bool kProj = (TT == ICA_HOMOGRAPHY);

  int pair = blockIdx.y;
  if (a.act_list) {
    if (a.host_slot && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
      __hip_atomic_store(a.host_slot, list_total(a.act_count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((int)blockIdx.y >= list_total(a.act_count)) return;
    pair = list_entry(a.act_list, a.act_count, a.list_cap, blockIdx.y);
  }
  const int tile = blockIdx.x;
  const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int H = a.H, W = a.W;
  const size_t plane = (size_t)H * W;
  const float* I1 = a.I1 + (size_t)pair * C * plane;
  const float* I2 = a.I2 + (size_t)pair * C * plane;

  float A[8];
  float lo = 0.f, hi = 0.f, lam2 = 0.f;
  if constexpr (kB) {
    double ad[8];
    params_to_displacement(a.p + (size_t)pair * 8, TT, ad);
#pragma unroll
    for (int i = 0; i < 8; ++i) A[i] = (float)ad[i];
    lo = a.minmax[pair * 2];
    hi = a.minmax[pair * 2 + 1];
    if constexpr (MODE == kModeRobust) {
      const float l = (float)a.lambda_it[pair];
      lam2 = l * l;
    }
  }

  float bacc[N], hacc[NT];
  double bsum[N], hsum[NT];
#pragma unroll
  for (int k = 0; k < N; ++k) { bacc[k] = 0.f; bsum[k] = 0.0; }
#pragma unroll
  for (int k = 0; k < NT; ++k) { hacc[k] = 0.f; hsum[k] = 0.0; }

  const int x = tx * kTileW + lane;
  const int y_begin = ty * a.tile_h;
  const int y_end = min(H, y_begin + a.tile_h);
  const bool frame_on = a.nanif && a.delta > 0;
  const bool col_ok = (x < W) && !(frame_on && (x < a.delta || x >= W - a.delta));
  const float xf = (float)x;
  int since_flush = 0;

  for (int y = y_begin + wave; y < y_end; y += kWaves) {
    const bool row_ok = !(frame_on && (y < a.delta || y >= H - a.delta));  // wave-uniform
    if (row_ok) {
      const bool act = col_ok;
      const float yf = (float)y;
      float di[C], D[C][N];
      bool valid = false;
      if (act) {
        float dxw = 0.f, dyw = 0.f;
        float iw[C];
        if constexpr (kB) {
          displacement<kProj>(A, xf, yf, dxw, dyw);
          valid = warp_sample<C, float>(I2, H, W, x, y, dxw, dyw, lo, hi, iw, a.p + (size_t)pair * 8, TT);
        }
        const float* c1 = I1 + (size_t)y * W + x;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float* q = c1 + c * plane;
          const float ix = (x >= 1 && x <= W - 2) ? 0.5f * (q[1] - q[-1]) : 0.0f;       // ref :81
          const float iy = (y >= 1 && y <= H - 2) ? 0.5f * (q[W] - q[-(ptrdiff_t)W]) : 0.0f;  // ref :82
          steepest_descent<TT>(ix, iy, xf, yf, D[c]);
          if constexpr (kB) {
            const float d = valid ? iw[c] - q[0] : 0.0f;                               // ref :114
            di[c] = (fabsf(d) <= 3.0e38f) ? d : 0.0f;                                  // non-finite -> 0
          }
        }
        float w = 1.0f;
        if constexpr (MODE == kModeRobust) {
          float t2 = 0.f;
#pragma unroll
          for (int c = 0; c < C; ++c) t2 = fmaf(di[c], di[c], t2);                     // io.py:74-76
          w = rho_prime(t2, lam2, a.robust_type);
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
          if constexpr (kB) {
            const float wd = w * di[c];
#pragma unroll
            for (int k = 0; k < N; ++k) bacc[k] = fmaf(wd, D[c][k], bacc[k]);
          }
          if constexpr (kH) {
            int idx = 0;
#pragma unroll
            for (int k = 0; k < N; ++k) {
              const float wk = w * D[c][k];
#pragma unroll
              for (int m = k; m < N; ++m) { hacc[idx] = fmaf(wk, D[c][m], hacc[idx]); ++idx; }
            }
          }
        }
      }
      if (++since_flush == kFlush) {
        since_flush = 0;
        if constexpr (kB) {
#pragma unroll
          for (int k = 0; k < N; ++k) { bsum[k] += (double)bacc[k]; bacc[k] = 0.f; }
        }
        if constexpr (kH) {
#pragma unroll
          for (int k = 0; k < NT; ++k) { hsum[k] += (double)hacc[k]; hacc[k] = 0.f; }
        }
      }
    }
  }
  if constexpr (kB) {
#pragma unroll
    for (int k = 0; k < N; ++k) bsum[k] += (double)bacc[k];
  }
  if constexpr (kH) {
#pragma unroll
    for (int k = 0; k < NT; ++k) hsum[k] += (double)hacc[k];
  }

  // fixed-order reduction: lanes (shuffle tree) -> 4 waves (LDS, summed in wave order) -> one partial
  __shared__ double red[kWaves][kAccStride];
  if constexpr (kB) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const double s = wave_sum(bsum[k]);
      if (lane == 0) red[wave][k] = s;
    }
  }
  if constexpr (kH) {
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      const double s = wave_sum(hsum[k]);
      if (lane == 0) red[wave][kMaxN + k] = s;
    }
  }
  __syncthreads();
  double* out = a.partials + ((size_t)blockIdx.y * (a.tiles_x * a.tiles_y) + tile) * kAccStride;
  const int t = threadIdx.x;
  if (kB && t < N) out[t] = ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t];
  if (kH && t >= kMaxN && t < kMaxN + NT) out[t] = ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t];
}

// ---- per-pair small kernels ---------------------------------------------------------------------

struct SolveArgs {
  int tt, n, robust, flags;
  int tiles;
  int max_iter;
  int level, nscales;
  int iter;               // iteration index of this launch (for the count history)
  double tol, lambda_user, lambdaN, lambda_ratio;
  const double* partials;
  double* p;              // [B][8]
  double* p_prev;         // [B][8]
  double* Hinv;           // [B][64] (L2: read, robust: scratch not used)
  double* lambda_it;      // [B]
  double* error;          // [B]
  int* niter;             // [B]
  const int* cur_list; const int* cur_count;
  int* nxt_list; int* nxt_count;
  int* clr_count;
  int list_cap;  // entries per list shard
  int* count_hist;        // [nscales][max_iter]
  double* trace;          // nullable [B][nscales][max_iter][10]
  int invert_h;           // this launch carries H partials: invert them (robust loop; L2: first iteration of a level)
  int store_hinv;         // L2, first iteration of a level: keep inv(H) in Hinv[pair] for the next iterations
};

// Sum the tile partials of one launch slot in a fixed order (deterministic): 256 threads = 4 groups x 64 lanes;
// lane k < 44 of group g adds tiles g, g+4, g+8, ... of accumulator k (4 loads in flight), then the 4 group sums are
// added in group order.  Only accumulators [k_lo, k_hi) are touched.
constexpr int kSolveThreads = 256;
// doubles of LDS a solve needs: s[44] | gather scratch | H | H^-1 | LU factors [64] + pivots [8] + one x[8] per column
constexpr int kSolveScratch = 5 * kAccStride + 2 * kMaxN * kMaxN + kMaxN * kMaxN + kMaxN + kMaxN * kMaxN;
__device__ __forceinline__ void gather_partials(const double* partials, int slot, int tiles, double* s /*LDS[44]*/,
                                                double* part /*LDS[4][44], or [32][8] for the b-only case*/,
                                                int k_lo = 0, int k_hi = kAccStride) {
  if (k_lo == 0 && k_hi <= kMaxN) {
    // b only (every L2 iteration after the first): 8 sums over the tiles.  All 256 threads load -- 32 tile groups x
    // 8 values, a thread's loads all in flight together -- because with hundreds of tiles (short tiles of the late,
    // small launches) this gather IS the latency of the solve.  Fixed summation order.
    const int k = threadIdx.x & 7, g = threadIdx.x >> 3;
    const double* q = partials + (size_t)slot * tiles * kAccStride + k;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int t = g;
    for (; t + 96 < tiles; t += 128) {
      a0 += q[(size_t)t * kAccStride];
      a1 += q[(size_t)(t + 32) * kAccStride];
      a2 += q[(size_t)(t + 64) * kAccStride];
      a3 += q[(size_t)(t + 96) * kAccStride];
    }
    for (; t < tiles; t += 32) a0 += q[(size_t)t * kAccStride];
    part[g * 8 + k] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (threadIdx.x < kAccStride) {
      double v = 0.0;
      if (threadIdx.x < k_hi) {
#pragma unroll
        for (int gg = 0; gg < 32; ++gg) v += part[gg * 8 + threadIdx.x];
      }
      s[threadIdx.x] = v;
    }
    __syncthreads();
    return;
  }
  const int k = threadIdx.x & 63, g = threadIdx.x >> 6;
  if (k >= k_lo && k < k_hi) {
    const double* q = partials + (size_t)slot * tiles * kAccStride + k;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int t = g;
    for (; t + 12 < tiles; t += 16) {
      a0 += q[(size_t)t * kAccStride];
      a1 += q[(size_t)(t + 4) * kAccStride];
      a2 += q[(size_t)(t + 8) * kAccStride];
      a3 += q[(size_t)(t + 12) * kAccStride];
    }
    for (; t < tiles; t += 4) a0 += q[(size_t)t * kAccStride];
    part[g * kAccStride + k] = (a0 + a1) + (a2 + a3);
  }
  __syncthreads();
  if (threadIdx.x < kAccStride) {
    const int kk = threadIdx.x;
    s[kk] = (kk >= k_lo && kk < k_hi)
                ? ((part[kk] + part[kAccStride + kk]) + part[2 * kAccStride + kk]) + part[3 * kAccStride + kk]
                : 0.0;
  }
  __syncthreads();
}

// L2 setup: H^-1 once per level (ref :102-103).
__global__ __launch_bounds__(kSolveThreads) void ica_setup_solve_kernel(const double* partials, int tiles, int n, double* Hinv) {
  __shared__ double scratch[kSolveScratch];
  double* s = scratch;
  double* Hm = scratch + 5 * kAccStride;
  double* ws = Hm + 2 * kMaxN * kMaxN;
  const int pair = blockIdx.x;
  gather_partials(partials, pair, tiles, s, scratch + kAccStride, kMaxN, kAccStride);
  if (threadIdx.x == 0) {
    for (int k = 0; k < n; ++k)
      for (int m = k; m < n; ++m) Hm[k * kMaxN + m] = Hm[m * kMaxN + k] = s[kMaxN + tri_index(k, m, n)];
    lu_inverse(n, Hm, Hinv + (size_t)pair * 64, kMaxN, ws);
  }
}

// lu_factor() of ica_math.hpp with the row operations of each elimination step spread over the lanes of ONE wave
// (lane i owns row i; same arithmetic per element, so the factors are bit-identical to the serial ones).  Called by
// a whole wave; `a` and `pivd` are LDS.  LDS operations of a wave complete in program order, and every lane runs
// the same instruction stream, so a lane reads what other lanes stored in an earlier statement.
__device__ __forceinline__ bool lu_factor_wave(int n, const double* H, int ld, double* a, double* pivd, int lane) {
  if (lane < n)
    for (int j = 0; j < n; ++j) a[lane * kMaxN + j] = H[lane * ld + j];
  bool ok = true;
  for (int k = 0; k < n; ++k) {
    int pr = k;  // every lane finds the pivot itself (same answer, no broadcast needed)
    double best = fabs(a[k * kMaxN + k]);
    for (int i = k + 1; i < n; ++i) {
      const double v = fabs(a[i * kMaxN + k]);
      if (v > best) { best = v; pr = i; }
    }
    if (lane == 0) pivd[k] = (double)pr;
    if (pr != k && lane < n) {  // lane j swaps column j of the two rows
      const double t = a[k * kMaxN + lane];
      a[k * kMaxN + lane] = a[pr * kMaxN + lane];
      a[pr * kMaxN + lane] = t;
    }
    const double piv = a[k * kMaxN + k];
    if (piv == 0.0) { ok = false; continue; }
    const double r = 1.0 / piv;
    if (lane > k && lane < n) {
      const double l = a[lane * kMaxN + k] * r;
      a[lane * kMaxN + k] = l;
      for (int j = k + 1; j < n; ++j) a[lane * kMaxN + j] -= l * a[k * kMaxN + j];
    }
  }
  return ok;
}

// Solve + compositional update + loop control for one active pair (ref :117-131 / :233-259).  Called by all 256
// threads of a workgroup (ica_solve_kernel); `scratch` is kSolveScratch doubles of LDS.  `slot` indexes the launch's partials.
__device__ __forceinline__ void solve_pair(const SolveArgs& a, int slot, int pair, double* scratch) {
  double* s = scratch;
  double* Hm = scratch + 5 * kAccStride;
  double* Hi = Hm + kMaxN * kMaxN;
  double* ws = Hi + kMaxN * kMaxN;
  // the stored inverse (L2, iterations >= 1) is fetched while the partials are being gathered
  double hpre = 0.0;
  if (!a.invert_h && threadIdx.x < kMaxN * kMaxN) hpre = a.Hinv[(size_t)pair * 64 + threadIdx.x];
  // ... and so is the per-pair state thread 0 needs at the end (one memory round trip instead of three in its tail)
  double pcur[8] = {0, 0, 0, 0, 0, 0, 0, 0}, lam = 0.0;
  int it = 0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) pcur[i] = a.p[(size_t)pair * 8 + i];
    lam = a.lambda_it[pair];
    it = a.niter[pair];
  }
  gather_partials(a.partials, slot, a.tiles, s, scratch + kAccStride, 0, a.invert_h ? kAccStride : kMaxN);
  const int n = a.n;
  const double* hinv = Hi;
  if (!a.invert_h) {
    if (threadIdx.x < kMaxN * kMaxN) Hi[threadIdx.x] = hpre;
    __syncthreads();
  }
  if (a.invert_h) {
    // inv(H) like np.linalg.inv (derivatives.py:110-130; zeros if singular): wave 0 factorises (one lane per row), then
    // the n columns of the inverse are solved by n threads (same arithmetic per element as the serial lu_inverse)
    double* fa = ws;                       // [n][kMaxN] factors
    double* pivd = fa + kMaxN * kMaxN;     // [kMaxN]
    double* xs = pivd + kMaxN;             // [n][kMaxN] per-column scratch
    if (threadIdx.x < 64) {  // wave 0: expand the packed upper triangle, then factorise with one lane per row
      const int k = threadIdx.x >> 3, m = threadIdx.x & 7;
      if (k < n && m < n) Hm[k * kMaxN + m] = s[kMaxN + (k <= m ? tri_index(k, m, n) : tri_index(m, k, n))];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const bool okf = lu_factor_wave(n, Hm, kMaxN, fa, pivd, threadIdx.x);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (threadIdx.x == 0) Hm[kMaxN * kMaxN - 1] = okf ? 1.0 : 0.0;  // (H has been copied into the factors by now)
    }
    __syncthreads();
    const bool ok = Hm[kMaxN * kMaxN - 1] != 0.0;
    if ((int)threadIdx.x < n) {
      if (ok) {
        lu_inverse_column(n, fa, pivd, threadIdx.x, xs + threadIdx.x * kMaxN, Hi, kMaxN);
      } else {
        for (int i = 0; i < n; ++i) Hi[i * kMaxN + threadIdx.x] = 0.0;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  if (a.invert_h && a.store_hinv)
    for (int i = 0; i < kMaxN * kMaxN; ++i) a.Hinv[(size_t)pair * 64 + i] = Hi[i];
  double dp[kMaxN], e2 = 0.0;
  for (int i = 0; i < n; ++i) {  // image_optimisation.py:146-155
    double v = 0.0;
    for (int j = 0; j < n; ++j) v += hinv[i * kMaxN + j] * s[j];
    dp[i] = v;
    e2 += v * v;
  }
  const double err = sqrt(e2);
  double* p = a.p + (size_t)pair * 8;
  double* pp = a.p_prev + (size_t)pair * 8;
#pragma unroll
  for (int i = 0; i < 8; ++i) pp[i] = pcur[i];
  update_transform(pcur, dp, a.tt, (a.flags & ICA_FLAG_EXACT_UPDATE) != 0);  // transformation.py:35-141
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = pcur[i];
  if (a.robust && a.lambda_user <= 0.0 && lam > a.lambdaN) {  // ref :235-238 (decay after use)
    lam *= a.lambda_ratio;
    if (lam < a.lambdaN) lam = a.lambdaN;
    a.lambda_it[pair] = lam;
  }
  if (a.trace) {
    double* tr = a.trace + (((size_t)pair * a.nscales + a.level) * a.max_iter + it) * ICA_TRACE_STRIDE;
    tr[0] = err;
    for (int i = 0; i < 8; ++i) tr[1 + i] = (i < n) ? pcur[i] : 0.0;
    tr[9] = lam;
  }
  a.error[pair] = err;
  a.niter[pair] = it + 1;
  if (err > a.tol && it + 1 < a.max_iter) {  // ref :109 / :225
    list_append(a.nxt_list, a.nxt_count, a.list_cap, slot, pair);
  }
}

__global__ __launch_bounds__(kSolveThreads) void ica_solve_kernel(SolveArgs a) {
  __shared__ double scratch[kSolveScratch];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    list_clear(a.clr_count);
    if (a.count_hist) a.count_hist[a.level * a.max_iter + a.iter] = list_total(a.cur_count);
  }
  if ((int)blockIdx.x >= list_total(a.cur_count)) return;
  solve_pair(a, blockIdx.x, list_entry(a.cur_list, a.cur_count, a.list_cap, blockIdx.x), scratch);
}

struct LevelInitArgs {
  int batch, tt, level, nscales;
  int w_coarse, h_coarse, w_fine, h_fine;  // sizes for zoom_in_parameters (coarse = level+1)
  double lambda_user, lambda0;
  double* p;
  double* lambda_it;
  double* error;
  int* niter;
  int* niter_out;      // [B][nscales]: receives the finished (level+1) counts
  int* list0; int* counts;  // 3 lists x kListShards shard counters
};

// Start of a level (ref :106-107, :221-223, :327,:337,:371-372): p from the coarser level through
// zoom_in_parameters (or zeros / the caller's p at the coarsest level), iteration state reset.
__global__ void ica_level_init_kernel(LevelInitArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < 3 * kListShards) {  // list 0 = all pairs (pair i in shard i & 7), lists 1 and 2 empty
    const int l = b / kListShards, sh = b % kListShards;
    a.counts[(l * kListShards + sh) * kShardStride] = l == 0 ? (a.batch - sh + kListShards - 1) / kListShards : 0;
  }
  if (b >= a.batch) return;
  double* p = a.p + (size_t)b * 8;
  if (a.level == a.nscales - 1) {
    if (a.nscales > 1)
      for (int i = 0; i < 8; ++i) p[i] = 0.0;
    else
      for (int i = nparams_of(a.tt); i < 8; ++i) p[i] = 0.0;
  } else {
    a.niter_out[(size_t)b * a.nscales + a.level + 1] = a.niter[b];
    zoom_in_parameters(p, a.tt, a.w_coarse, a.h_coarse, a.w_fine, a.h_fine);
  }
  a.lambda_it[b] = a.lambda_user > 0.0 ? a.lambda_user : a.lambda0;
  a.error[b] = 1e10;
  a.niter[b] = 0;
  a.list0[(b & (kListShards - 1)) * list_shard_cap(a.batch) + (b >> 3)] = b;
}

__global__ void ica_finish_kernel(int batch, int nscales, const int* niter, const double* error, int* niter_out,
                                  double* error_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  niter_out[(size_t)b * nscales] = niter[b];
  error_out[b] = error[b];
}

// ---- warp operator (Iw / DI write-out, data generation) -------------------------------------------
template <int C, class TIn>
__global__ __launch_bounds__(kBlock) void ica_warp_kernel(const TIn* __restrict__ I2, const TIn* __restrict__ I1,
                                                          int H, int W, int tt, const double* __restrict__ p,
                                                          const float* __restrict__ minmax, float* __restrict__ Iw,
                                                          float* __restrict__ DI) {
  const int pair = blockIdx.z;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * kWaves + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  double ad[8];
  params_to_displacement(p + (size_t)pair * 8, tt, ad);
  float A[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) A[i] = (float)ad[i];
  float dx, dy;
  if (tt == ICA_HOMOGRAPHY) displacement<true>(A, (float)x, (float)y, dx, dy);
  else displacement<false>(A, (float)x, (float)y, dx, dy);
  const size_t plane = (size_t)H * W;
  const size_t base = (size_t)pair * C * plane;
  float v[C];
  const bool ok = warp_sample<C, TIn>(I2 + base, H, W, x, y, dx, dy, minmax[pair * 2], minmax[pair * 2 + 1], v,
                                 p + (size_t)pair * 8, tt);
  const float nanv = __builtin_nanf("");
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const size_t o = base + c * plane + (size_t)y * W + x;
    const float iw = ok ? v[c] : nanv;
    if (Iw) Iw[o] = iw;
    if (DI) DI[o] = iw - (float)I1[o];
  }
}

// ---- transform_image (transformation.py:266-318): bilinear resampling under a 3x3 matrix, cval 0, clipped ----
template <int C>
__global__ __launch_bounds__(kBlock) void ica_transform_image_kernel(const float* __restrict__ I, int H, int W,
                                                                     const double* __restrict__ M,
                                                                     const float* __restrict__ minmax,
                                                                     float* __restrict__ out) {
  const int pair = blockIdx.z;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * kWaves + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  const double* m = M + (size_t)pair * 9;
  // test-data generation, not the hot path: positions in fp64 like skimage
  const double w = m[6] * x + m[7] * y + m[8];
  const double xs = (m[0] * x + m[1] * y + m[2]) / w, ys = (m[3] * x + m[4] * y + m[5]) / w;
  const double x0 = floor(xs), y0 = floor(ys), x1 = ceil(xs), y1 = ceil(ys);
  const float dx = (float)(xs - x0), dy = (float)(ys - y0);
  const size_t plane = (size_t)H * W;
  const float lo = minmax[pair * 2], hi = minmax[pair * 2 + 1];
  auto px = [&](const float* q, double r, double c) -> float {
    return (r >= 0 && r < H && c >= 0 && c < W) ? q[(size_t)r * W + (size_t)c] : 0.0f;
  };
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float* q = I + ((size_t)pair * C + c) * plane;
    const float top = (1.0f - dx) * px(q, y0, x0) + dx * px(q, y0, x1);
    const float bot = (1.0f - dx) * px(q, y1, x0) + dx * px(q, y1, x1);
    const float v = (1.0f - dy) * top + dy * bot;
    out[((size_t)pair * C + c) * plane + (size_t)y * W + x] = fminf(fmaxf(v, lo), hi);
  }
}

// ---- min / max of each image ----------------------------------------------------------------------
template <class TIn>
__global__ __launch_bounds__(kBlock) void ica_minmax_partial_kernel(const TIn* __restrict__ img, size_t elems,
                                                                    int chunks, float* __restrict__ part) {
  const int pair = blockIdx.y, chunk = blockIdx.x;
  const TIn* q = img + (size_t)pair * elems;
  const size_t per = (elems + chunks - 1) / chunks;
  const size_t beg = (size_t)chunk * per, end = min(elems, beg + per);
  float lo = INFINITY, hi = -INFINITY;
  for (size_t i = beg + threadIdx.x; i < end; i += kBlock) {
    const float v = (float)q[i];
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = fminf(lo, __shfl_down(lo, off, 64));
    hi = fmaxf(hi, __shfl_down(hi, off, 64));
  }
  __shared__ float slo[kWaves], shi[kWaves];
  if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kWaves; ++w) { lo = fminf(lo, slo[w]); hi = fmaxf(hi, shi[w]); }
    part[((size_t)pair * chunks + chunk) * 2] = lo;
    part[((size_t)pair * chunks + chunk) * 2 + 1] = hi;
  }
}

// One wave per pair: the lanes stride over the pair's partials (coalesced), then a wave reduction.
__global__ __launch_bounds__(kBlock) void ica_minmax_final_kernel(const float* __restrict__ part, int chunks, int batch,
                                                                  float* __restrict__ out) {
  const int b = blockIdx.x * kWaves + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= batch) return;  // whole wave
  const float2* q = reinterpret_cast<const float2*>(part) + (size_t)b * chunks;
  float lo = INFINITY, hi = -INFINITY;
  for (int c = lane; c < chunks; c += 64) {
    const float2 v = q[c];
    lo = fminf(lo, v.x);
    hi = fmaxf(hi, v.y);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = fminf(lo, __shfl_down(lo, off, 64));
    hi = fmaxf(hi, __shfl_down(hi, off, 64));
  }
  if (lane == 0) {
    out[b * 2] = lo;
    out[b * 2 + 1] = hi;
  }
}

// Range of a fused pyramid step's CLIPPED output from its tiles' unclipped ranges: clip(x, lo, hi) maps [a, b] to
// [max(a, lo), min(b, hi)] (a <= hi and b >= lo always hold here: the operator reproduces constants).  One wave per pair.
__global__ __launch_bounds__(kBlock) void ica_minmax_clipped_kernel(const float* __restrict__ part, int chunks, int batch,
                                                                    const float* __restrict__ clip, float* __restrict__ out) {
  const int b = blockIdx.x * kWaves + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= batch) return;  // whole wave
  const float2* q = reinterpret_cast<const float2*>(part) + (size_t)b * chunks;
  float lo = INFINITY, hi = -INFINITY;
  for (int c = lane; c < chunks; c += 64) {
    const float2 v = q[c];
    lo = fminf(lo, v.x);
    hi = fmaxf(hi, v.y);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = fminf(lo, __shfl_down(lo, off, 64));
    hi = fmaxf(hi, __shfl_down(hi, off, 64));
  }
  if (lane == 0) {
    out[b * 2] = fminf(fmaxf(lo, clip[b * 2]), clip[b * 2 + 1]);
    out[b * 2 + 1] = fminf(fmaxf(hi, clip[b * 2]), clip[b * 2 + 1]);
  }
}

// ---- per-operator hooks -----------------------------------------------------------------------------
__global__ void ica_gather_normal_equations_kernel(const double* partials, int tiles, int n, int batch, int have_b,
                                                   int have_h, const double* Hsetup_partials, double* Hout,
                                                   double* bout) {
  __shared__ double s[kAccStride], part[4 * kAccStride];
  const int pair = blockIdx.x;
  gather_partials(partials, pair, tiles, s, part);
  if (threadIdx.x == 0) {
    for (int i = 0; i < 8; ++i) bout[(size_t)pair * 8 + i] = (have_b && i < n) ? s[i] : 0.0;
    if (have_h) {
      for (int i = 0; i < 64; ++i) Hout[(size_t)pair * 64 + i] = 0.0;
      for (int k = 0; k < n; ++k)
        for (int m = k; m < n; ++m)
          Hout[(size_t)pair * 64 + k * 8 + m] = Hout[(size_t)pair * 64 + m * 8 + k] = s[kMaxN + tri_index(k, m, n)];
    }
  }
}

__global__ void ica_solve_update_kernel(int batch, int tt, int flags, const double* H, const double* b, double* p,
                                        double* dp_out, double* err_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch) return;
  const int n = nparams_of(tt);
  double Hi[64], ws[kLuWork];  // test hook: one matrix per thread, private workspace
  lu_inverse(n, H + (size_t)i * 64, Hi, 8, ws);
  double dp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, e2 = 0.0;
  for (int r = 0; r < n; ++r) {
    double v = 0.0;
    for (int c = 0; c < n; ++c) v += Hi[r * 8 + c] * b[(size_t)i * 8 + c];
    dp[r] = v;
    e2 += v * v;
  }
  update_transform(p + (size_t)i * 8, dp, tt, (flags & ICA_FLAG_EXACT_UPDATE) != 0);
  if (dp_out)
    for (int r = 0; r < 8; ++r) dp_out[(size_t)i * 8 + r] = dp[r];
  if (err_out) err_out[i] = sqrt(e2);
}

__global__ void ica_zoom_params_kernel(int batch, int tt, int w, int h, int ww, int hh, double* p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < batch) zoom_in_parameters(p + (size_t)i * 8, tt, w, h, ww, hh);
}

// ---- pyramid: separable banded resampling (Gaussian + B-spline prefilter + zoom folded into one
//      operator per axis on the host, see ica_pyramid.hpp) ------------------------------------------
struct BandOp {
  const int* start;    // [n_out] first input index of the band
  const float* wT;     // [K][n_out] weights (horizontal pass of the generic fallback)
  const int* vj0;      // [groups] first input index of each group of kOutsV outputs
  const int* vspan;    // [groups] input samples spanned by the group
  const int* vown0;    // [groups] the groups' windows overlap; [vown0, vown1) partitions the input axis so that the
  const int* vown1;    //          vertical pass can fold the image's min/max into its single read
  const float* vw;     // [slots][kVWin][kOutsV] zero-padded weight windows, kOutsV outputs interleaved
  const int* vwslot;   // [groups] the group's window.  Groups with the same window share a slot: away from the image
                       // border a decimation by an integer factor is shift-invariant, the whole table is then a few KB
                       // and the weight fetches (scalar loads, one row of weights per input row) stay in the scalar cache
  int n_in, n_out, K;
  // Decimation by exactly 2: away from the border every output has the same kFastK weights, `sw` (made bit-identical on the
  // host), and the groups whose window is slot `vfast_slot` consist of such outputs only (-1: none)
  const float* sw;
  int vfast_slot;
};
constexpr int kFastK = 26;   // taps of the factor-2 operator (Gaussian 0.5 + B-spline prefilter + zoom)
template <int J> using IC_ = std::integral_constant<int, J>;
template <int J0, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {   // f(IC_<J0>{}), f(IC_<J0 + 1>{}), ... N calls, indices compile-time
  if constexpr (N > 0) {
    f(IC_<J0>{});
    static_for<J0 + 1, N - 1>(f);
  }
}

#ifndef ICA_OUTSV
#define ICA_OUTSV 16
#endif
constexpr int kOutsV = ICA_OUTSV;   // consecutive output rows per wave in the vertical pass
constexpr int kVWin = 160;   // max input samples spanned by 16 consecutive outputs (16*f + K), checked on the host
constexpr int kRowsH = 32;   // rows per workgroup in the horizontal pass (8 per wave)

// Vertical pass: in [planes][n_in][W] -> out [planes][n_out][W]  (+ optional min/max of `in` per pair).
// A lane owns VEC adjacent columns (VEC = 4: one 16-byte load per input row and lane, 1 KiB per wave -- with 4-byte
// loads the pass sits at the rate the texture addresser takes dword loads, not at a memory rate); each wave produces
// kOutsV consecutive output rows, so every input value is loaded once and feeds kOutsV FMAs whose weights are
// wave-uniform: they come through the scalar cache into SGPRs (s_load_dwordx8/x16 per input row), the vector ALU
// only sees v_fmac / v_pk_fma(acc, sgpr, value).
// The operator weights are float64 on the host (ica_pyramid.hpp); they are applied in fp32 like every other
// per-pixel quantity of the engine (the result is stored as fp32 anyway).
// TIn = unsigned char: the input is a uint8 level-0 image (4 columns per lane = one dword load); converted on load.
template <int VEC, class TIn>
__global__ __launch_bounds__(kBlock) void ica_resample_v_kernel(const TIn* __restrict__ in, float* __restrict__ out,
                                                                int planes, int channels, int W, BandOp op,
                                                                float* __restrict__ mm_part) {
  typedef float vecf __attribute__((ext_vector_type(VEC)));
  const int lane = threadIdx.x & 63;
  const int x = (blockIdx.x * 64 + lane) * VEC;  // (VEC > 1 only when W % VEC == 0: a lane is in or out as a whole)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = blockIdx.y * kWaves + wave;
  const int o0 = g * kOutsV;
  const int pl = blockIdx.z;
  if (o0 >= op.n_out) return;  // whole wave
  const int no = min(kOutsV, op.n_out - o0);
  const int j0 = op.vj0[g], span = op.vspan[g];
  const int own0 = mm_part ? op.vown0[g] - j0 : 0, own1 = mm_part ? op.vown1[g] - j0 : 0;
  const float* __restrict__ wz = op.vw + (size_t)op.vwslot[g] * kVWin * kOutsV;
  float acc[kOutsV][VEC];
#pragma unroll
  for (int i = 0; i < kOutsV; ++i)
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[i][v] = 0.f;
  const bool xin = x < W;
  const TIn* col = in + (size_t)pl * op.n_in * W + (size_t)j0 * W + (xin ? x : 0);
  float lo = INFINITY, hi = -INFINITY;
  auto run = [&](int a, int b, auto track_c) {  // rows [a, b) of the window; track_c: fold them into min/max
    constexpr bool TRACK = decltype(track_c)::value;
#pragma unroll 4
    for (int jj = a; jj < b; ++jj) {
      float val[VEC];
      if constexpr (VEC == 1) {
        val[0] = (float)col[(size_t)jj * W];
      } else if constexpr (std::is_same_v<TIn, float>) {
        const vecf t = *reinterpret_cast<const vecf*>(col + (size_t)jj * W);
#pragma unroll
        for (int v = 0; v < VEC; ++v) val[v] = t[v];
      } else {
        static_assert(VEC == 4, "uint8 rows are read 4 pixels (one dword) per lane");
        const unsigned t = *reinterpret_cast<const unsigned*>(col + (size_t)jj * W);
        val[0] = (float)(t & 0xffu); val[1] = (float)((t >> 8) & 0xffu); val[2] = (float)((t >> 16) & 0xffu); val[3] = (float)(t >> 24);
      }
      if constexpr (TRACK) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) { lo = fminf(lo, val[v]); hi = fmaxf(hi, val[v]); }
      }
#pragma unroll
      for (int i = 0; i < kOutsV; ++i) {
        const float w = wz[jj * kOutsV + i];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[i][v] = fmaf(w, val[v], acc[i][v]);
      }
    }
  };
  run(0, own0, std::false_type{});
  run(own0, own1, std::true_type{});
  run(own1, span, std::false_type{});
  if (xin) {
#pragma unroll
    for (int i = 0; i < kOutsV; ++i) {
      if (i < no) {
        float* dst = out + ((size_t)pl * op.n_out + o0 + i) * W + x;
        if constexpr (VEC == 1) {
          dst[0] = acc[i][0];
        } else {
          vecf t;
#pragma unroll
          for (int v = 0; v < VEC; ++v) t[v] = acc[i][v];
          *reinterpret_cast<vecf*>(dst) = t;
        }
      }
    }
  }
  if (mm_part) {  // one partial per wave, reduced by ica_minmax_final_kernel (no atomics: 256 waves share a pair)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo = fminf(lo, __shfl_down(lo, off, 64));
      hi = fmaxf(hi, __shfl_down(hi, off, 64));
    }
    if (lane == 0) {
      const int ngroups = (op.n_out + kOutsV - 1) / kOutsV;
      const size_t slot = ((size_t)pl * ngroups + g) * gridDim.x + blockIdx.x;  // [plane][group][column block]
      mm_part[slot * 2] = lo;
      mm_part[slot * 2 + 1] = hi;
    }
  }
}

// Horizontal pass + clip: in [planes][H][n_in] -> out [planes][H][n_out].
// A workgroup stages 32 rows x (input span of 64 outputs) in LDS with coalesced loads; lane = output column,
// each thread accumulates 8 rows so one (coalesced, transposed) weight load feeds 8 FMAs.  The staged rows are
// stored de-interleaved (even input columns, then odd ones): with a decimation factor of 2 consecutive lanes
// read input columns 2 apart, which this layout turns into consecutive LDS words (no 2-way bank conflict); any
// other factor is still correct, only less regular.
template <bool PAIRS>
__global__ __launch_bounds__(kBlock) void ica_resample_h_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                int planes, int channels, int H, BandOp op,
                                                                const float* __restrict__ minmax, int span_pad) {
  extern __shared__ float tile[];  // [kRowsH][span_pad]: per row, even columns at [0, half), odd ones at [half, ...)
  const int o_base = blockIdx.x * 64, r_base = blockIdx.y * kRowsH, pl = blockIdx.z;
  const int o_last = min(o_base + 63, op.n_out - 1);
  const int c0 = op.start[o_base];
  const int span = op.start[o_last] + op.K - c0;  // <= span_pad
  const int half = (span_pad + 1) >> 1;
  const float* src = in + (size_t)pl * H * op.n_in;
  if constexpr (PAIRS) {
    // staging with 8-byte loads: a lane reads the input columns (2j, 2j+1) of the global image -- whose relative
    // position c = x - c0 decides the half of the de-interleaved row they go to -- 8 rows x 2 segments of 128 columns
    // per wave in flight together (n_in is even and the rows 8-byte aligned, host-checked)
    const int lane_c = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x0 = c0 & ~1;  // first global column pair
    constexpr int kSeg = 2;
    for (int cs = 0; cs < span + 1; cs += 128 * kSeg) {
      float2 v[kRowsH / kWaves][kSeg];
#pragma unroll
      for (int i = 0; i < kRowsH / kWaves; ++i) {
        const int y = r_base + wv + i * kWaves;
#pragma unroll
        for (int j = 0; j < kSeg; ++j) {
          const int xg = x0 + cs + j * 128 + 2 * lane_c;  // even; xg + 1 < n_in whenever xg < n_in
          v[i][j] = (xg - c0 < span && y < H && xg < op.n_in)
                        ? *reinterpret_cast<const float2*>(src + (size_t)y * op.n_in + xg) : make_float2(0.f, 0.f);
        }
      }
#pragma unroll
      for (int i = 0; i < kRowsH / kWaves; ++i) {
        const int r = wv + i * kWaves;
#pragma unroll
        for (int j = 0; j < kSeg; ++j) {
          const int c = x0 + cs + j * 128 + 2 * lane_c - c0;  // -1 or even first lane when c0 is odd
          if (c >= 0 && c < span) tile[r * span_pad + (c & 1) * half + (c >> 1)] = v[i][j].x;
          if (c + 1 >= 0 && c + 1 < span) tile[r * span_pad + ((c + 1) & 1) * half + ((c + 1) >> 1)] = v[i][j].y;
        }
      }
    }
  } else {
    // staging: 8 rows x up to 4 column segments per wave are read into registers first (all loads in flight
    // together), then written to LDS
    const int lane_c = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int kSeg = 3;  // 64-column segments per row kept in flight (span <= 192 in one go; wider spans loop)
    for (int cs = 0; cs < span; cs += 64 * kSeg) {
      float v[kRowsH / kWaves][kSeg];
#pragma unroll
      for (int i = 0; i < kRowsH / kWaves; ++i) {
        const int y = r_base + wv + i * kWaves;
#pragma unroll
        for (int j = 0; j < kSeg; ++j) {
          const int c = cs + j * 64 + lane_c, xg = c0 + c;
          v[i][j] = (c < span && y < H && xg < op.n_in) ? src[(size_t)y * op.n_in + xg] : 0.0f;
        }
      }
#pragma unroll
      for (int i = 0; i < kRowsH / kWaves; ++i) {
        const int r = wv + i * kWaves;
#pragma unroll
        for (int j = 0; j < kSeg; ++j) {
          const int c = cs + j * 64 + lane_c;
          if (c < span) tile[r * span_pad + (c & 1) * half + (c >> 1)] = v[i][j];
        }
      }
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int o = o_base + lane;
  if (o >= op.n_out) return;
  const int s = op.start[o] - c0;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  const float* t0 = tile + (rg * 8) * span_pad;
#pragma unroll 2
  for (int k = 0; k < op.K; ++k) {
    const float wk = op.wT[(size_t)k * op.n_out + o];
    const int c = s + k;
    const float* q = t0 + (c & 1) * half + (c >> 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = fmaf(wk, q[i * span_pad], acc[i]);
  }
  const int pair = pl / channels;
  const float lo = minmax[pair * 2], hi = minmax[pair * 2 + 1];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int y = r_base + rg * 8 + i;
    if (y < H) out[((size_t)pl * H + y) * op.n_out + o] = fminf(fmaxf(acc[i], lo), hi);
  }
}

// ---- both passes in one kernel --------------------------------------------------------------------------------------
// The two-pass pyramid step moves 9 bytes per input pixel at factor 2 (read 4, write 2, read 2, write 1); producing the
// level in one kernel moves 5.  A wave takes kOutsV output rows x one column block: the vertical pass exactly as above
// over the block's kFusedW input columns (4 per lane), whose result -- 16 rows x 256 columns -- never leaves the CU: it
// goes through a wave-private LDS tile (8 rows at a time, de-interleaved like the horizontal pass's) and the same lanes
// apply the horizontal operator, two output columns each.  Column blocks overlap by the horizontal band (K - f of 256
// columns: 10 % at factor 2), which the L2 serves.
//
// The clip range is the min/max of the INPUT image, which this very kernel is still computing (folded into its only read
// of the input, like the vertical pass).  So the output is written unclipped together with each wave's own output
// range; once the pair's range is final (ica_minmax_final_kernel), ica_resample_clip_kernel revisits only the
// (wave) tiles whose range sticks out of it -- in practice the few that contain an overshoot next to a saturated
// region -- and clips them.  Same values as clipping every pixel, without a second pass over the data.
constexpr int kFusedW = 256;     // input columns per column block
#ifndef ICA_FUSED_HALF
#define ICA_FUSED_HALF 8
#endif
constexpr int kFusedHalf = ICA_FUSED_HALF;    // rows per horizontal half pass (LDS: 8 x 256 floats per wave)

constexpr int kPairWin = 28;      // input columns a lane reads for its two adjacent outputs (7 aligned groups of 4)
struct FusedCols {
  const int* c0;   // [blocks] first input column of each block (multiple of 4)
  int nb;          // output columns per block (<= 128: two per lane)
  // Adjacent-output horizontal pass (nullptr: the operator is not regular enough for it, see get_op): lane l of block b
  // produces outputs b * nb + 2l and + 2l + 1 from the kPairWin columns that start at c0[b] + 4 * (l + hoff[b]);
  // wpair[((b * kPairWin + j) * 64 + l) * 2 + q] = weight of window column j for output q of the pair (0 outside its band)
  const float* wpair;
  const int* hoff; // [blocks] window origin of lane 0 relative to c0, in groups of 4 columns (-3 .. 1)
};

// PAIR: the horizontal operator is applied by the adjacent-output pass (FusedCols::wpair), else by the generic one -- two
// instantiations, because one kernel holding both paid for the larger of the two in registers (measured: 4.5 instead of
// 2.1 ms for the generic pass of the 1024^2 level).
#ifndef ICA_PAIR_WAVES   // waves per SIMD the register allocator leaves room for in the adjacent-output instantiation (119
                         // registers unconstrained; forced to 96 for a fifth wave it spills 20 bytes and is 1-4 % slower)
#define ICA_PAIR_WAVES 4
#endif
template <class TIn, bool PAIR>
__global__ __launch_bounds__(kBlock, (PAIR ? ICA_PAIR_WAVES : 1)) void ica_resample_fused_kernel(const TIn* __restrict__ in, float* __restrict__ out,
                                                                    int W, BandOp oy, BandOp ox, FusedCols fc,
                                                                    float* __restrict__ mm_part, float* __restrict__ out_part) {
  typedef float vec4f __attribute__((ext_vector_type(4)));
  typedef float vec2f __attribute__((ext_vector_type(2)));
  constexpr int kTileFloats = kFusedHalf * kFusedW;   // a wave-private tile: kFusedHalf rows x kFusedW columns
  __shared__ __attribute__((aligned(16))) float tiles[kWaves * kTileFloats];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = blockIdx.y * kWaves + wave;
  const int o0 = g * kOutsV;
  const int pl = blockIdx.z;
  if (o0 >= oy.n_out) return;  // whole wave (the LDS tiles are wave-private: no workgroup barrier anywhere)
  const int no = min(kOutsV, oy.n_out - o0);
  const int c0 = fc.c0[blockIdx.x];
  const int x = c0 + 4 * lane;
  const int j0 = oy.vj0[g], span = oy.vspan[g];
  const int own0 = mm_part ? oy.vown0[g] - j0 : 0, own1 = mm_part ? oy.vown1[g] - j0 : 0;
  const float* __restrict__ wz = oy.vw + (size_t)oy.vwslot[g] * kVWin * kOutsV;
  float acc[kOutsV][4];
#pragma unroll
  for (int i = 0; i < kOutsV; ++i)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[i][v] = 0.f;
  const bool xin = x < W;  // (W % 4 == 0: a lane's four columns are in or out together)
  const TIn* col = in + (size_t)pl * oy.n_in * W + (size_t)j0 * W + (xin ? x : 0);
  float lo = INFINITY, hi = -INFINITY;
  float* const tile = tiles + wave * kTileFloats;   // tile[i * kFusedW + c]
  // (Built and measured instead of the plain loads below: the rows through a ring of 8 LDS-DMA slots in the wave's idle
  //  tile, 8 KB in flight per wave and no registers spent.  Next to the DMA intrinsic the compiler no longer keeps the
  //  wave-uniform weights on the scalar path, and with hand-written s_load + wait per row the scalar-cache misses of
  //  the 3.6 KB weight window sit exposed in every row: 2.36 ms against 1.95 ms for the 1024^2 level.)
#ifndef ICA_FUSED_UNROLL
#define ICA_FUSED_UNROLL 4
#endif
  // (Also built and measured, none of them moving the 1024^2 level off ~2.0 ms: 2 / 6 / 8 rows of loads in flight; 8
  //  output rows per wave at 64 registers and twice the waves; the block's horizontal weights staged in LDS; odd groups
  //  walking their window bottom-up so that the rows two neighbours share are read close in time.  Counters
  //  (profiles/r02_pyramid_fused_counters.txt): 1.7x the input fetched from HBM -- the overlap of the windows, 57 rows
  //  per 32 new ones, mostly misses the L2 -- at 3.7 TB/s, the vector ALU half busy with the zero-padded windows' 2.3x
  //  multiply-adds, the LDS pipe a quarter busy: three moderately loaded resources that do not overlap well in one wave.)
  auto run = [&](int a, int b, auto track_c) {
    constexpr bool TRACK = decltype(track_c)::value;
#pragma unroll ICA_FUSED_UNROLL
    for (int jj = a; jj < b; ++jj) {
      float val[4];
      if constexpr (std::is_same_v<TIn, float>) {
        const vec4f v4 = *reinterpret_cast<const vec4f*>(col + (size_t)jj * W);
#pragma unroll
        for (int v = 0; v < 4; ++v) val[v] = v4[v];
      } else {
        const unsigned u = *reinterpret_cast<const unsigned*>(col + (size_t)jj * W);
        val[0] = (float)(u & 0xffu); val[1] = (float)((u >> 8) & 0xffu); val[2] = (float)((u >> 16) & 0xffu); val[3] = (float)(u >> 24);
      }
      if constexpr (TRACK) {
#pragma unroll
        for (int v = 0; v < 4; ++v) { lo = fminf(lo, val[v]); hi = fmaxf(hi, val[v]); }
      }
#pragma unroll
      for (int i = 0; i < kOutsV; ++i) {
        const float w = wz[jj * kOutsV + i];
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[i][v] = fmaf(w, val[v], acc[i][v]);
      }
    }
  };
  // Interior groups of a decimation by 2: output i of the group takes rows 2 i .. 2 i + 25 of the window with the same 26
  // weights, so row jj only reaches the outputs (jj - 25) / 2 .. jj / 2 -- 26 x 16 multiply-adds per column instead of the
  // 56 x 16 of the zero-padded window (the vertical pass's multiply-adds, on scalar weights, were two thirds of this
  // kernel's vector time).  Every index is compile-time; the terms of an output are added in the same order.
  constexpr int kFastSpan = 2 * (kOutsV - 1) + kFastK;
  bool fast = false;
  if constexpr (PAIR) fast = oy.vfast_slot >= 0 && oy.vwslot[g] == oy.vfast_slot && span == kFastSpan && no == kOutsV;
  if (fast) {
    if (xin) {
      float wv[kFastK];   // wave-uniform (scalar registers)
#pragma unroll
      for (int k = 0; k < kFastK; ++k)
        wv[k] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, oy.sw[k])));
      typedef std::conditional_t<std::is_same_v<TIn, float>, vec4f, unsigned> row_t;
      constexpr int kAhead = 4;   // rows of loads in flight (a queue in registers with compile-time slots)
      row_t queue[kAhead];
#pragma unroll
      for (int q = 0; q < kAhead; ++q) queue[q] = *reinterpret_cast<const row_t*>(col + (size_t)q * W);
      static_for<0, kFastSpan>([&](auto jj_c) {
        constexpr int jj = decltype(jj_c)::value;
        const row_t raw = queue[jj % kAhead];
        if constexpr (jj + kAhead < kFastSpan) queue[jj % kAhead] = *reinterpret_cast<const row_t*>(col + (size_t)(jj + kAhead) * W);
        float val[4];
        if constexpr (std::is_same_v<TIn, float>) {
#pragma unroll
          for (int v = 0; v < 4; ++v) val[v] = raw[v];
        } else {
          val[0] = (float)(raw & 0xffu); val[1] = (float)((raw >> 8) & 0xffu); val[2] = (float)((raw >> 16) & 0xffu); val[3] = (float)(raw >> 24);
        }
        if (jj >= own0 && jj < own1) {   // wave-uniform
#pragma unroll
          for (int v = 0; v < 4; ++v) { lo = fminf(lo, val[v]); hi = fmaxf(hi, val[v]); }
        }
        constexpr int ilo = jj > kFastK - 1 ? (jj - (kFastK - 1) + 1) / 2 : 0;
        constexpr int ihi = jj / 2 < kOutsV - 1 ? jj / 2 : kOutsV - 1;
#pragma unroll
        for (int i = ilo; i <= ihi; ++i) {
#pragma unroll
          for (int v = 0; v < 4; ++v) acc[i][v] = fmaf(wv[jj - 2 * i], val[v], acc[i][v]);
          // (the sums are formed HERE, as the row arrives: the optimiser would sink them to where an output completes
          //  and keep 26 input rows in registers instead)
          asm volatile("" : "+v"(acc[i][0]), "+v"(acc[i][1]), "+v"(acc[i][2]), "+v"(acc[i][3]));
        }
        __builtin_amdgcn_sched_barrier(0);   // (keeps the queue at kAhead rows)
      });
    }
  } else
  if (xin) {  // (columns past the image stay zero: the horizontal weights that reach them are zero as well)
    run(0, own0, std::false_type{});
    run(own0, own1, std::true_type{});
    run(own1, span, std::false_type{});
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // ---- horizontal operator on the wave's 16 x 256 block, 8 rows at a time through LDS ----
  const int ob = blockIdx.x * fc.nb, oe = min(ob + fc.nb, ox.n_out);
  float olo = INFINITY, ohi = -INFINITY;
#pragma unroll
  for (int hh = 0; hh < kOutsV / kFusedHalf; ++hh) {
    if constexpr (PAIR) {
      // Adjacent-output pass (regular operators: decimation by 2).  A lane produces outputs ob + 2 lane and ob + 2 lane + 1,
      // whose bands lie inside the same 28 consecutive columns: the rows go to LDS as they are (one ds_write_b128 per
      // lane and row) and come back as 7 aligned ds_read_b128 per lane and row -- 28 tile values for 2 x 26 taps, where
      // one output per read needs 52 ds_read_b32 (the LDS reads were what bounded the pyramid step).  Per-lane weights
      // on the window's columns (zero outside an output's band), so image borders need nothing special; the terms of
      // an output are added in the same order as in the generic pass below (ascending column, zero-weight terms change
      // nothing): bit-identical results.
#pragma unroll
      for (int i = 0; i < kFusedHalf; ++i) {
        vec4f v4;
#pragma unroll
        for (int v = 0; v < 4; ++v) v4[v] = acc[hh * kFusedHalf + i][v];
        *reinterpret_cast<vec4f*>(&tile[i * kFusedW + 4 * lane]) = v4;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int o = ob + 2 * lane;
      const float* wp = fc.wpair + ((size_t)blockIdx.x * kPairWin * 64 + lane) * 2;
      // group m of the lane's window sits at group lane + hoff + m of a tile row.  Groups before the row's first or
      // after its last one (block 0's left edge, the masked lanes at the right) carry zero weights: any finite value
      // does for them, so the group index is clamped into the row instead of padding the tile (the 32 KB of four
      // tiles are what lets five workgroups share a CU).
      const int g0 = lane + fc.hoff[blockIdx.x];
      float aA[kFusedHalf], aB[kFusedHalf];
#pragma unroll
      for (int i = 0; i < kFusedHalf; ++i) { aA[i] = 0.f; aB[i] = 0.f; }
      vec2f wcur[4], wnxt[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) wcur[t] = *reinterpret_cast<const vec2f*>(wp + (size_t)t * 128);
#pragma unroll
      for (int m = 0; m < kPairWin / 4; ++m) {
        if (m + 1 < kPairWin / 4) {   // the next group's weights (per-lane global loads, L1 / L2 hits) under this group's arithmetic
#pragma unroll
          for (int t = 0; t < 4; ++t) wnxt[t] = *reinterpret_cast<const vec2f*>(wp + (size_t)(4 * (m + 1) + t) * 128);
        }
        const float* tq = tile + 4 * min(max(g0 + m, 0), kFusedW / 4 - 1);
#pragma unroll
        for (int i = 0; i < kFusedHalf; ++i) {
          const vec4f v4 = *reinterpret_cast<const vec4f*>(tq + i * kFusedW);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            aA[i] = fmaf(wcur[t][0], v4[t], aA[i]);
            aB[i] = fmaf(wcur[t][1], v4[t], aB[i]);
          }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) wcur[t] = wnxt[t];
        // (the sums are formed HERE: left alone, the optimiser sinks every multiply-add to the stores below and keeps all
        //  56 tile reads -- 224 registers -- alive until then)
#pragma unroll
        for (int i = 0; i < kFusedHalf; ++i) asm volatile("" : "+v"(aA[i]), "+v"(aB[i]));
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < kFusedHalf; ++i) {
        const int r = hh * kFusedHalf + i;
        if (r < no) {
          float* dst = out + ((size_t)pl * oy.n_out + o0 + r) * ox.n_out + o;
          if (o + 1 < oe) {
            if ((ox.n_out & 1) == 0) {
              vec2f st; st[0] = aA[i]; st[1] = aB[i];
              *reinterpret_cast<vec2f*>(dst) = st;   // (even row pitch and even o: 8-byte aligned)
            } else {
              dst[0] = aA[i]; dst[1] = aB[i];
            }
            olo = fminf(olo, fminf(aA[i], aB[i]));
            ohi = fmaxf(ohi, fmaxf(aA[i], aB[i]));
          } else if (o < oe) {
            dst[0] = aA[i];
            olo = fminf(olo, aA[i]);
            ohi = fmaxf(ohi, aA[i]);
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();  // everyone has read this half before the next one overwrites it
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
    // relative column c = 4 * lane + v goes to plane (c & 1), slot (c >> 1): evens at [0, 128), odds at [128, 256)
#pragma unroll
    for (int i = 0; i < kFusedHalf; ++i) {
      vec2f ev, od;
      ev[0] = acc[hh * kFusedHalf + i][0]; ev[1] = acc[hh * kFusedHalf + i][2];
      od[0] = acc[hh * kFusedHalf + i][1]; od[1] = acc[hh * kFusedHalf + i][3];
      *reinterpret_cast<vec2f*>(&tile[i * kFusedW + 2 * lane]) = ev;
      *reinterpret_cast<vec2f*>(&tile[i * kFusedW + kFusedW / 2 + 2 * lane]) = od;
    }
    // the tile is this wave's own and a wave's LDS operations execute in order: only the compiler has to be told
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int slot = 0; slot < 2; ++slot) {
      const int o = ob + lane + 64 * slot;
      const bool valid = o < oe;
      const int s = valid ? ox.start[o] - c0 : 0;  // s + K <= 256 for every output of the block (host-checked)
      float a8[kFusedHalf];
#pragma unroll
      for (int i = 0; i < kFusedHalf; ++i) a8[i] = 0.f;
      // The weights are per-lane global loads (L1 / L2 hits, but a round trip each): 16 at a time, all in flight together.
      // (Staging the block's weights in LDS once per workgroup was built and measured too: the extra 12.8 KB cost an
      //  occupancy step and the pass got slower, 2.2 ms against 1.95 ms for the 1024^2 level.)
      const float* wcol = ox.wT + (valid ? o : 0);
      for (int k0 = 0; k0 < ox.K; k0 += 16) {
        float wk[16];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) wk[kk] = (valid && k0 + kk < ox.K) ? wcol[(size_t)(k0 + kk) * ox.n_out] : 0.0f;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
          if (k0 + kk < ox.K) {  // wave-uniform
            const int c = s + k0 + kk;
            const float* q = &tile[(c & 1) * (kFusedW / 2) + (c >> 1)];
#pragma unroll
            for (int i = 0; i < kFusedHalf; ++i) a8[i] = fmaf(wk[kk], q[i * kFusedW], a8[i]);
          }
        }
      }
      if (valid) {
#pragma unroll
        for (int i = 0; i < kFusedHalf; ++i) {
          const int r = hh * kFusedHalf + i;
          if (r < no) {
            out[((size_t)pl * oy.n_out + o0 + r) * ox.n_out + o] = a8[i];
            olo = fminf(olo, a8[i]);
            ohi = fmaxf(ohi, a8[i]);
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // everyone has read this half before the next one overwrites it
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = fminf(lo, __shfl_down(lo, off, 64));
    hi = fmaxf(hi, __shfl_down(hi, off, 64));
    olo = fminf(olo, __shfl_down(olo, off, 64));
    ohi = fmaxf(ohi, __shfl_down(ohi, off, 64));
  }
  if (lane == 0) {
    const int ngroups = (oy.n_out + kOutsV - 1) / kOutsV;
    const size_t slot = ((size_t)pl * ngroups + g) * gridDim.x + blockIdx.x;  // [plane][group][column block]
    if (mm_part) { mm_part[slot * 2] = lo; mm_part[slot * 2 + 1] = hi; }
    out_part[slot * 2] = olo;
    out_part[slot * 2 + 1] = ohi;
  }
}

// Clip the (wave) tiles of a fused pyramid step whose output range leaves the pair's input range (see above): one wave
// per tile, same grid as the step itself.  (One THREAD per tile's range check with the flagged tiles clipped by the wave
// one after the other was slower, 0.48 ms against 0.17 ms on the benchmark's 1024^2 level: images with saturated
// regions flag many tiles, and those want to be clipped in parallel.)
__global__ __launch_bounds__(kBlock) void ica_resample_clip_kernel(float* __restrict__ out, int channels, int n_out_y, int n_out_x,
                                                                   int nb, const float* __restrict__ out_part,
                                                                   const float* __restrict__ minmax) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = blockIdx.y * kWaves + wave, o0 = g * kOutsV, pl = blockIdx.z;
  if (o0 >= n_out_y) return;
  const int ngroups = (n_out_y + kOutsV - 1) / kOutsV;
  const size_t slot = ((size_t)pl * ngroups + g) * gridDim.x + blockIdx.x;
  const int pair = pl / channels;
  const float lo = minmax[pair * 2], hi = minmax[pair * 2 + 1];
  if (out_part[slot * 2] >= lo && out_part[slot * 2 + 1] <= hi) return;
  const int no = min(kOutsV, n_out_y - o0);
  const int ob = blockIdx.x * nb, oe = min(ob + nb, n_out_x);
  for (int r = 0; r < no; ++r)
    for (int o = ob + lane; o < oe; o += 64) {
      float* q = out + ((size_t)pl * n_out_y + o0 + r) * n_out_x + o;
      *q = fminf(fmaxf(*q, lo), hi);
    }
}

// Generic banded pass for downsampling factors the tiled passes are not sized for (a window of 16 outputs wider than
// kVWin input rows, or a 64-output row tile that does not fit the LDS): one thread per output element, taps straight
// from memory.  Slow and only reachable with nu below ~0.12, which the reference accepts like any other value.
// (The vertical pass reads the level's own storage, float32 or uint8.)
// VERTICAL: in [planes][n_in][W] -> out [planes][n_out][W].  Otherwise: in [planes][H][n_in] -> out [planes][H][n_out],
// clipped to the pair's range (the horizontal pass is the last step of a pyramid level).
template <bool VERTICAL, class TIn = float>
__global__ __launch_bounds__(kBlock) void ica_resample_generic_kernel(const TIn* __restrict__ in, float* __restrict__ out,
                                                                      int channels, int other, BandOp op,
                                                                      const float* __restrict__ minmax) {
  const int pl = blockIdx.z;
  const int x = blockIdx.x * kBlock + threadIdx.x;  // VERTICAL: column; else: output column
  const int y = blockIdx.y;                         // VERTICAL: output row; else: row
  if (VERTICAL) {
    if (x >= other) return;
    const int j0 = op.start[y];
    const TIn* q = in + ((size_t)pl * op.n_in + j0) * other + x;
    float acc = 0.f;
    for (int k = 0; k < op.K && j0 + k < op.n_in; ++k) acc = fmaf(op.wT[(size_t)k * op.n_out + y], (float)q[(size_t)k * other], acc);
    out[((size_t)pl * op.n_out + y) * other + x] = acc;
  } else {
    if (x >= op.n_out) return;
    const int j0 = op.start[x];
    const TIn* q = in + ((size_t)pl * other + y) * op.n_in + j0;
    float acc = 0.f;
    for (int k = 0; k < op.K && j0 + k < op.n_in; ++k) acc = fmaf(op.wT[(size_t)k * op.n_out + x], (float)q[k], acc);
    const int pair = pl / channels;
    out[((size_t)pl * other + y) * op.n_out + x] = fminf(fmaxf(acc, minmax[pair * 2]), minmax[pair * 2 + 1]);
  }
}

}  // namespace ica
