// C ABI of libica_hip.so (include/ica_hip.h): context, workspaces, the pyramid / level / iteration
// driver and the per-operator hooks.  Everything here is host code that enqueues the kernels of
// ica_kernels.hpp on the context's stream; there is no CPU compute path.
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <thread>
#include <sched.h>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "../../include/ica_hip.h"
#include "ica_kernels.hpp"
#include "ica_refops.hpp"
#include "ica_iterate_launch.hpp"
#include "ica_pyramid.hpp"

using namespace ica;

namespace {

struct DevBuf {
  void* ptr = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
    hipError_t e = hipMalloc(&ptr, bytes);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
  }
  template <class T> T* as() const { return reinterpret_cast<T*>(ptr); }
};

struct DeviceBandOp {
  DevBuf start, wT, vj0, vspan, vown0, vown1, vw, vwslot, hc0, wpair, hoff, sw;
  int vfast_slot = -1;     // window slot of the groups made of shift-invariant outputs only (BandOp::vfast_slot)
  bool pair_pass = false;  // the adjacent-output horizontal pass applies (ica_kernels.hpp, FusedCols)
  int n_in = 0, n_out = 0, K = 0;
  int hnb = 0, hblocks = 0;  // fused step, this operator on the x axis: outputs per column block (0: not usable), blocks
  int vspan_max = 0;   // widest input span of kOutsV consecutive outputs (window of the vertical pass)
  int span_pad = 0;    // LDS row pitch (floats) of the horizontal pass: widest 64-output input span, made odd
  bool vcover = false; // the groups' "own" ranges cover the whole input axis (min/max can be folded into the pass)
  BandOp view() const {
    return BandOp{start.as<int>(), wT.as<float>(), vj0.as<int>(), vspan.as<int>(), vown0.as<int>(), vown1.as<int>(),
                  vw.as<float>(), vwslot.as<int>(), n_in, n_out, K, sw.as<float>(), vfast_slot};
  }
  FusedCols cols() const {
    return FusedCols{hc0.as<int>(), hnb, pair_pass ? wpair.as<float>() : nullptr, pair_pass ? hoff.as<int>() : nullptr};
  }
  void release() { start.release(); wT.release(); vj0.release(); vspan.release(); vown0.release(); vown1.release(); vw.release(); vwslot.release(); hc0.release(); wpair.release(); hoff.release(); sw.release(); }
};

struct LevelGeom {
  int W = 0, H = 0;
  int tiles_x = 0, tiles_y = 0, tile_h = 0, tiles = 0;
};

}  // namespace

struct ica_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t handover = nullptr;   // orders a newly bound stream behind the work queued on the previous one
  bool profiling = false;
  // switches read once, when the context is created; per context, not per process
  int accum_variant = 2;           // ICA_ACCUM=1: the direct-gather iteration kernel (in-library cross-check)
  bool fold_hessian = true;        // ICA_SEPARATE_HESSIAN=1: separate L2 Hessian pass per level
  int max_tile_h = 512;            // ICA_MAX_TILE_H
  bool fused_pyramid = true;       // ICA_PYRAMID_2PASS=1: vertical and horizontal pyramid passes as two kernels
  double board_timeout_s = 10.0;   // ICA_BOARD_TIMEOUT_MS: how long the host waits for an iteration's count
  std::string last_error;
  // workspaces
  DevBuf pyr1[ICA_MAX_SCALES], pyr2[ICA_MAX_SCALES];  // levels >= 1
  DevBuf tmp_resample;                                // intermediate of the separable pyramid pass
  DevBuf partials, hinv, p_prev, lambda_it, error, niter, lists, counts, count_hist, minmax, mm_part, mm_in, out_part, clock_probe, clock_acc;
  DevBuf scratch_p, ref_part;
  int* pinned_count = nullptr;   // one list's shard counters for synchronous reads + a [scales][max_iter] board the kernels post to
  int* board = nullptr;          // host pointer of the board (pinned, host-coherent)
  int* board_dev = nullptr;      // device pointer of the same memory
  size_t board_cap = 0;
  bool board_dirty = false;      // kernels of an earlier call may still post to the board
  std::map<std::pair<int, int>, DeviceBandOp*> ops;
  std::vector<hipEvent_t> ev_pool;
  ~ica_ctx() {
    for (auto& kv : ops) {
      kv.second->release();
      delete kv.second;
    }
    for (auto e : ev_pool) (void)hipEventDestroy(e);
    if (handover) (void)hipEventDestroy(handover);
    for (int i = 0; i < ICA_MAX_SCALES; ++i) { pyr1[i].release(); pyr2[i].release(); }
    DevBuf* all[] = {&tmp_resample, &partials, &hinv, &p_prev, &lambda_it, &error, &niter, &lists, &counts,
                     &count_hist, &minmax, &mm_part, &mm_in, &out_part, &scratch_p, &ref_part, &clock_probe, &clock_acc};
    for (auto* b : all) b->release();
    if (pinned_count) (void)hipHostFree(pinned_count);
    if (board) (void)hipHostFree(board);
  }
};

namespace {

#define ICA_HIP(ctx, expr)                                                                 \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);               \
      return _e == hipErrorOutOfMemory ? ICA_ENOMEM : ICA_EHIP;                            \
    }                                                                                      \
  } while (0)

#define ICA_TRY(expr)              \
  do {                             \
    int _s = (expr);               \
    if (_s != ICA_OK) return _s;   \
  } while (0)

int validate(const ica_problem* pr) {
  if (!pr) return ICA_EINVAL_ARG;
  if (pr->channels < 1 || pr->channels > ICA_MAX_CHANNELS) return ICA_EINVAL_SHAPE;
  if (pr->batch <= 0 || pr->height <= 0 || pr->width <= 0) return ICA_EINVAL_ARG;
  if (pr->height >= (1 << 20) || pr->width >= (1 << 20)) return ICA_EINVAL_ARG;  // 20-bit fields in the chunk plan
  // the iteration kernel addresses one pair's planes through 32-bit buffer offsets (num_records and byte offsets are int)
  if ((size_t)pr->channels * (size_t)pr->height * (size_t)pr->width * 4 >= ((size_t)1 << 31)) return ICA_EINVAL_ARG;
  if (!(pr->tol < 0.01)) return ICA_EINVAL_TOL;  // ref :59-60
  if (nparams_of(pr->transform_type) < 0) return ICA_EINVAL_TRANSFORM;
  if (pr->robust_type < ICA_QUADRATIC || pr->robust_type > ICA_CHARBONNIER) return ICA_EINVAL_ROBUST;
  if (pr->nscales < 1 || pr->nscales > ICA_MAX_SCALES) return ICA_EINVAL_ARG;
  if (pr->max_iter < 0 || pr->max_iter > 100000) return ICA_EINVAL_ARG;
  if (pr->storage != ICA_STORAGE_F32 && pr->storage != ICA_STORAGE_U8) return ICA_EINVAL_ARG;
  // uint8 images: gray or RGB (the LDS-staged kernel); the 2- and 4-channel direct-gather kernel is float32 only
  if (pr->storage == ICA_STORAGE_U8 && pr->channels != 1 && pr->channels != 3) return ICA_EINVAL_SHAPE;
  return ICA_OK;
}

// Tile height for a launch over `pairs` pairs: the tallest tile (taller tiles amortise the per-tile fold) that still
// gives the launch >= ~4 workgroups per CU.  Late iterations, when only a few pairs are still iterating, therefore
// use short tiles and spread those pairs over the whole chip instead of running 16 long workgroups per pair.
#ifndef ICA_MIN_TH
#define ICA_MIN_TH 16
#endif
#ifndef ICA_MIN_BLOCKS   // a launch is cut into shorter tiles until it has this many workgroups (1536 are resident)
#define ICA_MIN_BLOCKS 2048
#endif
// `chunk` = rows per chunk of the kernel that will run the level (iter_chunk_rows): tile heights are whole chunks, so
// that only an image's last tile ends in a partial chunk (a partial chunk takes the careful band).
void set_tiling(LevelGeom& g, int pairs, int max_tile_h, int chunk) {
  g.tiles_x = (g.W + kTileW - 1) / kTileW;
  // (at most 32 chunks per tile: the chunk plans of a tile live in one register)
  auto whole = [&](int th) { return th >= chunk ? (th / chunk) * chunk : th; };
  int th = whole(std::max(ICA_MIN_TH, std::min(std::min(512, 32 * chunk), max_tile_h)));
  while (th > ICA_MIN_TH) {
    const long blocks = (long)pairs * g.tiles_x * ((g.H + th - 1) / th);
    if (blocks >= ICA_MIN_BLOCKS && th <= std::max(ICA_MIN_TH, g.H)) break;
    th = std::max(ICA_MIN_TH, whole(th >> 1));
  }
  g.tile_h = th;
  g.tiles_y = (g.H + th - 1) / th;
  g.tiles = g.tiles_x * g.tiles_y;
}

// rows per chunk of the iteration kernel of level `s` of a problem (uint8 storage applies to level 0 only)
int level_chunk(const ica_problem* pr, int s) {
  return iter_chunk_rows(pr->channels, pr->storage == ICA_STORAGE_U8 && s == 0);
}

void level_sizes(const ica_ctx* ctx, const ica_problem* pr, LevelGeom* g) {
  g[0].W = pr->width;
  g[0].H = pr->height;
  for (int s = 1; s < pr->nscales; ++s) {  // zoom.py:8-22
    g[s].W = std::max(1, round_half_even(g[s - 1].W * pr->nu));
    g[s].H = std::max(1, round_half_even(g[s - 1].H * pr->nu));
  }
  for (int s = 0; s < pr->nscales; ++s) set_tiling(g[s], pr->batch, ctx->max_tile_h, level_chunk(pr, s));
}

int get_op(ica_ctx* ctx, int n_in, int n_out, DeviceBandOp** out) {
  auto key = std::make_pair(n_in, n_out);
  auto it = ctx->ops.find(key);
  if (it != ctx->ops.end()) {
    *out = it->second;
    return ICA_OK;
  }
  HostBandOp h = build_rescale_op(n_in, n_out);
  // decimation by exactly 2: the interior outputs' weights made bit-identical (ica_pyramid.hpp)
  int inv_first = 0, inv_end = 0;   // outputs [inv_first, inv_end)
  std::vector<float> sw(kFastK, 0.0f);
  if (n_out >= 4 * kOutsV) canonicalise_interior(h, kFastK, inv_first, inv_end, sw);
  std::vector<float> wT((size_t)h.K * n_out);
  for (int o = 0; o < n_out; ++o)
    for (int k = 0; k < h.K; ++k) wT[(size_t)k * n_out + o] = (float)h.w[(size_t)o * h.K + k];
  auto* d = new DeviceBandOp();
  d->n_in = n_in;
  d->n_out = n_out;
  d->K = h.K;
  // vertical pass: per group of kOutsV outputs, a zero-padded window of weights indexed by input row
  const int groups = (n_out + kOutsV - 1) / kOutsV;
  std::vector<int> vj0(groups), vspan(groups);
  std::vector<float> vw;           // distinct windows, [slot][kVWin][kOutsV]
  std::vector<int> vwslot(groups, 0);
  std::vector<float> win((size_t)kVWin * kOutsV);
  for (int g = 0; g < groups; ++g) {
    const int o0 = g * kOutsV, no = std::min(kOutsV, n_out - o0);
    vj0[g] = h.start[o0];
    vspan[g] = std::min(n_in, h.start[o0 + no - 1] + h.K) - vj0[g];
    d->vspan_max = std::max(d->vspan_max, vspan[g]);
    if (vspan[g] > kVWin) continue;  // rejected by the caller
    std::fill(win.begin(), win.end(), 0.0f);
    for (int i = 0; i < no; ++i)
      for (int k = 0; k < h.K; ++k) {
        const int jj = h.start[o0 + i] + k - vj0[g];
        if (jj >= 0 && jj < vspan[g]) win[(size_t)jj * kOutsV + i] = (float)h.w[(size_t)(o0 + i) * h.K + k];
      }
    // same window as an earlier group (to the last bit or two: the entries come out of a float64 recursion)?
    int slot = -1;
    const int nslots = (int)(vw.size() / win.size());
    for (int t = nslots - 1; t >= 0 && t >= nslots - 4 && slot < 0; --t) {
      float dmax = 0.f;
      for (size_t e = 0; e < win.size(); ++e) dmax = std::max(dmax, std::fabs(vw[(size_t)t * win.size() + e] - win[e]));
      if (dmax <= 1.2e-7f) slot = t;
    }
    if (slot < 0) {
      slot = nslots;
      vw.insert(vw.end(), win.begin(), win.end());
    }
    vwslot[g] = slot;
  }
  if (vw.empty()) vw.assign(win.size(), 0.0f);
  {  // the window slot of a group that lies inside the shift-invariant range (all such groups share one slot)
    const int gf = (inv_first + kOutsV - 1) / kOutsV, ge = inv_end / kOutsV;
    if (ge > gf) {
      d->vfast_slot = vwslot[gf];
      for (int g = gf; g < ge; ++g)
        if (vwslot[g] != d->vfast_slot) d->vfast_slot = -1;
      // a border group must not share that slot (its outputs are not all shift-invariant)
      for (int g = 0; g < groups && d->vfast_slot >= 0; ++g)
        if ((g < gf || g >= ge) && vwslot[g] == d->vfast_slot) d->vfast_slot = -1;
    }
  }
  if (getenv("ICA_DEBUG_OPS"))
    fprintf(stderr, "ica: rescale operator %d -> %d: K %d, %d groups of %d outputs, %d distinct weight windows\n", n_in, n_out,
            h.K, groups, kOutsV, (int)(vw.size() / win.size()));
  // partition of the input axis among the groups (for the fused min/max): cut at the middle of each overlap
  std::vector<int> vown0(groups), vown1(groups);
  d->vcover = vj0[0] == 0 && vj0[groups - 1] + vspan[groups - 1] == n_in;
  for (int g = 0; g < groups; ++g) {
    vown0[g] = g == 0 ? vj0[0] : (vj0[g] + vj0[g - 1] + vspan[g - 1] + 1) / 2;
    vown1[g] = g == groups - 1 ? vj0[g] + vspan[g] : (vj0[g + 1] + vj0[g] + vspan[g] + 1) / 2;
    if (g > 0 && vj0[g] > vj0[g - 1] + vspan[g - 1]) d->vcover = false;  // a gap between windows
  }
  for (int g = 0; g < groups; ++g)
    if (vown0[g] < vj0[g] || vown1[g] > vj0[g] + vspan[g] || vown1[g] < vown0[g]) d->vcover = false;
  {
    int span = 1;
    for (int o = 0; o < n_out; o += 64) span = std::max(span, h.start[std::min(o + 63, n_out - 1)] + h.K - h.start[o]);
    d->span_pad = span | 1;
  }
  // fused step (ica_resample_fused_kernel): column blocks of `hnb` outputs whose input span, from a 16-byte aligned
  // first column, fits the kFusedW columns a wave holds; starts are non-decreasing, so the last output decides
  std::vector<int> hc0;
  {
    auto fits = [&](int nb) {
      for (int ob = 0; ob < n_out; ob += nb)
        if (h.start[std::min(ob + nb, n_out) - 1] + h.K - (h.start[ob] & ~3) > kFusedW) return false;
      return true;
    };
    int nb = 128;
    while (nb >= 16 && !fits(nb)) nb -= 8;
    if (nb >= 16) {
      d->hnb = nb;
      d->hblocks = (n_out + nb - 1) / nb;
      for (int ob = 0; ob < n_out; ob += nb) hc0.push_back(h.start[ob] & ~3);
    }
  }
  // adjacent-output horizontal pass (ica_kernels.hpp, FusedCols): per-lane weights on a regular window, if the operator has one
  std::vector<float> wpair;
  std::vector<int> hoff;
  if (d->hnb > 0 && !getenv("ICA_NO_PAIR_PASS")) d->pair_pass = build_pair_pass(h, d->hnb, hc0, kPairWin, wpair, hoff);
  auto up = [&](DevBuf& b, const void* src, size_t bytes) -> hipError_t {
    hipError_t e = b.ensure(bytes);
    if (e == hipSuccess) e = hipMemcpy(b.ptr, src, bytes, hipMemcpyHostToDevice);
    return e;
  };
  hipError_t e = up(d->start, h.start.data(), h.start.size() * sizeof(int));
  if (e == hipSuccess) e = up(d->sw, sw.data(), sw.size() * sizeof(float));
  if (e == hipSuccess) e = up(d->wT, wT.data(), wT.size() * sizeof(float));
  if (e == hipSuccess) e = up(d->vj0, vj0.data(), vj0.size() * sizeof(int));
  if (e == hipSuccess) e = up(d->vspan, vspan.data(), vspan.size() * sizeof(int));
  if (e == hipSuccess) e = up(d->vown0, vown0.data(), vown0.size() * sizeof(int));
  if (e == hipSuccess) e = up(d->vown1, vown1.data(), vown1.size() * sizeof(int));
  if (e == hipSuccess) e = up(d->vw, vw.data(), vw.size() * sizeof(float));
  if (e == hipSuccess) e = up(d->vwslot, vwslot.data(), vwslot.size() * sizeof(int));
  if (e == hipSuccess && !hc0.empty()) e = up(d->hc0, hc0.data(), hc0.size() * sizeof(int));
  if (e == hipSuccess && d->pair_pass) e = up(d->wpair, wpair.data(), wpair.size() * sizeof(float));
  if (e == hipSuccess && d->pair_pass) e = up(d->hoff, hoff.data(), hoff.size() * sizeof(int));
  if (e != hipSuccess) {
    ctx->last_error = std::string("pyramid operator upload: ") + hipGetErrorString(e);
    d->release();
    delete d;
    return e == hipErrorOutOfMemory ? ICA_ENOMEM : ICA_EHIP;
  }
  ctx->ops[key] = d;
  *out = d;
  return ICA_OK;
}

template <class TIn>
int launch_minmax(ica_ctx* ctx, int batch, size_t elems, const TIn* img, float* out) {
  int chunks = (int)std::min<size_t>(64, (elems + (size_t)kBlock * 16 - 1) / ((size_t)kBlock * 16));
  chunks = std::max(chunks, 1);
  ICA_HIP(ctx, ctx->mm_part.ensure((size_t)batch * chunks * 2 * sizeof(float)));
  hipLaunchKernelGGL(ica_minmax_partial_kernel<TIn>, dim3(chunks, batch), dim3(kBlock), 0, ctx->stream, img, elems, chunks,
                     ctx->mm_part.as<float>());
  hipLaunchKernelGGL(ica_minmax_final_kernel, dim3((batch + kWaves - 1) / kWaves), dim3(kBlock), 0, ctx->stream,
                     ctx->mm_part.as<float>(), chunks, batch, out);
  ICA_HIP(ctx, hipGetLastError());
  return ICA_OK;
}

// One pyramid step.  `mm_out` ([batch][2]) receives the min/max of `in` (the clip range of skimage's rescale, and --
// for I2 -- of the warp at that level); it is folded into the vertical pass, which reads every pixel of `in` anyway.
// TIn: element type of `in` (uint8 for the level-0 images of an ICA_STORAGE_U8 problem); `out` is always float32.
// `mm_next` (optional): the min/max of `out`, when the step can give it for free (single-kernel step: from the tiles'
// ranges); *have_next says whether it was written.
template <class TIn>
int launch_pyramid_down(ica_ctx* ctx, int batch, int C, int H, int W, int h, int w, const TIn* in, float* out,
                        float* mm_out, float* mm_next = nullptr, bool* have_next = nullptr) {
  if (have_next) *have_next = false;
  DeviceBandOp *ox = nullptr, *oy = nullptr;
  ICA_TRY(get_op(ctx, W, w, &ox));
  ICA_TRY(get_op(ctx, H, h, &oy));
  const int planes = batch * C;
  const size_t lds = (size_t)kRowsH * ox->span_pad * sizeof(float);
  const bool generic_v = oy->vspan_max > kVWin, generic_h = lds > 64 * 1024;
  // 4 columns per lane (one 16-byte load, or one dword of uint8 pixels) in the vertical pass when rows and base
  // pointers are aligned for it, else one column per lane
  const bool vec4 = W % 4 == 0 && ((uintptr_t)in % (4 * sizeof(TIn))) == 0;
  constexpr bool kF32 = std::is_same<TIn, float>::value;
  if (ctx->fused_pyramid && vec4 && !generic_v && ox->hnb > 0) {
    // both passes in one kernel (ica_resample_fused_kernel): no intermediate image
    const int vgroups = (h + kOutsV - 1) / kOutsV, xb = ox->hblocks;
    const int per_pair = C * vgroups * xb;
    float* part = nullptr;
    if (oy->vcover) {
      ICA_HIP(ctx, ctx->mm_part.ensure((size_t)batch * per_pair * 2 * sizeof(float)));
      part = ctx->mm_part.as<float>();
    } else {
      ICA_TRY(launch_minmax(ctx, batch, (size_t)C * H * W, in, mm_out));
    }
    ICA_HIP(ctx, ctx->out_part.ensure((size_t)batch * per_pair * 2 * sizeof(float)));
    float* opart = ctx->out_part.as<float>();
    const int step = std::max(C, (32768 / C) * C);  // planes ride on grid.z: chunk on whole pairs
    for (int p0 = 0; p0 < planes; p0 += step) {
      const int np = std::min(step, planes - p0);
      const size_t po = (size_t)p0 * vgroups * xb * 2;
      if (ox->pair_pass)
        hipLaunchKernelGGL((ica_resample_fused_kernel<TIn, true>), dim3(xb, (vgroups + kWaves - 1) / kWaves, np), dim3(kBlock), 0,
                           ctx->stream, in + (size_t)p0 * H * W, out + (size_t)p0 * h * w, W, oy->view(), ox->view(),
                           ox->cols(), part ? part + po : nullptr, opart + po);
      else
        hipLaunchKernelGGL((ica_resample_fused_kernel<TIn, false>), dim3(xb, (vgroups + kWaves - 1) / kWaves, np), dim3(kBlock), 0,
                           ctx->stream, in + (size_t)p0 * H * W, out + (size_t)p0 * h * w, W, oy->view(), ox->view(),
                           ox->cols(), part ? part + po : nullptr, opart + po);
    }
    if (part)
      hipLaunchKernelGGL(ica_minmax_final_kernel, dim3((batch + kWaves - 1) / kWaves), dim3(kBlock), 0, ctx->stream, part,
                         per_pair, batch, mm_out);
    for (int p0 = 0; p0 < planes; p0 += step) {
      const int np = std::min(step, planes - p0);
      hipLaunchKernelGGL(ica_resample_clip_kernel, dim3(xb, (vgroups + kWaves - 1) / kWaves, np), dim3(kBlock), 0, ctx->stream,
                         out + (size_t)p0 * h * w, C, h, w, ox->hnb, opart + (size_t)p0 * vgroups * xb * 2,
                         mm_out + (size_t)(p0 / C) * 2);
    }
    if (mm_next) {
      hipLaunchKernelGGL(ica_minmax_clipped_kernel, dim3((batch + kWaves - 1) / kWaves), dim3(kBlock), 0, ctx->stream, opart,
                         per_pair, batch, mm_out, mm_next);
      if (have_next) *have_next = true;
    }
    ICA_HIP(ctx, hipGetLastError());
    return ICA_OK;
  }
  ICA_HIP(ctx, ctx->tmp_resample.ensure((size_t)planes * h * W * sizeof(float)));
  float* tmp = ctx->tmp_resample.as<float>();
  const int vgroups = (h + kOutsV - 1) / kOutsV, xblocks = vec4 ? (W + 255) / 256 : (W + 63) / 64;
  const int per_pair = C * vgroups * xblocks;  // min/max partials per pair written by the vertical pass
  float* part = nullptr;
  if (oy->vcover && !generic_v) {
    ICA_HIP(ctx, ctx->mm_part.ensure((size_t)batch * per_pair * 2 * sizeof(float)));
    part = ctx->mm_part.as<float>();
  } else {
    ICA_TRY(launch_minmax(ctx, batch, (size_t)C * H * W, in, mm_out));
  }
  // planes ride on grid.z (<= 65535): chunk on whole pairs so the per-pair indices stay aligned
  const int step = std::max(C, (32768 / C) * C);
  for (int p0 = 0; p0 < planes; p0 += step) {
    const int np = std::min(step, planes - p0);
    const dim3 gv(xblocks, (vgroups + kWaves - 1) / kWaves, np);
    float* pp = part ? part + (size_t)p0 * vgroups * xblocks * 2 : nullptr;
    if (generic_v) {  // very small nu: window of 16 outputs wider than the tiled pass is sized for
      hipLaunchKernelGGL((ica_resample_generic_kernel<true, TIn>), dim3((W + kBlock - 1) / kBlock, h, np), dim3(kBlock), 0,
                         ctx->stream, in + (size_t)p0 * H * W, tmp + (size_t)p0 * h * W, C, W, oy->view(),
                         (const float*)nullptr);
    } else if (vec4)
      hipLaunchKernelGGL((ica_resample_v_kernel<4, TIn>), gv, dim3(kBlock), 0, ctx->stream, in + (size_t)p0 * H * W,
                         tmp + (size_t)p0 * h * W, np, C, W, oy->view(), pp);
    else
      hipLaunchKernelGGL((ica_resample_v_kernel<1, TIn>), gv, dim3(kBlock), 0, ctx->stream, in + (size_t)p0 * H * W,
                         tmp + (size_t)p0 * h * W, np, C, W, oy->view(), pp);
  }
  if (part)
    hipLaunchKernelGGL(ica_minmax_final_kernel, dim3((batch + kWaves - 1) / kWaves), dim3(kBlock), 0, ctx->stream, part, per_pair, batch,
                       mm_out);
  for (int p0 = 0; p0 < planes; p0 += step) {
    const int np = std::min(step, planes - p0);
    const dim3 gh((w + 63) / 64, (h + kRowsH - 1) / kRowsH, np);
    if (generic_h)
      hipLaunchKernelGGL((ica_resample_generic_kernel<false, float>), dim3((w + kBlock - 1) / kBlock, h, np), dim3(kBlock), 0,
                         ctx->stream, tmp + (size_t)p0 * h * W, out + (size_t)p0 * h * w, C, h, ox->view(),
                         mm_out + (size_t)(p0 / C) * 2);
    else if (W % 2 == 0)  // rows of the intermediate image are 8-byte aligned: stage with 8-byte loads
      hipLaunchKernelGGL(ica_resample_h_kernel<true>, gh, dim3(kBlock), lds, ctx->stream, tmp + (size_t)p0 * h * W,
                         out + (size_t)p0 * h * w, np, C, h, ox->view(), mm_out + (size_t)(p0 / C) * 2, ox->span_pad);
    else
      hipLaunchKernelGGL(ica_resample_h_kernel<false>, gh, dim3(kBlock), lds, ctx->stream, tmp + (size_t)p0 * h * W,
                         out + (size_t)p0 * h * w, np, C, h, ox->view(), mm_out + (size_t)(p0 / C) * 2, ox->span_pad);
  }
  ICA_HIP(ctx, hipGetLastError());
  return ICA_OK;
}

// variant 1 (ICA_ACCUM=1) is the simple direct-gather kernel, kept as an in-library cross-check of the LDS-staged one
template <int TT, int MODE>
bool launch_accumulate_c(int variant, int C, dim3 grid, hipStream_t st, const LevelArgs& a) {
  // 1 and 3 channels (gray / RGB, the reference's data) run the LDS-staged kernel; 2 and 4 channels (the reference's
  // robust entry point takes any channel count, ref :168-177) run the direct-gather kernel
  if ((variant == 1 && !a.u8) || C == 2 || C == 4) {  // (the cross-check kernel reads float32 images only)
    switch (C) {
      case 1: hipLaunchKernelGGL((ica_accumulate_kernel<TT, MODE, 1>), grid, dim3(kBlock), 0, st, a); break;
      case 2: hipLaunchKernelGGL((ica_accumulate_kernel<TT, MODE, 2>), grid, dim3(kBlock), 0, st, a); break;
      case 3: hipLaunchKernelGGL((ica_accumulate_kernel<TT, MODE, 3>), grid, dim3(kBlock), 0, st, a); break;
      default: hipLaunchKernelGGL((ica_accumulate_kernel<TT, MODE, 4>), grid, dim3(kBlock), 0, st, a); break;
    }
    return true;
  } else {
    // 1-D grid: (pairs rounded up to a multiple of 8) x tiles, decoded XCD-aware inside the kernel; launches of fewer
    // than kFlatMapBelow pairs (a.n_pairs = pairs of the launch) use a flat pairs x tiles grid instead
    const unsigned blocks = (unsigned)((int)grid.y < kFlatMapBelow ? grid.y * grid.x : ((grid.y + 7) / 8) * 8 * grid.x);
    LevelArgs al = a;
    al.n_pairs = (int)grid.y;  // the kernel picks the same map from it
    return launch_iterate(TT, MODE, C, a.robust_type, blocks, st, al);
  }
}

template <int MODE>
bool launch_accumulate(int variant, int tt, int C, dim3 grid, hipStream_t st, const LevelArgs& a) {
  switch (tt) {
    case ICA_TRANSLATION: return launch_accumulate_c<ICA_TRANSLATION, MODE>(variant, C, grid, st, a);
    case ICA_EUCLIDEAN: return launch_accumulate_c<ICA_EUCLIDEAN, MODE>(variant, C, grid, st, a);
    case ICA_SIMILARITY: return launch_accumulate_c<ICA_SIMILARITY, MODE>(variant, C, grid, st, a);
    case ICA_AFFINITY: return launch_accumulate_c<ICA_AFFINITY, MODE>(variant, C, grid, st, a);
    default: return launch_accumulate_c<ICA_HOMOGRAPHY, MODE>(variant, C, grid, st, a);
  }
}

int launch_accumulate_mode(ica_ctx* ctx, int mode, int tt, int C, dim3 grid, hipStream_t st, const LevelArgs& a) {
  const int v = ctx->accum_variant;
  bool ok;
  if (mode == kModeL2) ok = launch_accumulate<kModeL2>(v, tt, C, grid, st, a);
  else if (mode == kModeRobust) ok = launch_accumulate<kModeRobust>(v, tt, C, grid, st, a);
  // (uint8 levels always run the LDS-staged kernel -- the direct-gather kernel reads float32 only -- so with ICA_ACCUM=1
  //  their first L2 iteration takes the staged kernel's own first-iteration mode too)
  else if (mode == kModeL2First && (v == 2 || a.u8) && (C == 1 || C == 3)) ok = launch_accumulate<kModeL2First>(v, tt, C, grid, st, a);
  else if (mode == kModeL2First) ok = launch_accumulate<kModeRobust>(v, tt, C, grid, st, a);  // v1: weights 1 at run time
  else ok = launch_accumulate<kModeHessian>(v, tt, C, grid, st, a);
  if (!ok) {  // never silently: a missing launch would leave stale partial sums and an unposted board slot behind
    ctx->last_error = "no iteration kernel is instantiated for mode " + std::to_string(mode) + " / robust function " +
                      std::to_string(a.robust_type) + " / " + std::to_string(C) + " channels" + (a.u8 ? " / uint8 storage" : "");
    return ICA_EINVAL_ARG;
  }
  return ICA_OK;
}

// Host side of the launch-sizing board: wait until the kernel of an earlier iteration has posted its
// pair count (>= 0).  The usual wait is a few microseconds (the GPU is one iteration behind the host), so: a short
// spin, then sched_yield, then 50 us sleeps -- a waiting context does not hold a core -- under a wall-clock limit.
// Returns the count, or -1 on timeout.
int wait_board(volatile int* slot, double timeout_s) {
  // The count is posted when the previous iteration's kernel STARTS, i.e. normally long before it is asked for; when
  // it is not there yet, the kernel in front of it is still running (at most a few milliseconds).  Poll for up to 4 ms
  // (yielding the core between bursts of reads), and only then go to sleep between polls: a sleeping waiter wakes
  // 50-100 us late, which showed as five ~40 us holes per solve of configs[1].
  int v = *slot;
  if (v >= 0) return v;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned n = 0;; ++n) {
    for (int i = 0; i < 64 && (v = *slot) < 0; ++i) {}
    if (v >= 0) return v;
    const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (waited > timeout_s) return v;
    if (waited < 4e-3) sched_yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}

template <class TIn>
int launch_warp(ica_ctx* ctx, int batch, int C, int H, int W, int tt, const TIn* I2, const double* p,
                const float* minmax, const TIn* I1, float* Iw, float* DI) {
  for (int b0 = 0; b0 < batch; b0 += 32768) {
    const int nb = std::min(32768, batch - b0);
    const size_t off = (size_t)b0 * C * H * W;
    dim3 grid((W + 63) / 64, (H + kWaves - 1) / kWaves, nb);
#define ICA_WARP_CASE(CC)                                                                                              \
  case CC:                                                                                                             \
    hipLaunchKernelGGL((ica_warp_kernel<CC, TIn>), grid, dim3(kBlock), 0, ctx->stream, I2 + off, I1 ? I1 + off : nullptr, H, W, \
                       tt, p + (size_t)b0 * 8, minmax + (size_t)b0 * 2, Iw ? Iw + off : nullptr,                       \
                       DI ? DI + off : nullptr);                                                                       \
    break;
    switch (C) { ICA_WARP_CASE(1) ICA_WARP_CASE(2) ICA_WARP_CASE(3) ICA_WARP_CASE(4) default: break; }
#undef ICA_WARP_CASE
  }
  ICA_HIP(ctx, hipGetLastError());
  return ICA_OK;
}

int ensure_state(ica_ctx* ctx, const ica_problem* pr, const LevelGeom* g) {
  const size_t B = (size_t)pr->batch;
  size_t max_slots_tiles = 1;  // largest (pairs in a launch) x (tiles per pair) over every launch geometry
  for (int s = 0; s < pr->nscales; ++s)
    for (int nb = 1; nb <= pr->batch; nb = (nb < pr->batch && nb * 2 > pr->batch) ? pr->batch : nb * 2) {
      LevelGeom t = g[s];
      set_tiling(t, nb, ctx->max_tile_h, level_chunk(pr, s));
      // the tiling is chosen for the polled count, which can only over-estimate the pairs of later launches
      max_slots_tiles = std::max(max_slots_tiles, (size_t)std::min<long>(pr->batch, 2L * nb) * t.tiles);
      if (nb == pr->batch) break;
    }
  ICA_HIP(ctx, ctx->partials.ensure(max_slots_tiles * kAccStride * sizeof(double)));
  ICA_HIP(ctx, ctx->hinv.ensure(B * 64 * sizeof(double)));
  ICA_HIP(ctx, ctx->p_prev.ensure(B * 8 * sizeof(double)));
  ICA_HIP(ctx, ctx->lambda_it.ensure(B * sizeof(double)));
  ICA_HIP(ctx, ctx->error.ensure(B * sizeof(double)));
  ICA_HIP(ctx, ctx->niter.ensure(B * sizeof(int)));
  ICA_HIP(ctx, ctx->lists.ensure(3 * (size_t)kListShards * list_shard_cap((int)B) * sizeof(int)));
  ICA_HIP(ctx, ctx->counts.ensure(3 * kListShards * kShardStride * sizeof(int)));
  ICA_HIP(ctx, ctx->count_hist.ensure((size_t)pr->nscales * std::max(1, pr->max_iter) * sizeof(int)));
  ICA_HIP(ctx, ctx->minmax.ensure((size_t)pr->nscales * B * 2 * sizeof(float)));
  ICA_HIP(ctx, ctx->mm_in.ensure(B * 2 * sizeof(float)));
  for (int s = 1; s < pr->nscales; ++s) {
    const size_t bytes = B * pr->channels * g[s].H * g[s].W * sizeof(float);
    ICA_HIP(ctx, ctx->pyr1[s].ensure(bytes));
    ICA_HIP(ctx, ctx->pyr2[s].ensure(bytes));
    DeviceBandOp* tmp;
    ICA_TRY(get_op(ctx, g[s - 1].W, g[s].W, &tmp));
    ICA_TRY(get_op(ctx, g[s - 1].H, g[s].H, &tmp));
  }
  if (pr->nscales > 1)
    ICA_HIP(ctx, ctx->tmp_resample.ensure(B * pr->channels * g[1].H * g[0].W * sizeof(float)));
  if (!ctx->pinned_count) ICA_HIP(ctx, hipHostMalloc((void**)&ctx->pinned_count, kListShards * kShardStride * sizeof(int), hipHostMallocDefault));
  const size_t need = (size_t)pr->nscales * std::max(1, pr->max_iter);
  if (need > ctx->board_cap) {
    if (ctx->board) { ICA_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipHostFree(ctx->board); ctx->board = nullptr; }
    ICA_HIP(ctx, hipHostMalloc((void**)&ctx->board, need * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
    ICA_HIP(ctx, hipHostGetDevicePointer((void**)&ctx->board_dev, ctx->board, 0));
    ctx->board_cap = need;
  }
  return ICA_OK;
}

}  // namespace

extern "C" {

int ica_version(void) { return ICA_ABI_VERSION; }

int ica_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* ica_strerror(int status) {
  switch (status) {
    case ICA_OK: return "ok";
    case ICA_EINVAL_SHAPE: return "I1 and I2 must be RGB images with channels in the last dimension";
    case ICA_EINVAL_DIMS: return "I1 and I2 must have the same dimensions";
    case ICA_EINVAL_TOL: return "TOL must be positive and very small (less than 0.01)";
    case ICA_EINVAL_TRANSFORM: return "Unknown transform type";
    case ICA_EINVAL_ROBUST: return "Unknown type for robust error function";
    case ICA_EINVAL_ARG: return "invalid argument";
    case ICA_ENOMEM: return "device memory allocation failed";
    case ICA_EHIP: return "HIP runtime error";
    case ICA_ENODEVICE: return "no usable HIP device";
    default: return "unknown status";
  }
}

const char* ica_last_error(const ica_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int ica_create(int device, ica_ctx** out) {
  if (!out) return ICA_EINVAL_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return ICA_ENODEVICE;
  if (hipSetDevice(device) != hipSuccess) return ICA_ENODEVICE;
  auto* c = new ica_ctx();
  c->device = device;
  if (const char* e = getenv("ICA_ACCUM")) c->accum_variant = (atoi(e) == 1) ? 1 : 2;
  if (const char* e = getenv("ICA_SEPARATE_HESSIAN")) c->fold_hessian = atoi(e) == 0;
  if (const char* e = getenv("ICA_MAX_TILE_H")) c->max_tile_h = std::max(16, atoi(e));
  if (const char* e = getenv("ICA_PYRAMID_2PASS")) c->fused_pyramid = atoi(e) == 0;
  if (const char* e = getenv("ICA_BOARD_TIMEOUT_MS")) c->board_timeout_s = std::max(1, atoi(e)) * 1e-3;
  *out = c;
  return ICA_OK;
}

int ica_destroy(ica_ctx* ctx) {
  if (!ctx) return ICA_EINVAL_ARG;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  delete ctx;
  return ICA_OK;
}

int ica_set_stream(ica_ctx* ctx, void* s) {
  if (!ctx) return ICA_EINVAL_ARG;
  hipStream_t ns = reinterpret_cast<hipStream_t>(s);
  if (ns != ctx->stream) {
    // A context owns ONE set of workspaces (partials, lists, pyramid levels, the host board): work queued on the
    // stream it was bound to so far must finish before anything queued on the new stream touches them.  An event on
    // the old stream, awaited by the new one, gives that order without stalling the host.  (A context is still
    // single-threaded: one host thread at a time may call into it.)
    ICA_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->handover) ICA_HIP(ctx, hipEventCreateWithFlags(&ctx->handover, hipEventDisableTiming));
    ICA_HIP(ctx, hipEventRecord(ctx->handover, ctx->stream));
    ICA_HIP(ctx, hipStreamWaitEvent(ns, ctx->handover, 0));
    ctx->stream = ns;
  }
  return ICA_OK;
}

int ica_synchronize(ica_ctx* ctx) {
  if (!ctx) return ICA_EINVAL_ARG;
  ICA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return ICA_OK;
}

int ica_set_profiling(ica_ctx* ctx, int on) {
  if (!ctx) return ICA_EINVAL_ARG;
  ctx->profiling = on != 0;
  return ICA_OK;
}

// one wave: shader cycles (s_memtime) per tick of the constant 100 MHz counter (s_memrealtime) over `ticks` ticks
__global__ void ica_shader_clock_kernel(long long ticks, unsigned long long* out) {
  const long long r0 = wall_clock64(), c0 = clock64();
  long long r1 = r0;
  while (r1 - r0 < ticks) {
    __builtin_amdgcn_s_sleep(32);
    r1 = wall_clock64();
  }
  const long long c1 = clock64();
  r1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = (unsigned long long)(c1 - c0); out[1] = (unsigned long long)(r1 - r0); }
}

int ica_op_shader_clock_dev(ica_ctx* ctx, int duration_us, double* ghz_out) {
  if (!ctx || !ghz_out || duration_us <= 0 || duration_us > 10000000) return ICA_EINVAL_ARG;
  ICA_HIP(ctx, hipSetDevice(ctx->device));
  ICA_HIP(ctx, ctx->clock_probe.ensure(2 * sizeof(unsigned long long)));
  unsigned long long* d = ctx->clock_probe.as<unsigned long long>();
  hipLaunchKernelGGL(ica_shader_clock_kernel, dim3(1), dim3(64), 0, ctx->stream, (long long)duration_us * 100, d);
  ICA_HIP(ctx, hipGetLastError());
  unsigned long long h[2] = {0, 0};
  ICA_HIP(ctx, hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  ICA_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *ghz_out = h[1] ? (double)h[0] / ((double)h[1] * 10.0) : 0.0;
  return ICA_OK;
}

int ica_nparams(int tt) { return nparams_of(tt); }

int ica_zoom_size(int width, int height, double nu, int* ow, int* oh) {
  if (!ow || !oh) return ICA_EINVAL_ARG;
  *ow = round_half_even(width * nu);
  *oh = round_half_even(height * nu);
  return ICA_OK;
}

int ica_prepare(ica_ctx* ctx, const ica_problem* pr) {
  if (!ctx) return ICA_EINVAL_ARG;
  ICA_TRY(validate(pr));
  ICA_HIP(ctx, hipSetDevice(ctx->device));
  LevelGeom g[ICA_MAX_SCALES];
  level_sizes(ctx, pr, g);
  return ensure_state(ctx, pr, g);
}

int ica_solve_batch_dev(ica_ctx* ctx, const ica_problem* pr, const void* I1v, const void* I2v, double* p,
                        double* error_out, int32_t* niter_out, float* DI_out, float* Iw_out, double* trace_out,
                        ica_stats* stats) {
  // level 0 is the caller's images, float32 or uint8 (ica_problem.storage); every coarser level is float32
  const float* I1 = static_cast<const float*>(I1v);
  const float* I2 = static_cast<const float*>(I2v);
  const unsigned char* I1b = static_cast<const unsigned char*>(I1v);
  const unsigned char* I2b = static_cast<const unsigned char*>(I2v);
  if (!ctx || !I1 || !I2 || !p || !error_out || !niter_out) return ICA_EINVAL_ARG;
  ICA_TRY(validate(pr));
  ICA_HIP(ctx, hipSetDevice(ctx->device));
  LevelGeom g[ICA_MAX_SCALES];
  level_sizes(ctx, pr, g);
  ICA_TRY(ensure_state(ctx, pr, g));
  hipStream_t st = ctx->stream;
  const int B = pr->batch, C = pr->channels, tt = pr->transform_type, n = nparams_of(tt);
  const bool robust = pr->robust_type != ICA_QUADRATIC;
  const int ns = pr->nscales, max_iter = pr->max_iter;
  const bool u8 = pr->storage == ICA_STORAGE_U8;

  // ---- pyramid (ref :331-337) ----
  const float* L1[ICA_MAX_SCALES];
  const float* L2[ICA_MAX_SCALES];
  L1[0] = I1;
  L2[0] = I2;
  // min/max of every I2 level is needed twice (clip range of the next pyramid step and of the warp at that level):
  // computed once per level into mm2[s]
  ICA_HIP(ctx, ctx->minmax.ensure((size_t)ns * B * 2 * sizeof(float)));
  ICA_HIP(ctx, ctx->mm_in.ensure((size_t)B * 2 * sizeof(float)));
  float* mm2 = ctx->minmax.as<float>();
  bool have_coarsest = false;  // the coarsest I2 level's range, needed by its warp, may come out of the last step
  for (int s = 1; s < ns; ++s) {
    float* mm_next = s == ns - 1 ? mm2 + (size_t)(ns - 1) * B * 2 : nullptr;
    if (s == 1 && u8) {
      ICA_TRY(launch_pyramid_down(ctx, B, C, g[0].H, g[0].W, g[1].H, g[1].W, I1b, ctx->pyr1[1].as<float>(),
                                  ctx->mm_in.as<float>()));
      ICA_TRY(launch_pyramid_down(ctx, B, C, g[0].H, g[0].W, g[1].H, g[1].W, I2b, ctx->pyr2[1].as<float>(), mm2, mm_next,
                                  &have_coarsest));
    } else {
      ICA_TRY(launch_pyramid_down(ctx, B, C, g[s - 1].H, g[s - 1].W, g[s].H, g[s].W, L1[s - 1], ctx->pyr1[s].as<float>(),
                                  ctx->mm_in.as<float>()));
      ICA_TRY(launch_pyramid_down(ctx, B, C, g[s - 1].H, g[s - 1].W, g[s].H, g[s].W, L2[s - 1], ctx->pyr2[s].as<float>(),
                                  mm2 + (size_t)(s - 1) * B * 2, mm_next, &have_coarsest));
    }
    L1[s] = ctx->pyr1[s].as<float>();
    L2[s] = ctx->pyr2[s].as<float>();
  }
  if (ns == 1 && u8)
    ICA_TRY(launch_minmax(ctx, B, (size_t)C * g[0].H * g[0].W, I2b, mm2));
  else if (!have_coarsest)
    ICA_TRY(launch_minmax(ctx, B, (size_t)C * g[ns - 1].H * g[ns - 1].W, L2[ns - 1], mm2 + (size_t)(ns - 1) * B * 2));
  if (ctx->board_dirty) ICA_HIP(ctx, hipStreamSynchronize(st));  // nothing may post to the board while it is reset
  for (size_t i = 0; i < (size_t)ns * std::max(1, max_iter); ++i) ctx->board[i] = -1;
  ctx->board_dirty = true;
  ICA_HIP(ctx, hipMemsetAsync(ctx->count_hist.ptr, 0, (size_t)ns * std::max(1, max_iter) * sizeof(int), st));
  if (trace_out)
    ICA_HIP(ctx, hipMemsetAsync(trace_out, 0, (size_t)B * ns * max_iter * ICA_TRACE_STRIDE * sizeof(double), st));
  ICA_HIP(ctx, hipMemsetAsync(niter_out, 0, (size_t)B * ns * sizeof(int32_t), st));

  int* lists = ctx->lists.as<int>();
  int* counts = ctx->counts.as<int>();
  const int list_cap = list_shard_cap(B);                       // three sharded lists (ica_kernels.hpp)
  const size_t list_stride = (size_t)kListShards * list_cap, cnt_stride = (size_t)kListShards * kShardStride;
  if (ctx->profiling) {   // the iteration kernel's clock accumulators (ica_common.hpp, LevelArgs::clock_acc)
    ICA_HIP(ctx, ctx->clock_acc.ensure(4 * sizeof(unsigned long long)));
    ICA_HIP(ctx, hipMemsetAsync(ctx->clock_acc.ptr, 0, 4 * sizeof(unsigned long long), st));
  }
  struct EvRec { int level; size_t e0, e1; bool first; };
  std::vector<EvRec> evrecs;
  size_t ev_used = 0;
  auto next_event = [&](size_t* idx) -> int {
    if (ev_used == ctx->ev_pool.size()) {
      hipEvent_t e;
      ICA_HIP(ctx, hipEventCreate(&e));
      ctx->ev_pool.push_back(e);
    }
    *idx = ev_used++;
    return ICA_OK;
  };
  int64_t launches[ICA_MAX_SCALES] = {0}, first_launches[ICA_MAX_SCALES] = {0};

  // ---- coarse to fine (ref :340-372) ----
  for (int s = ns - 1; s >= 0; --s) {
    LevelInitArgs li{};
    li.batch = B; li.tt = tt; li.level = s; li.nscales = ns;
    if (s < ns - 1) { li.w_coarse = g[s + 1].W; li.h_coarse = g[s + 1].H; li.w_fine = g[s].W; li.h_fine = g[s].H; }
    li.lambda_user = pr->lambda; li.lambda0 = pr->lambda0;
    li.p = p; li.lambda_it = ctx->lambda_it.as<double>(); li.error = ctx->error.as<double>();
    li.niter = ctx->niter.as<int>(); li.niter_out = niter_out; li.list0 = lists; li.counts = counts;
    hipLaunchKernelGGL(ica_level_init_kernel, dim3((B + 255) / 256), dim3(256), 0, st, li);

    LevelArgs la{};
    la.I1 = L1[s]; la.I2 = L2[s]; la.H = g[s].H; la.W = g[s].W;
    la.u8 = (s == 0 && u8) ? 1 : 0;
    la.tiles_x = g[s].tiles_x; la.tiles_y = g[s].tiles_y; la.tile_h = g[s].tile_h;
    la.delta = pr->delta; la.nanif = pr->nanifoutside; la.robust_type = pr->robust_type;
    la.p = p; la.lambda_it = ctx->lambda_it.as<double>(); la.minmax = mm2 + (size_t)s * B * 2;
    la.partials = ctx->partials.as<double>();
    la.n_pairs = B;
    la.clock_acc = ctx->profiling ? ctx->clock_acc.as<unsigned long long>() : nullptr;

    // L2 (ref :96-103): the steepest-descent Hessian does not depend on p and is needed once per level.  Instead of a
    // separate pass over I1, iteration 0 runs the kernel variant that accumulates H next to b (weights identically 1)
    // and its solve stores inv(H) for the remaining iterations.  ICA_SEPARATE_HESSIAN=1 restores the separate pass.
    const bool l2_fold_hessian = !robust && ctx->fold_hessian;
    if (!robust && !l2_fold_hessian) {
      la.act_list = nullptr; la.act_count = nullptr; la.host_slot = nullptr;
      ICA_TRY(launch_accumulate_mode(ctx, kModeHessian, tt, C, dim3(g[s].tiles, B), st, la));
      hipLaunchKernelGGL(ica_setup_solve_kernel, dim3(B), dim3(kSolveThreads), 0, st, la.partials, g[s].tiles, n,
                         ctx->hinv.as<double>());
    }

    int nb = B;  // upper bound of the number of active pairs
    LevelGeom lg = g[s];
    for (int it = 0; it < max_iter && nb > 0; ++it) {
      const int cur = it % 3, nxt = (it + 1) % 3, clr = (it + 2) % 3;
      if (it >= 1) {
        // The fused kernel of iteration it-1 posts "pairs that entered it-1" to host-visible memory when it starts;
        // that is an upper bound for iteration `it`.  Waiting for it keeps the host exactly one iteration ahead of
        // the GPU -- the queue never runs dry and no stream synchronisation is needed.
        int v = wait_board(ctx->board + (size_t)s * max_iter + (it - 1), ctx->board_timeout_s);
        if (v < 0) {
          // The count did not arrive in time.  Either the GPU is far behind (another process owns it) or a kernel of
          // this solve has died: the synchronous read below tells the two apart -- a HIP error is reported to the
          // caller with last_error set, a slow GPU just costs this one synchronisation.
          ICA_HIP(ctx, hipMemcpyAsync(ctx->pinned_count, counts + (size_t)cur * kListShards * kShardStride,
                                      kListShards * kShardStride * sizeof(int), hipMemcpyDeviceToHost, st));
          ICA_HIP(ctx, hipStreamSynchronize(st));
          v = 0;
          for (int sh = 0; sh < kListShards; ++sh) v += ctx->pinned_count[sh * kShardStride];
        }
        nb = std::min(nb, v);
        if (nb == 0) break;
      }
      la.host_slot = ctx->board_dev + (size_t)s * max_iter + it;
      set_tiling(lg, nb, ctx->max_tile_h, level_chunk(pr, s));  // accumulate and solve of one iteration always agree on the tiling
      la.tiles_x = lg.tiles_x; la.tiles_y = lg.tiles_y; la.tile_h = lg.tile_h;
      la.act_list = lists + (size_t)cur * list_stride;
      la.act_count = counts + (size_t)cur * cnt_stride;
      la.list_cap = list_cap;
      size_t e0 = 0, e1 = 0;
      if (ctx->profiling) {
        ICA_TRY(next_event(&e0));
        ICA_HIP(ctx, hipEventRecord(ctx->ev_pool[e0], st));
      }
      SolveArgs sa{};
      sa.tt = tt; sa.n = n; sa.robust = robust; sa.flags = pr->flags; sa.tiles = lg.tiles; sa.max_iter = max_iter;
      sa.level = s; sa.nscales = ns; sa.iter = it; sa.tol = pr->tol; sa.lambda_user = pr->lambda;
      sa.lambdaN = pr->lambdaN; sa.lambda_ratio = pr->lambda_ratio;
      sa.partials = la.partials; sa.p = p; sa.p_prev = ctx->p_prev.as<double>(); sa.Hinv = ctx->hinv.as<double>();
      sa.lambda_it = ctx->lambda_it.as<double>(); sa.error = ctx->error.as<double>(); sa.niter = ctx->niter.as<int>();
      sa.cur_list = lists + (size_t)cur * list_stride; sa.cur_count = counts + (size_t)cur * cnt_stride;
      sa.nxt_list = lists + (size_t)nxt * list_stride; sa.nxt_count = counts + (size_t)nxt * cnt_stride;
      sa.clr_count = counts + (size_t)clr * cnt_stride; sa.list_cap = list_cap;
      sa.count_hist = ctx->count_hist.as<int>(); sa.trace = trace_out;
      const bool first_l2 = l2_fold_hessian && it == 0;
      sa.invert_h = robust || first_l2;  // the solve inverts the H it receives ...
      sa.store_hinv = first_l2;          // ... and keeps the inverse for the later L2 iterations
      la.robust_type = first_l2 ? (int)ICA_QUADRATIC : pr->robust_type;
      ICA_TRY(launch_accumulate_mode(ctx, robust ? kModeRobust : first_l2 ? kModeL2First : kModeL2, tt, C, dim3(lg.tiles, nb), st, la));
      if (ctx->profiling) {
        ICA_TRY(next_event(&e1));
        ICA_HIP(ctx, hipEventRecord(ctx->ev_pool[e1], st));
        evrecs.push_back({s, e0, e1, first_l2});
      }
      if (first_l2) first_launches[s]++;
      else launches[s]++;
      hipLaunchKernelGGL(ica_solve_kernel, dim3(nb), dim3(kSolveThreads), 0, st, sa);
    }
    ICA_HIP(ctx, hipGetLastError());
  }
  hipLaunchKernelGGL(ica_finish_kernel, dim3((B + 255) / 256), dim3(256), 0, st, B, ns, ctx->niter.as<int>(),
                     ctx->error.as<double>(), niter_out, error_out);
  if (DI_out || Iw_out) {
    // Iw / DI of the last executed iteration at scale 0 = warp under the parameters BEFORE the last
    // update (ref :111-114 then :123, returned at :133).  minmax still holds the scale-0 range.
    if (u8)
      ICA_TRY(launch_warp(ctx, B, C, g[0].H, g[0].W, tt, I2b, ctx->p_prev.as<double>(), mm2, I1b, Iw_out, DI_out));
    else
      ICA_TRY(launch_warp(ctx, B, C, g[0].H, g[0].W, tt, I2, ctx->p_prev.as<double>(), mm2, I1, Iw_out, DI_out));
  }
  ICA_HIP(ctx, hipGetLastError());

  if (stats) {
    memset(stats, 0, sizeof(*stats));
    std::vector<int> hist((size_t)ns * std::max(1, max_iter));
    ICA_HIP(ctx, hipMemcpyAsync(hist.data(), ctx->count_hist.ptr, hist.size() * sizeof(int), hipMemcpyDeviceToHost, st));
    ICA_HIP(ctx, hipStreamSynchronize(st));
    ctx->board_dirty = false;
    for (int s = 0; s < ns; ++s) {
      int64_t units = 0;
      for (int it = 0; it < max_iter; ++it) units += hist[(size_t)s * max_iter + it];
      stats->pair_iterations[s] = units;
      stats->pair_iterations_total += units;
      const double unit_bytes = 2.0 * g[s].H * g[s].W * C * 4.0;
      const int64_t first_units = first_launches[s] ? hist[(size_t)s * max_iter] : 0;
      stats->iter_launches[s] = launches[s];
      stats->iter_kernel_bytes[s] = (double)(units - first_units) * unit_bytes;
      stats->first_launches[s] = first_launches[s];
      stats->first_kernel_bytes[s] = (double)first_units * unit_bytes;
      stats->level_width[s] = g[s].W;
      stats->level_height[s] = g[s].H;
    }
    for (const auto& r : evrecs) {
      float ms = 0.f;
      ICA_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev_pool[r.e0], ctx->ev_pool[r.e1]));
      (r.first ? stats->first_kernel_ms : stats->iter_kernel_ms)[r.level] += ms;
    }
    if (ctx->profiling) {
      unsigned long long acc[2] = {0, 0};
      ICA_HIP(ctx, hipMemcpy(acc, ctx->clock_acc.ptr, sizeof(acc), hipMemcpyDeviceToHost));
      stats->shader_clock_ghz = acc[1] ? (double)acc[0] / ((double)acc[1] * 10.0) : 0.0;   // cycles per 10 ns tick
    }
  }
  return ICA_OK;
}

int ica_solve_batch(ica_ctx* ctx, const ica_problem* pr, const void* I1, const void* I2, double* p, double* error_out,
                    int32_t* niter_out, float* DI_out, float* Iw_out, double* trace_out, ica_stats* stats) {
  if (!ctx || !I1 || !I2 || !p || !error_out || !niter_out) return ICA_EINVAL_ARG;
  ICA_TRY(validate(pr));
  ICA_HIP(ctx, hipSetDevice(ctx->device));
  const size_t B = pr->batch, img = B * pr->channels * pr->height * pr->width;
  const size_t es = pr->storage == ICA_STORAGE_U8 ? 1 : 4;  // bytes per input pixel
  const size_t tr = B * pr->nscales * pr->max_iter * ICA_TRACE_STRIDE;
  DevBuf d1, d2, dp, de, dn, dDI, dIw, dtr;
  auto cleanup = [&]() { d1.release(); d2.release(); dp.release(); de.release(); dn.release(); dDI.release(); dIw.release(); dtr.release(); };
  int rc = ICA_OK;
  auto H = [&](hipError_t e, const char* what) {
    if (e != hipSuccess && rc == ICA_OK) {
      ctx->last_error = std::string(what) + ": " + hipGetErrorString(e);
      rc = e == hipErrorOutOfMemory ? ICA_ENOMEM : ICA_EHIP;
    }
  };
  H(d1.ensure(img * es), "alloc I1"); H(d2.ensure(img * es), "alloc I2"); H(dp.ensure(B * 8 * 8), "alloc p");
  H(de.ensure(B * 8), "alloc err"); H(dn.ensure(B * pr->nscales * 4), "alloc niter");
  if (DI_out) H(dDI.ensure(img * 4), "alloc DI");
  if (Iw_out) H(dIw.ensure(img * 4), "alloc Iw");
  if (trace_out) H(dtr.ensure(std::max<size_t>(tr, 1) * 8), "alloc trace");
  if (rc == ICA_OK) {
    H(hipMemcpyAsync(d1.ptr, I1, img * es, hipMemcpyHostToDevice, ctx->stream), "copy I1");
    H(hipMemcpyAsync(d2.ptr, I2, img * es, hipMemcpyHostToDevice, ctx->stream), "copy I2");
    H(hipMemcpyAsync(dp.ptr, p, B * 64, hipMemcpyHostToDevice, ctx->stream), "copy p");
  }
  if (rc == ICA_OK)
    rc = ica_solve_batch_dev(ctx, pr, d1.ptr, d2.ptr, dp.as<double>(), de.as<double>(), dn.as<int32_t>(),
                             DI_out ? dDI.as<float>() : nullptr, Iw_out ? dIw.as<float>() : nullptr,
                             trace_out ? dtr.as<double>() : nullptr, stats);
  if (rc == ICA_OK) {
    H(hipMemcpyAsync(p, dp.ptr, B * 64, hipMemcpyDeviceToHost, ctx->stream), "copy p back");
    H(hipMemcpyAsync(error_out, de.ptr, B * 8, hipMemcpyDeviceToHost, ctx->stream), "copy err back");
    H(hipMemcpyAsync(niter_out, dn.ptr, B * pr->nscales * 4, hipMemcpyDeviceToHost, ctx->stream), "copy niter back");
    if (DI_out) H(hipMemcpyAsync(DI_out, dDI.ptr, img * 4, hipMemcpyDeviceToHost, ctx->stream), "copy DI back");
    if (Iw_out) H(hipMemcpyAsync(Iw_out, dIw.ptr, img * 4, hipMemcpyDeviceToHost, ctx->stream), "copy Iw back");
    if (trace_out && tr) H(hipMemcpyAsync(trace_out, dtr.ptr, tr * 8, hipMemcpyDeviceToHost, ctx->stream), "copy trace back");
    H(hipStreamSynchronize(ctx->stream), "sync");
  } else {
    (void)hipStreamSynchronize(ctx->stream);
  }
  cleanup();
  return rc;
}

// ---- per-operator hooks --------------------------------------------------------------------------

int ica_op_minmax_dev(ica_ctx* ctx, int batch, int channels, int height, int width, const float* I, float* out) {
  if (!ctx || !I || !out || batch <= 0 || channels <= 0 || height <= 0 || width <= 0) return ICA_EINVAL_ARG;
  ICA_HIP(ctx, hipSetDevice(ctx->device));
  return launch_minmax(ctx, batch, (size_t)channels * height * width, I, out);
}

int ica_op_warp_dev(ica_ctx* ctx, int batch, int channels, int height, int width, int tt, const float* I2,
                    const double* p, const float* minmax, const float* I1, float* Iw_out, float* DI_out) {
  if (!ctx || !I2 || !p || batch <= 0 || height <= 0 || width <= 0) return ICA_EINVAL_ARG;
  if (channels < 1 || channels > ICA_MAX_CHANNELS) return ICA_EINVAL_SHAPE;
  if (nparams_of(tt) < 0) return ICA_EINVAL_TRANSFORM;
  if (DI_out && !I1) return ICA_EINVAL_ARG;
  ICA_HIP(ctx, hipSetDevice(ctx->device));
  if (!minmax) {
    ICA_HIP(ctx, ctx->minmax.ensure((size_t)batch * 2 * sizeof(float)));
    ICA_TRY(launch_minmax(ctx, batch, (size_t)channels * height * width, I2, ctx->minmax.as<float>()));
    minmax = ctx->minmax.as<float>();
  }
  return launch_warp(ctx, batch, channels, height, width, tt, I2, p, minmax, I1, Iw_out, DI_out);
}

int ica_op_normal_equations_dev(ica_ctx* ctx, const ica_problem* pr, const void* I1, const void* I2, const double* p,
                                const double* lambda_it, double* H_out, double* b_out) {
  if (!ctx || !I1 || !I2 || !p || !H_out || !b_out) return ICA_EINVAL_ARG;
  ICA_TRY(validate(pr));
  const bool robust = pr->robust_type != ICA_QUADRATIC;
  if (robust && !lambda_it) return ICA_EINVAL_ARG;
  ICA_HIP(ctx, hipSetDevice(ctx->device));
  ica_problem one = *pr;
  one.nscales = 1;
  LevelGeom g[ICA_MAX_SCALES];
  level_sizes(ctx, &one, g);
  ICA_TRY(ensure_state(ctx, &one, g));
  const int B = pr->batch, C = pr->channels, tt = pr->transform_type, n = nparams_of(tt);
  const bool u8 = pr->storage == ICA_STORAGE_U8;
  if (u8) ICA_TRY(launch_minmax(ctx, B, (size_t)C * g[0].H * g[0].W, static_cast<const unsigned char*>(I2), ctx->minmax.as<float>()));
  else ICA_TRY(launch_minmax(ctx, B, (size_t)C * g[0].H * g[0].W, static_cast<const float*>(I2), ctx->minmax.as<float>()));
  LevelArgs la{};
  la.I1 = static_cast<const float*>(I1); la.I2 = static_cast<const float*>(I2); la.H = g[0].H; la.W = g[0].W;
  la.u8 = u8 ? 1 : 0;
  la.tiles_x = g[0].tiles_x; la.tiles_y = g[0].tiles_y; la.tile_h = g[0].tile_h;
  la.delta = pr->delta; la.nanif = pr->nanifoutside; la.robust_type = pr->robust_type;
  la.p = p; la.lambda_it = lambda_it; la.minmax = ctx->minmax.as<float>();
  la.partials = ctx->partials.as<double>();
  la.n_pairs = B;
  la.act_list = nullptr; la.act_count = nullptr;
  if (robust) {
    ICA_TRY(launch_accumulate_mode(ctx, kModeRobust, tt, C, dim3(g[0].tiles, B), ctx->stream, la));
    hipLaunchKernelGGL(ica_gather_normal_equations_kernel, dim3(B), dim3(kSolveThreads), 0, ctx->stream, la.partials, g[0].tiles, n,
                       B, 1, 1, (const double*)nullptr, H_out, b_out);
  } else {
    ICA_TRY(launch_accumulate_mode(ctx, kModeHessian, tt, C, dim3(g[0].tiles, B), ctx->stream, la));
    hipLaunchKernelGGL(ica_gather_normal_equations_kernel, dim3(B), dim3(kSolveThreads), 0, ctx->stream, la.partials, g[0].tiles, n,
                       B, 0, 1, (const double*)nullptr, H_out, b_out);
    ICA_TRY(launch_accumulate_mode(ctx, kModeL2, tt, C, dim3(g[0].tiles, B), ctx->stream, la));
    hipLaunchKernelGGL(ica_gather_normal_equations_kernel, dim3(B), dim3(kSolveThreads), 0, ctx->stream, la.partials, g[0].tiles, n,
                       B, 1, 0, (const double*)nullptr, H_out, b_out);
  }
  ICA_HIP(ctx, hipGetLastError());
  return ICA_OK;
}

int ica_op_solve_update_dev(ica_ctx* ctx, int batch, int tt, int flags, const double* H, const double* b, double* p,
                            double* dp_out, double* error_out) {
  if (!ctx || !H || !b || !p || batch <= 0) return ICA_EINVAL_ARG;
  if (nparams_of(tt) < 0) return ICA_EINVAL_TRANSFORM;
  ICA_HIP(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(ica_solve_update_kernel, dim3((batch + 63) / 64), dim3(64), 0, ctx->stream, batch, tt, flags, H, b, p,
                     dp_out, error_out);
  ICA_HIP(ctx, hipGetLastError());
  return ICA_OK;
}

int ica_op_pyramid_down_dev(ica_ctx* ctx, int batch, int channels, int height, int width, double nu, const float* in,
                            float* out) {
  if (!ctx || !in || !out || batch <= 0 || channels <= 0 || height <= 0 || width <= 0 || !(nu > 0)) return ICA_EINVAL_ARG;
  ICA_HIP(ctx, hipSetDevice(ctx->device));
  const int w = std::max(1, round_half_even(width * nu)), h = std::max(1, round_half_even(height * nu));
  ICA_HIP(ctx, ctx->mm_in.ensure((size_t)batch * 2 * sizeof(float)));
  return launch_pyramid_down(ctx, batch, channels, height, width, h, w, in, out, ctx->mm_in.as<float>());
}

int ica_op_ref_jacobian_dev(ica_ctx* ctx, int tt, int nx, int ny, double* J) {
  if (!ctx || !J || nx < 0 || ny < 0) return ICA_EINVAL_ARG;
  const int n = nparams_of(tt);
  if (n < 0) return ICA_EINVAL_TRANSFORM;
  const size_t px = (size_t)nx * ny;
  if (px == 0) return ICA_OK;
  ICA_HIP(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(ica_ref_jacobian_kernel, dim3((unsigned)((px + kRefBlock - 1) / kRefBlock)), dim3(kRefBlock), 0, ctx->stream,
                     tt, n, nx, ny, J);
  ICA_HIP(ctx, hipGetLastError());
  return ICA_OK;
}

int ica_op_ref_steepest_descent_dev(ica_ctx* ctx, int ny, int nx, int nz, int n, const double* Ix, const double* Iy,
                                    const double* J, double* DIJ) {
  if (!ctx || ny < 0 || nx < 0 || nz < 0 || n < 1 || n > kRefMaxN) return ICA_EINVAL_ARG;
  const size_t total = (size_t)ny * nx * nz * n;
  if (total == 0) return ICA_OK;
  if (!Ix || !Iy || !J || !DIJ) return ICA_EINVAL_ARG;
  ICA_HIP(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(ica_ref_sd_images_kernel, dim3((unsigned)((total + kRefBlock - 1) / kRefBlock)), dim3(kRefBlock), 0,
                     ctx->stream, (size_t)ny * nx, nz, n, Ix, Iy, J, DIJ);
  ICA_HIP(ctx, hipGetLastError());
  return ICA_OK;
}

int ica_op_ref_normal_dev(ica_ctx* ctx, int ny, int nx, int nz, int n, const double* DIJ, const double* DI, const double* rho,
                          double* H, double* b) {
  if (!ctx || ny < 0 || nx < 0 || nz < 0 || n < 1 || n > kRefMaxN || (b && !DI)) return ICA_EINVAL_ARG;
  ICA_HIP(ctx, hipSetDevice(ctx->device));
  if (H) ICA_HIP(ctx, hipMemsetAsync(H, 0, (size_t)n * n * sizeof(double), ctx->stream));
  if (b) ICA_HIP(ctx, hipMemsetAsync(b, 0, (size_t)n * sizeof(double), ctx->stream));
  const size_t px = (size_t)ny * nx;
  if (px == 0 || nz == 0 || (!H && !b)) return ICA_OK;
  if (!DIJ) return ICA_EINVAL_ARG;
  const int blocks = (int)std::min<size_t>(kRefBlocks, (px + kRefBlock - 1) / kRefBlock);
  ICA_HIP(ctx, ctx->ref_part.ensure((size_t)blocks * kRefAcc * sizeof(double)));
  double* part = ctx->ref_part.as<double>();
#define ICA_REF_CASE(NN)                                                                                              \
  case NN:                                                                                                            \
    hipLaunchKernelGGL(ica_ref_normal_partial_kernel<NN>, dim3(blocks), dim3(kRefBlock), 0, ctx->stream, px, nz, DIJ, DI, rho, \
                       part);                                                                                         \
    hipLaunchKernelGGL(ica_ref_normal_final_kernel<NN>, dim3(1), dim3(64), 0, ctx->stream, blocks, part, H, b);        \
    break;
  switch (n) {
    ICA_REF_CASE(1) ICA_REF_CASE(2) ICA_REF_CASE(3) ICA_REF_CASE(4) ICA_REF_CASE(5) ICA_REF_CASE(6) ICA_REF_CASE(7)
    ICA_REF_CASE(8)
    default: break;
  }
#undef ICA_REF_CASE
  ICA_HIP(ctx, hipGetLastError());
  return ICA_OK;
}

int ica_op_ref_rhop_dev(ica_ctx* ctx, size_t count, const double* t2, double lambda, int robust_type, double* out) {
  if (!ctx) return ICA_EINVAL_ARG;
  if (robust_type < ICA_QUADRATIC || robust_type > ICA_CHARBONNIER) return ICA_EINVAL_ROBUST;
  if (count == 0) return ICA_OK;
  if (!t2 || !out) return ICA_EINVAL_ARG;
  ICA_HIP(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(ica_ref_rhop_kernel, dim3((unsigned)((count + kRefBlock - 1) / kRefBlock)), dim3(kRefBlock), 0, ctx->stream,
                     count, t2, lambda, robust_type, out);
  ICA_HIP(ctx, hipGetLastError());
  return ICA_OK;
}

int ica_op_ref_robust_error_dev(ica_ctx* ctx, int ny, int nx, int nz, const double* DI, double lambda, int robust_type,
                                double* rho) {
  if (!ctx || ny < 0 || nx < 0 || nz < 0) return ICA_EINVAL_ARG;
  if (robust_type < ICA_QUADRATIC || robust_type > ICA_CHARBONNIER) return ICA_EINVAL_ROBUST;
  const size_t px = (size_t)ny * nx;
  if (px == 0) return ICA_OK;
  if (!rho || (nz > 0 && !DI)) return ICA_EINVAL_ARG;
  ICA_HIP(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(ica_ref_robust_error_kernel, dim3((unsigned)((px + kRefBlock - 1) / kRefBlock)), dim3(kRefBlock), 0,
                     ctx->stream, px, nz, DI, lambda, robust_type, rho);
  ICA_HIP(ctx, hipGetLastError());
  return ICA_OK;
}

int ica_op_zoom_in_parameters_dev(ica_ctx* ctx, int batch, int tt, int w, int h, int ww, int hh, double* p) {
  if (!ctx || !p || batch <= 0 || w <= 0 || h <= 0) return ICA_EINVAL_ARG;
  if (nparams_of(tt) < 0) return ICA_EINVAL_TRANSFORM;
  ICA_HIP(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(ica_zoom_params_kernel, dim3((batch + 255) / 256), dim3(256), 0, ctx->stream, batch, tt, w, h, ww, hh, p);
  ICA_HIP(ctx, hipGetLastError());
  return ICA_OK;
}

int ica_op_transform_image_dev(ica_ctx* ctx, int batch, int channels, int height, int width, const float* I,
                               const double* M, float* out) {
  if (!ctx || !I || !M || !out || batch <= 0 || height <= 0 || width <= 0) return ICA_EINVAL_ARG;
  if (channels < 1 || channels > ICA_MAX_CHANNELS) return ICA_EINVAL_SHAPE;
  ICA_HIP(ctx, hipSetDevice(ctx->device));
  ICA_HIP(ctx, ctx->mm_in.ensure((size_t)batch * 2 * sizeof(float)));
  ICA_TRY(launch_minmax(ctx, batch, (size_t)channels * height * width, I, ctx->mm_in.as<float>()));
  for (int b0 = 0; b0 < batch; b0 += 32768) {
    const int nb = std::min(32768, batch - b0);
    const size_t off = (size_t)b0 * channels * height * width;
    dim3 grid((width + 63) / 64, (height + kWaves - 1) / kWaves, nb);
#define ICA_TI_CASE(CC)                                                                                                  \
  case CC:                                                                                                             \
    hipLaunchKernelGGL(ica_transform_image_kernel<CC>, grid, dim3(kBlock), 0, ctx->stream, I + off, height, width,      \
                       M + (size_t)b0 * 9, ctx->mm_in.as<float>() + (size_t)b0 * 2, out + off);                        \
    break;
    switch (channels) { ICA_TI_CASE(1) ICA_TI_CASE(2) ICA_TI_CASE(3) ICA_TI_CASE(4) default: break; }
#undef ICA_TI_CASE
  }
  ICA_HIP(ctx, hipGetLastError());
  return ICA_OK;
}

}  // extern "C"
