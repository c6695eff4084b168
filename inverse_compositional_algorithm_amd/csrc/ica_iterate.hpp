// Fused per-iteration pass, LDS-staged (the production kernel).
//
// One workgroup (4 waves) owns a 64-column x tile_h-row tile of one image pair and walks it in chunks of
// CH rows (32 for one channel, 16 for three).  For every chunk:
//   1. The footprint of the chunk under the current warp is the convex quadrilateral of its 4 mapped
//      corners; its bounding box (+ bicubic halo) of I2 and the chunk (+1-px halo) of I1 are moved
//      HBM/L2 -> LDS by LDS-DMA (buffer_load ... lds, 16 B per lane, no VGPRs) into ONE tile per workgroup: barrier, copy
//      chunk k, barrier, arithmetic on chunk k -- a workgroup's copy is not hidden behind its own arithmetic, the other
//      five workgroups of the CU cover it (two tiles with the next chunk copied under the current one were measured
//      slower, like everything else that trades workgroups per CU for buffering: DESIGN.md 5.1).  A footprint that
//      does not fit the tile falls back to direct cached gathers for that chunk -- same arithmetic, same results.
//   2. Each lane owns one image column; each wave walks RW = CH / 4 consecutive rows.  All 16 bicubic taps and the
//      5-point gradient stencil come from LDS with compile-time offsets; nothing but the two images is ever
//      read from memory (the reference materialises 192 B/pixel of gradient/Jacobian/steepest-descent tensors).
//   3. "Moment" accumulation.  Every Jacobian entry is a monomial c*x^a*y^b (derivatives.py:33-68), so
//        b_k  = sum_pix (Jx_k Gx + Jy_k Gy),                 Gx = rho' * sum_c Ix_c DI_c, Gy likewise
//        H_km = sum_pix (Jx_k Jx_m Sxx + (Jx_k Jy_m + Jy_k Jx_m) Sxy + Jy_k Jy_m Syy),  Sxx = rho' sum_c Ix_c^2 ...
//      x^a is constant per lane and factors out of the row loop: per pixel only the y-moments
//      sum_y y^b G* and sum_y y^b S** are accumulated (4 FMAs for affine L2 instead of 6 products + 6 FMAs;
//      12 + 5 instead of (36 + 8 + 8) * C for robust homography, independent of C).  The x powers are applied
//      once per tile, in fp64, when the 64 lanes' moments are folded into the <= 44 sums.
//   fp32 per pixel, promoted to fp64 at the end of the tile; fixed summation order throughout (deterministic).
//
// Three instruction streams per chunk.  The kernel is not bound by HBM bytes but by three stations at 60-85 % each -- the
// L1's line requests for the tile copies, the LDS reads of the taps, vector-instruction issue -- and on
// gfx950 the issue cost of a vector instruction depends on its FORM (tools/ubench/valu_floor.hip: 2 cycles for VOP2 /
// VOP3 on registers, inline constants or a literal; 4 with an SGPR operand, for v_floor / v_fract / v_cvt / v_max /
// v_med3 / shifts and for packed fp32 (two results); 8 for v_rcp / v_rsq; an fp32 MFMA blocks the issue port for its
// whole duration).  So:
//   * INTERIOR chunks -- no pixel in the NaN frame or on the gradient border, every bicubic footprint inside I2,
//     decided once per chunk from the mapped corners (the large majority of a level-0 image) -- run `band_fast`:
//     no masks, no validity tests, no selects; the wave's rows unrolled; every wave-uniform operand (y powers, warp
//     coefficients, clip range, lambda^2) is kept in a VGPR so the multiply-adds are VOP2 v_fmac on registers; the
//     robust function is a template argument (no branches in the loop); the tap address formed in fp32 (exact small
//     integers); the Keys weights in their factored form (9 operations per axis).
//   * MASKED-border chunks (one channel) run the same stream with the frame pixels' residual and weight selected to 0.
//   * The remaining BORDER chunks run `band_careful`, which keeps every test of the reference's pixel masks, and
//     re-evaluates a sampling position in float64 when its fp32 value is within 2^-16 of an integer: there floor()
//     decides whether a border pixel's 4x4 footprint counts as inside the image (bicubic_interpolation.py:199,
//     cval=nan), and fp32 and the reference's float64 could otherwise part ways.
#pragma once
#include <type_traits>

#ifndef ICA_ITER_WAVES
#define ICA_ITER_WAVES 6
#endif
#ifndef ICA_ITER_WAVES_U8   // waves per SIMD asked of the one-channel uint8-storage kernel (28.6 KB of LDS: 5 workgroups per CU)
#define ICA_ITER_WAVES_U8 5
#endif
#ifndef ICA_ROW_FENCE
#define ICA_ROW_FENCE 1
#endif
#ifndef ICA_NBUF   // float tiles per workgroup for one channel (2: chunk k+1 copied under chunk k's arithmetic)
#define ICA_NBUF 1
#endif
#ifndef ICA_CH     // image rows per chunk, one channel
#define ICA_CH 32
#endif
#ifndef ICA_TAP_BATCH   // interior band: all LDS reads of a pixel in one block, one wait (0: reads left to the compiler)
#define ICA_TAP_BATCH 0
#endif
#ifndef ICA_POS_LINEAR   // interior band, homography: displacement from per-column polynomials in y (fewer instructions)
#define ICA_POS_LINEAR 0
#endif

#include "ica_common.hpp"

namespace ica {

#ifndef ICA_CWAVES3   // computing waves per workgroup, three channels (8 with ICA_CH3=32: one 68.7 KB tile, 2 workgroups per CU)
#define ICA_CWAVES3 4
#endif
// waves of a workgroup that walk rows (ch / compute_waves each)
constexpr int compute_waves(int channels) { return channels == 3 ? ICA_CWAVES3 : 4; }

// Shape of a workgroup per instantiation.
// ICA_LOADER = n > 0 (an experiment, off by default): n extra LOADER waves issue every LDS-DMA copy of the workgroup and
// compute nothing, the four computing waves never issue or wait for a copy; chunk k + 1 is copied into a second tile
// while chunk k is computed (two 18 KB tiles of 24-row chunks, 4 workgroups per CU), one barrier per chunk.  The idea:
// in the round-2 kernel a wave spends 8.5 % of its time issuing its share of the copies (~180 cycles per 1 KB
// instruction), 10.8 % waiting for them to land and 2.4 % at the barrier in front (profiles/r03_stamps_*), time in which
// it does not compete for the vector ALU -- which the kernel without copies keeps ~85 % busy and the kernel with copies
// 68 %.  Measured (profiles/r03_ablation.txt): 3395 us with one loader wave and 3887 us with two against 2660 us for the
// robust homography kernel, 3144 against 2238 for L2 homography: the copies of a chunk, issued by one or two waves
// one after the other, take longer than the four computing waves need for the chunk.
template <int C, bool U8> struct IterCfg {
  static constexpr bool loader = (ICA_LOADER != 0) && C == 1 && !U8;
  static constexpr int cw = compute_waves(C);
  static constexpr int copy_waves = loader ? ICA_LOADER : cw;   // ICA_LOADER = number of loader waves
  static constexpr int waves = cw + (loader ? copy_waves : 0);
  static constexpr int block = 64 * waves;
};

#ifdef ICA_STAMP
// Diagnostic build (tools/mkvariant.sh stamp -DICA_STAMP): shader cycles a wave spends in each phase of a chunk, summed
// over all waves of all launches since the last read (ica_debug_stamps_tt<N> in ica_iterate_inst.hip):
// [0] waiting at the barrier that ends the previous chunk  [1] issuing the LDS-DMA copies  [2] waiting for them to land
// [3] the band (arithmetic + LDS reads)  [4] prologue (plan)  [5] fold  [6] whole kernel  [7] whole kernel in ticks of the constant 100 MHz clock
constexpr int kStampWaves = 1 << 18;   // one record of 8 counters per wave of the LAST launch (no atomics: 65536 waves
// adding to one line congest the L2 and the copies queue behind them)
static __device__ unsigned long long ica_stamp_acc[kStampWaves * 16];
#endif

struct Mono { signed char c, a, b; };  // c * x^a * y^b; c == 0: structurally zero

// Jacobian rows dW/dp (derivatives.py:33-68) as monomials; jac_x(tt, k) = dx'/dp_k, jac_y(tt, k) = dy'/dp_k
__host__ __device__ constexpr Mono jac_x(int tt, int k) {
  constexpr Mono Z{0, 0, 0}, ONE{1, 0, 0}, X{1, 1, 0}, Y{1, 0, 1}, NY{-1, 0, 1}, NXX{-1, 2, 0}, NXY{-1, 1, 1};
  switch (tt) {
    case ICA_TRANSLATION: return k == 0 ? ONE : Z;
    case ICA_EUCLIDEAN: return k == 0 ? ONE : k == 2 ? NY : Z;
    case ICA_SIMILARITY: return k == 0 ? ONE : k == 2 ? X : k == 3 ? NY : Z;
    case ICA_AFFINITY: return k == 0 ? ONE : k == 2 ? X : k == 3 ? Y : Z;
    case ICA_HOMOGRAPHY: return k == 0 ? X : k == 1 ? Y : k == 2 ? ONE : k == 6 ? NXX : k == 7 ? NXY : Z;
    default: return Z;
  }
}
__host__ __device__ constexpr Mono jac_y(int tt, int k) {
  constexpr Mono Z{0, 0, 0}, ONE{1, 0, 0}, X{1, 1, 0}, Y{1, 0, 1}, NXY{-1, 1, 1}, NYY{-1, 0, 2};
  switch (tt) {
    case ICA_TRANSLATION: return k == 1 ? ONE : Z;
    case ICA_EUCLIDEAN: return k == 1 ? ONE : k == 2 ? X : Z;
    case ICA_SIMILARITY: return k == 1 ? ONE : k == 2 ? Y : k == 3 ? X : Z;
    case ICA_AFFINITY: return k == 1 ? ONE : k == 4 ? X : k == 5 ? Y : Z;
    case ICA_HOMOGRAPHY: return k == 3 ? X : k == 4 ? Y : k == 5 ? ONE : k == 6 ? NXY : k == 7 ? NYY : Z;
    default: return Z;
  }
}

// highest y power among the Jx (mbx) / Jy (mby) entries of each transform
template <int TT> struct MomShape;
template <> struct MomShape<ICA_TRANSLATION> { static constexpr int mbx = 0, mby = 0; };
template <> struct MomShape<ICA_EUCLIDEAN> { static constexpr int mbx = 1, mby = 0; };
template <> struct MomShape<ICA_SIMILARITY> { static constexpr int mbx = 1, mby = 1; };
template <> struct MomShape<ICA_AFFINITY> { static constexpr int mbx = 1, mby = 1; };
template <> struct MomShape<ICA_HOMOGRAPHY> { static constexpr int mbx = 1, mby = 2; };

template <int TT> struct MomLayout {
  static constexpr int mbx = MomShape<TT>::mbx, mby = MomShape<TT>::mby;
  static constexpr int ngx = mbx + 1, ngy = mby + 1, nxx = 2 * mbx + 1, nxy = mbx + mby + 1, nyy = 2 * mby + 1;
  static constexpr int ogx = 0, ogy = ngx, oxx = ngx + ngy, oxy = oxx + nxx, oyy = oxy + nxy;
  static constexpr int nmom = oyy + nyy;  // <= 17
  static constexpr int maxpow = 2 * mby > 2 * mbx ? 2 * mby : 2 * mbx;  // highest y power any moment needs
};

template <int C, bool LOADER = false, bool DUP = false> struct TileCfg {
  // Image rows per chunk: 4 waves x ch/4 rows.  One channel: 32 rows in ONE 22.9 KB tile (6 workgroups per CU), copy and
  // arithmetic of a workgroup alternate and the other five overlap them.  Measured against 16-row chunks in two tiles
  // (chunk k+1 copied under chunk k's arithmetic, 26.4 KB): 3-7 % faster on every one-channel kernel -- half the
  // workgroup barriers and copy set-ups per row, 41 instead of 50 footprint rows fetched per 32 rows of output -- where
  // more waves, more buffers or fewer instructions had bought nothing (profiles/r02_ablation.txt).  The uint8 kernel
  // uses the same chunk (bit-identical results need the same rows in the same waves); with its staging area it holds
  // 28.6 KB, 5 workgroups per CU, and is still 4 % faster than with 16-row chunks.
  // Three channels: 16 rows, one 39.5 KB tile, 4 workgroups per CU.
  // Three channels, ICA_CSEQ = 1 (an experiment, off by default): the tile holds ONE channel at a time, at the one-channel
  // geometry (32-row chunks, 22.9 KB).  A chunk is walked in three passes -- copy channel c, run the band on it -- with the
  // per-pixel sums over the channels (gx, gy, sxx, sxy, syy, t^2: six registers per row of the wave's band) carried across
  // the passes; rho' and the moments follow the third.  The idea: three RESIDENT channels only fit as 16-row chunks, whose
  // 9 + 2 halo rows per chunk cost 1.61x the algorithmic bytes from HBM.  Measured on 256 pairs of 1080 x 1920 x 3
  // (profiles/r03_ablation.txt): traffic 1.61x -> 1.28x, but 4310 us against 3640 us for robust homography, 1866 against
  // 1230 us for the Hessian pass, L2 rhs passes unchanged -- three copy / barrier phases per 32 rows instead of two, 38 %
  // more vector instructions (positions and weights once per pass), and the same LDS reads, which are what this kernel is
  // bound by (57 dwords per pixel row).
  static constexpr bool seq = (C == 3) && (ICA_CSEQ != 0);
  static constexpr int lc = seq ? 1 : C;               // channels resident in the tile
  static constexpr bool dup = DUP;                     // a second, one-float-shifted copy of the I2 footprint (ICA_DUP)
  static constexpr int ch = LOADER ? ICA_CH_LOADER : DUP ? ICA_CH_DUP : (C == 1) ? ICA_CH : ICA_CH3;
  static constexpr int rw = ch / compute_waves(C);
  static_assert(ch % compute_waves(C) == 0, "a chunk is split evenly among the computing waves");
  static constexpr int p2 = 80, r2 = ch + 9;           // I2 footprint buffer: pitch (floats) x rows
  static constexpr int p1 = 72, r1 = ch + 2;           // I1 buffer: columns x_t0-4 .. x_t0+67, rows yc0-1 .. yc0+ch
  static constexpr int nbuf = LOADER ? 2 : (C == 1) ? ICA_NBUF : 1;
  static constexpr int buf_floats = lc * (p2 * r2 + p1 * r1) + (DUP ? p2 * r2 : 0);
};

// 4 bytes per lane, global -> LDS at (wave-uniform) lds_base + 4*lane: rows that are not 16-byte aligned
__device__ __forceinline__ void glds4(const float* gsrc, float* lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_base, 4, 0, 0);
}

// one LDS-DMA instruction: 4 (uint8 storage) or 16 (float32) bytes per lane from descriptor `rs` at byte offset `off`
// to LDS `dst` + bytes * lane.  (The size operand of the builtin must be a literal.)
#ifndef ICA_DMA_AUX1   // cache policy bits of the I1 / I2 tile copies (0: default, 2: nt, 1: sc0, 16: sc1) -- experiment hooks
#define ICA_DMA_AUX1 0
#endif
#ifndef ICA_DMA_AUX2
#define ICA_DMA_AUX2 0
#endif
#define ICA_LDS_DMA_AUX(U8, rs, dst, off, AUX)                                                                        \
  do {                                                                                                                \
    if constexpr (U8)                                                                                                 \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst), 4, (int)(off), 0, 0, AUX); \
    else                                                                                                              \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst), 16, (int)(off), 0, 0, AUX); \
  } while (0)
#define ICA_LDS_DMA(U8, rs, dst, off) ICA_LDS_DMA_AUX(U8, rs, dst, off, 0)

// A wave-uniform value the compiler would keep in an SGPR (a VALU operand read from an SGPR costs 1.5-2.4x the
// issue time of the same instruction on a VGPR, see the header): pin it to a VGPR.
__device__ __forceinline__ float in_vgpr(float v) {
  asm("" : "+v"(v));
  return v;
}
// the same, but placed where it is written (a volatile asm is neither hoisted out of a loop nor merged)
__device__ __forceinline__ float pin_vgpr(float v) {
  float r;
  asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(v));
  return r;
}
__device__ __forceinline__ int in_vgpr(int v) {
  asm("" : "+v"(v));
  return v;
}
template <int J> using IC = std::integral_constant<int, J>;
// floor(x) as an integer in one instruction (v_floor_f32 + v_cvt_i32_f32 otherwise)
__device__ __forceinline__ int floor_to_int(float x) {
  int r;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}


// ---- batched LDS reads of the interior band ------------------------------------------------------------------------
// Left to the compiler, every pair of taps is read and waited for on its own (ds_read2_b32, s_waitcnt lgkmcnt(0), two
// multiply-adds: ten exposed LDS round trips per pixel row, the registers being short), and a wave spends most of its
// band waiting for the LDS: measured 1260 cycles per row and wave where its ~105 vector instructions need ~600 at the
// rate one wave can issue them (profiles/r03_stamps_*.txt).  Here all 16 taps of a pixel and the three I1 values of
// its row are requested in ONE block of reads, the Keys weights are computed while they are in flight (the weights are
// passed through the block so that the compiler cannot move them in front of it, and through the wait so that it cannot
// move them behind it), and one wait follows.
typedef float ica_f2 __attribute__((ext_vector_type(2)));
struct Taps16 { ica_f2 p[8]; };   // p[2r] = taps 0,1 of footprint row r, p[2r+1] = taps 2,3

template <int P2>
__device__ __forceinline__ void lds_read_taps(unsigned addr, Taps16& t, float& fx, float& fy) {
  static_assert(P2 == 80, "ds_read2_b32 offsets are written out for a pitch of 80 floats");
  asm volatile(
      "ds_read2_b32 %0, %10 offset1:1\n"
      "ds_read2_b32 %1, %10 offset0:2 offset1:3\n"
      "ds_read2_b32 %2, %10 offset0:80 offset1:81\n"
      "ds_read2_b32 %3, %10 offset0:82 offset1:83\n"
      "ds_read2_b32 %4, %10 offset0:160 offset1:161\n"
      "ds_read2_b32 %5, %10 offset0:162 offset1:163\n"
      "ds_read2_b32 %6, %10 offset0:240 offset1:241\n"
      "ds_read2_b32 %7, %10 offset0:242 offset1:243"
      : "=&v"(t.p[0]), "=&v"(t.p[1]), "=&v"(t.p[2]), "=&v"(t.p[3]), "=&v"(t.p[4]), "=&v"(t.p[5]), "=&v"(t.p[6]),
        "=&v"(t.p[7]), "+v"(fx), "+v"(fy)
      : "v"(addr)
      : "memory");
}
// the centre value of the row below (`dn`) and the left / right neighbours of this row's centre, from the I1 tile;
// `off` = byte offset of this row's centre from `addr` (a compile-time constant in the unrolled band)
template <int OFF, int P1>
__device__ __forceinline__ void lds_read_i1(unsigned addr, float& dn, float& lf, float& rt) {
  asm volatile(
      "ds_read_b32 %0, %3 offset:%4\n"
      "ds_read_b32 %1, %3 offset:%5\n"
      "ds_read_b32 %2, %3 offset:%6"
      : "=&v"(dn), "=&v"(lf), "=&v"(rt)
      : "v"(addr), "n"(OFF + P1 * 4), "n"(OFF - 4), "n"(OFF + 4)
      : "memory");
}
__device__ __forceinline__ void lds_wait_taps(Taps16& t, float (&wx)[4], float (&wy)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(t.p[0]), "+v"(t.p[1]), "+v"(t.p[2]), "+v"(t.p[3]), "+v"(t.p[4]), "+v"(t.p[5]), "+v"(t.p[6]),
                 "+v"(t.p[7]), "+v"(wx[0]), "+v"(wx[1]), "+v"(wx[2]), "+v"(wx[3]), "+v"(wy[0]), "+v"(wy[1]), "+v"(wy[2]),
                 "+v"(wy[3])
               :
               : "memory");
}
__device__ __forceinline__ void lds_wait_i1(float& dn, float& lf, float& rt) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dn), "+v"(lf), "+v"(rt) : : "memory");
}

template <bool PROJ>
__device__ __forceinline__ void displacement_col(const float* A, float cx, float cy, float gx, float xf, float yf,
                                                 float& dx, float& dy) {
  // displacement<>() with the per-column terms (cx = a0 x + a2, cy = a3 x + a5, gx = a6 x) hoisted
  float nx = fmaf(A[1], yf, cx);
  float ny = fmaf(A[4], yf, cy);
  if constexpr (PROJ) {
    const float g = fmaf(A[7], yf, gx);
    const float r = rcp_nr(1.0f + g);
    nx = (nx - xf * g) * r;
    ny = (ny - yf * g) * r;
  }
  dx = nx;
  dy = ny;
}

// rho'(t^2; lambda) with the error function known at compile time (no branches in the pixel loop).  The hardware
// reciprocal / reciprocal square root are good to 1 ulp, which is all a per-pixel WEIGHT needs (the same weight
// multiplies the pixel's contribution to H and to b); the sampling POSITION keeps its Newton step (rcp_nr).
template <int RT>
__device__ __forceinline__ float rho_prime_t(float t2, float lam2) {
  if constexpr (RT == ICA_TRUNCATED_QUADRATIC) return t2 < lam2 ? 1.0f : 0.0f;
  else if constexpr (RT == ICA_GERMAN_MCCLURE) {
    const float s = lam2 + t2;
    return lam2 * __builtin_amdgcn_rcpf(s * s);
  } else if constexpr (RT == ICA_LORENTZIAN) return __builtin_amdgcn_rcpf(lam2 + t2);
  else if constexpr (RT == ICA_CHARBONNIER) return __builtin_amdgcn_rsqf(t2 + lam2);
  else return 1.0f;
}

// U8: the level's images are uint8 in HBM (ICA_STORAGE_U8, level 0 only).  The copies then move BYTES into
// a staging area of the LDS (a quarter of the traffic: same lane -> (row, 4-pixel group) map, 4 bytes per lane instead
// of 16), and each chunk is expanded once, 4 pixels per thread and step (one ds_read_b32, four v_cvt_f32_ubyte, one
// ds_write_b128), into the float tile the bands read -- the bands, the masks and every sum are the float path's, so
// on the same (uint8-valued) images the two storages give bit-identical results.  Converting inside the tap loop
// instead (4 taps per dword) would put ~20 more vector instructions on every pixel of a loop that is bound by
// vector-instruction issue; expanding costs ~3 per pixel.  One float tile + one staging area per
// workgroup: barrier, expand chunk k, barrier, start the copy of chunk k+1, compute chunk k.
template <int TT, int MODE, int C, int RT, bool U8>
// (second launch bound = waves per SIMD the register allocator must leave room for: what the LDS footprint allows)
__global__ __launch_bounds__((IterCfg<C, U8>::block), (IterCfg<C, U8>::loader ? (IterCfg<C, U8>::waves * 4 + 3) / 4 : C == 1 ? (U8 ? ICA_ITER_WAVES_U8 : ICA_DUP ? 5 : ICA_ITER_WAVES) : U8 ? 3 : 4)) void ica_iterate_kernel(LevelArgs a) {
  using WG = IterCfg<C, U8>;
  constexpr bool kLoader = WG::loader;
  constexpr int kIterWaves = WG::waves, kIterBlock = WG::block, kCopyWaves = WG::copy_waves, kComputeWaves = WG::cw;
  using TIn = std::conditional_t<U8, unsigned char, float>;
  constexpr int ES = (int)sizeof(TIn);    // bytes per image element in HBM
  constexpr int N = NP<TT>::n;
  constexpr int NT = N * (N + 1) / 2;
  constexpr bool kB = (MODE != kModeHessian);
  constexpr bool kH = (MODE != kModeL2);
  constexpr bool kProj = (TT == ICA_HOMOGRAPHY);
  constexpr bool kRobust = (MODE == kModeRobust);
  using ML = MomLayout<TT>;
  constexpr bool kDup = (ICA_DUP != 0) && C == 1 && !U8 && !kLoader;
  using TC = TileCfg<C, kLoader, kDup>;
  constexpr int kI2Copies = kDup ? 2 : 1;   // footprint copies in front of the I1 tile
  constexpr int CH = TC::ch, RW = TC::rw, P2 = TC::p2, R2 = TC::r2, P1 = TC::p1, R1 = TC::r1;
  constexpr bool kSeq = TC::seq;                          // one channel in the tile at a time (see TileCfg)
  constexpr int LC = TC::lc, kPasses = kSeq ? C : 1;
  constexpr int NBUF = U8 ? 1 : TC::nbuf;                 // float tiles
  constexpr int kStageFloats = U8 ? TC::buf_floats / 4 : 0;  // uint8 staging area, in floats (4 pixels each)
  static_assert(TC::buf_floats % 16 == 0 && (LC * P2 * R2) % 4 == 0, "tile regions must be whole 16-byte groups");
  constexpr int NMOM = ML::nmom;
  constexpr unsigned kPitchBits = P2 == 80 ? 0x43a00000u : P2 == 96 ? 0x43c00000u : 0u;  // (float)(P2 * 4)
  static_assert(kPitchBits != 0, "add the bit pattern of (float)(P2 * 4)");

  // The ring buffers; the end-of-tile fold reuses the same LDS.  The fold works on the moments in two halves when a
  // single pass would need more LDS than the ring (robust homography), so that it never costs an occupancy step.
  constexpr int kRingFloats = NBUF * TC::buf_floats + kStageFloats;
  constexpr int kFoldAll = 2 * (NMOM + N + NT) * 64;
  constexpr int kLdsFloats = kRingFloats > 2 * (NMOM + (N + NT + 1) / 2) * 64 ? kRingFloats : 2 * (NMOM + (N + NT + 1) / 2) * 64;
  constexpr bool kFoldSplit = kFoldAll > kLdsFloats;
  __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];

  // XCD-aware block -> (pair, tile) map.  Workgroups are dealt round-robin to the 8 XCDs (blocks b and b+8 share
  // an L2), so consecutive block ids take different PAIRS and the blocks that share an XCD walk through the tiles
  // of the same pair: tile halos and the rows two chunks share are then served by that XCD's L2.  (Placement is a
  // speed matter only; nothing depends on it for correctness.)
  // With few pairs in the launch that map leaves XCDs idle or unevenly loaded (a single straggler pair would run on
  // one eighth of the chip): then the tiles of a pair are simply dealt to consecutive workgroups, i.e. round-robin
  // over all XCDs (those launches are latency-bound, not traffic-bound).
  const int ntiles = a.tiles_x * a.tiles_y;
  const bool flat = a.n_pairs < kFlatMapBelow;
  const int slot = flat ? (int)blockIdx.x : (int)(blockIdx.x >> 3), xcd = blockIdx.x & 7;
  const int pair_slot = flat ? slot / ntiles : (slot / ntiles) * 8 + xcd;
  const int tile_id = slot - (slot / ntiles) * ntiles;
  int pair = pair_slot;
  if (a.act_list) {
    // tell the host how many pairs entered this iteration (it sizes the next launch from it, without a sync)
    if (a.host_slot && blockIdx.x == 0 && threadIdx.x == 0)
      __hip_atomic_store(a.host_slot, list_total(a.act_count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (pair_slot >= list_total(a.act_count)) return;
    pair = list_entry(a.act_list, a.act_count, a.list_cap, pair_slot);
  } else if (pair_slot >= a.n_pairs) {
    return;
  }
#ifdef ICA_STAMP
  long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long stamp_begin = clock64(), stamp_rt_begin = wall_clock64();
#endif
  // Profiling: the workgroup in the middle of the grid notes when it starts (in memory: no register lives across the
  // kernel for it) and, at the end, adds its lifetime in shader cycles and in ticks of the constant 100 MHz counter to
  // the context's accumulators -- the clock the chip sustains under THIS kernel, measured without a second stream.
  const bool clock_wg = a.clock_acc != nullptr && blockIdx.x == (gridDim.x >> 1);
  if (clock_wg && threadIdx.x == 0) {
    a.clock_acc[2] = (unsigned long long)clock64();
    a.clock_acc[3] = (unsigned long long)wall_clock64();
  }
  const int tx = tile_id % a.tiles_x, ty = tile_id / a.tiles_x;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int H = a.H, W = a.W;
  const size_t plane = (size_t)H * W;
  const TIn* I1 = reinterpret_cast<const TIn*>(a.I1) + (size_t)pair * C * plane;
  const TIn* I2 = reinterpret_cast<const TIn*>(a.I2) + (size_t)pair * C * plane;

  float A[8];
  float lo = 0.f, hi = 0.f, lam2 = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) A[i] = 0.f;
  bool ident = true;  // all |p_i| < 1e-10: the reference warps with the identity (bicubic_interpolation.py:173-175)
  if constexpr (kB) {
    double ad[8];
    params_to_displacement(a.p + (size_t)pair * 8, TT, ad);
    // (params_to_displacement returns all zeros exactly when it took the identity shortcut)
#pragma unroll
    for (int i = 0; i < 8; ++i) ident = ident & (ad[i] == 0.0);
#pragma unroll
    for (int i = 0; i < 8; ++i) A[i] = (float)ad[i];
    lo = a.minmax[pair * 2];
    hi = a.minmax[pair * 2 + 1];
    if constexpr (kRobust) {
      const float l = (float)a.lambda_it[pair];
      lam2 = l * l;
    }
  }

  const int x_t0 = tx * kTileW;
  const int x = x_t0 + lane;
  const float xf = (float)x;
  // (Dealing the chunks of two vertically adjacent tiles alternately to their two workgroups, so that the 9 footprint rows
  //  consecutive chunks share are asked for at about the same time, was built and measured: HBM traffic UP, 1.18x instead
  //  of 1.12x the algorithmic bytes for one channel, 1.78x instead of 1.61x for three, times unchanged --
  //  profiles/r03_ablation.txt.)
  const int y_begin = ty * a.tile_h;
  const int y_end = min(H, y_begin + a.tile_h);
  const int nchunks = (y_end - y_begin + CH - 1) / CH;  // <= 32: one lane plans one chunk (tile_h <= 32 * CH, host-checked)
  const bool frame_on = a.nanif && a.delta > 0;
  const bool col_ok = (x < W) & !(frame_on & ((x < a.delta) | (x >= W - a.delta)));
  const float cmx = ((x >= 1) & (x <= W - 2)) ? 1.0f : 0.0f;  // ref :81: the 1-px border keeps a zero x-derivative
  // (the per-column terms of the displacement -- cx = a0 x + a2, cy = a3 x + a5, gxc = a6 x -- are formed where a band
  //  starts, from xf: three registers that do not have to live across the chunk loop)
  const int xm1 = x - 1;
  const unsigned ux_max = (unsigned)(W - 4), uy_max = (unsigned)(H - 4);
  const bool can_sample = (W >= 4) & (H >= 4);  // else the unsigned range tests below would wrap
  const bool w4 = (W & 3) == 0;                         // rows start on 4-pixel boundaries -> whole-group LDS-DMA

  // ---- plan: lane c (< 32) computes the I2 footprint of chunk c: x_lo | flags (bit 30 = stage through LDS, bit 29 =
  //      skip, bit 28 = interior: no frame / border / invalid pixel in the chunk, bit 27 = every pixel outside the
  //      NaN frame has its footprint inside I2: frame pixels only need masking), which stays in lane c, and
  //      y_lo | rows << 20, which moves to lane 32 + c of the same register (a tile has at most 32 chunks) ----
  int plan = 0;
  int plan_x = 0, plan_y = 0;
  {
    const int yc0 = y_begin + lane * CH;
    const int rows = min(CH, y_end - yc0);
    const bool skip = (lane >= nchunks) | (frame_on & ((yc0 + rows <= a.delta) | (yc0 >= H - a.delta)));
    // pixels that take part with weight 1 and a full gradient stencil: outside the frame (or, without a frame, off
    // the 1-px border where the reference keeps a zero derivative)
    const int mrg = frame_on ? a.delta : 1;
    bool interior = (rows == CH) & (yc0 >= mrg) & (yc0 + CH <= H - mrg) & (x_t0 >= mrg) & (x_t0 + kTileW <= W - mrg);
    bool stage = false, semi = false;
    // (wave-uniform: a tile that does not touch the NaN frame has no border chunk, and skips the second corner test)
    const bool tile_on_border = (x_t0 < mrg) | (x_t0 + kTileW > W - mrg) | (y_begin < mrg) | (y_end > H - mrg);
    int x_lo = 0, y_lo = 0;
    if constexpr (kB) {
      // the image of the chunk rectangle under an affine / projective map is the convex quadrilateral of its
      // mapped corners (denominator checked positive), so the corner positions bound every sampling position
      const float xl = (float)x_t0, xr = (float)min(x_t0 + kTileW - 1, W - 1);
      const float yt = (float)yc0, yb = (float)(yc0 + rows - 1);
      float dxs[4], dys[4];
      displacement<kProj>(A, xl, yt, dxs[0], dys[0]);
      displacement<kProj>(A, xr, yt, dxs[1], dys[1]);
      displacement<kProj>(A, xl, yb, dxs[2], dys[2]);
      displacement<kProj>(A, xr, yb, dxs[3], dys[3]);
      const float px0 = fminf(fminf(xl + dxs[0], xr + dxs[1]), fminf(xl + dxs[2], xr + dxs[3]));
      const float px1 = fmaxf(fmaxf(xl + dxs[0], xr + dxs[1]), fmaxf(xl + dxs[2], xr + dxs[3]));
      const float py0 = fminf(fminf(yt + dys[0], yt + dys[1]), fminf(yb + dys[2], yb + dys[3]));
      const float py1 = fmaxf(fmaxf(yt + dys[0], yt + dys[1]), fmaxf(yb + dys[2], yb + dys[3]));
      bool sane = (fabsf(px0) < 1.0e6f) & (fabsf(px1) < 1.0e6f) & (fabsf(py0) < 1.0e6f) & (fabsf(py1) < 1.0e6f);
      if constexpr (kProj) {
        const float g0 = fmaf(A[6], xl, A[7] * yt), g1 = fmaf(A[6], xr, A[7] * yt);
        const float g2 = fmaf(A[6], xl, A[7] * yb), g3 = fmaf(A[6], xr, A[7] * yb);
        sane = sane & (fminf(fminf(g0, g1), fminf(g2, g3)) > -0.5f);
      }
      if (sane) {
        // taps of a pixel sampled at X: columns floor(X)-1 .. floor(X)+2; one more pixel of slack for fp32
        // rounding; x_lo is rounded down to a multiple of 4 so that 16-byte copies stay aligned
        x_lo = max(0, (int)floorf(px0) - 2) & ~3;
        y_lo = max(0, (int)floorf(py0) - 2);
        const int x_hi = min(W - 1, (int)floorf(px1) + 3);
        const int y_hi = min(H - 1, (int)floorf(py1) + 3);
        const int bw = x_hi - x_lo + 1, bh = y_hi - y_lo + 1;
        stage = (bw >= 1) & (bh >= 1) & (bw <= P2) & (bh <= R2);
        plan_y = bh << 20;
        // every footprint inside the image with a pixel to spare: floor(X) - 1 >= 1 and floor(X) + 2 <= W - 2
        interior = interior & (px0 >= 2.0f) & (px1 <= (float)(W - 4)) & (py0 >= 2.0f) & (py1 <= (float)(H - 4));
        // the same test for the LIVE pixels of the chunk only (those outside the NaN frame): what a border chunk needs
        if (frame_on & !skip & tile_on_border) {
          const float lxl = (float)max(x_t0, a.delta), lxr = (float)min(x_t0 + kTileW - 1, W - 1 - a.delta);
          const float lyt = (float)max(yc0, a.delta), lyb = (float)min(yc0 + rows - 1, H - 1 - a.delta);
          if (lxl <= lxr && lyt <= lyb) {
            float ex[4], ey[4];
            displacement<kProj>(A, lxl, lyt, ex[0], ey[0]);
            displacement<kProj>(A, lxr, lyt, ex[1], ey[1]);
            displacement<kProj>(A, lxl, lyb, ex[2], ey[2]);
            displacement<kProj>(A, lxr, lyb, ex[3], ey[3]);
            const float qx0 = fminf(fminf(lxl + ex[0], lxr + ex[1]), fminf(lxl + ex[2], lxr + ex[3]));
            const float qx1 = fmaxf(fmaxf(lxl + ex[0], lxr + ex[1]), fmaxf(lxl + ex[2], lxr + ex[3]));
            const float qy0 = fminf(fminf(lyt + ey[0], lyt + ey[1]), fminf(lyb + ey[2], lyb + ey[3]));
            const float qy1 = fmaxf(fmaxf(lyt + ey[0], lyt + ey[1]), fmaxf(lyb + ey[2], lyb + ey[3]));
            semi = (qx0 >= 2.0f) & (qx1 <= (float)(W - 4)) & (qy0 >= 2.0f) & (qy1 <= (float)(H - 4));
          } else {
            semi = true;  // no live pixel in this lane's chunk at all
          }
        }
      }
      interior = interior & sane & stage;
      semi = semi & sane & stage;
    } else {
      semi = frame_on;  // no warp in this mode: the frame mask is all a border chunk needs
    }
    plan_x = x_lo | (stage ? (1 << 30) : 0) | (skip ? (1 << 29) : 0) | (interior ? (1 << 28) : 0) | (semi ? (1 << 27) : 0);
    plan_y |= y_lo;
    const int up = __shfl(plan_y, lane & 31, 64);  // lane 32 + c takes chunk c's word
    plan = lane < 32 ? plan_x : up;
  }
  // Chunks by instruction stream (see the bands below): bit c of a mask = chunk c.  Kernels that accumulate the Hessian
  // walk the tile class by class -- interior chunks, then masked-border ones, then the rest -- so that each loop contains
  // ONE band: with two bands behind a branch in one loop the register allocator keeps the running sums in a second set of
  // registers inside the straight-line band and copies them back at the back edge (17 + 17 instructions per band).
  unsigned cls_mask[3];
  {
    const bool mine = (lane < nchunks) & (lane < 32) & !(plan_x & (1 << 29));
    const bool st = (plan_x & (1 << 30)) != 0, in = (plan_x & (1 << 28)) != 0, se = (plan_x & (1 << 27)) != 0;
#ifdef ICA_NO_SEMI
    constexpr bool kSemi = false;
#else
    // (one channel only: with three, the extra code costs the kernel the registers of an occupancy step -- measured 10 %
    //  slower on configs[4])
    constexpr bool kSemi = (C == 1);
#endif
    cls_mask[0] = (unsigned)__ballot(mine & st & in);
    cls_mask[1] = (unsigned)__ballot(mine & st & !in & se & kSemi);
    cls_mask[2] = (unsigned)__ballot(mine) & ~(cls_mask[0] | cls_mask[1]);
#if defined(ICA_PROBE) && ICA_PROBE == 1   // register-pressure probe: fast band only
    cls_mask[0] |= cls_mask[1] | cls_mask[2]; cls_mask[1] = cls_mask[2] = 0;
#elif defined(ICA_PROBE) && ICA_PROBE == 2  // careful band only
    cls_mask[2] |= cls_mask[0] | cls_mask[1]; cls_mask[0] = cls_mask[1] = 0;
#endif
  }
  auto plan_px = [&](int k) { return __builtin_amdgcn_readlane(plan, k); };
  auto plan_ylo = [&](int k) { return __builtin_amdgcn_readlane(plan, k + 32) & 0xFFFFF; };
  auto plan_bh = [&](int k) { return __builtin_amdgcn_readlane(plan, k + 32) >> 20; };

  // ---- LDS-DMA plumbing.  Buffer descriptors give hardware range checking (rows above / below the image are
  // simply dropped, the LDS slot gets 0) and 32-bit offsets.  One wave-instruction moves RP whole tile rows: lane l
  // carries float4 number (l % V) of row (l / V) of the group, V = float4 per tile row, so the only per-lane state is
  // ONE byte offset per image (off1, off2), which does not depend on the chunk; everything else about a copy (first
  // row, row shift, how many rows) is wave-uniform.  The tile's row groups are dealt round-robin to the 4 waves.
  const auto rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)I1, 0, (int)(C * plane * ES), 0x00020000);
  constexpr int V1 = P1 / 4, V2 = P2 / 4;                       // float4 per buffer row
  constexpr int RP1 = 64 / V1, RP2 = 64 / V2;                   // rows per wave-instruction
  constexpr int NQ1 = (R1 + RP1 - 1) / RP1, NQ2 = (R2 + RP2 - 1) / RP2;  // row groups per tile
  constexpr unsigned kPoison = 0x80000000u;                     // offset that is always out of range
  unsigned off1, off2;
  {
    const int r1 = lane / V1, c1 = lane - r1 * V1, r2 = lane / V2, c2 = lane - r2 * V2;
    const int xx = x_t0 - 4 + 4 * c1;
    off1 = (xx >= 0 && xx < W) ? (unsigned)((r1 * W + 4 * c1 - 4) * ES) : kPoison;
    off2 = (unsigned)((r2 * W + 4 * c2) * ES);
  }

  // issue the LDS-DMA copies of chunk k into buffer `buf` (no waits here).
  // (Fetching only the rows a chunk does NOT share with its predecessor and moving the shared ones LDS -> LDS -- 9 of
  //  the 25 I2 rows, 2 of the 18 I1 rows -- was built and measured: a quarter fewer L1 requests, but 9 % MORE time;
  //  the move and its dependency sit in front of the copies.  profiles/r02_ablation.txt)
  const bool copies = !kLoader || wave >= kComputeWaves;   // (with loader waves: they alone copy)
  auto issue_impl = [&](int k, int buf, int pass) {   // pass: the channel to copy (channel-sequential tiles only)
#if defined(ICA_ABLATE) && (ICA_ABLATE & 2)   // (ICA_ABLATE is a set of bits: 1 copies only, 2 no copies, 4 lane-linear taps, 8 no barriers, 16 half the tap reads, 64 no Hessian moments, 128 no I1 copies, 256 no I2 copies)
    return;  // timing-only build: no LDS-DMA
#endif
    const int px = plan_px(k), y_lo = plan_ylo(k);
    // destination: the float tile `buf`, or (U8) the staging area behind the tile, laid out like the tile, in bytes
    TIn* b2 = U8 ? reinterpret_cast<TIn*>(lds + NBUF * TC::buf_floats) : reinterpret_cast<TIn*>(lds + buf * TC::buf_floats);
    TIn* b1 = b2 + LC * kI2Copies * P2 * R2;
    const int yc0 = y_begin + k * CH;
    [[maybe_unused]] const unsigned pass_off = kSeq ? (unsigned)((size_t)pass * plane * ES) : 0u;
    if (w4) {
      // a skipped (all-frame) chunk still issues its copies, with every lane out of range (the tile is zero-filled)
      const bool skip = (px & (1 << 29)) != 0;
      // may wrap below 0 (row -1): then out of range.  A skipped chunk gets an out-of-range base.
      const unsigned base1 = skip ? 0x40000000u : (unsigned)(((yc0 - 1) * W + x_t0) * ES);
#if defined(ICA_ABLATE) && (ICA_ABLATE & 128)   // timing-only build: no I1 copies
      if (false)
#endif
#pragma unroll
      for (int qq = 0; qq < (NQ1 + kCopyWaves - 1) / kCopyWaves; ++qq) {
        const int q = (kLoader ? wave - kComputeWaves : wave) + qq * kCopyWaves;      // this wave's row group (wave-uniform)
        // Lanes past the tile's last row must not take part at all: an out-of-range lane still writes its (zero) result
        // to LDS, which would land in the neighbouring tile.
        const int rows = min(RP1, R1 - RP1 * q);
        if (q < NQ1 && lane < V1 * rows) {
          TIn* dst = b1 + RP1 * q * P1;
          unsigned u = base1 + (unsigned)(RP1 * q * W * ES);
          asm volatile("" : "+s"(u));  // (opaque: the sum with off1 is formed here, not hoisted out of the chunk loop into a register)
          const unsigned o = off1 + u;  // a poisoned offset stays out of range
#pragma unroll
          for (int c = 0; c < LC; ++c)
            ICA_LDS_DMA_AUX(U8, rs1, dst + c * P1 * R1, o + pass_off + (unsigned)(c * plane * ES), ICA_DMA_AUX1);
        }
      }
#if defined(ICA_ABLATE) && (ICA_ABLATE & 256)   // timing-only build: no I2 copies
      if constexpr (false) {
#else
      if constexpr (kB) {
#endif
        const bool stage = !skip && (px & (1 << 30));
        const int x_lo = px & 0xFFFFF;
        const int bh = stage ? plan_bh(k) : 0;
        // rows beyond the footprint fall outside this descriptor (a chunk that is not staged gets 0 rows) and are
        // zero-filled
        // (channel-sequential: the descriptor ends behind the footprint's last row of THIS channel)
        const auto rs2 = __builtin_amdgcn_make_buffer_rsrc(
            (void*)I2, 0, bh ? (int)(((size_t)(kSeq ? pass : C - 1) * plane + (size_t)(y_lo + bh) * W) * ES) : 0, 0x00020000);
        const unsigned base2 = (unsigned)((y_lo * W + x_lo) * ES);
#pragma unroll
        for (int qq = 0; qq < (NQ2 + kCopyWaves - 1) / kCopyWaves; ++qq) {
          const int q = (kLoader ? wave - kComputeWaves : wave) + qq * kCopyWaves;
          const int rows = min(RP2, R2 - RP2 * q);
          if (q < NQ2 && lane < V2 * rows) {
            TIn* dst = b2 + RP2 * q * P2;
            unsigned u = base2 + (unsigned)(RP2 * q * W * ES);
            asm volatile("" : "+s"(u));
            const unsigned o = off2 + u;  // all P2 columns of a row are copied (a few more than needed)
#pragma unroll
            for (int c = 0; c < LC; ++c)
              ICA_LDS_DMA_AUX(U8, rs2, dst + c * P2 * R2, o + pass_off + (unsigned)(c * plane * ES), ICA_DMA_AUX2);
            // second copy, one float to the left (B[j] = A[j + 1]): an odd tap address in A is an even one in B
            if constexpr (kDup) ICA_LDS_DMA(U8, rs2, dst + P2 * R2, o + 4u);
          }
        }
      }
    } else if constexpr (!U8) {
      if (px & (1 << 29)) return;
      for (int r = (kLoader ? wave - kComputeWaves : wave); r < R1; r += kCopyWaves) {
        const int yy = yc0 - 1 + r;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int c = 0; c < LC; ++c) {
          const float* src = I1 + (size_t)(kSeq ? pass : c) * plane + (size_t)yy * W + (x_t0 - 4);
          float* dst = b1 + c * P1 * R1 + r * P1;
          if ((x_t0 - 4 + lane >= 0) & (x_t0 - 4 + lane < W)) glds4(src + lane, dst);
          if ((lane < P1 - 64) & (x_t0 + 60 + lane < W)) glds4(src + 64 + lane, dst + 64);
        }
      }
      if (kB && (px & (1 << 30))) {
        const int x_lo = px & 0xFFFFF;
        for (int r = (kLoader ? wave - kComputeWaves : wave); r < R2; r += kCopyWaves) {
          const int yy = y_lo + r;
          if (yy >= H) break;
#pragma unroll
          for (int c = 0; c < LC; ++c) {
            const float* src = I2 + (size_t)(kSeq ? pass : c) * plane + (size_t)yy * W + x_lo;
            float* dst = b2 + c * P2 * R2 + r * P2;
            if (x_lo + lane < W) glds4(src + lane, dst);
            if ((lane < P2 - 64) & (x_lo + 64 + lane < W)) glds4(src + 64 + lane, dst + 64);
            if constexpr (kDup) {
              if (x_lo + 1 + lane < W) glds4(src + 1 + lane, dst + P2 * R2);
              if ((lane < P2 - 64) & (x_lo + 65 + lane < W)) glds4(src + 65 + lane, dst + P2 * R2 + 64);
            }
          }
        }
      }
    }
  };

  auto issue = [&](int k, int buf, int pass = 0) {
    if (copies) issue_impl(k, buf, pass);
  };
  if (!w4) {  // the dword copy path skips out-of-image lanes instead of zero-filling them: start from finite LDS
    for (int i = threadIdx.x; i < kLdsFloats; i += kIterBlock) lds[i] = 0.f;
    __syncthreads();
  }

  float m32[NMOM];
#pragma unroll
  for (int i = 0; i < NMOM; ++i) m32[i] = 0.f;

  // wave-uniform values live in SGPRs between the bands (the fast band copies what it needs into VGPRs)
#pragma unroll
  for (int i = 0; i < 8; ++i) A[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, A[i])));
  lo = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, lo)));
  hi = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, hi)));
  lam2 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, lam2)));

  static_assert(NBUF == 1 || NBUF == 2, "one or two staging buffers");
  // U8: staging area -> float tile, 4 pixels per thread and step; rows whose width is not a multiple of 4 pixels (no
  // whole-group copies) are read straight from memory, element by element (cold path: odd widths at level 0)
  [[maybe_unused]] auto expand = [&](int k, [[maybe_unused]] int pass) {
    if (w4) {
      const unsigned* s32 = reinterpret_cast<const unsigned*>(lds + TC::buf_floats);
      for (int e = threadIdx.x; e < TC::buf_floats / 4; e += kIterBlock) {
        const unsigned u = s32[e];
        float4 v;
        v.x = (float)(u & 0xffu); v.y = (float)((u >> 8) & 0xffu); v.z = (float)((u >> 16) & 0xffu); v.w = (float)(u >> 24);
        *reinterpret_cast<float4*>(lds + 4 * e) = v;
      }
    } else {
      const int px = plan_px(k), y_lo = plan_ylo(k), x_lo = px & 0xFFFFF, yc0 = y_begin + k * CH;
      for (int e = threadIdx.x; e < TC::buf_floats; e += kIterBlock) {
        const bool two = e < LC * P2 * R2;                 // I2 footprint region, else I1 tile region
        const int f = two ? e : e - LC * P2 * R2, pitch = two ? P2 : P1, rows = two ? R2 : R1;
        const int cl = f / (pitch * rows), g = f - cl * pitch * rows, r = g / pitch, col = g - r * pitch;
        const int c = kSeq ? pass : cl;                    // image channel
        const int yy = two ? y_lo + r : yc0 - 1 + r, xx = two ? x_lo + col : x_t0 - 4 + col;
        const TIn* img = two ? I2 : I1;
        const bool in = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W) & (two ? (kB && (px & (1 << 30)) != 0) : true);
        lds[e] = in ? (float)img[c * plane + (size_t)yy * W + xx] : 0.0f;
      }
    }
  };
  // Walking order.  Class by class (kH: 12-17 running sums, the copies of a one-loop walk cost 3 % of the kernel), or
  // top to bottom in one loop (b only: a handful of sums, nothing to gain from separate loops -- and the out-of-order
  // walk fetches more from HBM: neighbouring chunks share 9 of their 25 footprint rows and the column tiles next door
  // walk the same rows at the same time; measured 1.26x against 1.06x the algorithmic bytes for affine L2 at 512^2,
  // 1.14x against 1.08x for the robust homography kernel, which takes the trade for its 3 %).
#ifdef ICA_WALK
  constexpr int kWalk = ICA_WALK;
#else
  constexpr int kWalk = kH ? 0 : 1;
#endif
  // first chunk of class >= c that is still to do, -1 if none (all wave-uniform)
  auto first_from = [&](int c) -> int {
    for (; c < 3; ++c)
      if (cls_mask[c]) return __builtin_ctz(cls_mask[c]);
    return -1;
  };
#ifdef ICA_STAMP
  stamp[4] = clock64() - stamp_begin;
#endif
  int ring = 0;  // the float tile the next computed chunk sits in (two tiles: they alternate)
  {
    const unsigned all = cls_mask[0] | cls_mask[1] | cls_mask[2];
    const int k0 = kWalk == 1 ? (all ? __builtin_ctz(all) : -1) : first_from(0);
    if (k0 >= 0 && ((U8 && w4 && !kSeq) || (!U8 && NBUF == 2))) issue(k0, 0);
  }
  // the lane's y-moments from a pixel's channel sums (what accumulate() does behind its channel loops)
  [[maybe_unused]] auto finish_seq = [&](float w, const float* s6, float yf) {
    [[maybe_unused]] const float y2 = yf * yf;
    if constexpr (kB) {
      float gx = s6[0], gy = s6[1];
      if constexpr (kH) { gx *= w; gy *= w; }
      m32[ML::ogx] += gx;
      if constexpr (ML::mbx >= 1) m32[ML::ogx + 1] = fmaf(gx, yf, m32[ML::ogx + 1]);
      m32[ML::ogy] += gy;
      if constexpr (ML::mby >= 1) m32[ML::ogy + 1] = fmaf(gy, yf, m32[ML::ogy + 1]);
      if constexpr (ML::mby >= 2) m32[ML::ogy + 2] = fmaf(gy, y2, m32[ML::ogy + 2]);
    }
    if constexpr (kH) {
      const float sxx = s6[2] * w, sxy = s6[3] * w, syy = s6[4] * w;
      float yp[5];
      yp[0] = 1.0f; yp[1] = yf; yp[2] = y2; yp[3] = y2 * yf; yp[4] = y2 * y2;
#pragma unroll
      for (int b = 0; b < 5; ++b) {
        if (b < ML::nxx) m32[ML::oxx + b] = (b == 0) ? m32[ML::oxx] + sxx : fmaf(sxx, yp[b], m32[ML::oxx + b]);
        if (b < ML::nxy) m32[ML::oxy + b] = (b == 0) ? m32[ML::oxy] + sxy : fmaf(sxy, yp[b], m32[ML::oxy + b]);
        if (b < ML::nyy) m32[ML::oyy + b] = (b == 0) ? m32[ML::oyy] + syy : fmaf(syy, yp[b], m32[ML::oyy + b]);
      }
    }
  };
  // one chunk: ring protocol, then the band of its class.  `next` = the chunk that follows in walking order (-1: none).
  auto step = [&](auto cls_c, const int k, const int next) {
    constexpr int CLS = decltype(cls_c)::value;
    // channel-sequential tiles: the sums over the channels of each pixel of this wave's band, carried from pass to pass
    // ([0] gx = sum Ix DI, [1] gy, [2] sxx = sum Ix^2, [3] sxy, [4] syy, [5] t^2 = sum DI^2; gradients doubled as elsewhere)
    [[maybe_unused]] float sg[kSeq ? RW : 1][6];
    if constexpr (kSeq) {
      // A careful chunk (border pixels whose footprint leaves I2, a partial chunk, a footprint too large to stage) is
      // walked ONCE, all channels per pixel, straight from memory through the cache: it needs every test of the reference's
      // pixel masks and the float64 re-evaluation, code that is kept small (rolled) and so cannot carry sums from pass to
      // pass in registers.  It touches neither tile nor staging area, so it needs no copy and no barrier.
      const bool careful = CLS == 2 || (CLS == 3 && !(((cls_mask[0] | cls_mask[1]) >> k) & 1));
      if (careful) {
        const int yb0 = y_begin + k * CH + wave * RW;
        [[maybe_unused]] const float cx = fmaf(A[0], xf, A[2]), cy = fmaf(A[3], xf, A[5]), gxc = A[6] * xf;
        const int xc = min(x, W - 1), xl = max(xc - 1, 0), xr = min(xc + 1, W - 1);
#pragma unroll 1
        for (int i = 0; i < RW; ++i) {
          const int y = yb0 + i;  // wave-uniform
          const bool row_live = (y < y_end) & !(frame_on & ((y < a.delta) | (y >= H - a.delta)));
          const float rmy = ((y >= 1) & (y <= H - 2)) ? 1.0f : 0.0f;  // ref :82: zero y-derivative on the border rows
          const bool live = col_ok & row_live;
          const float yf = (float)y;
          const int yc = min(y, H - 1), yu = max(yc - 1, 0), yd = min(yc + 1, H - 1);
          float wx[4], wy[4];
          [[maybe_unused]] bool valid = false;
          [[maybe_unused]] size_t toff = 0;
          if constexpr (kB) {
            float dxw, dyw;
            displacement_col<kProj>(A, cx, cy, gxc, xf, yf, dxw, dyw);
            float flx = floorf(dxw), fly = floorf(dyw);
            float fx = dxw - flx, fy = dyw - fly;
            int ix = (int)flx, iy = (int)fly;
            const bool near_x = live & (fminf(fx, 1.0f - fx) < kNearInteger), near_y = live & (fminf(fy, 1.0f - fy) < kNearInteger);
#ifndef ICA_NO_EXACT
            if (near_x) exact_axis<TT, 0>(a.p + (size_t)pair * 8, ident, x, y, x, ix, fx);
            if (near_y) exact_axis<TT, 1>(a.p + (size_t)pair * 8, ident, x, y, y, iy, fy);
#endif
            const int ux = ix + xm1, uy = iy + (y - 1);
            valid = live & can_sample & ((unsigned)ux <= ux_max) & ((unsigned)uy <= uy_max);
            toff = valid ? (size_t)uy * W + ux : 0;
            keys_weights(fx, wx);
            keys_weights(fy, wy);
          }
          float s6[6];
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const TIn* i1c = I1 + (size_t)c * plane;
            // (clamped coordinates: where a neighbour does not exist its difference is masked, any finite value will do)
            const float mid = (float)i1c[(size_t)yc * W + xc];
            const float ix2 = ((float)i1c[(size_t)yc * W + xr] - (float)i1c[(size_t)yc * W + xl]) * cmx;
            const float iy2 = ((float)i1c[(size_t)yd * W + xc] - (float)i1c[(size_t)yu * W + xc]) * rmy;
            float di = 0.f;
            if constexpr (kB) {
              const TIn* q = I2 + (size_t)c * plane + toff;
              float v = 0.f;
              if (valid) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const TIn* rp = q + (size_t)r * W;
                  float s = wx[0] * (float)rp[0];
                  s = __builtin_fmaf(wx[1], (float)rp[1], s);
                  s = __builtin_fmaf(wx[2], (float)rp[2], s);
                  s = __builtin_fmaf(wx[3], (float)rp[3], s);
                  v = (r == 0) ? wy[0] * s : __builtin_fmaf(wy[r], s, v);
                }
              }
              const float d = __builtin_amdgcn_fmed3f(v, lo, hi) - mid;     // clip (skimage warp), ref :114
              di = (valid & (fabsf(d) <= 3.0e38f)) ? d : 0.0f;              // NaN / outside -> 0 (io.py:99)
            }
            if constexpr (!kH) di *= live ? 1.0f : 0.0f;   // unweighted sums: the mask goes into the residual
            if constexpr (kB) {
              s6[0] = (c == 0) ? ix2 * di : fmaf(ix2, di, s6[0]);
              s6[1] = (c == 0) ? iy2 * di : fmaf(iy2, di, s6[1]);
            }
            if constexpr (kH) {
              s6[2] = (c == 0) ? ix2 * ix2 : fmaf(ix2, ix2, s6[2]);
              s6[3] = (c == 0) ? ix2 * iy2 : fmaf(ix2, iy2, s6[3]);
              s6[4] = (c == 0) ? iy2 * iy2 : fmaf(iy2, iy2, s6[4]);
            }
            if constexpr (kRobust) s6[5] = (c == 0) ? di * di : fmaf(di, di, s6[5]);
          }
          float w = live ? 1.0f : 0.0f;  // pixels in the NaN frame contribute nothing (derivatives.py:84)
          if constexpr (kRobust) w = live ? rho_prime_t<RT>(s6[5], lam2) : 0.0f;  // rho'(0) where the warp is outside
          finish_seq(w, s6, yf);
        }
        return;
      }
    }
   auto one_pass = [&](auto pass_c) {
    constexpr int PC = decltype(pass_c)::value;   // the channel in the tile (channel-sequential), else 0
    const int buf = (NBUF >= 2) ? ring : 0;
    if constexpr (U8) {
      // (channel-sequential: a chunk's first channel is asked for here, not under the previous chunk -- that one may be
      //  a careful chunk, which runs without the tile; the staging area is free: its last user expanded it a band ago)
      if constexpr (kSeq && PC == 0) { if (w4) issue(k, 0, 0); }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // chunk k's bytes are in the staging area; everyone is done with the float tile
      expand(k, PC);
      __syncthreads();                // the float tile holds chunk k; the staging area is free
      if constexpr (kSeq) {
        if (w4 && PC + 1 < kPasses) issue(k, 0, PC + 1);   // the next channel's bytes, under this channel's band
      } else {
        if (w4 && next >= 0) issue(next, 0);
      }
    } else if (NBUF == 1) {
#ifdef ICA_STAMP
      const long long s0 = clock64();
#endif
#if !(defined(ICA_ABLATE) && (ICA_ABLATE & 8))
      __syncthreads();  // everyone is done reading the single buffer
#endif
#ifdef ICA_STAMP
      const long long s1 = clock64();
#endif
      issue(k, 0, PC);
#ifdef ICA_STAMP
      const long long s2 = clock64();
#endif
#if !(defined(ICA_ABLATE) && (ICA_ABLATE & 8))
      __syncthreads();  // drains this wave's LDS-DMA (vmcnt(0)) and publishes chunk k
#endif
#ifdef ICA_STAMP
      const long long s3 = clock64();
      stamp[0] += s1 - s0; stamp[1] += s2 - s1; stamp[2] += s3 - s2; stamp[3] -= s3;
#endif
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      ring ^= 1;
      if (next >= 0) issue(next, ring);
    }

    const int px = plan_px(k), y_lo = plan_ylo(k);
    [[maybe_unused]] const bool staged = (px & (1 << 30)) != 0;
    const int x_lo = px & 0xFFFFF;
    const float* b2 = lds + buf * TC::buf_floats;
    const float* b1 = b2 + LC * kI2Copies * P2 * R2;
    const int yc0 = y_begin + k * CH;
    const int yb0 = yc0 + wave * RW;
    const float* i1p = b1 + (wave * RW) * P1 + lane + 4;  // row yb0-1 of this lane's column

    // one pixel's contribution to the lane's y-moments: w = weight (1 for L2), ix2 / iy2 = 2*Ix, 2*Iy, di = residual.
    // `yf` must be a VGPR value (fast band) or may be anything (careful band).
    auto accumulate = [&](float w, const float* ix2, const float* iy2, const float* di, float yf) {
      // one channel and a weighted sum: fold the weight into the gradient once (w Ix, w Iy) instead of scaling
      // gx, gy, sxx, sxy, syy one by one
      constexpr bool kFoldW = (C == 1) && kH;
      float wix = 0.f, wiy = 0.f;
      if constexpr (kFoldW) { wix = w * ix2[0]; wiy = w * iy2[0]; }
      [[maybe_unused]] const float y2 = yf * yf;
      if constexpr (kB) {
        float gx = 0.f, gy = 0.f;
        if constexpr (kFoldW) {
          gx = wix * di[0];
          gy = wiy * di[0];
        } else {
#pragma unroll
          for (int c = 0; c < C; ++c) {
            gx = (c == 0) ? ix2[0] * di[0] : fmaf(ix2[c], di[c], gx);
            gy = (c == 0) ? iy2[0] * di[0] : fmaf(iy2[c], di[c], gy);
          }
          if constexpr (kH) { gx *= w; gy *= w; }
        }
        m32[ML::ogx] += gx;
        if constexpr (ML::mbx >= 1) m32[ML::ogx + 1] = fmaf(gx, yf, m32[ML::ogx + 1]);
        m32[ML::ogy] += gy;
        if constexpr (ML::mby >= 1) m32[ML::ogy + 1] = fmaf(gy, yf, m32[ML::ogy + 1]);
        if constexpr (ML::mby >= 2) m32[ML::ogy + 2] = fmaf(gy, y2, m32[ML::ogy + 2]);
      }
#if defined(ICA_ABLATE) && (ICA_ABLATE & 64)   // timing-only build: no Hessian moments (15 vector instructions per pixel fewer)
      if constexpr (false) {
#else
      if constexpr (kH) {
#endif
        float sxx = 0.f, sxy = 0.f, syy = 0.f;
        if constexpr (kFoldW) {
          sxx = wix * ix2[0];
          sxy = wix * iy2[0];
          syy = wiy * iy2[0];
        } else {
#pragma unroll
          for (int c = 0; c < C; ++c) {
            sxx = (c == 0) ? ix2[0] * ix2[0] : fmaf(ix2[c], ix2[c], sxx);
            sxy = (c == 0) ? ix2[0] * iy2[0] : fmaf(ix2[c], iy2[c], sxy);
            syy = (c == 0) ? iy2[0] * iy2[0] : fmaf(iy2[c], iy2[c], syy);
          }
          sxx *= w; sxy *= w; syy *= w;
        }
        float yp[5];
        yp[0] = 1.0f; yp[1] = yf; yp[2] = y2; yp[3] = y2 * yf; yp[4] = y2 * y2;
#pragma unroll
        for (int b = 0; b < 5; ++b) {
          if (b < ML::nxx) m32[ML::oxx + b] = (b == 0) ? m32[ML::oxx] + sxx : fmaf(sxx, yp[b], m32[ML::oxx + b]);
          if (b < ML::nxy) m32[ML::oxy + b] = (b == 0) ? m32[ML::oxy] + sxy : fmaf(sxy, yp[b], m32[ML::oxy + b]);
          if (b < ML::nyy) m32[ML::oyy + b] = (b == 0) ? m32[ML::oyy] + syy : fmaf(syy, yp[b], m32[ML::oyy + b]);
        }
      }
    };

    // ---------------------------------------------------------------------------------------------------------
    // INTERIOR chunk: every pixel takes part, every footprint is inside I2 and inside the staged window.
    // MASKED: a border chunk whose pixels outside the NaN frame all have their footprint inside I2 -- the same
    // stream, with the frame pixels' residual and weight SELECTED to zero (v_cndmask, not a multiplication by a 0 / 1
    // mask: a frame pixel samples wherever its position points, possibly at bytes of the uint8 staging area, and the
    // bits found there may be a NaN).  Coarse pyramid levels consist of such chunks almost entirely.
    // ---------------------------------------------------------------------------------------------------------
    auto band_fast = [&](auto masked_c) {
      constexpr bool MASKED = decltype(masked_c)::value;
      // byte address of tap (0,0) of the pixel in row i  =  b2 + ((yb0 + i - 1 - y_lo) * P2 + (x - 1 - x_lo)) * 4
      //                                                     + (floor(dy) * P2 + floor(dx)) * 4;
      // the row part i * P2 * 4 is a compile-time LDS offset, the rest of the first line is computed once per chunk,
      // and the second line is formed in fp32 (small integers, exact) with two multiply-adds: the integer forms
      // (v_mul_i32_i24, v_mad) cost 2.7x a VOP2 on this chip
      const int lds_b2 = (int)(unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)b2;
      const float tap0f = (float)(lds_b2 + ((yb0 - 1 - y_lo) * P2 + (xm1 - x_lo)) * 4);
      const float yf0 = in_vgpr((float)yb0);
      [[maybe_unused]] const float cx = fmaf(A[0], xf, A[2]), cy = fmaf(A[3], xf, A[5]), gxc = A[6] * xf, nxf = -xf;
      // wave-uniform operands, copied into VGPRs here (not kept in VGPRs across the chunk loop: registers are short)
      [[maybe_unused]] const float a1v = pin_vgpr(A[1]), a4v = pin_vgpr(A[4]), a7v = pin_vgpr(A[7]);
#if ICA_POS_LINEAR
      [[maybe_unused]] const float pd0 = 1.0f + gxc, psx = fmaf(-xf, a7v, a1v), pix = fmaf(nxf, gxc, cx), psy = a4v - gxc,
                                   na7v = -a7v;
#endif
      [[maybe_unused]] const float hi_v = pin_vgpr(hi), lam2_v = pin_vgpr(lam2);
      // I1: a row's up / down neighbours are the centre values of the rows before / after it, so the column is read
      // once per row (rolling window of three values per channel)
      float cu[C], cm[C], cd[C];
#pragma unroll
      for (int c = 0; c < C; ++c) { cm[c] = i1p[c * P1 * R1]; cd[c] = i1p[c * P1 * R1 + P1]; }
      [[maybe_unused]] const unsigned i1a = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)i1p;
#ifdef ICA_PROBE_I1LOAD   // timing probe: what do per-lane global loads of the I1 row (left, centre, right) cost in the band?
      typedef int probe_v3 __attribute__((ext_vector_type(3)));
      probe_v3 pre[RW + 2];
      const int probe_off = ((yb0 - 1) * W + (x - 1)) * 4;
      pre[0] = __builtin_amdgcn_raw_buffer_load_b96(rs1, probe_off, 0, 0);
      pre[1] = __builtin_amdgcn_raw_buffer_load_b96(rs1, probe_off + W * 4, 0, 0);
#endif
#pragma unroll
      for (int i = 0; i < RW; ++i) {
        const float yf = i == 0 ? yf0 : yf0 + (float)i;  // exact (small integers); a VGPR because yf0 is one
#ifdef ICA_PROBE_I1LOAD
        pre[i + 2] = __builtin_amdgcn_raw_buffer_load_b96(rs1, probe_off + (i + 2) * W * 4, 0, 0);
        asm volatile("" ::"v"(pre[i].x), "v"(pre[i].y), "v"(pre[i].z));
#endif
        [[maybe_unused]] bool live = true;
        if constexpr (MASKED) {
          const int y = yb0 + i;  // wave-uniform
          live = col_ok & (y < y_end) & !((y < a.delta) | (y >= H - a.delta));
        }
        float ix2[C], iy2[C];
        constexpr bool kBatch = (ICA_TAP_BATCH != 0) && kB && (C == 1);
        [[maybe_unused]] float i1dn = 0.f, i1lf = 0.f, i1rt = 0.f;
        if constexpr (kBatch) {
          // (the I1 reads are requested first: they do not depend on the sampling position)
          switch (i) {   // the row offset must be an immediate
            case 0: lds_read_i1<1 * P1 * 4, P1>(i1a, i1dn, i1lf, i1rt); break;
            case 1: lds_read_i1<2 * P1 * 4, P1>(i1a, i1dn, i1lf, i1rt); break;
            case 2: lds_read_i1<3 * P1 * 4, P1>(i1a, i1dn, i1lf, i1rt); break;
            case 3: lds_read_i1<4 * P1 * 4, P1>(i1a, i1dn, i1lf, i1rt); break;
            case 4: lds_read_i1<5 * P1 * 4, P1>(i1a, i1dn, i1lf, i1rt); break;
            case 5: lds_read_i1<6 * P1 * 4, P1>(i1a, i1dn, i1lf, i1rt); break;
            case 6: lds_read_i1<7 * P1 * 4, P1>(i1a, i1dn, i1lf, i1rt); break;
            default: lds_read_i1<8 * P1 * 4, P1>(i1a, i1dn, i1lf, i1rt); break;
          }
        } else {
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const float* q = i1p + c * P1 * R1 + (i + 1) * P1;
            cu[c] = cm[c]; cm[c] = cd[c]; cd[c] = q[P1];
            ix2[c] = q[1] - q[-1];
            iy2[c] = cd[c] - cu[c];
          }
        }
        float di[C];
#pragma unroll
        for (int c = 0; c < C; ++c) di[c] = 0.f;
        if constexpr (kB) {
          float dxw, dyw;
          if constexpr (kProj) {
#if ICA_POS_LINEAR
            // numerators and denominator of the displacement as polynomials in y with per-column coefficients (formed
            // once per band): x' - x = ((a1 - a7 x) y + (cx - x gxc)) / d,  y' - y = (-a7 y^2 + (a4 - gxc) y + cy) / d,
            // d = a7 y + (1 + gxc): 5 multiply-adds and a reciprocal per pixel instead of 7 and an addition
            const float d = fmaf(a7v, yf, pd0);
            const float r = rcp_nr(d);
            dxw = fmaf(psx, yf, pix) * r;
            dyw = fmaf(na7v, yf * yf, fmaf(psy, yf, cy)) * r;
#else
            dxw = fmaf(a1v, yf, cx); dyw = fmaf(a4v, yf, cy);
            const float g = fmaf(a7v, yf, gxc);
            const float r = rcp_nr(1.0f + g);
            dxw = fmaf(nxf, g, dxw) * r;
            dyw = fmaf(-yf, g, dyw) * r;
#endif
          } else {
            dxw = fmaf(a1v, yf, cx); dyw = fmaf(a4v, yf, cy);
          }
          const float flx = floorf(dxw), fly = floorf(dyw);
          float fx = dxw - flx, fy = dyw - fly;
          float af;  // fly * (P2 * 4) + tap0f with the pitch as a literal (no register for it)
          asm("v_fmamk_f32 %0, %1, %3, %2" : "=v"(af) : "v"(fly), "v"(tap0f), "n"(kPitchBits));
          af = fmaf(flx, 4.0f, af);
          // (signed conversion: in a border chunk the address of row 0 of the band may lie below the tile -- a frame
          //  row above the image -- while a later, live row of the same band is inside)
#if defined(ICA_ABLATE) && (ICA_ABLATE & 4)  // timing-only build: every tap read is lane-linear (no LDS bank conflicts)
          int zero_s;
          asm volatile("s_mov_b32 %0, 0" : "=s"(zero_s));
          const int lane_b = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) * 4;
          const unsigned addr = (unsigned)(((int)af & zero_s) + lane_b + (lds_b2 + i * P2 * 4));
#else
          unsigned addr = (unsigned)((int)af + i * P2 * 4);
          // (opaque: folded into the read offsets, the row part overflows ds_read2's 8-bit offsets for i >= 1 and
          //  every pair of taps gets an address add of its own)
          if (i > 0) asm("" : "+v"(addr));
#endif
          float wx[4], wy[4];
          if constexpr (kBatch) {
            Taps16 t;
            lds_read_taps<P2>(addr, t, fx, fy);
            keys_weights(fx, wx);     // while the reads are in flight
            keys_weights(fy, wy);
            lds_wait_taps(t, wx, wy);
            lds_wait_i1(i1dn, i1lf, i1rt);
            cu[0] = cm[0]; cm[0] = cd[0]; cd[0] = i1dn;
            ix2[0] = i1rt - i1lf;
            iy2[0] = cd[0] - cu[0];
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float sr = wx[0] * t.p[2 * r].x;
              sr = __builtin_fmaf(wx[1], t.p[2 * r].y, sr);
              sr = __builtin_fmaf(wx[2], t.p[2 * r + 1].x, sr);
              sr = __builtin_fmaf(wx[3], t.p[2 * r + 1].y, sr);
              v = (r == 0) ? wy[0] * sr : __builtin_fmaf(wy[r], sr, v);
            }
            di[0] = __builtin_amdgcn_fmed3f(v, lo, hi_v) - cm[0];  // clip (skimage warp), ref :114
            if constexpr (MASKED) di[0] = live ? di[0] : 0.0f;
          } else {
          keys_weights(fx, wx);
          keys_weights(fy, wy);
          if constexpr (kDup) {
            // an odd tap (0,0) address: the same four floats start one float earlier -- 8-byte aligned -- in the second copy
            const unsigned odd = addr & 4u;
            const unsigned a2 = (unsigned)__umul24(odd, (unsigned)((P2 * R2 * 4 - 4) / 4)) + addr;
            const __attribute__((address_space(3))) ica_f2* q2 = (const __attribute__((address_space(3))) ica_f2*)(uintptr_t)a2;
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const ica_f2 t01 = q2[r * (P2 / 2)], t23 = q2[r * (P2 / 2) + 1];
              float s = wx[0] * t01.x;
              s = __builtin_fmaf(wx[1], t01.y, s);
              s = __builtin_fmaf(wx[2], t23.x, s);
              s = __builtin_fmaf(wx[3], t23.y, s);
              v = (r == 0) ? wy[0] * s : __builtin_fmaf(wy[r], s, v);
            }
            di[0] = __builtin_amdgcn_fmed3f(v, lo, hi_v) - cm[0];  // clip (skimage warp), ref :114
            if constexpr (MASKED) di[0] = live ? di[0] : 0.0f;
          } else
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const __attribute__((address_space(3))) float* q =
                (const __attribute__((address_space(3))) float*)(uintptr_t)(addr + (unsigned)(c * P2 * R2 * 4));
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#if defined(ICA_ABLATE) && (ICA_ABLATE & 16)   // timing-only build: half the tap reads (footprint rows 2, 3 take the values of rows 0, 1)
              const int rr = r & 1;
#else
              const int rr = r;
#endif
              float s = wx[0] * q[rr * P2];
              s = __builtin_fmaf(wx[1], q[rr * P2 + 1], s);
              s = __builtin_fmaf(wx[2], q[rr * P2 + 2], s);
              s = __builtin_fmaf(wx[3], q[rr * P2 + 3], s);
              v = (r == 0) ? wy[0] * s : __builtin_fmaf(wy[r], s, v);
            }
            di[c] = __builtin_amdgcn_fmed3f(v, lo, hi_v) - cm[c];  // clip (skimage warp), ref :114
            if constexpr (MASKED) di[c] = live ? di[c] : 0.0f;
          }
          }
        }
        float w = 1.0f;
        if constexpr (kRobust) {
          float t2 = di[0] * di[0];
#pragma unroll
          for (int c = 1; c < C; ++c) t2 = fmaf(di[c], di[c], t2);
          w = rho_prime_t<RT>(t2, lam2_v);
        }
        if constexpr (MASKED && kH) w = live ? w : 0.0f;  // pixels in the NaN frame contribute nothing (derivatives.py:84)
        accumulate(w, ix2, iy2, di, yf);
#if ICA_ROW_FENCE
        // keep the rows apart: a row hoisted over the previous one's sums costs registers
        if ((i + 1) % ICA_ROW_FENCE == 0) __builtin_amdgcn_sched_barrier(0);
#endif
      }
    };

    // ---------------------------------------------------------------------------------------------------------
    // BORDER chunk (or one whose footprint was not staged): every test of the reference's pixel masks.
    // ---------------------------------------------------------------------------------------------------------
    auto band_careful = [&](auto staged_c) {
      constexpr bool STAGED = decltype(staged_c)::value;
      [[maybe_unused]] const float cx = fmaf(A[0], xf, A[2]), cy = fmaf(A[3], xf, A[5]), gxc = A[6] * xf;
#pragma unroll 1   // the cold path: keep it small (registers, code), not fast
      for (int i = 0; i < RW; ++i) {
        const int y = yb0 + i;  // wave-uniform
        const bool row_live = (y < y_end) & !(frame_on & ((y < a.delta) | (y >= H - a.delta)));
        const float rmy = ((y >= 1) & (y <= H - 2)) ? 1.0f : 0.0f;  // ref :82: zero y-derivative on the border rows
        const bool live = col_ok & row_live;
        const float yf = (float)y;
        float mid[C], ix2[C], iy2[C];  // 2*Ix, 2*Iy (the 1/2 is applied once per tile)
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const float* q = i1p + c * P1 * R1 + i * P1;
          // unconditional LDS reads (the buffer only ever holds image values or the zeros the DMA writes for
          // out-of-image elements, so masked-out lanes see finite values) and multiplicative 0/1 masks
          mid[c] = q[P1];
          ix2[c] = (q[P1 + 1] - q[P1 - 1]) * cmx;
          iy2[c] = (q[2 * P1] - q[0]) * rmy;
        }
        float di[C];
#pragma unroll
        for (int c = 0; c < C; ++c) di[c] = 0.f;
        if constexpr (kB) {
          float dxw, dyw;
          displacement_col<kProj>(A, cx, cy, gxc, xf, yf, dxw, dyw);
          float flx = floorf(dxw), fly = floorf(dyw);
          float fx = dxw - flx, fy = dyw - fly;
          int ix = (int)flx, iy = (int)fly;
          // A position within 2^-16 px of an integer: floor() decides whether the 4x4 footprint of a border pixel is
          // inside the image, and fp32 (position good to ~1e-6 px) may land on the other side of the integer than the
          // reference's float64.  Re-evaluate those lanes in float64 (rare: exact-integer translations, "round"
          // parameters).  The interpolated VALUE is continuous across the integer, only the in/out decision is not.
          const bool near_x = live & (fminf(fx, 1.0f - fx) < kNearInteger), near_y = live & (fminf(fy, 1.0f - fy) < kNearInteger);
#ifndef ICA_NO_EXACT
          if (near_x) exact_axis<TT, 0>(a.p + (size_t)pair * 8, ident, x, y, x, ix, fx);
          if (near_y) exact_axis<TT, 1>(a.p + (size_t)pair * 8, ident, x, y, y, iy, fy);
#endif
          // 4x4 footprint inside the image (bicubic_interpolation.py:199, cval=nan), tested on the integer position of
          // tap (0,0): ux = x0 - 1 in [0, W-4], uy = y0 - 1 in [0, H-4], one unsigned compare each.  An infinite position
          // saturates in the conversion and lands outside; a NaN position converts to 0, may pass, and is caught by the
          // finiteness test on the residual below (NaN weights -> NaN sample -> residual 0), which is what "outside"
          // amounts to as well.
          const int ux = ix + xm1, uy = iy + (y - 1);
          const bool valid = live & can_sample & ((unsigned)ux <= ux_max) & ((unsigned)uy <= uy_max);
          float wx[4], wy[4];
          keys_weights(fx, wx);
          keys_weights(fy, wy);
          float iw[C];
          if constexpr (STAGED) {
            const int aoff_raw = __mul24(uy, P2) + ux - (y_lo * P2 + x_lo);
            const int aoff = valid ? aoff_raw : 0;
            const float* t = b2 + aoff;
#pragma unroll
            for (int c = 0; c < C; ++c) {
              const float* q = t + c * P2 * R2;
              float v = 0.f;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float s = wx[0] * q[r * P2];
                s = __builtin_fmaf(wx[1], q[r * P2 + 1], s);
                s = __builtin_fmaf(wx[2], q[r * P2 + 2], s);
                s = __builtin_fmaf(wx[3], q[r * P2 + 3], s);
                v = (r == 0) ? wy[0] * s : __builtin_fmaf(wy[r], s, v);
              }
              iw[c] = v;
            }
          } else {
            const TIn* t = I2 + (valid ? (size_t)uy * W + ux : 0);
#pragma unroll
            for (int c = 0; c < C; ++c) {
              const TIn* q = t + c * plane;
              float v = 0.f;
              if (valid) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const TIn* rp = q + (size_t)r * W;
                  float s = wx[0] * (float)rp[0];
                  s = __builtin_fmaf(wx[1], (float)rp[1], s);
                  s = __builtin_fmaf(wx[2], (float)rp[2], s);
                  s = __builtin_fmaf(wx[3], (float)rp[3], s);
                  v = (r == 0) ? wy[0] * s : __builtin_fmaf(wy[r], s, v);
                }
              }
              iw[c] = v;
            }
          }
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const float d = __builtin_amdgcn_fmed3f(iw[c], lo, hi) - mid[c];  // clip (skimage warp), ref :114
            di[c] = (valid & (fabsf(d) <= 3.0e38f)) ? d : 0.0f;           // NaN / outside -> 0 (io.py:99)
          }
        }
        float w = live ? 1.0f : 0.0f;  // pixels in the NaN frame contribute nothing (derivatives.py:84)
        if constexpr (kRobust) {
          float t2 = di[0] * di[0];
#pragma unroll
          for (int c = 1; c < C; ++c) t2 = fmaf(di[c], di[c], t2);
          w = live ? rho_prime_t<RT>(t2, lam2) : 0.0f;  // rho'(0) where the warp is outside
        }
        if constexpr (!kH) {  // unweighted sums: the mask goes into the residual (one channel or three)
#pragma unroll
          for (int c = 0; c < C; ++c) di[c] *= w;
        }
        accumulate(w, ix2, iy2, di, yf);
      }
    };

    // ---------------------------------------------------------------------------------------------------------
    // Channel-sequential tiles (three channels, TileCfg::seq): the same two bands, run once per channel PC on the
    // channel the tile holds.  The sampling position and the Keys weights are formed again in every pass (the vector ALU
    // has the room: this kernel is bound by its copies and its LDS reads), the sums over the channels live in sg[][],
    // in the order of the resident-channel bands (channel 0 first), and the weight, its products and the moments follow
    // the last channel.
    // ---------------------------------------------------------------------------------------------------------
    // channel PC's terms of a pixel's sums (first channel: products, later ones: multiply-adds, as in accumulate())
    [[maybe_unused]] auto add_channel = [&](float* s6, float ix2, float iy2, float di) {
      if constexpr (kB) {
        s6[0] = (PC == 0) ? ix2 * di : fmaf(ix2, di, s6[0]);
        s6[1] = (PC == 0) ? iy2 * di : fmaf(iy2, di, s6[1]);
      }
      if constexpr (kH) {
        s6[2] = (PC == 0) ? ix2 * ix2 : fmaf(ix2, ix2, s6[2]);
        s6[3] = (PC == 0) ? ix2 * iy2 : fmaf(ix2, iy2, s6[3]);
        s6[4] = (PC == 0) ? iy2 * iy2 : fmaf(iy2, iy2, s6[4]);
      }
      if constexpr (kRobust) s6[5] = (PC == 0) ? di * di : fmaf(di, di, s6[5]);
    };
    [[maybe_unused]] auto band_fast_seq = [&](auto masked_c) {
      constexpr bool MASKED = decltype(masked_c)::value;
      const int lds_b2 = (int)(unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)b2;
      const float tap0f = (float)(lds_b2 + ((yb0 - 1 - y_lo) * P2 + (xm1 - x_lo)) * 4);
      // (volatile: the three passes form the same positions and weights from the same inputs -- left to itself the compiler
      //  computes them once for the whole chunk and keeps them in scratch memory across the passes)
      float yf0 = (float)yb0;
      asm volatile("" : "+v"(yf0));
      [[maybe_unused]] const float cx = fmaf(A[0], xf, A[2]), cy = fmaf(A[3], xf, A[5]), gxc = A[6] * xf, nxf = -xf;
      [[maybe_unused]] const float a1v = pin_vgpr(A[1]), a4v = pin_vgpr(A[4]), a7v = pin_vgpr(A[7]);
      [[maybe_unused]] const float hi_v = pin_vgpr(hi), lam2_v = pin_vgpr(lam2);
      float cu, cm = i1p[0], cd = i1p[P1];
#pragma unroll
      for (int i = 0; i < RW; ++i) {
        const float yf = i == 0 ? yf0 : yf0 + (float)i;
        [[maybe_unused]] bool live = true;
        if constexpr (MASKED) {
          const int y = yb0 + i;  // wave-uniform
          live = col_ok & (y < y_end) & !((y < a.delta) | (y >= H - a.delta));
        }
        const float* q1 = i1p + (i + 1) * P1;
        cu = cm; cm = cd; cd = q1[P1];
        const float ix2 = q1[1] - q1[-1], iy2 = cd - cu;
        float di = 0.f;
        if constexpr (kB) {
          float dxw, dyw;
          if constexpr (kProj) {
            dxw = fmaf(a1v, yf, cx); dyw = fmaf(a4v, yf, cy);
            const float g = fmaf(a7v, yf, gxc);
            const float r = rcp_nr(1.0f + g);
            dxw = fmaf(nxf, g, dxw) * r;
            dyw = fmaf(-yf, g, dyw) * r;
          } else {
            dxw = fmaf(a1v, yf, cx); dyw = fmaf(a4v, yf, cy);
          }
          const float flx = floorf(dxw), fly = floorf(dyw);
          const float fx = dxw - flx, fy = dyw - fly;
          float af;
          asm("v_fmamk_f32 %0, %1, %3, %2" : "=v"(af) : "v"(fly), "v"(tap0f), "n"(kPitchBits));
          af = fmaf(flx, 4.0f, af);
          unsigned addr = (unsigned)((int)af + i * P2 * 4);
          if (i > 0) asm("" : "+v"(addr));
          float wx[4], wy[4];
          keys_weights(fx, wx);
          keys_weights(fy, wy);
          const __attribute__((address_space(3))) float* q = (const __attribute__((address_space(3))) float*)(uintptr_t)addr;
          float v = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float sr = wx[0] * q[r * P2];
            sr = __builtin_fmaf(wx[1], q[r * P2 + 1], sr);
            sr = __builtin_fmaf(wx[2], q[r * P2 + 2], sr);
            sr = __builtin_fmaf(wx[3], q[r * P2 + 3], sr);
            v = (r == 0) ? wy[0] * sr : __builtin_fmaf(wy[r], sr, v);
          }
          di = __builtin_amdgcn_fmed3f(v, lo, hi_v) - cm;  // clip (skimage warp), ref :114
          if constexpr (MASKED) di = live ? di : 0.0f;
        }
        add_channel(sg[i], ix2, iy2, di);
        if constexpr (PC + 1 < kPasses) {
          // (the sums are not used before the last pass: without this the compiler defers the whole arithmetic to that
          //  pass and carries the raw LDS values of the earlier ones -- 19 per pixel -- through scratch memory)
          if constexpr (kB) asm volatile("" : "+v"(sg[i][0]), "+v"(sg[i][1]));
          if constexpr (kH) asm volatile("" : "+v"(sg[i][2]), "+v"(sg[i][3]), "+v"(sg[i][4]));
          if constexpr (kRobust) asm volatile("" : "+v"(sg[i][5]));
        }
        if constexpr (PC == kPasses - 1) {
          float w = 1.0f;
          if constexpr (kRobust) w = rho_prime_t<RT>(sg[i][5], lam2_v);
          if constexpr (MASKED && kH) w = live ? w : 0.0f;  // pixels in the NaN frame contribute nothing (derivatives.py:84)
          finish_seq(w, sg[i], yf);
        }
#if ICA_ROW_FENCE
        if ((i + 1) % ICA_ROW_FENCE == 0) __builtin_amdgcn_sched_barrier(0);
#endif
      }
    };
#if defined(ICA_ABLATE) && (ICA_ABLATE & 1)
    if (a.delta < 0)  // timing-only build: staging and barriers only
#endif
    if constexpr (kSeq) {   // (careful chunks never get here: step() runs them in one pass, without the tile)
      if constexpr (CLS == 0) band_fast_seq(std::false_type{});
      else if constexpr (CLS == 1) band_fast_seq(std::true_type{});
      else if ((cls_mask[0] >> k) & 1) band_fast_seq(std::false_type{});
      else band_fast_seq(std::true_type{});
    } else
    if (!kLoader || wave < kComputeWaves) {
      if constexpr (CLS == 0) band_fast(std::false_type{});
      else if constexpr (CLS == 1) band_fast(std::true_type{});
      else if constexpr (CLS == 3) {  // (one loop, the class looked up per chunk)
        if ((cls_mask[0] >> k) & 1) band_fast(std::false_type{});
        else if ((cls_mask[1] >> k) & 1) band_fast(std::true_type{});
        else if (staged) band_careful(std::true_type{});
        else band_careful(std::false_type{});
      }
      else if (staged) band_careful(std::true_type{});
      else band_careful(std::false_type{});
    }
#ifdef ICA_STAMP
    if constexpr (!U8 && NBUF == 1) stamp[3] += clock64();
#endif
   };
    if constexpr (kSeq) {
      one_pass(IC<0>{});
      one_pass(IC<1>{});
      one_pass(IC<2>{});
    } else {
      one_pass(IC<0>{});
    }
  };
  // Ring protocol (two tiles): chunk `next` is in flight while chunk k is computed.  A wave's copies retire in order;
  // the barrier makes every wave's part visible, and also means everyone has finished reading the chunk before k,
  // whose tile is the one `next` is about to overwrite.  Chunks that lie in the NaN frame entirely are in no class.
  if constexpr (kWalk == 1) {  // top to bottom, one loop
    unsigned rem = cls_mask[0] | cls_mask[1] | cls_mask[2];
    while (rem) {
      const int k = __builtin_ctz(rem);
      rem &= rem - 1;
      step(IC<3>{}, k, rem ? __builtin_ctz(rem) : -1);
    }
  } else {
    while (cls_mask[0]) {
      const int k = __builtin_ctz(cls_mask[0]);
      cls_mask[0] &= cls_mask[0] - 1;
      step(IC<0>{}, k, first_from(0));
    }
    while (cls_mask[1]) {
      const int k = __builtin_ctz(cls_mask[1]);
      cls_mask[1] &= cls_mask[1] - 1;
      step(IC<1>{}, k, first_from(1));
    }
    while (cls_mask[2]) {
      const int k = __builtin_ctz(cls_mask[2]);
      cls_mask[2] &= cls_mask[2] - 1;
      step(IC<2>{}, k, first_from(2));
    }
  }
  // The y-moments of one lane's column over the whole tile (<= 64 rows per lane) stay in fp32 registers; they are
  // promoted to fp64 here, before x powers up to x^4 are applied and 64 lanes x 4 waves x tiles are summed.
#ifdef ICA_STAMP
  const long long stamp_fold = clock64();
#endif
  double m64[NMOM];
#pragma unroll
  for (int i = 0; i < NMOM; ++i) m64[i] = (double)m32[i];

  // ---- fold the per-column moments into the tile's partial sums (fixed order) ----
  __syncthreads();
  double* mom = reinterpret_cast<double*>(lds);   // [NMOM][64] moments of the whole tile, per column
  double* con = mom + NMOM * 64;                  // [N + NT][64] per-column contribution to each output sum
  for (int wv = 0; wv < kIterWaves; ++wv) {           // waves take turns: deterministic wave order
    if (wave == wv) {
#pragma unroll
      for (int i = 0; i < NMOM; ++i) {
        if (wv == 0) mom[i * 64 + lane] = m64[i];
        else mom[i * 64 + lane] += m64[i];
      }
    }
    __syncthreads();
  }
  // every lane applies its own x powers; the (N + NT) outputs are dealt round-robin to the 4 waves.  When the
  // contributions of all outputs do not fit the LDS next to the moments (robust homography), they are produced and
  // summed in two halves.
  constexpr int kOut = N + NT;
  constexpr int kHalf = kFoldSplit ? (kOut + 1) / 2 : kOut;
#pragma unroll
  for (int part = 0; part < (kFoldSplit ? 2 : 1); ++part) {
    const int e_lo = part * kHalf, e_hi = (part == 0 ? kHalf : kOut);
    {
      double xp[5];
      xp[0] = 1.0;
#pragma unroll
      for (int i = 1; i < 5; ++i) xp[i] = xp[i - 1] * (double)x;
      auto term = [&](Mono j1, Mono j2, int base) -> double {  // j1 * j2 * moment, all indices compile-time
        return (double)(j1.c * j2.c) * xp[j1.a + j2.a] * mom[(base + j1.b + j2.b) * 64 + lane];
      };
      int e = 0;
#pragma unroll
      for (int k = 0; k < N; ++k) {   // b_k (halved: the gradients were accumulated as 2*Ix, 2*Iy)
        if (kB && ((e & 3) == wave) && e >= e_lo && e < e_hi) {
          constexpr Mono one{1, 0, 0};
          const Mono jx = jac_x(TT, k), jy = jac_y(TT, k);
          double v = 0.0;
          if (jx.c) v += term(jx, one, ML::ogx);
          if (jy.c) v += term(jy, one, ML::ogy);
          con[(e - e_lo) * 64 + lane] = 0.5 * v;
        }
        ++e;
      }
#pragma unroll
      for (int k = 0; k < N; ++k) {
#pragma unroll
        for (int m = k; m < N; ++m) {  // H_km (quartered)
          if (kH && ((e & 3) == wave) && e >= e_lo && e < e_hi) {
            const Mono jxk = jac_x(TT, k), jyk = jac_y(TT, k), jxm = jac_x(TT, m), jym = jac_y(TT, m);
            double v = 0.0;
            if (jxk.c && jxm.c) v += term(jxk, jxm, ML::oxx);
            if (jxk.c && jym.c) v += term(jxk, jym, ML::oxy);
            if (jyk.c && jxm.c) v += term(jyk, jxm, ML::oxy);
            if (jyk.c && jym.c) v += term(jyk, jym, ML::oyy);
            con[(e - e_lo) * 64 + lane] = 0.25 * v;
          }
          ++e;
        }
      }
    }
    __syncthreads();
    {
      const int t = threadIdx.x;
      const int e = e_lo + (t >> 2), q = t & 3;   // 4 threads per output, 16 columns each, then a fixed 2-step tree
      double val = 0.0;
      const bool have = (e < e_hi) && ((e < N) ? kB : (e < N + NT ? kH : false));
      if (have) {
#pragma unroll
        for (int l = 0; l < 16; ++l) val += con[(e - e_lo) * 64 + q * 16 + l];
      }
      val += __shfl_down(val, 2, 64);
      val += __shfl_down(val, 1, 64);
      if (q == 0 && have) {
        double* out = a.partials + ((size_t)pair_slot * ntiles + tile_id) * kAccStride;  // indexed by launch slot
        double* dst = out + (e < N ? e : kMaxN + (e - N));
        *dst = val;
      }
    }
    if (kFoldSplit && part == 0) __syncthreads();
  }
  if (clock_wg && threadIdx.x == 0) {
    a.clock_acc[0] += (unsigned long long)clock64() - a.clock_acc[2];
    a.clock_acc[1] += (unsigned long long)wall_clock64() - a.clock_acc[3];
  }
#ifdef ICA_STAMP
  {
    const long long e = clock64();
    stamp[5] = e - stamp_fold;
    stamp[6] = e - stamp_begin;
    stamp[7] = wall_clock64() - stamp_rt_begin;  // 100 MHz ticks
    if (lane == 0) {
      unsigned long long* rec = ica_stamp_acc + (size_t)((blockIdx.x * kIterWaves + wave) & (kStampWaves - 1)) * 16;
      for (int i = 0; i < 8; ++i) rec[i] = (unsigned long long)stamp[i];
      rec[8] = (unsigned long long)stamp_rt_begin;                       // absolute, 100 MHz
      rec[9] = (unsigned long long)(stamp_rt_begin + stamp[7]);
      unsigned hw_id, xcc_id;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
      rec[10] = hw_id;
      rec[11] = xcc_id;
    }
  }
#endif
}

}  // namespace ica
