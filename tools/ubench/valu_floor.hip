// What does one vector instruction cost a SIMD of gfx950 when the chip is full -- in shader cycles AND in nanoseconds?
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_floor valu_floor.hip -lpthread && ./valu_floor
//
// Method (differs from issue_cost.hip, whose wall-clock column mixed launch overhead and uneven block placement into the
// per-instruction figure):
//   * occupancy is FORCED: every block asks for 160 KB / W of dynamic LDS, so a CU holds exactly W blocks of 4 waves
//     (W waves per SIMD) however the dispatcher deals them;
//   * the grid is 12 rounds of (256 CUs x W blocks), each block runs for ~1 ms: the tail and the launch are < 1 %;
//   * nanoseconds per wave-instruction per SIMD = event time x 1024 SIMDs / (wave-instructions issued);
//   * every wave brackets its loop with s_memtime (shader clock) and s_memrealtime (constant 100 MHz): the ratio is
//     the clock the shader really ran at under THIS load, and ticks / instruction is the cost in cycles;
//   * a host thread samples the driver's view of the clock (hwmon freq1_input / pp_dpm_sclk) while the kernel runs.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>
#include <dirent.h>

#define S_(x) #x
#define S(x) S_(x)

// v8..v39 sources (never written in the loop), v40..v71 destinations
#define PROLOGUE                                                                                                        \
  "v_mov_b32 v8, 1.0\n v_mov_b32 v9, 0.5\n v_mov_b32 v10, 2.0\n v_mov_b32 v11, 4.0\n"                                     \
  "v_mov_b32 v12, 1.0\n v_mov_b32 v13, 0.5\n v_mov_b32 v14, 2.0\n v_mov_b32 v15, 4.0\n"                                   \
  "v_mov_b32 v16, 1.0\n v_mov_b32 v17, 0.5\n v_mov_b32 v18, 2.0\n v_mov_b32 v19, 4.0\n"                                   \
  "v_mov_b32 v20, 1.0\n v_mov_b32 v21, 0.5\n v_mov_b32 v22, 2.0\n v_mov_b32 v23, 4.0\n"                                   \
  "v_mov_b32 v24, 1.0\n v_mov_b32 v25, 0.5\n v_mov_b32 v26, 2.0\n v_mov_b32 v27, 4.0\n"                                   \
  "v_mov_b32 v28, 1.0\n v_mov_b32 v29, 0.5\n v_mov_b32 v30, 2.0\n v_mov_b32 v31, 4.0\n"                                   \
  "v_mov_b32 v32, 1.0\n v_mov_b32 v33, 0.5\n v_mov_b32 v34, 2.0\n v_mov_b32 v35, 4.0\n"                                   \
  "v_mov_b32 v36, 1.0\n v_mov_b32 v37, 0.5\n v_mov_b32 v38, 2.0\n v_mov_b32 v39, 4.0\n"                                   \
  "v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n"                                           \
  "v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n"                                           \
  "v_mov_b32 v48, 0\n v_mov_b32 v49, 0\n v_mov_b32 v50, 0\n v_mov_b32 v51, 0\n"                                           \
  "v_mov_b32 v52, 0\n v_mov_b32 v53, 0\n v_mov_b32 v54, 0\n v_mov_b32 v55, 0\n"                                           \
  "v_mov_b32 v56, 0\n v_mov_b32 v57, 0\n v_mov_b32 v58, 0\n v_mov_b32 v59, 0\n"                                           \
  "v_mov_b32 v60, 0\n v_mov_b32 v61, 0\n v_mov_b32 v62, 0\n v_mov_b32 v63, 0\n"                                           \
  "v_mov_b32 v64, 0\n v_mov_b32 v65, 0\n v_mov_b32 v66, 0\n v_mov_b32 v67, 0\n"                                           \
  "v_mov_b32 v68, 0\n v_mov_b32 v69, 0\n v_mov_b32 v70, 0\n v_mov_b32 v71, 0\n"                                           \
  "s_mov_b32 s20, 1.0\n s_mov_b32 s21, 0.5\n"

#define CLOBBERS                                                                                                        \
  "memory", "vcc", "s20", "s21", "s24", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19",  \
      "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35",     \
      "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51",     \
      "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67",     \
      "v68", "v69", "v70", "v71", "v72", "v73"

// out[4 * wave_slot + {0,1}] = shader ticks, 100 MHz ticks of this wave's loop
#define KERNEL(name, body)                                                                                              \
  __global__ __launch_bounds__(256) void name(unsigned long long* out, int iters) {                                     \
    extern __shared__ float lds[];                                                                                      \
    lds[threadIdx.x] = (float)threadIdx.x;                                                                              \
    __syncthreads();                                                                                                    \
    unsigned long long t0, t1, r0, r1;                                                                                  \
    const unsigned la = (threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 1024;                                             \
    asm volatile(PROLOGUE "v_mov_b32 v72, %4\n s_mov_b32 s24, %5\n"                                                     \
                 "s_waitcnt vmcnt(0) lgkmcnt(0)\n"                                                                      \
                 "s_memrealtime %2\n s_memtime %0\n s_waitcnt lgkmcnt(0)\n"                                             \
                 "1:\n" body                                                                                            \
                 "s_sub_u32 s24, s24, 1\n s_cmp_lg_u32 s24, 0\n s_cbranch_scc1 1b\n"                                    \
                 "s_waitcnt vmcnt(0) lgkmcnt(0)\n"                                                                      \
                 "s_memtime %1\n s_memrealtime %3\n s_waitcnt lgkmcnt(0)\n"                                             \
                 : "=&s"(t0), "=&s"(t1), "=&s"(r0), "=&s"(r1)                                                           \
                 : "v"(la), "s"(iters)                                                                                  \
                 : CLOBBERS);                                                                                           \
    if ((threadIdx.x & 63) == 0) {                                                                                      \
      const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);                                                     \
      out[2 * w] = t1 - t0;                                                                                             \
      out[2 * w + 1] = r1 - r0;                                                                                         \
    }                                                                                                                   \
    if (lds[threadIdx.x] < -1.f) out[0] = 0;                                                                            \
  }

// 32 instructions per body.  f(dst, a, b, c)
#define B32(f)                                                                                                          \
  f(40, 8, 9, 10) f(41, 11, 12, 13) f(42, 14, 15, 16) f(43, 17, 18, 19) f(44, 20, 21, 22) f(45, 23, 24, 25)               \
  f(46, 26, 27, 28) f(47, 29, 30, 31) f(48, 32, 33, 34) f(49, 35, 36, 37) f(50, 38, 39, 8) f(51, 9, 10, 11)               \
  f(52, 12, 13, 14) f(53, 15, 16, 17) f(54, 18, 19, 20) f(55, 21, 22, 23) f(56, 24, 25, 26) f(57, 27, 28, 29)             \
  f(58, 30, 31, 32) f(59, 33, 34, 35) f(60, 36, 37, 38) f(61, 39, 8, 9) f(62, 10, 11, 12) f(63, 13, 14, 15)               \
  f(64, 16, 17, 18) f(65, 19, 20, 21) f(66, 22, 23, 24) f(67, 25, 26, 27) f(68, 28, 29, 30) f(69, 31, 32, 33)             \
  f(70, 34, 35, 36) f(71, 37, 38, 39)
// the same 32 with only D distinct destinations (dependent chains of fmac: each destination is reused every D instructions)
#define B32D2(f) f(40, 8, 9, 10) f(41, 11, 12, 13) f(40, 14, 15, 16) f(41, 17, 18, 19) f(40, 20, 21, 22) f(41, 23, 24, 25) \
  f(40, 26, 27, 28) f(41, 29, 30, 31) f(40, 32, 33, 34) f(41, 35, 36, 37) f(40, 38, 39, 8) f(41, 9, 10, 11)               \
  f(40, 12, 13, 14) f(41, 15, 16, 17) f(40, 18, 19, 20) f(41, 21, 22, 23) f(40, 24, 25, 26) f(41, 27, 28, 29)             \
  f(40, 30, 31, 32) f(41, 33, 34, 35) f(40, 36, 37, 38) f(41, 39, 8, 9) f(40, 10, 11, 12) f(41, 13, 14, 15)               \
  f(40, 16, 17, 18) f(41, 19, 20, 21) f(40, 22, 23, 24) f(41, 25, 26, 27) f(40, 28, 29, 30) f(41, 31, 32, 33)             \
  f(40, 34, 35, 36) f(41, 37, 38, 39)
#define B32D1(f) f(40, 8, 9, 10) f(40, 11, 12, 13) f(40, 14, 15, 16) f(40, 17, 18, 19) f(40, 20, 21, 22) f(40, 23, 24, 25) \
  f(40, 26, 27, 28) f(40, 29, 30, 31) f(40, 32, 33, 34) f(40, 35, 36, 37) f(40, 38, 39, 8) f(40, 9, 10, 11)               \
  f(40, 12, 13, 14) f(40, 15, 16, 17) f(40, 18, 19, 20) f(40, 21, 22, 23) f(40, 24, 25, 26) f(40, 27, 28, 29)             \
  f(40, 30, 31, 32) f(40, 33, 34, 35) f(40, 36, 37, 38) f(40, 39, 8, 9) f(40, 10, 11, 12) f(40, 13, 14, 15)               \
  f(40, 16, 17, 18) f(40, 19, 20, 21) f(40, 22, 23, 24) f(40, 25, 26, 27) f(40, 28, 29, 30) f(40, 31, 32, 33)             \
  f(40, 34, 35, 36) f(40, 37, 38, 39)
#define B32D4(f) f(40, 8, 9, 10) f(41, 11, 12, 13) f(42, 14, 15, 16) f(43, 17, 18, 19) f(40, 20, 21, 22) f(41, 23, 24, 25) \
  f(42, 26, 27, 28) f(43, 29, 30, 31) f(40, 32, 33, 34) f(41, 35, 36, 37) f(42, 38, 39, 8) f(43, 9, 10, 11)               \
  f(40, 12, 13, 14) f(41, 15, 16, 17) f(42, 18, 19, 20) f(43, 21, 22, 23) f(40, 24, 25, 26) f(41, 27, 28, 29)             \
  f(42, 30, 31, 32) f(43, 33, 34, 35) f(40, 36, 37, 38) f(41, 39, 8, 9) f(42, 10, 11, 12) f(43, 13, 14, 15)               \
  f(40, 16, 17, 18) f(41, 19, 20, 21) f(42, 22, 23, 24) f(43, 25, 26, 27) f(40, 28, 29, 30) f(41, 31, 32, 33)             \
  f(42, 34, 35, 36) f(43, 37, 38, 39)
// every source of an instruction in the same register bank (index mod 4)
#define B32SB(f)                                                                                                        \
  f(40, 8, 12, 16) f(41, 9, 13, 17) f(42, 10, 14, 18) f(43, 11, 15, 19) f(44, 20, 24, 28) f(45, 21, 25, 29)               \
  f(46, 22, 26, 30) f(47, 23, 27, 31) f(48, 32, 36, 8) f(49, 33, 37, 9) f(50, 34, 38, 10) f(51, 35, 39, 11)               \
  f(52, 12, 16, 20) f(53, 13, 17, 21) f(54, 14, 18, 22) f(55, 15, 19, 23) f(56, 24, 28, 32) f(57, 25, 29, 33)             \
  f(58, 26, 30, 34) f(59, 27, 31, 35) f(60, 36, 8, 12) f(61, 37, 9, 13) f(62, 38, 10, 14) f(63, 39, 11, 15)               \
  f(64, 16, 20, 24) f(65, 17, 21, 25) f(66, 18, 22, 26) f(67, 19, 23, 27) f(68, 28, 32, 36) f(69, 29, 33, 37)             \
  f(70, 30, 34, 38) f(71, 31, 35, 39)

#define FMAC(d, a, b, c) "v_fmac_f32 v" S_(d) ", v" S_(a) ", v" S_(b) "\n"
#define FMA3(d, a, b, c) "v_fma_f32 v" S_(d) ", v" S_(a) ", v" S_(b) ", v" S_(c) "\n"
#define ADD2(d, a, b, c) "v_add_f32 v" S_(d) ", v" S_(a) ", v" S_(b) "\n"
#define MUL2(d, a, b, c) "v_mul_f32 v" S_(d) ", v" S_(a) ", v" S_(b) "\n"
#define MOV1(d, a, b, c) "v_mov_b32 v" S_(d) ", v" S_(a) "\n"
#define FMACS(d, a, b, c) "v_fmac_f32 v" S_(d) ", s20, v" S_(b) "\n"
#define FMAMK(d, a, b, c) "v_fmamk_f32 v" S_(d) ", v" S_(a) ", 0x40490fdb, v" S_(b) "\n"
#define FLOOR(d, a, b, c) "v_floor_f32 v" S_(d) ", v" S_(a) "\n"
#define FRACT(d, a, b, c) "v_fract_f32 v" S_(d) ", v" S_(a) "\n"
#define MAX2(d, a, b, c) "v_max_f32 v" S_(d) ", v" S_(a) ", v" S_(b) "\n"
#define MED3(d, a, b, c) "v_med3_f32 v" S_(d) ", v" S_(a) ", v" S_(b) ", v" S_(c) "\n"
#define CVTI(d, a, b, c) "v_cvt_i32_f32 v" S_(d) ", v" S_(a) "\n"
#define RCP(d, a, b, c) "v_rcp_f32 v" S_(d) ", v" S_(a) "\n"
#define ADDU(d, a, b, c) "v_add_u32 v" S_(d) ", v" S_(a) ", v" S_(b) "\n"
#define LSHL(d, a, b, c) "v_lshlrev_b32 v" S_(d) ", 2, v" S_(a) "\n"
#define SUB2(d, a, b, c) "v_sub_f32 v" S_(d) ", v" S_(a) ", v" S_(b) "\n"
#define CNDM(d, a, b, c) "v_cndmask_b32 v" S_(d) ", v" S_(a) ", v" S_(b) ", vcc\n"
#define DSR(d, a, b, c) "ds_read_b32 v" S_(d) ", v72 offset:" S_(a) "*16\n"
#define DSR2(d, a, b, c) "ds_read2_b32 v[" S_(d) ":" S_(d) "+1], v72 offset0:" S_(a) " offset1:" S_(b) "\n"
// packed: register pairs
#define B16P(f) f(40, 8, 10, 12) f(42, 14, 16, 18) f(44, 20, 22, 24) f(46, 26, 28, 30) f(48, 32, 34, 36) f(50, 38, 8, 10) \
  f(52, 12, 14, 16) f(54, 18, 20, 22) f(56, 24, 26, 28) f(58, 30, 32, 34) f(60, 36, 38, 8) f(62, 10, 12, 14) f(64, 16, 18, 20) \
  f(66, 22, 24, 26) f(68, 28, 30, 32) f(70, 34, 36, 38)
#define PKFMA(d, a, b, c) "v_pk_fma_f32 v[" S_(d) ":" S_(d) "+1], v[" S_(a) ":" S_(a) "+1], v[" S_(b) ":" S_(b) "+1], v[" S_(c) ":" S_(c) "+1]\n"
#define PKMUL(d, a, b, c) "v_pk_mul_f32 v[" S_(d) ":" S_(d) "+1], v[" S_(a) ":" S_(a) "+1], v[" S_(b) ":" S_(b) "+1]\n"

KERNEL(k_fmac, B32(FMAC))
KERNEL(k_fmac_d4, B32D4(FMAC))
KERNEL(k_fmac_d2, B32D2(FMAC))
KERNEL(k_fmac_d1, B32D1(FMAC))
KERNEL(k_fmac_samebank, B32SB(FMAC))
KERNEL(k_fma3, B32(FMA3))
KERNEL(k_fma3_samebank, B32SB(FMA3))
KERNEL(k_add, B32(ADD2))
KERNEL(k_mul, B32(MUL2))
KERNEL(k_mov, B32(MOV1))
KERNEL(k_fmac_sgpr, B32(FMACS))
KERNEL(k_fmamk, B32(FMAMK))
KERNEL(k_floor, B32(FLOOR))
KERNEL(k_fract, B32(FRACT))
KERNEL(k_max, B32(MAX2))
KERNEL(k_med3, B32(MED3))
KERNEL(k_cvt_i32, B32(CVTI))
KERNEL(k_rcp, B32(RCP))
KERNEL(k_add_u32, B32(ADDU))
KERNEL(k_lshl, B32(LSHL))
KERNEL(k_cndmask_vcc, B32(CNDM))
KERNEL(k_pk_fma, B16P(PKFMA) B16P(PKFMA))
KERNEL(k_pk_mul, B16P(PKMUL) B16P(PKMUL))
// the iteration kernel's rough mix: per 32 instructions 24 fast-class, 5 reads of LDS, 3 slow-class
#define MIXA                                                                                                            \
  FMAC(40, 8, 9, 0) FMAC(41, 11, 12, 0) DSR(56, 1, 0, 0) FMAC(42, 14, 15, 0) FMAC(43, 17, 18, 0) MUL2(44, 20, 21, 0)      \
  FMAC(45, 23, 24, 0) DSR(57, 2, 0, 0) FMAC(46, 26, 27, 0) ADD2(47, 29, 30, 0) FLOOR(48, 32, 0, 0) FMAC(49, 35, 36, 0)    \
  DSR(58, 3, 0, 0) FMAC(50, 38, 39, 0) FMAC(51, 9, 10, 0) MUL2(52, 12, 13, 0) FMAC(53, 15, 16, 0) DSR(59, 4, 0, 0)        \
  FMAC(54, 18, 19, 0) SUB2(55, 21, 22, 0) MED3(60, 24, 25, 26) FMAC(61, 27, 28, 0) FMAC(62, 30, 31, 0) DSR(63, 5, 0, 0)   \
  FMAC(64, 33, 34, 0) MUL2(65, 36, 37, 0) CVTI(66, 39, 0, 0) FMAC(67, 8, 10, 0) FMAC(68, 11, 13, 0) FMAC(69, 14, 16, 0)   \
  ADD2(70, 17, 19, 0) FMAC(71, 20, 22, 0) "s_waitcnt lgkmcnt(0)\n"
KERNEL(k_mix_24fast_5lds_3slow, MIXA)
#define MIXB                                                                                                            \
  FMAC(40, 8, 9, 0) FMAC(41, 11, 12, 0) FMAC(42, 14, 15, 0) FMAC(43, 17, 18, 0) MUL2(44, 20, 21, 0)                       \
  FMAC(45, 23, 24, 0) FMAC(46, 26, 27, 0) ADD2(47, 29, 30, 0) FLOOR(48, 32, 0, 0) FMAC(49, 35, 36, 0)                     \
  FMAC(50, 38, 39, 0) FMAC(51, 9, 10, 0) MUL2(52, 12, 13, 0) FMAC(53, 15, 16, 0)                                          \
  FMAC(54, 18, 19, 0) SUB2(55, 21, 22, 0) MED3(60, 24, 25, 26) FMAC(61, 27, 28, 0) FMAC(62, 30, 31, 0)                    \
  FMAC(64, 33, 34, 0) MUL2(65, 36, 37, 0) CVTI(66, 39, 0, 0) FMAC(67, 8, 10, 0) FMAC(68, 11, 13, 0) FMAC(69, 14, 16, 0)   \
  ADD2(70, 17, 19, 0) FMAC(71, 20, 22, 0)
KERNEL(k_mix_24fast_3slow, MIXB)
KERNEL(k_ds_read_b32, B32(DSR) "s_waitcnt lgkmcnt(0)\n")


// ---- one pixel row of the iteration kernel's interior band, as an instruction mix: 85 fast-class, 4 slow-class, 2
// transcendental vector instructions, 8 ds_read2_b32 (taps) + 3 ds_read_b32 (I1) ----
#define F(d, a, b) "v_fmac_f32 v[" S_(d) "], v" S_(a) ", v" S_(b) "\n"
#define F8(d) F(d, 8, 9) F(d + 1, 10, 11) F(d + 2, 12, 13) F(d + 3, 14, 15) F(d + 4, 16, 17) F(d + 5, 18, 19) F(d + 6, 20, 21) F(d + 7, 22, 23)
#define TAPS(base)                                                                                                      \
  "ds_read2_b32 v[40:41], v72 offset0:" S_(base) " offset1:" S_(base) "+1\n"                                              \
  "ds_read2_b32 v[42:43], v72 offset0:" S_(base) "+2 offset1:" S_(base) "+3\n"                                            \
  "ds_read2_b32 v[44:45], v72 offset0:" S_(base) "+80 offset1:" S_(base) "+81\n"                                          \
  "ds_read2_b32 v[46:47], v72 offset0:" S_(base) "+82 offset1:" S_(base) "+83\n"                                          \
  "ds_read2_b32 v[48:49], v72 offset0:" S_(base) "+160 offset1:" S_(base) "+161\n"                                        \
  "ds_read2_b32 v[50:51], v72 offset0:" S_(base) "+162 offset1:" S_(base) "+163\n"                                        \
  "ds_read2_b32 v[52:53], v72 offset0:" S_(base) "+240 offset1:" S_(base) "+241\n"                                        \
  "ds_read2_b32 v[54:55], v72 offset0:" S_(base) "+242 offset1:" S_(base) "+243\n"                                        \
  "ds_read_b32 v56, v72 offset:4000\n ds_read_b32 v57, v72 offset:4288\n ds_read_b32 v58, v72 offset:4296\n"
#define SLOW4 "v_floor_f32 v60, v24\n v_floor_f32 v61, v25\n v_cvt_i32_f32 v62, v26\n v_med3_f32 v63, v27, v28, v29\n"
#define TR2 "v_rcp_f32 v64, v30\n v_rcp_f32 v65, v31\n"
// batched: 16 position instructions, all reads, 18 weight instructions, one wait, the rest
#define ROW_BATCHED F8(56) F8(64) TAPS(0) F8(56) F8(64) F(70, 8, 9) F(71, 10, 11) "s_waitcnt lgkmcnt(0)\n"                 \
  F8(40) F8(48) F8(56) F8(64) F8(40) F8(48) F(56, 8, 9) F(57, 8, 9) F(58, 8, 9) SLOW4 TR2
// serial (what the compiler emitted): every pair of taps read, waited for and used on its own
#define PAIR(r, o0, o1) "ds_read2_b32 v[40:41], v72 offset0:" S_(o0) " offset1:" S_(o1) "\n s_waitcnt lgkmcnt(0)\n" F(r, 40, 8) F(r, 41, 9)
#define ROW_SERIAL F8(56) F8(64) "ds_read_b32 v56, v72 offset:4000\n ds_read2_b32 v[58:59], v72 offset0:250 offset1:252\n s_waitcnt lgkmcnt(0)\n" \
  F8(56) F8(64) F(70, 8, 9) F(71, 10, 11)                                                                               \
  PAIR(60, 0, 1) PAIR(60, 2, 3) F(66, 60, 9) PAIR(61, 80, 81) PAIR(61, 82, 83) F(66, 61, 9) PAIR(62, 160, 161) PAIR(62, 162, 163) \
  F(66, 62, 9) PAIR(63, 240, 241) PAIR(63, 242, 243) F(66, 63, 9)                                                      \
  F8(40) F8(48) F8(56) F8(64) F(40, 8, 9) F(41, 8, 9) F(42, 8, 9) F(43, 8, 9) F(44, 8, 9) F(45, 8, 9) F(46, 8, 9) SLOW4 TR2
KERNEL(k_row_batched, ROW_BATCHED)
KERNEL(k_row_serial, ROW_SERIAL)
// the batched row without its LDS reads, and with the reads alone
#define ROW_NOLDS F8(56) F8(64) F8(56) F8(64) F(70, 8, 9) F(71, 10, 11)                 \
  F8(40) F8(48) F8(56) F8(64) F8(40) F8(48) F(56, 8, 9) F(57, 8, 9) F(58, 8, 9) SLOW4 TR2
KERNEL(k_row_nolds, ROW_NOLDS)
KERNEL(k_row_ldsonly, TAPS(0) "s_waitcnt lgkmcnt(0)\n")

typedef void (*kern_t)(unsigned long long*, int);
struct Entry { const char* name; kern_t fn; int ninstr; int nvalu; };
#define E(n, c, v) {#n, n, c, v}

// ---- the driver's view of the shader clock, sampled while a kernel runs ----
static std::string find_hwmon_freq() {
  for (int card = 0; card < 64; ++card) {
    const std::string base = "/sys/class/drm/card" + std::to_string(card) + "/device/hwmon";
    DIR* d = opendir(base.c_str());
    if (!d) continue;
    while (dirent* e = readdir(d)) {
      if (strncmp(e->d_name, "hwmon", 5) != 0) continue;
      const std::string f = base + "/" + e->d_name + "/freq1_input";
      std::ifstream in(f);
      long long v;
      if (in >> v) { closedir(d); return f; }
    }
    closedir(d);
  }
  return "";
}
static std::string find_dpm_sclk() {
  for (int card = 0; card < 64; ++card) {
    const std::string f = "/sys/class/drm/card" + std::to_string(card) + "/device/pp_dpm_sclk";
    std::ifstream in(f);
    std::string line;
    if (std::getline(in, line)) return f;
  }
  return "";
}
static double read_hwmon_mhz(const std::string& f) {
  std::ifstream in(f);
  long long v = 0;
  in >> v;
  return v / 1e6;
}
static std::string read_dpm(const std::string& f) {
  std::ifstream in(f);
  std::string line, all;
  while (std::getline(in, line)) {
    if (line.find('*') != std::string::npos) all += line + " ";
  }
  return all;
}

int main(int argc, char** argv) {
  std::vector<Entry> es = {
      E(k_fmac, 32, 32), E(k_fmac_d4, 32, 32), E(k_fmac_d2, 32, 32), E(k_fmac_d1, 32, 32), E(k_fmac_samebank, 32, 32),
      E(k_fma3, 32, 32), E(k_fma3_samebank, 32, 32), E(k_add, 32, 32), E(k_mul, 32, 32), E(k_mov, 32, 32),
      E(k_fmac_sgpr, 32, 32), E(k_fmamk, 32, 32), E(k_floor, 32, 32), E(k_fract, 32, 32), E(k_max, 32, 32),
      E(k_med3, 32, 32), E(k_cvt_i32, 32, 32), E(k_rcp, 32, 32), E(k_add_u32, 32, 32), E(k_lshl, 32, 32),
      E(k_cndmask_vcc, 32, 32), E(k_pk_fma, 32, 32), E(k_pk_mul, 32, 32),
      E(k_mix_24fast_5lds_3slow, 32, 27), E(k_mix_24fast_3slow, 27, 27), E(k_ds_read_b32, 32, 0),
      E(k_row_batched, 1, 91), E(k_row_serial, 1, 91), E(k_row_nolds, 1, 91), E(k_row_ldsonly, 1, 0)};
  const char* only = argc > 1 ? argv[1] : nullptr;
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount;
  printf("device: %s, %d CUs, clockRate %d kHz (the driver's maximum)\n", prop.gcnArchName, ncu, prop.clockRate);
  const std::string hw = find_hwmon_freq(), dpm = find_dpm_sclk();
  printf("clock files: hwmon '%s'  dpm '%s'\n", hw.c_str(), dpm.c_str());
  const int rounds = 12;
  const int Ws[] = {1, 2, 3, 4, 6, 8};
  const size_t max_waves = (size_t)ncu * 8 * rounds * 4;
  unsigned long long* d;
  (void)hipMalloc(&d, sizeof(unsigned long long) * 2 * max_waves);
  std::vector<unsigned long long> h(2 * max_waves);
  printf("%-26s %2s %9s %9s %9s %8s %8s %s\n", "kernel", "W", "ns/SIMD", "cyc/SIMD", "cyc/wave", "GHz(tk)", "GHz(drv)",
         "  [ns, cycles: per wave-instruction; GHz(tk) = s_memtime / s_memrealtime inside the loop]");
  for (auto& e : es) {
    if (only && !strstr(e.name, only)) continue;
    for (int w : Ws) {
      // dynamic LDS so that exactly w blocks fit a CU (160 KB per CU; the static part is nil)
      const int lds = std::min((160 * 1024 / w) & ~2047, (int)prop.sharedMemPerBlock & ~2047);
      const int wreal = 160 * 1024 / lds;                              // blocks that really fit
      if (wreal != w) continue;
      (void)hipFuncSetAttribute((const void*)e.fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      const int blocks = ncu * w * rounds;
      // ~1 ms per block: iters such that w waves x iters x ninstr x ~1.2 ns ~ 1 ms
      const int iters = std::max(200, (int)(1.0e6 / (1.2 * w * e.ninstr)));
      hipLaunchKernelGGL(e.fn, dim3(ncu * w), dim3(256), lds, 0, d, 20);  // warm-up
      (void)hipDeviceSynchronize();
      std::atomic<bool> stop{false};
      std::vector<double> mhz;
      std::string dpm_seen;
      std::thread sampler([&] {
        while (!stop.load()) {
          if (!hw.empty()) mhz.push_back(read_hwmon_mhz(hw));
          if (!dpm.empty() && dpm_seen.empty()) dpm_seen = read_dpm(dpm);
          std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
      });
      hipEvent_t e0, e1;
      (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0, 0);
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), lds, 0, d, iters);
      (void)hipEventRecord(e1, 0);
      const hipError_t err = hipDeviceSynchronize();
      stop.store(true);
      sampler.join();
      if (err != hipSuccess) { printf("%s: launch failed: %s\n", e.name, hipGetErrorString(err)); return 1; }
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
      const size_t nw = (size_t)blocks * 4;
      (void)hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * nw, hipMemcpyDeviceToHost);
      std::vector<double> cyc(nw), ghz(nw);
      for (size_t i = 0; i < nw; ++i) {
        cyc[i] = (double)h[2 * i] / ((double)iters * e.ninstr);
        ghz[i] = (double)h[2 * i] / ((double)h[2 * i + 1] * 10.0);
      }
      std::sort(cyc.begin(), cyc.end());
      std::sort(ghz.begin(), ghz.end());
      const double ns_simd = (double)ms * 1e6 * (ncu * 4.0) / ((double)nw * iters * e.ninstr);
      double drv = 0;
      for (double m : mhz) drv += m;
      drv = mhz.empty() ? 0.0 : drv / mhz.size() / 1e3;
      printf("%-26s %2d %9.3f %9.3f %9.2f %8.3f %8.3f   %.1f ms, %d iters%s%s\n", e.name, wreal, ns_simd,
             cyc[nw / 2] / wreal, cyc[nw / 2], ghz[nw / 2], drv, ms, iters, dpm_seen.empty() ? "" : "  dpm: ", dpm_seen.c_str());
      fflush(stdout);
    }
  }
  return 0;
}
