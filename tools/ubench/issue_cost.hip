// Issue cost of the instruction forms the iteration kernel is made of (gfx950), measured in shader cycles with
// s_memtime: every wave runs a loop of kBody independent instructions; W waves per SIMD on every CU.
//   hipcc --offload-arch=gfx950 -O3 -o issue_cost issue_cost.hip && ./issue_cost
// Output: cycles per wave-instruction seen by ONE wave, and per SIMD (= that / W) for W = 1, 2, 4, 6, 8.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)

// The loop body is written with explicit registers so that operand banks are under control (bank = index % 4).
// v[40..71] destinations, v[8..39] sources; s[20..27] scalar sources.
#define KERNEL(name, body, ninstr)                                                                          \
  __global__ __launch_bounds__(256) void name(unsigned long long* out, int iters) {                         \
    __shared__ float lds[4096];                                                                             \
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)i;                                        \
    __syncthreads();                                                                                        \
    unsigned long long t0, t1;                                                                              \
    const unsigned la = (threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 4096;                                 \
    const unsigned lb = (threadIdx.x & 63) * 8 + (threadIdx.x >> 6) * 4096;                                 \
    asm volatile(                                                                                           \
        "v_mov_b32 v8, 1.0\n v_mov_b32 v9, 0.5\n v_mov_b32 v10, 2.0\n v_mov_b32 v11, 4.0\n"                 \
        "v_mov_b32 v12, 1.0\n v_mov_b32 v13, 0.5\n v_mov_b32 v14, 2.0\n v_mov_b32 v15, 4.0\n"               \
        "v_mov_b32 v16, 1.0\n v_mov_b32 v17, 0.5\n v_mov_b32 v18, 2.0\n v_mov_b32 v19, 4.0\n"               \
        "v_mov_b32 v20, 1.0\n v_mov_b32 v21, 0.5\n v_mov_b32 v22, 2.0\n v_mov_b32 v23, 4.0\n"               \
        "v_mov_b32 v24, 1.0\n v_mov_b32 v25, 0.5\n v_mov_b32 v26, 2.0\n v_mov_b32 v27, 4.0\n"               \
        "v_mov_b32 v28, 1.0\n v_mov_b32 v29, 0.5\n v_mov_b32 v30, 2.0\n v_mov_b32 v31, 4.0\n"               \
        "v_mov_b32 v32, 1.0\n v_mov_b32 v33, 0.5\n v_mov_b32 v34, 2.0\n v_mov_b32 v35, 4.0\n"               \
        "v_mov_b32 v36, 1.0\n v_mov_b32 v37, 0.5\n v_mov_b32 v38, 2.0\n v_mov_b32 v39, 4.0\n"               \
        "v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n"                       \
        "v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n"                       \
        "v_mov_b32 v48, 0\n v_mov_b32 v49, 0\n v_mov_b32 v50, 0\n v_mov_b32 v51, 0\n"                       \
        "v_mov_b32 v52, 0\n v_mov_b32 v53, 0\n v_mov_b32 v54, 0\n v_mov_b32 v55, 0\n"                       \
        "v_mov_b32 v56, 0\n v_mov_b32 v57, 0\n v_mov_b32 v58, 0\n v_mov_b32 v59, 0\n"                       \
        "v_mov_b32 v60, 0\n v_mov_b32 v61, 0\n v_mov_b32 v62, 0\n v_mov_b32 v63, 0\n"                       \
        "v_mov_b32 v64, 0\n v_mov_b32 v65, 0\n v_mov_b32 v66, 0\n v_mov_b32 v67, 0\n"                       \
        "v_mov_b32 v68, 0\n v_mov_b32 v69, 0\n v_mov_b32 v70, 0\n v_mov_b32 v71, 0\n"                       \
        "v_mov_b32 v72, %2\n v_mov_b32 v73, %3\n v_lshlrev_b32 v74, 1, %3\n v_and_b32 v74, 0x1ff0, v74\n"                                                           \
        "s_mov_b32 s20, 1.0\n s_mov_b32 s21, 0.5\n s_mov_b32 s22, 2.0\n s_mov_b32 s23, 4.0\n"               \
        "s_mov_b32 s24, %4\n s_mov_b32 s26, 0x55555555\n s_mov_b32 s27, 0x33333333\n"                                                                               \
        "s_mov_b32 vcc_lo, 0x55555555\n s_mov_b32 vcc_hi, 0x55555555\n"                                                               \
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n s_barrier\n"                                                       \
        "s_memtime %0\n s_waitcnt lgkmcnt(0)\n"                                                             \
        "1:\n" body                                                                                         \
        "s_sub_u32 s24, s24, 1\n s_cmp_lg_u32 s24, 0\n s_cbranch_scc1 1b\n"                                 \
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n"                                                                   \
        "s_memtime %1\n s_waitcnt lgkmcnt(0)\n"                                                             \
        : "=&s"(t0), "=&s"(t1)                                                                              \
        : "v"(la), "v"(lb), "s"(iters)                                                                      \
        : "memory", "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "v8", "v9", "v10", "v11", "v12", \
          "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", \
          "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", \
          "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", \
          "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75");               \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                        \
    if (lds[threadIdx.x] < -1.f) out[0] = 0;                                                                \
  }

// ---- 16 independent instructions per body (dst v40+i; sources rotate through v8..v39) ----
#define B16(f) f(40, 8, 9, 10) f(41, 11, 12, 13) f(42, 14, 15, 16) f(43, 17, 18, 19) f(44, 20, 21, 22) f(45, 23, 24, 25) \
  f(46, 26, 27, 28) f(47, 29, 30, 31) f(48, 32, 33, 34) f(49, 35, 36, 37) f(50, 38, 39, 8) f(51, 9, 10, 11) f(52, 12, 13, 14) \
  f(53, 15, 16, 17) f(54, 18, 19, 20) f(55, 21, 22, 23)
// same-bank sources: all three source indices congruent mod 4
#define B16S(f) f(40, 8, 12, 16) f(41, 9, 13, 17) f(42, 10, 14, 18) f(43, 11, 15, 19) f(44, 20, 24, 28) f(45, 21, 25, 29) \
  f(46, 22, 26, 30) f(47, 23, 27, 31) f(48, 32, 36, 8) f(49, 33, 37, 9) f(50, 34, 38, 10) f(51, 35, 39, 11) f(52, 12, 16, 20) \
  f(53, 13, 17, 21) f(54, 14, 18, 22) f(55, 15, 19, 23)
// register pairs: dst v[40+2i:41+2i], sources even-aligned pairs
#define B16P(f) f(40, 8, 10, 12) f(42, 14, 16, 18) f(44, 20, 22, 24) f(46, 26, 28, 30) f(48, 32, 34, 36) f(50, 38, 8, 10) \
  f(52, 12, 14, 16) f(54, 18, 20, 22) f(56, 24, 26, 28) f(58, 30, 32, 34) f(60, 36, 38, 8) f(62, 10, 12, 14) f(64, 16, 18, 20) \
  f(66, 22, 24, 26) f(68, 28, 30, 32) f(70, 34, 36, 38)

#define S_(x) #x
#define FMA3(d, a, b, c) "v_fma_f32 v" S_(d) ", v" S_(a) ", v" S_(b) ", v" S_(c) "\n"
#define FMAC(d, a, b, c) "v_fmac_f32 v" S_(d) ", v" S_(a) ", v" S_(b) "\n"
#define FMA2S(d, a, b, c) "v_fma_f32 v" S_(d) ", s20, v" S_(b) ", v" S_(c) "\n"
#define FMA1S(d, a, b, c) "v_fma_f32 v" S_(d) ", s20, v" S_(b) ", 2.0\n"
#define FMACS(d, a, b, c) "v_fmac_f32 v" S_(d) ", s20, v" S_(b) "\n"
#define FMAMK(d, a, b, c) "v_fmamk_f32 v" S_(d) ", v" S_(a) ", 0x3fc00000, v" S_(b) "\n"
#define FMAK1(d, a, b, c) "v_fma_f32 v" S_(d) ", v" S_(a) ", v" S_(b) ", 1.0\n"
#define MUL2(d, a, b, c) "v_mul_f32 v" S_(d) ", v" S_(a) ", v" S_(b) "\n"
#define ADD2(d, a, b, c) "v_add_f32 v" S_(d) ", v" S_(a) ", v" S_(b) "\n"
#define MULS(d, a, b, c) "v_mul_f32 v" S_(d) ", s20, v" S_(b) "\n"
#define PKFMA(d, a, b, c) "v_pk_fma_f32 v[" S_(d) ":" S_(d) "+1], v[" S_(a) ":" S_(a) "+1], v[" S_(b) ":" S_(b) "+1], v[" S_(c) ":" S_(c) "+1]\n"
#define PKFMAC(d, a, b, c) "v_pk_fma_f32 v[" S_(d) ":" S_(d) "+1], v[" S_(a) ":" S_(a) "+1], v[" S_(b) ":" S_(b) "+1], v[" S_(d) ":" S_(d) "+1]\n"
#define PKMUL(d, a, b, c) "v_pk_mul_f32 v[" S_(d) ":" S_(d) "+1], v[" S_(a) ":" S_(a) "+1], v[" S_(b) ":" S_(b) "+1]\n"
#define PKADD(d, a, b, c) "v_pk_add_f32 v[" S_(d) ":" S_(d) "+1], v[" S_(a) ":" S_(a) "+1], v[" S_(b) ":" S_(b) "+1]\n"
#define PKFMAS(d, a, b, c) "v_pk_fma_f32 v[" S_(d) ":" S_(d) "+1], v[" S_(a) ":" S_(a) "+1], s[20:21], v[" S_(d) ":" S_(d) "+1]\n"
#define RCP(d, a, b, c) "v_rcp_f32 v" S_(d) ", v" S_(a) "\n"
#define FLOOR(d, a, b, c) "v_floor_f32 v" S_(d) ", v" S_(a) "\n"
#define FRACT(d, a, b, c) "v_fract_f32 v" S_(d) ", v" S_(a) "\n"
#define CVTI(d, a, b, c) "v_cvt_i32_f32 v" S_(d) ", v" S_(a) "\n"
#define CVTFLR(d, a, b, c) "v_cvt_flr_i32_f32 v" S_(d) ", v" S_(a) "\n"
#define CNDM(d, a, b, c) "v_cndmask_b32 v" S_(d) ", v" S_(a) ", v" S_(b) ", vcc\n"
#define MED3(d, a, b, c) "v_med3_f32 v" S_(d) ", v" S_(a) ", v" S_(b) ", v" S_(c) "\n"
#define MADU24(d, a, b, c) "v_mad_u32_u24 v" S_(d) ", v" S_(a) ", v" S_(b) ", v" S_(c) "\n"
#define LSHLADD(d, a, b, c) "v_lshl_add_u32 v" S_(d) ", v" S_(a) ", 2, v" S_(c) "\n"
#define ADDU(d, a, b, c) "v_add_u32 v" S_(d) ", v" S_(a) ", v" S_(b) "\n"
#define DPPMOV(d, a, b, c) "v_mov_b32_dpp v" S_(d) ", v" S_(a) " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define DPPSUB(d, a, b, c) "v_sub_f32_dpp v" S_(d) ", v" S_(a) ", v" S_(b) " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define DPPROW(d, a, b, c) "v_mov_b32_dpp v" S_(d) ", v" S_(a) " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define RDLANE(d, a, b, c) "v_readlane_b32 s25, v" S_(a) ", 5\n"
#define CVTUB(d, a, b, c) "v_cvt_f32_ubyte1 v" S_(d) ", v" S_(a) "\n"
#define PERM(d, a, b, c) "v_perm_b32 v" S_(d) ", v" S_(a) ", v" S_(b) ", v" S_(c) "\n"
#define ALIGNB(d, a, b, c) "v_alignbyte_b32 v" S_(d) ", v" S_(a) ", v" S_(b) ", v" S_(c) "\n"
#define CMPU(d, a, b, c) "v_cmp_ge_u32 vcc, v" S_(a) ", v" S_(b) "\n"
#define FMAMIX(d, a, b, c) "v_fma_mix_f32 v" S_(d) ", v" S_(a) ", v" S_(b) ", v" S_(c) " op_sel_hi:[0,1,0]\n"

#define MULE64(d, a, b, c) "v_mul_f32_e64 v" S_(d) ", v" S_(a) ", v" S_(b) "\n"
#define MULC(d, a, b, c) "v_mul_f32 v" S_(d) ", 0.5, v" S_(b) "\n"
#define ADDC(d, a, b, c) "v_add_f32 v" S_(d) ", 1.0, v" S_(b) "\n"
#define SUBF(d, a, b, c) "v_sub_f32 v" S_(d) ", v" S_(a) ", v" S_(b) "\n"
#define MAXF(d, a, b, c) "v_max_f32 v" S_(d) ", v" S_(a) ", v" S_(b) "\n"
#define CNDS(d, a, b, c) "v_cndmask_b32_e64 v" S_(d) ", v" S_(a) ", v" S_(b) ", s[26:27]\n"
#define CMPE64(d, a, b, c) "v_cmp_ge_u32_e64 s[26:27], v" S_(a) ", v" S_(b) "\n"
#define CMPF(d, a, b, c) "v_cmp_le_f32 vcc, v" S_(a) ", v" S_(b) "\n"
#define LSHL(d, a, b, c) "v_lshlrev_b32 v" S_(d) ", 2, v" S_(a) "\n"
#define ANDB(d, a, b, c) "v_and_b32 v" S_(d) ", v" S_(a) ", v" S_(b) "\n"
#define MULI24(d, a, b, c) "v_mul_i32_i24 v" S_(d) ", v" S_(a) ", v" S_(b) "\n"
#define MADI24(d, a, b, c) "v_mad_i32_i24 v" S_(d) ", v" S_(a) ", v" S_(b) ", v" S_(c) "\n"
#define CVTFI(d, a, b, c) "v_cvt_f32_i32 v" S_(d) ", v" S_(a) "\n"
#define RSQ(d, a, b, c) "v_rsq_f32 v" S_(d) ", v" S_(a) "\n"
#define MOV(d, a, b, c) "v_mov_b32 v" S_(d) ", v" S_(a) "\n"
#define FMANEG(d, a, b, c) "v_fma_f32 v" S_(d) ", -v" S_(a) ", v" S_(b) ", v" S_(c) "\n"
#define FMACC(d, a, b, c) "v_fmac_f32 v" S_(d) ", 0.5, v" S_(b) "\n"
#define CNDMIX(d, a, b, c) "v_cndmask_b32 v" S_(d) ", v" S_(a) ", v" S_(b) ", vcc\n v_mul_f32 v" S_(d) ", v" S_(a) ", v" S_(b) "\n v_add_f32 v" S_(d) ", v" S_(a) ", v" S_(b) "\n v_mul_f32 v" S_(d) ", v" S_(b) ", v" S_(c) "\n"
#define DSR32ADJ(d, a, b, c) "ds_read_b32 v" S_(d) ", v72 offset:" S_(a) "*4\n"
#define DSR32CONF(d, a, b, c) "ds_read_b32 v" S_(d) ", v73 offset:" S_(a) "*16\n"
#define DSR128(d, a, b, c) "ds_read_b128 v[" S_(d) ":" S_(d) "+3], v74 offset:" S_(a) "*16\n"
// LDS reads into v40.. (v72 = lane*4 + wave*4096: conflict-free dwords; v73 = lane*8 + ...: for 8-byte reads)
#define DSR32(d, a, b, c) "ds_read_b32 v" S_(d) ", v72 offset:" S_(a) "*16\n"
#define DSR2(d, a, b, c) "ds_read2_b32 v[" S_(d) ":" S_(d) "+1], v72 offset0:" S_(a) " offset1:" S_(a) "+1\n"
// lane stride 4 bytes (v72): every other lane's 8-/16-byte read is misaligned, neighbouring lanes overlap
#define DSR64U(d, a, b, c) "ds_read_b64 v[" S_(d) ":" S_(d) "+1], v72 offset:" S_(a) "*16+4\n"
#define DSR128U(d, a, b, c) "ds_read_b128 v[" S_(d) ":" S_(d) "+3], v72 offset:" S_(a) "*16+4\n"
#define DSR64(d, a, b, c) "ds_read_b64 v[" S_(d) ":" S_(d) "+1], v73 offset:" S_(a) "*16\n"

KERNEL(k_fma3, B16(FMA3), 16)
KERNEL(k_fma3_samebank, B16S(FMA3), 16)
KERNEL(k_fmac, B16(FMAC), 16)
KERNEL(k_fmac_samebank, B16S(FMAC), 16)
KERNEL(k_fma_2v1s, B16(FMA2S), 16)
KERNEL(k_fma_1v1s1c, B16(FMA1S), 16)
KERNEL(k_fmac_s, B16(FMACS), 16)
KERNEL(k_fmamk, B16(FMAMK), 16)
KERNEL(k_fma_2v1c, B16(FMAK1), 16)
KERNEL(k_mul2, B16(MUL2), 16)
KERNEL(k_add2, B16(ADD2), 16)
KERNEL(k_mul_s, B16(MULS), 16)
KERNEL(k_pk_fma, B16P(PKFMA), 16)
KERNEL(k_pk_fmac, B16P(PKFMAC), 16)
KERNEL(k_pk_mul, B16P(PKMUL), 16)
KERNEL(k_pk_add, B16P(PKADD), 16)
KERNEL(k_pk_fma_s, B16P(PKFMAS), 16)
KERNEL(k_rcp, B16(RCP), 16)
KERNEL(k_floor, B16(FLOOR), 16)
KERNEL(k_fract, B16(FRACT), 16)
KERNEL(k_cvt_i32, B16(CVTI), 16)
KERNEL(k_cvt_flr, B16(CVTFLR), 16)
KERNEL(k_cndmask, B16(CNDM), 16)
KERNEL(k_med3, B16(MED3), 16)
KERNEL(k_mad_u24, B16(MADU24), 16)
KERNEL(k_lshl_add, B16(LSHLADD), 16)
KERNEL(k_add_u32, B16(ADDU), 16)
KERNEL(k_dpp_wave_shr, B16(DPPMOV), 16)
KERNEL(k_dpp_sub, B16(DPPSUB), 16)
KERNEL(k_dpp_row_shr, B16(DPPROW), 16)
KERNEL(k_readlane, B16(RDLANE), 16)
KERNEL(k_cvt_ubyte, B16(CVTUB), 16)
KERNEL(k_perm, B16(PERM), 16)
KERNEL(k_alignbyte, B16(ALIGNB), 16)
KERNEL(k_cmp_u32, B16(CMPU), 16)
KERNEL(k_fma_mix, B16(FMAMIX), 16)
KERNEL(k_ds_read_b32, B16(DSR32) "s_waitcnt lgkmcnt(0)\n", 16)
KERNEL(k_ds_read2_b32, B16P(DSR2) "s_waitcnt lgkmcnt(0)\n", 16)
KERNEL(k_ds_read_b64, B16P(DSR64) "s_waitcnt lgkmcnt(0)\n", 16)
// mixes: VALU next to LDS reads (do they overlap?)
KERNEL(k_mix_16fmac_16ds32, B16(FMAC) B16(DSR32) "s_waitcnt lgkmcnt(0)\n", 32)
KERNEL(k_mix_32fmac_16ds32, B16(FMAC) B16S(FMAC) B16(DSR32) "s_waitcnt lgkmcnt(0)\n", 48)
KERNEL(k_mix_64fmac_16ds2, B16(FMAC) B16S(FMAC) B16(FMAC) B16S(FMAC) B16P(DSR2) "s_waitcnt lgkmcnt(0)\n", 80)
KERNEL(k_mix_16fmac_16mul2, B16(FMAC) B16(MUL2), 32)
KERNEL(k_mix_16fma3_16mul2, B16(FMA3) B16(MUL2), 32)
KERNEL(k_mix_16pkfma_16mul2, B16P(PKFMA) B16(MUL2), 32)


KERNEL(k_mul_e64, B16(MULE64), 16)
KERNEL(k_mul_const, B16(MULC), 16)
KERNEL(k_add_const, B16(ADDC), 16)
KERNEL(k_sub, B16(SUBF), 16)
KERNEL(k_max, B16(MAXF), 16)
KERNEL(k_cndmask_sgpr, B16(CNDS), 16)
KERNEL(k_cmp_e64, B16(CMPE64), 16)
KERNEL(k_cmp_f32, B16(CMPF), 16)
KERNEL(k_lshl, B16(LSHL), 16)
KERNEL(k_and, B16(ANDB), 16)
KERNEL(k_mul_i24, B16(MULI24), 16)
KERNEL(k_mad_i24, B16(MADI24), 16)
KERNEL(k_cvt_f32_i32, B16(CVTFI), 16)
KERNEL(k_rsq, B16(RSQ), 16)
KERNEL(k_mov, B16(MOV), 16)
KERNEL(k_fma_neg, B16(FMANEG), 16)
KERNEL(k_fmac_const, B16(FMACC), 16)
KERNEL(k_cndmask_1in4, B16(CNDMIX), 64)
KERNEL(k_ds_read_b32_adj, B16(DSR32ADJ) "s_waitcnt lgkmcnt(0)\n", 16)
KERNEL(k_ds_read_b32_2way, B16(DSR32CONF) "s_waitcnt lgkmcnt(0)\n", 16)
KERNEL(k_ds_read_b128, REP4(DSR128(40, 1, 0, 0) DSR128(44, 2, 0, 0) DSR128(48, 3, 0, 0) DSR128(52, 4, 0, 0)) "s_waitcnt lgkmcnt(0)\n", 16)

KERNEL(k_ds_read_b64_lane4B, B16P(DSR64U) "s_waitcnt lgkmcnt(0)\n", 16)
KERNEL(k_ds_read_b128_lane4B, REP4(DSR128U(40, 1, 0, 0) DSR128U(44, 2, 0, 0) DSR128U(48, 3, 0, 0) DSR128U(52, 4, 0, 0)) "s_waitcnt lgkmcnt(0)\n", 16)

// functional check of the misaligned forms: out[lane] = the dwords a lane got from byte address 4*lane+4
__global__ void k_unaligned_check(float* out) {
  __shared__ float lds[256];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (float)i;
  __syncthreads();
  const unsigned a = threadIdx.x * 4 + 4;
  float x0, x1, y0, y1, y2, y3;
  asm volatile("ds_read_b64 v[20:21], %6\n ds_read_b128 v[24:27], %6\n s_waitcnt lgkmcnt(0)\n"
               "v_mov_b32 %0, v20\n v_mov_b32 %1, v21\n v_mov_b32 %2, v24\n v_mov_b32 %3, v25\n v_mov_b32 %4, v26\n v_mov_b32 %5, v27\n"
               : "=v"(x0), "=v"(x1), "=v"(y0), "=v"(y1), "=v"(y2), "=v"(y3) : "v"(a)
               : "v20", "v21", "v24", "v25", "v26", "v27", "memory");
  float* o = out + threadIdx.x * 6;
  o[0] = x0; o[1] = x1; o[2] = y0; o[3] = y1; o[4] = y2; o[5] = y3;
}

typedef void (*kern_t)(unsigned long long*, int);
struct Entry { const char* name; kern_t fn; int ninstr; };
#define E(n, c) {#n, n, c}

int main(int argc, char** argv) {
  std::vector<Entry> es = {
      E(k_fma3, 16), E(k_fma3_samebank, 16), E(k_fmac, 16), E(k_fmac_samebank, 16), E(k_fma_2v1s, 16), E(k_fma_1v1s1c, 16),
      E(k_fmac_s, 16), E(k_fmamk, 16), E(k_fma_2v1c, 16), E(k_mul2, 16), E(k_add2, 16), E(k_mul_s, 16), E(k_pk_fma, 16),
      E(k_pk_fmac, 16), E(k_pk_mul, 16), E(k_pk_add, 16), E(k_pk_fma_s, 16), E(k_rcp, 16), E(k_floor, 16), E(k_fract, 16),
      E(k_cvt_i32, 16), E(k_cvt_flr, 16), E(k_cndmask, 16), E(k_med3, 16), E(k_mad_u24, 16), E(k_lshl_add, 16),
      E(k_add_u32, 16), E(k_dpp_wave_shr, 16), E(k_dpp_sub, 16), E(k_dpp_row_shr, 16), E(k_readlane, 16), E(k_cvt_ubyte, 16),
      E(k_perm, 16), E(k_alignbyte, 16), E(k_cmp_u32, 16), E(k_fma_mix, 16), E(k_ds_read_b32, 16), E(k_ds_read2_b32, 16),
      E(k_ds_read_b64, 16), E(k_mix_16fmac_16ds32, 32), E(k_mix_32fmac_16ds32, 48), E(k_mix_64fmac_16ds2, 80),
      E(k_mix_16fmac_16mul2, 32), E(k_mix_16fma3_16mul2, 32), E(k_mix_16pkfma_16mul2, 32),
      E(k_mul_e64, 16), E(k_mul_const, 16), E(k_add_const, 16), E(k_sub, 16), E(k_max, 16), E(k_cndmask_sgpr, 16),
      E(k_cmp_e64, 16), E(k_cmp_f32, 16), E(k_lshl, 16), E(k_and, 16), E(k_mul_i24, 16), E(k_mad_i24, 16),
      E(k_cvt_f32_i32, 16), E(k_rsq, 16), E(k_mov, 16), E(k_fma_neg, 16), E(k_fmac_const, 16), E(k_cndmask_1in4, 64),
      E(k_ds_read_b32_adj, 16), E(k_ds_read_b32_2way, 16), E(k_ds_read_b128, 16),
      E(k_ds_read_b64_lane4B, 16), E(k_ds_read_b128_lane4B, 16)};
  {
    float* dc;
    (void)hipMalloc(&dc, 64 * 6 * sizeof(float));
    k_unaligned_check<<<1, 64>>>(dc);
    std::vector<float> hc(64 * 6);
    (void)hipMemcpy(hc.data(), dc, hc.size() * sizeof(float), hipMemcpyDeviceToHost);
    int bad64 = 0, bad128 = 0;
    for (int l = 0; l < 64; ++l) {
      for (int j = 0; j < 2; ++j) bad64 += hc[l * 6 + j] != (float)(l + 1 + j);
      for (int j = 0; j < 4; ++j) bad128 += hc[l * 6 + 2 + j] != (float)(l + 1 + j);
    }
    printf("misaligned LDS reads (lane stride 4 B): ds_read_b64 %s, ds_read_b128 %s   lane1: %g %g | %g %g %g %g\n",
           bad64 ? "WRONG" : "exact", bad128 ? "WRONG" : "exact", hc[6], hc[7], hc[8], hc[9], hc[10], hc[11]);
    (void)hipFree(dc);
  }
  if (argc > 1) es.erase(es.begin(), es.begin() + atoi(argv[1]));
  const int iters = 4000;
  unsigned long long* d;
  (void)hipMalloc(&d, sizeof(unsigned long long) * 4 * 256 * 8);
  std::vector<unsigned long long> h(4 * 256 * 8);
  printf("%-24s", "cycles/instr: wave|SIMD");
  const int Ws[] = {2, 4, 5, 6};
  for (int w : Ws) printf("        W=%d           ", w);
  printf("\n");
  for (auto& e : es) {
    printf("%-24s", e.name);
    for (int w : Ws) {
      const int blocks = 256 * w;
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d, 10);
      hipEvent_t e0, e1;
      (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0, 0);
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d, iters);
      (void)hipEventRecord(e1, 0);
      if (hipDeviceSynchronize() != hipSuccess) { printf(" launch failed\n"); return 1; }
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      const double ns_per = ms * 1e6 / ((double)iters * e.ninstr * w);  // wall ns per wave-instruction per SIMD
      (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
      (void)hipMemcpy(h.data(), d, sizeof(unsigned long long) * 4 * blocks, hipMemcpyDeviceToHost);
      std::vector<unsigned long long> v(h.begin(), h.begin() + 4 * blocks);
      std::sort(v.begin(), v.end());
      const double med = (double)v[v.size() / 2] / ((double)iters * e.ninstr);
      printf("  %6.2f|%5.2f|%4.2fns ", med, med / w, ns_per);
    }
    printf("\n");
    fflush(stdout);
  }
  return 0;
}
