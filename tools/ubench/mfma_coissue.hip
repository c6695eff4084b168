// Does the matrix pipe take work off the vector ALU's issue port?  The iteration kernel's 17 y-moment accumulations per
// pixel (lane = image column, one value S per lane and row, sums of S * y^k) map onto v_mfma_f32_4x4x1_16b_f32 exactly:
// 16 blocks of 4 lanes, D_b[i][j] += A_b[i] * B_b[j] with B = the lane's own S and A = (1, y, y^2, y^3) replicated in
// every block -- one MFMA = four multiply-adds per lane, accumulated in four registers.  It only pays if issuing the
// MFMA costs the SIMD less than the four v_fmac it replaces (8 cycles).  Measured here: a loop of 24 independent v_fmac
// with 0 / 4 / 8 MFMAs interleaved, and MFMAs alone, every SIMD of the chip loaded with >= 3 waves.
//
//   hipcc --offload-arch=gfx950 -O3 -o mfma_coissue mfma_coissue.hip && ./mfma_coissue
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define S_(x) #x
#define S(x) S_(x)
#define F(d, a, b) "v_fmac_f32 v[" S_(d) "], v" S_(a) ", v" S_(b) "\n"
#define M(d) "v_mfma_f32_4x4x1_16b_f32 v[" S_(d) ":" S_(d) "+3], v8, v9, v[" S_(d) ":" S_(d) "+3]\n"
#define M16(d) "v_mfma_f32_16x16x4_f32 v[" S_(d) ":" S_(d) "+3], v8, v9, v[" S_(d) ":" S_(d) "+3]\n"
#define F3(d) F(d, 10, 11) F(d + 1, 12, 13) F(d + 2, 14, 15)

#define INIT                                                                                                            \
  "v_mov_b32 v8, 1.0\n v_mov_b32 v9, 0.5\n v_mov_b32 v10, 2.0\n v_mov_b32 v11, 4.0\n v_mov_b32 v12, 1.0\n"               \
  "v_mov_b32 v13, 0.5\n v_mov_b32 v14, 2.0\n v_mov_b32 v15, 4.0\n"
#define CLOB                                                                                                            \
  "memory", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47",   \
      "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63",     \
      "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79",     \
      "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "s24"

#define KERNEL(name, body)                                                                                              \
  __global__ __launch_bounds__(256) void name(float* out, int iters) {                                                  \
    asm volatile(INIT "s_mov_b32 s24, %0\n 1:\n" body "s_sub_u32 s24, s24, 1\n s_cmp_lg_u32 s24, 0\n s_cbranch_scc1 1b\n"  \
                 "s_nop 7\n s_nop 7\n"                                                                                  \
                 :                                                                                                      \
                 : "s"(iters)                                                                                           \
                 : CLOB);                                                                                               \
    if (iters < 0) out[threadIdx.x] = 0.f;                                                                              \
  }

// 24 v_fmac into v72..v95 (24 independent destinations)
#define G(d) F3(d)
KERNEL(k_fmac24, G(72) G(75) G(78) G(81) G(84) G(87) G(90) G(93))
// + 4 MFMAs 4x4x1 (accumulators v[40:43] .. v[52:55]), one after every 6 v_fmac
KERNEL(k_fmac24_mfma4, G(72) G(75) M(40) G(78) G(81) M(44) G(84) G(87) M(48) G(90) G(93) M(52))
// + 8 MFMAs (v[40:43] .. v[68:71]), one after every 3 v_fmac
KERNEL(k_fmac24_mfma8, G(72) M(40) G(75) M(44) G(78) M(48) G(81) M(52) G(84) M(56) G(87) M(60) G(90) M(64) G(93) M(68))
// 8 MFMAs alone
KERNEL(k_mfma8, M(40) M(44) M(48) M(52) M(56) M(60) M(64) M(68))
// 24 v_fmac + 8 more v_fmac (what 8 MFMAs' worth of issue would have to beat: 32 multiply-adds per lane = 8 MFMAs x 4)
KERNEL(k_fmac56, G(72) G(75) G(78) G(81) G(84) G(87) G(90) G(93) G(40) G(43) G(46) G(49) G(52) G(55) G(58) G(61) F(64, 10, 11) F(65, 12, 13)
           F(66, 14, 15) F(67, 10, 11) F(68, 12, 13) F(69, 14, 15) F(70, 10, 11) F(71, 12, 13))
// the 16x16x4 form (one issue = 16 multiply-adds per lane, 32 cycles of the matrix pipe): 24 v_fmac + 2
KERNEL(k_fmac24_mfma16x2, G(72) G(75) G(78) G(81) M16(40) G(84) G(87) G(90) G(93) M16(44))
KERNEL(k_mfma16x8, M16(40) M16(44) M16(48) M16(52) M16(56) M16(60) M16(64) M16(68))

typedef void (*kern_t)(float*, int);
struct E { const char* name; kern_t fn; int fmac, mfma; };

int main() {
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount;
  float* d;
  (void)hipMalloc(&d, 4096);
  std::vector<E> es = {{"k_fmac24", k_fmac24, 24, 0},           {"k_fmac24_mfma4", k_fmac24_mfma4, 24, 4},
                       {"k_fmac24_mfma8", k_fmac24_mfma8, 24, 8}, {"k_mfma8", k_mfma8, 0, 8},
                       {"k_fmac56", k_fmac56, 56, 0},           {"k_fmac24_mfma16x2", k_fmac24_mfma16x2, 24, 2},
                       {"k_mfma16x8", k_mfma16x8, 0, 8}};
  printf("device %s, %d CUs; blocks of 4 waves (one per SIMD), 5 blocks per CU resident, 8 rounds\n", prop.gcnArchName, ncu);
  printf("%-20s %6s %6s %12s %14s\n", "kernel", "v_fmac", "mfma", "ns/body/SIMD", "(body = one loop pass of one wave)");
  for (auto& e : es) {
    const int per_cu = 5, rounds = 8, iters = 40000;
    int occ = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)e.fn, 256, 0);
    const int blocks = ncu * per_cu * rounds;
    hipLaunchKernelGGL(e.fn, dim3(ncu * per_cu), dim3(256), 0, 0, d, 100);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1, 0);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s failed\n", e.name); return 1; }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // every SIMD ran blocks / ncu waves one after / beside the other, `iters` bodies each
    const double ns = (double)ms * 1e6 / ((double)blocks / ncu * iters);
    printf("%-20s %6d %6d %12.2f   occupancy %d blocks/CU, %.1f ms\n", e.name, e.fmac, e.mfma, ns, occ, ms);
    fflush(stdout);
  }
  return 0;
}
