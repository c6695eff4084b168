// Instantiations of the fused iteration kernel for ONE transform type (-DICA_INST_TT=1..5): 2 channel counts x 2 image
// storages (float32, uint8) x (L2, L2-first, Hessian + 4 robust error functions).  build.py compiles this file five times, in parallel.
#include "ica_iterate.hpp"
#include "ica_iterate_launch.hpp"

#ifndef ICA_INST_TT
#error "compile with -DICA_INST_TT=<transform type 1..5>"
#endif

namespace ica {
namespace {

template <int TT, int MODE, int RT>
void launch_c(int channels, unsigned blocks, hipStream_t st, const LevelArgs& a) {
  if (a.u8) {  // this level's images are uint8 in HBM (level 0 of an ICA_STORAGE_U8 problem)
    if (channels == 1) hipLaunchKernelGGL((ica_iterate_kernel<TT, MODE, 1, RT, true>), dim3(blocks), dim3(IterCfg<1, true>::block), 0, st, a);
    else hipLaunchKernelGGL((ica_iterate_kernel<TT, MODE, 3, RT, true>), dim3(blocks), dim3(IterCfg<3, true>::block), 0, st, a);
  } else {
    if (channels == 1) hipLaunchKernelGGL((ica_iterate_kernel<TT, MODE, 1, RT, false>), dim3(blocks), dim3(IterCfg<1, false>::block), 0, st, a);
    else hipLaunchKernelGGL((ica_iterate_kernel<TT, MODE, 3, RT, false>), dim3(blocks), dim3(IterCfg<3, false>::block), 0, st, a);
  }
}

template <int TT>
bool launch_tt(int mode, int channels, int robust_type, unsigned blocks, hipStream_t st, const LevelArgs& a) {
  switch (mode) {
    case kModeL2: launch_c<TT, kModeL2, ICA_QUADRATIC>(channels, blocks, st, a); break;
    case kModeL2First: launch_c<TT, kModeL2First, ICA_QUADRATIC>(channels, blocks, st, a); break;
    case kModeHessian: launch_c<TT, kModeHessian, ICA_QUADRATIC>(channels, blocks, st, a); break;
    default:
      // the robust error function is a template argument of the kernel (no branches in its pixel loop)
      switch (robust_type) {
        case ICA_TRUNCATED_QUADRATIC: launch_c<TT, kModeRobust, ICA_TRUNCATED_QUADRATIC>(channels, blocks, st, a); break;
        case ICA_GERMAN_MCCLURE: launch_c<TT, kModeRobust, ICA_GERMAN_MCCLURE>(channels, blocks, st, a); break;
        case ICA_LORENTZIAN: launch_c<TT, kModeRobust, ICA_LORENTZIAN>(channels, blocks, st, a); break;
        case ICA_CHARBONNIER: launch_c<TT, kModeRobust, ICA_CHARBONNIER>(channels, blocks, st, a); break;
        default: return false;  // QUADRATIC has no robust instantiation: the caller reports it (nothing was launched)
      }
  }
  return true;
}

}  // namespace

#define ICA_CAT2(a, b) a##b
#define ICA_CAT(a, b) ICA_CAT2(a, b)
bool ICA_CAT(launch_iterate_tt, ICA_INST_TT)(int mode, int channels, int robust_type, unsigned blocks, hipStream_t st,
                                             const LevelArgs& a) {
  return launch_tt<ICA_INST_TT>(mode, channels, robust_type, blocks, st, a);
}

}  // namespace ica

#ifdef ICA_STAMP
// diagnostic builds: copy out and reset the per-wave records of this unit's kernels (see ica_iterate.hpp):
// out = [kStampWaves][16] unsigned long long (the waves of the last launch; a record with [6] == 0 is unused)
#define ICA_STAMP_FN2(n) ica_debug_stamps_tt##n
#define ICA_STAMP_FN(n) ICA_STAMP_FN2(n)
extern "C" __attribute__((visibility("default"))) int ICA_STAMP_FN(ICA_INST_TT)(unsigned long long* out, int max_records) {
  if (max_records < ica::kStampWaves) return ica::kStampWaves;
  const size_t bytes = (size_t)ica::kStampWaves * 16 * sizeof(unsigned long long);
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ica::ica_stamp_acc), bytes) != hipSuccess) return -2;
  void* dev = nullptr;
  if (hipGetSymbolAddress(&dev, HIP_SYMBOL(ica::ica_stamp_acc)) != hipSuccess) return -3;
  if (hipMemset(dev, 0, bytes) != hipSuccess) return -4;
  return 0;
}
#endif
