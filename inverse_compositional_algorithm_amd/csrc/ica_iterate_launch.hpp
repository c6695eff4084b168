// Launch interface of the fused iteration kernel (ica_iterate.hpp).  The kernel is instantiated per transform type in
// separate translation units (ica_iterate_inst.hip, compiled once per type, in parallel) so that the library builds in
// seconds; ica_api.hip only sees these five functions.
#pragma once
#include <hip/hip_runtime.h>

#include "ica_common.hpp"

namespace ica {

// mode: kModeL2 / kModeRobust / kModeHessian / kModeL2First; channels 1 or 3; robust_type: the ICA_* error function
// (used by kModeRobust only: a template argument of the kernel).  `blocks` = 1-D grid (see launch_iterate).
void launch_iterate_tt1(int mode, int channels, int robust_type, unsigned blocks, hipStream_t st, const LevelArgs& a);
void launch_iterate_tt2(int mode, int channels, int robust_type, unsigned blocks, hipStream_t st, const LevelArgs& a);
void launch_iterate_tt3(int mode, int channels, int robust_type, unsigned blocks, hipStream_t st, const LevelArgs& a);
void launch_iterate_tt4(int mode, int channels, int robust_type, unsigned blocks, hipStream_t st, const LevelArgs& a);
void launch_iterate_tt5(int mode, int channels, int robust_type, unsigned blocks, hipStream_t st, const LevelArgs& a);

inline void launch_iterate(int tt, int mode, int channels, int robust_type, unsigned blocks, hipStream_t st,
                           const LevelArgs& a) {
  switch (tt) {
    case ICA_TRANSLATION: launch_iterate_tt1(mode, channels, robust_type, blocks, st, a); break;
    case ICA_EUCLIDEAN: launch_iterate_tt2(mode, channels, robust_type, blocks, st, a); break;
    case ICA_SIMILARITY: launch_iterate_tt3(mode, channels, robust_type, blocks, st, a); break;
    case ICA_AFFINITY: launch_iterate_tt4(mode, channels, robust_type, blocks, st, a); break;
    default: launch_iterate_tt5(mode, channels, robust_type, blocks, st, a); break;
  }
}

}  // namespace ica
