// Launch interface of the fused iteration kernel (ica_iterate.hpp).  The kernel is instantiated per transform type in
// separate translation units (ica_iterate_inst.hip, compiled once per type, in parallel) so that the library builds in
// seconds; ica_api.hip only sees these five functions.
#pragma once
#include <hip/hip_runtime.h>

#include "ica_common.hpp"

namespace ica {

// mode: kModeL2 / kModeRobust / kModeHessian / kModeL2First; channels 1 or 3; robust_type: the ICA_* error function
// (used by kModeRobust only: a template argument of the kernel).  `blocks` = 1-D grid (see launch_iterate).
bool launch_iterate_tt1(int mode, int channels, int robust_type, unsigned blocks, hipStream_t st, const LevelArgs& a);
bool launch_iterate_tt2(int mode, int channels, int robust_type, unsigned blocks, hipStream_t st, const LevelArgs& a);
bool launch_iterate_tt3(int mode, int channels, int robust_type, unsigned blocks, hipStream_t st, const LevelArgs& a);
bool launch_iterate_tt4(int mode, int channels, int robust_type, unsigned blocks, hipStream_t st, const LevelArgs& a);
bool launch_iterate_tt5(int mode, int channels, int robust_type, unsigned blocks, hipStream_t st, const LevelArgs& a);

// false: no kernel is instantiated for this combination and nothing was launched
inline bool launch_iterate(int tt, int mode, int channels, int robust_type, unsigned blocks, hipStream_t st,
                           const LevelArgs& a) {
  switch (tt) {
    case ICA_TRANSLATION: return launch_iterate_tt1(mode, channels, robust_type, blocks, st, a);
    case ICA_EUCLIDEAN: return launch_iterate_tt2(mode, channels, robust_type, blocks, st, a);
    case ICA_SIMILARITY: return launch_iterate_tt3(mode, channels, robust_type, blocks, st, a);
    case ICA_AFFINITY: return launch_iterate_tt4(mode, channels, robust_type, blocks, st, a);
    default: return launch_iterate_tt5(mode, channels, robust_type, blocks, st, a);
  }
}

}  // namespace ica
