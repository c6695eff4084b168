// Compile-only probe: instantiates a few iteration kernels so that tools/isa_cost.py can look at their ISA in seconds
// (the full library takes ~45 s).   hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only iter_probe.hip
#include "../../inverse_compositional_algorithm_amd/csrc/ica_iterate.hpp"
using namespace ica;
template __global__ void ica::ica_iterate_kernel<ICA_HOMOGRAPHY, kModeRobust, 1, ICA_LORENTZIAN, false>(LevelArgs);
template __global__ void ica::ica_iterate_kernel<ICA_HOMOGRAPHY, kModeL2, 1, ICA_QUADRATIC, false>(LevelArgs);
template __global__ void ica::ica_iterate_kernel<ICA_AFFINITY, kModeL2, 1, ICA_QUADRATIC, false>(LevelArgs);
template __global__ void ica::ica_iterate_kernel<ICA_HOMOGRAPHY, kModeRobust, 3, ICA_LORENTZIAN, false>(LevelArgs);
template __global__ void ica::ica_iterate_kernel<ICA_HOMOGRAPHY, kModeRobust, 1, ICA_LORENTZIAN, true>(LevelArgs);
template __global__ void ica::ica_iterate_kernel<ICA_HOMOGRAPHY, kModeRobust, 3, ICA_LORENTZIAN, true>(LevelArgs);
