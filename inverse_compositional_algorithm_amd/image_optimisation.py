"""Host-side mirror of the reference's ``image_optimisation`` public names that callers import
(reference: src/image_optimisation.py:10-15).  The operators themselves (rho', steepest-descent images,
independent vector, solve) run inside the fused HIP kernels (csrc/ica_kernels.hpp)."""
from enum import Enum


class RobustErrorFunctionType(Enum):
    QUADRATIC = 0
    TRUNCATED_QUADRATIC = 1
    GERMAN_MCCLURE = 2
    LORENTZIAN = 3
    CHARBONNIER = 4
