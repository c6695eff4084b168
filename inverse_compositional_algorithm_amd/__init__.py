"""MI355X-native inverse-compositional image alignment.

Importing the package never touches the GPU.  The compute entry points (``engine.IcaEngine`` and the
drop-in functions in ``inverse_compositional_algorithm``) load ``libica_hip.so`` on first use and raise
if it has not been built or no HIP device is present -- there is no CPU fallback.
"""
from .constants import LAMBDA_0, LAMBDA_N, LAMBDA_RATIO, MAX_ITER  # noqa: F401
from .image_optimisation import RobustErrorFunctionType  # noqa: F401
from .transformation import TransformType  # noqa: F401

__all__ = ["TransformType", "RobustErrorFunctionType", "MAX_ITER", "LAMBDA_0", "LAMBDA_N", "LAMBDA_RATIO"]
