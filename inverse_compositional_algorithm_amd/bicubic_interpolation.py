"""Drop-in for the one function of the reference's `bicubic_interpolation` module that its drivers use
(`src/bicubic_interpolation.py:154-206`): the bicubic warp of an image under a parametric transform, on the GPU
(`ica_op_warp_dev`, the same device code as the fused iteration kernel's sampler).

Semantics kept from `skimage.transform.warp(order=3, mode="constant", cval=nan, preserve_range=True)`: Keys / Catmull-Rom
weights, NaN wherever any of the 16 taps falls outside the image, result clipped to `[image.min(), image.max()]`;
`nanifoutside` and `delta` are accepted and ignored, as in the reference (its NaN frame is commented out, `:200-204`).
The reference's unused pure-Python / Numba interpolators (`:8-152`) are out of scope."""
import numpy as np

from .inverse_compositional_algorithm import _to_device, _to_host, _use_engine
from .transformation import _tt


def bicubic_interpolation_skimage(image: np.ndarray, params: np.ndarray, transformation_type, nanifoutside: bool,
                                  delta: int) -> np.ndarray:
    image = np.asarray(image)
    if image.ndim != 3:
        raise ValueError("image must have its channels in the last dimension")
    p = np.zeros((1, 8))
    p[0, :len(params)] = np.asarray(params, dtype=np.float64)
    with _use_engine() as eng:
        return _to_host(eng.warp(_to_device(image, eng), p, _tt(transformation_type)))
