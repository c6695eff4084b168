"""Batched torch-facing layers over the HIP engine.

Mirror the constructor and call arguments of the reference's Keras layers `InverseCompositional`,
`RobustInverseCompositional` and `PyramidalInverseCompositional`
(reference src/keras-tf/tf_inverse_compositional_algorithm.py:61-253, :255-465, :467-583): inputs are
`[I1, I2]` (or `[I1, I2, p]`) batches in channels-last layout `(B, H, W, C)`, the result is `(p (B, 8), error (B,),
DI (B, H, W, C), Iw (B, H, W, C))`, and a call may override `transform_type`.

Two deliberate differences from those layers, both in favour of the NumPy path this repository is a drop-in for:
  * each pair stops on its OWN |dp| < TOL (the Keras layers stop the whole batch on one aggregate error,
    tf_inverse_compositional_algorithm.py:434-446), so a pair's result does not depend on its batch mates;
  * the arithmetic is the NumPy path's (Keys bicubic warp with range clip, rescale() pyramid, float64 solve), not the
    TF re-implementation's (tf.image.resize pyramid, float32 solve).
The layers hold no weights; they are `torch.nn.Module`s so they compose with the rest of a torch program.
Tensors already on the engine's GPU are used in place (`channels_first=True` skips the layout change as well).
"""
from typing import Optional, Sequence

import threading

import torch

from . import constants as cts
from .engine import IcaEngine
from .image_optimisation import RobustErrorFunctionType
from .transformation import TransformType

_engines = {}
_engines_lock = threading.Lock()


def _engine(device: int):
    """One engine (= one library context, single-threaded by contract) per GPU, with the lock that serialises its users."""
    with _engines_lock:
        if device not in _engines:
            _engines[device] = (IcaEngine(device), threading.Lock())
        return _engines[device]


class _IcaLayer(torch.nn.Module):
    def __init__(self, transform_type, nscales, nu, TOL, robust_type, lambda_, nanifoutside, delta, verbose, max_iter,
                 device, channels_first):
        super().__init__()
        if not TOL < 0.01:
            raise ValueError("TOL must be positive and very small (less than 0.01)")
        self.transform_type = transform_type
        self.nscales, self.nu, self.TOL = int(nscales), float(nu), float(TOL)
        self.robust_type, self.lambda_ = robust_type, float(lambda_)
        self.nanifoutside, self.delta, self.verbose = bool(nanifoutside), int(delta), bool(verbose)
        self.max_iter = int(max_iter)
        self.device_index = int(device)
        self.channels_first = bool(channels_first)
        self.last_niter: Optional[torch.Tensor] = None

    def _planar(self, x: torch.Tensor, dev) -> torch.Tensor:
        x = torch.as_tensor(x)
        if x.dim() != 4:
            raise ValueError("I1 and I2 must be batches of images, (B, H, W, C)")
        channels = x.shape[1] if self.channels_first else x.shape[3]
        # 8-bit gray / RGB stays 8-bit in device memory (uint8 storage); the 2- and 4-channel kernel reads float32 only
        keep_u8 = x.dtype == torch.uint8 and channels in (1, 3)
        x = x.to(device=dev) if keep_u8 else x.to(device=dev, dtype=torch.float32)
        return x.contiguous() if self.channels_first else x.permute(0, 3, 1, 2).contiguous()

    def forward(self, inputs: Sequence[torch.Tensor], transform_type: Optional[TransformType] = None):
        if transform_type is not None:  # same override as the Keras layers' call(inputs, transform_type=...)
            self.transform_type = transform_type
        if self.transform_type is None:
            raise ValueError("transform_type must be given to the constructor or to the call")
        eng, lock = _engine(self.device_index)
        I1, I2 = self._planar(inputs[0], eng.device), self._planar(inputs[1], eng.device)
        if I1.shape != I2.shape:
            raise ValueError("I1 and I2 must have the same dimensions")
        p0 = inputs[2] if len(inputs) > 2 and self.nscales == 1 else None  # the pyramid starts from zero (ref :345)
        B, C, H, W = I1.shape
        prob = eng.problem(B, H, W, C, int(self.transform_type.value), int(self.robust_type.value),
                           nscales=self.nscales, nu=self.nu, tol=self.TOL, lambda_=self.lambda_, delta=self.delta,
                           nanifoutside=self.nanifoutside, max_iter=self.max_iter)
        with lock:
            res = eng.solve(I1, I2, prob, p0=p0, want_images=True)
        self.last_niter = res.niter
        if self.verbose:
            nit = res.niter.sum(dim=1)
            print(f"pairs: {B}, iterations per pair: min {int(nit.min())} max {int(nit.max())}, "
                  f"converged: {int((res.error < self.TOL).sum())}")
        DI, Iw = res.DI, res.Iw
        if not self.channels_first:
            DI, Iw = DI.permute(0, 2, 3, 1), Iw.permute(0, 2, 3, 1)
        return res.p, res.error, DI, Iw


class InverseCompositional(_IcaLayer):
    """L2 inverse compositional algorithm on one scale (Keras: InverseCompositional, :61-253)."""

    def __init__(self, TOL: float = 1e-3, nanifoutside: bool = True, delta: int = 10, verbose: bool = True,
                 max_iter: int = cts.MAX_ITER, transform_type: Optional[TransformType] = None, device: int = 0,
                 channels_first: bool = False):
        super().__init__(transform_type, 1, 0.5, TOL, RobustErrorFunctionType.QUADRATIC, 0.0, nanifoutside, delta,
                         verbose, max_iter, device, channels_first)


class RobustInverseCompositional(_IcaLayer):
    """Robust inverse compositional algorithm on one scale (Keras: RobustInverseCompositional, :255-465)."""

    def __init__(self, TOL: float = 1e-3, robust_type: RobustErrorFunctionType = RobustErrorFunctionType.CHARBONNIER,
                 lambda_: float = 0.0, nanifoutside: bool = True, delta: int = 10, verbose: bool = True,
                 max_iter: int = cts.MAX_ITER, transform_type: Optional[TransformType] = None, device: int = 0,
                 channels_first: bool = False):
        super().__init__(transform_type, 1, 0.5, TOL, robust_type, lambda_, nanifoutside, delta, verbose, max_iter,
                         device, channels_first)


class PyramidalInverseCompositional(_IcaLayer):
    """Coarse-to-fine driver (Keras: PyramidalInverseCompositional, :467-583)."""

    def __init__(self, transform_type: TransformType, nscales: int = 3, nu: float = 0.5, TOL: float = 1e-3,
                 robust_type: RobustErrorFunctionType = RobustErrorFunctionType.CHARBONNIER, lambda_: float = 0.0,
                 nanifoutside: bool = True, delta: int = 10, verbose: bool = True, device: int = 0,
                 channels_first: bool = False):
        super().__init__(transform_type, nscales, nu, TOL, robust_type, lambda_, nanifoutside, delta, verbose,
                         cts.MAX_ITER, device, channels_first)
