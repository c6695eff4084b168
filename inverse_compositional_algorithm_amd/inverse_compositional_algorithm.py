"""Drop-in for the reference's ``src/inverse_compositional_algorithm.py``: same three functions, same
keyword names, same return tuple ``(p, error, DI, Iw)``, same ValueErrors -- computed on an MI355X by
libica_hip.so.  There is no CPU path here: without the HIP library / a GPU these functions raise.

reference: inverse_compositional_algorithm.py:17-133, :135-261, :264-374
"""
from __future__ import annotations

import contextlib
import threading

import numpy as np
import torch

from . import constants as cts
from .engine import IcaEngine
from .image_optimisation import RobustErrorFunctionType
from .transformation import TransformType

_ENGINES: dict = {}
_ENGINES_LOCK = threading.Lock()
_DEVICE = 0
_FINITE_MSG = "I1 and I2 must contain finite values only"
_RGB_MSG = "I1 and I2 must be RGB images with channels in the last dimension"
_DIM_MSG = "I1 and I2 must have the same dimensions"
_TOL_MSG = "TOL must be positive and very small (less than 0.01)"


def set_device(device: int) -> None:
    """GPU the three drop-in functions run on (default 0).  The reference has no such notion; a process that drives
    several GPUs calls this once per worker thread's turn, or uses engine.IcaEngine / sharding.py directly."""
    global _DEVICE
    _DEVICE = int(device)


def _engine_entry(device: int = None):
    device = _DEVICE if device is None else int(device)
    with _ENGINES_LOCK:
        if device not in _ENGINES:
            _ENGINES[device] = (IcaEngine(device), threading.RLock())
        return _ENGINES[device]


def _engine(device: int = None) -> IcaEngine:
    """The engine of one GPU (not locked: for single-threaded callers such as the tests)."""
    return _engine_entry(device)[0]


@contextlib.contextmanager
def _use_engine(device: int = None):
    """The engine of one GPU, held for the duration of the block.  A library context is single-threaded
    (include/ica_hip.h), so concurrent callers of the drop-in functions are serialised per device."""
    eng, lock = _engine_entry(device)
    with lock:
        yield eng


def _to_device(img: np.ndarray, eng: IcaEngine, u8: bool = False) -> torch.Tensor:
    """(H, W, C) -> [1, C, H, W] on the GPU.  8-bit images (the reference's whole data set; it casts them to float64 on
    entry, ref :63-65, :314-316) travel and stay as uint8 -- same values, a quarter of the bytes."""
    a = np.ascontiguousarray(np.moveaxis(np.asarray(img, dtype=np.uint8 if u8 else np.float32), 2, 0))[None]
    return torch.from_numpy(a).to(eng.device)


def _to_host(t: torch.Tensor) -> np.ndarray:
    return np.moveaxis(t[0].cpu().numpy().astype(np.float64), 0, 2)


def _print_trace(trace, niter, nparams, nscales, robust, pyramidal, robust_type):
    # reference: :125-129 (L2), :253-257 (robust), :341-345,:357-358 (pyramid headers)
    for s in range(nscales - 1, -1, -1):
        if pyramidal:
            print(f"Scale: {s}")
            print("(L2 norm)" if not robust else f"(Robust error function {robust_type})")
        for it in range(int(niter[s])):
            row = trace[s, it]
            ps = " ".join(repr(float(v)) for v in row[1:1 + nparams])
            if robust:
                print(f"|Dp|={float(row[0])}: p=({ps}), lambda_={float(row[9])}")
            else:
                print(f"Iteration {it}: |Dp|={float(row[0])}: p=({ps})")


def _run(I1, I2, p, transform_type, TOL, robust_type, lambda_, nanifoutside, delta, verbose, nscales, nu,
         need_rgb, pyramidal):
    I1 = np.asarray(I1)
    I2 = np.asarray(I2)
    if need_rgb and (I1.ndim != 3 or I2.ndim != 3 or I1.shape[2] != 3 or I2.shape[2] != 3):
        raise ValueError(_RGB_MSG)                                     # ref :48-49, :300-301
    if I1.ndim != 3:
        raise ValueError("not enough values to unpack (expected 3, got %d)" % I1.ndim)  # ref :169 unpack
    if I1.shape != I2.shape:
        raise ValueError(_DIM_MSG)                                     # ref :55-56
    if TOL >= 0.01:
        raise ValueError(_TOL_MSG)                                     # ref :59-60
    if not 1 <= I1.shape[2] <= 4:
        raise ValueError(_RGB_MSG)  # the library is built for 1..4 channels (differences)
    # Non-finite pixels: the reference zero-fills non-finite residuals and derivatives element by element (image_optimisation.py:99,
    # derivatives.py:84) but clips the warp to [I2.min(), I2.max()], which a single NaN turns into NaN bounds and an
    # all-NaN warp: its result on such inputs is an accident, not a contract.  The kernels' fast path does no per-pixel
    # finiteness tests, so such inputs are refused here, loudly (differences).
    for img in (I1, I2):
        if img.dtype.kind == "f" and not np.isfinite(img).all():
            raise ValueError(_FINITE_MSG)
    tt = TransformType(transform_type) if not isinstance(transform_type, TransformType) else transform_type
    rt = RobustErrorFunctionType(robust_type) if not isinstance(robust_type, RobustErrorFunctionType) else robust_type
    n = tt.nparams()
    eng, lock = _engine_entry()
    H, W, C = I1.shape
    prob = eng.problem(1, H, W, C, tt.value, rt.value, nscales, nu, TOL, float(lambda_), bool(nanifoutside is True),
                       int(delta), cts.MAX_ITER)
    p0 = torch.as_tensor(np.asarray(p, dtype=np.float64)[:n]).reshape(1, n)
    u8 = I1.dtype == np.uint8 and I2.dtype == np.uint8 and C in (1, 3)
    with lock:
        res = eng.solve(_to_device(I1, eng, u8), _to_device(I2, eng, u8), prob, p0=p0, want_images=True,
                        want_trace=bool(verbose))
    p_out = res.p[0, :n].cpu().numpy()
    error = float(res.error[0].cpu())
    if verbose:
        _print_trace(res.trace[0].cpu().numpy(), res.niter[0].cpu().numpy(), n, nscales,
                     rt != RobustErrorFunctionType.QUADRATIC, pyramidal, rt)
    if not pyramidal and isinstance(p, np.ndarray) and p.dtype == np.float64 and p.shape[0] >= n:
        # the single-scale entry points mutate the caller's array and return it (transformation.py:54,68-70,...);
        # the pyramidal driver works on its own copy, ps[0] = np.copy(p) (ref :323), and leaves the caller's p alone
        p[:n] = p_out
        p_out = p
    return p_out, error, _to_host(res.DI), _to_host(res.Iw)


def inverse_compositional_algorithm(I1, I2, p, transform_type, TOL, nanifoutside, delta, verbose):
    """Quadratic (L2) inverse compositional algorithm.  reference: :17-133"""
    return _run(I1, I2, p, transform_type, TOL, RobustErrorFunctionType.QUADRATIC, 0.0, nanifoutside, delta, verbose,
                1, 0.5, need_rgb=True, pyramidal=False)


def robust_inverse_compositional_algorithm(I1, I2, p, transform_type, TOL, robust_type, lambda_, nanifoutside, delta,
                                           verbose):
    """Robust inverse compositional algorithm.  reference: :135-261.  For QUADRATIC the reference's robust loop
    re-accumulates and re-inverts the same Hessian every iteration (rho' = 1); the library computes it once per level
    (the L2 schedule) -- the iterates are the same.  Any channel count from 1 to 4 is accepted here (the reference's
    robust entry point has no channel check, :168-177)."""
    return _run(I1, I2, p, transform_type, TOL, robust_type, lambda_, nanifoutside, delta, verbose, 1, 0.5,
                need_rgb=False, pyramidal=False)


def pyramidal_inverse_compositional_algorithm(I1, I2, p, transform_type, nscales, nu, TOL, robust_type, lambda_,
                                              nanifoutside, delta, verbose):
    """Coarse-to-fine driver.  reference: :264-374"""
    return _run(I1, I2, p, transform_type, TOL, robust_type, lambda_, nanifoutside, delta, verbose, int(nscales),
                float(nu), need_rgb=True, pyramidal=True)
