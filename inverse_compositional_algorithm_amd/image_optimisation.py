"""Drop-in for the reference's ``image_optimisation`` module (reference: src/image_optimisation.py): the enum callers
import plus the per-operator array functions, same names, arguments and return shapes.  Inside a solve these operators
run fused in the iteration kernel (csrc/ica_iterate.hpp) and nothing is materialised; the functions here are the
reference's materialised formulation on the GPU (``ica_op_ref_*_dev``, float64), for code written against them."""
from __future__ import annotations

from enum import Enum

import numpy as np


class RobustErrorFunctionType(Enum):
    QUADRATIC = 0
    TRUNCATED_QUADRATIC = 1
    GERMAN_MCCLURE = 2
    LORENTZIAN = 3
    CHARBONNIER = 4


def _eng():
    from .inverse_compositional_algorithm import _use_engine
    return _use_engine()


def _rt(type_) -> int:
    if isinstance(type_, Enum):
        v = type_.value
    else:
        try:
            v = int(type_)
        except (TypeError, ValueError):
            raise ValueError("Unknown type for robust error function")
    if v not in (0, 1, 2, 3, 4):
        raise ValueError("Unknown type for robust error function")
    return v


def rhop(t2, lambda_: float, type_) -> np.ndarray:
    """reference: image_optimisation.py:16-53.  TRUNCATED_QUADRATIC is evaluated element-wise (the reference's
    ``if t2 < lambda2`` raises on arrays, :40)."""
    with _eng() as eng:
        return eng.ref_rhop(np.asarray(t2, dtype=np.float64), lambda_, _rt(type_))


def robust_error_function(DI: np.ndarray, lambda_: float, type_) -> np.ndarray:
    """reference: image_optimisation.py:56-78 -- rho'(sum over channels of finite(DI)^2), (ny, nx)."""
    with _eng() as eng:
        return eng.ref_robust_error(np.asarray(DI, dtype=np.float64), lambda_, _rt(type_))


def independent_vector(DIJ: np.ndarray, DI: np.ndarray, nparams: int) -> np.ndarray:
    """reference: image_optimisation.py:82-110 -- b = sum DIJ^T DI; non-finite entries count as 0."""
    with _eng() as eng:
        return eng.ref_normal(np.asarray(DIJ, dtype=np.float64), DI=np.asarray(DI, dtype=np.float64), want_h=False, want_b=True)[1]


def independent_vector_robust(DIJ: np.ndarray, DI: np.ndarray, rho: np.ndarray, nparams: int) -> np.ndarray:
    """reference: image_optimisation.py:113-143 -- b = sum rho * DIJ^T DI."""
    with _eng() as eng:
        return eng.ref_normal(np.asarray(DIJ, dtype=np.float64), DI=np.asarray(DI, dtype=np.float64),
                              rho=np.asarray(rho, dtype=np.float64), want_h=False, want_b=True)[1]


def parametric_solve(H_1: np.ndarray, b: np.ndarray, nparams: int):
    """reference: image_optimisation.py:146-155 -- (||dp||, dp) with dp = H^-1 b (8 numbers: host algebra)."""
    dp = np.asarray(H_1) @ np.asarray(b)
    return np.sqrt(np.sum(dp ** 2)), dp


def steepest_descent_images(Ix: np.ndarray, Iy: np.ndarray, J: np.ndarray, nparams: int) -> np.ndarray:
    """reference: image_optimisation.py:159-200 -- DIJ[:, :, c, n] = Ix[:, :, c] J[:, :, n] + Iy[:, :, c] J[:, :, n + nparams]."""
    Ix, Iy = np.asarray(Ix, dtype=np.float64), np.asarray(Iy, dtype=np.float64)
    if Ix.shape != Iy.shape:
        raise ValueError("Ix and Iy must have the same dimensions")
    with _eng() as eng:
        return eng.ref_steepest_descent(Ix, Iy, np.asarray(J, dtype=np.float64), int(nparams))
