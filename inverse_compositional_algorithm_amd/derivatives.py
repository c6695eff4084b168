"""Drop-in for the reference's ``derivatives`` module (reference: src/derivatives.py): same names, arguments and return
shapes.  The image-sized work runs on the GPU (HIP kernels ``ica_ref_*`` behind ``ica_op_ref_*_dev``, float64 like the
reference); the 8 x 8 inverse is host algebra.  These are the reference's MATERIALISED formulation -- the fused iteration
kernel never builds J or DIJ -- provided so that code written against the operator functions keeps running."""
from __future__ import annotations

import numpy as np

from .transformation import TransformType, _tt


def _eng():
    from .inverse_compositional_algorithm import _use_engine
    return _use_engine()


def jacobian(transform_type: TransformType, nx: int, ny: int) -> np.ndarray:
    """reference: derivatives.py:7-70 -- (ny, nx, 2 * nparams): x row of the Jacobian first, then the y row."""
    with _eng() as eng:
        return eng.ref_jacobian(_tt(transform_type), nx, ny)


def hessian(DIJ: np.ndarray) -> np.ndarray:
    """reference: derivatives.py:73-89 -- sum of DIJ^T DIJ over pixels and channels; non-finite entries count as 0."""
    DIJ = np.asarray(DIJ, dtype=np.float64)
    with _eng() as eng:
        return eng.ref_normal(DIJ)[0]


def hessian_robust(DIJ: np.ndarray, rho: np.ndarray, nparams: int) -> np.ndarray:
    """reference: derivatives.py:91-107 -- the same sum weighted by rho (ny, nx)."""
    DIJ = np.asarray(DIJ, dtype=np.float64)
    with _eng() as eng:
        return eng.ref_normal(DIJ, rho=np.asarray(rho, dtype=np.float64))[0]


def inverse_hessian(H: np.ndarray, nparams: int) -> np.ndarray:
    """reference: derivatives.py:110-130 -- np.linalg.inv, zeros when H is singular."""
    try:
        return np.linalg.inv(H)
    except np.linalg.LinAlgError:
        return np.zeros((nparams, nparams), dtype=np.float64)
