"""Host-side mirror of the reference's ``transformation`` module (reference: src/transformation.py).

Only the 8-number algebra lives here (NumPy, float64); image warping runs on the GPU.
"""
from __future__ import annotations

import math
from enum import Enum

import numpy as np


class TransformType(Enum):
    """reference: transformation.py:8-13"""
    TRANSLATION = 1
    EUCLIDEAN = 2
    SIMILARITY = 3
    AFFINITY = 4
    HOMOGRAPHY = 5

    def nparams(self) -> int:
        """reference: transformation.py:16-32"""
        return {1: 2, 2: 3, 3: 4, 4: 6, 5: 8}[self.value]


def _tt(transform_type) -> int:
    return transform_type.value if isinstance(transform_type, Enum) else int(transform_type)


def params2matrix(p, transform_type) -> np.ndarray:
    """reference: transformation.py:188-236"""
    t = _tt(transform_type)
    m = np.identity(3)
    if t == 1:
        m[0, 2], m[1, 2] = p[0], p[1]
    elif t == 2:
        c, s = math.cos(p[2]), math.sin(p[2])
        m[0, :], m[1, :] = (c, -s, p[0]), (s, c, p[1])
    elif t == 3:
        m[0, :], m[1, :] = (1 + p[2], -p[3], p[0]), (p[3], 1 + p[2], p[1])
    elif t == 4:
        m[0, :], m[1, :] = (1 + p[2], p[3], p[0]), (p[4], 1 + p[5], p[1])
    elif t == 5:
        m[0, :], m[1, :], m[2, :2] = (1 + p[0], p[1], p[2]), (p[3], 1 + p[4], p[5]), (p[6], p[7])
    return m


_TYPE_OF_NPARAMS = {2: 1, 3: 2, 4: 3, 6: 4, 8: 5}


def project(x, y, p, nparams):
    """reference: transformation.py:144-185 -- the point (x, y) under the transform whose type is given by its NUMBER of
    parameters (2, 3, 4, 6 or 8); anything else raises the reference's ValueError.  Scalars or arrays."""
    if nparams not in _TYPE_OF_NPARAMS:
        raise ValueError("Invalid transformation type")
    m = params2matrix(p, _TYPE_OF_NPARAMS[nparams])
    xp = m[0, 0] * x + m[0, 1] * y + m[0, 2]
    yp = m[1, 0] * x + m[1, 1] * y + m[1, 2]
    if nparams == 8:
        d = m[2, 0] * x + m[2, 1] * y + 1
        return xp / d, yp / d
    return xp, yp


def matrix2params(matrix, transform_type):
    """reference: transformation.py:238-263 (returns a list, as the reference does)."""
    t = _tt(transform_type)
    m = matrix
    if t == 1:
        return [m[0, 2], m[1, 2]]
    if t == 2:
        return [m[0, 2], m[1, 2], np.arctan2(m[1, 0], m[0, 0])]
    if t == 3:
        return [m[0, 2], m[1, 2], m[0, 0] - 1, m[1, 0]]
    if t == 4:
        return [m[0, 2], m[1, 2], m[0, 0] - 1, m[0, 1], m[1, 0], m[1, 1] - 1]
    if t == 5:
        return [m[0, 0] - 1, m[0, 1], m[0, 2], m[1, 0], m[1, 1] - 1, m[1, 2], m[2, 0], m[2, 1]]
    return np.zeros(0)


def update_transform(p: np.ndarray, dp: np.ndarray, transform_type, exact: bool = False) -> np.ndarray:
    """p <- params(M(p) . M(dp)^-1), in place and returned (reference: transformation.py:35-141).

    Host twin of the device routine in csrc/ica_math.hpp (same construction: 3x3 composition through
    the adjugate plus the reference's two scalar-formula deviations unless ``exact``)."""
    t = _tt(transform_type)
    if t == 1:
        p[:2] -= dp[:2]
        return p
    P, A = params2matrix(p, t), params2matrix(dp, t)
    adj = np.linalg.inv(A) * np.linalg.det(A)
    m = P @ adj
    if t == 2:
        p[0], p[1], p[2] = m[0, 2], m[1, 2], math.atan2(m[1, 0], m[0, 0])
    elif t == 3:
        det = A[0, 0] ** 2 + A[1, 0] ** 2
        if det * det > 1e-10:
            p[:] = [m[0, 2] / det, m[1, 2] / det, m[0, 0] / det - 1, m[1, 0] / det]
    elif t == 4:
        det = A[0, 0] * A[1, 1] - A[0, 1] * A[1, 0]
        if det * det > 1e-10:
            ep, c, d = p[5], dp[0], dp[4]
            p[:] = [m[0, 2] / det, m[1, 2] / det, m[0, 0] / det - 1, m[0, 1] / det, m[1, 0] / det, m[1, 1] / det - 1]
            if not exact:
                p[1] += (d * d * ep - c * d * ep) / det
    elif t == 5:
        det = -m[2, 2]
        if det * det > 1e-10:
            a, c, g = dp[0], dp[2], dp[6]
            m = m / m[2, 2]
            p[:] = [m[0, 0] - 1, m[0, 1], m[0, 2], m[1, 0], m[1, 1] - 1, m[1, 2], m[2, 0], m[2, 1]]
            if not exact:
                p[4] -= (-a + c * g) / det
    return p


def transform_image(image, transformation_type, gt):
    """Test-data generator of the reference (transformation.py:266-318): bilinear warp of `image` by the inverse of
    the transform `gt`, zeros outside, float64 (H, W, C) result.  Runs on the GPU (ica_op_transform_image_dev)."""
    import torch

    from .inverse_compositional_algorithm import _use_engine
    image = np.asarray(image)
    if image.dtype != np.float64:
        image = image.astype(np.float64)
        print("Image type: ", image.dtype)                              # the reference prints these two lines
        print("Range of image: ", np.min(image), np.max(image))
    t = _tt(transformation_type)
    gt = np.asarray(gt, dtype=np.float64)
    if np.all(np.abs(gt) < 1e-10):
        M = np.eye(3)
    elif t == 2:   # EuclideanTransform(rotation=-gt[2], translation=gt[:2])  (transformation.py:309)
        M = params2matrix([gt[0], gt[1], -gt[2]], 2)
    elif t == 1:   # EuclideanTransform(translation=gt)                         (transformation.py:305)
        M = params2matrix(gt, 1)
    else:
        M = params2matrix(gt, t)
    x = np.ascontiguousarray(np.moveaxis(image.astype(np.float32), 2, 0))[None]
    with _use_engine() as eng:
        out = eng.transform_image(torch.from_numpy(x).to(eng.device), np.linalg.inv(M)[None])
        return np.moveaxis(out[0].cpu().numpy().astype(np.float64), 0, 2)
