"""Deterministic synthetic warped-pair workloads (BASELINE.json configs).

``make_pair`` is the canonical CPU generator (NumPy/SciPy, float64 internally, uint8-valued
output) used for the parity fixtures: the same seed gives the same pixels on every machine.
``make_batch_device`` (engine.py) builds bulk batches of the same distribution on the GPU.

The texture is band-limited multi-octave noise (roughly 1/f, like natural images) so that the
coarse pyramid levels still see structure; ground-truth transforms are small (a few percent of
the image size), the regime the inverse-compositional algorithm is built for.
"""
from __future__ import annotations

import numpy as np

TRANSLATION, EUCLIDEAN, SIMILARITY, AFFINITY, HOMOGRAPHY = 1, 2, 3, 4, 5
NPARAMS = {1: 2, 2: 3, 3: 4, 4: 6, 5: 8}

OCTAVE_SIGMAS = (1.5, 3.0, 6.0, 12.0, 24.0)


def texture(rng: np.random.Generator, H: int, W: int, C: int) -> np.ndarray:
    """uint8-valued float64 (H, W, C) texture."""
    from scipy.ndimage import gaussian_filter

    acc = np.zeros((H, W, C))
    for k, s in enumerate(OCTAVE_SIGMAS):
        n = rng.standard_normal((H, W, C))
        g = gaussian_filter(n, sigma=(s, s, 0), mode="reflect")
        acc += g / g.std() * (1.5 ** k)
    lo, hi = acc.min(), acc.max()
    return np.round((acc - lo) / (hi - lo) * 255.0)


def ground_truth(rng: np.random.Generator, ttype: int, H: int, W: int) -> np.ndarray:
    """Small random transform parameters (reference parametrisation, transformation.py:188-236)."""
    u = lambda s: rng.uniform(-s, s)
    tx, ty = u(0.02) * W, u(0.02) * H
    if ttype == TRANSLATION:
        return np.array([tx, ty])
    if ttype == EUCLIDEAN:
        return np.array([tx, ty, u(0.05)])
    if ttype == SIMILARITY:
        return np.array([tx, ty, u(0.02), u(0.02)])
    if ttype == AFFINITY:
        return np.array([tx, ty, u(0.02), u(0.02), u(0.02), u(0.02)])
    if ttype == HOMOGRAPHY:
        s = 2e-5 * (1024.0 / W)
        return np.array([u(0.02), u(0.02), tx, u(0.02), u(0.02), ty, u(s), u(s)])
    raise ValueError("unknown transform type")


def matrix_of(p, ttype: int) -> np.ndarray:
    p = np.asarray(p, dtype=np.float64)
    m = np.eye(3)
    if ttype == TRANSLATION:
        m[0, 2], m[1, 2] = p[0], p[1]
    elif ttype == EUCLIDEAN:
        m[:2] = [[np.cos(p[2]), -np.sin(p[2]), p[0]], [np.sin(p[2]), np.cos(p[2]), p[1]]]
    elif ttype == SIMILARITY:
        m[:2] = [[1 + p[2], -p[3], p[0]], [p[3], 1 + p[2], p[1]]]
    elif ttype == AFFINITY:
        m[:2] = [[1 + p[2], p[3], p[0]], [p[4], 1 + p[5], p[1]]]
    else:
        m[:] = [[1 + p[0], p[1], p[2]], [p[3], 1 + p[4], p[5]], [p[6], p[7], 1]]
    return m


def resample(img: np.ndarray, M: np.ndarray) -> np.ndarray:
    """out(x) = Catmull-Rom interpolation of img at M.x, 0 where the 4x4 footprint leaves the image."""
    H, W = img.shape[:2]
    ii, jj = np.mgrid[0:H, 0:W].astype(np.float64)
    d = M[2, 0] * jj + M[2, 1] * ii + M[2, 2]
    x = (M[0, 0] * jj + M[0, 1] * ii + M[0, 2]) / d
    y = (M[1, 0] * jj + M[1, 1] * ii + M[1, 2]) / d
    x0, y0 = np.floor(x), np.floor(y)
    ok = (x0 >= 1) & (x0 <= W - 3) & (y0 >= 1) & (y0 <= H - 3)
    xi = np.where(ok, x0, 1).astype(np.int64)
    yi = np.where(ok, y0, 1).astype(np.int64)
    t, s = (x - x0), (y - y0)

    def w(t):
        return (((-0.5 * t + 1.0) * t - 0.5) * t, (1.5 * t - 2.5) * t * t + 1.0,
                ((-1.5 * t + 2.0) * t + 0.5) * t, (0.5 * t - 0.5) * t * t)
    wx, wy = w(t), w(s)
    out = np.zeros_like(img, dtype=np.float64)
    for r in range(4):
        row = sum(wx[k][..., None] * img[yi + r - 1, xi + k - 1] for k in range(4))
        out += wy[r][..., None] * row
    out[~ok] = 0.0
    return out


def make_pair(seed: int, H: int, W: int, C: int, ttype: int, robust_clutter: bool = False):
    """Returns (I1, I2, p_gt): float32 (H, W, C) arrays with uint8 values; solving I1 <- I2 gives p ~ p_gt."""
    rng = np.random.default_rng(seed)
    I2 = texture(rng, H, W, C)
    p_gt = ground_truth(rng, ttype, H, W)
    I1 = resample(I2, matrix_of(p_gt, ttype))
    if robust_clutter:
        I1 = I1 + rng.normal(0.0, 5.0, I1.shape)
        oh, ow = int(round(H * 0.316)), int(round(W * 0.316))  # ~10 % of the area
        oy, ox = int(rng.integers(0, H - oh)), int(rng.integers(0, W - ow))
        I1[oy:oy + oh, ox:ox + ow] = rng.integers(0, 256, (oh, ow, C))
    I1 = np.clip(np.round(I1), 0, 255)
    return I1.astype(np.float32), I2.astype(np.float32), p_gt
