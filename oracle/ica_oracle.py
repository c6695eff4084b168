"""CPU oracle for the inverse-compositional (IC) hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a float64 NumPy restatement of the reference algorithm
(mfournigault/inverse_compositional_algorithm, ``src/*.py``).  It is the *checker*:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  Nothing under ``inverse_compositional_algorithm_amd/`` imports
it, and the product path never falls back to it.

Pinning status (see DESIGN.md "Oracle"):
  * every arithmetic function below is compared against the reference's own
    function, imported in the build container by ``oracle/ref_harness.py``
    (tests/test_oracle_vs_reference.py, skipped where /root/reference is absent);
  * the two third-party calls the reference makes (``skimage.transform.warp`` and
    ``skimage.transform.rescale``; scikit-image is not installed in this image) are
    restated here (``keys_warp`` / ``rescale_restated``) and pinned by the
    per-iteration traces stored in the reference's notebooks
    (tests/golden/kat_traces.json, tests/test_oracle_kat.py);
  * TRUNCATED_QUADRATIC cannot run in the reference (``rhop`` raises on arrays,
    image_optimisation.py:40) -> "parity unpinned" for that robust function.

Structure mirrors the reference on purpose (materialised steepest-descent tensor +
einsum reductions + explicit inverse) so that timing it is a fair stand-in for
"the repo's NumPy CPU path" (bench.py cpu_baseline.kind == "port").

All file:line citations are relative to /root/reference/.
"""
from __future__ import annotations

import math

import numpy as np

# --- enums (transformation.py:8-32, image_optimisation.py:10-15) -------------------
TRANSLATION, EUCLIDEAN, SIMILARITY, AFFINITY, HOMOGRAPHY = 1, 2, 3, 4, 5
QUADRATIC, TRUNCATED_QUADRATIC, GERMAN_MCCLURE, LORENTZIAN, CHARBONNIER = 0, 1, 2, 3, 4
_NPARAMS = {TRANSLATION: 2, EUCLIDEAN: 3, SIMILARITY: 4, AFFINITY: 6, HOMOGRAPHY: 8}

# constants.py:1-4
MAX_ITER = 30
LAMBDA_0 = 80.0
LAMBDA_N = 5.0
LAMBDA_RATIO = 0.9


def nparams(ttype: int) -> int:
    """transformation.py:16-32."""
    return _NPARAMS[int(ttype)]


# --- transform algebra ----------------------------------------------------------------
def params2matrix(p, ttype: int) -> np.ndarray:
    """p -> 3x3 matrix (transformation.py:188-236)."""
    p = np.asarray(p, dtype=np.float64)
    m = np.eye(3)
    if ttype == TRANSLATION:
        m[0, 2], m[1, 2] = p[0], p[1]
    elif ttype == EUCLIDEAN:
        c, s = math.cos(p[2]), math.sin(p[2])
        m[:2] = [[c, -s, p[0]], [s, c, p[1]]]
    elif ttype == SIMILARITY:
        m[:2] = [[1 + p[2], -p[3], p[0]], [p[3], 1 + p[2], p[1]]]
    elif ttype == AFFINITY:
        m[:2] = [[1 + p[2], p[3], p[0]], [p[4], 1 + p[5], p[1]]]
    elif ttype == HOMOGRAPHY:
        m[:] = [[1 + p[0], p[1], p[2]], [p[3], 1 + p[4], p[5]], [p[6], p[7], 1.0]]
    else:
        raise ValueError("unknown transform type")
    return m


def matrix2params(m, ttype: int) -> np.ndarray:
    """3x3 matrix -> p (transformation.py:238-263)."""
    m = np.asarray(m, dtype=np.float64)
    if ttype == TRANSLATION:
        return np.array([m[0, 2], m[1, 2]])
    if ttype == EUCLIDEAN:
        return np.array([m[0, 2], m[1, 2], math.atan2(m[1, 0], m[0, 0])])
    if ttype == SIMILARITY:
        return np.array([m[0, 2], m[1, 2], m[0, 0] - 1, m[1, 0]])
    if ttype == AFFINITY:
        return np.array([m[0, 2], m[1, 2], m[0, 0] - 1, m[0, 1], m[1, 0], m[1, 1] - 1])
    if ttype == HOMOGRAPHY:
        return np.array([m[0, 0] - 1, m[0, 1], m[0, 2], m[1, 0], m[1, 1] - 1, m[1, 2], m[2, 0], m[2, 1]])
    raise ValueError("unknown transform type")


def _adjugate3(a: np.ndarray) -> np.ndarray:
    c = np.empty((3, 3))
    c[0, 0] = a[1, 1] * a[2, 2] - a[1, 2] * a[2, 1]
    c[0, 1] = a[0, 2] * a[2, 1] - a[0, 1] * a[2, 2]
    c[0, 2] = a[0, 1] * a[1, 2] - a[0, 2] * a[1, 1]
    c[1, 0] = a[1, 2] * a[2, 0] - a[1, 0] * a[2, 2]
    c[1, 1] = a[0, 0] * a[2, 2] - a[0, 2] * a[2, 0]
    c[1, 2] = a[0, 2] * a[1, 0] - a[0, 0] * a[1, 2]
    c[2, 0] = a[1, 0] * a[2, 1] - a[1, 1] * a[2, 0]
    c[2, 1] = a[0, 1] * a[2, 0] - a[0, 0] * a[2, 1]
    c[2, 2] = a[0, 0] * a[1, 1] - a[0, 1] * a[1, 0]
    return c


def update_transform(p: np.ndarray, dp: np.ndarray, ttype: int, exact: bool = False) -> np.ndarray:
    """Compositional update p <- params(M(p) . M(dp)^-1), IN PLACE (transformation.py:35-141).

    The reference writes the composition out as closed-form scalar expressions.  This
    restatement composes 3x3 matrices (M(p) . adj(M(dp)) / normaliser), which is the same
    function to ~1e-15, and then re-applies the two places where the reference's scalar
    expressions differ from the exact composition:
      * AFFINITY  p[1]: last numerator term is d*d*ep where the composition has c*d*ep
        (transformation.py:106);
      * HOMOGRAPHY p[4]: numerator lacks (-a + c*g)                  (transformation.py:136).
    ``exact=True`` drops those two corrections.  The update is skipped when the reference's
    ``det*det <= 1e-10`` guard fires (transformation.py:79,98,131).
    """
    n = nparams(ttype)
    if ttype == TRANSLATION:
        p[:n] -= dp[:n]  # transformation.py:52-54
        return p
    P = params2matrix(p, ttype)
    A = params2matrix(dp, ttype)
    if ttype == EUCLIDEAN:  # transformation.py:56-70 (no guard: det == 1)
        m = P @ _adjugate3(A)
        p[0], p[1] = m[0, 2], m[1, 2]
        p[2] = math.atan2(m[1, 0], m[0, 0])
        return p
    if ttype == SIMILARITY:  # transformation.py:72-87
        det = A[0, 0] * A[0, 0] + A[1, 0] * A[1, 0]
        if det * det > 1e-10:
            m = (P @ _adjugate3(A)) / det
            p[:] = [m[0, 2], m[1, 2], m[0, 0] - 1, m[1, 0]]
        return p
    if ttype == AFFINITY:  # transformation.py:89-110
        det = A[0, 0] * A[1, 1] - A[0, 1] * A[1, 0]
        if det * det > 1e-10:
            m = (P @ _adjugate3(A)) / det
            ep = p[5]
            c, d = dp[0], dp[4]
            p[:] = [m[0, 2], m[1, 2], m[0, 0] - 1, m[0, 1], m[1, 0], m[1, 1] - 1]
            if not exact:
                p[1] += (d * d * ep - c * d * ep) / det
        return p
    if ttype == HOMOGRAPHY:  # transformation.py:112-139
        m = P @ _adjugate3(A)
        det = -m[2, 2]  # == the reference's `det` expression (transformation.py:130)
        if det * det > 1e-10:
            a, c, g = dp[0], dp[2], dp[6]
            m = m / m[2, 2]
            p[:] = [m[0, 0] - 1, m[0, 1], m[0, 2], m[1, 0], m[1, 1] - 1, m[1, 2], m[2, 0], m[2, 1]]
            if not exact:
                p[4] -= (-a + c * g) / det
        return p
    raise ValueError("unknown transform type")


# --- warp: restatement of skimage.transform.warp(order=3, mode='constant', cval=nan) ---
def _cubic(v0, v1, v2, v3, t):
    # Catmull-Rom / Keys a=-1/2 (same polynomial as bicubic_interpolation.py:39-41)
    return v1 + 0.5 * t * (v2 - v0 + t * (2.0 * v0 - 5.0 * v1 + 4.0 * v2 - v3 + t * (3.0 * (v1 - v2) + v3 - v0)))


def keys_warp(img: np.ndarray, M: np.ndarray) -> np.ndarray:
    """Iw[i,j,c] = bicubic I2 at M.(j,i,1); NaN if any of the 16 taps is outside the image;
    clipped to [img.min(), img.max()] (bicubic_interpolation.py:199 -> skimage.transform.warp
    with order=3, mode='constant', cval=nan, preserve_range=True, clip=True)."""
    img = np.asarray(img, dtype=np.float64)
    H, W = img.shape[:2]
    ii, jj = np.mgrid[0:H, 0:W].astype(np.float64)
    xw = M[0, 0] * jj + M[0, 1] * ii + M[0, 2]
    yw = M[1, 0] * jj + M[1, 1] * ii + M[1, 2]
    ww = M[2, 0] * jj + M[2, 1] * ii + M[2, 2]
    x = xw / ww
    y = yw / ww
    x0 = np.floor(x)
    y0 = np.floor(y)
    fx = (x - x0)[..., None]
    fy = (y - y0)[..., None]
    valid = (x0 - 1 >= 0) & (x0 + 2 <= W - 1) & (y0 - 1 >= 0) & (y0 + 2 <= H - 1)
    xi = np.where(valid, x0, 1).astype(np.int64)
    yi = np.where(valid, y0, 1).astype(np.int64)
    rows = []
    for r in (-1, 0, 1, 2):
        taps = [img[yi + r, xi + k] for k in (-1, 0, 1, 2)]
        rows.append(_cubic(taps[0], taps[1], taps[2], taps[3], fx))
    out = _cubic(rows[0], rows[1], rows[2], rows[3], fy)
    out = np.clip(out, img.min(), img.max())
    out[~valid] = np.nan
    return out


def warp_params(img, p, ttype):
    """bicubic_interpolation.py:154-206: identity if all |p_i| < 1e-10, else params2matrix."""
    p = np.asarray(p, dtype=np.float64)
    M = np.eye(3) if np.all(np.abs(p) < 1e-10) else params2matrix(p, ttype)
    return keys_warp(img, M)


# --- gradient / Jacobian / steepest descent -------------------------------------------
def gradient(I1: np.ndarray, nanifoutside: bool, delta: int):
    """Central differences with a zero 1-px border, then a NaN frame of width delta
    (inverse_compositional_algorithm.py:81-93)."""
    Ix = np.zeros_like(I1)
    Iy = np.zeros_like(I1)
    Ix[:, 1:-1] = 0.5 * (I1[:, 2:] - I1[:, :-2])
    Iy[1:-1, :] = 0.5 * (I1[2:, :] - I1[:-2, :])
    if nanifoutside is True and delta > 0:
        for g in (Ix, Iy):
            g[:delta] = np.nan
            g[-delta:] = np.nan
            g[:, :delta] = np.nan
            g[:, -delta:] = np.nan
    return Ix, Iy


def jacobian(ttype: int, nx: int, ny: int) -> np.ndarray:
    """dW/dp at p=0, layout [...,0:n] = x-row, [...,n:2n] = y-row (derivatives.py:7-70)."""
    n = nparams(ttype)
    J = np.zeros((ny, nx, 2 * n))
    y, x = np.mgrid[0:ny, 0:nx].astype(np.float64)
    one = 1.0
    if ttype == TRANSLATION:
        cols = {0: one, 3: one}
    elif ttype == EUCLIDEAN:
        cols = {0: one, 2: -y, 4: one, 5: x}
    elif ttype == SIMILARITY:
        cols = {0: one, 2: x, 3: -y, 5: one, 6: y, 7: x}
    elif ttype == AFFINITY:
        cols = {0: one, 2: x, 3: y, 7: one, 10: x, 11: y}
    elif ttype == HOMOGRAPHY:
        cols = {0: x, 1: y, 2: one, 6: -x * x, 7: -x * y, 11: x, 12: y, 13: one, 14: -x * y, 15: -y * y}
    else:
        raise ValueError("unknown transform type")
    for k, v in cols.items():
        J[:, :, k] = v
    return J


def steepest_descent_images(Ix, Iy, J, n):
    """DIJ[y,x,c,k] = Ix*J[k] + Iy*J[k+n] (image_optimisation.py:158-194)."""
    return Ix[..., None] * J[:, :, None, :n] + Iy[..., None] * J[:, :, None, n:]


def hessian(DIJ, rho=None):
    """H = sum_pix rho * DIJ^T DIJ with element-wise non-finite -> 0
    (derivatives.py:73-88, robust :91-107)."""
    D = np.where(np.isfinite(DIJ), DIJ, 0.0)
    if rho is None:
        return np.einsum("ijck,ijcm->km", D, D)
    return np.einsum("ij,ijck,ijcm->km", rho, D, D)


def independent_vector(DIJ, DI, rho=None):
    """b = sum_pix rho * DIJ^T DI, non-finite -> 0 on both (image_optimisation.py:82-143)."""
    D = np.where(np.isfinite(DIJ), DIJ, 0.0)
    R = np.where(np.isfinite(DI), DI, 0.0)
    prod = np.einsum("ijck,ijc->ijk", D, R)
    if rho is not None:
        prod = rho[..., None] * prod
    return prod.sum(axis=(0, 1))


def rhop(t2, lam: float, rtype: int):
    """rho'(t^2; lambda) (image_optimisation.py:17-53).  TRUNCATED_QUADRATIC raises on arrays in
    the reference (:40); the obvious element-wise reading is used here (parity unpinned)."""
    l2 = lam * lam
    if rtype == QUADRATIC:
        return np.ones_like(t2)
    if rtype == TRUNCATED_QUADRATIC:
        return np.where(t2 < l2, 1.0, 0.0)
    if rtype == GERMAN_MCCLURE:
        return l2 / ((l2 + t2) * (l2 + t2))
    if rtype == LORENTZIAN:
        return 1.0 / (l2 + t2)
    if rtype == CHARBONNIER:
        return 1.0 / np.sqrt(t2 + l2)
    raise ValueError("Unknown type for robust error function")


def robust_error_function(DI, lam, rtype):
    """rho[y,x] = rho'(sum_c DI_c^2) with non-finite DI -> 0 (image_optimisation.py:56-79)."""
    R = np.where(np.isfinite(DI), DI, 0.0)
    t2 = np.einsum("ijc,ijc->ij", R, R)
    return np.where(np.isfinite(t2), rhop(t2, lam, rtype), 0.0)


def inverse_hessian(H):
    """np.linalg.inv, zeros when singular (derivatives.py:110-130)."""
    try:
        return np.linalg.inv(H)
    except np.linalg.LinAlgError:
        return np.zeros_like(H)


def parametric_solve(H_1, b):
    """dp = H^-1 b ; error = ||dp||_2 (image_optimisation.py:146-155)."""
    dp = H_1 @ b
    return math.sqrt(float(np.sum(dp ** 2))), dp


# --- drivers ---------------------------------------------------------------------------
def _check(I1, I2, TOL, need_rgb):
    if need_rgb and (I1.ndim != 3 or I2.ndim != 3 or I1.shape[2] != 3 or I2.shape[2] != 3):
        raise ValueError("I1 and I2 must be RGB images with channels in the last dimension")
    if I1.shape != I2.shape:
        raise ValueError("I1 and I2 must have the same dimensions")
    if TOL >= 0.01:
        raise ValueError("TOL must be positive and very small (less than 0.01)")


def ica_single_scale(I1, I2, p, ttype, TOL, robust_type=QUADRATIC, lambda_=0.0,
                     nanifoutside=True, delta=10, exact_update=False, trace=None,
                     max_iter=MAX_ITER, need_rgb=False):
    """One pyramid level: the L2 loop (inverse_compositional_algorithm.py:17-133) when
    robust_type == QUADRATIC, else the robust loop (:135-261).  Mutates and returns p.

    ``trace`` (a list) receives one dict per iteration: dp_norm, p (copy after update), lambda
    (value *after* the decay, as the reference prints it, :253-257), b, H."""
    _check(I1, I2, TOL, need_rgb)
    I1 = np.asarray(I1, dtype=np.float64)
    I2 = np.asarray(I2, dtype=np.float64)
    ny, nx = I1.shape[:2]
    n = nparams(ttype)
    Ix, Iy = gradient(I1, nanifoutside, delta)
    J = jacobian(ttype, nx, ny)
    DIJ = steepest_descent_images(Ix, Iy, J, n)
    robust = robust_type != QUADRATIC
    if not robust:
        H = hessian(DIJ)
        H_1 = inverse_hessian(H)
    error, niter = 1e10, 0
    lambda_it = lambda_ if lambda_ > 0 else LAMBDA_0
    DI = np.zeros_like(I1)
    Iw = np.zeros_like(I1)
    while error > TOL and niter < max_iter:
        Iw = warp_params(I2, p, ttype)
        DI = Iw - I1
        if robust:
            rho = robust_error_function(DI, lambda_it, robust_type)
            if lambda_ <= 0 and lambda_it > LAMBDA_N:  # decay AFTER use (:235-238)
                lambda_it *= LAMBDA_RATIO
                if lambda_it < LAMBDA_N:
                    lambda_it = LAMBDA_N
            b = independent_vector(DIJ, DI, rho)
            H = hessian(DIJ, rho)
            H_1 = inverse_hessian(H)
        else:
            b = independent_vector(DIJ, DI)
        error, dp = parametric_solve(H_1, b)
        update_transform(p, dp, ttype, exact=exact_update)
        if trace is not None:
            trace.append(dict(dp_norm=error, p=p.copy(), lam=lambda_it, b=b.copy(), H=H.copy()))
        niter += 1
    return p, error, DI, Iw, niter


def zoom_size(nx, ny, factor):
    """zoom.py:8-22 (np.round = half-to-even)."""
    return int(np.round(nx * factor)), int(np.round(ny * factor))


def zoom_in_parameters(p, ttype, nx, ny, nxx, nyy):
    """zoom.py:62-125: translations x nu', projective terms / nu', nu' = max(nxx/nx, nyy/ny)."""
    nu = max(nxx / nx, nyy / ny)
    q = np.array(p, dtype=np.float64)
    if ttype in (TRANSLATION, EUCLIDEAN, SIMILARITY, AFFINITY):
        q[0] *= nu
        q[1] *= nu
    elif ttype == HOMOGRAPHY:
        q[2] *= nu
        q[5] *= nu
        q[6] /= nu
        q[7] /= nu
    else:
        raise ValueError("Unsupported transformation type")
    return q


def rescale_restated(img: np.ndarray, nu: float) -> np.ndarray:
    """skimage.transform.rescale(img, nu, mode='constant', cval=0, order=3, anti_aliasing=True,
    channel_axis=2, preserve_range=True) as called at inverse_compositional_algorithm.py:333-336
    (scikit-image >= 0.19): Gaussian sigma=(f-1)/2 per spatial axis with zero padding, cubic
    B-spline zoom with grid_mode, clip to the input range."""
    from scipy import ndimage as ndi

    img = np.asarray(img, dtype=np.float64)
    out_hw = [max(int(np.round(s * nu)), 1) for s in img.shape[:2]]
    f = np.array([img.shape[0] / out_hw[0], img.shape[1] / out_hw[1], 1.0])
    sigma = np.maximum(0.0, (f - 1.0) / 2.0)
    g = ndi.gaussian_filter(img, sigma, cval=0.0, mode="grid-constant")
    z = ndi.zoom(g, [1.0 / f[0], 1.0 / f[1], 1.0], order=3, mode="grid-constant", cval=0.0, grid_mode=True)
    return np.clip(z, img.min(), img.max())


def ica_pyramidal(I1, I2, p, ttype, nscales, nu, TOL, robust_type=QUADRATIC, lambda_=0.0,
                  nanifoutside=True, delta=10, exact_update=False, traces=None, need_rgb=False):
    """Coarse-to-fine driver (inverse_compositional_algorithm.py:264-374).  ``traces`` (a list)
    receives (scale, [iteration dicts]) per level, coarse to fine.  Returns the level-0 p and the
    level-0 last-iteration error/DI/Iw, plus niter per level (index = scale)."""
    _check(I1, I2, TOL, need_rgb)
    I1s = [np.asarray(I1, dtype=np.float64)]
    I2s = [np.asarray(I2, dtype=np.float64)]
    n = nparams(ttype)
    ps = np.zeros((nscales, n))
    ps[0] = np.asarray(p, dtype=np.float64)[:n]
    nx = [I1s[0].shape[1]]
    ny = [I1s[0].shape[0]]
    for s in range(1, nscales):
        w, h = zoom_size(nx[s - 1], ny[s - 1], nu)
        nx.append(w)
        ny.append(h)
        I1s.append(rescale_restated(I1s[s - 1], nu))
        I2s.append(rescale_restated(I2s[s - 1], nu))
    niters = [0] * nscales
    error, DI, Iw = 1e10, None, None
    for s in range(nscales - 1, -1, -1):
        tr = [] if traces is not None else None
        _, error, DI, Iw, niters[s] = ica_single_scale(
            I1s[s], I2s[s], ps[s], ttype, TOL, robust_type, lambda_, nanifoutside, delta,
            exact_update=exact_update, trace=tr)
        if traces is not None:
            traces.append((s, tr))
        if s > 0:
            ps[s - 1] = zoom_in_parameters(ps[s], ttype, nx[s], ny[s], nx[s - 1], ny[s - 1])
    return ps[0].copy(), error, DI, Iw, niters
