"""Parity of the HIP path (through the C ABI) with the CPU oracle on the same seeded inputs.

Tolerances.  The reference computes in float64; the kernels compute per-pixel quantities in float32
and accumulate / solve in float64.  north_star's bar: recovered parameters within 1e-4 relative.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from oracle import ica_oracle as oc  # noqa: E402
from inverse_compositional_algorithm_amd import synthetic as syn  # noqa: E402

TYPES = [oc.TRANSLATION, oc.EUCLIDEAN, oc.SIMILARITY, oc.AFFINITY, oc.HOMOGRAPHY]
P_REL_TOL = 1e-4        # north_star: recovered parameters within 1e-4 relative
CORNER_TOL_PX = 1e-3    # max corner displacement between the two solutions


@pytest.fixture(scope="module")
def eng():
    from inverse_compositional_algorithm_amd.engine import IcaEngine
    return IcaEngine(0)


def to_dev(eng, *imgs):
    """(H,W,C) numpy -> [1,C,H,W] float32 device tensors (batched along dim 0)."""
    a = np.stack([np.moveaxis(np.asarray(i, dtype=np.float32), 2, 0) for i in imgs])
    return torch.from_numpy(np.ascontiguousarray(a)).to(eng.device)


def to_np(t):
    return np.moveaxis(t.cpu().numpy().astype(np.float64), 1, 3)  # [B,H,W,C]


def corners_disp(p, q, t, H, W):
    Mp, Mq = oc.params2matrix(p, t), oc.params2matrix(q, t)
    d = 0.0
    for x, y in ((0, 0), (W - 1, 0), (0, H - 1), (W - 1, H - 1)):
        a = Mp @ [x, y, 1.0]
        b = Mq @ [x, y, 1.0]
        d = max(d, float(np.hypot(a[0] / a[2] - b[0] / b[2], a[1] / a[2] - b[1] / b[2])))
    return d


def pairs(t, n, H=96, W=128, C=3, clutter=False, seed0=500):
    out = [syn.make_pair(seed0 + 17 * t + i, H, W, C, t, robust_clutter=clutter) for i in range(n)]
    return [o[0] for o in out], [o[1] for o in out], [o[2] for o in out]


@pytest.mark.parametrize("t", TYPES)
@pytest.mark.parametrize("C", [1, 3])
def test_warp_operator(eng, t, C):
    """ica_op_warp_dev vs oracle.warp_params (bicubic_interpolation.py:154-206)."""
    I1s, I2s, pg = pairs(t, 3, C=C)
    ps = [g * s for g, s in zip(pg, (1.0, -0.7, 2.5))] + [np.zeros(oc.nparams(t))]
    I2s, I1s = I2s + [I2s[0]], I1s + [I1s[0]]
    p = np.zeros((4, 8))
    for i, q in enumerate(ps):
        p[i, :len(q)] = q
    Iw, DI = eng.warp(to_dev(eng, *I2s), torch.from_numpy(p), t, I1=to_dev(eng, *I1s))
    Iw, DI = to_np(Iw), to_np(DI)
    for i in range(4):
        want = oc.warp_params(I2s[i].astype(np.float64), ps[i], t)
        assert np.array_equal(np.isnan(Iw[i]), np.isnan(want)), "NaN (out-of-image) mask differs"
        np.testing.assert_allclose(np.nan_to_num(Iw[i]), np.nan_to_num(want), atol=2e-3)
        np.testing.assert_allclose(np.nan_to_num(DI[i]), np.nan_to_num(want - I1s[i]), atol=2e-3)


def test_minmax(eng):
    x = torch.randn(5, 3, 37, 91, device=eng.device)
    mm = eng.minmax(x).cpu().numpy()
    np.testing.assert_array_equal(mm[:, 0], x.amin(dim=(1, 2, 3)).cpu().numpy())
    np.testing.assert_array_equal(mm[:, 1], x.amax(dim=(1, 2, 3)).cpu().numpy())


@pytest.mark.parametrize("t", TYPES)
@pytest.mark.parametrize("r", [oc.QUADRATIC, oc.LORENTZIAN])
@pytest.mark.parametrize("C", [1, 3])
def test_normal_equations(eng, t, r, C):
    """H and b of one iteration vs the oracle's materialised-tensor einsums."""
    I1s, I2s, pg = pairs(t, 2, C=C, clutter=(r != 0))
    n = oc.nparams(t)
    H_, W_ = I1s[0].shape[:2]
    prob = eng.problem(2, H_, W_, C, t, r, delta=5)
    p = np.zeros((2, 8))
    p[0, :n] = pg[0] * 0.6
    lam = 37.0
    Hm, bv = eng.normal_equations(to_dev(eng, *I1s), to_dev(eng, *I2s), torch.from_numpy(p), prob, lambda_it=lam)
    Hm, bv = Hm.cpu().numpy(), bv.cpu().numpy()
    for i in range(2):
        I1, I2 = I1s[i].astype(np.float64), I2s[i].astype(np.float64)
        Ix, Iy = oc.gradient(I1, True, 5)
        DIJ = oc.steepest_descent_images(Ix, Iy, oc.jacobian(t, W_, H_), n)
        DI = oc.warp_params(I2, p[i, :n], t) - I1
        rho = oc.robust_error_function(DI, lam, r) if r else None
        Hw, bw = oc.hessian(DIJ, rho), oc.independent_vector(DIJ, DI, rho)
        sc = np.sqrt(np.diag(Hw))
        np.testing.assert_allclose(Hm[i, :n, :n] / np.outer(sc, sc), Hw / np.outer(sc, sc), atol=2e-6)
        # b is a cancelling sum: compare against the scale sum|terms| ~ sc * |DI|_2
        bscale = sc * np.sqrt(np.nansum((DI * (np.sqrt(rho)[..., None] if r else 1.0)) ** 2))
        np.testing.assert_allclose(bv[i, :n] / bscale, bw / bscale, atol=5e-6)
        assert np.all(Hm[i, n:, :] == 0) and np.all(bv[i, n:] == 0)


# (generic, non-round parameter values: with "nice" decimals many sampling positions fall exactly on integers,
#  where floor() -- hence which pixels have their footprint inside the image -- is decided by the last bit)
@pytest.mark.parametrize("t,p", [(oc.EUCLIDEAN, [3.1371, -20.2113, 0.5031]), (oc.SIMILARITY, [5.0731, 4.1177, 0.4519, 0.1033]),
                                 (oc.AFFINITY, [-6.0317, 2.0911, 0.0513, 0.3019, -0.2531, 0.0217]),
                                 (oc.HOMOGRAPHY, [0.2011, 0.1013, 4.0331, -0.1017, 0.1507, -3.0119, 6.1e-4, -4.3e-4]),
                                 (oc.TRANSLATION, [300.37, 10.21])])
@pytest.mark.parametrize("r", [oc.QUADRATIC, oc.CHARBONNIER])
def test_normal_equations_large_warp_fallback(eng, t, p, r):
    """Warps whose chunk footprint does not fit the LDS buffer take the direct-gather branch (some or all chunks);
    a footprint entirely outside the image leaves every pixel invalid.  Same arithmetic, same results."""
    C, H_, W_ = 1, 160, 200
    I1s, I2s, _ = pairs(oc.AFFINITY, 2, H=H_, W=W_, C=C, clutter=(r != 0), seed0=777)
    n = oc.nparams(t)
    prob = eng.problem(2, H_, W_, C, t, r, delta=5)
    pp = np.zeros((2, 8))
    pp[0, :n] = p
    pp[1, :n] = np.array(p) * 0.5
    lam = 25.0
    Hm, bv = eng.normal_equations(to_dev(eng, *I1s), to_dev(eng, *I2s), torch.from_numpy(pp), prob, lambda_it=lam)
    Hm, bv = Hm.cpu().numpy(), bv.cpu().numpy()
    for i in range(2):
        I1, I2 = I1s[i].astype(np.float64), I2s[i].astype(np.float64)
        Ix, Iy = oc.gradient(I1, True, 5)
        DIJ = oc.steepest_descent_images(Ix, Iy, oc.jacobian(t, W_, H_), n)
        DI = oc.warp_params(I2, pp[i, :n], t) - I1
        rho = oc.robust_error_function(DI, lam, r) if r else None
        Hw, bw = oc.hessian(DIJ, rho), oc.independent_vector(DIJ, DI, rho)
        sc = np.sqrt(np.diag(Hw))
        np.testing.assert_allclose(Hm[i, :n, :n] / np.outer(sc, sc), Hw / np.outer(sc, sc), atol=2e-6)
        bscale = sc * max(np.sqrt(np.nansum((DI * (np.sqrt(rho)[..., None] if r else 1.0)) ** 2)), 1.0)
        np.testing.assert_allclose(bv[i, :n] / bscale, bw / bscale, atol=5e-6)


@pytest.mark.parametrize("t", TYPES)
def test_solve_update_operator(eng, t):
    rng = np.random.default_rng(t)
    n = oc.nparams(t)
    B = 16
    H = np.zeros((B, 8, 8))
    b = np.zeros((B, 8))
    p = np.zeros((B, 8))
    sc = np.ones(n)
    if t == oc.HOMOGRAPHY:
        sc[[2, 5]], sc[[6, 7]] = 30.0, 1e-3
    for i in range(B):
        A = rng.normal(size=(n + 3, n)) / sc
        H[i, :n, :n] = A.T @ A
        b[i, :n] = rng.normal(size=n) / sc * 0.05
        p[i, :n] = rng.normal(size=n) * sc * 0.05
    H[0] = 0.0  # singular -> H^-1 = 0 -> dp = 0 (derivatives.py:125-130)
    pn, dp, err = eng.solve_update(torch.from_numpy(H), torch.from_numpy(b), torch.from_numpy(p), t)
    pn, dp, err = pn.cpu().numpy(), dp.cpu().numpy(), err.cpu().numpy()
    for i in range(B):
        e_w, dp_w = oc.parametric_solve(oc.inverse_hessian(H[i, :n, :n]), b[i, :n])
        p_w = oc.update_transform(p[i, :n].copy(), dp_w, t)
        np.testing.assert_allclose(dp[i, :n], dp_w, rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(err[i], e_w, rtol=1e-9)
        np.testing.assert_allclose(pn[i, :n], p_w, rtol=1e-9, atol=1e-14)
    assert err[0] == 0.0 and np.allclose(pn[0], p[0], rtol=1e-14, atol=0)


@pytest.mark.parametrize("shape", [(96, 128), (135, 101), (68, 120)])
@pytest.mark.parametrize("C", [1, 3])
def test_pyramid_down(eng, shape, C):
    """ica_op_pyramid_down_dev vs the restated skimage.rescale (ref :333-336)."""
    H_, W_ = shape
    rng = np.random.default_rng(H_ + C)
    imgs = [np.round(rng.uniform(0, 255, (H_, W_, C))) for _ in range(2)]
    got = to_np(eng.pyramid_down(to_dev(eng, *imgs), 0.5))
    for i in range(2):
        want = oc.rescale_restated(imgs[i], 0.5)
        assert got[i].shape == want.shape
        np.testing.assert_allclose(got[i], want, atol=1e-4)


@pytest.mark.parametrize("nu", [0.75, 0.6, 0.4, 0.3, 0.25])
def test_pyramid_down_other_factors(eng, nu):
    """`nu` is a caller's choice (config.ini): non-dyadic and stronger decimations use wider operators (more taps,
    longer input windows per group of outputs) through the same two passes."""
    rng = np.random.default_rng(int(nu * 100))
    imgs = [np.round(rng.uniform(0, 255, (150, 202, 1))) for _ in range(2)]
    got = to_np(eng.pyramid_down(to_dev(eng, *imgs), nu))
    for i in range(2):
        want = oc.rescale_restated(imgs[i], nu)
        assert got[i].shape == want.shape
        np.testing.assert_allclose(got[i], want, atol=2e-4)


@pytest.fixture(scope="module")
def eng_two_pass():
    """A second context whose pyramid steps run as two kernels (vertical, then horizontal + clip)."""
    import os
    os.environ["ICA_PYRAMID_2PASS"] = "1"
    try:
        from inverse_compositional_algorithm_amd.engine import IcaEngine
        e = IcaEngine(0)
    finally:
        del os.environ["ICA_PYRAMID_2PASS"]
    return e


def _saturated_image(rng, H_, W_, C):
    """Smooth texture with flat 0 / 255 regions: the cubic prefilter overshoots next to them, so the clip to the input's
    range (skimage.rescale) is active."""
    img = np.round(rng.uniform(40, 215, (H_, W_, C)))
    img[H_ // 5:H_ // 2, W_ // 6:W_ // 2] = 255.0
    img[H_ // 2 + 3:H_ - 7, W_ // 2 + 5:W_ - 9] = 0.0
    img[3:9, 5:W_ - 5] = 255.0
    return img


@pytest.mark.parametrize("shape,C,nu", [((64, 1024), 1, 0.5), ((200, 600), 3, 0.5), ((96, 512), 1, 0.5), ((150, 400), 1, 0.75),
                                        ((180, 256), 1, 0.3), ((37, 1028), 1, 0.5)])
def test_pyramid_down_single_kernel(eng, eng_two_pass, shape, C, nu):
    """Widths that are a multiple of 4 run both passes in one kernel (several overlapping column blocks; output written
    unclipped, tiles outside the input range revisited).  Same values as the two-kernel step, bit for bit, and both
    agree with the restated skimage.rescale; the images have saturated regions so that the clip matters."""
    H_, W_ = shape
    rng = np.random.default_rng(H_ * 7 + W_ + C)
    imgs = [_saturated_image(rng, H_, W_, C), np.round(rng.uniform(0, 255, (H_, W_, C)))]
    got = to_np(eng.pyramid_down(to_dev(eng, *imgs), nu))
    two = to_np(eng_two_pass.pyramid_down(to_dev(eng_two_pass, *imgs), nu))
    clipped = 0
    for i in range(2):
        want = oc.rescale_restated(imgs[i], nu)
        assert got[i].shape == want.shape
        np.testing.assert_array_equal(got[i], two[i])
        np.testing.assert_allclose(got[i], want, atol=2e-4)
        assert got[i].min() >= imgs[i].min() and got[i].max() <= imgs[i].max()
        clipped += int(np.sum((got[i] == imgs[i].max()) | (got[i] == imgs[i].min())))
    assert clipped > 0  # the saturated image does get clipped


def test_pyramid_down_single_kernel_random_shapes(eng, eng_two_pass):
    """Seeded sweep of shapes for the one-kernel pyramid step against the two-kernel step (bit for bit): heights from 1
    row up, widths from 4 columns (a single, mostly empty column block) to several blocks, batches of 1-3 images with 1-3
    channels, three factors.  The two-kernel step itself is pinned to the restated skimage.rescale by the tests above."""
    rng = np.random.default_rng(20261005)
    for case in range(40):
        H_ = int(rng.integers(1, 220))
        W_ = 4 * int(rng.integers(1, 280))
        C = int(rng.integers(1, 4))
        nu = [0.5, 0.5, 0.75, 0.3][case % 4]
        nimg = int(rng.integers(1, 4))
        if H_ * W_ * C * nimg > 600000:
            H_ = max(1, 600000 // (W_ * C * nimg))
        imgs = [np.round(rng.uniform(0, 255, (H_, W_, C))) for _ in range(nimg)]
        for im in imgs:  # flat saturated patches: the clip is active somewhere
            im[: max(1, H_ // 3), : max(1, W_ // 4)] = 255.0
            im[H_ // 2:, W_ // 2:] = 0.0
        got = to_np(eng.pyramid_down(to_dev(eng, *imgs), nu))
        two = to_np(eng_two_pass.pyramid_down(to_dev(eng_two_pass, *imgs), nu))
        assert len(got) == len(two) == nimg
        for i in range(nimg):
            assert got[i].shape == two[i].shape, (case, H_, W_, C, nu)
            np.testing.assert_array_equal(got[i], two[i], err_msg=str((case, H_, W_, C, nu)))


@pytest.mark.parametrize("nu", [0.1, 0.07])
def test_pyramid_down_very_small_factor(eng, nu):
    """The reference accepts any nu; factors below ~0.12 exceed what the tiled passes are sized for and run the generic
    banded pass (vertical: a 16-output window wider than 160 input rows)."""
    rng = np.random.default_rng(int(nu * 1000))
    imgs = [np.round(rng.uniform(0, 255, (300, 260, 3)))]
    got = to_np(eng.pyramid_down(to_dev(eng, *imgs), nu))
    want = oc.rescale_restated(imgs[0], nu)
    assert got[0].shape == want.shape
    np.testing.assert_allclose(got[0], want, atol=2e-4)


def test_zoom_in_parameters(eng):
    for t in TYPES:
        p = np.zeros((3, 8))
        p[:, :oc.nparams(t)] = np.random.default_rng(t).normal(size=(3, oc.nparams(t)))
        got = eng.zoom_in_parameters(torch.from_numpy(p), t, 68, 120, 135, 240).cpu().numpy()
        for i in range(3):
            np.testing.assert_allclose(got[i, :oc.nparams(t)],
                                       oc.zoom_in_parameters(p[i, :oc.nparams(t)], t, 68, 120, 135, 240), rtol=1e-15)


CASES = [(oc.TRANSLATION, oc.QUADRATIC, 1, 3), (oc.EUCLIDEAN, oc.QUADRATIC, 2, 3), (oc.SIMILARITY, oc.CHARBONNIER, 1, 3),
         (oc.AFFINITY, oc.QUADRATIC, 2, 1), (oc.AFFINITY, oc.GERMAN_MCCLURE, 2, 3), (oc.HOMOGRAPHY, oc.QUADRATIC, 2, 1),
         (oc.HOMOGRAPHY, oc.LORENTZIAN, 2, 3), (oc.HOMOGRAPHY, oc.CHARBONNIER, 1, 1),
         (oc.AFFINITY, oc.TRUNCATED_QUADRATIC, 1, 1)]


@pytest.mark.parametrize("t,r,ns,C", CASES)
def test_full_solve_matches_oracle(eng, t, r, ns, C):
    """Whole pyramidal solve, iteration by iteration, against the oracle on the same inputs."""
    B = 3
    I1s, I2s, _ = pairs(t, B, H=128, W=160, C=C, clutter=(r != 0), seed0=900 + r)
    n = oc.nparams(t)
    prob = eng.problem(B, 128, 160, C, t, r, nscales=ns, nu=0.5, tol=1e-3, lambda_=0.0, delta=5)
    res = eng.solve(to_dev(eng, *I1s), to_dev(eng, *I2s), prob, want_images=True, want_trace=True, want_stats=True)
    p, err, nit, tr = res.p.cpu().numpy(), res.error.cpu().numpy(), res.niter.cpu().numpy(), res.trace.cpu().numpy()
    Iw, DI = to_np(res.Iw), to_np(res.DI)
    total = 0
    for i in range(B):
        traces = []
        p_w, e_w, DI_w, Iw_w, nit_w = oc.ica_pyramidal(I1s[i].astype(np.float64), I2s[i].astype(np.float64), np.zeros(n),
                                                       t, ns, 0.5, 1e-3, r, 0.0, True, 5, traces=traces)
        assert list(nit[i]) == list(nit_w), f"iterations per scale differ: {nit[i]} vs {nit_w}"
        total += sum(nit_w)
        rel = np.linalg.norm(p[i, :n] - p_w) / np.linalg.norm(p_w)
        assert rel <= P_REL_TOL, f"pair {i}: relative parameter error {rel:.2e}"
        assert corners_disp(p[i, :n], p_w, t, 128, 160) <= CORNER_TOL_PX
        np.testing.assert_allclose(err[i], e_w, rtol=2e-2, atol=1e-6)
        for s, trs in traces:  # per-iteration trace parity
            for k, itd in enumerate(trs):
                row = tr[i, s, k]
                assert corners_disp(row[1:1 + n], itd["p"], t, *[(128, 160), (64, 80)][s]) <= 5e-3
                if r:
                    np.testing.assert_allclose(row[9], itd["lam"], rtol=1e-12)
        assert np.array_equal(np.isnan(Iw[i]), np.isnan(Iw_w))
        np.testing.assert_allclose(np.nan_to_num(Iw[i]), np.nan_to_num(Iw_w), atol=2e-2)
        np.testing.assert_allclose(np.nan_to_num(DI[i]), np.nan_to_num(DI_w), atol=2e-2)
    assert res.stats["pair_iterations_total"] == total


def test_profiling_reports_kernel_time_and_the_clock_the_kernel_ran_at(eng):
    """ica_set_profiling: HIP-event time of the iteration kernel's launches and ica_stats.shader_clock_ghz, measured inside the
    kernel (one workgroup's lifetime in shader cycles over the same in 100 MHz ticks); both zero with profiling off."""
    t = oc.HOMOGRAPHY
    I1s, I2s, _ = pairs(t, 8, H=256, W=320, C=1, clutter=True, seed0=4100)
    prob = eng.problem(8, 256, 320, 1, t, oc.LORENTZIAN, nscales=2, delta=5)
    a, b = to_dev(eng, *I1s), to_dev(eng, *I2s)
    try:
        eng.set_profiling(True)
        on = eng.solve(a, b, prob, want_stats=True)
    finally:
        eng.set_profiling(False)
    off = eng.solve(a, b, prob, want_stats=True)
    assert torch.equal(on.p, off.p) and torch.equal(on.niter, off.niter)      # measuring changes nothing
    assert 0.8 < on.stats["shader_clock_ghz"] < 2.6, on.stats["shader_clock_ghz"]
    assert sum(on.stats["iter_kernel_ms"]) > 0.0
    assert off.stats["shader_clock_ghz"] == 0.0 and sum(off.stats["iter_kernel_ms"]) == 0.0


def test_initial_p_used_only_for_single_scale(eng):
    """ref :327,:337,:372 -- with nscales > 1 the caller's p is discarded."""
    t = oc.AFFINITY
    I1s, I2s, pg = pairs(t, 1, H=128, W=160, C=1)
    d1, d2 = to_dev(eng, *I1s), to_dev(eng, *I2s)
    p0 = torch.from_numpy(pg[0][None] * 0.9)
    a = eng.solve(d1, d2, eng.problem(1, 128, 160, 1, t, nscales=1), p0=p0)
    b = eng.solve(d1, d2, eng.problem(1, 128, 160, 1, t, nscales=1))
    assert int(a.niter[0, 0]) < int(b.niter[0, 0])
    c = eng.solve(d1, d2, eng.problem(1, 128, 160, 1, t, nscales=2), p0=p0)
    d = eng.solve(d1, d2, eng.problem(1, 128, 160, 1, t, nscales=2))
    assert torch.equal(c.p, d.p)


def test_deterministic(eng):
    t = oc.HOMOGRAPHY
    I1s, I2s, _ = pairs(t, 4, H=128, W=160, C=3, clutter=True)
    d1, d2 = to_dev(eng, *I1s), to_dev(eng, *I2s)
    prob = eng.problem(4, 128, 160, 3, t, oc.LORENTZIAN, nscales=2)
    a, b = eng.solve(d1, d2, prob), eng.solve(d1, d2, prob)
    assert torch.equal(a.p, b.p) and torch.equal(a.niter, b.niter)


def test_alternative_schedules_give_the_same_iterates(eng):
    """The library's optional code paths are alternative schedules of the same arithmetic: a separate Hessian pass per
    level instead of the folded first iteration (ICA_SEPARATE_HESSIAN), the direct-gather iteration kernel
    (ICA_ACCUM=1), 16-row tiles everywhere (ICA_MAX_TILE_H).  Identical iteration counts, parameters equal to rounding -- on an L2 and on a robust
    problem, with float32 and with uint8 images (the options are read when a context is created, hence one process per setting;
    with ICA_ACCUM=1 a uint8 level still runs the LDS-staged kernel, and its first L2 iteration must launch it)."""
    import os
    import subprocess
    import sys
    code = ("import numpy as np, torch, sys; sys.path.insert(0, %r);"
            "from inverse_compositional_algorithm_amd.engine import IcaEngine;"
            "e = IcaEngine(0); out = [];\n"
            "for t, r, u8 in ((5, 3, 0), (4, 0, 0), (5, 3, 1), (4, 0, 1)):\n"
            "    I1, I2, _ = e.make_batch(5, 16, 128, 160, 1, t, robust_clutter=(r != 0))\n"
            "    if u8: I1, I2 = I1.to(torch.uint8), I2.to(torch.uint8)\n"
            "    q = e.solve(I1, I2, e.problem(16, 128, 160, 1, t, r, nscales=2, delta=5))\n"
            "    out += [q.p.cpu().numpy().ravel(), q.niter.cpu().numpy().ravel().astype(float)]\n"
            "np.save(sys.argv[1], np.concatenate(out))"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    outs = {}
    for name, extra in (("default", {}), ("separate_hessian", {"ICA_SEPARATE_HESSIAN": "1"}), ("direct_gather", {"ICA_ACCUM": "1"}),
                        ("short_tiles", {"ICA_MAX_TILE_H": "16"})):
        path = f"/tmp/ica_schedule_{name}.npy"
        subprocess.run([sys.executable, "-c", code, path], check=True, env=dict(os.environ, **extra), timeout=300)
        outs[name] = np.load(path)
    # per problem (L2 and robust, float32 and uint8 storage): 16 x 8 parameters, then 16 x 2 iteration counts
    ref = outs.pop("default").reshape(4, -1)
    np.testing.assert_array_equal(ref[:2], ref[2:])  # uint8 storage: the float32 storage's results, bit for bit
    for name, o in outs.items():
        o = o.reshape(4, -1)
        np.testing.assert_array_equal(o[:, 128:], ref[:, 128:], err_msg=name)                  # iteration counts
        # (the direct-gather kernel forms its per-pixel products in another order: fp32 rounding differs)
        tol = 1e-4 if name == "direct_gather" else 1e-6
        np.testing.assert_allclose(o[:, :128], ref[:, :128], rtol=tol, atol=tol * 1e-2, err_msg=name)


EDGE = [
    # name, H, W, C, t, r, ns, delta, nanifoutside, p0
    ("odd_width_dword_copy_path", 90, 101, 3, oc.HOMOGRAPHY, oc.LORENTZIAN, 2, 5, True, None),
    ("odd_width_affine_l2", 75, 83, 1, oc.AFFINITY, oc.QUADRATIC, 2, 4, True, None),
    ("narrower_than_a_tile", 48, 40, 3, oc.SIMILARITY, oc.QUADRATIC, 1, 3, True, None),
    ("tiny_image", 24, 24, 1, oc.TRANSLATION, oc.CHARBONNIER, 1, 2, True, None),
    ("no_nan_frame_delta0", 64, 96, 3, oc.EUCLIDEAN, oc.QUADRATIC, 1, 0, True, None),
    ("nanifoutside_false", 64, 96, 1, oc.AFFINITY, oc.GERMAN_MCCLURE, 1, 10, False, None),
    ("frame_swallows_everything", 32, 48, 1, oc.AFFINITY, oc.QUADRATIC, 1, 40, True, None),          # H = 0 -> dp = 0
    ("warp_entirely_outside", 64, 96, 1, oc.TRANSLATION, oc.QUADRATIC, 1, 5, True, [500.0, 300.0]),  # b = 0
    ("warp_entirely_outside_robust", 64, 96, 3, oc.HOMOGRAPHY, oc.CHARBONNIER, 1, 5, True,
     [0.0, 0.0, 900.0, 0.0, 0.0, -700.0, 0.0, 0.0]),
    ("five_scales_small", 320, 416, 1, oc.EUCLIDEAN, oc.QUADRATIC, 5, 2, True, None),               # 20x26 at the top
]


@pytest.mark.parametrize("case", EDGE, ids=[c[0] for c in EDGE])
def test_edge_cases_match_oracle(eng, case):
    name, H_, W_, C, t, r, ns, delta, nanif, p0 = case
    n = oc.nparams(t)
    I1, I2, _ = syn.make_pair(4000 + H_ + W_, H_, W_, C, t, robust_clutter=(r != 0))
    prob = eng.problem(1, H_, W_, C, t, r, nscales=ns, delta=delta, nanifoutside=nanif)
    pin = None if p0 is None else torch.tensor([p0], dtype=torch.float64)
    res = eng.solve(to_dev(eng, I1), to_dev(eng, I2), prob, p0=pin, want_images=True)
    p_w, e_w, DI_w, Iw_w, nit_w = oc.ica_pyramidal(I1.astype(np.float64), I2.astype(np.float64),
                                                   np.zeros(n) if p0 is None else np.array(p0, dtype=np.float64), t, ns, 0.5,
                                                   1e-3, r, 0.0, nanif, delta)
    assert list(res.niter[0].cpu().numpy()) == nit_w, name
    p = res.p[0, :n].cpu().numpy()
    scale = max(np.linalg.norm(p_w), 1e-3)
    assert np.linalg.norm(p - p_w) / scale <= 1e-4, (name, p, p_w)
    np.testing.assert_allclose(float(res.error[0]), e_w, rtol=2e-2, atol=1e-6)
    Iw = to_np(res.Iw)[0]
    assert np.array_equal(np.isnan(Iw), np.isnan(Iw_w)), name
    np.testing.assert_allclose(np.nan_to_num(Iw), np.nan_to_num(Iw_w), atol=2e-2)


@pytest.mark.parametrize("t", [oc.AFFINITY, oc.HOMOGRAPHY])
def test_exact_update_flag_matches_oracle_exact_composition(eng, t):
    """ICA_FLAG_EXACT_UPDATE: the exact composition p o dp^-1 instead of the reference's two deviating closed forms
    (transformation.py:106, :136); the oracle implements both."""
    from inverse_compositional_algorithm_amd._lib import ICA_FLAG_EXACT_UPDATE
    n = oc.nparams(t)
    I1, I2, _ = syn.make_pair(611 + t, 112, 144, 3, t)
    prob = eng.problem(1, 112, 144, 3, t, 0, nscales=2, delta=5, flags=ICA_FLAG_EXACT_UPDATE)
    res = eng.solve(to_dev(eng, I1), to_dev(eng, I2), prob)
    want, _, _, _, nit = oc.ica_pyramidal(I1.astype(np.float64), I2.astype(np.float64), np.zeros(n), t, 2, 0.5, 1e-3, 0, 0.0,
                                         True, 5, exact_update=True)
    ref_form, _, _, _, _ = oc.ica_pyramidal(I1.astype(np.float64), I2.astype(np.float64), np.zeros(n), t, 2, 0.5, 1e-3, 0,
                                            0.0, True, 5, exact_update=False)
    p = res.p[0, :n].cpu().numpy()
    assert list(res.niter[0].cpu().numpy()) == nit
    assert np.linalg.norm(p - want) / np.linalg.norm(want) <= 1e-4
    assert np.linalg.norm(p - want) < 0.1 * np.linalg.norm(want - ref_form) or np.allclose(want, ref_form, rtol=1e-7)


@pytest.mark.parametrize("nu,ns", [(0.6, 3), (0.75, 3), (0.4, 2)])
def test_full_solve_with_other_pyramid_factors(eng, nu, ns):
    """nu != 0.5: level sizes by round-half-even (zoom.py:20-21), non-dyadic rescale operators, zoom_in_parameters with
    nu' = max of the two size ratios (zoom.py:62-125)."""
    t, n = oc.SIMILARITY, 4
    I1, I2, _ = syn.make_pair(777, 150, 202, 3, t)
    prob = eng.problem(1, 150, 202, 3, t, oc.CHARBONNIER, nscales=ns, nu=nu, delta=4)
    res = eng.solve(to_dev(eng, I1), to_dev(eng, I2), prob)
    want, _, _, _, nit = oc.ica_pyramidal(I1.astype(np.float64), I2.astype(np.float64), np.zeros(n), t, ns, nu, 1e-3,
                                         oc.CHARBONNIER, 0.0, True, 4)
    assert list(res.niter[0].cpu().numpy()) == nit
    p = res.p[0, :n].cpu().numpy()
    assert np.linalg.norm(p - want) / np.linalg.norm(want) <= 1e-4


def test_odd_batch_and_short_iteration_cap(eng):
    """13 pairs (not a multiple of the 8 list shards / XCDs) and max_iter = 4: every pair stops at the cap or at TOL,
    exactly like max_iter single-scale oracle runs."""
    t, n, B = oc.AFFINITY, 6, 13
    pairs = [syn.make_pair(900 + i, 96, 128, 1, t) for i in range(B)]
    I1s, I2s = [a for a, _, _ in pairs], [b for _, b, _ in pairs]
    prob = eng.problem(B, 96, 128, 1, t, 0, nscales=1, delta=5, max_iter=4)
    res = eng.solve(to_dev(eng, *I1s), to_dev(eng, *I2s), prob)
    for i in range(B):
        want, _, _, _, nit = oc.ica_single_scale(I1s[i].astype(np.float64), I2s[i].astype(np.float64), np.zeros(n), t, 1e-3, 0,
                                                0.0, True, 5, max_iter=4)
        assert int(res.niter[i, 0]) == nit
        assert np.linalg.norm(res.p[i, :n].cpu().numpy() - want) / np.linalg.norm(want) <= 1e-4


@pytest.mark.parametrize("t,r", [(oc.HOMOGRAPHY, oc.LORENTZIAN), (oc.AFFINITY, oc.QUADRATIC)])
def test_batch_of_1024_pairs_against_the_oracle(eng, t, r):
    """The benchmark's batch size (1024 pairs per launch: full-chip grids, eight list shards of 128, converged pairs
    leaving the active lists at different iterations), at an image size the oracle can follow: every 37th pair of the
    batch is solved by the oracle too -- same iteration counts per scale, parameters within 1e-4."""
    B, H_, W_, n = 1024, 64, 80, oc.nparams(t)
    base = [syn.make_pair(7000 + 31 * t + i, H_, W_, 1, t, robust_clutter=(r != 0)) for i in range(32)]
    # 1024 distinct pairs from 32 generated ones: pair j = base pair (j mod 32) with I2 shifted by a few grey levels
    # (the clip range and every residual change with it, so the 32 copies of a base pair converge differently)
    I1 = np.stack([np.moveaxis(base[j % 32][0], 2, 0) for j in range(B)]).astype(np.float32)
    I2 = np.stack([np.clip(np.moveaxis(base[j % 32][1], 2, 0) + (j // 32) * 0.75, 0, 255) for j in range(B)]).astype(np.float32)
    prob = eng.problem(B, H_, W_, 1, t, r, nscales=2, nu=0.5, tol=1e-3, lambda_=0.0, delta=4)
    res = eng.solve(torch.from_numpy(I1).to(eng.device), torch.from_numpy(I2).to(eng.device), prob, want_stats=True)
    p, nit = res.p.cpu().numpy(), res.niter.cpu().numpy()
    assert res.stats["pair_iterations_total"] == int(nit.sum())
    assert len({tuple(x) for x in nit}) > 2, "the batch was meant to converge at different iterations"
    checked_converged = 0
    for j in range(0, B, 37):
        a = np.moveaxis(I1[j], 0, 2).astype(np.float64)
        b = np.moveaxis(I2[j], 0, 2).astype(np.float64)
        p_w, _, _, _, nit_w = oc.ica_pyramidal(a, b, np.zeros(n), t, 2, 0.5, 1e-3, r, 0.0, True, 4)
        if max(nit_w) < 30:
            assert list(nit[j]) == list(nit_w), f"pair {j}: iterations per scale {nit[j]} vs {nit_w}"
            rel = np.linalg.norm(p[j, :n] - p_w) / np.linalg.norm(p_w)
            assert rel <= P_REL_TOL, f"pair {j}: relative parameter error {rel:.2e}"
            assert corners_disp(p[j, :n], p_w, t, H_, W_) <= CORNER_TOL_PX
            checked_converged += 1
        else:
            # a scale that exhausts its 30 iterations is a non-converged trajectory (see the seeded random cases below):
            # the coarsest scale must still agree in its count, the final transforms by where they put the corners
            assert nit[j][-1] == nit_w[-1], f"pair {j}: iterations at the coarsest scale {nit[j]} vs {nit_w}"
            assert corners_disp(p[j, :n], p_w, t, H_, W_) <= 0.25
    assert checked_converged >= (8 if r == 0 else 0)


def _fuzz_cases(count, seed, hmax=150, wmax=210):
    rng = np.random.default_rng(seed)
    out = []
    for k in range(count):
        H_, W_ = int(rng.integers(40, hmax)), int(rng.integers(40, wmax))
        C = int(rng.choice([1, 3]))
        t = int(rng.integers(1, 6))
        r = int(rng.choice([0, 0, 2, 3, 4]))
        ns = int(rng.integers(1, 4))
        nu = float(rng.choice([0.5, 0.5, 0.6, 0.7]))
        delta = int(rng.integers(0, 8))
        lam = float(rng.choice([0.0, 0.0, 25.0]))
        out.append((k, H_, W_, C, t, r, ns, nu, delta, lam))
    return out


@pytest.mark.parametrize("case", _fuzz_cases(24, 20260101), ids=lambda c: "case%d_%dx%dx%d_t%d_r%d_s%d" % c[:7])
def test_seeded_random_configurations_match_oracle(eng, case):
    """Randomly drawn (size, channels, transform, error function, scales, nu, delta, lambda) combinations: iteration
    counts per scale and recovered parameters against the oracle."""
    k, H_, W_, C, t, r, ns, nu, delta, lam = case
    while min(H_, W_) * nu ** (ns - 1) < 24 + 2 * delta:  # keep a few valid pixels on the coarsest level
        ns -= 1
    ns = max(ns, 1)
    n = oc.nparams(t)
    I1, I2, _ = syn.make_pair(5000 + k, H_, W_, C, t, robust_clutter=(r != 0))
    prob = eng.problem(1, H_, W_, C, t, r, nscales=ns, nu=nu, delta=delta, lambda_=lam)
    res = eng.solve(to_dev(eng, I1), to_dev(eng, I2), prob, want_trace=True)
    traces = []
    want, e_w, _, _, nit = oc.ica_pyramidal(I1.astype(np.float64), I2.astype(np.float64), np.zeros(n), t, ns, nu, 1e-3, r, lam,
                                           True, delta, traces=traces)
    got_nit = list(res.niter[0].cpu().numpy())
    p = res.p[0, :n].cpu().numpy()
    if max(nit) < 30:
        assert got_nit == nit, (case, got_nit, nit)
        assert np.linalg.norm(p - want) / max(np.linalg.norm(want), 1e-3) <= 1e-4, (case, p, want)
        return
    # A scale that exhausts its 30 iterations is a non-converged trajectory, the kind most sensitive to fp32-vs-float64
    # drift.  Every scale before it (coarse to fine) must still agree in iteration count and iterate by iterate; on the
    # exhausting scale itself both sides must run all 30 iterations and the first iterates must agree; the scales
    # after it started from slightly different parameters, so only the final transforms are compared, by how far
    # they move the image corners.
    tr = res.trace[0].cpu().numpy()              # [scale][iteration][|dp|, p0..p7, lambda]
    sizes = [(W_, H_)]
    for _ in range(1, ns):
        sizes.append(oc.zoom_size(sizes[-1][0], sizes[-1][1], nu))
    first_cap = next(s for s, _ in traces if nit[s] >= 30)   # traces run coarse to fine
    for s, its in traces:
        if s < first_cap:
            break
        assert got_nit[s] == nit[s], (case, s, got_nit, nit)
        upto = len(its) if s > first_cap else 8
        for it in range(min(upto, len(its))):
            d = corners_disp(tr[s, it, 1:1 + n], its[it]["p"], t, sizes[s][1], sizes[s][0])
            assert d <= 5e-3, (case, s, it, d)
    assert corners_disp(p, want, t, H_, W_) <= 0.25, (case, p, want)


def test_one_engine_under_two_streams_matches_serial(eng):
    """ica_set_stream orders a newly bound stream behind the work queued on the previous one (a context owns one set of
    workspaces): two solves issued back to back on ONE engine under two torch streams give exactly the serial results."""
    t, n, B = oc.AFFINITY, 6, 24
    sets = []
    for base in (3100, 3200):
        prs = [syn.make_pair(base + i, 96, 128, 1, t) for i in range(B)]
        sets.append((to_dev(eng, *[a for a, _, _ in prs]), to_dev(eng, *[b for _, b, _ in prs])))
    prob = eng.problem(B, 96, 128, 1, t, 0, nscales=2, delta=5)
    serial = [eng.solve(a, b, prob) for a, b in sets]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=eng.device) for _ in sets]
    out = []
    for (a, b), st in zip(sets, streams):
        with torch.cuda.stream(st):
            out.append(eng.solve(a, b, prob))
    torch.cuda.synchronize()
    for r, w in zip(out, serial):
        assert torch.equal(r.p, w.p) and torch.equal(r.niter, w.niter) and torch.equal(r.error, w.error)


@pytest.mark.parametrize("C", [2, 4])
def test_two_and_four_channel_images(eng, C):
    """The reference's robust entry point takes any channel count (ref :168-177); 2 and 4 channels run the
    direct-gather kernel.  One robust single-scale solve and one L2 pyramid against the oracle."""
    t, n = oc.SIMILARITY, 4
    I1, I2, _ = syn.make_pair(4100 + C, 90, 120, C, t, robust_clutter=True)
    for robust, ns in ((oc.LORENTZIAN, 1), (oc.QUADRATIC, 2)):
        prob = eng.problem(1, 90, 120, C, t, robust, nscales=ns, delta=5)
        res = eng.solve(to_dev(eng, I1), to_dev(eng, I2), prob, want_images=True)
        want, _, DI, Iw, nit = oc.ica_pyramidal(I1.astype(np.float64), I2.astype(np.float64), np.zeros(n), t, ns, 0.5, 1e-3,
                                               robust, 0.0, True, 5)
        assert list(res.niter[0].cpu().numpy()) == nit
        p = res.p[0, :n].cpu().numpy()
        assert np.linalg.norm(p - want) / np.linalg.norm(want) <= P_REL_TOL
        assert np.array_equal(np.isnan(to_np(res.Iw)[0]), np.isnan(Iw))


def _skimage_cases():
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "skimage_warp.npz"))
    kind = {"translation": oc.TRANSLATION, "integer_translation": oc.TRANSLATION, "euclidean": oc.EUCLIDEAN,
            "similarity": oc.SIMILARITY, "affinity": oc.AFFINITY, "affinity_round_numbers": oc.AFFINITY,
            "homography": oc.HOMOGRAPHY, "homography_gray_large_shift": oc.HOMOGRAPHY}
    return [(n, kind[n], z[n + "__img"].astype(np.float64), z[n + "__M"], z[n + "__out"]) for n in sorted(kind)]


@pytest.mark.parametrize("case", _skimage_cases(), ids=lambda c: c[0])
def test_warp_against_scikit_image_outputs(eng, case):
    """The HIP warp against genuine scikit-image 0.18.3 outputs (tests/golden/skimage_warp.npz; reference call site
    bicubic_interpolation.py:199): NaN mask identical -- including the maps whose sampling positions are exact
    integers, where the in/out decision of a border pixel hangs on floor() -- and values within fp32 rounding."""
    name, t, img, M, want = case
    p = oc.matrix2params(M, t)
    np.testing.assert_allclose(oc.params2matrix(p, t), M, atol=1e-14)
    Iw = to_np(eng.warp(to_dev(eng, img), torch.from_numpy(p[None]), t))[0]
    assert np.array_equal(np.isnan(Iw), np.isnan(want)), name
    np.testing.assert_allclose(np.nan_to_num(Iw), np.nan_to_num(want), atol=2e-3)


@pytest.mark.parametrize("t", TYPES)
@pytest.mark.parametrize("C,W_", [(1, 128), (3, 128), (1, 131), (3, 90)])
def test_uint8_storage_matches_float_storage_bit_for_bit(eng, t, C, W_):
    """uint8 level-0 images (ICA_STORAGE_U8): every transform, gray and RGB, L2 and robust, one scale and a pyramid, widths
    that are / are not a multiple of 4 pixels (whole-group copies vs the element-wise fill), a batch in which pairs
    converge at different iterations -- results identical to the float32 storage, bit for bit."""
    n, B, H_ = oc.nparams(t), 5, 96
    prs = [syn.make_pair(6100 + 13 * t + i, H_, W_, C, t, robust_clutter=True) for i in range(B)]
    f1, f2 = to_dev(eng, *[a for a, _, _ in prs]), to_dev(eng, *[b for _, b, _ in prs])
    u1, u2 = f1.to(torch.uint8), f2.to(torch.uint8)
    assert torch.equal(u1.float(), f1) and torch.equal(u2.float(), f2)
    for robust, ns in ((oc.QUADRATIC, 2), (oc.CHARBONNIER, 1), (oc.LORENTZIAN, 3)):
        prob = eng.problem(B, H_, W_, C, t, robust, nscales=ns, delta=4)
        a = eng.solve(f1, f2, prob, want_images=True, want_trace=True)
        b = eng.solve(u1, u2, prob, want_images=True, want_trace=True)
        assert torch.equal(a.p, b.p) and torch.equal(a.niter, b.niter) and torch.equal(a.error, b.error)
        assert torch.equal(torch.nan_to_num(a.trace), torch.nan_to_num(b.trace))
        assert torch.equal(torch.isnan(a.Iw), torch.isnan(b.Iw)) and torch.equal(torch.nan_to_num(a.DI), torch.nan_to_num(b.DI))
    # the operator hook too: H and b of one iteration
    prob = eng.problem(B, H_, W_, C, t, oc.LORENTZIAN, delta=4)
    p = np.zeros((B, 8))
    for i, (_, _, g) in enumerate(prs):
        p[i, :n] = 0.8 * g
    Ha, ba = eng.normal_equations(f1, f2, torch.from_numpy(p), prob, lambda_it=30.0)
    Hb, bb = eng.normal_equations(u1, u2, torch.from_numpy(p), prob, lambda_it=30.0)
    assert torch.equal(Ha, Hb) and torch.equal(ba, bb)


def test_uint8_storage_rejects_what_it_does_not_cover(eng):
    x = torch.zeros(1, 2, 32, 32, dtype=torch.uint8, device=eng.device)
    from inverse_compositional_algorithm_amd._lib import IcaError
    with pytest.raises(IcaError) as e:
        eng.solve(x, x, eng.problem(1, 32, 32, 2, oc.TRANSLATION, 0))
    assert e.value.status == -1  # 2- and 4-channel images are float32 only
    with pytest.raises(ValueError, match="same dtype"):
        eng.solve(x[:, :1].contiguous(), x[:, :1].float().contiguous(), eng.problem(1, 32, 32, 1, oc.TRANSLATION, 0))


@pytest.mark.parametrize("nu", [0.1, 0.07])
def test_uint8_storage_with_a_very_small_pyramid_factor(eng, nu):
    """uint8 level-0 images through the generic pyramid pass (nu below ~0.12: a window of 16 outputs wider than the tiled
    pass is sized for): the same results as float32 storage, like every other nu."""
    B, H_, W_, C, t = 3, 200, 240, 1, oc.TRANSLATION
    prs = [syn.make_pair(7300 + i, H_, W_, C, t) for i in range(B)]
    f1, f2 = to_dev(eng, *[a for a, _, _ in prs]), to_dev(eng, *[b for _, b, _ in prs])
    prob = eng.problem(B, H_, W_, C, t, oc.QUADRATIC, nscales=2, nu=nu, delta=2)
    a = eng.solve(f1, f2, prob)
    b = eng.solve(f1.to(torch.uint8), f2.to(torch.uint8), prob)
    assert torch.equal(a.p, b.p) and torch.equal(a.niter, b.niter) and torch.equal(a.error, b.error)


def test_a_pair_too_large_for_32_bit_offsets_is_refused(eng):
    """One pair's planes are addressed through 32-bit buffer offsets: C * H * W * 4 bytes must stay below 2^31."""
    from inverse_compositional_algorithm_amd._lib import IcaError
    prob = eng.problem(1, 13400, 13400, 3, oc.TRANSLATION, 0)
    import ctypes
    from inverse_compositional_algorithm_amd._lib import check
    with pytest.raises(IcaError) as e:  # the descriptor is refused before anything is allocated or launched
        check(eng.lib.ica_prepare(eng.ctx, ctypes.byref(prob)), eng.ctx)
    assert e.value.status == -6  # ICA_EINVAL_ARG


def test_operator_wrappers_check_shapes(eng):
    """The reference's NumPy operators fail on mismatched operands; so do the wrappers (before any device access)."""
    Ix = np.zeros((6, 7, 3))
    with pytest.raises(ValueError):
        eng.ref_steepest_descent(Ix, np.zeros((6, 7, 2)), np.zeros((6, 7, 4)), 2)
    with pytest.raises(ValueError):
        eng.ref_steepest_descent(Ix, Ix, np.zeros((6, 7, 6)), 2)   # J must be (ny, nx, 2 * nparams)
    with pytest.raises(ValueError):
        eng.ref_normal(np.zeros((6, 7, 3, 2)), DI=np.zeros((6, 7, 2)), want_b=True)
    with pytest.raises(ValueError):
        eng.ref_normal(np.zeros((6, 7, 3, 2)), rho=np.zeros((6, 6)))
    with pytest.raises(ValueError):
        eng.ref_robust_error(np.zeros((6, 7)), 5.0, 3)
