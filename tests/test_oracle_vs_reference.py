"""Compare every oracle function with the reference function it restates, imported from
/root/reference by oracle/ref_harness.py (build container only; skipped on the GPU box)."""
import numpy as np
import pytest

from conftest import needs_reference
from oracle import ica_oracle as oc
from oracle import ref_harness

pytestmark = needs_reference
TYPES = [oc.TRANSLATION, oc.EUCLIDEAN, oc.SIMILARITY, oc.AFFINITY, oc.HOMOGRAPHY]


@pytest.fixture(scope="module")
def ref():
    return ref_harness.load()


def _rt(ref, t):
    return ref.tr.TransformType(t)


@pytest.mark.parametrize("t", TYPES)
def test_nparams_matrix_roundtrip(ref, t):
    rng = np.random.default_rng(t)
    assert oc.nparams(t) == _rt(ref, t).nparams()
    p = rng.normal(size=oc.nparams(t)) * 0.1
    M = oc.params2matrix(p, t)
    np.testing.assert_allclose(M, ref.tr.params2matrix(p, _rt(ref, t)), rtol=0, atol=0)
    np.testing.assert_allclose(oc.matrix2params(M, t), np.asarray(ref.tr.matrix2params(M, _rt(ref, t))), atol=0)


@pytest.mark.parametrize("t", TYPES)
def test_dropin_project_matches_reference(ref, t):
    """transformation.project (ref :144-185): type given by the number of parameters; scalars and arrays; error text."""
    from inverse_compositional_algorithm_amd import transformation as ours
    rng = np.random.default_rng(40 + t)
    n = oc.nparams(t)
    for _ in range(20):
        p = rng.normal(size=n) * (1e-3 if t == oc.HOMOGRAPHY else 0.1)
        x, y = rng.uniform(0, 500, 7), rng.uniform(0, 400, 7)
        gx, gy = ours.project(x, y, p, n)
        wx, wy = ref.tr.project(x, y, p, n)
        np.testing.assert_allclose(gx, wx, rtol=1e-13, atol=1e-11)
        np.testing.assert_allclose(gy, wy, rtol=1e-13, atol=1e-11)
        sx, sy = ours.project(3.0, 5.0, p, n)
        rx, ry = ref.tr.project(3.0, 5.0, p, n)
        assert abs(sx - rx) < 1e-11 and abs(sy - ry) < 1e-11
    with pytest.raises(ValueError) as e_ref:
        ref.tr.project(1.0, 2.0, np.zeros(5), 5)
    with pytest.raises(ValueError) as e_ours:
        ours.project(1.0, 2.0, np.zeros(5), 5)
    assert str(e_ref.value) == str(e_ours.value)


@pytest.mark.parametrize("t", TYPES)
def test_update_transform_matches_reference_formulas(ref, t):
    """Includes the AFFINITY p[1] and HOMOGRAPHY p[4] deviations (transformation.py:106,136)."""
    rng = np.random.default_rng(10 + t)
    n = oc.nparams(t)
    scale = np.ones(n) * 0.05
    if t == oc.HOMOGRAPHY:
        scale[[2, 5]] = 5.0
        scale[[6, 7]] = 1e-4
    else:
        scale[:2] = 5.0
    for _ in range(200):
        p = rng.normal(size=n) * scale
        dp = rng.normal(size=n) * scale * 0.3
        want = ref.tr.update_transform(p.copy(), dp.copy(), _rt(ref, t))
        got = oc.update_transform(p.copy(), dp.copy(), t)
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-13)
    # in place + returned
    p = rng.normal(size=n) * scale
    out = oc.update_transform(p, rng.normal(size=n) * scale * 0.1, t)
    assert out is p


def test_update_transform_degenerate_det_is_skipped(ref):
    p = np.array([1.0, 2.0, 0.1, 0.2])
    dp = np.array([0.3, 0.4, -1.0, 0.0])  # similarity det = (1+a)^2+b^2 = 0
    want = ref.tr.update_transform(p.copy(), dp.copy(), ref.tr.TransformType.SIMILARITY)
    np.testing.assert_array_equal(oc.update_transform(p.copy(), dp, oc.SIMILARITY), want)
    np.testing.assert_array_equal(want, p)


def test_exact_update_differs_only_where_documented():
    rng = np.random.default_rng(5)
    for t, idx in ((oc.AFFINITY, 1), (oc.HOMOGRAPHY, 4)):
        n = oc.nparams(t)
        p, dp = rng.normal(size=n) * 0.05, rng.normal(size=n) * 0.05
        a = oc.update_transform(p.copy(), dp, t)
        e = oc.update_transform(p.copy(), dp, t, exact=True)
        M = oc.params2matrix(p, t) @ np.linalg.inv(oc.params2matrix(dp, t))
        np.testing.assert_allclose(e, oc.matrix2params(M / M[2, 2], t), atol=1e-13)
        diff = np.abs(a - e) > 1e-15
        assert diff[idx] and diff.sum() == 1


@pytest.mark.parametrize("t", TYPES)
def test_jacobian_sd_hessian_rhs(ref, t):
    rng = np.random.default_rng(20 + t)
    ny, nx, nz = 13, 17, 3
    n = oc.nparams(t)
    J = oc.jacobian(t, nx, ny)
    np.testing.assert_array_equal(J, ref.de.jacobian(_rt(ref, t), nx, ny))
    I1 = rng.uniform(0, 255, (ny, nx, nz))
    Ix, Iy = oc.gradient(I1, True, 2)
    DIJ = oc.steepest_descent_images(Ix, Iy, J, n)
    want = ref.io.steepest_descent_images(Ix, Iy, J, n)
    np.testing.assert_array_equal(np.isnan(DIJ), np.isnan(want))
    np.testing.assert_allclose(np.nan_to_num(DIJ), np.nan_to_num(want), rtol=1e-15)
    np.testing.assert_allclose(oc.hessian(DIJ), ref.de.hessian(want), rtol=1e-13)
    rho = rng.uniform(0.1, 1, (ny, nx))
    np.testing.assert_allclose(oc.hessian(DIJ, rho), ref.de.hessian_robust(want, rho, n), rtol=1e-13)
    DI = rng.normal(size=(ny, nx, nz))
    DI[3:5, 4:9] = np.nan
    np.testing.assert_allclose(oc.independent_vector(DIJ, DI), ref.io.independent_vector(want, DI, n), rtol=1e-12)
    np.testing.assert_allclose(oc.independent_vector(DIJ, DI, rho),
                               ref.io.independent_vector_robust(want, DI, rho, n), rtol=1e-12)


def test_reference_unit_test_tables():
    """Known answers of test/test_derivatives.py:11-68 (2x2 grid Jacobians)."""
    J = oc.jacobian(oc.EUCLIDEAN, 2, 2)
    np.testing.assert_array_equal(J[1, 1], [1, 0, -1, 0, 1, 1])
    J = oc.jacobian(oc.SIMILARITY, 2, 2)
    np.testing.assert_array_equal(J[1, 0], [1, 0, 0, -1, 0, 1, 1, 0])
    J = oc.jacobian(oc.AFFINITY, 2, 2)
    np.testing.assert_array_equal(J[1, 1], [1, 0, 1, 1, 0, 0, 0, 1, 0, 0, 1, 1])
    np.testing.assert_array_equal(oc.jacobian(oc.TRANSLATION, 2, 2)[0, 1], [1, 0, 0, 1])
    # test_derivatives.py:71-83 / :97-101
    DIJ = np.random.default_rng(0).random((3, 3, 4, 2))
    want = sum(DIJ[i, j].T @ DIJ[i, j] for i in range(3) for j in range(3))
    np.testing.assert_allclose(oc.hessian(DIJ), want)
    assert oc.hessian(np.zeros((0, 0, 0, 0))).shape == (0, 0)


@pytest.mark.parametrize("r", [oc.QUADRATIC, oc.GERMAN_MCCLURE, oc.LORENTZIAN, oc.CHARBONNIER])
def test_rho(ref, r):
    rng = np.random.default_rng(30 + r)
    DI = rng.normal(size=(9, 11, 3)) * 20
    DI[2, 3] = np.nan
    rr = ref.io.RobustErrorFunctionType(r)
    np.testing.assert_allclose(oc.robust_error_function(DI, 7.5, r), ref.io.robust_error_function(DI, 7.5, rr),
                               rtol=1e-15)


def test_truncated_quadratic_is_unpinned(ref):
    with pytest.raises(ValueError):  # image_optimisation.py:40 -- array truth value
        ref.io.robust_error_function(np.ones((4, 4, 3)), 1.0, ref.io.RobustErrorFunctionType.TRUNCATED_QUADRATIC)
    out = oc.robust_error_function(np.array([[[0.1, 0, 0], [3.0, 0, 0]]]), 1.0, oc.TRUNCATED_QUADRATIC)
    np.testing.assert_array_equal(out, [[1.0, 0.0]])


def test_inverse_and_solve(ref):
    rng = np.random.default_rng(3)
    A = rng.normal(size=(8, 8))
    H = A @ A.T
    np.testing.assert_array_equal(oc.inverse_hessian(H), ref.de.inverse_hessian(H, 8))
    np.testing.assert_array_equal(oc.inverse_hessian(np.zeros((3, 3))), np.zeros((3, 3)))
    np.testing.assert_array_equal(ref.de.inverse_hessian(np.zeros((3, 3)), 3), np.zeros((3, 3)))
    b = rng.normal(size=8)
    e1, d1 = oc.parametric_solve(oc.inverse_hessian(H), b)
    e2, d2 = ref.io.parametric_solve(ref.de.inverse_hessian(H, 8), b, 8)
    assert e1 == e2 and np.array_equal(d1, d2)


@pytest.mark.parametrize("t", TYPES)
def test_zoom(ref, t):
    p = np.arange(1, oc.nparams(t) + 1, dtype=float) * 0.01
    np.testing.assert_array_equal(oc.zoom_in_parameters(p, t, 68, 120, 135, 240),
                                  ref.zm.zoom_in_parameters(p, _rt(ref, t), 68, 120, 135, 240))
    assert oc.zoom_size(135, 1080, 0.5) == ref.zm.zoom_size(135, 1080, 0.5) == (68, 540)


@pytest.mark.parametrize("t,r,ns", [(oc.TRANSLATION, oc.QUADRATIC, 1), (oc.EUCLIDEAN, oc.QUADRATIC, 2),
                                     (oc.SIMILARITY, oc.LORENTZIAN, 1), (oc.AFFINITY, oc.QUADRATIC, 2),
                                     (oc.AFFINITY, oc.GERMAN_MCCLURE, 1), (oc.HOMOGRAPHY, oc.QUADRATIC, 2),
                                     (oc.HOMOGRAPHY, oc.LORENTZIAN, 2), (oc.HOMOGRAPHY, oc.CHARBONNIER, 1)])
def test_drivers_match_reference_loops(ref, t, r, ns, capsys):
    """Whole drivers: the oracle's loops vs the reference's own loops (reference code verbatim + the two
    restated skimage calls), iteration by iteration."""
    from inverse_compositional_algorithm_amd import synthetic as syn
    I1, I2, _ = syn.make_pair(77 + 10 * t + r, 96, 128, 3, t, robust_clutter=(r != 0))
    I1, I2 = I1.astype(np.float64), I2.astype(np.float64)
    n = oc.nparams(t)
    traces = []
    p_o, e_o, DI_o, Iw_o, nit = oc.ica_pyramidal(I1, I2, np.zeros(n), t, ns, 0.5, 1e-3, r, 0.0, True, 5,
                                                 traces=traces, need_rgb=True)
    p_r, e_r, DI_r, Iw_r = ref.ica.pyramidal_inverse_compositional_algorithm(
        I1=I1, I2=I2, p=np.zeros(n), transform_type=_rt(ref, t), nscales=ns, nu=0.5, TOL=1e-3,
        robust_type=ref.io.RobustErrorFunctionType(r), lambda_=0.0, nanifoutside=True, delta=5, verbose=True)
    out = capsys.readouterr().out
    n_lines = sum(1 for ln in out.splitlines() if "|Dp|=" in ln)
    assert n_lines == sum(nit)
    np.testing.assert_allclose(p_o, p_r, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(e_o, e_r, rtol=1e-7)
    np.testing.assert_array_equal(np.isnan(Iw_o), np.isnan(Iw_r))
    np.testing.assert_allclose(np.nan_to_num(Iw_o), np.nan_to_num(Iw_r), atol=1e-9)
    np.testing.assert_allclose(np.nan_to_num(DI_o), np.nan_to_num(DI_r), atol=1e-9)


def test_error_messages(ref):
    I = np.zeros((32, 32, 3))
    for fn in (lambda: oc.ica_single_scale(I, I[:-1], np.zeros(2), 1, 1e-3),
               lambda: ref.ica.robust_inverse_compositional_algorithm(I, I[:-1], np.zeros(2), _rt(ref, 1), 1e-3,
                                                                      ref.io.RobustErrorFunctionType(4), 0., True, 5, False)):
        with pytest.raises(ValueError, match="same dimensions"):
            fn()
    with pytest.raises(ValueError, match="TOL must be positive"):
        oc.ica_single_scale(I, I, np.zeros(2), 1, 0.01)
    with pytest.raises(ValueError, match="RGB images"):
        oc.ica_pyramidal(I[..., :1], I[..., :1], np.zeros(2), 1, 1, 0.5, 1e-3, need_rgb=True)
