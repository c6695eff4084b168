"""Host-side mirror of the reference interface (no GPU): enums, 8-number algebra, config reader, argument checks."""
import os

import numpy as np
import pytest

from inverse_compositional_algorithm_amd import configuration_handler as cfh
from inverse_compositional_algorithm_amd import constants, transformation as tr, zoom
from inverse_compositional_algorithm_amd.image_optimisation import RobustErrorFunctionType
from inverse_compositional_algorithm_amd.transformation import TransformType
from oracle import ica_oracle as oc

DATA = os.path.join(os.path.dirname(__file__), "data")


def test_enums_and_constants():
    assert [t.value for t in TransformType] == [1, 2, 3, 4, 5] and [t.nparams() for t in TransformType] == [2, 3, 4, 6, 8]
    assert [r.name for r in RobustErrorFunctionType] == ["QUADRATIC", "TRUNCATED_QUADRATIC", "GERMAN_MCCLURE",
                                                         "LORENTZIAN", "CHARBONNIER"]
    assert (constants.MAX_ITER, constants.LAMBDA_0, constants.LAMBDA_N, constants.LAMBDA_RATIO) == (30, 80.0, 5.0, 0.9)


@pytest.mark.parametrize("t", list(TransformType))
def test_transform_algebra_matches_oracle(t):
    rng = np.random.default_rng(t.value)
    n = t.nparams()
    sc = np.full(n, 0.05)
    if t == TransformType.HOMOGRAPHY:
        sc[[2, 5]], sc[[6, 7]] = 5.0, 1e-4
    else:
        sc[:2] = 5.0
    for _ in range(50):
        p, dp = rng.normal(size=n) * sc, rng.normal(size=n) * sc * 0.3
        np.testing.assert_array_equal(tr.params2matrix(p, t), oc.params2matrix(p, t.value))
        np.testing.assert_allclose(tr.matrix2params(tr.params2matrix(p, t), t), p, atol=1e-15)
        for exact in (False, True):
            np.testing.assert_allclose(tr.update_transform(p.copy(), dp, t, exact=exact),
                                       oc.update_transform(p.copy(), dp, t.value, exact=exact), rtol=1e-11, atol=1e-13)
        np.testing.assert_array_equal(zoom.zoom_in_parameters(p, t, 68, 120, 135, 240),
                                      oc.zoom_in_parameters(p, t.value, 68, 120, 135, 240))
    q = rng.normal(size=n) * sc
    assert tr.update_transform(q, np.zeros(n), t) is q  # in place, like the reference (transformation.py:54,...)
    assert zoom.zoom_size(135, 1080, 0.5) == (68, 540)


def test_config_reader_accepts_both_reference_flavours(tmp_path):
    a = cfh.read_config_file(os.path.join(DATA, "config_plain.ini"))
    assert a["inverse_compositional_algorithm"] == {"TOL": 1e-3, "transform_type": TransformType.AFFINITY, "verbose": False}
    assert a["robust_inverse_compositional_algorithm"]["robust_type"] == RobustErrorFunctionType.LORENTZIAN
    assert a["robust_inverse_compositional_algorithm"]["lambda"] == 12.5
    assert a["pyramidal_inverse_compositional_algorithm"]["pyramid_levels"] == 4
    b = cfh.read_config_file(os.path.join(DATA, "config_inline_comments.ini"))  # the root config.ini style
    assert b["pyramidal_inverse_compositional_algorithm"]["transform_type"] == TransformType.TRANSLATION
    assert b["robust_inverse_compositional_algorithm"]["robust_type"] == RobustErrorFunctionType.GERMAN_MCCLURE
    out = tmp_path / "c.ini"
    cfh.create_config_file(str(out))
    c = cfh.read_config_file(str(out))
    assert set(c) == {"inverse_compositional_algorithm", "robust_inverse_compositional_algorithm",
                      "pyramidal_inverse_compositional_algorithm"}
    assert c["pyramidal_inverse_compositional_algorithm"]["pyramid_levels"] == 2


def test_dropin_signatures_and_argument_errors():
    """Same keyword names and the same ValueErrors as the reference (ref :48-60, :169-177, :300-311), raised before
    anything touches the GPU."""
    import inspect
    from inverse_compositional_algorithm_amd import inverse_compositional_algorithm as ica
    assert list(inspect.signature(ica.inverse_compositional_algorithm).parameters) == [
        "I1", "I2", "p", "transform_type", "TOL", "nanifoutside", "delta", "verbose"]
    assert list(inspect.signature(ica.robust_inverse_compositional_algorithm).parameters) == [
        "I1", "I2", "p", "transform_type", "TOL", "robust_type", "lambda_", "nanifoutside", "delta", "verbose"]
    assert list(inspect.signature(ica.pyramidal_inverse_compositional_algorithm).parameters) == [
        "I1", "I2", "p", "transform_type", "nscales", "nu", "TOL", "robust_type", "lambda_", "nanifoutside", "delta",
        "verbose"]
    rgb, gray = np.zeros((32, 40, 3)), np.zeros((32, 40, 1))
    T, R = TransformType.AFFINITY, RobustErrorFunctionType.CHARBONNIER
    with pytest.raises(ValueError, match="RGB images with channels in the last dimension"):
        ica.inverse_compositional_algorithm(gray, gray, np.zeros(6), T, 1e-3, True, 5, False)
    with pytest.raises(ValueError, match="RGB images"):
        ica.pyramidal_inverse_compositional_algorithm(rgb[..., 0], rgb[..., 0], np.zeros(6), T, 2, 0.5, 1e-3, R, 0., True, 5,
                                                      False)
    with pytest.raises(ValueError, match="same dimensions"):
        ica.robust_inverse_compositional_algorithm(rgb, rgb[:-1], np.zeros(6), T, 1e-3, R, 0., True, 5, False)
    with pytest.raises(ValueError, match="TOL must be positive and very small"):
        ica.inverse_compositional_algorithm(rgb, rgb, np.zeros(6), T, 0.01, True, 5, False)
    # non-finite pixels: the reference's result on them is an accident of np.clip with NaN bounds (its residuals are
    # zero-filled element by element, image_optimisation.py:99, but the warp is clipped to [I2.min(), I2.max()]); the
    # drop-in refuses them by name instead of computing something else silently
    bad = rgb.copy()
    bad[3, 4, 1] = np.nan
    for a, b_ in ((bad, rgb), (rgb, bad)):
        with pytest.raises(ValueError, match="finite values only"):
            ica.robust_inverse_compositional_algorithm(a, b_, np.zeros(6), T, 1e-3, R, 0., True, 5, False)
    bad[3, 4, 1] = np.inf
    with pytest.raises(ValueError, match="finite values only"):
        ica.pyramidal_inverse_compositional_algorithm(rgb, bad, np.zeros(6), T, 2, 0.5, 1e-3, R, 0., True, 5, False)


def test_synthetic_generator_is_deterministic():
    from inverse_compositional_algorithm_amd import synthetic as syn
    a = syn.make_pair(5, 64, 80, 3, 5, robust_clutter=True)
    b = syn.make_pair(5, 64, 80, 3, 5, robust_clutter=True)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert a[0].dtype == np.float32 and a[0].min() >= 0 and a[0].max() <= 255 and np.all(a[0] == np.round(a[0]))


def test_project_is_the_matrix_action():
    """transformation.project (ref transformation.py:144-185): the type is given by the number of parameters."""
    from inverse_compositional_algorithm_amd import transformation as tr
    rng = np.random.default_rng(5)
    x, y = rng.uniform(0, 300, 9), rng.uniform(0, 200, 9)
    for t, n in ((1, 2), (2, 3), (3, 4), (4, 6), (5, 8)):
        p = rng.normal(size=n) * (1e-3 if t == 5 else 0.1)
        M = tr.params2matrix(p, t)
        h = M @ np.stack([x, y, np.ones_like(x)])
        gx, gy = tr.project(x, y, p, n)
        np.testing.assert_allclose(gx, h[0] / h[2], rtol=1e-13)
        np.testing.assert_allclose(gy, h[1] / h[2], rtol=1e-13)
    with pytest.raises(ValueError, match="Invalid transformation type"):
        tr.project(1.0, 2.0, np.zeros(5), 5)
