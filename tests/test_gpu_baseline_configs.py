"""BASELINE.json configs as parity cases at (or near) their full image sizes.

Against the oracle where it finishes in seconds (one pair each), and through size-independent properties on small
batches at full size: the solver recovers the known ground-truth transform of a synthetic pair, results do not depend
on the position of a pair in the batch or on the batch size, and reruns are bit-identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from inverse_compositional_algorithm_amd import synthetic as syn  # noqa: E402
from oracle import ica_oracle as oc  # noqa: E402

# (name, H, W, C, transform, robust, nscales)
CONFIGS = {
    "configs[0]": (256, 256, 1, oc.AFFINITY, oc.QUADRATIC, 1),
    "configs[1]": (512, 512, 1, oc.AFFINITY, oc.QUADRATIC, 3),
    "configs[2]": (1024, 1024, 1, oc.HOMOGRAPHY, oc.QUADRATIC, 4),
    "configs[3]": (1024, 1024, 1, oc.HOMOGRAPHY, oc.LORENTZIAN, 4),
    "configs[4]": (1080, 1920, 3, oc.HOMOGRAPHY, oc.LORENTZIAN, 5),
}


@pytest.fixture(scope="module")
def eng():
    from inverse_compositional_algorithm_amd.engine import IcaEngine
    return IcaEngine(0)


def corner_err(p, q, t, H, W):
    Mp, Mq = oc.params2matrix(p, t), oc.params2matrix(q, t)
    d = 0.0
    for x, y in ((0, 0), (W - 1, 0), (0, H - 1), (W - 1, H - 1)):
        a, b = Mp @ [x, y, 1.0], Mq @ [x, y, 1.0]
        d = max(d, float(np.hypot(a[0] / a[2] - b[0] / b[2], a[1] / a[2] - b[1] / b[2])))
    return d


@pytest.mark.parametrize("name", ["configs[0]", "configs[1]", "configs[2]"])
def test_one_pair_against_the_oracle_at_full_size(eng, name):
    H, W, C, t, r, ns = CONFIGS[name]
    n = oc.nparams(t)
    I1, I2, _ = syn.make_pair(31337 + H, H, W, C, t, robust_clutter=(r != 0))
    d1 = torch.from_numpy(np.ascontiguousarray(np.moveaxis(I1, 2, 0))[None]).to(eng.device)
    d2 = torch.from_numpy(np.ascontiguousarray(np.moveaxis(I2, 2, 0))[None]).to(eng.device)
    res = eng.solve(d1, d2, eng.problem(1, H, W, C, t, r, nscales=ns, delta=10))
    p = res.p[0, :n].cpu().numpy()
    want, _, _, _, nit = oc.ica_pyramidal(I1.astype(np.float64), I2.astype(np.float64), np.zeros(n), t, ns, 0.5, 1e-3, r,
                                         0.0, True, 10)
    assert list(res.niter[0].cpu().numpy()) == nit
    assert np.linalg.norm(p - want) / np.linalg.norm(want) <= 1e-4
    assert corner_err(p, want, t, H, W) <= 1e-3


@pytest.mark.parametrize("name", list(CONFIGS))
def test_properties_at_full_size(eng, name):
    H, W, C, t, r, ns = CONFIGS[name]
    n = oc.nparams(t)
    B = 6 if C == 1 else 3
    I1, I2, p_gt = eng.make_batch(4242, B, H, W, C, t, robust_clutter=(r != 0))
    prob = eng.problem(B, H, W, C, t, r, nscales=ns, delta=10)
    a = eng.solve(I1, I2, prob)
    # 1. the known transform is recovered (I1 was resampled from I2 under p_gt, then rounded to 8 bits / cluttered)
    p, gt = a.p.cpu().numpy(), p_gt.cpu().numpy()
    conv = a.error.cpu().numpy() < 1e-3
    assert conv.sum() >= B - 1
    for i in np.nonzero(conv)[0]:
        assert corner_err(p[i, :n], gt[i, :n], t, H, W) < (0.05 if r == 0 else 1.0), (name, i)  # clutter limits accuracy
    # 2. bit-identical rerun
    b = eng.solve(I1, I2, prob)
    assert torch.equal(a.p, b.p) and torch.equal(a.niter, b.niter)
    # 3. a pair's result does not depend on where it sits in the batch, nor on the batch size
    perm = torch.arange(B - 1, -1, -1, device=eng.device)
    c = eng.solve(I1[perm].contiguous(), I2[perm].contiguous(), prob)
    assert torch.equal(c.p[perm], a.p) and torch.equal(c.niter[perm], a.niter)
    d = eng.solve(I1[:1].contiguous(), I2[:1].contiguous(), eng.problem(1, H, W, C, t, r, nscales=ns, delta=10))
    # (the tile height, hence the fp64 summation order, is chosen from the batch size: equal to rounding, not bitwise)
    np.testing.assert_allclose(d.p[0].cpu().numpy(), a.p[0].cpu().numpy(), rtol=1e-6, atol=1e-10)
    assert torch.equal(d.niter[0], a.niter[0])
