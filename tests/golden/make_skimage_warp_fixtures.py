#!/opt/conda/bin/python3.9
"""Golden vectors for the bicubic warp from GENUINE scikit-image (the third-party call on the hot path,
reference src/bicubic_interpolation.py:199):

    skimage.transform.warp(img, ProjectiveTransform(matrix=M), order=3, mode="constant", cval=nan,
                           preserve_range=True)

Run in the build container with the second interpreter, the only one that has scikit-image (0.18.3):

    /opt/conda/bin/python3.9 tests/golden/make_skimage_warp_fixtures.py

Writes tests/golden/skimage_warp.npz (inputs and outputs only: small random uint8-valued images, the 3x3 matrices and
skimage's results).  tests/test_oracle_kat.py::test_keys_warp_reproduces_scikit_image checks oracle.keys_warp -- the
restatement every other parity test leans on -- against it, for a translation, a Euclidean map, a similarity, an
affinity and two homographies (the transform classes that have NO stored trace in the reference's notebooks are the
last three).  The rescale() restatement is pinned separately by the notebooks' multi-scale traces (KAT-4/5).
"""
import os

import numpy as np
import skimage
from skimage import transform

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: 3x3 matrix, (H, W, C)
    "translation": (np.array([[1, 0, 2.3], [0, 1, -1.7], [0, 0, 1.0]]), (24, 32, 3)),
    "euclidean": (np.array([[np.cos(0.07), -np.sin(0.07), 1.25], [np.sin(0.07), np.cos(0.07), -0.5], [0, 0, 1.0]]), (24, 32, 3)),
    "similarity": (np.array([[1.03, -0.02, 0.4], [0.02, 1.03, 1.1], [0, 0, 1.0]]), (31, 27, 3)),
    "affinity": (np.array([[1.04, 0.03, -1.2], [-0.025, 0.97, 2.05], [0, 0, 1.0]]), (24, 32, 3)),
    "homography": (np.array([[1.02, 0.015, 1.5], [-0.01, 0.99, -0.75], [4e-4, -3e-4, 1.0]]), (24, 32, 3)),
    "homography_gray_large_shift": (np.array([[0.98, -0.03, 4.25], [0.02, 1.01, -3.5], [-6e-4, 5e-4, 1.0]]), (40, 36, 1)),
    # sampling positions that fall EXACTLY on integers / half-integers (the in/out decision at the border)
    "integer_translation": (np.array([[1, 0, 3.0], [0, 1, -2.0], [0, 0, 1.0]]), (20, 24, 3)),
    "affinity_round_numbers": (np.array([[1.25, 0.0, 3.0], [0.0, 1.5, -2.0], [0, 0, 1.0]]), (24, 32, 3)),
}


def main():
    rng = np.random.default_rng(20261004)
    out = {"skimage_version": np.array(skimage.__version__)}
    for name, (M, shape) in CASES.items():
        img = rng.integers(0, 256, shape).astype(np.float64)
        res = transform.warp(img, transform.ProjectiveTransform(matrix=M), order=3, mode="constant", cval=np.nan,
                             preserve_range=True)
        out[name + "__img"] = img.astype(np.uint8)
        out[name + "__M"] = M
        out[name + "__out"] = res
    np.savez_compressed(os.path.join(HERE, "skimage_warp.npz"), **out)
    print("wrote", os.path.join(HERE, "skimage_warp.npz"), "with", len(CASES), "cases; scikit-image", skimage.__version__)


if __name__ == "__main__":
    main()
