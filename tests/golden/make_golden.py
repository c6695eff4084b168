"""Generate the committed golden vectors (build container only; needs /root/reference).

    python tests/golden/make_golden.py

For each case below a small synthetic pair (inverse_compositional_algorithm_amd.synthetic.make_pair: seeded,
uint8-valued) is solved by the REFERENCE's own drivers, imported from /root/reference/src through
oracle/ref_harness.py (reference code verbatim; the two scikit-image calls the image lacks are the oracle's
restatements, themselves pinned by the notebook traces in kat_traces.json).  What is stored is data only:
the generator arguments, a sha256 of the generated pixels, and the per-iteration |dp| / p / lambda trace that the
reference prints with verbose=True, plus the returned p, error and NaN count of the returned warped image.
"""
import contextlib
import hashlib
import io
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from inverse_compositional_algorithm_amd import synthetic as syn  # noqa: E402
from oracle import ref_harness  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_runs.json")
# (name, transform, robust, nscales, H, W, C, seed, lambda, delta, clutter)
CASES = [
    ("translation_l2_1s", 1, 0, 1, 96, 128, 3, 11, 0.0, 5, False),
    ("euclidean_l2_2s", 2, 0, 2, 128, 160, 3, 12, 0.0, 5, False),
    ("similarity_charbonnier_2s", 3, 4, 2, 128, 160, 3, 13, 0.0, 5, True),
    ("affinity_l2_3s", 4, 0, 3, 256, 256, 3, 14, 0.0, 10, False),
    ("affinity_lorentzian_2s", 4, 3, 2, 128, 160, 3, 15, 0.0, 5, True),
    ("affinity_german_mcclure_fixed_lambda", 4, 2, 1, 96, 128, 3, 16, 30.0, 5, True),
    ("homography_l2_2s", 5, 0, 2, 128, 160, 3, 17, 0.0, 5, False),
    ("homography_l2_3s_256", 5, 0, 3, 256, 256, 3, 18, 0.0, 10, False),
    ("homography_lorentzian_2s", 5, 3, 2, 128, 160, 3, 19, 0.0, 5, True),
    ("homography_charbonnier_1s", 5, 4, 1, 96, 128, 3, 20, 0.0, 5, True),
    ("affinity_l2_gray_as_rgb", 4, 0, 2, 128, 160, 1, 21, 0.0, 5, False),  # gray replicated x3 for the reference
    ("translation_l2_odd_size", 1, 0, 3, 135, 101, 3, 22, 0.0, 4, False),
]
IT_L2 = re.compile(r"^Iteration (\d+): \|Dp\|=(\S+): p=\(([^)]*)\)\s*$")
IT_RB = re.compile(r"^\|Dp\|=(\S+): p=\(([^)]*)\), lambda_=(\S+)\s*$")


def run_case(ref, case):
    name, t, r, ns, H, W, C, seed, lam, delta, clutter = case
    I1, I2, p_gt = syn.make_pair(seed, H, W, C, t, robust_clutter=clutter)
    sha = hashlib.sha256(I1.tobytes() + I2.tobytes()).hexdigest()
    J1, J2 = I1.astype(np.float64), I2.astype(np.float64)
    if C == 1:
        J1, J2 = np.repeat(J1, 3, axis=2), np.repeat(J2, 3, axis=2)
    n = ref.tr.TransformType(t).nparams()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        p, err, DI, Iw = ref.ica.pyramidal_inverse_compositional_algorithm(
            I1=J1, I2=J2, p=np.zeros(n), transform_type=ref.tr.TransformType(t), nscales=ns, nu=0.5, TOL=1e-3,
            robust_type=ref.io.RobustErrorFunctionType(r), lambda_=lam, nanifoutside=True, delta=delta, verbose=True)
    scales, cur = [], None
    for line in buf.getvalue().splitlines():
        m = re.match(r"^Scale: (\d+)", line)
        if m:
            cur = {"scale": int(m.group(1)), "iterations": []}
            scales.append(cur)
            continue
        m = IT_L2.match(line)
        if m:
            cur["iterations"].append({"dp_norm": float(m.group(2)), "p": [float(v) for v in m.group(3).split()]})
            continue
        m = IT_RB.match(line)
        if m:
            cur["iterations"].append({"dp_norm": float(m.group(1)), "p": [float(v) for v in m.group(2).split()],
                                      "lambda": float(m.group(3))})
    return {"name": name, "transform": t, "robust": r, "nscales": ns, "H": H, "W": W, "C": C, "seed": seed,
            "lambda": lam, "delta": delta, "clutter": clutter, "tol": 1e-3, "nu": 0.5, "inputs_sha256": sha,
            "p_gt": p_gt.tolist(), "p": np.asarray(p).tolist(), "error": float(err),
            "iw_nan_count": int(np.isnan(Iw[..., 0]).sum()),
            "iw_checksum": float(np.nansum(Iw[..., 0])), "scales": scales}


def main():
    ref = ref_harness.load()
    runs = [run_case(ref, c) for c in CASES]
    json.dump({"generator": "tests/golden/make_golden.py", "source": "reference drivers via oracle/ref_harness.py",
               "runs": runs}, open(OUT, "w"), indent=1)
    for r in runs:
        print(r["name"], [(s["scale"], len(s["iterations"])) for s in r["scales"]], "err", r["error"])


if __name__ == "__main__":
    main()
