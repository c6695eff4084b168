#!/usr/bin/env python3
"""CPU baseline (A) of BASELINE.md section 4: the REFERENCE's own loops timed in the build container, next to the oracle
(the restatement bench.py times on the GPU box as `cpu_baseline`) on identical inputs -- so that the oracle's rate can
stand in for "the repo's NumPy CPU path".  Needs /root/reference (oracle/ref_harness.py); never runs on the GPU box.

    python tools/cpu_baseline_reference.py > profiles/r02_cpu_baseline_A.json

config (1): full single-scale solve of one 256x256 pair (gray replicated to 3 channels: the reference accepts nothing
else), AFFINITY, L2.  configs (2)-(4): one pair each at full size; per pyramid scale the level setup (gradient, Jacobian,
steepest-descent images, L2 Hessian: ref :81-103 / :200-218) and K = 2 passes of the loop body (warp, residual, rho',
J^T r, J^T J, inverse, solve, update: ref :109-131 / :225-259), exactly what bench.py's cpu_baseline times on the oracle.
The warp inside both is the NumPy restatement of skimage.transform.warp (scikit-image cannot be imported under this
interpreter); upstream's own figure for skimage's Cython warp is 32.8 ms at 584x388x3 (test/time_profiling.ipynb:306-307).
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from inverse_compositional_algorithm_amd import synthetic as syn  # noqa: E402
from oracle import ica_oracle as oc, ref_harness  # noqa: E402

import bench  # noqa: E402  (CONFIGS, host_info)


def level_times(ref, L1, L2, p, t, rt, K=2):
    """(setup seconds, seconds per loop-body pass) with the reference's functions and with the oracle's."""
    ny, nx = L1.shape[:2]
    n = oc.nparams(t)
    tt = ref.tr.TransformType(t)
    rob = ref.io.RobustErrorFunctionType(rt)
    out = {}
    # ---- reference ----
    t0 = time.perf_counter()
    Ix, Iy = np.zeros_like(L1), np.zeros_like(L1)
    Ix[:, 1:-1] = 0.5 * (L1[:, 2:] - L1[:, :-2])          # ref :81-82 (inline in the reference's drivers)
    Iy[1:-1, :] = 0.5 * (L1[2:, :] - L1[:-2, :])
    for a in (Ix, Iy):                                     # ref :85-93 NaN frame, delta = 10
        a[:10] = np.nan; a[-10:] = np.nan; a[:, :10] = np.nan; a[:, -10:] = np.nan
    J = ref.de.jacobian(tt, nx, ny)
    DIJ = ref.io.steepest_descent_images(Ix, Iy, J, n)
    if rt == 0:
        H_1 = ref.de.inverse_hessian(ref.de.hessian(DIJ), n)
    setup_ref = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(K):
        Iw = ref.bi.bicubic_interpolation_skimage(L2, p, tt, True, 10)
        DI = Iw - L1
        if rt != 0:
            rho = ref.io.robust_error_function(DI, 40.0, rob)
            b = ref.io.independent_vector_robust(DIJ, DI, rho, n)
            H_1 = ref.de.inverse_hessian(ref.de.hessian_robust(DIJ, rho, n), n)
        else:
            b = ref.io.independent_vector(DIJ, DI, n)
        err, dp = ref.io.parametric_solve(H_1, b, n)
        ref.tr.update_transform(p.copy(), dp, tt)
    iter_ref = (time.perf_counter() - t0) / K
    # ---- oracle ----
    t0 = time.perf_counter()
    Ix, Iy = oc.gradient(L1, True, 10)
    J = oc.jacobian(t, nx, ny)
    DIJ = oc.steepest_descent_images(Ix, Iy, J, n)
    if rt == 0:
        H_1 = oc.inverse_hessian(oc.hessian(DIJ))
    setup_or = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(K):
        Iw = oc.warp_params(L2, p, t)
        DI = Iw - L1
        if rt != 0:
            rho = oc.robust_error_function(DI, 40.0, rt)
            b = oc.independent_vector(DIJ, DI, rho)
            H_1 = oc.inverse_hessian(oc.hessian(DIJ, rho))
        else:
            b = oc.independent_vector(DIJ, DI)
        _, dp = oc.parametric_solve(H_1, b)
        oc.update_transform(p.copy(), dp, t)
    iter_or = (time.perf_counter() - t0) / K
    return dict(setup_reference_s=setup_ref, iteration_reference_s=iter_ref, setup_oracle_s=setup_or, iteration_oracle_s=iter_or)


def main():
    ref = ref_harness.load()
    out = {"host": bench.host_info(), "note": "build container, 8 cores; NumPy element-wise / einsum code is single-threaded, "
           "OpenBLAS threads only serve the 8x8 inverse", "configs": {}}
    # config (1): the reference's own driver, full solve
    I1, I2, _ = syn.make_pair(1000, 256, 256, 1, 4)
    J1, J2 = np.repeat(I1.astype(np.float64), 3, axis=2), np.repeat(I2.astype(np.float64), 3, axis=2)
    t0 = time.perf_counter()
    p, err, DI, Iw = ref.ica.inverse_compositional_algorithm(J1, J2, np.zeros(6), ref.tr.TransformType(4), 1e-3, True, 10, False)
    t_ref = time.perf_counter() - t0
    t0 = time.perf_counter()
    po, _, _, _, nit = oc.ica_single_scale(J1, J2, np.zeros(6), 4, 1e-3, 0, 0.0, True, 10, need_rgb=True)
    t_or = time.perf_counter() - t0
    out["configs"]["configs[0] 256x256x3 AFFINITY L2 1 scale, full solve"] = dict(
        iterations=nit, reference_s=t_ref, oracle_s=t_or, reference_iterations_per_s=nit / t_ref,
        oracle_iterations_per_s=nit / t_or, max_abs_param_diff=float(np.abs(p - po).max()))
    for cid in (2, 3, 4):
        cfg = bench.CONFIGS[cid]
        t, rt, ns = cfg["transform"], cfg["robust"], cfg["nscales"]
        I1, I2, g = syn.make_pair(1000 * 2, cfg["H"], cfg["W"], cfg["C"], t, robust_clutter=rt != 0)
        # the reference only takes 3 channels: replicate the gray pair (what the goldens did too)
        L1, L2 = [np.repeat(I1.astype(np.float64), 3, axis=2)], [np.repeat(I2.astype(np.float64), 3, axis=2)]
        for s in range(1, ns):
            L1.append(oc.rescale_restated(L1[-1], 0.5)); L2.append(oc.rescale_restated(L2[-1], 0.5))
        per = []
        for s in range(ns - 1, -1, -1):
            p = 0.9 * np.asarray(g, dtype=np.float64)
            if t == 5:
                p[[2, 5]] *= 0.5 ** s; p[[6, 7]] /= 0.5 ** s
            else:
                p[:2] *= 0.5 ** s
            r = level_times(ref, L1[s], L2[s], p, t, rt)
            r["scale"] = s; r["size"] = list(L1[s].shape)
            per.append(r)
            print(cfg["name"], r, file=sys.stderr, flush=True)
        out["configs"][cfg["name"] + " (gray replicated to 3 channels)"] = per
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
