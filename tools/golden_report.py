#!/usr/bin/env python3
"""Parity of the HIP path against the committed reference runs, as a table (GPU box):
tests/golden/reference_configs.json (BASELINE config sizes) and tests/golden/reference_runs.json (small runs).
Per run: end-point error (relative, corner displacement), iteration counts, and the worst deviation along the whole
trajectory (corner displacement of every iterate, relative error of every |dp|) -- the figures the tolerances of
tests/test_golden.py are set from."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import test_golden as tg

worst = dict(rel=0.0, corners=0.0, traj=0.0, dp=0.0, dp_abs=0.0)
for group, runs in (("config sizes", tg.CFG_RUNS), ("small runs", tg.RUNS)):
    print(f"---- {group} ----")
    for run in runs:
        t, n = run["transform"], tg.oc.nparams(run["transform"])
        res = tg._hip_solve(run, want_trace=True)
        p = res.p[0, :n].cpu().numpy()
        tr = res.trace[0].cpu().numpy()
        want = np.array(run["p"])
        nit = [int(v) for v in res.niter[0].cpu().numpy()]
        want_nit = [len(sc["iterations"]) for sc in sorted(run["scales"], key=lambda s: s["scale"])]
        rel = np.linalg.norm(p - want) / np.linalg.norm(want)
        cd = tg._corner_disp(p, want, t, run["H"], run["W"])
        traj = dpr = dpa = 0.0
        for sc in run["scales"]:
            s, h, w = sc["scale"], run["H"], run["W"]
            for _ in range(s):
                w, h = tg.oc.zoom_size(w, h, run["nu"])
            for k, it in enumerate(sc["iterations"][:nit[s]]):
                traj = max(traj, tg._corner_disp(tr[s, k, 1:1 + n], np.array(it["p"]), t, h, w))
                d = abs(tr[s, k, 0] - it["dp_norm"])
                dpa = max(dpa, d)
                dpr = max(dpr, d / max(it["dp_norm"], 1e-3))   # relative to max(|dp|, TOL)
        ok = "" if nit == want_nit else "   <-- ITERATION COUNTS DIFFER"
        print(f'{run["name"]:<34} rel {rel:.2e}  corners {cd:.2e} px  trajectory {traj:.2e} px  |dp| rel {dpr:.2e} abs {dpa:.2e}  '
              f'niter {nit} ref {want_nit}{ok}')
        worst = dict(rel=max(worst["rel"], rel), corners=max(worst["corners"], cd), traj=max(worst["traj"], traj),
                     dp=max(worst["dp"], dpr), dp_abs=max(worst["dp_abs"], dpa))
print("worst:", " ".join(f"{k} {v:.2e}" for k, v in worst.items()))
