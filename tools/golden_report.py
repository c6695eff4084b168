#!/usr/bin/env python3
"""Parity of the HIP path against tests/golden/reference_configs.json, as a table (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import test_golden as tg
for run in tg.CFG_RUNS:
    n = tg.oc.nparams(run["transform"])
    res = tg._hip_solve(run)
    p = res.p[0, :n].cpu().numpy()
    want = np.array(run["p"])
    nit = [int(v) for v in res.niter[0].cpu().numpy()]
    want_nit = [len(sc["iterations"]) for sc in sorted(run["scales"], key=lambda s: s["scale"])]
    print(f'{run["name"]:<24} rel {np.linalg.norm(p - want) / np.linalg.norm(want):.2e}  corners {tg._corner_disp(p, want, run["transform"], run["H"], run["W"]):.2e} px  niter {nit} ref {want_nit}')
