#!/usr/bin/env python3
"""One-off wider sweep of tests/test_gpu_parity.py::test_seeded_random_configurations_match_oracle (GPU box)."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import test_gpu_parity as tp
from inverse_compositional_algorithm_amd.engine import IcaEngine
eng = IcaEngine(0)
n, seed = int(sys.argv[1]), int(sys.argv[2])
hmax, wmax = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (150, 210)
bad = capped = 0
for case in tp._fuzz_cases(n, seed, hmax, wmax):
    try:
        tp.test_seeded_random_configurations_match_oracle(eng, case)
    except AssertionError as e:
        bad += 1
        print("FAIL", case, str(e)[:300])
print("cases", n, "failures", bad)
