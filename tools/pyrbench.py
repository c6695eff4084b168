#!/usr/bin/env python3
"""Micro-benchmark of one pyramid step (ica_op_pyramid_down_dev)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from inverse_compositional_algorithm_amd.engine import IcaEngine
eng = IcaEngine(0)
B, H, W = 1024, 512, 512
x = torch.rand(B, 1, H, W, device=eng.device) * 255
for _ in range(2): y = eng.pyramid_down(x, 0.5)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): y = eng.pyramid_down(x, 0.5)
torch.cuda.synchronize(); print("wall ms per step", (time.perf_counter() - t0) / 5 * 1e3, float(y.sum()))
