#!/usr/bin/env python3
"""Pyramid-step microbenchmark: eng.pyramid_down on B planes of HxW, a few repetitions (for rocprofv3)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from inverse_compositional_algorithm_amd.engine import IcaEngine
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, nargs=2, default=[1024, 1024])
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--nu", type=float, default=0.5)
a = ap.parse_args()
eng = IcaEngine(0)
g = torch.Generator(device="cpu").manual_seed(1)
x = (torch.rand(a.batch, 1, a.size[0], a.size[1], generator=g) * 255).round().to(eng.device)
y = eng.pyramid_down(x, a.nu)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(a.reps):
    y = eng.pyramid_down(x, a.nu)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / a.reps
gb = x.numel() * 4 / 1e9 + y.numel() * 4 / 1e9
print("wall %.3f ms per step; compulsory traffic %.3f GB -> %.2f TB/s" % (dt * 1e3, gb, gb / dt / 1e3))
