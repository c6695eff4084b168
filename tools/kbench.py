#!/usr/bin/env python3
"""Micro-benchmark of the fused iteration kernel alone (through ica_op_normal_equations_dev).
Run under rocprofv3 --kernel-trace --stats to get the kernel's own duration; the wall time printed here also
includes the min/max pre-pass and the partial-sum gather."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from inverse_compositional_algorithm_amd.engine import IcaEngine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--size", type=int, nargs=2, default=[512, 512])
ap.add_argument("--channels", type=int, default=1)
ap.add_argument("--transform", type=int, default=4)
ap.add_argument("--robust", type=int, default=0)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--scale", type=float, default=0.9, help="p = scale * p_gt")
args = ap.parse_args()
eng = IcaEngine(0)
H, W = args.size
I1, I2, pg = eng.make_batch(7, args.batch, H, W, args.channels, args.transform, robust_clutter=args.robust != 0)
prob = eng.problem(args.batch, H, W, args.channels, args.transform, args.robust)
p = pg * args.scale
for _ in range(2):
    Hm, bv = eng.normal_equations(I1, I2, p, prob, lambda_it=40.0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.reps):
    Hm, bv = eng.normal_equations(I1, I2, p, prob, lambda_it=40.0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.reps
gb = 2.0 * args.batch * H * W * args.channels * 4 / 1e9
print(f"wall {dt*1e3:.3f} ms per call (L2: hessian + rhs kernels; robust: one kernel); algorithmic {gb:.3f} GB per kernel")
print("checksum", float(Hm.sum()), float(bv.sum()))
