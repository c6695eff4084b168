#!/usr/bin/env python3
"""Where a wave of the fused iteration kernel spends its cycles, the shader clock it runs at, and how many waves a CU
really holds (diagnostic build: tools/mkvariant.sh stamp -DICA_STAMP).

usage (GPU box): ICA_LIB=tools/abl/lib_stamp.so python3 tools/stamp_phases.py [--batch ... like kbench]
Every wave of the last launch leaves a record (ica_iterate.hpp): cycles (s_memtime) per phase, its lifetime on the
constant 100 MHz clock (s_memrealtime), and the hardware id of the CU / SIMD it ran on."""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from inverse_compositional_algorithm_amd import _lib  # noqa: E402
from inverse_compositional_algorithm_amd.engine import IcaEngine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--size", type=int, nargs=2, default=[1024, 1024])
ap.add_argument("--channels", type=int, default=1)
ap.add_argument("--transform", type=int, default=5)
ap.add_argument("--robust", type=int, default=3)
ap.add_argument("--scale", type=float, default=0.9)
args = ap.parse_args()
eng = IcaEngine(0)
H, W = args.size
I1, I2, pg = eng.make_batch(7, args.batch, H, W, args.channels, args.transform, robust_clutter=args.robust != 0)
prob = eng.problem(args.batch, H, W, args.channels, args.transform, args.robust)
p = pg * args.scale
lib = _lib.load()
fn = getattr(lib, f"ica_debug_stamps_tt{args.transform}")
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
nrec = fn(None, 0)
rec = np.zeros((nrec, 16), dtype=np.uint64)
eng.normal_equations(I1, I2, p, prob, lambda_it=40.0)
torch.cuda.synchronize()
fn(rec.ctypes.data, nrec)  # reset
t0 = time.perf_counter()
eng.normal_equations(I1, I2, p, prob, lambda_it=40.0)   # the records hold the last launch
torch.cuda.synchronize()
print(f"wall of the stamped call {1e3 * (time.perf_counter() - t0):.2f} ms (with the min/max pass and the gather)")
assert fn(rec.ctypes.data, nrec) == 0
rec = rec[rec[:, 6] != 0].astype(np.int64)
waves = len(rec)
v = rec[:, :8].sum(axis=0)
print(f"shader clock inside the kernel: {v[6] / (v[7] * 10.0):.3f} GHz (s_memtime / s_memrealtime over every wave's lifetime)")
names = ["barrier 1 (previous chunk done)", "issue LDS-DMA", "barrier 2 (copies landed)", "band", "prologue", "fold",
         "whole kernel"]
print(f"{waves} waves in the last launch")
for i, n in enumerate(names):
    print(f"  {n:34s} {v[i] / waves:12.0f} cycles per wave  {100.0 * v[i] / max(v[6], 1):5.1f} %")
# residency: waves per CU over the launch, from the absolute begin / end times
begin, end = rec[:, 8], rec[:, 9]
span = (end.max() - begin.min()) / 100.0  # us
hw, xcc = rec[:, 10], rec[:, 11] & 0xF
# gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
ncu = len(np.unique(cu_key))
busy = (end - begin).sum() / 100.0  # wave-us
print(f"launch span {span:.1f} us on {ncu} CUs seen; mean resident waves per CU {busy / span / ncu:.2f} "
      f"(per SIMD {busy / span / ncu / 4:.2f}); wave lifetime {np.mean(end - begin) / 100.0:.1f} us")
# middle half of the launch only (no ramp, no tail)
lo, hi = begin.min() + (end.max() - begin.min()) // 4, begin.min() + 3 * (end.max() - begin.min()) // 4
ov = (np.minimum(end, hi) - np.maximum(begin, lo)).clip(min=0).sum() / float(hi - lo)
print(f"middle half of the launch: {ov / ncu:.2f} resident waves per CU ({ov / ncu / 4:.2f} per SIMD)")
per_cu = np.bincount(np.unique(cu_key, return_inverse=True)[1], weights=(np.minimum(end, hi) - np.maximum(begin, lo)).clip(min=0)) / float(hi - lo)
print(f"  per CU: min {per_cu.min():.1f}  median {np.median(per_cu):.1f}  max {per_cu.max():.1f} waves;  SIMD split of waves: "
      + " ".join(f"{(simd == k).mean():.2f}" for k in range(4)))
