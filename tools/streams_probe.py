#!/usr/bin/env python3
"""Probe: does running the batch as S concurrent sub-batches (one engine + stream + host thread each) raise throughput?"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from inverse_compositional_algorithm_amd.engine import IcaEngine
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B, H, W = 1024, 512, 512
gen = IcaEngine(0)
I1, I2, _ = gen.make_batch(2000, B, H, W, 1, 4)
engs = [IcaEngine(0) for _ in range(S)]
streams = [torch.cuda.Stream() for _ in range(S)]
per = B // S
probs = [e.problem(per, H, W, 1, 4, 0, nscales=3) for e in engs]
parts = [(I1[i * per:(i + 1) * per].contiguous(), I2[i * per:(i + 1) * per].contiguous()) for i in range(S)]
for e, p in zip(engs, probs): e.prepare(p)
units = [0] * S
def work(i, steps):
    with torch.cuda.stream(streams[i]):
        for _ in range(steps):
            r = engs[i].solve(parts[i][0], parts[i][1], probs[i], want_stats=True)
            units[i] += r.stats["pair_iterations_total"]
def run(steps):
    for i in range(S): units[i] = 0
    th = [threading.Thread(target=work, args=(i, steps)) for i in range(S)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0
run(2)
dt = run(10)
print(f"streams={S} ms_per_step={dt/10*1e3:.3f} iters/s={sum(units)/dt:.0f}")
