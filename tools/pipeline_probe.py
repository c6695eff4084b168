#!/usr/bin/env python3
"""Probe: D full-batch solves in flight (one engine + stream + host thread each) vs one after another.
The straggler launches of one solve then overlap the pyramid / coarse levels of the next."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from inverse_compositional_algorithm_amd.engine import IcaEngine
D = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B, H, W = 1024, 512, 512
gen = IcaEngine(0)
I1, I2, _ = gen.make_batch(2000, B, H, W, 1, 4)
engs = [IcaEngine(0) for _ in range(D)]
streams = [torch.cuda.Stream() for _ in range(D)]
probs = [e.problem(B, H, W, 1, 4, 0, nscales=3) for e in engs]
for e, p in zip(engs, probs): e.prepare(p)
units = [0] * D
def work(i, steps):
    with torch.cuda.stream(streams[i]):
        for _ in range(steps):
            r = engs[i].solve(I1, I2, probs[i], want_stats=True)
            units[i] += r.stats["pair_iterations_total"]
def run(total):
    for i in range(D): units[i] = 0
    th = [threading.Thread(target=work, args=(i, total // D)) for i in range(D)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0
run(2 * D)
total = 12
dt = run(total)
print(f"in_flight={D} ms_per_solve={dt/total*1e3:.3f} iters/s={sum(units)/dt:.0f}")
