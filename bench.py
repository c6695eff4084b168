#!/usr/bin/env python3
"""Benchmark of the inverse-compositional hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1, one rank per GPU)

One "step" = one full pyramidal solve of a batch of synthetic warped pairs that is already resident in HBM:
pyramid construction, per-level setup, every inverse-compositional iteration (fused warp + steepest descent +
J^T J / J^T r kernel, then the per-pair solve/update kernel) until every pair has converged or hit MAX_ITER.
The metric is BASELINE.json's "IC iterations/sec": executed (pair, scale, iteration) units per second, whole job.
Pairs are independent, so N GPUs run N shards of the same per-GPU batch with no collective (weak scaling).
"""
from __future__ import annotations

import argparse
import contextlib
import threading
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # id: (batch, H, W, C, transform, robust, nscales)   -- BASELINE.json configs[1..4]
    2: dict(batch=1024, H=512, W=512, C=1, transform=4, robust=0, nscales=3,
            name="configs[1]: B=1024 512x512 gray, AFFINITY, L2, 3 scales"),
    3: dict(batch=1024, H=1024, W=1024, C=1, transform=5, robust=0, nscales=4,
            name="configs[2]: B=1024 1024x1024 gray, HOMOGRAPHY, L2, 4 scales"),
    4: dict(batch=1024, H=1024, W=1024, C=1, transform=5, robust=3, nscales=4,
            name="configs[3]: B=1024 1024x1024 gray, HOMOGRAPHY, LORENTZIAN, 4 scales"),
    5: dict(batch=1024, H=1080, W=1920, C=3, transform=5, robust=3, nscales=5,
            name="configs[4]: 1024 pairs/GPU 1920x1080x3, HOMOGRAPHY, LORENTZIAN, 5 scales"),
}
IN_FLIGHT_DEFAULT = 1
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


def measured_traffic_ratio():
    """HBM bytes / algorithmic bytes of the fused iteration kernel, from the committed PMC run
    (profiles/r01_traffic.json: rocprofv3 --pmc FETCH_SIZE, x1024 B, x2 gfx950 correction).  bench.py cannot collect
    PMC counters itself; the per-launch `traffic` it reports is this measured ratio times the launch's algorithmic bytes."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as fh:
            return float(json.load(fh)["fetch_bytes_over_algorithmic_bytes"])
    except Exception:
        return None


def cpu_baseline(cfg, seconds_budget=15.0):
    """The oracle (float64 NumPy restatement with the reference's structure) on a bounded sample of the same
    workload, timed on this host.  Reported next to the GPU number; it is never the thing measured as `value`."""
    import numpy as np
    from inverse_compositional_algorithm_amd import synthetic as syn
    from oracle import ica_oracle as oc
    n = oc.nparams(cfg["transform"])
    iters, pairs, t_total = 0, 0, 0.0
    while t_total < seconds_budget and pairs < 64:
        I1, I2, _ = syn.make_pair(1000 * 2 + pairs, cfg["H"], cfg["W"], cfg["C"], cfg["transform"],
                                  robust_clutter=cfg["robust"] != 0)
        I1, I2 = I1.astype(np.float64), I2.astype(np.float64)
        t0 = time.perf_counter()
        _, _, _, _, nit = oc.ica_pyramidal(I1, I2, np.zeros(n), cfg["transform"], cfg["nscales"], 0.5, 1e-3,
                                           cfg["robust"], 0.0, True, 10)
        t_total += time.perf_counter() - t0
        iters += sum(nit)
        pairs += 1
    return {"value": iters / t_total, "unit": "pair-iterations/s", "cores": 1, "kind": "port",
            "sample": f"{pairs} pairs of the same workload ({iters} pair-iterations, {t_total:.1f} s), oracle/ica_oracle.py "
                      f"(float64 NumPy, materialised steepest-descent tensor + einsum, as the reference)",
            "pairs_per_s": pairs / t_total}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="pairs per GPU (default: the config's)")
    ap.add_argument("--in-flight", type=int, default=IN_FLIGHT_DEFAULT,
                    help="solves kept in flight per GPU (one engine + HIP stream + host thread each): the straggler "
                         "launches of one solve overlap the pyramid and coarse levels of the next")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch

    from inverse_compositional_algorithm_amd import sharding
    info = sharding.rank_info()
    world, rank, local = info.world, info.rank, info.local_rank
    if world > 1:
        torch.cuda.set_device(local)
    sharding.init("nccl", info)  # backend "nccl" is RCCL on ROCm; used for the barrier and 3 scalars only
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)

    from inverse_compositional_algorithm_amd.engine import IcaEngine
    cfg = dict(CONFIGS[args.config])
    if args.batch:
        cfg["batch"] = args.batch
    eng = IcaEngine(local)
    torch.cuda.set_device(local)
    B, H, W, C = cfg["batch"], cfg["H"], cfg["W"], cfg["C"]
    # each rank owns its own shard of pairs, generated on its own device (no host staging, no exchange)
    I1, I2, p_gt = eng.make_batch(sharding.shard_seed(args.config, rank), B, H, W, C, cfg["transform"],
                                  robust_clutter=cfg["robust"] != 0)
    prob = eng.problem(B, H, W, C, cfg["transform"], cfg["robust"], nscales=cfg["nscales"], nu=0.5, tol=1e-3,
                       lambda_=0.0, nanifoutside=True, delta=10)
    eng.prepare(prob)
    # HIP events around every fused-iteration launch, on the stream it runs on.  ~90 event pairs per solve cost 0.3 ms
    # of a 5.4 ms step (measured: BENCH_EVENTS=all vs none), so they bracket every 5th timed step (step 0, 5, ...):
    # the roofline figures are live measurements inside the timed region, taken on that sample of its steps.
    ev_mode = os.environ.get("BENCH_EVENTS", "sample")  # sample | all | none
    def profiled(i):
        return ev_mode == "all" or (ev_mode == "sample" and i % 5 == 0)

    def barrier():
        sharding.barrier(info, eng.device)

    # D solves in flight: D engines (own workspaces, own HIP stream, own host thread -- the host follows each solve to
    # size its launches), all reading the same resident batch.  Steps are handed out one by one, so exactly
    # `steps` solves run between the two barriers however they interleave.
    D = max(1, args.in_flight)
    engines = [eng] + [IcaEngine(local) for _ in range(D - 1)]
    for e in engines[1:]:
        e.prepare(prob)
    streams = [None] if D == 1 else [torch.cuda.Stream(device=eng.device) for _ in range(D)]
    lock = threading.Lock()

    class Gate:
        """Event-bracketed steps run ALONE (the other in-flight solves pause around them), so that the kernel durations
        they measure are those of the kernel, not of kernels sharing the GPU with another solve's."""

        def __init__(self):
            self.cv, self.active, self.excl, self.want_excl = threading.Condition(), 0, False, 0

        def enter(self, exclusive):
            with self.cv:
                if exclusive:
                    self.want_excl += 1
                    while self.excl or self.active > 0:
                        self.cv.wait()
                    self.want_excl -= 1
                    self.excl = True
                else:
                    while self.excl or self.want_excl > 0:
                        self.cv.wait()
                self.active += 1

        def leave(self, exclusive):
            with self.cv:
                self.active -= 1
                if exclusive:
                    self.excl = False
                self.cv.notify_all()

    gate = Gate()
    errors = []
    acc = dict(units=0, dom_ms=0.0, dom_bytes=0.0, dom_n=0, s0_ms=0.0, s0_bytes=0.0, first_ms=0.0, n_prof=0, res=None,
               latency=0.0)

    def run_steps(n_steps, timed):
        counter = iter(range(n_steps))

        def worker(k):
            torch.cuda.set_device(local)
            ctx = torch.cuda.stream(streams[k]) if streams[k] is not None else contextlib.nullcontext()
            with ctx:
                while True:
                    with lock:
                        i = next(counter, None)
                    if i is None:
                        return
                    prof = profiled(i) if timed else True
                    alone = prof and timed and D > 1
                    gate.enter(alone)
                    try:
                        engines[k].set_profiling(prof)
                        t_s = time.perf_counter()
                        r = engines[k].solve(I1, I2, prob, want_stats=True)
                        t_e = time.perf_counter()
                    except BaseException as exc:  # re-raised by the main thread after the join
                        with lock:
                            errors.append(exc)
                        return
                    finally:
                        gate.leave(alone)
                    if not timed:
                        continue
                    with lock:
                        acc["units"] += r.stats["pair_iterations_total"]
                        acc["latency"] += t_e - t_s
                        acc["res"] = r
                        if prof:
                            acc["n_prof"] += 1
                            acc["dom_ms"] += sum(r.stats["iter_kernel_ms"])
                            acc["dom_bytes"] += sum(r.stats["iter_kernel_bytes"])
                            acc["dom_n"] += sum(r.stats["iter_launches"])
                            acc["s0_ms"] += r.stats["iter_kernel_ms"][0]
                            acc["s0_bytes"] += r.stats["iter_kernel_bytes"][0]
                            acc["first_ms"] += sum(r.stats["first_kernel_ms"])

        if D == 1:
            worker(0)
        else:
            th = [threading.Thread(target=worker, args=(k,)) for k in range(D)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        if errors:
            raise errors[0]

    # the dominant kernel = the steady-state iteration kernel (one template instantiation), ALL its launches of the
    # event-bracketed steps at every pyramid scale -- the same set rocprofv3 --stats averages under that kernel's name
    run_steps(max(args.warmup, D if D > 1 else 0), timed=False)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps, timed=True)
    barrier()
    dt = time.perf_counter() - t0
    units, res, n_prof = acc["units"], acc["res"], acc["n_prof"]
    dom_ms, dom_bytes, dom_n = acc["dom_ms"], acc["dom_bytes"], acc["dom_n"]
    s0_ms, s0_bytes, first_ms = acc["s0_ms"], acc["s0_bytes"], acc["first_ms"]

    # solution quality on this shard (not part of the timed region)
    n = eng.lib.ica_nparams(cfg["transform"])
    perr = (res.p[:, :n] - p_gt[:, :n]).abs().amax(dim=1)
    converged = float((res.error < 1e-3).double().mean())

    dt, units_all, pairs_all = sharding.aggregate(info, dt, float(units), float(B * args.steps), eng.device)

    if rank == 0:
        # the PMC pass was taken on the configs[1] kernel; other workloads report no traffic figure
        ratio = measured_traffic_ratio() if args.config == 2 else None
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        tname = {1: "TRANSLATION", 2: "EUCLIDEAN", 3: "SIMILARITY", 4: "AFFINITY", 5: "HOMOGRAPHY"}[cfg["transform"]]
        out = {
            "metric": "IC iterations/sec",
            "value": units_all / dt,
            "unit": "pair-iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": cfg["name"], "pairs_per_gpu": B, "height": H, "width": W, "channels": C,
                       "transform": tname, "robust_type": cfg["robust"], "nscales": cfg["nscales"], "tol": 1e-3,
                       "delta": 10, "parallelism": f"{world} x independent pair shards (no collective)"
                       + (f", {D} solves in flight per GPU" if D > 1 else ""), "in_flight": D},
            "step_latency_ms": acc["latency"] / max(args.steps, 1) * 1e3,
            "image_pairs_per_s": pairs_all / dt,
            "pair_iterations_per_step": units_all / args.steps,
            "roofline": {
                "bound": "hbm",
                "kernel": f"ica_accumulate2_kernel<{cfg['transform']}, {1 if cfg['robust'] else 0}, {C}> "
                          f"({tname}, {'robust: H+b every iteration' if cfg['robust'] else 'L2: b only, iterations >= 1'}), "
                          "all its launches of the timed steps at all scales (grids shrink as pairs converge)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": (ratio * dom_bytes / max(dom_n, 1)) if ratio else None,
                "traffic_source": "profiles/r01_traffic.json (PMC FETCH_SIZE of this kernel at full batch)" if ratio else None,
                "algorithmic_bytes_per_launch": dom_bytes / max(dom_n, 1),
                "avg_launch_ms": dom_ms / max(dom_n, 1),
                "launches": dom_n,
                "event_bracketed_steps": n_prof,
                "kernel_share_of_step": dom_ms / max(n_prof, 1) / (dt / args.steps * 1e3),
                "scale0_only_GBps": (s0_bytes / (s0_ms * 1e-3) / 1e9) if s0_ms > 0 else None,
                "first_iteration_kernel_share_of_step": first_ms / max(n_prof, 1) / (dt / args.steps * 1e3),
            },
            "quality": {"converged_frac": converged, "median_abs_param_err_vs_gt": float(perr.median()),
                        "mean_iters_per_pair": units_all / pairs_all},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out))
    sharding.finalize(info)


if __name__ == "__main__":
    main()
