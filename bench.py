#!/usr/bin/env python3
"""Benchmark of the inverse-compositional hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no torchrun environment, bench.py starts the N ranks itself (a child `python -m torch.distributed.run
--nproc-per-node N bench.py ...`, started before this process touches the GPU) and relays rank 0's JSON line; under
torchrun (RANK / WORLD_SIZE set) it is one of the ranks.  The ranks count themselves with an all-reduce and the run
FAILS if that count differs from --gpus: an N-GPU number can never silently be a 1-GPU number.

One "step" = one full pyramidal solve of a batch of synthetic warped pairs that is already resident in HBM:
pyramid construction, per-level setup, every inverse-compositional iteration (fused warp + steepest descent +
J^T J / J^T r kernel, then the per-pair solve/update kernel) until every pair has converged or hit MAX_ITER.
The metric is BASELINE.json's "IC iterations/sec": executed (pair, scale, iteration) units per second, whole job.
Pairs are independent, so N GPUs run N shards of the same per-GPU batch with no collective (weak scaling).

The headline workload is BASELINE.json configs[3] (1024 pairs of 1024x1024, homography, Lorentzian, 4 scales): the
configuration the 70 %-of-roofline target is quoted on, whose iteration kernel is the fused
warp + steepest-descent + Hessian pass with the robust weight, `ica_iterate_kernel<5, 1, 1, 3>`.  At N = 1 the same
JSON line carries `per_config`: shorter runs of configs[1], configs[2] and the per-GPU shard of configs[4].
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import socket
import subprocess
import sys
import threading
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
DEFAULT_CONFIG = 4                 # BASELINE configs[3]
# (config id, timed steps, image storage, solves in flight) of the per_config entries; the last one is the headline
# workload again with its level-0 images stored as uint8, reported under both byte conventions.
# configs[1]'s solve is short (5 ms: 3 small levels, the last 14 iterations of scale 0 run one straggling pair): three
# solves of the resident batch in flight (three contexts, streams and host threads) overlap those tails, 4.6 instead
# of 5.0 ms per solve; the larger workloads gain 3-5 % from two and are reported with one.
SIDE_CONFIGS = ((2, 6, "f32", 3), (3, 3, "f32", 1), (5, 2, "f32", 1), (4, 3, "u8", 1))   # (.., solves in flight)
SIDE_DEADLINE_S = 330.0            # no side config is started later than this after process start
IN_FLIGHT_DEFAULT = 1
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
TNAME = {1: "TRANSLATION", 2: "EUCLIDEAN", 3: "SIMILARITY", 4: "AFFINITY", 5: "HOMOGRAPHY"}
T_START = time.perf_counter()


def kernel_key(cfg):
    """Template arguments <transform, mode, channels> of the steady-state iteration kernel of a workload."""
    return f"<{cfg['transform']}, {1 if cfg['robust'] else 0}, {cfg['C']}, {cfg['robust']}>"


TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r03_traffic.json")


def library_fingerprint():
    """Source fingerprint of the library this process loaded (the build stamp's first line); None for an experiment
    build selected with ICA_LIB, which has no stamp."""
    if "ICA_LIB" in os.environ:
        return None
    try:
        from inverse_compositional_algorithm_amd import build
        return build.read_stamp()[0] or None
    except Exception:
        return None


def measured_traffic(cfg, storage="f32"):
    """HBM bytes / algorithmic bytes of one full-batch launch of this workload's iteration kernel, from the committed
    PMC passes (profiles/r03_traffic.json, built by tools/pmc_traffic.sh from rocprofv3 --pmc FETCH_SIZE CSVs; x1024 B,
    x2 gfx950 correction).  bench.py cannot collect PMC counters itself: the per-launch `traffic` it reports is this
    measured ratio times the launch's algorithmic bytes -- and ONLY when the file was measured on the very library that is
    loaded now (the file records the library's source fingerprint); after any change to the kernels it is null until
    the counters have been collected again."""
    try:
        with open(TRAFFIC_FILE) as fh:
            doc = json.load(fh)
        if not doc.get("library_fingerprint") or doc["library_fingerprint"] != library_fingerprint():
            return None, f"stale: {os.path.basename(TRAFFIC_FILE)} was measured on another build of the library"
        # (the uint8-storage instantiation is a kernel of its own: its counters, or nothing)
        e = doc["kernels"][kernel_key(cfg) + (" u8" if storage == "u8" else "")]
        return float(e["fetch_bytes_over_algorithmic_bytes"]), e.get("source")
    except Exception:
        return None, None


def host_info():
    info = {"nproc": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)),
            "OMP_NUM_THREADS": os.environ.get("OMP_NUM_THREADS"), "MKL_NUM_THREADS": os.environ.get("MKL_NUM_THREADS")}
    try:
        from threadpoolctl import threadpool_info
        info["blas"] = [{k: d.get(k) for k in ("internal_api", "num_threads")} for d in threadpool_info()]
    except Exception:
        info["blas"] = None
    return info


def cpu_baseline(cfg, mean_iters_per_scale, budget_s=25.0):
    """The oracle (float64 NumPy restatement with the reference's structure) timed on this host on a BOUNDED sample of
    the same workload: ONE pair; at every pyramid scale the per-level setup of `oracle.ica_single_scale` (gradient,
    Jacobian, steepest-descent tensor, L2 Hessian: ref :81-103 / :200-218) and K passes of its loop body (warp,
    residual, rho', J^T r, J^T J, inverse, solve, update: ref :109-131 / :225-259) are timed, and both are
    extrapolated to the mean iteration counts per scale the GPU run executed.  It is reported next to the GPU
    number; it is never the thing measured as `value`."""
    import numpy as np
    from inverse_compositional_algorithm_amd import synthetic as syn
    from oracle import ica_oracle as oc
    t, rt, ns = cfg["transform"], cfg["robust"], cfg["nscales"]
    n = oc.nparams(t)
    I1, I2, p_gt = syn.make_pair(1000 * 2, cfg["H"], cfg["W"], cfg["C"], t, robust_clutter=rt != 0)
    L1, L2 = [I1.astype(np.float64)], [I2.astype(np.float64)]
    t0 = time.perf_counter()
    for s in range(1, ns):
        L1.append(oc.rescale_restated(L1[-1], 0.5))
        L2.append(oc.rescale_restated(L2[-1], 0.5))
    t_pyr = time.perf_counter() - t0
    setup, per_iter, spent, passes = [0.0] * ns, [0.0] * ns, t_pyr, []
    per_scale_s = 0.8 * budget_s / ns   # time budget of the loop-body passes of one scale
    for s in range(ns - 1, -1, -1):   # coarse to fine; the fine scales are the expensive ones
        if spent > budget_s:  # out of budget: extrapolate the remaining (finer) scales by pixel count
            per_iter[s], setup[s] = per_iter[s + 1] * 4.0, setup[s + 1] * 4.0
            continue
        ny, nx = L1[s].shape[:2]
        p = 0.9 * np.asarray(p_gt, dtype=np.float64)  # a mid-trajectory iterate: the ground truth at this level, scaled
        if t == oc.HOMOGRAPHY:
            p[[2, 5]] *= 0.5 ** s
            p[[6, 7]] /= 0.5 ** s
        else:
            p[:2] *= 0.5 ** s
        t0 = time.perf_counter()
        Ix, Iy = oc.gradient(L1[s], True, 10)
        J = oc.jacobian(t, nx, ny)
        DIJ = oc.steepest_descent_images(Ix, Iy, J, n)
        if rt == oc.QUADRATIC:
            H_1 = oc.inverse_hessian(oc.hessian(DIJ))
        setup[s] = time.perf_counter() - t0

        def body():
            Iw = oc.warp_params(L2[s], p, t)
            DI = Iw - L1[s]
            if rt != oc.QUADRATIC:
                rho = oc.robust_error_function(DI, 40.0, rt)
                b = oc.independent_vector(DIJ, DI, rho)
                Hi = oc.inverse_hessian(oc.hessian(DIJ, rho))
            else:
                b, Hi = oc.independent_vector(DIJ, DI), H_1
            _, dp = oc.parametric_solve(Hi, b)
            oc.update_transform(p.copy(), dp, t)

        t0 = time.perf_counter()
        body()
        t1 = time.perf_counter() - t0
        K = max(2, min(400, int(per_scale_s / max(t1, 1e-6))))  # enough passes for ~per_scale_s seconds at this scale
        t0 = time.perf_counter()
        for _ in range(K):
            body()
        per_iter[s] = (time.perf_counter() - t0) / K
        passes.append(K)
        spent += setup[s] + (K + 1) * per_iter[s]
        del DIJ, J, Ix, Iy
    iters = sum(mean_iters_per_scale)
    t_pair = t_pyr + sum(setup[s] + mean_iters_per_scale[s] * per_iter[s] for s in range(ns))
    return {"value": iters / t_pair, "unit": "pair-iterations/s", "cores": 1, "kind": "port",
            "sample": f"1 pair of the same workload; per pyramid scale the level setup and {passes} (coarse to fine) passes of the loop body of "
                      f"oracle.ica_single_scale ({spent:.1f} s of CPU work), extrapolated to the GPU run's mean iterations "
                      f"per scale {[round(v, 2) for v in mean_iters_per_scale]}; oracle/ica_oracle.py = float64 NumPy, "
                      f"materialised steepest-descent tensor + einsum, as the reference",
            "pairs_per_s": 1.0 / t_pair, "seconds_per_iteration_by_scale": per_iter, "setup_seconds_by_scale": setup,
            "pyramid_seconds": t_pyr, "host": host_info()}


# ---- launching the ranks -----------------------------------------------------------------------------------

def spawn_ranks(n, argv):
    """Start the N ranks as a CHILD process (nothing in this process has touched the GPU yet; no exec of a
    GPU-initialised process) and exit with its code.  The child's rank 0 prints the JSON line on our stdout."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


class StubEngine:
    """CPU stand-in for the engine (BENCH_STUB=1): lets the launcher, the rank count check and the aggregation be
    tested with gloo on a machine without a GPU.  It computes nothing and is never used for a reported number."""

    def __init__(self):
        import torch
        self.device = torch.device("cpu")


def run_stub(args, info, sharding):
    import torch
    eng = StubEngine()
    sharding.barrier(info, eng.device)
    t0 = time.perf_counter()
    time.sleep(0.01 * args.steps)
    units = 100.0 * args.steps * (info.rank + 1)
    sharding.barrier(info, eng.device)
    dt = time.perf_counter() - t0
    dt, units_all, pairs_all = sharding.aggregate(info, dt, units, 8.0 * args.steps, eng.device)
    return dict(dt=dt, units_all=units_all, pairs_all=pairs_all)


# ---- one measured workload ---------------------------------------------------------------------------------

def measure(args, info, sharding, cfg_id, steps, warmup, D, batch=0, storage="f32"):
    """Generate this rank's shard of `cfg_id`, run `warmup` untimed and `steps` timed solves between barriers.
    Returns this job's aggregate numbers (all ranks) and rank-local quality figures."""
    import torch
    from inverse_compositional_algorithm_amd.engine import IcaEngine
    local = info.local_rank
    cfg = dict(CONFIGS[cfg_id])
    if batch:
        cfg["batch"] = batch
    torch.cuda.set_device(local)
    B, H, W, C = cfg["batch"], cfg["H"], cfg["W"], cfg["C"]
    # the shard must fit before anything is allocated (configs[4]: ~68 GB per GPU of 288)
    free_b, total_b = torch.cuda.mem_get_info(local)
    need_b = sharding.shard_device_bytes(B, H, W, C, cfg["nscales"], 0.5, 1 if storage == "u8" else 4)
    sharding.check_memory_budget(need_b, free_b, total_b, f"rank {info.rank}: the shard of {cfg['name']}")
    eng = IcaEngine(local)
    # each rank owns its own shard of pairs, generated on its own device (no host staging, no exchange)
    t_gen = time.perf_counter()
    I1, I2, p_gt = eng.make_batch(sharding.shard_seed(cfg_id, info.rank), B, H, W, C, cfg["transform"],
                                  robust_clutter=cfg["robust"] != 0,
                                  dtype=torch.uint8 if storage == "u8" else torch.float32)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    prob = eng.problem(B, H, W, C, cfg["transform"], cfg["robust"], nscales=cfg["nscales"], nu=0.5, tol=1e-3,
                       lambda_=0.0, nanifoutside=True, delta=10)
    eng.prepare(prob)
    # HIP events around every fused-iteration launch, on the stream it runs on.  Event pairs cost GPU time (~3.4 us
    # each), so they bracket every 5th timed step (step 0, 5, ...): the roofline figures are live measurements inside
    # the timed region, taken on that sample of its steps.
    ev_mode = os.environ.get("BENCH_EVENTS", "sample")  # sample | all | none

    def profiled(i):
        return ev_mode == "all" or (ev_mode == "sample" and i % 5 == 0)

    # D solves in flight: D engines (own workspaces, own HIP stream, own host thread -- the host follows each solve to
    # size its launches), all reading the same resident batch.  Steps are handed out one by one, so exactly
    # `steps` solves run between the two barriers however they interleave.
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
    ns = cfg["nscales"]
    acc = dict(units=0, per_scale=[0] * ns, dom_ms=0.0, dom_bytes=0.0, dom_n=0, s0_ms=0.0, s0_bytes=0.0, s0_n=0,
               first_ms=0.0, n_prof=0, res=None, latency=0.0, lvl_ms=[0.0] * ns, lvl_bytes=[0.0] * ns, lvl_n=[0] * ns,
               lvl_units=[0] * ns, lvl_first_ms=[0.0] * ns, lvl_first_n=[0] * ns, lvl_wh=[(0, 0)] * ns)

    clock = dict(samples=[])   # shader clock of each event-bracketed step (measured inside the iteration kernel)

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
                        for s in range(ns):
                            acc["per_scale"][s] += r.stats["pair_iterations"][s]
                        acc["latency"] += t_e - t_s
                        acc["res"] = r
                        if prof:
                            acc["n_prof"] += 1
                            acc["dom_ms"] += sum(r.stats["iter_kernel_ms"])
                            acc["dom_bytes"] += sum(r.stats["iter_kernel_bytes"])
                            acc["dom_n"] += sum(r.stats["iter_launches"])
                            acc["s0_ms"] += r.stats["iter_kernel_ms"][0]
                            acc["s0_bytes"] += r.stats["iter_kernel_bytes"][0]
                            acc["s0_n"] += r.stats["iter_launches"][0]
                            acc["first_ms"] += sum(r.stats["first_kernel_ms"])
                            if r.stats.get("shader_clock_ghz", 0.0) > 0.0:
                                clock["samples"].append(r.stats["shader_clock_ghz"])
                            for s in range(ns):
                                acc["lvl_ms"][s] += r.stats["iter_kernel_ms"][s]
                                acc["lvl_bytes"][s] += r.stats["iter_kernel_bytes"][s]
                                acc["lvl_n"][s] += r.stats["iter_launches"][s]
                                acc["lvl_units"][s] += r.stats["pair_iterations"][s]
                                acc["lvl_first_ms"][s] += r.stats["first_kernel_ms"][s]
                                acc["lvl_first_n"][s] += r.stats["first_launches"][s]
                                acc["lvl_wh"][s] = tuple(r.stats["level_size"][s])

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
    run_steps(max(warmup, D if D > 1 else 0), timed=False)
    # The shader clock the chip sustains under this load is measured INSIDE the iteration kernel of the event-bracketed
    # steps (ica_stats.shader_clock_ghz: cycles / 100 MHz ticks of one workgroup's lifetime in the middle of every
    # launch).  (Until round 3 a probe wave on a second stream measured it beside the timed region; when the two streams
    # happened to share a hardware queue the probe's 3 ms in every 15 serialised with the solve -- seen once as a step
    # 20 % longer at the idle clock, 2.41 GHz -- so nothing but the solves runs in the timed region now.)
    sharding.barrier(info, eng.device)
    t0 = time.perf_counter()
    run_steps(steps, timed=True)
    sharding.barrier(info, eng.device)
    dt = time.perf_counter() - t0
    res = acc["res"]

    # solution quality on this shard (not part of the timed region)
    n = eng.lib.ica_nparams(cfg["transform"])
    perr = (res.p[:, :n] - p_gt[:, :n]).abs().amax(dim=1)
    converged = float((res.error < 1e-3).double().mean())
    hit_cap = float((res.niter >= 30).any(dim=1).double().mean())
    dt, units_all, pairs_all = sharding.aggregate(info, dt, float(acc["units"]), float(B * steps), eng.device)

    dom_ms, dom_bytes, dom_n, n_prof = acc["dom_ms"], acc["dom_bytes"], acc["dom_n"], acc["n_prof"]
    ms_per_step = dt / steps * 1e3
    # iterations/s per pyramid level -- executed pair-iterations of a level over the time its iteration
    # kernel launches took (HIP events, event-bracketed steps), next to the roofline fraction of those launches.  The
    # peak-rate column is what 8 TB/s would allow at that level's size (954 k/s at 1024^2, 3.81 M/s at 512^2, ...).
    per_level = []
    for s in range(ns):
        w, h = acc["lvl_wh"][s]
        ms, by, un = acc["lvl_ms"][s] + acc["lvl_first_ms"][s], acc["lvl_bytes"][s], acc["lvl_units"][s]
        per_level.append({
            "level": s, "width": w, "height": h,
            "mean_iterations_per_pair": acc["per_scale"][s] / float(B * steps),
            "pair_iterations_per_s": (un / (ms * 1e-3)) if ms > 0 else None,
            "peak_pair_iterations_per_s": HBM_PEAK_GBS * 1e9 / (2.0 * w * h * C * 4) if w and h else None,
            "frac": (by / (acc["lvl_ms"][s] * 1e-3) / 1e9 / HBM_PEAK_GBS) if acc["lvl_ms"][s] > 0 else None,
            "iteration_kernel_ms_per_step": ms / max(n_prof, 1),
            "launches_per_step": (acc["lvl_n"][s] + acc["lvl_first_n"][s]) / max(n_prof, 1)})
    cs = sorted(clock["samples"])
    shader_clock = ({"mean_ghz": sum(cs) / len(cs), "min_ghz": cs[0], "max_ghz": cs[-1], "samples": len(cs),
                     "method": "inside the iteration kernel: s_memtime / s_memrealtime over one workgroup's lifetime in the middle of "
                               "every launch, one sample per event-bracketed step of the timed region"}
                    if cs else None)
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    ratio, src = measured_traffic(cfg, storage)
    out = dict(
        cfg=cfg, dt=dt, units_all=units_all, pairs_all=pairs_all, ms_per_step=ms_per_step,
        step_latency_ms=acc["latency"] / max(steps, 1) * 1e3, gen_seconds=t_gen,
        mean_iters_per_scale=[acc["per_scale"][s] / float(B * steps) for s in range(ns)],
        per_level=per_level, shader_clock=shader_clock,
        roofline={
            "bound": "hbm",
            "kernel": f"ica_iterate_kernel{kernel_key(cfg)} ({TNAME[cfg['transform']]}, "
                      f"{'robust: H+b every iteration' if cfg['robust'] else 'L2: b only, iterations >= 1'}), "
                      "all its launches of the event-bracketed steps at all scales (grids shrink as pairs converge)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": (ratio * dom_bytes / max(dom_n, 1)) if ratio else None,
            "traffic_source": src,
            "algorithmic_bytes_per_launch": dom_bytes / max(dom_n, 1),
            "avg_launch_ms": dom_ms / max(dom_n, 1), "launches": dom_n, "event_bracketed_steps": n_prof,
            "kernel_share_of_step": dom_ms / max(n_prof, 1) / ms_per_step,
            "scale0": {"GBps": (acc["s0_bytes"] / (acc["s0_ms"] * 1e-3) / 1e9) if acc["s0_ms"] > 0 else None,
                       "frac": (acc["s0_bytes"] / (acc["s0_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if acc["s0_ms"] > 0 else None,
                       "avg_launch_ms": acc["s0_ms"] / max(acc["s0_n"], 1), "launches": acc["s0_n"]},
            "first_iteration_kernel_share_of_step": acc["first_ms"] / max(n_prof, 1) / ms_per_step,
            # `achieved` / `frac` use SURVEY 8(d)'s figure, 2*H*W*C*4 bytes per pair-iteration, whatever the storage.
            # With uint8 level-0 images the bytes that exist in HBM at scale 0 are a quarter of that: frac_1B is the
            # scale-0 launches priced at 1 byte per element.
            "bytes_per_element_convention": 4,
            "storage": storage,
            "scale0_frac_1B_convention": ((acc["s0_bytes"] / 4.0 / (acc["s0_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS)
                                          if (storage == "u8" and acc["s0_ms"] > 0) else None),
        },
        quality={"converged_frac": converged, "hit_iteration_cap_frac": hit_cap,
                 "median_abs_param_err_vs_gt": float(perr.median()), "mean_iters_per_pair": units_all / pairs_all},
    )
    del I1, I2, p_gt, res
    for e in engines:
        e.close()
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=DEFAULT_CONFIG, choices=sorted(CONFIGS),
                    help="2..5 = BASELINE.json configs[1..4]; default 4 = configs[3], the headline workload")
    ap.add_argument("--batch", type=int, default=0, help="pairs per GPU (default: the config's)")
    ap.add_argument("--in-flight", type=int, default=IN_FLIGHT_DEFAULT,
                    help="solves kept in flight per GPU (one engine + HIP stream + host thread each): the straggler "
                         "launches of one solve overlap the pyramid and coarse levels of the next")
    ap.add_argument("--storage", choices=("f32", "u8"), default="f32",
                    help="element type of the level-0 images in HBM (uint8: SURVEY 8(f)3; same results, a quarter of the bytes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-per-config", action="store_true", help="skip the per_config side measurements (N = 1 only)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, before anything here touches the GPU
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    import torch

    from inverse_compositional_algorithm_amd import sharding
    stub = os.environ.get("BENCH_STUB") == "1"
    info = sharding.rank_info()
    world, rank, local = info.world, info.rank, info.local_rank
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to report a "
              f"number for a job of another size", file=sys.stderr)
        sys.exit(2)
    if world > 1:
        sharding.pin_rank(info)  # one polling host thread per rank: every rank keeps its own cores
    if world > 1 and not stub:
        torch.cuda.set_device(local)
    sharding.init("gloo" if stub else "nccl", info)  # backend "nccl" is RCCL on ROCm; barrier + 3 scalars only
    dev = torch.device("cpu") if stub else torch.device("cuda", local)
    n_ranks = sharding.count_ranks(info, dev)  # every rank contributes a 1: what the collective library really spans
    if n_ranks != args.gpus:
        print(f"bench.py: the ranks count {n_ranks} of themselves, --gpus says {args.gpus}", file=sys.stderr)
        sharding.finalize(info)
        sys.exit(3)

    if stub:
        r = run_stub(args, info, sharding)
        if rank == 0:
            print(json.dumps({"metric": "IC iterations/sec", "value": r["units_all"] / r["dt"], "unit": "pair-iterations/s",
                              "n_gpus": n_ranks, "steps": args.steps, "warmup": args.warmup, "stub": True,
                              "units": r["units_all"], "pairs": r["pairs_all"]}))
        sharding.finalize(info)
        return

    D = max(1, args.in_flight)
    m = measure(args, info, sharding, args.config, args.steps, args.warmup, D, args.batch, args.storage)
    cfg = m["cfg"]

    per_config = None
    if world == 1 and not args.no_per_config and not args.batch:
        per_config = []
        for cid, steps, stor, infl in SIDE_CONFIGS:
            if cid == args.config and stor == args.storage:
                continue
            if time.perf_counter() - T_START > SIDE_DEADLINE_S:
                per_config.append({"workload": CONFIGS[cid]["name"], "skipped": "time budget of the default run"})
                continue
            try:
                s = measure(args, info, sharding, cid, steps, 1, infl, storage=stor)
            except Exception as exc:  # a side measurement must never cost the headline line
                per_config.append({"workload": CONFIGS[cid]["name"], "error": repr(exc)[:200]})
                continue
            per_config.append({"workload": s["cfg"]["name"] + (" [uint8 level-0 images]" if stor == "u8" else ""),
                               "storage": stor, "in_flight": infl, "value": s["units_all"] / s["dt"], "unit": "pair-iterations/s",
                               "steps": steps, "warmup": 1, "ms_per_step": s["ms_per_step"],
                               "image_pairs_per_s": s["pairs_all"] / s["dt"],
                               "per_level": s["per_level"], "shader_clock": s["shader_clock"],
                               "roofline": {k: s["roofline"][k] for k in ("kernel", "achieved", "frac", "avg_launch_ms",
                                                                          "launches", "traffic", "algorithmic_bytes_per_launch",
                                                                          "scale0", "scale0_frac_1B_convention")},
                               "converged_frac": s["quality"]["converged_frac"],
                               "hit_iteration_cap_frac": s["quality"]["hit_iteration_cap_frac"],
                               "mean_iters_per_pair": s["quality"]["mean_iters_per_pair"]})

    if rank == 0:
        out = {
            "metric": "IC iterations/sec",
            "value": m["units_all"] / m["dt"],
            "unit": "pair-iterations/s",
            "n_gpus": n_ranks,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": m["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",   # the arithmetic type of the path (per pixel; sums and solve are f64), whatever the storage
            "data": "synthetic",
            "config": {"workload": cfg["name"], "pairs_per_gpu": cfg["batch"], "height": cfg["H"], "width": cfg["W"],
                       "channels": cfg["C"], "transform": TNAME[cfg["transform"]], "robust_type": cfg["robust"],
                       "nscales": cfg["nscales"], "tol": 1e-3, "delta": 10, "storage": args.storage,
                       "parallelism": f"{n_ranks} x independent pair shards (no collective)"
                       + (f", {D} solves in flight per GPU" if D > 1 else ""), "in_flight": D,
                       "note": "synthetic pairs: multi-octave noise texture, ground-truth transform of a few percent, "
                               "robust workloads add sigma=5 noise and a 10 % occluder; pairs that exhaust the 30 "
                               "iterations of a scale (quality.hit_iteration_cap_frac) keep iterating like in the reference"},
            "step_latency_ms": m["step_latency_ms"],
            "image_pairs_per_s": m["pairs_all"] / m["dt"],
            "pair_iterations_per_step": m["units_all"] / args.steps,
            "mean_iters_per_scale": m["mean_iters_per_scale"],
            "per_level": m["per_level"],
            "shader_clock": m["shader_clock"],
            "library_fingerprint": library_fingerprint(),
            "roofline": m["roofline"],
            "quality": m["quality"],
        }
        if per_config is not None:
            out["per_config"] = per_config
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, m["mean_iters_per_scale"])
        out["wall_seconds_total"] = time.perf_counter() - T_START
        print(json.dumps(out))
    sharding.finalize(info)


if __name__ == "__main__":
    main()
