"""Multi-GPU execution of the hot path: independent image pairs shard across ranks with NO data-path collective
(SURVEY.md 8e).  One process per GPU; each rank owns `pairs_per_gpu` pairs generated on its own device.  The only
communication is the benchmark's bookkeeping: a barrier around the timed region, MAX of the wall time and SUM of
the processed units -- three scalars per run (RCCL on GPUs, gloo in the CPU tests)."""
from __future__ import annotations

import os
from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass
class RankInfo:
    rank: int
    world: int
    local_rank: int


def rank_info() -> RankInfo:
    return RankInfo(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
                    int(os.environ.get("LOCAL_RANK", "0")))


def init(backend: str, info: RankInfo) -> None:
    if info.world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", info.local_rank)
        dist.init_process_group(backend, rank=info.rank, world_size=info.world, **kw)


def shard_seed(config_id: int, rank: int) -> int:
    """Seed of a rank's shard of synthetic pairs: distinct per (config, rank), identical across runs."""
    return 1000 * config_id + 7919 * rank


def shard_range(total_pairs: int, rank: int, world: int):
    """Contiguous block of a fixed global batch (strong-scaling helper; the bench uses weak scaling)."""
    per = (total_pairs + world - 1) // world
    lo = min(total_pairs, rank * per)
    return lo, min(total_pairs, lo + per)


def barrier(info: RankInfo, device=None) -> None:
    if info.world > 1:
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def count_ranks(info: RankInfo, device) -> int:
    """How many ranks the process group really spans: every rank contributes a 1 to a SUM all-reduce.  bench.py
    prints this as `n_gpus` and refuses to report when it differs from --gpus."""
    if info.world <= 1:
        return 1
    one = torch.ones(1, dtype=torch.float64, device=device)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    return int(round(float(one[0])))


def aggregate(info: RankInfo, seconds: float, units: float, pairs: float, device) -> tuple:
    """(max seconds over ranks, total units, total pairs)."""
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    s = torch.tensor([units, pairs], dtype=torch.float64, device=device)
    if info.world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return float(t[0]), float(s[0]), float(s[1])


def finalize(info: RankInfo) -> None:
    if info.world > 1 and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
