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


def shard_device_bytes(pairs: int, height: int, width: int, channels: int, nscales: int, nu: float = 0.5,
                       storage_bytes: int = 4) -> int:
    """Device memory one rank's shard needs while it is solved: the two level-0 image batches as stored (float32 or
    uint8), the float32 pyramid levels of both (library workspaces), and the per-(pair, tile) partial sums.  For
    BASELINE configs[4] (1024 pairs of 1920 x 1080 x 3 per GPU, 5 scales) this is ~68 GB of the 288 GB."""
    from .zoom import zoom_size
    total = 2 * pairs * channels * height * width * storage_bytes
    w, h = width, height
    tiles0 = ((width + 63) // 64) * ((height + 511) // 512)
    for _ in range(1, nscales):
        w, h = zoom_size(w, h, nu)
        total += 2 * pairs * channels * max(1, h) * max(1, w) * 4
    total += pairs * max(tiles0, 8) * 48 * 8          # partial sums: 44 float64 per (pair, tile), padded
    total += pairs * (8 + 8 + 64 + 4) * 8              # p, p_prev, H^-1, scalars
    return int(total)


def check_memory_budget(need_bytes: int, free_bytes: int, total_bytes: int, what: str, headroom: float = 0.08) -> None:
    """Refuse a shard that cannot fit BEFORE anything is allocated, with the numbers in the message (an out-of-memory
    error in the middle of data generation names neither the shard nor the remedy)."""
    budget = free_bytes - headroom * total_bytes
    if need_bytes > budget:
        raise MemoryError(
            f"{what} needs {need_bytes / 2**30:.1f} GiB of device memory; {free_bytes / 2**30:.1f} GiB of "
            f"{total_bytes / 2**30:.1f} GiB are free ({headroom:.0%} of the total is kept as headroom): use fewer pairs "
            f"per GPU (--batch), uint8 storage (--storage u8), or more GPUs")


def cores_for_rank(local_rank: int, local_world: int, cores) -> list:
    """The slice of this process's allowed cores that rank `local_rank` of `local_world` keeps for itself.  A rank's
    solving thread polls the launch-sizing board for the whole solve (include/ica_hip.h): ranks that share a core
    take turns at it.  Contiguous, disjoint slices; with fewer cores than ranks the ranks share round-robin."""
    cores = sorted(cores)
    if not cores or local_world <= 1:
        return cores
    if len(cores) < local_world:
        return [cores[local_rank % len(cores)]]
    per = len(cores) // local_world
    return cores[local_rank * per:(local_rank + 1) * per]


def pin_rank(info: RankInfo, local_world: int = None) -> list:
    """Pin this rank's process to its own cores (Linux; a no-op elsewhere).  Returns the cores it now runs on."""
    if not hasattr(os, "sched_setaffinity"):
        return []
    local_world = local_world or int(os.environ.get("LOCAL_WORLD_SIZE", info.world))
    mine = cores_for_rank(info.local_rank, local_world, os.sched_getaffinity(0))
    if mine:
        os.sched_setaffinity(0, mine)
    return mine


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
