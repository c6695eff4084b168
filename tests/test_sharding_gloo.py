"""The N > 1 path on CPU: two gloo ranks run the same sharding / aggregation code bench.py uses on GPUs."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from inverse_compositional_algorithm_amd import sharding, synthetic as syn
    info = sharding.rank_info()
    sharding.init("gloo", info)
    dev = torch.device("cpu")
    # every rank owns its own pairs: different seeds -> different data, no exchange of pixels
    I1, I2, p_gt = syn.make_pair(sharding.shard_seed(2, info.rank), 32, 40, 1, 4)
    sharding.barrier(info, dev)
    seconds = 0.5 + 0.25 * info.rank           # rank 1 is the slow one
    units = 100.0 * (info.rank + 1)             # pair-iterations this rank executed
    t, u, p = sharding.aggregate(info, seconds, units, 8.0, dev)
    out[rank] = (t, u, p, float(I2.sum()), sharding.shard_range(10, info.rank, info.world))
    sharding.finalize(info)


def test_two_ranks_weak_scaling_bookkeeping():
    world, port = 2, _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        r0, r1 = out[0], out[1]
    assert r0[:3] == r1[:3] == (0.75, 300.0, 16.0)    # MAX time, SUM units, SUM pairs on every rank
    assert r0[3] != r1[3]                              # shards hold different pairs
    assert r0[4] == (0, 5) and r1[4] == (5, 10)        # contiguous blocks cover the batch exactly once


def test_single_rank_is_a_no_op():
    from inverse_compositional_algorithm_amd import sharding
    info = sharding.RankInfo(0, 1, 0)
    assert sharding.aggregate(info, 1.5, 10.0, 2.0, torch.device("cpu")) == (1.5, 10.0, 2.0)
    assert sharding.shard_range(7, 0, 1) == (0, 7)


# ---- bench.py's own launch path (CPU: the stub engine + gloo stand in for the GPUs + RCCL) -------------------
def _bench(args, env_extra, timeout=240):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r.returncode, (json.loads(lines[-1]) if lines else None), r.stderr


def test_bench_gpus_2_spawns_two_ranks_and_counts_them():
    """`python bench.py --gpus 2` without torchrun starts its two ranks itself; the ranks count themselves with an
    all-reduce and that count is what the JSON line reports."""
    rc, out, err = _bench(["--gpus", "2", "--steps", "3", "--warmup", "0"], {"BENCH_STUB": "1"})
    assert rc == 0, err[-2000:]
    assert out["n_gpus"] == 2 and out["stub"] is True
    assert out["units"] == 100.0 * 3 * (1 + 2) and out["pairs"] == 2 * 8.0 * 3  # SUM over both ranks


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    """Started as ONE rank (as a launcher that ignored --gpus would) bench.py exits non-zero instead of printing a
    1-rank number under an N-GPU label."""
    rc, out, err = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"],
                          {"BENCH_STUB": "1", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert rc != 0 and out is None
    assert "refusing" in err


def test_bench_without_a_gpu_fails_loudly():
    """No GPU, no stub: the product path must fail, not fall back to a CPU computation."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    rc, out, err = _bench(["--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-per-config"], {})
    assert rc != 0 and out is None


def test_bench_gpus_8_rehearsal_on_cpu():
    """The 8-rank launch path the driver uses at round end, rehearsed with gloo and the stub engine: eight ranks are
    started, count themselves, pin themselves to their own cores and aggregate MAX time / SUM units."""
    rc, out, err = _bench(["--gpus", "8", "--steps", "2", "--warmup", "0"], {"BENCH_STUB": "1"}, timeout=600)
    assert rc == 0, err[-2000:]
    assert out["n_gpus"] == 8 and out["stub"] is True
    assert out["units"] == 100.0 * 2 * sum(range(1, 9)) and out["pairs"] == 8 * 8.0 * 2


def test_shard_memory_budget():
    from inverse_compositional_algorithm_amd import sharding
    # BASELINE configs[4]: 1024 pairs of 1920 x 1080 x 3 per GPU, 5 scales -- ~51 GB of level-0 images + ~17 GB of levels
    need = sharding.shard_device_bytes(1024, 1080, 1920, 3, 5, 0.5, 4)
    assert 66e9 < need < 72e9
    assert sharding.shard_device_bytes(1024, 1080, 1920, 3, 5, 0.5, 1) < 0.45 * need   # uint8 level-0 storage
    gib = 2 ** 30
    sharding.check_memory_budget(need, 280 * gib, 288 * gib, "the shard")          # fits an MI355X
    with pytest.raises(MemoryError, match="needs .* GiB of device memory.*fewer pairs"):
        sharding.check_memory_budget(need, 60 * gib, 288 * gib, "the shard")        # someone else holds the card
    with pytest.raises(MemoryError):
        sharding.check_memory_budget(5 * need, 280 * gib, 288 * gib, "a 5120-pair shard")


def test_ranks_keep_disjoint_cores():
    from inverse_compositional_algorithm_amd import sharding
    cores = list(range(3, 19))                      # 16 allowed cores
    got = [sharding.cores_for_rank(r, 8, cores) for r in range(8)]
    assert all(len(g) == 2 for g in got)
    assert sorted(c for g in got for c in g) == cores          # disjoint and complete
    assert sharding.cores_for_rank(0, 1, cores) == cores        # a single rank keeps everything
    few = [sharding.cores_for_rank(r, 8, [0, 1, 2]) for r in range(8)]
    assert all(len(g) == 1 for g in few) and {g[0] for g in few} == {0, 1, 2}
