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
