"""The N>1 path of bench.py on CPU: world_size 2, gloo backend. Proofs shard across ranks with no data-path collective
(replicas); the only collectives are the barriers around the timed region and the MAX-reduce of the elapsed time. The
device prover is replaced by a stand-in here (no GPU on this box); shard() and timed_region() are the real bench code."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    import time
    import numpy as np
    import bench
    conc, steps, warmup = 3, 2, 1
    per_rank = (steps + warmup) * conc
    mine = bench.shard(world * per_rank, world, rank)
    inputs = np.array(mine, dtype=np.int64).reshape(-1, 1)
    seen = []

    def fake_prove_batch(xs):
        time.sleep(0.02 * (rank + 1))  # rank 1 is slower: the MAX over ranks must win
        seen.extend(int(v) for v in xs[:, 0])
        return xs

    elapsed, last = bench.timed_region(fake_prove_batch, inputs, conc, steps, warmup, dist)
    # every rank reports the same (max) time, which is at least the slow rank's sleep
    t = torch.tensor([elapsed], dtype=torch.float64)
    lo, hi = t.clone(), t.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    assert abs(float(lo) - float(hi)) < 1e-9
    assert elapsed >= 0.02 * world * steps * 0.9
    assert seen == mine and len(last) == conc
    # the shards are a partition of the global input set
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    flat = sorted(v for g in gathered for v in g)
    assert flat == list(range(world * per_rank))
    open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    dist.destroy_process_group()


def test_two_rank_replica_sharding_and_timing(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


def test_shard_is_round_robin_partition():
    import bench
    for world in (1, 2, 4, 8):
        parts = [bench.shard(64, world, r) for r in range(world)]
        assert sorted(v for p in parts for v in p) == list(range(64))
        assert all(len(p) == 64 // world for p in parts)
