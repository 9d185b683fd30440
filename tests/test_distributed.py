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


# ---------------------------------------------------------------------------------------------- sharded sumcheck
def _sharded_case(nv=7, seed=5):
    """k = 3 base tables and a second term, like the standalone sumcheck bench plus a mixed-degree term"""
    import numpy as np
    P = 0xFFFFFFFF00000001
    rng = np.random.default_rng(seed)
    tabs = [rng.integers(0, P, size=1 << nv, dtype=np.uint64) for _ in range(3)]
    terms = [((1, 0), [0, 1, 2]), ((5, 7), [1, 2])]
    return tabs, terms


def _prove_sharded_py(tabs, terms, nv, world, rank_or_none, exchange):
    """drive deep_prove_amd.sharded.prove_sharded with the pure-Python shard backend"""
    import deep_prove_amd as dpa
    from support.py_shard import PyShard, words_to_exts
    chunk = (1 << nv) // world
    ranks = range(world) if rank_or_none is None else [rank_or_none]
    shards = [PyShard([[(int(v), 0) for v in t[g * chunk:(g + 1) * chunk]] for t in tabs], terms) for g in ranks]
    return dpa.sharded.prove_sharded(shards, exchange, nv, terms, dpa.Transcript(b"test"),
                                     lambda tw: PyShard([words_to_exts(w) for w in tw], terms))


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_sharded_sumcheck_is_bit_identical_to_the_unsharded_oracle_proof(oracle, world):
    """W workers own contiguous slices (top variables select the worker); shares are added mod p; the last
    log2(W) rounds run on the merged W-entry tables. The proof stream must equal the oracle's prove_parallel."""
    import deep_prove_amd as dpa
    nv = 7
    tabs, terms = _sharded_case(nv)
    proof, finals = _prove_sharded_py(tabs, terms, nv, world, None, dpa.sharded.LocalExchange(world))
    oproof, ofinals = oracle.sumcheck_prove(nv, tabs, [False] * 3, terms, oracle.transcript(b"test"))
    assert proof.size == oproof.size and (proof == oproof).all()
    assert (finals == ofinals).all()


def _gloo_shard_worker(rank, world, port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import deep_prove_amd as dpa
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    nv = 7
    tabs, terms = _sharded_case(nv)
    proof, finals = _prove_sharded_py(tabs, terms, nv, world, rank, dpa.sharded.TorchExchange())
    q.put((rank, proof.tolist(), finals.tolist()))
    dist.destroy_process_group()


def test_sharded_sumcheck_over_gloo_world_size_2(oracle):
    """two processes, one shard each, shares all-gathered with torch.distributed (gloo here, RCCL on the GPUs)"""
    import multiprocessing as mp
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_gloo_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    tabs, terms = _sharded_case(7)
    oproof, ofinals = oracle.sumcheck_prove(7, tabs, [False] * 3, terms, oracle.transcript(b"test"))
    for rank, proof, finals in res:
        assert proof == oproof.tolist() and finals == ofinals.tolist(), f"rank {rank} differs"


def test_bench_guarded_section():
    """bench.guarded: the optional sharded-sumcheck section of a multi-GPU run may fail or hang without taking the headline
    JSON line down"""
    import threading
    import time
    import bench
    fired = threading.Event()
    assert bench.guarded(lambda: {"ok": 1}, 5.0, fired.set) == {"ok": 1}
    out = bench.guarded(lambda: (_ for _ in ()).throw(RuntimeError("rank died")), 5.0, fired.set)
    assert out == {"error": "RuntimeError: rank died"} and not fired.is_set()
    bench.guarded(lambda: time.sleep(0.6), 0.2, fired.set)  # a section that outlives its budget: the watchdog fires
    assert fired.is_set()
    fired.clear()
    bench.guarded(lambda: None, 0.2, fired.set)  # ... and is disarmed when the section returns in time
    time.sleep(0.4)
    assert not fired.is_set()


def _strong_worker(rank, world, port, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    import numpy as np
    import bench
    B, steps, warmup = 7, 2, 1  # an odd batch: the shares differ by one
    mine = bench.strong_share(B, world, rank)
    ids = [s * B + j for s in range(steps + warmup) for j in bench.shard(B, world, rank)]
    inputs = np.array(ids, dtype=np.int64).reshape(-1, 1)
    seen = []
    elapsed, last = bench.timed_region(lambda xs: (seen.extend(int(v) for v in xs[:, 0]), xs)[1], inputs, mine, steps, warmup, dist)
    assert seen == ids and len(last) == mine
    gathered = [None] * world
    dist.all_gather_object(gathered, ids)
    assert sorted(v for g in gathered for v in g) == list(range((steps + warmup) * B))  # every proof of every step exactly once
    open(os.path.join(tmpdir, f"strong{rank}"), "w").write("ok")
    dist.destroy_process_group()


def test_config4_fixed_batch_split_over_two_ranks(tmp_path):
    """bench.py --batch B (BASELINE config 4, strong scaling): the shares of one fixed batch partition it, step by step"""
    import bench
    for world in (1, 2, 4, 8):
        assert sum(bench.strong_share(64, world, r) for r in range(world)) == 64
        assert max(bench.strong_share(64, world, r) for r in range(world)) == 64 // world
    world = 2
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_strong_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"strong{r}") for r in range(world))


@pytest.mark.parametrize("extra", [[], ["--batch", "64"]])
def test_bench_spawns_its_own_ranks(extra):
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (how the driver's BENCH command line looks) re-executes
    itself under torch.distributed.run: two ranks come up over gloo on 127.0.0.1, all-reduce, rank 0 prints one JSON line.
    DP_BENCH_LAUNCH_CHECK stops before the device is touched (no GPU on this box)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DP_BENCH_LAUNCH_CHECK="1", DP_DIST_BACKEND="gloo", DP_FORCE_DEVICE="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"] + extra,
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["launch_check"] and rec["world"] == 2 and rec["ranks_seen"] == 2 and rec["batch"] == (64 if extra else 0)


def test_clock_sampler_reads_a_hwmon_tree(tmp_path):
    """bench.ClockSampler on a fake amdgpu device directory: units (Hz, uW, millidegrees), summary statistics, and silence when nothing can be read"""
    import time
    import bench
    hw = tmp_path / "hwmon" / "hwmon3"
    hw.mkdir(parents=True)
    (hw / "freq1_input").write_text("2349000000\n")
    (hw / "power1_input").write_text("769000000\n")
    (hw / "temp2_input").write_text("50000\n")
    (tmp_path / "gpu_busy_percent").write_text("100\n")
    s = bench.ClockSampler(0, period_s=0.01, sysfs_device=str(tmp_path)).start()
    time.sleep(0.1)
    (hw / "freq1_input").write_text("2313000000\n")
    time.sleep(0.1)
    r = s.stop()
    assert r["samples"] >= 4 and r["sclk_mhz"]["max"] == 2349.0 and r["sclk_mhz"]["min"] == 2313.0 and 2313.0 < r["sclk_mhz"]["mean"] < 2349.0
    assert r["power_w"]["mean"] == 769.0 and r["busy_pct"]["median"] == 100.0 and r["temp_c"]["max"] == 50.0
    empty = tmp_path / "none"
    empty.mkdir()
    e = bench.ClockSampler(0, sysfs_device=str(empty))
    e.smi = None  # (no rocm-smi fallback in the test)
    assert e.start().stop() is None
