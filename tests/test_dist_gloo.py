"""The N>1 harness path on CPU: world_size-2 gloo processes exercise the unit sharding, the barrier, the max-over-ranks
timing reduction and the host-side arg-max that replaces a collective."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from hbetune_rs_amd import dist as D

    dist = D.init(backend="gloo")
    units = D.shard_units(8, world, rank)  # C3: 8 optimiser runs
    D.barrier(dist)
    elapsed = D.max_over_ranks(dist, 1.0 + rank)  # rank 1 is "slower"
    # each rank reports its best (lml, theta) over its own runs; lml grows with the run index here
    my_best = max(units)
    best_rank, best_val, payload = D.argmax_over_ranks(dist, float(my_best), {"run": my_best})
    D.barrier(dist)
    q.put((rank, units, elapsed, best_rank, best_val, payload))
    dist.destroy_process_group()


def test_two_rank_gloo_harness():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort()
    assert results[0][1] == [0, 2, 4, 6] and results[1][1] == [1, 3, 5, 7]  # no unit lost or duplicated
    assert all(r[2] == 2.0 for r in results)  # MAX over ranks
    assert all(r[3] == 1 and r[4] == 7.0 and r[5] == {"run": 7} for r in results)  # same winner everywhere


def test_shard_units_partition():
    from hbetune_rs_amd import dist as D

    for world in (1, 2, 3, 4, 8):
        seen = sorted(u for r in range(world) for u in D.shard_units(8, world, r))
        assert seen == list(range(8))


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without a torchrun environment must start two ranks itself and report n_gpus = 2
    (rehearsed on CPU with --dry: gloo, no GPU work; the barrier / max-over-ranks / rank-0 print are the real ones)."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "0", "--dry"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout  # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    assert out["ms_per_step"] >= 19.0  # rank 1 takes 20 ms per step: MAX over ranks, not rank 0's 10 ms


def test_bench_refuses_a_mismatched_world(tmp_path):
    import subprocess

    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry"], capture_output=True, text=True,
                         timeout=120, env=env, cwd=str(tmp_path))
    assert res.returncode == 2 and "WORLD_SIZE" in res.stderr
