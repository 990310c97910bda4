"""Multi-GPU harness helpers: one process per GPU, torch.distributed only for the barrier and the max-over-ranks timing.

The GP path shards as independent units (optimiser runs / whole fits) with no exchange step, so there is
no data-path collective here: `shard_units` is a static partition and the only communication is the timing reduction.
"""
import os


def rank_info():
    """(rank, local_rank, world_size) from the torchrun environment (1 process = 1 GPU)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def shard_units(n_units, world, rank):
    """Units (optimiser runs of one fit, or whole fits) owned by `rank`: unit u -> rank u mod world (gradmin.rs:19-31 axis)."""
    return [u for u in range(n_units) if u % world == rank]


def init(backend=None):
    """Initialise torch.distributed when WORLD_SIZE > 1.  backend: 'nccl' (= RCCL on ROCm) on GPUs, 'gloo' on CPU."""
    rank, local_rank, world = rank_info()
    if world == 1:
        return None
    import torch
    import torch.distributed as dist

    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend=backend)
    return dist


def barrier(dist):
    import torch

    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def max_over_ranks(dist, value):
    """MAX all-reduce of a Python float (timings)."""
    if dist is None:
        return float(value)
    import torch

    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def argmax_over_ranks(dist, value, payload):
    """Host-side arg-max over ranks of (value, payload) — the only 'collective' of a sharded fit (fit.rs:116-125 capture
    across runs): gathers one (lml, theta) pair per rank and keeps the best; ties go to the lowest rank."""
    if dist is None:
        return 0, value, payload
    gathered = [None] * dist.get_world_size()
    dist.all_gather_object(gathered, (float(value), payload))
    best = max(range(len(gathered)), key=lambda r: (gathered[r][0], -r))
    return best, gathered[best][0], gathered[best][1]
