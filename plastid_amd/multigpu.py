"""One-process-per-GPU plumbing for the counting path.

The path shards by independent units (genome ranges / chains / samples): every
rank stages its own records and counts its own intervals; count vectors are never
exchanged.  The only collective is a tiny all-reduce of summary totals (RCCL over
xGMI with the ``nccl`` backend; ``gloo`` on CPU in the tests).  Float totals are
reduced in fixed rank order so the result does not depend on the ring schedule.
"""
import os

import numpy as np


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend, device=None):
    """Join the process group described by the torchrun environment (no-op for world size 1)."""
    import torch.distributed as dist
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kwargs = {}
        if backend == "nccl" and device is not None:
            kwargs["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, local_rank, world


def shard_chains(n_chains, rank, world):
    """Contiguous, balanced shard of chain indices for `rank` (chains are independent units)."""
    bounds = np.linspace(0, n_chains, world + 1).astype(np.int64)
    return np.arange(bounds[rank], bounds[rank + 1])


def allreduce_int_totals(values, device="cpu"):
    """Sum a small vector of int64 totals over all ranks (exact)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(x) for x in t.tolist()]


def reduce_float_totals_ordered(values, device="cpu"):
    """Sum float64 totals over ranks in FIXED rank order (all-gather, then a left-to-right sum),
    so center-mapping totals are reproducible bit for bit."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return t.tolist()
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    acc = torch.zeros_like(t)
    for p in parts:  # rank order
        acc = acc + p
    return acc.tolist()


def max_over_ranks(x, device="cpu"):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
