"""Multi-GPU decomposition of the hot path: reads are independent units, so rank r of W maps the
contiguous range shard_range(r, W, n) with the index and reference replicated, and the only exchange is the sum of
the five MappingStats counters (reference src/FEM_map.c:200-212) — one 40-byte all-reduce (RCCL on GPUs, gloo in
the CPU tests)."""
import numpy as np


def shard_range(rank, world, n_reads):
    """Contiguous, balanced [lo, hi) of n_reads for this rank; concatenating the shards in rank order restores the batch."""
    lo = n_reads * rank // world
    hi = n_reads * (rank + 1) // world
    return lo, hi


def reduce_stats(stats, device=None):
    """Sum the five counters over all ranks of the default process group; returns uint64[5] on every rank."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.asarray(stats, dtype=np.uint64).astype(np.int64))
    if device is not None:
        t = t.to(device)
    if dist.is_available() and dist.is_initialized():  # (a group of one rank reduces too: the same RCCL call as eight)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy().astype(np.uint64)
