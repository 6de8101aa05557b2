"""Multi-GPU layout of the batch: one process per GPU, contiguous instance slices, no data-path collective; the only
exchange is one gather of results to rank 0.  ``torch.distributed`` backend "nccl" is RCCL on ROCm (xGMI);
"gloo" is used by the CPU tests.
"""
from __future__ import annotations

import numpy as np


def shard_range(total, world, rank):
    """Contiguous, balanced slice of ``total`` instances for ``rank`` (earlier ranks take the remainder)."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_by_horizon(horizons, world, rank):
    """Sweep grids (run_mpc_horizons.sh / run_mpc_alphas.sh): group instances by horizon first (a solver handle has one
    N), then slice every group across ranks.  Returns {N: indices owned by this rank}."""
    horizons = np.asarray(horizons)
    out = {}
    for N in np.unique(horizons):
        idx = np.where(horizons == N)[0]
        lo, hi = shard_range(len(idx), world, rank)
        out[int(N)] = idx[lo:hi]
    return out


def gather_to_root(local, sizes=None, dst=0):
    """Gather per-rank result tensors of shape [B_r, ...] on ``dst``; returns the concatenation on dst, None elsewhere.
    Uneven shards are padded to the largest one (torch.distributed.gather needs equal shapes)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    if sizes is None:
        n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        all_n = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(all_n, n)
        sizes = [int(v.item()) for v in all_n]
    m = max(sizes)
    pad = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], dim=0)
