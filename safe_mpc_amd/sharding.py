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


def gather_buffers(local, sizes, rank, dst=0):
    """Receive buffers of :func:`gather_to_root`, allocated once by a caller that gathers every step."""
    import torch
    m = max(sizes)
    pad = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    recv = [torch.empty_like(pad) for _ in sizes] if rank == dst else None
    return pad, recv


def gather_to_root(local, sizes=None, dst=0, bufs=None, concat=True):
    """Gather per-rank result tensors of shape [B_r, ...] on ``dst``; returns the concatenation on dst, None elsewhere.
    Uneven shards are padded to the largest one (torch.distributed.gather needs equal shapes).  ``bufs`` (from
    :func:`gather_buffers`) keeps the pad / receive buffers out of the per-step path; with ``concat=False`` the root
    gets the list of per-rank views instead of a fresh concatenation."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    if sizes is None:
        n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        all_n = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(all_n, n)
        sizes = [int(v.item()) for v in all_n]
    pad, recv = bufs if bufs is not None else gather_buffers(local, sizes, rank, dst)
    if pad.shape[0] == local.shape[0]:
        src = local                          # even shards: no staging copy at all
    else:
        pad[:local.shape[0]] = local
        src = pad
    dist.gather(src, recv, dst=dst)
    if rank != dst:
        return None
    views = [b[:s] for b, s in zip(recv, sizes)]
    return torch.cat(views, dim=0) if concat else views
