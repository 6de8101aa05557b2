"""Multi-GPU path on CPU: world_size-2 gloo, each rank solves its slice with the CPU test double, one gather."""
import os
import socket

import numpy as np
import pytest

from conftest import constant_guess, make_problem, sample_instances
from safe_mpc_amd.sharding import shard_by_horizon, shard_range


def test_shard_range_partitions():
    for total in (0, 1, 7, 4096, 4099):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_shard_by_horizon_groups():
    hor = np.repeat([20, 25, 30], [5, 4, 7])
    parts = [shard_by_horizon(hor, 2, r) for r in range(2)]
    for N in (20, 25, 30):
        joined = np.concatenate([p[N] for p in parts])
        assert sorted(joined.tolist()) == np.where(hor == N)[0].tolist()


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from fake_solver import OracleSolver
    from safe_mpc_amd.sharding import gather_to_root
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    par, prob, net = make_problem('st', N=6)
    B = 7                                                   # uneven on purpose
    x0 = sample_instances(prob, B, seed=11)
    xg, ug, p = constant_guess(prob, x0)
    lo, hi = shard_range(B, world, rank)
    s = OracleSolver(prob, net)
    x, u, st, it = s.solve(x0[lo:hi], xg[lo:hi], ug[lo:hi], p[lo:hi])
    payload = torch.tensor(np.hstack([u[:, 0], st[:, None].astype(float)]))
    out = gather_to_root(payload)
    if rank == 0:
        xa, ua, sa, ia = s.solve(x0, xg, ug, p)
        q.put((out.numpy(), np.hstack([ua[:, 0], sa[:, None].astype(float)])))
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gather_matches_single_process():
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    got, want = q.get(timeout=120)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert got.shape == (7, 7) and np.array_equal(got, want)
