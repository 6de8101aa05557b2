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


def _worker_prealloc(rank, world, port, q):
    """the bench's per-step form: buffers allocated once (gather_buffers), two steps, no concatenation on the root."""
    import torch
    import torch.distributed as dist
    from safe_mpc_amd.sharding import gather_buffers, gather_to_root
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    total = 9                                               # strong scaling: 9 instances in total over 2 ranks (5 + 4)
    sizes = [shard_range(total, world, r)[1] - shard_range(total, world, r)[0] for r in range(world)]
    lo, hi = shard_range(total, world, rank)
    local = torch.zeros((hi - lo, 3), dtype=torch.float64)
    bufs = gather_buffers(local, sizes, rank)
    got = []
    for step in range(2):
        local[:] = torch.arange(lo, hi, dtype=torch.float64)[:, None] + 100.0 * step
        views = gather_to_root(local, sizes=sizes, bufs=bufs, concat=False)
        if rank == 0:
            got.append(torch.cat(views).numpy().copy())
    if rank == 0:
        q.put(got)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gather_with_preallocated_buffers():
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_prealloc, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = q.get(timeout=120)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    for step in range(2):
        assert np.array_equal(got[step][:, 0], np.arange(9) + 100.0 * step)


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus N` must never fall back to fewer GPUs silently: with no launcher environment it
    spawns the ranks itself, and exits non-zero before touching a GPU when N exceeds what is visible."""
    import subprocess
    import sys
    import torch
    n = torch.cuda.device_count()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(max(n + 1, 2)), '--steps', '1', '--warmup', '0'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == '' and 'refusing' in r.stderr
    # a launcher environment that disagrees with --gpus is an error as well
    env2 = dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1'], env=env2,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ''
