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


# ---- C3's partition end to end: strong scaling over a horizon x alpha grid ----------------------------------------
def _grid_case():
    """a small horizon x alpha grid in the layout of run_mpc_horizons.sh / run_mpc_alphas.sh: 11 instances (uneven on purpose)"""
    horizons = np.array([4, 4, 4, 4, 6, 6, 6, 5, 5, 5, 5])
    alphas = np.array([10.0, 20.0, 30.0, 40.0, 10.0, 20.0, 30.0, 10.0, 20.0, 30.0, 40.0])
    return horizons, alphas


def _solve_group(N, idx, alphas, x0_all):
    """two closed-loop steps of the instances `idx` (global indices) at horizon N with the CPU test double: [u applied at step 1, status]"""
    from fake_solver import OracleSolver
    par, prob, net = make_problem('st', N=int(N))
    s = OracleSolver(prob, net)
    x = x0_all[idx]
    xg, ug, p = constant_guess(prob, x)
    p[:, :, 3] = alphas[idx][:, None]
    out = np.zeros((len(idx), prob.nu + 1))
    for _ in range(2):
        xg = s.guess_correction(xg, ug)
        xt, ut, st, it = s.solve(x, xg, ug, p)
        xg, ug, u = s.provide_control((st == 0).astype(np.int32), xt, ut, xg, ug)
        x = s.plant_step(x, u)[0]
        out[:, :prob.nu], out[:, prob.nu] = u, st
    return out


def _worker_grid(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from safe_mpc_amd.sharding import gather_to_root
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    horizons, alphas = _grid_case()
    par, prob, net = make_problem('st', N=4)
    x0_all = sample_instances(prob, len(horizons), seed=5)
    owned = shard_by_horizon(horizons, world, rank)                 # {N: global indices of this rank}
    rows = []
    for N, idx in sorted(owned.items()):
        if len(idx):
            rows.append(np.hstack([idx[:, None].astype(float), _solve_group(N, idx, alphas, x0_all)]))
    local = torch.tensor(np.vstack(rows)) if rows else torch.zeros((0, 1 + prob.nu + 1), dtype=torch.float64)
    got = gather_to_root(local)                                     # ONE gather: [global index | u | status], ragged over ranks
    if rank == 0:
        q.put(got.numpy())
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_strong_scaling_grouped_by_horizon_matches_single_process():
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_grid, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = q.get(timeout=180)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    horizons, alphas = _grid_case()
    # every instance of the grid exactly once, whatever rank and horizon group it went through
    order = np.argsort(got[:, 0])
    assert np.array_equal(got[order, 0], np.arange(len(horizons)))
    # ... and with the result of the unsharded run (instances are independent: bit-identical on the same code path)
    par, prob, net = make_problem('st', N=4)
    x0_all = sample_instances(prob, len(horizons), seed=5)
    for N in np.unique(horizons):
        idx = np.where(horizons == N)[0]
        want = _solve_group(N, idx, alphas, x0_all)
        assert np.array_equal(got[order][idx, 1:], want), N
    assert (got[:, -1] == 0).all()


# ---- C4's partition end to end: the 7-DoF problem split by shard_range (bench.py --config c4 --scaling strong) -----
def _solve_fr7(idx, x0_all):
    """two closed-loop steps of the instances `idx` of the 7-DoF / row-on-every-node problem with the CPU test double"""
    from conftest import make_problem_fr7
    from fake_solver import OracleSolver
    par, prob, net = make_problem_fr7(N=6)
    s = OracleSolver(prob, net)
    x = x0_all[idx]
    xg, ug, p = constant_guess(prob, x, alpha=par.alpha)
    out = np.zeros((len(idx), prob.nu + 1))
    for _ in range(2):
        xg = s.guess_correction(xg, ug)
        xt, ut, st, it = s.solve(x, xg, ug, p)
        xg, ug, u = s.provide_control((st == 0).astype(np.int32), xt, ut, xg, ug)
        x = s.plant_step(x, u)[0]
        out[:, :prob.nu], out[:, prob.nu] = u, st
    return out


def _worker_c4(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from conftest import make_problem_fr7
    from safe_mpc_amd.sharding import gather_to_root
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    par, prob, net = make_problem_fr7(N=6)
    total = 7                                                        # (uneven on purpose: 4 + 3)
    x0_all = sample_instances(prob, total, seed=9)
    lo, hi = shard_range(total, world, rank)
    idx = np.arange(lo, hi)
    local = torch.tensor(np.hstack([idx[:, None].astype(float), _solve_fr7(idx, x0_all)]))
    got = gather_to_root(local)                                      # ONE gather: [global index | u | status]
    if rank == 0:
        q.put(got.numpy())
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_c4_split_matches_single_process():
    import torch.multiprocessing as mp
    from conftest import make_problem_fr7
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_c4, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = q.get(timeout=240)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert np.array_equal(got[:, 0], np.arange(7))                  # contiguous slices arrive in rank order
    par, prob, net = make_problem_fr7(N=6)
    x0_all = sample_instances(prob, 7, seed=9)
    want = _solve_fr7(np.arange(7), x0_all)
    assert np.array_equal(got[:, 1:], want)                         # instances are independent: bit-identical on the same code path
    assert got.shape[1] == 1 + 7 + 1
