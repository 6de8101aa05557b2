"""Multi-GPU readiness on ONE GPU: the path the driver takes on an 8-GPU node -- bench.py as a torch.distributed
rank over RCCL, the rollout log gathered to rank 0, exactly one JSON line on stdout -- exercised with a single rank
(SMPC_FORCE_DIST=1).  The file sorts first on purpose: bench.py must be started as a fresh child by a process that has not
touched the GPU yet (a GPU-initialised process must never exec another program on this pool)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_runs_as_a_torch_distributed_rank_over_rccl():
    import torch
    from safe_mpc_amd import _lib
    if torch.cuda.is_initialized() or _lib._lib is not None:
        pytest.skip('needs a process that has not initialised the GPU (run the whole suite, or this file alone)')
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, SMPC_FORCE_DIST='1', RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1', '--no-cpu-baseline'],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]                      # the contract: ONE JSON line (RCCL's banner goes to stderr)
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 1 and d['value'] > 0 and d['scaling'] == 'weak'
    assert d['config']['failed_instance_steps'] == 0 and 3.0 < d['config']['mean_ipm_iterations'] < 12.0
    assert d['roofline']['bound'] == 'hbm' and d['roofline']['achieved'] > 0 and d['cpu_baseline'] is None
    assert d['roofline']['kernel_ms_in_loop']['launches_sampled'] > 0
    # the host side of the loop, for the day eight ranks share one host
    assert 0.0 < d['host_issue_ms_per_step'] <= d['ms_per_step'] * 1.05 and d['usable_cores'] >= 1
    assert d['config']['hip_graphs'] is False


@pytest.mark.gpu
def test_bench_graph_replay_is_legal_under_torch_distributed():
    """--graphs 1 as a torch.distributed rank: the captured step takes its log slot from a device-side counter, so the rollout
    log that is gathered holds every step (SMPC_BENCH_CHECK_LOG: bench.py verifies its own log after the timed region)."""
    import torch
    from safe_mpc_amd import _lib
    if torch.cuda.is_initialized() or _lib._lib is not None:
        pytest.skip('needs a process that has not initialised the GPU (run the whole suite, or this file alone)')
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, SMPC_FORCE_DIST='1', RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'),
               SMPC_BENCH_CHECK_LOG='1')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '6', '--warmup', '3', '--graphs', '1',
                        '--no-cpu-baseline', '--no-survey-window'], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    assert d['config']['hip_graphs'] is True and d['steps'] == 6
    assert d['config']['failed_instance_steps'] == 0 and 3.0 < d['config']['mean_ipm_iterations'] < 12.0
    assert d['log_check'] == 'ok: 6 distinct log rows, statuses all zero'


@pytest.mark.gpu
@pytest.mark.parametrize('cfg,extra', [('c2', ['--batch', '2048']), ('c3', ['--batch', '640']), ('c4', ['--batch', '768']),
                                       ('c3', ['--batch', '5120', '--scaling', 'strong'])])
def test_every_baseline_config_runs_through_the_one_launcher_as_an_rccl_rank(cfg, extra):
    """bench.py --config c2 | c3 | c4: the same launcher, the same single RCCL gather of the rollout log, `roofline`
    in every line -- at reduced sizes, as one torch.distributed rank.  (bench.py runs as a child process: it starts its own GPU context.)"""
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, SMPC_FORCE_DIST='1', RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'), SMPC_BENCH_CHECK_LOG='1')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--config', cfg, '--steps', '3', '--warmup', '1',
                        '--no-cpu-baseline', '--no-survey-window'] + extra, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['config']['name'] == cfg and d['n_gpus'] == 1 and d['value'] > 0
    # (C2: a perturbed plant leaves the model's prediction, and now and then a QP starts outside its bounds: under 1 % of the solves)
    assert d['config']['failed_instance_steps'] <= (0.01 * 3 * d['config']['batch_per_gpu'] if cfg == 'c2' else 0)
    assert d['config']['workload'].startswith(cfg.upper())
    assert d['roofline']['bound'] == 'hbm' and d['roofline']['achieved'] > 0 and 0 < d['roofline']['frac'] < 1
    assert ('3 distinct' in d['log_check']) and (cfg == 'c2' or d['log_check'].startswith('ok:'))      # (C2's few failed solves are in the log)
    if cfg == 'c3':
        assert d['config']['horizon'] == [20, 25, 30, 35, 40] and d['config']['streams_per_gpu'] == 5
        assert d['config']['batch_per_gpu'] in (644, 5120) and 'in_loop_all_groups' in d['roofline']      # (644: rank 0's share of 8 x 640, remainders included)
    if cfg == 'c4':
        assert d['config']['controller'] == 'constraint_everywhere' and d['config']['horizon'] == 40
