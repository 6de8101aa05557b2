#!/usr/bin/env python3
"""BASELINE config 3 timed: one GPU's share (4 096 of 32 768 instances) of the horizon x alpha grid of run_mpc_horizons.sh /
run_mpc_alphas.sh as ONE closed loop: the instances are grouped by horizon (sharding.shard_by_horizon), every group is a
controller object on its own engine handle / HIP stream (its own N), alpha rides per instance in p[:, :, 3]; one "step" = one
closed-loop step of every group.  Prints ms per step and instance-steps/s.
    python scripts/c3_bench.py [steps] [warmup] [rank]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import numpy as np, torch
import bench
from safe_mpc_amd.controller import get_controller
from safe_mpc_amd.sharding import shard_by_horizon
from safe_mpc_amd.solver import BatchedOcpSolver

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 3
par, prob, net = bench.build_problem()
dev = torch.device('cuda', 0)
n_full = 32768
horizons = np.repeat([20, 25, 30, 35, 40], n_full // 5 + 1)[:n_full]
alphas = np.tile([20.0, 30.0, 40.0, 50.0], n_full // 4)
owned = shard_by_horizon(horizons, 8, rank)
probe = BatchedOcpSolver(prob, net)
starts = bench.initial_states(probe, prob, 512, rank)
t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
groups = []
for N, idx in owned.items():
    par_g, prob_g, net_g = bench.build_problem()
    par_g.N = int(N)
    prob_n = type(prob_g)(par_g, bench.CONTROLLER, 'ext', N=int(N))
    prob_n.set_normalisation(net_g.mean, net_g.std)
    sv = BatchedOcpSolver(prob_n, net_g)
    n = len(idx)
    stream = torch.cuda.ExternalStream(sv.L.smpc_stream(sv.h), device=dev)
    with torch.cuda.stream(stream):
        ctrl = get_controller(bench.CONTROLLER, par_g, n, cost='ext', N=int(N), solver=sv, net=net_g, device_state=True)
        x0 = starts[idx % 512]
        ctrl.setGuess(t(np.repeat(x0[:, None, :], N + 1, axis=1)), t(np.zeros((n, N, 6))))
        ctrl.p[:, :, 3] = t(alphas[idx])[:, None]
        g = dict(N=int(N), n=n, sv=sv, ctrl=ctrl, stream=stream, x=t(x0), xn=t(x0), u=torch.empty((n, 6), dtype=torch.float64, device=dev),
                 ue=torch.empty((n, 6), dtype=torch.float64, device=dev), acc=torch.zeros((3,), dtype=torch.int64, device=dev))
    # (step_on_device hands ctrl.p to the engine as it is: the per-instance alpha column stays)
    groups.append(g)


def step():
    for g in groups:
        with torch.cuda.stream(g['stream']):
            g['ctrl'].step_on_device(g['x'], u_out=g['u'])
            g['sv'].plant_step(g['x'], g['u'], out=(g['xn'], g['ue']))
            g['x'], g['xn'] = g['xn'], g['x']
            g['sv'].accumulate_stats(g['ctrl'].last_status, g['ctrl'].qp_iter, g['acc'])


for _ in range(warm):
    step()
torch.cuda.synchronize()
for g in groups:
    g['sv'].sync(); g['acc'].zero_()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
for g in groups:
    g['sv'].sync()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
B = sum(g['n'] for g in groups)
acc_g = [g['acc'].cpu().numpy().copy() for g in groups]        # (per group, before the event-timed continuation of the loop adds to them)
acc = sum(acc_g)


def sync():
    for g in groups:
        g['sv'].sync()
    torch.cuda.synchronize()


# per-kernel split of the loop's own launches (a continuation of the loop with the event rings on, outside the timed region) and
# the roofline block of the dominant kernel: SURVEY 8(d)'s algorithmic bytes per instance-step, by horizon group
tk = bench.in_loop_kernel_times([g['sv'] for g in groups], step, sync, max(steps, 6))
names = ['linearise', 'mlp', 'qp_setup', 'qp_ipm', 'solve_total']
in_loop = {str(g['N']): {n_: float(t_[i]) for i, n_ in enumerate(names)} for g, t_ in zip(groups, tk) if t_ is not None}
# HBM traffic: k_qp_ipm<6,6,*> moves a fixed number of bytes per STAGE and iteration, so the C1 measurement (31 stages,
# profiles/r05_pmc_traffic.json) gives bytes per instance-iteration at horizon N as x (N + 1) / 31; weighted with the loop's own
# instance-iterations of every horizon group
bpi31, tsrc = None, None
tf = os.path.join(ROOT, 'profiles', 'r05_pmc_traffic.json')
if os.path.exists(tf):
    bpi31, tsrc = json.load(open(tf)).get('bytes_per_instance_iteration'), 'profiles/r05_pmc_traffic.json (x (N + 1) / 31 per horizon group)'
its = [float(a_[0]) / steps for a_ in acc_g]                                     # instance-iterations per step, per group
roof = bench.roofline_of_launches([bench.algorithmic_bytes(6, g['N']) * g['n'] for g, t_ in zip(groups, tk) if t_ is not None],
                                  [t_[3] for t_ in tk if t_ is not None])
if bpi31:
    tr = sum(bpi31 * (g['N'] + 1) / 31.0 * it_ for g, it_ in zip(groups, its))   # bytes per step over all groups' launches
    roof['traffic'] = tr / max(len(groups), 1)
    roof['traffic_source'], roof['traffic_bytes_per_instance_iteration_at_N30'] = tsrc, bpi31
    roof['wasted_traffic_ratio'] = roof['traffic'] / roof['algorithmic_bytes_per_launch']
roof['algorithmic_bytes_per_instance_step'] = {str(g['N']): bench.algorithmic_bytes(6, g['N']) for g in groups}
print(json.dumps({'roofline': roof, 'kernel_ms_in_loop_by_horizon': in_loop,'workload': 'C3: rank %d of 8 of the 32768-instance horizon x alpha grid (N in 20..40, alpha in 20..50), controller %s' % (rank, bench.CONTROLLER),
                  'instances': B, 'groups': {str(g['N']): g['n'] for g in groups}, 'steps': steps, 'warmup': warm,
                  'ms_per_step': 1e3 * dt / steps, 'instance_steps_per_s': B * steps / dt,
                  'mean_ipm_iterations': float(acc[0]) / max(int(acc[2]), 1), 'failed_instance_steps': int(acc[1])}))
