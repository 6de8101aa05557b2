#!/usr/bin/env python3
"""BASELINE config 4 timed: one GPU's share (16 384 of 131 072 instances) of the 7-DoF Franka-class problem (config_fr7.yaml:
N = 40, sphere obstacle + floor rows, safe-set row on EVERY node through the MLP, controller 'constraint_everywhere') as a closed
loop on three sub-batch streams -- the same loop as bench.py.  Prints ms per step and instance-steps/s.
    python scripts/c4_bench.py [steps] [warmup] [instances] [streams]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import numpy as np, torch
import bench
from safe_mpc_amd.controller import get_controller
from safe_mpc_amd.parser import Parameters
from safe_mpc_amd.problem import OcpProblem
from safe_mpc_amd.safe_set import SafeSetNet
from safe_mpc_amd.sharding import shard_range
from safe_mpc_amd.solver import BatchedOcpSolver

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
N, S, CONT = 40, (int(sys.argv[4]) if len(sys.argv) > 4 else 3), 'constraint_everywhere'
par = Parameters({}, 'fr7', filename=os.path.join(ROOT, 'config_fr7.yaml'))
par.N = N
prob = OcpProblem(par, CONT, 'ext', N=N)
net = SafeSetNet.from_params(par, prob.x_min, prob.x_max)
prob.set_normalisation(net.mean, net.std)
dev = torch.device('cuda', 0)
probe = BatchedOcpSolver(prob, net)
nq = prob.nq
x0 = bench.initial_states(probe, prob, B, 0)
x0[:, nq:] = 0.1 * np.random.default_rng(0).uniform(-1, 1, (B, nq)) * prob.ubx[nq:]
t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
groups = []
for i in range(S):
    lo, hi = shard_range(B, S, i)
    sv = BatchedOcpSolver(prob, net)
    stream = torch.cuda.ExternalStream(sv.L.smpc_stream(sv.h), device=dev)
    with torch.cuda.stream(stream):
        ctrl = get_controller(CONT, par, hi - lo, cost='ext', N=N, solver=sv, net=net, device_state=True)
        xs = x0[lo:hi]
        ctrl.setGuess(t(np.repeat(xs[:, None, :], N + 1, axis=1)), t(np.zeros((hi - lo, N, nq))))
        groups.append(dict(n=hi - lo, sv=sv, ctrl=ctrl, stream=stream, x=t(xs), xn=t(xs),
                           u=torch.empty((hi - lo, nq), dtype=torch.float64, device=dev), ue=torch.empty((hi - lo, nq), dtype=torch.float64, device=dev),
                           acc=torch.zeros((3,), dtype=torch.int64, device=dev), itmax=torch.zeros((1,), dtype=torch.int32, device=dev)))


def step():
    for g in groups:
        with torch.cuda.stream(g['stream']):
            g['ctrl'].step_on_device(g['x'], u_out=g['u'])
            g['sv'].plant_step(g['x'], g['u'], out=(g['xn'], g['ue']))
            g['x'], g['xn'] = g['xn'], g['x']
            g['sv'].accumulate_stats(g['ctrl'].last_status, g['ctrl'].qp_iter, g['acc'])


def sync():
    for g in groups:
        g['sv'].sync()
    torch.cuda.synchronize()


for _ in range(warm):
    step()
sync()
for g in groups:
    g['acc'].zero_()
sync()
t0 = time.perf_counter()
for _ in range(steps):
    step()
sync()
dt = time.perf_counter() - t0
acc = sum(g['acc'].cpu().numpy() for g in groups)
# per-kernel split of the loop's own launches (a continuation of the loop with the event rings on, outside the timed region) and
# the roofline block of the dominant kernel: SURVEY 8(d)'s algorithmic bytes per instance-step (15 420 B at 7-DoF, N = 40)
tk = bench.in_loop_kernel_times([g['sv'] for g in groups], step, sync, max(steps, 6))
names = ['linearise', 'mlp', 'qp_setup', 'qp_ipm', 'solve_total']
in_loop = {n_: float(np.mean([t_[i] for t_ in tk if t_ is not None])) for i, n_ in enumerate(names)}
alg1 = bench.algorithmic_bytes(nq, N)
# HBM traffic of the dominant kernel: bytes per instance-iteration of k_qp_ipm<7,4,true> from the rocprofv3 --pmc passes over this very
# problem (scripts/prof_pmc.sh <dir> c4 -> profiles/r05_pmc_traffic_c4.json) x the loop's own instance-iterations per launch
bpi, tsrc = None, None
tf = os.path.join(ROOT, 'profiles', 'r05_pmc_traffic_c4.json')
if os.path.exists(tf):
    bpi, tsrc = json.load(open(tf)).get('bytes_per_instance_iteration'), 'profiles/r05_pmc_traffic_c4.json'
n_launch = len([t_ for t_ in tk if t_ is not None])
roof = bench.roofline_of_launches([alg1 * g['n'] for g, t_ in zip(groups, tk) if t_ is not None], [t_[3] for t_ in tk if t_ is not None],
                                  bpi, (float(acc[0]) / steps) if bpi else None)
roof['traffic_source'], roof['traffic_bytes_per_instance_iteration'] = tsrc, bpi
if roof.get('traffic'):
    roof['wasted_traffic_ratio'] = roof['traffic'] / roof['algorithmic_bytes_per_launch']
    roof['traffic_GBps'] = roof['traffic'] / (roof['avg_launch_ms'] * 1e-3) / 1e9
roof['algorithmic_bytes_per_instance_step'] = alg1
print(json.dumps({'roofline': roof, 'kernel_ms_in_loop': in_loop,'workload': 'C4: 7-DoF Franka-class, N=40, %d instances (one GPU of 8 x 16384), controller %s, sphere + floor rows, NN row on every node' % (B, CONT),
                  'instances': B, 'streams': S, 'steps': steps, 'warmup': warm, 'ms_per_step': 1e3 * dt / steps,
                  'instance_steps_per_s': B * steps / dt, 'mean_ipm_iterations': float(acc[0]) / max(int(acc[2]), 1),
                  'failed_instance_steps': int(acc[1]), 'last_step_max_iterations': int(max(int(g['ctrl'].qp_iter.max().item()) for g in groups))}))
