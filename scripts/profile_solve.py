"""Runs a few RTI solves of the C1 workload (B=4096, N=30, 'st') for rocprofv3; prints nothing but a summary line."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from safe_mpc_amd.solver import BatchedOcpSolver

par, prob, net = bench.build_problem()
s = BatchedOcpSolver(prob, net)
B = int(os.environ.get('SMPC_B', '4096'))
x0 = bench.initial_states(s, prob, B, 0)
N = prob.N
xg = np.repeat(x0[:, None, :], N + 1, axis=1); ug = np.zeros((B, N, 6)); p = np.zeros((B, N + 1, 5))
p[:, :, :3], p[:, :, 3], p[:, :, 4] = prob.ee_ref, par.alpha, 1.0
x = x0
for i in range(int(os.environ.get('SMPC_STEPS', '6'))):
    xo, uo, st, it = s.solve(x, xg, ug, p)
    acc = (st == 0).astype(np.int32)
    xg, ug, ua = s.provide_control(acc, xo, uo, xg, ug)
    x, _ = s.plant_step(x, ua)
    xg = s.guess_correction(xg, ug)
    print('step', i, 'iters mean', it.mean(), 'max', it.max(), 'fails', (st != 0).sum(), flush=True)
