#!/usr/bin/env python3
"""smpc_rollout_batch (the plain RTI policy + plant, the whole closed loop inside the engine) at the bench's workload size:
ms per 4096-instance step by number of sub-batch worker streams."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from safe_mpc_amd.solver import BatchedOcpSolver
par, prob, net = bench.build_problem()
s = BatchedOcpSolver(prob, net)
B, N, n = 4096, prob.N, int(os.environ.get('SMPC_STEPS', '40'))
x0 = bench.initial_states(s, prob, B, 0)
xg = np.repeat(x0[:, None, :], N + 1, axis=1); ug = np.zeros((B, N, 6)); p = np.zeros((B, N + 1, 5))
p[:, :, :3], p[:, :, 3], p[:, :, 4] = prob.ee_ref, par.alpha, 1.0
dev = torch.device('cuda:0')
t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
for streams in ('1', '2', '3', '4'):
    os.environ['SMPC_ROLLOUT_STREAMS'] = streams
    xd, xgd, ugd, pd = t(x0), t(xg), t(ug), t(p)
    s.rollout(xd, xgd, ugd, pd, 4); s.sync()                     # warm-up: workspaces of the workers
    xgd, ugd = t(xg), t(ug)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = s.rollout(xd, xgd, ugd, pd, n); s.sync()
    dt = time.perf_counter() - t0
    print(f'smpc_rollout_batch, B={B} N={N}, {streams} sub-batch stream(s): {1e3 * dt / n:.3f} ms per step '
          f'({B * n / dt / 1e6:.3f} M instance-steps/s), mean IPM iterations {out[3].double().mean().item():.2f}', flush=True)
