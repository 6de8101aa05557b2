"""First GPU bring-up check: component-by-component parity print-out (diagnostic, not a test)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from conftest import constant_guess, make_problem, sample_instances
from oracle.oracle import Oracle
from safe_mpc_amd.solver import BatchedOcpSolver

for controller in ['naive', 'st', 'constraint_everywhere']:
    par, prob, net = make_problem(controller, 'ext', N=30)
    s = BatchedOcpSolver(prob, net)
    o = Oracle(prob, (net.weights, net.biases))
    B = 64
    x0 = sample_instances(prob, B, seed=1, vel_scale=0.3)
    xg, ug, p = constant_guess(prob, x0)
    rng = np.random.default_rng(0)
    ug += rng.uniform(-3, 3, ug.shape)
    a, b = s.eval_nodes(xg, ug, p), o.eval_nodes(xg, ug, p)
    for f in a.dtype.names:
        d = np.abs(a[f] - b[f]).max()
        print(f'{controller:24s} {f:14s} maxabs diff {d:.3e}  scale {np.abs(b[f]).max():.3e}', flush=True)
    t = time.time()
    xa, ua, sa, ia = s.solve(x0, xg, ug, p)
    tg = time.time() - t
    t = time.time()
    xb, ub, sb, ib = o.solve_batch(x0, xg, ug, p)
    tc = time.time() - t
    print(f'{controller}: status gpu {np.bincount(sa, minlength=5)} cpu {np.bincount(sb, minlength=5)}; iters gpu {ia[:8]} cpu {ib[:8]}')
    print(f'   max|du| {np.abs(ua - ub).max():.3e} max|dx| {np.abs(xa - xb).max():.3e}  |u| {np.abs(ub).max():.2f}  t_gpu {tg:.3f}s t_cpu {tc:.3f}s', flush=True)

# throughput smoke at the headline size
par, prob, net = make_problem('st', 'ext', N=30)
s = BatchedOcpSolver(prob, net)
B = 4096
x0 = sample_instances(prob, B, seed=2)
xg, ug, p = constant_guess(prob, x0)
s.enable_timing(True)
for i in range(3):
    t = time.time()
    x, u, st, it = s.solve(x0, xg, ug, p)
    print(f'B={B} host-path solve {time.time() - t:.4f}s  status {np.bincount(st, minlength=5)} iters mean {it.mean():.2f} max {it.max()}  timing {s.timing()}', flush=True)
