"""GPU experiment: where the host thread spends its time in a device-resident policy loop (per step, both groups).
usage: python tests/experiments/policy_host_time.py [controller] [steps]"""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from safe_mpc_amd import closed_loop as cl
from safe_mpc_amd.solver import BatchedOcpSolver

T = collections.defaultdict(float)


def timed(cls, name, label=None):
    f = getattr(cls, name)
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            T[label or name] += time.perf_counter() - t
    setattr(cls, name, w)


timed(cl._Group, '_run_half')
timed(cl._Group, '_apply_inflight')
timed(cl._Group, 'handle_aborts')
_item = torch.Tensor.item
def item(self):
    t = time.perf_counter()
    try:
        return _item(self)
    finally:
        T['item (sync wait)'] += time.perf_counter() - t
torch.Tensor.item = item

name = sys.argv[1] if len(sys.argv) > 1 else 'receding'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 80
par, prob, net = bench.build_problem()
par.back_hor = 30
B, N = 4096, prob.N
s = BatchedOcpSolver(prob, net)
x0 = bench.initial_states(s, prob, B, 0)
xg = np.repeat(x0[:, None, :], N + 1, axis=1)
ug = np.zeros((B, N, prob.nu))
tm = {}
res = cl.run_mpc(par, name, xg, ug, n_steps=steps, on_device=True, timing=tm)
print(f"{name}: {tm['ms_per_step']:.3f} ms/step; host seconds over {steps} steps:")
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print(f'  {k:24s} {1e3 * v / steps:8.3f} ms/step')
