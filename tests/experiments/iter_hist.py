"""Experiment: IPM iteration counts per instance and step of the bench's closed loop (C1, 'st'): histogram, persistence of the
slow instances, and the critical path sum_t iters(b, t) of the slowest instances."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from safe_mpc_amd.solver import BatchedOcpSolver

par, prob, net = bench.build_problem()
s = BatchedOcpSolver(prob, net)
B, steps = 4096, int(os.environ.get('STEPS', '40'))
x0 = bench.initial_states(s, prob, B, 0)
N = prob.N
xg = np.repeat(x0[:, None, :], N + 1, axis=1); ug = np.zeros((B, N, 6)); p = np.zeros((B, N + 1, 5))
p[:, :, :3], p[:, :, 3], p[:, :, 4] = prob.ee_ref, par.alpha, 1.0
xt, ut, st, it, xg, ug = s.rollout(x0, xg, ug, p, steps)
it = np.asarray(it)            # [steps, B]
print('mean iterations per step:', np.round(it.mean(1), 2))
print('max  iterations per step:', it.max(1))
print('overall histogram (iterations: count):', {int(k): int(v) for k, v in zip(*np.unique(it, return_counts=True))})
tot = it.sum(0)
print('per-instance total over %d steps: mean %.1f  p50 %.0f  p90 %.0f  p99 %.0f  max %d' % (steps, tot.mean(), *np.percentile(tot, [50, 90, 99]), tot.max()))
print('sum over steps of the per-step max: %d   (the synchronous critical path, in iterations)' % it.max(1).sum())
for g in (3,):
    sub = np.array_split(np.arange(B), g)
    print('%d sub-batches: sum over steps of max within sub-batch:' % g, [int(it[:, ix].max(1).sum()) for ix in sub])
slow = it >= 9
print('instance-steps with >= 9 iterations: %.2f %%; instances ever >= 9: %d; mean run length of consecutive slow steps: %.2f' % (
    100 * slow.mean(), slow.any(0).sum(), np.mean([len(r) for b in range(B) for r in ''.join('1' if v else '0' for v in slow[:, b]).split('0') if r] or [0])))
