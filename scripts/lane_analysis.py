"""Would sub-batches sorted by difficulty shorten the closed loop?    A sub-batch launch of k_qp_ipm lasts as
long as its slowest instance iterates, so the loop's critical path is  sum over steps of (max iterations within a sub-batch).  This takes
the per-instance iteration counts of the bench's closed loop (C1, 110 steps) and compares the three contiguous sub-batches of bench.py with
partitions by the iteration counts of the first steps (a small "slow lane" + fast lanes).  usage: python scripts/lane_analysis.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from safe_mpc_amd.solver import BatchedOcpSolver

par, prob, net = bench.build_problem()
s = BatchedOcpSolver(prob, net)
B, steps, warm = 4096, 110, 10
x0 = bench.initial_states(s, prob, B, 0)
N = prob.N
xg = np.repeat(x0[:, None, :], N + 1, axis=1); ug = np.zeros((B, N, 6)); p = np.zeros((B, N + 1, 5))
p[:, :, :3], p[:, :, 3], p[:, :, 4] = prob.ee_ref, par.alpha, 1.0
xt, ut, st, it, xg, ug = s.rollout(x0, xg, ug, p, steps)
it = np.asarray(it)[:, :]                    # [steps, B]
T = it[warm:]                                # the timed window
print('mean iterations %.2f; per-step max over the whole batch: mean %.2f' % (T.mean(), T.max(1).mean()))
contig = np.array_split(np.arange(B), 3)
base = [T[:, ix].max(1).sum() for ix in contig]
print('three contiguous sub-batches (bench.py): sum over steps of the per-step max = %s -> critical path %d iterations' % (base, max(base)))
score = it[3:warm].mean(0)                   # difficulty seen during the warm-up
order = np.argsort(-score)
for n_slow in (128, 256, 512, 1024):
    for n_fast in (2, 3):
        lanes = [order[:n_slow]] + list(np.array_split(order[n_slow:], n_fast))
        cp = [T[:, ix].max(1).sum() for ix in lanes]
        print('slow lane of %4d + %d fast lanes: per-lane critical paths %s -> %d (%.0f %% of the contiguous partition)' % (
            n_slow, n_fast, cp, max(cp), 100.0 * max(cp) / max(base)))
# oracle partition (hindsight): by the mean over the timed window itself
order2 = np.argsort(-T.mean(0))
for n_slow in (256, 512):
    lanes = [order2[:n_slow]] + list(np.array_split(order2[n_slow:], 3))
    cp = [T[:, ix].max(1).sum() for ix in lanes]
    print('hindsight partition, slow lane of %4d + 3 fast lanes: %s -> %d (%.0f %%)' % (n_slow, cp, max(cp), 100.0 * max(cp) / max(base)))
print('slowest single instance over the window: %d iterations' % T.sum(0).max())
