import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from safe_mpc_amd.solver import BatchedOcpSolver
par, prob, net = bench.build_problem()
s = BatchedOcpSolver(prob, net)
B = 4096
x0 = bench.initial_states(s, prob, B, 0)
N = prob.N
xg = np.repeat(x0[:, None, :], N + 1, axis=1); ug = np.zeros((B, N, 6)); p = np.zeros((B, N + 1, 5))
p[:, :, :3], p[:, :, 3], p[:, :, 4] = prob.ee_ref, par.alpha, 1.0
x = x0
saved = False
for i in range(14):
    xo, uo, st, it = s.solve(x, xg, ug, p)
    print('step', i, 'iters mean %.2f' % it.mean(), 'hist', np.bincount(np.minimum(it, 30))[1:].tolist(), 'n>=30', (it >= 30).sum(), 'max', it.max(), 'fails', (st != 0).sum(), flush=True)
    if (it.max() >= 50 and not saved) or (i == 13 and not saved):
        j = int(np.argmax(it))
        np.savez(os.path.join(ROOT, 'gpurun_out', 'straggler.npz'), x=x[j], xg=xg[j], ug=ug[j], p=p[j], it=it[j], step=i)
        saved = True
    xg, ug, ua = s.provide_control((st == 0).astype(np.int32), xo, uo, xg, ug)
    x, _ = s.plant_step(x, ua)
    xg = s.guess_correction(xg, ug)
