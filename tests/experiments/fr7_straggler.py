"""CPU experiment (oracle only): iteration statistics of the C4 workload's first step and a trace of its slowest instance."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from conftest import make_problem_fr7, sample_instances, constant_guess
from oracle.oracle import Oracle

par, prob, net = make_problem_fr7(N=40)
cache = '/tmp/fr7_x0.npy'
if os.path.exists(cache):
    x0 = np.load(cache)
else:
    x0 = sample_instances(prob, 4096, seed=3, vel_scale=0.0)
    np.save(cache, x0)
o = Oracle(prob, (net.weights, net.biases))
xg, ug, p = constant_guess(prob, x0, ee_ref=prob.ee_ref)
t = time.time()
xo, uo, st, it = o.solve_batch(x0, xg, ug, p)
print('first step: mean it %.2f max %d p99 %.0f fails %d (%.1fs)' % (it.mean(), it.max(), np.quantile(it, 0.99), (st != 0).sum(), time.time() - t))
worst = np.argsort(-it)[:8]
print('worst', worst, it[worst])
if len(sys.argv) > 1:
    i = int(sys.argv[1]) if sys.argv[1] != 'worst' else int(worst[0])
    os.environ['SMPC_ORACLE_TRACE'] = '1'
    o.solve_batch(x0[i:i + 1], xg[i:i + 1], ug[i:i + 1], p[i:i + 1])
