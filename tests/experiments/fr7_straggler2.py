"""CPU experiment (oracle only): tests/experiments/fr7_bench.py's workload (vel_scale 0.1), slowest instance traced."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from conftest import make_problem_fr7, sample_instances, constant_guess
from oracle.oracle import Oracle
par, prob, net = make_problem_fr7(N=40)
cache = '/tmp/fr7_x0_v01.npy'
if os.path.exists(cache):
    x0 = np.load(cache)
else:
    x0 = sample_instances(prob, 4096, seed=3, vel_scale=0.1); np.save(cache, x0)
o = Oracle(prob, (net.weights, net.biases))
xg, ug, p = constant_guess(prob, x0, ee_ref=prob.ee_ref)
xo, uo, st, it = o.solve_batch(x0, xg, ug, p)
print('first step: mean it %.2f max %d p99 %.0f fails %d' % (it.mean(), it.max(), np.quantile(it, 0.99), (st != 0).sum()))
worst = np.argsort(-it)[:6]; print('worst', worst, it[worst])
if len(sys.argv) > 1:
    i = int(worst[0]); os.environ['SMPC_ORACLE_TRACE'] = '1'
    o.solve_batch(x0[i:i + 1], xg[i:i + 1], ug[i:i + 1], p[i:i + 1])
    ev = o.eval_nodes(xg[i:i+1], ug[i:i+1], p[i:i+1])
    print('row_val node1', ev[0,1]['row_val'][:4], 'lb', prob.row_lb, 'nn_val', ev[0,1]['nn_val'])
