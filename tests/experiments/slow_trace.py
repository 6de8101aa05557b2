"""CPU experiment (oracle only): iteration trace of the slowest QP of a short C1 closed loop (SMPC_ORACLE_TRACE).
usage: python tests/experiments/slow_trace.py [B] [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from conftest import make_problem, sample_instances, constant_guess
from oracle.oracle import Oracle

B = int(sys.argv[1]) if len(sys.argv) > 1 else 384
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 14
par, prob, net = make_problem('st', 'ext', N=30)
x0 = sample_instances(prob, B, seed=0)
o = Oracle(prob, (net.weights, net.biases))
xg, ug, p = constant_guess(prob, x0)
x = x0.copy()
worst = []
for t in range(STEPS):
    xo, uo, st, it = o.solve_batch(x, xg, ug, p)
    order = np.argsort(-it)[:3]
    for b in order:
        worst.append((int(it[b]), t, int(b), x[b].copy(), xg[b].copy(), ug[b].copy(), p[b].copy()))
    print('step', t, 'mean', it.mean().round(2), 'max', it.max(), 'hist', np.bincount(it)[4:])
    xg, ug, ua = o.provide_control((st == 0).astype(np.int32), xo, uo, xg, ug)
    x, _ = o.plant_step(x, ua)
    xg = o.guess_correction(xg, ug)
worst.sort(key=lambda w: -w[0])
os.environ['SMPC_ORACLE_TRACE'] = '1'
for w in worst[:3]:
    print('=== iterations', w[0], 'step', w[1], 'instance', w[2], file=sys.stderr)
    o.solve_batch(w[3][None], w[4][None], w[5][None], w[6][None])
