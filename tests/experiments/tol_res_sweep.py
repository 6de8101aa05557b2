"""CPU experiment (oracle only, not product): what the exit tolerance on the LINEAR residuals costs.
include/smpc.h notes that HPIPM's BALANCE mode (config.yaml:15) asks 1e-6 of the stationarity residual and 1e-8 of the rest; the
engine asks 1e-8 of everything.  Closed loop of the oracle over 100 steps for qp_tol_res in {1e-8 (default), 1e-6, 1e-4}:
mean IPM iterations, mean of the per-step maxima (what sets a launch's duration), and the deviation of the closed-loop
trajectory from the default run's.
usage: python tests/experiments/tol_res_sweep.py [controller] [B] [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from conftest import make_problem, sample_instances, constant_guess
from oracle.oracle import Oracle

CONT = sys.argv[1] if len(sys.argv) > 1 else 'st'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 100
N = 30


def run(tol_res, tol=1e-8):
    par, prob, net = make_problem(CONT, 'ext', N=N, qp_tol=tol, qp_tol_res=tol_res)
    o = Oracle(prob, (net.weights, net.biases))
    x = sample_instances(prob, B, seed=0)
    xg, ug, p = constant_guess(prob, x)
    fails = np.zeros(B, int)
    X, IT, nfail = [x.copy()], [], 0
    for j in range(STEPS):
        xg = o.guess_correction(xg, ug)
        xt, ut, st, it = o.solve_batch(x, xg, ug, p)
        IT.append(it.copy())
        nfail += int((st != 0).sum())
        fails = np.where(st == 0, 0, fails + 1)
        xg, ug, u = o.provide_control((fails == 0).astype(np.int32), xt, ut, xg, ug)
        x, _ = o.plant_step(x, u)
        X.append(x.copy())
    return np.array(X), np.array(IT), nfail


base = None
print(f'controller {CONT}, B = {B}, {STEPS} closed-loop steps, N = {N}')
print('qp_tol   qp_tol_res | mean it | mean of per-step max | steps 10..: mean it | failed solves | max |x - x_default| over the loop')
for tol, tr in ((1e-8, 1e-8), (1e-8, 1e-6), (1e-8, 1e-4), (1e-7, 1e-6)):
    X, IT, nf = run(tr, tol)
    if base is None:
        base = X
    print(f'{tol:.0e}   {tr:.0e}      | {IT.mean():6.3f}  | {IT.max(1).mean():6.2f}               | {IT[10:].mean():6.3f}            | {nf:5d}         | {np.abs(X - base).max():.2e}',
          flush=True)
