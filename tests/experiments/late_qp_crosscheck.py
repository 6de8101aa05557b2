"""CPU experiment (oracle only, not product): how well is the solution of a LATE closed-loop QP pinned?  
The oracle's closed loop is run for `steps` steps; at the chosen steps every instance's stage QP is exported and solved by the
independent dense log-barrier solver of tests/qp_ref.py (exit 1e-11); reported per step: the gap between the IPM's step and
the dense optimum in the controls, in the objective value and the IPM solution's violation of the dense QP's rows.
usage: python tests/experiments/late_qp_crosscheck.py [controller] [B] [steps] [qp_tol] [qp_tol_res]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from conftest import make_problem, sample_instances, constant_guess
from oracle.oracle import Oracle
from qp_ref import condense, solve_condensed

CONT = sys.argv[1] if len(sys.argv) > 1 else 'st'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 101
over = {}
if len(sys.argv) > 4: over['qp_tol'] = float(sys.argv[4])
if len(sys.argv) > 5: over['qp_tol_res'] = float(sys.argv[5])
N = 30
par, prob, net = make_problem(CONT, 'ext', N=N, **over)
o = Oracle(prob, (net.weights, net.biases))
x0 = sample_instances(prob, B, seed=0)
xg, ug, p = constant_guess(prob, x0)
x = x0.copy()
fails = np.zeros(B, int)


def qp_objective(cq, v):
    return 0.5 * v @ cq['H'] @ v + cq['g'] @ v if cq['Z'] is None else None


for j in range(STEPS):
    xg = o.guess_correction(xg, ug)
    xt, ut, st, it = o.solve_batch(x, xg, ug, p)
    if j in (0, 40, 100, STEPS - 1):
        gaps, objg, viol, nits = [], [], [], []
        for b in range(B):
            if st[b] != 0:
                continue
            cq = condense(o.build_qp(x[b], xg[b], ug[b], p[b]), N, 6, par.dt)
            v, s_, lam, nit = solve_condensed(cq)
            du = (ut[b] - ug[b]).reshape(-1)
            gaps.append(np.abs(du - v).max() / (1 + np.abs(v).max()))
            if cq['Z'] is None:
                f = lambda w: 0.5 * w @ cq['H'] @ w + cq['g'] @ w
                # soft rows: add the L1 penalty of the violation
                def full(w):
                    r = cq['G'] @ w - cq['h']
                    sw = cq['soft_w']
                    return f(w) + np.sum(np.where(sw >= 0, sw * np.maximum(r, 0), 0.0))
                objg.append((full(du) - full(v)) / (1 + abs(full(v))))
                r = cq['G'] @ du - cq['h']
                viol.append(np.max(np.where(cq['soft_w'] >= 0, 0.0, r)))
            nits.append(nit)
        gaps = np.array(gaps)
        print(f'step {j:3d}: {len(gaps)} QPs, IPM iterations mean {it.mean():.2f} max {it.max()};  |du - v*|/(1+|v*|): median {np.median(gaps):.2e} '
              f'max {gaps.max():.2e};  objective gap max {np.max(objg):.2e};  hard-row violation max {np.max(viol):.2e};  dense its max {max(nits)}', flush=True)
    ok = st == 0
    fails = np.where(ok, 0, fails + 1)
    xg, ug, u = o.provide_control((fails == 0).astype(np.int32), xt, ut, xg, ug)
    x, _ = o.plant_step(x, u)
