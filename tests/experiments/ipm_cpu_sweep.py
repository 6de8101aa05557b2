"""CPU experiment (oracle only, not product): closed-loop C1 iteration statistics of the IPM under descriptor options.
usage: python tests/experiments/ipm_cpu_sweep.py [B] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from conftest import make_problem, make_problem_fr7, sample_instances, constant_guess
from oracle.oracle import Oracle

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 25


def closed_loop(prob, net, x0, ee_ref=None, steps=STEPS):
    o = Oracle(prob, (net.weights, net.biases))
    xg, ug, p = constant_guess(prob, x0, ee_ref=ee_ref)
    x = x0.copy()
    its, fails = [], 0
    for t in range(steps):
        xo, uo, st, it = o.solve_batch(x, xg, ug, p)
        its.append(it.copy())
        fails += int((st != 0).sum())
        xg, ug, ua = o.provide_control((st == 0).astype(np.int32), xo, uo, xg, ug)
        x, _ = o.plant_step(x, ua)
        xg = o.guess_correction(xg, ug)
    its = np.array(its)
    return its, fails, x


if __name__ == '__main__':
    par, prob, net = make_problem('st', 'ext', N=30)
    x0 = sample_instances(prob, B, seed=0)
    base = None
    for mu0, tol in [(1.0, 1e-8), (0.1, 1e-8), (0.01, 1e-8), (1.0, 1e-6), (10.0, 1e-8)]:
        prob.desc.qp_mu0, prob.desc.qp_tol = mu0, tol
        t = time.time()
        its, fails, x = closed_loop(prob, net, x0)
        if base is None:
            base = x
        print(f'mu0 {mu0:6g} tol {tol:g}: mean it {its.mean():.2f} (first step {its[0].mean():.2f}, later {its[5:].mean():.2f}) '
              f'max {its.max()} p99 {np.quantile(its, 0.99):.0f} fails {fails} |x - x_base| {np.abs(x - base).max():.2e} ({time.time() - t:.1f}s)')
