#!/usr/bin/env python3
"""Regenerates the frozen parity fixtures under tests/golden/.  BUILD-GENERATED DATA, not reference output.

The reference holds no golden vectors for this path and none of its native stack (acados / CasADi / l4casadi / adam) can
run here, so these vectors come from this repository's own CPU oracle (oracle/smpc_oracle.cpp) and, for the
network, from torch (an implementation independent of both the oracle and the HIP kernels).  Their purpose is to FREEZE
the target: tests/test_golden.py compares the live oracle (CPU suite) and the HIP engine (-m gpu) with these files, so a
change that touches kernel and oracle together can no longer move the expectation silently.  Re-running this script is a
deliberate act that shows up in `git diff tests/golden/`.

    python tests/golden/make_golden.py          # rewrites the .npz files

What is frozen is what is unique: inputs, linearisation records, and the solution of each (strictly convex) stage QP to the
test tolerance -- not the IPM's iteration path (iteration counts are stored for information and as a regression ceiling).

Cases:
  c0_{naive,st}   C0 plumbing: Z1-class 6-DoF, N = 10, the single start q0 = [-0.3, 0.8, -1.65, 0.658, 0, 0] of
                  guess_acados.py:103 (extended to 6 joints), ee_ref of config.yaml:73; one RTI solve from the constant guess,
                  then three closed-loop steps (shift, guessCorrection, nominal plant)
  c1_st           a 32-instance slice of C1: N = 30, Halton starts, controller 'st' (soft terminal safe-set row)
  c1_nls_zerovel  16 instances, NONLINEAR_LS cost + terminal zero velocity (the guess generator's OCP, guess_acados.py:33-34)
  c1_receding     16 instances, safe-set row switched per node through p[4]
  c4_fr7          8 instances of C4: 7-DoF, N = 40, sphere + floor rows, safe-set row on every node
  mlp_torch       the seeded 12-256-256-256-1 GELU(tanh) network on 64 inputs: value and input gradient from torch autograd
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from conftest import constant_guess, make_problem, make_problem_fr7, sample_instances   # noqa: E402
from oracle.oracle import Oracle                                                            # noqa: E402

EV_FIELDS = ('tau', 'M', 'dtau_dq', 'dtau_dv', 'ee', 'cost_grad_q', 'cost_hess_qq', 'row_val', 'row_grad', 'nn_val', 'nn_grad')


def solve_case(prob, net, x0, xg, ug, p, closed_loop_steps=0):
    o = Oracle(prob, (net.weights, net.biases) if net is not None else None)
    ev = o.eval_nodes(xg, ug, p)
    N = prob.N
    nodes = np.array(sorted({0, 1, N // 2, N - 1, N}))      # linearisation records of these nodes only (file size)
    out = {'x0': x0, 'xg': xg, 'ug': ug, 'p': p, 'ev_nodes': nodes}
    for f in EV_FIELDS:
        out['ev_' + f] = np.asarray(ev[f])[:, nodes]
    x, u, st, it = o.solve_batch(x0, xg, ug, p)
    out.update(x=x, u=u, status=st, qp_iter=it)
    # closed loop: provideControl (accept where status == 0), nominal plant, guessCorrection, solve again
    xs, xgs, ugs = x0, xg, ug
    for t in range(closed_loop_steps):
        acc = (st == 0).astype(np.int32)
        xgs, ugs, ua = o.provide_control(acc, x, u, xgs, ugs)
        xs, _ = o.plant_step(xs, ua)
        xgs = o.guess_correction(xgs, ugs)
        x, u, st, it = o.solve_batch(xs, xgs, ugs, p)
        out[f'cl{t}_x0'], out[f'cl{t}_xg'], out[f'cl{t}_ug'] = xs, xgs, ugs
        out[f'cl{t}_x'], out[f'cl{t}_u'], out[f'cl{t}_status'], out[f'cl{t}_qp_iter'] = x, u, st, it
    return out


def main():
    cases = {}
    # ---- C0
    for cont in ('naive', 'st'):
        par, prob, net = make_problem(cont, 'ext', N=10)
        x0 = np.array([[-0.3, 0.8, -1.65, 0.658, 0.0, 0.0] + [0.0] * 6])
        xg, ug, p = constant_guess(prob, x0)
        cases[f'c0_{cont}'] = solve_case(prob, net, x0, xg, ug, p, closed_loop_steps=3)
    # ---- C1 slices
    par, prob, net = make_problem('st', 'ext', N=30)
    x0 = sample_instances(prob, 32, seed=0)
    cases['c1_st'] = solve_case(prob, net, x0, *constant_guess(prob, x0), closed_loop_steps=2)
    par, prob, net = make_problem('zerovel', 'nls', N=30)
    x0 = sample_instances(prob, 16, seed=1, vel_scale=0.1)
    cases['c1_nls_zerovel'] = solve_case(prob, net, x0, *constant_guess(prob, x0))
    par, prob, net = make_problem('receding', 'ext', N=30)
    x0 = sample_instances(prob, 16, seed=2, vel_scale=0.1)
    xg, ug, p = constant_guess(prob, x0)
    p[:, 1:30, 4] = -1.0
    p[np.arange(16), 1 + (7 * np.arange(16)) % 29, 4] = 1.0
    cases['c1_receding'] = solve_case(prob, net, x0, xg, ug, p)
    # ---- C4 slice
    par, prob, net = make_problem_fr7(N=40)
    x0 = sample_instances(prob, 8, seed=3, vel_scale=0.1)
    cases['c4_fr7'] = solve_case(prob, net, x0, *constant_guess(prob, x0, ee_ref=prob.ee_ref))
    # ---- the network through torch (independent of oracle and engine)
    par, prob, net = make_problem('st', 'ext', N=10)
    rng = np.random.default_rng(0)
    s = rng.standard_normal((64, 12)).astype(np.float32)
    y, g = net.torch_value_and_grad(s)
    cases['mlp_torch'] = {'s': s, 'y': y.astype(np.float32), 'g': g.astype(np.float32),
                          'w_checksum': np.array([float(np.sum(np.abs(w), dtype=np.float64)) for w in net.weights])}
    for name, d in cases.items():
        path = os.path.join(HERE, name + '.npz')
        np.savez_compressed(path, **d)
        print(f'{name}: {os.path.getsize(path) / 1024:.0f} KiB',
              {k: v.tolist() for k, v in d.items() if k.endswith('status') or k.endswith('qp_iter')} if 'x0' in d and len(d['x0']) <= 1 else '')


if __name__ == '__main__':
    main()
