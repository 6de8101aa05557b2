"""Frozen parity fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py -- build-generated, see its header).

CPU suite: the live oracle must still reproduce the frozen vectors (a co-edit of kernel and oracle cannot move the target).
-m gpu:    the HIP engine, through the C ABI, against the same frozen vectors.

Tolerances: linearisation records 1e-9 relative (fp32 network rows 2e-5 / 2e-4); QP solution 1e-6 (1 + |u|_inf) without
the network, 1e-4 (1 + |u|_inf) with it, 2e-5 for the zero-width terminal velocity box (as tests/test_gpu_parity.py).
Iteration counts are a regression ceiling (frozen + 2), not a parity quantity: the QP solution is unique, the path is not.
"""
import os

import numpy as np
import pytest

from conftest import make_problem, make_problem_fr7

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

# case -> (problem factory, solution tolerance relative to 1 + |u|_inf, state tolerance)
CASES = {
    'c0_naive': (lambda: make_problem('naive', 'ext', N=10), 1e-6, 1e-6),
    'c0_st': (lambda: make_problem('st', 'ext', N=10), 1e-4, 1e-4),
    'c1_st': (lambda: make_problem('st', 'ext', N=30), 1e-4, 1e-4),
    'c1_nls_zerovel': (lambda: make_problem('zerovel', 'nls', N=30), 2e-5, 2e-5),
    'c1_receding': (lambda: make_problem('receding', 'ext', N=30), 1e-4, 1e-4),
    'c4_fr7': (lambda: make_problem_fr7(N=40), 1e-4, 1e-4),
}
EV_TIGHT = ('tau', 'M', 'dtau_dq', 'dtau_dv', 'ee', 'cost_grad_q', 'cost_hess_qq', 'row_val', 'row_grad')


def _load(name):
    return dict(np.load(os.path.join(GOLD, name + '.npz')))


def _check_solver(solver_like, name, eval_nodes, solve):
    make, tol_u, tol_x = CASES[name]
    g = _load(name)
    ev = eval_nodes(g['xg'], g['ug'], g['p'])
    nodes = g['ev_nodes']
    for f in EV_TIGHT:
        a, b = np.asarray(ev[f])[:, nodes], g['ev_' + f]
        assert np.abs(a - b).max() <= 1e-9 * (1e-12 + np.abs(b).max()) + 1e-13, (name, f)
    a, b = np.asarray(ev['nn_val'])[:, nodes], g['ev_nn_val']
    assert np.abs(a - b).max() <= 2e-5 * (1e-12 + np.abs(b).max()) + 1e-12, (name, 'nn_val')
    a, b = np.asarray(ev['nn_grad'])[:, nodes], g['ev_nn_grad']
    assert np.abs(a - b).max() <= 2e-4 * (1e-12 + np.abs(b).max()) + 1e-12, (name, 'nn_grad')
    steps = [('', '')] + [(f'cl{t}_', f'cl{t}_') for t in range(8) if f'cl{t}_x0' in g]
    for pre, _ in steps:
        x0, xg, ug = g[pre + 'x0'], g[pre + 'xg'], g[pre + 'ug']
        x, u, st, it = solve(x0, xg, ug, g['p'])
        assert np.array_equal(np.asarray(st), g[pre + 'status']), (name, pre)
        ok = g[pre + 'status'] == 0
        assert ok.sum() >= len(ok) - 2
        ub = g[pre + 'u'][ok]
        assert np.abs(np.asarray(u)[ok] - ub).max() < tol_u * (1 + np.abs(ub).max()), (name, pre)
        assert np.abs(np.asarray(x)[ok] - g[pre + 'x'][ok]).max() < tol_x, (name, pre)
        assert np.all(np.asarray(it)[ok] <= g[pre + 'qp_iter'][ok] + 2), (name, pre, np.asarray(it)[ok], g[pre + 'qp_iter'][ok])


@pytest.mark.parametrize('name', sorted(CASES))
def test_oracle_reproduces_frozen_vectors(name):
    from oracle.oracle import Oracle
    par, prob, net = CASES[name][0]()
    o = Oracle(prob, (net.weights, net.biases))
    _check_solver(o, name, o.eval_nodes, o.solve_batch)


def test_c0_start_is_the_reference_script_start():
    """guess_acados.py:103 (q0 extended to six joints), ee_ref of config.yaml:73, N = 10, one instance."""
    g = _load('c0_naive')
    assert g['x0'].shape == (1, 12)
    assert np.array_equal(g['x0'][0], [-0.3, 0.8, -1.65, 0.658, 0.0, 0.0] + [0.0] * 6)
    assert np.array_equal(g['p'][0, 0, :3], [0.7, 0.17, 0.13]) and g['xg'].shape == (1, 11, 12)


def test_network_matches_frozen_torch_autograd():
    """The fp32 MLP of the oracle against torch.autograd values frozen by make_golden.py (safe_set.py:26-43, parser.py:99)."""
    from oracle.oracle import Oracle
    par, prob, net = make_problem('st', 'ext', N=10)
    g = _load('mlp_torch')
    chk = np.array([float(np.sum(np.abs(w), dtype=np.float64)) for w in net.weights])
    assert np.allclose(chk, g['w_checksum'], rtol=1e-6), 'the seeded synthetic network changed'
    o = Oracle(prob, (net.weights, net.biases))
    for s, y, gr in zip(g['s'], g['y'], g['g']):
        yo, go = o.mlp(s)
        assert abs(yo - y) < 2e-5 * (1 + abs(y)) and np.abs(go - gr).max() < 2e-5 * (1 + np.abs(gr).max())


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(CASES))
def test_engine_reproduces_frozen_vectors(name):
    from safe_mpc_amd.solver import BatchedOcpSolver
    par, prob, net = CASES[name][0]()
    s = BatchedOcpSolver(prob, net)
    _check_solver(s, name, s.eval_nodes, s.solve)
