"""Pins the oracle's QP layer (a11, a12, a17): Riccati primal-dual IPM vs an independent dense solver, KKT
conditions, and the structural properties of one RTI step."""
import numpy as np
import pytest

from conftest import constant_guess, make_problem, sample_instances
from oracle.oracle import Oracle
from qp_ref import condense, solve_condensed


def _setup(controller, N, cost='ext'):
    par, prob, net = make_problem(controller, cost, N=N)
    o = Oracle(prob, (net.weights, net.biases))
    return par, prob, net, o


@pytest.mark.parametrize('controller,N', [('naive', 8), ('zerovel', 8), ('st', 6), ('htwa', 6),
                                          ('constraint_everywhere', 6)])
def test_ipm_matches_independent_dense_solver(controller, N):
    par, prob, net, o = _setup(controller, N)
    x0 = sample_instances(prob, 3, seed=1, vel_scale=0.2)
    xg, ug, p = constant_guess(prob, x0)
    rng = np.random.default_rng(0)
    ug += rng.uniform(-2, 2, ug.shape)                       # non-trivial warm start with dynamics defects
    xg[:, 1:] += 0.01 * rng.standard_normal(xg[:, 1:].shape)
    xo, uo, st, it = o.solve_batch(x0 + 0.002, xg, ug, p)    # x0 != x_guess[0] as in a closed loop
    for b in range(3):
        qp = o.build_qp(x0[b] + 0.002, xg[b], ug[b], p[b])
        cq = condense(qp, N, 6, par.dt)
        v, s, lam, nit = solve_condensed(cq)
        Phi, c = cq['Phi'], cq['c']
        assert nit < 150
        if st[b] != 0:
            continue
        du = (uo[b] - ug[b]).reshape(-1)
        scale = 1.0 + np.abs(v).max()
        assert np.abs(du - v).max() < 2e-6 * scale, (controller, b, np.abs(du - v).max())
        dxN = xo[b, N] - xg[b, N]
        assert np.allclose(dxN, Phi[N] @ v + c[N], atol=1e-7)
    assert (st == 0).sum() >= 2


def test_rti_step_structure():
    """x_out[0] = x0, the output satisfies the double integrator exactly, status 0, few iterations."""
    par, prob, net, o = _setup('st', 30)
    x0 = sample_instances(prob, 16, seed=2)
    xg, ug, p = constant_guess(prob, x0)
    xo, uo, st, it, res = o.solve_batch(x0, xg, ug, p, with_res=True)
    assert np.all(st == 0) and it.max() <= 25
    assert np.allclose(xo[:, 0], x0, atol=1e-12)
    dt = par.dt
    qn = xo[:, :-1, :6] + dt * xo[:, :-1, 6:] + 0.5 * dt * dt * uo
    vn = xo[:, :-1, 6:] + dt * uo
    assert np.allclose(xo[:, 1:, :6], qn, atol=1e-10) and np.allclose(xo[:, 1:, 6:], vn, atol=1e-10)
    assert res.max() <= 1e-8
    # box constraints hold on the solution, linearised torque rows hold
    assert np.all(xo[:, 1:] >= prob.lbx - 1e-7) and np.all(xo[:, 1:] <= prob.ubx + 1e-7)
    ev = o.eval_nodes(xg, ug, p)
    for b in range(4):
        for k in range(prob.N):
            e = ev[b, k]
            M = e['M'][:36].reshape(6, 6); dq = e['dtau_dq'][:36].reshape(6, 6); dv = e['dtau_dv'][:36].reshape(6, 6)
            dx = xo[b, k] - xg[b, k]
            tau_lin = e['tau'][:6] + M @ (uo[b, k] - ug[b, k]) + dq @ dx[:6] + dv @ dx[6:]
            assert np.all(np.abs(tau_lin) <= prob.tau_max + 1e-6)


def test_fixed_point_known_optimum():
    """Known answer: target = current EE position, robot at rest.  Staying put with u = 0 has zero cost and is
    feasible, so it is the optimum; one RTI step from that guess must return it unchanged."""
    par, prob, net, o = _setup('naive', 12)
    x0 = sample_instances(prob, 4, seed=3)
    ee = np.array([o.points(x[:6])[prob.desc.ee_point] for x in x0])
    xg, ug, p = constant_guess(prob, x0)
    p[:, :, :3] = ee[:, None, :]
    xo, uo, st, it = o.solve_batch(x0, xg, ug, p)
    assert np.all(st == 0)
    assert np.abs(xo - xg).max() < 1e-7 and np.abs(uo).max() < 1e-5


def test_infeasible_hard_terminal_set_reports_failure():
    """HTWA with an unreachable terminal set (alpha = 100 -> g = -|v| - eps < 0 unless v = 0 ... and a bias of -50)."""
    par, prob, net = make_problem('htwa', N=10)
    biases = [b.copy() for b in net.biases]
    biases[-1][:] = -500.0                                    # nn(s) << 0 everywhere: hard row cannot be met
    o = Oracle(prob, (net.weights, biases))
    x0 = sample_instances(prob, 2, seed=4, vel_scale=0.3)
    xg, ug, p = constant_guess(prob, x0)
    xo, uo, st, it = o.solve_batch(x0, xg, ug, p)
    # either the IPM breaks down (status 4) or it stops at the iteration cap, which acados' RTI reports as success
    assert np.all((st == 4) | (it >= prob.desc.qp_max_iter))
    # the soft version of the same problem is always solvable
    par2, prob2, _ = make_problem('st', N=10)
    o2 = Oracle(prob2, (net.weights, biases))
    xo, uo, st, it = o2.solve_batch(x0, xg, ug, p)
    assert np.all(st == 0) and it.max() < 150


def test_flag_switches_nn_row_off():
    """p[4] <= 0 removes the safe-set row (utils.py:207-210): 'st' with flag -1 equals 'naive'."""
    par, prob, net, o = _setup('st', 10)
    par_n, prob_n, _, o_n = _setup('naive', 10)
    x0 = sample_instances(prob, 3, seed=5, vel_scale=0.2)
    xg, ug, p = constant_guess(prob, x0, flag=-1.0)
    a = o.solve_batch(x0, xg, ug, p)
    b = o_n.solve_batch(x0, xg, ug, p)
    assert np.array_equal(a[2], b[2]) and np.allclose(a[0], b[0], atol=1e-12) and np.allclose(a[1], b[1], atol=1e-12)
