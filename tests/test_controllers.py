"""Policy automata: pure state-machine tests with scripted solver statuses, plus a short
closed loop through the whole policy layer on the CPU test double."""
import numpy as np
import pytest

from conftest import make_problem, sample_instances
from fake_solver import OracleSolver
from safe_mpc_amd import controller as C


def _make(name, B=4, N=6, **over):
    par, prob, net = make_problem(name if name != 'stwa' else 'st', N=N, **over)
    cls = C.CONTROLLERS[name]
    ctrl = cls.__new__(cls)
    # build the controller around the test double (the real constructor would open the GPU engine)
    solver_prob = C.OcpProblem(par, cls.cont_name, 'ext', N=N)
    solver_prob.set_normalisation(net.mean, net.std)
    C.AbstractController.__init__(ctrl, par, B, 'ext', N, solver=OracleSolver(solver_prob, net), net=net)
    x0 = sample_instances(prob, B, seed=7)
    ctrl.setGuess(np.repeat(x0[:, None, :], N + 1, axis=1), np.zeros((B, N, 6)))
    return par, ctrl, x0


def test_naive_fail_counter_and_rollback():
    """controller.py:274-284 + 169-184: a failed solve keeps the shifted OLD guess and applies u_guess[0]."""
    par, c, x0 = _make('naive')
    c.u_guess[:] = np.arange(c.N)[None, :, None] + 1.0          # recognisable warm start
    c.ocp_solver.scripted_status = [np.array([0, 4, 0, 4])]
    u, abort = c.step(x0)
    assert not abort.any()
    assert c.fails.tolist() == [0, 1, 0, 1]
    assert np.allclose(u[1], 1.0) and np.allclose(u[3], 1.0)            # old u_guess[0]
    assert np.allclose(c.u_guess[1, :, 0], [2, 3, 4, 5, 6, 6])          # rolled, last duplicated
    assert not np.allclose(c.u_guess[0, :, 0], [2, 3, 4, 5, 6, 6])      # accepted instances took the new solution
    c.ocp_solver.scripted_status = [np.array([0, 0, 4, 4])]
    c.step(c.x_guess[:, 0])
    assert c.fails.tolist() == [0, 0, 1, 2] and c.current_step.tolist() == [2, 2, 2, 2]


def test_stwa_abort_after_n_minus_one_failures():
    """controller.py:375-388: x_viable is taken at the first failure, abort is raised when fails == N-1, and the aborting
    instance neither shifts its guess nor advances its step counter."""
    par, c, x0 = _make('htwa', B=2, N=4)
    x = x0
    viable0 = None
    for k in range(5):
        c.ocp_solver.scripted_status = [np.array([0, 4])]
        guess_before = c.x_guess[1].copy()
        u, abort = c.step(x)
        if k == 0:
            viable0 = c.x_viable[1].copy()
        if k < 3:
            assert not abort.any() and c.fails[1] == k + 1
        elif k == 3:
            assert abort.tolist() == [False, True] and c.fails[1] == 3        # N-1 failures -> abort, counter frozen
            # guessCorrection ran, but no shift: node 0 of the guess is unchanged
            assert np.array_equal(c.x_guess[1, 0], guess_before[0])
            assert c.current_step.tolist() == [4, 3]
        x = c.x_guess[:, 0].copy()
    assert np.array_equal(c.x_viable[1], viable0)                              # captured once, at the first failure
    assert c.fails[0] == 0


def test_receding_index_automaton():
    """controller.py:448-498: flags follow r; r decreases every step, jumps to the last safe node, abort at r == 0."""
    par, c, x0 = _make('receding', B=3, N=5)
    assert c.r.tolist() == [5, 5, 5]
    # make the safe-set test scripted: instance 0 never safe, instance 1 safe everywhere, instance 2 safe up to node 3
    safe = np.zeros((3, 6), bool); safe[1] = True; safe[2, :4] = True
    c.checkSafeConstraints = lambda x: safe
    c.checkStateConstraints = lambda x: np.ones(3, bool)
    seen_flags = []
    orig_solve = c.solve
    def spy(x):
        seen_flags.append(c.p[:, :, 4].copy())
        return orig_solve(x)
    c.solve = spy
    x = x0
    rs, aborts = [], []
    for k in range(6):
        c.ocp_solver.scripted_status = [np.zeros(3, np.int32)]
        u, abort = c.step(x)
        rs.append(c.r.tolist()); aborts.append(abort.tolist())
        x = c.x_guess[:, 0].copy()
    # instance 0: r = 4,3,2,1 then hits 0 -> abort, r reset to N;  instance 1: every step jumps back to N-1
    assert [r[0] for r in rs] == [4, 3, 2, 1, 5, 4]
    assert [a[0] for a in aborts] == [False, False, False, False, True, False]
    assert all(r[1] == 4 for r in rs) and not any(a[1] for a in aborts)
    # instance 2: safe nodes 0..3 -> candidates i in r+2..N that are safe: only while r+2 <= 3
    assert [r[2] for r in rs][:3] == [4, 3, 2]
    # flags of the first solve: all running nodes off (r = N), terminal on
    assert np.all(seen_flags[0][:, 1:5] == -1) and np.all(seen_flags[0][:, 5] == 1)
    # second solve of instance 0: node r = 4 switched on
    assert seen_flags[1][0, 4] == 1 and np.all(seen_flags[1][0, 1:4] == -1)


@pytest.mark.parametrize('name', ['naive', 'st', 'htwa', 'receding', 'real_receding', 'constraint_everywhere'])
def test_closed_loop_on_cpu_double(name):
    """A few closed-loop steps of every registered policy: finite controls, states stay inside the widened box."""
    par, c, x0 = _make(name, B=3, N=8)
    x = x0.copy()
    for k in range(4):
        u, abort = c.step(x)
        assert np.all(np.isfinite(u))
        x, _ = c.ocp_solver.plant_step(x, u)
    assert np.all(x >= c.problem.x_min - par.tol_x) and np.all(x <= c.problem.x_max + par.tol_x)
    assert c.x_guess.shape == (3, 9, 12) and np.all(c.current_step <= 4)


def test_get_controller_registry():
    assert set(C.CONTROLLERS) == {'naive', 'zerovel', 'st', 'stwa', 'htwa', 'receding', 'real_receding',
                                  'constraint_everywhere'}
    with pytest.raises(ValueError):
        C.get_controller('nope', None, 1)


def test_real_receding_tube_bounds_reach_the_solver():
    """controller.py:530-536: node r of every instance is boxed to x_guess[r+1] +- 1e-3; the other nodes get the model box."""
    par, c, x0 = _make('real_receding', B=3, N=6)
    seen = {}
    orig = c.ocp_solver.set_instance_bounds
    c.ocp_solver.set_instance_bounds = lambda lo, hi: (seen.update(lo=lo.copy(), hi=hi.copy()), orig(lo, hi))[1]
    c.checkSafeConstraints = lambda x: np.zeros((3, 7), bool)
    c.step(x0)                                   # r = N: no tube yet
    assert np.allclose(seen['lo'][:, 1:6], c.problem.x_min) and np.allclose(seen['hi'][:, 1:6], c.problem.x_max)
    assert c.r.tolist() == [5, 5, 5]
    guess_before = c.x_guess.copy()
    u, ab = c.step(c.x_guess[:, 0].copy())       # r = 5 < N: node 5 boxed around x_guess[6]
    assert np.allclose(seen['lo'][:, 5], guess_before[:, 6] - 1e-3) and np.allclose(seen['hi'][:, 5], guess_before[:, 6] + 1e-3)
    # the solution honours the tube (status 0 instances)
    ok = c.last_status == 0
    assert ok.any()
    assert np.all(np.abs(c.x_temp[ok, 5] - guess_before[ok, 6]) <= 1e-3 + 1e-6)
