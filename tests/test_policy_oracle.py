"""SURVEY 8 row a14 (+ f1): the product's two statements of the policy layer -- the batched numpy mirror
(safe_mpc_amd/controller.py, closed_loop.py; CPU here) and the engine kernels (kernels_policy.hpp; tests/test_gpu_parity.py) --
against oracle/policy_oracle.py, an independent scalar restatement of /root/reference/src/safe_mpc/controller.py:274-284,
369-388, 448-498, 524-565, 651-661 and /root/reference/scripts/mpc.py:118-287.  Statuses, safe-set verdicts and backup
outcomes are scripted so that every branch is taken: rejection, abort after N - 1 failures, abort_flag = False, the receding
index moving, RealReceding's abort, a failing backup OCP, and an abort raised on the very step MPC is resumed."""
import numpy as np
import pytest

from conftest import sample_instances
from fake_solver import OracleSolver, make_double_controller
from policy_numerics import SolverNumerics
from oracle import policy_oracle as po
from safe_mpc_amd import closed_loop as cl
from safe_mpc_amd.parser import Parameters


class ScriptedSolver(OracleSolver):
    """the CPU double with a status matrix [call][instance] (None entries: the solver's own status)"""

    def __init__(self, problem, net=None):
        super().__init__(problem, net)
        self.matrix, self.calls = None, 0

    def solve(self, x0, xg, ug, p, out=None):
        x, u, st, it = super().solve(x0, xg, ug, p, out)
        st = np.array(st, np.int32)
        if self.matrix is not None and self.calls < len(self.matrix):
            for b, v in enumerate(self.matrix[self.calls][:len(st)]):
                if v is not None:
                    st[b] = v
        self.calls += 1
        return x, u, st, it


class ScriptedBackup(OracleSolver):
    def __init__(self, problem, net=None):
        super().__init__(problem, net)
        self.script = None

    def solve(self, x0, xg, ug, p, out=None):
        x, u, st, it = super().solve(x0, xg, ug, p, out)
        st = np.array(st, np.int32)
        if self.script is not None:
            for i in range(len(st)):
                v = self.script(np.asarray(x0)[i])
                if v is not None:
                    st[i] = v
        return x, u, st, it


def _params(N=5, Nb=4, abort_flag=True):
    par = Parameters({}, 'z1')
    par.nq, par.n_dof_safe_set, par.net_size, par.N, par.back_hor = 6, 6, [12, 32, 1], N, Nb
    par.abort_flag = abort_flag
    return par


def _swap_solver(ctrl, cls):
    """replace the double's solver object by a scripted one around the same problem / network"""
    net = ctrl.net
    ctrl.ocp_solver = cls(ctrl.problem, net)
    return ctrl


def _safe_rule(x):
    """a scripted safe-set verdict, a pure function of the node's state: both statements see the same answers"""
    return bool(np.sin(37.0 * x[0] + 11.0 * x[1]) > -0.2)


def _vec_safe(x):
    x = np.asarray(x, float)
    flat = x.reshape(-1, x.shape[-1])
    return np.array([_safe_rule(v) for v in flat]).reshape(x.shape[:-1])


def _status_matrix(kind, n_steps, B, N):
    """[step][instance]: 0 / 4 / None.  Instance 0 never fails; 1 fails long enough to abort, and fails again on the step it
    resumes; 2 fails once in a while; 3 aborts into a backup OCP that fails; the rest follow the solver."""
    M = [[None] * B for _ in range(n_steps)]
    for j in range(n_steps):
        M[j][0] = 0
        if B > 1:
            M[j][1] = 4 if (2 <= j < 2 + N + 2 or 14 <= j < 40) else 0
        if B > 2:
            M[j][2] = 4 if j % 4 == 1 else 0
        if B > 3:
            M[j][3] = 4 if j >= 1 else 0
    return M


def _run_mirror(par, kind, xg, ug, n_steps, matrix, backup_rule, scripted_safe):
    ctrls = []

    def make_controller(name, batch):
        c = _swap_solver(make_double_controller(name, par, batch), ScriptedSolver)
        c.ocp_solver.matrix = matrix
        if scripted_safe:
            c.checkSafeConstraints = lambda x: _vec_safe(x)
        ctrls.append(c)
        return c

    def make_backup(batch):
        b = _swap_solver(make_double_controller('backup', par, batch), ScriptedBackup)
        b.ocp_solver.script = backup_rule
        return b
    return cl.run_mpc(par, kind, xg, ug, n_steps=n_steps, make_controller=make_controller, make_backup=make_backup), ctrls[0]


def _run_oracle(par, kind, xg, ug, n_steps, matrix, backup_rule, scripted_safe):
    B = len(xg)
    ref = make_double_controller(kind, par, 1)
    bref = make_double_controller('backup', par, 1)
    out = []
    for b in range(B):
        num = SolverNumerics(OracleSolver(ref.problem, ref.net), ref.problem, par, OracleSolver(bref.problem, bref.net), bref.problem)
        num.status_script = (lambda j, b=b: matrix[j][b] if j < len(matrix) else None)
        num.backup_script = backup_rule
        if scripted_safe:
            num.safe_script = _safe_rule
        inst = po.PolicyInstance('stwa' if kind == 'stwa' else kind, par.N, ref.nx, ref.nu, abort_flag=par.abort_flag)
        out.append(po.run_closed_loop(inst, num, xg[b], ug[b], n_steps, bref.N, ref.nq, on_step=num.on_step))
    return out


@pytest.mark.parametrize('kind,abort_flag', [('naive', True), ('constraint_everywhere', True), ('htwa', True), ('stwa', True),
                                             ('receding', True), ('receding', False), ('real_receding', True)])
def test_numpy_mirror_equals_scalar_oracle_closed_loop(kind, abort_flag):
    N, Nb, B, n_steps = 5, 4, 6, 44
    par = _params(N, Nb, abort_flag)
    probe = make_double_controller(kind if kind != 'stwa' else 'htwa', par, 1)
    x0 = sample_instances(probe.problem, B, seed=5, vel_scale=0.05)
    xg, ug = np.repeat(x0[:, None, :], N + 1, axis=1), np.zeros((B, N, 6))
    matrix = _status_matrix(kind, n_steps, B, N)
    thr = np.sort(x0[:, 0])[B // 2]
    backup_rule = lambda xv: (4 if xv[0] > 10.0 else None)           # (instance 3's viable state is moved past 10 below)
    scripted_safe = kind in ('receding', 'real_receding')
    if B > 3:
        # instance 3: every backup OCP from its viable state fails -> lost at the step of the event
        x3 = float(x0[3, 0])
        backup_rule = lambda xv, x3=x3: (4 if abs(xv[0] - x3) < 0.3 else None)
    mir, ctrl = _run_mirror(par, kind, xg, ug, n_steps, matrix, backup_rule, scripted_safe)
    ora = _run_oracle(par, kind, xg, ug, n_steps, matrix, backup_rule, scripted_safe)
    xo, uo = np.array([r['x'] for r in ora]), np.array([r['u'] for r in ora])
    assert np.array_equal(np.isnan(mir['x']), np.isnan(xo)), np.where(np.isnan(mir['x']).any(2) != np.isnan(xo).any(2))
    assert np.array_equal(np.isnan(mir['u']), np.isnan(uo))
    # (Tolerances: the scalar statement integrates its guesses in numpy, the double in C++ -- rounding-level differences in the
    #  viable state.  The backup OCP has zero cost (+ 1e-4 regularisation), so its solution is only pinned to ~1e-4 by the
    #  IPM's exit test and moves by that much with such differences; the PD law then carries it into u.)
    assert np.nanmax(np.abs(mir['x'] - xo)) < 1e-4 and np.nanmax(np.abs(mir['u'] - uo)) < 2e-3 * (1 + np.nanmax(np.abs(uo)))
    conv, coll, viable, unconv = po.outcome_lists(ora)
    assert (mir['conv_idx'], mir['collisions_idx'], sorted(mir['viable_idx']), mir['unconv_idx']) == (conv, coll, sorted(viable), unconv)
    ev = [(b, j, xv) for b, r in enumerate(ora) for (j, xv) in r['events']]
    assert mir['x_viable'].shape[0] == len(ev)
    if ev:
        assert np.abs(mir['x_viable'] - np.array([e[2] for e in ev])).max() < 1e-7
    if kind in ('receding', 'real_receding'):
        assert np.array_equal(mir['r_receding'][:, :, 0], np.array([r['r'] for r in ora]))
    if kind in ('htwa', 'stwa', 'receding', 'real_receding') and abort_flag:
        assert len(ev) >= 2                                          # the abort branches ran ...
        assert coll                                                  # ... a backup OCP failed ...
        assert any(r['n_viable'] > 0 for r in ora)                   # ... and one succeeded
    if not abort_flag:
        assert not ev                                                # controller.py:476-481: r stops at 0, no abort


def test_abort_on_the_resume_step_opens_no_new_event():
    """scripts/mpc.py:137-146 -- the controller step of an instance that resumes MPC runs inside the
    `if sa_flag:` branch; if it aborts again no viable state is appended and no backup OCP solved: x_viable keeps one row, the
    old backup trajectory's last node is held and the abort clock keeps counting."""
    N, Nb, n_steps = 5, 4, 40
    par = _params(N, Nb, True)
    probe = make_double_controller('htwa', par, 1)
    x0 = sample_instances(probe.problem, 2, seed=5, vel_scale=0.0)
    xg, ug = np.repeat(x0[:, None, :], N + 1, axis=1), np.zeros((2, N, 6))
    matrix = [[0, 4 if j >= 1 else 0] for j in range(n_steps)]       # instance 1 never solves again after step 0
    mir, _ = _run_mirror(par, 'htwa', xg, ug, n_steps, matrix, None, False)
    ora = _run_oracle(par, 'htwa', xg, ug, n_steps, matrix, None, False)
    assert len(ora[1]['events']) == 1                                 # the reference's loop: ONE event, however often it re-aborts
    assert mir['x_viable'].shape[0] == 1
    assert np.nanmax(np.abs(mir['x'][1] - ora[1]['x'])) < 1e-4 and np.nanmax(np.abs(mir['u'][1] - ora[1]['u'])) < 2e-3 * (1 + np.nanmax(np.abs(ora[1]['u'])))
    # after the backup trajectory (Nb nodes) the instance alternates "resume -> abort again" and PD hold: it did step again
    assert not np.isnan(ora[1]['u']).any()
    # without the quirk every abort is an event (the behaviour before round 3)
    par.reference_quirks = False
    mir2, _ = _run_mirror(par, 'htwa', xg, ug, n_steps, matrix, None, False)
    assert mir2['x_viable'].shape[0] > 1


@pytest.mark.parametrize('kind', ['naive', 'constraint_everywhere', 'htwa', 'receding', 'real_receding'])
def test_numpy_mirror_step_equals_scalar_oracle_step(kind):
    """<Controller>.step alone, step by step, on states that are NOT produced by the loop (random kicks): controls, abort
    flags, fails, r, viable state, shifted guesses, current_step."""
    N, B = 6, 5
    par = _params(N, 4, True)
    rng = np.random.default_rng(1)
    mirror = _swap_solver(make_double_controller(kind, par, B), ScriptedSolver)
    x0 = sample_instances(mirror.problem, B, seed=2, vel_scale=0.1)
    xg, ug = np.repeat(x0[:, None, :], N + 1, axis=1), rng.normal(scale=0.1, size=(B, N, 6))
    mirror.setGuess(xg.copy(), ug.copy())
    mirror.reset_controller()
    if kind in ('receding', 'real_receding'):
        mirror.checkSafeConstraints = lambda x: _vec_safe(x)
    steps = 3 * N
    matrix = [[(4 if (j + b) % 3 == 0 or (b == 1 and j >= 2) else 0) for b in range(B)] for j in range(steps)]
    mirror.ocp_solver.matrix = matrix
    insts, nums = [], []
    for b in range(B):
        num = SolverNumerics(OracleSolver(mirror.problem, mirror.net), mirror.problem, par)
        num.status_script = (lambda j, b=b: matrix[j][b])
        if kind in ('receding', 'real_receding'):
            num.safe_script = _safe_rule
        inst = po.PolicyInstance(kind, N, mirror.nx, mirror.nu, abort_flag=True)
        inst.set_guess(xg[b], ug[b])
        inst.reset()
        insts.append(inst)
        nums.append(num)
    x = x0.copy()
    seen_abort = False
    for j in range(steps):
        um, am = mirror.step(x)
        for b in range(B):
            nums[b].on_step(j)
            uo, ao = po.step(insts[b], nums[b], x[b])
            assert bool(am[b]) == ao, (j, b)
            assert np.abs(um[b] - uo).max() < 1e-8 * (1 + np.abs(uo).max()), (j, b)
            assert int(mirror.fails[b]) == insts[b].fails and int(mirror.current_step[b]) == insts[b].current_step, (j, b)
            assert np.abs(mirror.x_guess[b] - np.array(insts[b].x_guess)).max() < 1e-8
            assert np.abs(mirror.u_guess[b] - np.array(insts[b].u_guess)).max() < 1e-8 * (1 + np.abs(mirror.u_guess[b]).max())
            if hasattr(mirror, 'r'):
                assert int(mirror.r[b]) == insts[b].r, (j, b)
            if mirror.can_abort:
                assert np.abs(mirror.x_viable[b] - insts[b].x_viable).max() < 1e-8, (j, b)
            seen_abort |= ao
        x = x + par.dt * np.hstack([x[:, 6:], um]) + rng.normal(scale=1e-3, size=x.shape)
    if mirror.can_abort:
        assert seen_abort


def _moving_traj(ee_ref, n_cols):
    """a reference that moves with the step (controller.py:153-156 indexes cost.traj by current_step + i): a small circle
    around the configured ee_ref"""
    t = np.arange(n_cols) * 0.15
    return np.asarray(ee_ref, float)[:, None] + 0.05 * np.vstack([np.cos(t) - 1.0, np.sin(t), 0.5 * np.sin(2 * t)])


@pytest.mark.parametrize('kind', ['st', 'htwa', 'receding'])
def test_reference_trajectory_advances_with_the_step(kind):
    """p[i][0:3] = cost.traj[:, current_step + i] at every solve (controller.py:153-156,
    cost_definition.py:30-31).  The numpy mirror with setTrajectory against the scalar oracle with inst.traj, step by step, with
    rejected solves and (htwa / receding) aborts in the script: an aborting instance returns before `current_step += 1`
    (controller.py:384-385, 483-487) and must see the SAME columns again at its next solve."""
    N, B = 6, 4
    par = _params(N, 4, True)
    rng = np.random.default_rng(3)
    mirror = _swap_solver(make_double_controller(kind, par, B), ScriptedSolver)
    x0 = sample_instances(mirror.problem, B, seed=2, vel_scale=0.05)
    xg, ug = np.repeat(x0[:, None, :], N + 1, axis=1), rng.normal(scale=0.1, size=(B, N, 6))
    steps = 3 * N
    traj = _moving_traj(mirror.problem.ee_ref, steps + 1 + N)
    mirror.setGuess(xg.copy(), ug.copy())
    mirror.reset_controller()
    mirror.setTrajectory(traj)
    if kind == 'receding':
        mirror.checkSafeConstraints = lambda x: _vec_safe(x)
    matrix = [[(4 if (b == 1 and j >= 2) or (b == 2 and j % 5 == 1) else 0) for b in range(B)] for j in range(steps)]
    mirror.ocp_solver.matrix = matrix
    seen_p = []
    orig = mirror.ocp_solver.solve
    mirror.ocp_solver.solve = lambda x0_, xg_, ug_, p_, out=None: (seen_p.append(np.array(p_)), orig(x0_, xg_, ug_, p_, out))[1]
    insts, nums = [], []
    for b in range(B):
        num = SolverNumerics(OracleSolver(mirror.problem, mirror.net), mirror.problem, par)
        num.status_script = (lambda j, b=b: matrix[j][b])
        if kind == 'receding':
            num.safe_script = _safe_rule
        inst = po.PolicyInstance(kind, N, mirror.nx, mirror.nu, abort_flag=True)
        inst.set_guess(xg[b], ug[b])
        inst.reset()
        inst.traj = traj
        insts.append(inst)
        nums.append(num)
    x = x0.copy()
    seen_abort = False
    for j in range(steps):
        cs = [inst.current_step for inst in insts]
        um, am = mirror.step(x)
        for b in range(B):
            # what the mirror handed the solver at this step: the columns current_step + i of traj
            assert np.array_equal(seen_p[-1][b, :, :3], traj[:, cs[b]:cs[b] + N + 1].T), (j, b)
            nums[b].on_step(j)
            uo, ao = po.step(insts[b], nums[b], x[b])
            assert bool(am[b]) == ao, (j, b)
            assert np.abs(um[b] - uo).max() < 1e-8 * (1 + np.abs(uo).max()), (j, b)
            assert int(mirror.current_step[b]) == insts[b].current_step, (j, b)
            seen_abort |= ao
        x = x + par.dt * np.hstack([x[:, 6:], um]) + rng.normal(scale=1e-3, size=x.shape)
    if mirror.can_abort:
        assert seen_abort
    # the moving reference matters: the same loop with the constant ee_ref gives other controls
    const = _swap_solver(make_double_controller(kind, par, B), ScriptedSolver)
    const.setGuess(xg.copy(), ug.copy())
    const.reset_controller()
    u_c, _ = const.step(x0)
    moving = _swap_solver(make_double_controller(kind, par, B), ScriptedSolver)
    moving.setGuess(xg.copy(), ug.copy())
    moving.reset_controller()
    moving.setTrajectory(traj + 0.1)
    u_m, _ = moving.step(x0)
    assert np.abs(u_c - u_m).max() > 1e-3
    moving.setTrajectory(None)                                       # back to the constant ee_ref
    assert np.array_equal(moving.p[:, :, :3], const.p[:, :, :3])
