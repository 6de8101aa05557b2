"""oracle/policy_oracle.py -- TEST INFRASTRUCTURE: a scalar, one-instance restatement of the reference's policy layer.

Only tests/ may import this file; the product (safe_mpc_amd/, bench.py, scripts/) never does.  It states, in plain per-instance
``if`` / ``for`` code, what the reference does around its OCP solver:

  * the ``step()`` automata of the registered controllers
      Naive / TerminalZeroVelocity / ST      /root/reference/src/safe_mpc/controller.py:274-284, 308-317
      ControllerSafeSetEverywhere            controller.py:651-661
      STWA / HTWA                            controller.py:369-388
      Receding                               controller.py:448-498
      RealReceding                           controller.py:524-565
    with ``solve`` (:136-167), ``provideControl`` (:169-184) and ``guessCorrection`` (:226-231);
  * the driver's closed loop with safe abort, PD tracking of the backup trajectory and the failure taxonomy
      /root/reference/scripts/mpc.py:118-287.

Nothing here is vectorised and nothing is shared with safe_mpc_amd/controller.py or kernels_policy.hpp (the two product
statements of the same automata, which the tests compare with this one).  All numerics are injected through a ``numerics``
object, so the automata can be driven by scripted statuses / scripted safe-set verdicts as well as by a real solver:

    numerics.integrate_naively(x, u) -> x_next                       env_model.py:210-212 (the controller's own model)
    numerics.solve(x0, x_guess, u_guess, flags, lo, hi, ref=None) -> (status, x_traj, u_traj)
            flags[k]: last entry of the node parameter p (> 0: safe-set row on, utils.py:207-210)
            lo / hi : per-node state bounds [N+1][nx], or None for the formulation's own
            ref     : per-node reference points [N+1][3] (first three entries of p, controller.py:153-156), or None for the
                      constant ee_ref (only passed when the instance carries a trajectory)
    numerics.check_state_constraints(x_traj) -> bool                  env_model.py:170-173
    numerics.check_safe(x_node) -> bool                               safe_set.py:61-68
  driver only:
    numerics.plant(x, u) -> x_next                                    env_model.py:192-206
    numerics.check_state_bounds(x) -> bool                            env_model.py:175-177
    numerics.check_collision(x) -> bool                               env_model.py:238-243
    numerics.backup_solve(x_viable) -> (status, x_abort, u_abort)     mpc.py:174-177 (constant guess, SafeBackupController)
    numerics.converged(x_last) -> bool                                mpc.py:273
"""
import numpy as np


class PolicyInstance:
    """The state one controller object of the reference carries for ONE closed loop (controller.py:112-125, 364-367, 402-407)."""

    def __init__(self, kind, N, nx, nu, abort_flag=True):
        assert kind in ('naive', 'zerovel', 'st', 'constraint_everywhere', 'stwa', 'htwa', 'receding', 'real_receding')
        self.kind, self.N, self.nx, self.nu = kind, int(N), int(nx), int(nu)
        self.abort_flag = bool(abort_flag)
        self.x_guess = [np.zeros(nx) for _ in range(self.N + 1)]
        self.u_guess = [np.zeros(nu) for _ in range(self.N)]
        self.x_temp = [np.zeros(nx) for _ in range(self.N + 1)]
        self.u_temp = [np.zeros(nu) for _ in range(self.N)]
        self.flags = [1.0] * (self.N + 1)          # last entry of p per node; the solver keeps whatever was set last (:153-156)
        self.lo = None                             # per-node state bounds set through constraints_set (RealReceding)
        self.hi = None
        self.x_viable = np.zeros(nx)
        self.last_status = 4                       # controller.py:125
        self.traj = None                           # cost.traj [3][n_steps + 1 + N] (cost_definition.py:30-31, 89); None = constant ee_ref
        self.reset()

    # controller.py:233-237 (+ :362-365 for the safe-set classes, :445-447 for the receding ones)
    def reset(self):
        self.fails = 0
        self.current_step = 0
        self.r = self.N

    # controller.py:196-198, 390-393
    def set_guess(self, x_guess, u_guess):
        self.x_guess = [np.array(x, float) for x in x_guess]
        self.u_guess = [np.array(u, float) for u in u_guess]
        if self.kind in ('stwa', 'htwa', 'receding', 'real_receding'):
            self.x_viable = np.array(x_guess[-1], float)


def guess_correction(inst, numerics):
    """controller.py:226-231"""
    for i in range(inst.N):
        inst.x_guess[i + 1] = np.array(numerics.integrate_naively(inst.x_guess[i], inst.u_guess[i]), float)


def solve(inst, numerics, x0):
    """controller.py:136-167: the iterate is kept whatever the status."""
    if inst.traj is not None:
        # :153-156  p_i = [cost.traj[:, current_step + i], alpha, flag_i]
        ref = [np.array([inst.traj[c][inst.current_step + i] for c in range(3)], float) for i in range(inst.N + 1)]
        status, x_traj, u_traj = numerics.solve(np.array(x0, float), inst.x_guess, inst.u_guess, inst.flags, inst.lo, inst.hi, ref=ref)
    else:
        status, x_traj, u_traj = numerics.solve(np.array(x0, float), inst.x_guess, inst.u_guess, inst.flags, inst.lo, inst.hi)
    for i in range(inst.N):
        inst.x_temp[i] = np.array(x_traj[i], float)
        inst.u_temp[i] = np.array(u_traj[i], float)
    inst.x_temp[inst.N] = np.array(x_traj[inst.N], float)
    inst.last_status = int(status)
    return int(status)


def provide_control(inst):
    """controller.py:169-184: after a failure the OLD guess is shifted and its first control applied, otherwise the new one."""
    N = inst.N
    if inst.fails > 0:
        u = np.array(inst.u_guess[0])
        src_x, src_u = inst.x_guess, inst.u_guess
    else:
        u = np.array(inst.u_temp[0])
        src_x, src_u = inst.x_temp, inst.u_temp
    new_x = [np.array(src_x[i + 1]) for i in range(N)]      # np.roll(., -1) followed by "last = one before last"
    new_x.append(np.array(src_x[N]))
    new_u = [np.array(src_u[i + 1]) for i in range(N - 1)]
    new_u.append(np.array(src_u[N - 1]))
    inst.x_guess, inst.u_guess = new_x, new_u
    return u


def step(inst, numerics, x):
    """One ``controller.step(x)`` -> (u, abort)."""
    k = inst.kind
    if k in ('naive', 'zerovel', 'st'):
        return _step_plain(inst, numerics, x, False)
    if k == 'constraint_everywhere':
        return _step_plain(inst, numerics, x, True)
    if k in ('stwa', 'htwa'):
        return _step_stwa(inst, numerics, x)
    if k == 'receding':
        return _step_receding(inst, numerics, x)
    return _step_real_receding(inst, numerics, x)


def _step_plain(inst, numerics, x, state_check):
    """controller.py:274-284 (308-317 is the same text); :651-661 adds the state test of the new trajectory."""
    guess_correction(inst, numerics)
    status = solve(inst, numerics, x)
    good = status == 0
    if good and state_check:
        good = bool(numerics.check_state_constraints(inst.x_temp))
    if good:
        inst.fails = 0
    else:
        inst.fails += 1
    inst.current_step += 1
    return provide_control(inst), False


def _step_stwa(inst, numerics, x):
    """controller.py:375-388"""
    guess_correction(inst, numerics)
    status = solve(inst, numerics, x)
    if status == 0 and numerics.check_state_constraints(inst.x_temp):
        inst.fails = 0
    else:
        if inst.fails == 0:
            inst.x_viable = np.array(inst.x_guess[-2])
        if inst.fails == inst.N - 1:
            return np.array(inst.u_guess[0]), True
        inst.fails += 1
    inst.current_step += 1
    return provide_control(inst), False


def _after_receding_solve(inst, numerics, status):
    """controller.py:471-498 == :540-565 (the two classes share this text)."""
    if inst.abort_flag:
        inst.r -= 1
    else:
        if inst.r > 0:
            inst.r -= 1
    if inst.r == 0 and inst.abort_flag:
        inst.x_viable = np.array(inst.x_guess[1])
        inst.r = inst.N
        return np.array(inst.u_guess[0]), True
    if status == 0 and numerics.check_state_constraints(inst.x_temp):
        inst.fails = 0
        first = inst.r + 2                      # range(self.r + 2, self.N + 1) is evaluated once, before r moves
        for i in range(first, inst.N + 1):
            if numerics.check_safe(inst.x_temp[i]):
                inst.r = i - 1
    else:
        inst.fails += 1
    inst.current_step += 1
    return provide_control(inst), False


def _step_receding(inst, numerics, x):
    """controller.py:448-498"""
    N = inst.N
    guess_correction(inst, numerics)
    for i in range(1, N):
        if i != inst.r:
            inst.flags[i] = -1.0
    inst.flags[N] = 1.0
    if inst.r < N:
        inst.flags[inst.r] = 1.0
    status = solve(inst, numerics, x)
    return _after_receding_solve(inst, numerics, status)


def _step_real_receding(inst, numerics, x):
    """controller.py:524-565: no guessCorrection; node r is boxed to the state the previous plan reaches one node later."""
    N = inst.N
    if inst.lo is None:
        lo_d, hi_d = numerics.default_state_bounds(N)
        inst.lo = [np.array(v, float) for v in lo_d]
        inst.hi = [np.array(v, float) for v in hi_d]
    x_min, x_max = numerics.model_state_bounds()
    if inst.r < N:
        inst.lo[inst.r] = np.array(inst.x_guess[inst.r + 1]) - 1e-3
        inst.hi[inst.r] = np.array(inst.x_guess[inst.r + 1]) + 1e-3
    for i in range(N):
        if i != inst.r:
            inst.lo[i] = np.array(x_min, float)
            inst.hi[i] = np.array(x_max, float)
    status = solve(inst, numerics, x)
    return _after_receding_solve(inst, numerics, status)


# =========================================================================================================================
# the driver's loop for ONE instance (scripts/mpc.py:118-287)
# =========================================================================================================================
def run_closed_loop(inst, numerics, x_guess, u_guess, n_steps, Nb, nq, on_step=None):
    """Returns a dict: x [n_steps+1][nx] and u [n_steps][nu] (NaN where the reference leaves its pre-filled NaN), the abort
    events [(step, x_viable)], r at every step the controller stepped (-1 otherwise), and what the outcome lists are built from
    (collided: in collisions_idx; n_viable: times appended to viable_idx; conv: the test of mpc.py:273 on the last state)."""
    nx, nu = inst.nx, inst.nu
    kp, kd = 1.0, 1e2                                          # mpc.py:97
    x_sim = np.full((n_steps + 1, nx), np.nan)
    u_log = np.full((n_steps, nu), np.nan)
    r_log = np.full(n_steps, -1, dtype=np.int64)
    x_sim[0] = np.array(x_guess[0], float)                     # mpc.py:113,117 (x_init is the first node of the guess)
    inst.set_guess(x_guess, u_guess)                           # mpc.py:119
    inst.reset()                                               # mpc.py:120
    ja = 0
    sa_flag = False
    x_abort = u_abort = None
    events = []
    collided = False
    n_viable = 0                                               # how often mpc.py:189 appended this instance to viable_idx
    has_r = inst.kind in ('receding', 'real_receding')
    for j in range(n_steps):
        if on_step is not None:
            on_step(j)
        x = x_sim[j]
        if sa_flag:                                            # mpc.py:130-146
            if ja < Nb:
                u = np.array(u_abort[ja]) - (kp * (x[:nq] - x_abort[ja][:nq]) + kd * (x[nq:] - x_abort[ja][nq:]))
            else:
                if np.all(x[nq:] < 5e-3):
                    sa_flag = False
                    if has_r:
                        r_log[j] = inst.r
                    u, sa_flag = step(inst, numerics, x)       # an abort raised HERE opens no new event: x_abort, ja are kept
                else:
                    u = -(kp * (x[:nq] - x_abort[-1][:nq]) + 3e2 * (x[nq:] - x_abort[-1][nq:]))
            ja += 1
            u_log[j] = u
        else:                                                  # mpc.py:149-190
            if has_r:
                r_log[j] = inst.r
            u, sa_flag = step(inst, numerics, x)
            u_log[j] = u
            if sa_flag:
                xv = np.array(inst.x_viable)
                events.append((j, xv))
                status, xa, ua = numerics.backup_solve(xv)
                if status != 0:
                    collided = True                            # mpc.py:186-190: lost at this step, nothing is integrated
                    break
                ja = 0
                n_viable += 1
                x_abort = [np.array(v, float) for v in xa]
                u_abort = [np.array(v, float) for v in ua]
        x_next = np.array(numerics.plant(x, u), float)         # mpc.py:240
        x_sim[j + 1] = x_next
        if not numerics.check_state_bounds(x_next):            # mpc.py:246-257
            collided = True
            break
        if not numerics.check_collision(x_next):               # mpc.py:258-264
            collided = True
            break
    last = x_sim[-1]
    conv = bool(not np.isnan(last).any() and numerics.converged(last))      # mpc.py:273 (a NaN norm compares false)
    return {'x': x_sim, 'u': u_log, 'r': r_log, 'events': events, 'collided': collided, 'n_viable': n_viable, 'conv': conv}


def outcome_lists(results):
    """mpc.py:273-286 over the per-instance results of :func:`run_closed_loop`.  Two things the reference's list handling does,
    restated as they are: a converged instance is taken out of viable_idx ONCE (list.remove, :277-278), so one that went through
    two abort events and still converged stays in both lists; and convergence is tested on x_sim[-1] whether or not the
    instance was also recorded as a failure at the very last step."""
    conv_idx = [i for i, r in enumerate(results) if r['conv']]
    coll_idx = [i for i, r in enumerate(results) if r['collided']]
    viable_idx = [i for i, r in enumerate(results) if r['n_viable'] - (1 if r['conv'] else 0) > 0 and i not in coll_idx]
    unconv = [i for i in range(len(results)) if i not in conv_idx and i not in coll_idx and i not in viable_idx]
    return conv_idx, coll_idx, viable_idx, unconv
