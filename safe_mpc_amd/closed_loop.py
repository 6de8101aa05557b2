"""Batched versions of the reference's two live entry points, as library functions:

* :func:`generate_guess`  -- scripts/guess_acados.py:72-158,235-244 (warm starts by SQP to convergence)
* :func:`run_mpc`         -- scripts/mpc.py:102-317 (closed loop with safe abort, failure taxonomy, result dict)

Every instance of the reference's outer loop (mpc.py:102) is one row of the arrays handled here; the inner loop over MPC
steps stays a Python loop, each iteration being a handful of batched engine calls.  ``scripts/mpc.py`` and
``scripts/guess_acados.py`` are thin CLI wrappers that keep the reference's flags, file names and pickle formats.
"""
from __future__ import annotations

import copy
import os
import pickle

import numpy as np

from ._xp import InPlaceState
from .controller import SafeBackupController, get_controller
from .problem import JOINT_DTYPE
from .urdf import Inertial, Origin, RobotDescription, SerialChain


# ---- file names --------------------------------------------------------------------------------------
def guess_file(params, model_name, cont_name, horizon, use_net):
    track = 'traj_track' if params.track_traj else ''
    return (f'{params.DATA_DIR}{model_name}_{cont_name}_{horizon}hor_{int(params.alpha)}sm_use_net{use_net}_{track}'
            f'_q_collision_margins_{params.q_margin}_{params.collision_margin}_guess.pkl')


def result_file(params, model_name, cont_name, horizon, use_net, noise, control_noise, jm, cm):
    track = 'traj_track' if params.track_traj else ''
    return (f'{params.DATA_DIR}{model_name}_{cont_name}_use_net{use_net}_{horizon}hor_{int(params.alpha)}sm_{track}'
            f'noise_{noise}_control_noise{control_noise}_q_collision_margins_{jm}_{cm}_mpc.pkl')


def halton(n, dim, skip=1):
    """Unscrambled Halton points (guess_acados.py:79 uses scipy's qmc.Halton(d, scramble=False))."""
    primes = [2, 3, 5, 7, 11, 13, 17, 19, 23, 29]
    idx = np.arange(skip, skip + n)
    out = np.zeros((n, dim))
    for d in range(dim):
        b, f, k, r = primes[d], 1.0, idx.copy(), np.zeros(n)
        while k.max() > 0:
            f /= b
            r += f * (k % b)
            k //= b
        out[:, d] = r
    return out


# ---- model noise (utils.py:126-171, generate_urdf_noise.py:20-36, env_model.py:321-328) ----------------------------------
def perturbed_joint_tables(params, nq, noise_pct, seeds):
    """Per-instance plant models: every URDF link's mass, inertia entries and COM coordinates are multiplied by
    1 + U(-noise, noise)% (utils.py:138-166) BEFORE lumping -- in memory, instead of one URDF file per instance."""
    base = params.robot_descr
    out = np.zeros((len(seeds), nq), JOINT_DTYPE)
    # The reference draws uniform(-n, n) for EVERY field, also where the field and hence n is zero (ixy = 0 ...): the draw returns 0
    # but advances the generator, so a replay of its seeded perturbations needs the same.  Default (reference_quirks); without it
    # zero-magnitude fields consume nothing (rounds 1-4).
    always_draw = bool(getattr(params, 'reference_quirks', True))
    for row, seed in enumerate(seeds):
        rng = np.random.default_rng(int(seed))
        draw = lambda n_: rng.uniform(-n_, n_) if (n_ > 0 or (always_draw and noise_pct > 0)) else 0.0
        links = []
        for l in base.links:
            l2 = copy.copy(l)
            if l.inertial is not None:
                m = l.inertial.mass
                m = m + draw(abs(m) * noise_pct / 100)
                I = l.inertial.inertia.copy()
                for (a, b) in [(0, 0), (1, 1), (2, 2), (0, 1), (1, 2), (0, 2)]:       # ixx iyy izz ixy iyz ixz (utils.py:128)
                    v = I[a, b] + draw(abs(I[a, b]) * noise_pct / 100)
                    I[a, b] = I[b, a] = v
                xyz = l.inertial.origin.xyz.copy()
                for k in range(3):
                    xyz[k] += draw(abs(xyz[k] * noise_pct / 100))
                l2.inertial = Inertial(Origin(xyz, l.inertial.origin.rpy), m, I)
            links.append(l2)
        chain = SerialChain(RobotDescription(links, base.joints, base.name), nq)
        for i, j in enumerate(chain.joints):
            o = out[row, i]
            o['R0'], o['p0'], o['axis'] = j.R0.reshape(-1), j.p0, j.axis
            o['mass'], o['com'] = j.mass, j.com
            I = j.inertia
            o['inertia'] = [I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]]
            o['q_min'], o['q_max'], o['v_max'], o['tau_max'] = j.q_min, j.q_max, j.v_max, j.tau_max
    return out


def _mv3(R, v):
    """R [3, 3] (or [..., 3, 3]) times v [..., 3] with the sums written out: elementwise numpy only, so that a row of a batch gets the
    same bits whatever the batch's size (a BLAS matmul may pick another kernel, and another summation order, per shape)."""
    return R[..., :, 0] * v[..., None, 0] + R[..., :, 1] * v[..., None, 1] + R[..., :, 2] * v[..., None, 2]


def _mm3(A, B):
    """A [..., 3, 3] times B [..., 3, 3], sums written out (see _mv3)."""
    return (A[..., :, 0, None] * B[..., None, 0, :] + A[..., :, 1, None] * B[..., None, 1, :]) + A[..., :, 2, None] * B[..., None, 2, :]


def perturbed_joint_tables_batched(params, nq, noise_pct, seeds):
    """The per-instance plant models of :func:`perturbed_joint_tables` for MANY seeds at once (BASELINE config 2: 65 536 instances,
    seed = instance id): the same draws in the same order -- one ``default_rng(seed)`` per instance, ten draws per URDF
    link in the reference's order (mass, ixx iyy izz ixy iyz ixz, x y z; utils.py:128, 138-166), bit for bit those of the per-seed
    function -- and the lumping into joint tables as elementwise array arithmetic over the batch.  The per-seed function stays the
    statement the tests hold this one against (draws equal to the bit, tables to 1e-13 of their scale: SerialChain lumps with
    BLAS-backed matmuls, this one with written-out sums)."""
    base = params.robot_descr
    seeds = np.asarray(seeds, np.int64).reshape(-1)
    B = len(seeds)
    always_draw = bool(getattr(params, 'reference_quirks', True))
    chain0 = SerialChain(base, nq)                      # structure: carriers of every link, joint frames, limits (not perturbed)
    # nominal value of every field that is drawn for, in the reference's order
    fields = []                                         # (link index, kind, a, b)
    for li, l in enumerate(base.links):
        if l.inertial is None:
            continue
        fields.append((li, 'm', 0, 0))
        fields += [(li, 'I', a, b) for (a, b) in [(0, 0), (1, 1), (2, 2), (0, 1), (1, 2), (0, 2)]]
        fields += [(li, 'c', k, 0) for k in range(3)]

    def nominal(f):
        l = base.links[f[0]].inertial
        return l.mass if f[1] == 'm' else (l.inertia[f[2], f[3]] if f[1] == 'I' else l.origin.xyz[f[2]])
    nom = np.array([nominal(f) for f in fields])
    # (the magnitudes exactly as the per-seed function forms them: abs(v) * noise / 100 for mass and inertia, abs(v * noise / 100) for
    #  the centre of mass)
    mag = np.array([abs(v * noise_pct / 100) if f[1] == 'c' else abs(v) * noise_pct / 100 for f, v in zip(fields, nom)])
    drawn = (mag > 0) | (always_draw and noise_pct > 0)
    nd = int(drawn.sum())
    U = np.zeros((B, len(fields)))
    if nd:
        R = np.empty((B, nd))
        for r_, sd in enumerate(seeds):                 # one generator per instance, as the reference reseeds per model
            R[r_] = np.random.default_rng(int(sd)).random(nd)
        lo, hi = -mag[drawn], mag[drawn]
        U[:, drawn] = lo + (hi - lo) * R                # Generator.uniform(low, high) = low + (high - low) * next_double
    val = nom[None, :] + U                              # [B, fields]
    # lump: every link's inertial into the actuated link that carries it
    m = np.zeros((B, nq))
    mc = np.zeros((B, nq, 3))
    parts = [[] for _ in range(nq)]
    col = 0
    for li, l in enumerate(base.links):
        if l.inertial is None:
            continue
        mass, ent, xyz = val[:, col], val[:, col + 1:col + 7], val[:, col + 7:col + 10]
        col += 10
        if l.name not in chain0._carrier:
            continue
        idx, Rc, pc = chain0._carrier[l.name]
        if idx < 0:
            continue
        I = np.empty((B, 3, 3))
        for e_, (a, b) in enumerate([(0, 0), (1, 1), (2, 2), (0, 1), (1, 2), (0, 2)]):
            I[:, a, b] = ent[:, e_]
            I[:, b, a] = ent[:, e_]
        c = pc + _mv3(Rc, xyz)
        Ro = Rc @ l.inertial.origin.R                   # (constants of the model)
        Ic = _mm3(_mm3(np.broadcast_to(Ro, (B, 3, 3)), I), np.broadcast_to(Ro.T, (B, 3, 3)))
        m[:, idx] += mass
        mc[:, idx] += mass[:, None] * c
        parts[idx].append((mass, c, Ic))
    out = np.zeros((B, nq), JOINT_DTYPE)
    eye = np.eye(3)
    for i, j in enumerate(chain0.joints):
        if np.any(m[:, i] <= 0):
            raise ValueError(f'link moved by {j.name} has no mass')
        com = mc[:, i] / m[:, i, None]
        I = np.zeros((B, 3, 3))
        for (mi, ci, Ici) in parts[i]:
            d = ci - com
            dd = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]
            I += Ici + mi[:, None, None] * (dd[:, None, None] * eye - d[:, :, None] * d[:, None, :])
        o = out[:, i]
        o['R0'], o['p0'], o['axis'] = j.R0.reshape(-1), j.p0, j.axis
        o['mass'], o['com'] = m[:, i], com
        o['inertia'] = np.stack([I[:, 0, 0], I[:, 0, 1], I[:, 0, 2], I[:, 1, 1], I[:, 1, 2], I[:, 2, 2]], axis=1)
        o['q_min'], o['q_max'], o['v_max'], o['tau_max'] = j.q_min, j.q_max, j.v_max, j.tau_max
    return out


# ---- warm-start generation ----------------------------------------------------------------------------------------------------
def merit_terms(ctrl, x0, x, u):
    """Pieces of the l1 merit function of the OCP at the iterate (x, u), per instance, from one batched linearisation
    (``smpc_eval_nodes``): cost f, its gradient (dq, du parts) and the l1 norm of every constraint's violation -- initial
    state, dynamics defects, state box, torque rows, collision rows, safe-set row (where the formulation has it)."""
    pr, d = ctrl.problem, ctrl.problem.desc
    nq, N, dt = ctrl.nq, ctrl.N, ctrl.params.dt
    x, u = np.asarray(x, float), np.asarray(u, float)
    ev = ctrl.ocp_solver.eval_nodes(x, u, ctrl.p)
    ee = np.asarray(ev['ee'])
    cs = np.full(N + 1, d.cost_scale_stage)
    cs[N] = d.cost_scale_term
    if d.cost_kind != 0:
        err = np.sum((ee - ctrl.p[:, :, :3]) ** 2, axis=2)                           # [B, N+1]
        f = (cs * d.Q * err).sum(1) + (cs[:N] * d.R * np.sum(u ** 2, axis=2)).sum(1)
        gq = cs[None, :, None] * np.asarray(ev['cost_grad_q'])[:, :, :nq]
        gu = cs[None, :N, None] * 2.0 * d.R * u
    else:
        f, gq, gu = np.zeros(len(x)), np.zeros((len(x), N + 1, nq)), np.zeros_like(u)
    pos = lambda a: np.maximum(a, 0.0)
    viol = np.abs(x[:, 0] - x0).sum(1)
    xn = np.empty_like(x[:, 1:])
    xn[:, :, :nq] = x[:, :-1, :nq] + dt * x[:, :-1, nq:] + 0.5 * dt * dt * u
    xn[:, :, nq:] = x[:, :-1, nq:] + dt * u
    viol += np.abs(x[:, 1:] - xn).sum(axis=(1, 2))
    lo = np.tile(pr.lbx, (N + 1, 1)); hi = np.tile(pr.ubx, (N + 1, 1))
    lo[N], hi[N] = pr.lbx_e, pr.ubx_e
    viol += (pos(lo[1:] - x[:, 1:]) + pos(x[:, 1:] - hi[1:])).sum(axis=(1, 2))
    tau = np.asarray(ev['tau'])[:, :N, :nq]
    viol += pos(np.abs(tau) - pr.tau_max).sum(axis=(1, 2))
    if d.n_rows:
        rv = np.asarray(ev['row_val'])[:, 1:, :d.n_rows]
        lb = np.where(np.abs(pr.row_lb) < 1e5, pr.row_lb, -np.inf)
        ub = np.where(np.abs(pr.row_ub) < 1e5, pr.row_ub, np.inf)
        viol += (pos(lb - rv) + pos(rv - ub)).sum(axis=(1, 2))
    if d.nn_mode != 0:
        g = np.asarray(ev['nn_val'])
        on = ctrl.p[:, :, 4] > 0
        on[:, 0] = False
        if d.nn_mode == 1:
            on[:, :N] = False
        viol += (pos(-g) * on).sum(1)
    return f, gq, gu, viol


def generate_guess(params, cont_name, n, make_controller=None, sqp_tol=1e-6, verbose=False, armijo=1e-4, alpha_reduction=0.7,
                   alpha_min=0.05, history=None):
    """guess_acados.py:98-158: Halton q0 in the joint box, collision filter, constant guess, SQP to convergence, checkGuess.

    SQP with merit backtracking (the reference runs acados with nlp_solver_type SQP, globalization MERIT_BACKTRACKING,
    parser.py:115-117,139): every iteration solves the engine's stage QP at the current iterate (one RTI solve from it),
    then backtracks the step length per instance on the l1 merit  f + mu |c|_1  until the Armijo condition
    merit(alpha) <= merit(0) + armijo * alpha * (grad f . d - mu |c(0)|_1)  holds (factor ``alpha_reduction``, floor
    ``alpha_min`` at which the step is taken regardless -- acados' defaults 0.7 / 0.05 [EXT-UNVERIFIED]; the reference parses
    alpha_reduction / alpha_min from config.yaml but never hands them to acados, parser.py:119-120).  The penalty mu grows
    so that the QP step is a descent direction of the merit.  All instances advance together; one batched linearisation per
    trial step length.  Returns (dict(xg [m,N+1,nx], ug [m,N,nu]) of the accepted instances in sampling order, good mask);
    ``history`` (a list) receives the per-iteration merit values [B] for inspection."""
    make_controller = make_controller or (lambda name, batch: get_controller(name, params, batch))
    ctrl = make_controller(cont_name, n)
    pr = ctrl.problem
    nq = pr.nq
    q = pr.x_min[:nq] + halton(4 * n + 16, nq) * (pr.x_max[:nq] - pr.x_min[:nq])        # guess_acados.py:100
    x_all = np.hstack([q, np.zeros_like(q)])
    free = np.asarray(ctrl.ocp_solver.check_trajectory(x_all[:, None, :], tol_x=0.0))      # guess_acados.py:109
    x0 = x_all[free][:n]
    if len(x0) < n:
        ctrl = make_controller(cont_name, len(x0))
    B = len(x0)
    ctrl.setGuess(np.repeat(x0[:, None, :], ctrl.N + 1, axis=1), np.zeros((B, ctrl.N, ctrl.nu)))
    done = np.zeros(B, bool)
    status = np.zeros(B, np.int32)
    mu = np.full(B, 10.0)
    for it in range(int(params.nlp_max_iter)):
        st = ctrl.solve(x0)
        dx, du = ctrl.x_temp - ctrl.x_guess, ctrl.u_temp - ctrl.u_guess
        step = np.maximum(np.abs(dx).reshape(B, -1).max(1), np.abs(du).reshape(B, -1).max(1))
        f0, gq, gu, c0 = merit_terms(ctrl, x0, ctrl.x_guess, ctrl.u_guess)
        gd = (gq * dx[:, :, :nq]).sum(axis=(1, 2)) + (gu * du).sum(axis=(1, 2))
        # penalty large enough for  D = grad f . d - mu |c|_1 < 0  wherever the iterate is infeasible
        need = np.where(c0 > 1e-12, 2.0 * np.maximum(gd, 0.0) / np.maximum(c0, 1e-12), 0.0)
        mu = np.minimum(np.maximum(mu, need), 1e8)
        m0 = f0 + mu * c0
        D = gd - mu * c0
        alpha = np.ones(B)
        settled = done | (st != 0)
        while True:
            xt = ctrl.x_guess + alpha[:, None, None] * dx
            ut = ctrl.u_guess + alpha[:, None, None] * du
            ft, _, _, ct = merit_terms(ctrl, x0, xt, ut)
            ok = (ft + mu * ct <= m0 + armijo * alpha * np.minimum(D, 0.0) + 1e-12 * (1.0 + np.abs(m0)))
            settled = settled | ok | (alpha <= alpha_min)
            if settled.all():
                break
            alpha = np.where(settled, alpha, np.maximum(alpha * alpha_reduction, alpha_min))
        upd = ~done & (st == 0)
        ctrl.x_guess = np.where(upd[:, None, None], xt, ctrl.x_guess)
        ctrl.u_guess = np.where(upd[:, None, None], ut, ctrl.u_guess)
        status = np.where(~done, st, status)
        if history is not None:
            history.append({'merit': np.where(upd, ft + mu * ct, m0), 'merit_before': m0.copy(), 'alpha': np.where(upd, alpha, 0.0),
                            'mu': mu.copy(), 'violation': np.where(upd, ct, c0), 'updated': upd.copy()})
        done |= (alpha * step < sqp_tol) | (st != 0)
        if verbose:
            print(f'SQP iteration {it}: {done.sum()}/{B} done, max step {step[upd].max() if upd.any() else 0:.2e}, '
                  f'min alpha {alpha[upd].min() if upd.any() else 1:.2f}')
        if done.all():
            break
    ctrl.x_temp, ctrl.u_temp = ctrl.x_guess.copy(), ctrl.u_guess.copy()
    good = ((status == 0) | (status == 2)) & ctrl.checkGuess()              # guess_acados.py:115 accepts status 0 or 2
    return {'xg': ctrl.x_guess[good], 'ug': ctrl.u_guess[good]}, good


# ---- closed loop -----------------------------------------------------------------------------------------------------------------
_STATE = ('x_guess', 'u_guess', 'fails', 'current_step', 'x_viable', 'r')


def _masked_step(ctrl, x, active):
    """controller.step on all rows, then roll back the rows that must not have stepped."""
    xp = ctrl.xp
    snap = {k: xp.copy(getattr(ctrl, k)) for k in _STATE if hasattr(ctrl, k)}
    u, abort = ctrl.step(x)
    for k, v in snap.items():
        cur = getattr(ctrl, k)
        m = active.reshape((-1,) + (1,) * (cur.ndim - 1))
        setattr(ctrl, k, xp.where(m, cur, v))
    return u, abort & active


class _Group(InPlaceState):
    """The closed loop of ONE group of instances (global indices first .. first + B): the driver's safe-abort automaton around
    the controller's ``step``.  A step has two enqueue-only halves, :meth:`part_a` (PD abort tracking, controller step) and
    :meth:`part_b` (plant, outcome tests, logs), with the step's single host decision between them (``did any instance
    raise abort?``).  On the device every state array keeps its address (InPlaceState), so after a few eager steps each half
    is captured once as a hipGraph and replayed: one graph launch instead of ~50 kernel launches per half."""

    def __init__(self, params, x_guess, u_guess, noise, control_noise, ctrl, backup, n_steps, first, callback, use_graphs,
                 collect_times=False):
        self._params, self._ctrl, self._backup, self._n_steps, self._first, self._callback = params, ctrl, backup, n_steps, first, callback
        B = x_guess.shape[0]
        xp = ctrl.xp
        self._xp, self._B = xp, B
        # stats.append(controller.getTime()) of scripts/mpc.py:239: one row of the seven time_fields per step of this group's
        # controller.  Device path: the engine keeps HIP events of its last 64 solves (smpc_get_timing_history); they are read
        # at least 32 solves late -- finished by then, so nothing waits inside the loop.
        self._collect_times = bool(collect_times) and hasattr(ctrl.ocp_solver, 'timing_history')
        self._time_rows, self._time_next, self._time_lost = [], 0, 0
        if self._collect_times:
            ctrl.ocp_solver.enable_timing(2)
        pr, nq, nx, nu = ctrl.problem, ctrl.nq, ctrl.nx, ctrl.nu
        self._Nb = backup.N
        seeds = np.arange(first, first + B)
        joints_noisy = perturbed_joint_tables(params, nq, noise, seeds) if noise > 0 else None          # mpc.py:106-107
        # model.reset_seed(i) is called at EVERY step (mpc.py:126): each instance sees the same torque-noise draw each step
        tau_noise = None
        if control_noise > 0:
            tau_noise = np.stack([np.random.default_rng(int(i)).normal(np.zeros(nu), pr.tau_max * control_noise / 100, nu)
                                  for i in seeds])
        if xp.on_device:
            if joints_noisy is not None:          # [B, nq] smpc_joint records as a float64 tensor (33 doubles each)
                joints_noisy = xp.asarray(np.ascontiguousarray(joints_noisy).view(np.float64).reshape(B, nq, -1), xp.f64)
            if tau_noise is not None:
                tau_noise = xp.asarray(tau_noise, xp.f64)
        self._joints_noisy, self._tau_noise = joints_noisy, tau_noise
        # step-major logs (one contiguous [B, .] slab per step); transposed to the reference's [B, step, .] at the end
        nan = float('nan')
        self.x_log = xp.full((n_steps + 1, B, nx), nan)
        self.u_log = xp.full((n_steps, B, nu), nan)
        self.r_log = xp.full((n_steps, B), -1, xp.i64)
        self.x_cur = xp.asarray(x_guess[:, 0], xp.f64)
        self.x_log[0] = self.x_cur
        ctrl.setGuess(x_guess, u_guess)                                                  # mpc.py:119-120
        ctrl.reset_controller()
        self.alive = xp.full((B,), True, xp.bool_)
        self.sa = xp.full((B,), False, xp.bool_)
        self.ja = xp.zeros((B,), xp.i64)
        self.x_abort = xp.zeros((B, self._Nb + 1, nx))
        self.u_abort = xp.zeros((B, self._Nb, nu))
        self.collided = xp.full((B,), False, xp.bool_)
        self.viable = xp.zeros((B,), xp.u8)            # successful abort events so far (mpc.py:189 appends once per event)
        self.resumed = xp.full((B,), False, xp.bool_)   # left the backup trajectory at this step (mpc.py:137-141)
        self._quirks = bool(getattr(params, 'reference_quirks', True))
        self.u = xp.zeros((B, nu))
        self.new_abort = xp.full((B,), False, xp.bool_)
        # last valid row of the state / input logs of every instance (mpc.py:114 pre-fills with NaN, :240-264 and :186-190 break)
        self.last_x, self.last_u = xp.full((B,), n_steps, xp.i64), xp.full((B,), n_steps - 1, xp.i64)
        self._jt = xp.step_index(0)           # the step counter (a device tensor on the torch backend: graphs replay it)
        self._zeros_u = xp.zeros((B, nu))
        self._abort_events = []
        self._ever_aborted = False
        self._use_graphs = bool(use_graphs and xp.on_device)
        self._graphs = {}                     # (half, ever_aborted) -> captured graph
        # device path: both halves of a step are engine kernels (smpc_loop_pre / smpc_policy_step / smpc_loop_post); the numpy
        # code below is the readable statement of the same automaton and what the host path runs
        self._fused = bool(xp.on_device and hasattr(ctrl.ocp_solver, 'loop_pre'))
        if self._fused:
            self._u_other = xp.zeros((B, nu))
            self._stepping = xp.full((B,), True, xp.bool_)
            self.new_abort = ctrl._abort_out          # (the controller's own output buffer: no copy per step)
            # abort events: the backup OCPs of a step's events are solved on the backup solver's own stream WHILE the next
            # step's solve runs; until their outcome is applied (smpc_loop_apply_backup) the instances are only kept from stepping
            self._pending = xp.full((B,), False, xp.bool_)
            self._resumed = xp.full((B,), False, xp.bool_)      # written by smpc_loop_pre
            self._any_event = xp.zeros((1,), xp.i32)            # written by smpc_loop_classify_aborts
            self._inflight = None
            import torch
            bs = backup.ocp_solver
            self._side = torch.cuda.ExternalStream(bs.L.smpc_stream(bs.h), device=torch.device('cuda', bs.device))
        self._inplace = xp.on_device

    # ---- first half: everything up to the controller's verdict ---------------------------------------------------------------
    def part_a(self):
        xp, ctrl, B, Nb = self._xp, self._ctrl, self._B, self._Nb
        nq = ctrl.nq
        if self._fused:
            sv = ctrl.ocp_solver
            sv.loop_pre(self, getattr(ctrl, 'r', None), self._pending, self._u_other, self._stepping)
            ctrl.step_on_device(self.x_cur, self._stepping, self._u_other, u_out=self.u)
            if ctrl.can_abort:
                sv.loop_classify_aborts(self, self.new_abort, self._any_event)
            return
        kp, kd = 1.0, 1e2                                                               # mpc.py:97
        x_cur = self.x_cur
        if self._ever_aborted:
            # --- instances following their safe-abort trajectory (mpc.py:130-146)
            in_abort = self.sa & self.alive
            follow = in_abort & (self.ja < Nb)
            idx = xp.clip_max(self.ja, Nb - 1)
            xa = xp.take_rows(self.x_abort, idx)
            ua = xp.take_rows(self.u_abort, idx)
            u_f = ua - (kp * (x_cur[:, :nq] - xa[:, :nq]) + kd * (x_cur[:, nq:] - xa[:, nq:]))
            u = xp.where(follow[:, None], u_f, self._zeros_u)
            hold = in_abort & (self.ja >= Nb)
            resume = hold & xp.all_tail(x_cur[:, nq:] < 5e-3)                         # mpc.py:138
            still = hold & ~resume
            xe = self.x_abort[:, -1]
            u_h = -(kp * (x_cur[:, :nq] - xe[:, :nq]) + 3e2 * (x_cur[:, nq:] - xe[:, nq:]))
            u = xp.where(still[:, None], u_h, u)
            self.sa = self.sa & ~resume
            self.ja = self.ja + xp.cast(in_abort, xp.i64)
            self.resumed = resume
        else:
            u = self._zeros_u
        # --- instances under MPC (mpc.py:151)
        stepping = self.alive & ~self.sa
        if hasattr(ctrl, 'r'):
            xp.put_row(self.r_log, self._jt, xp.where(stepping, ctrl.r, xp.full((B,), -1, xp.i64)))
        # (until the first abort event every live instance steps: no snapshot / roll-back needed; dead instances are never
        #  read again, so they may step along)
        if self._ever_aborted:
            u_m, ab = _masked_step(ctrl, x_cur, stepping)
        else:
            u_m, ab = ctrl.step(x_cur)
        self.u = xp.where(stepping[:, None], u_m, u)
        self.new_abort = ab & stepping
        if self._ever_aborted and self._quirks:
            # An abort raised on the very step an instance resumed MPC sits inside the reference's `if sa_flag:` branch
            # (mpc.py:137-141): no viable state is recorded and no backup OCP solved -- the instance is in safe abort again
            # with its OLD backup trajectory, its abort clock keeps running, and it re-tests its velocity at the next step.
            again = self.new_abort & self.resumed
            self.sa = self.sa | again
            self.new_abort = self.new_abort & ~again

    # ---- the step's host decision: abort events (mpc.py:161-190) -----------------------------------------------------------------
    def handle_aborts(self, j=None):
        xp, ctrl, backup, B, Nb = self._xp, self._ctrl, self._backup, self._B, self._Nb
        if not ctrl.can_abort:
            return
        if self._fused:
            return self._handle_aborts_fused(j)
        if not xp.any(self.new_abort):                                               # (the step's one host synchronisation)
            return
        self._ever_aborted = True
        # the backup OCP is solved for the aborting instances only (a compact batch), from their viable states
        rows = np.where(xp.host(self.new_abort))[0]
        j = int(xp.host(self._jt)[0]) if xp.on_device else self._jt[0]
        xv = ctrl.getLastViableState()
        rows_b = xp.asarray(rows, xp.i64)
        xv_c = xv[rows_b]
        n_c = len(rows)
        xg_c = xp.repeat_nodes(xv_c, Nb + 1)
        xo_c, uo_c, st_c, _ = backup.ocp_solver.solve(xv_c if xp.on_device else np.ascontiguousarray(xv_c), xg_c,
                                                      xp.zeros((n_c, Nb, ctrl.nu)), backup.p[:n_c])
        self._abort_events.append((rows + self._first, np.full(n_c, j), xp.host(xv_c)))
        ok_c = st_c == 0 if xp.on_device else np.asarray(st_c) == 0
        failed = xp.full((B,), False, xp.bool_)
        okb = xp.full((B,), False, xp.bool_)
        failed[rows_b] = ~ok_c
        okb[rows_b] = ok_c
        xa_new, ua_new = xp.copy(self.x_abort), xp.copy(self.u_abort)
        xa_new[rows_b] = xo_c if xp.on_device else np.asarray(xo_c)
        ua_new[rows_b] = uo_c if xp.on_device else np.asarray(uo_c)
        self.collided = self.collided | failed
        self.alive = self.alive & ~failed
        jj = xp.step_vec(self._jt, B)
        self.last_x, self.last_u = xp.where(failed, jj, self.last_x), xp.where(failed, jj, self.last_u)   # mpc.py:186-190
        self.x_abort = xp.where(okb[:, None, None], xa_new, self.x_abort)
        self.u_abort = xp.where(okb[:, None, None], ua_new, self.u_abort)
        self.ja = self.ja * xp.cast(~okb, xp.i64)
        self.sa = self.sa | okb
        self.viable = xp.cast(xp.clip_max(xp.cast(self.viable, xp.i64) + xp.cast(okb, xp.i64), 255), xp.u8)   # saturating, like k_loop_apply_backup

    def _apply_inflight(self):
        if self._inflight is None:
            return
        import torch
        rows_b, xv_c, xo_c, uo_c, st_c, ev = self._inflight
        torch.cuda.current_stream().wait_event(ev)
        self._ctrl.ocp_solver.loop_apply_backup(self, rows_b, st_c, xo_c, uo_c, self.viable, self.u, self._pending)
        self._inflight = None

    def _handle_aborts_fused(self, j):
        """Device path of :meth:`handle_aborts`.  Same events, same outcomes; what differs is WHEN the backup OCP's result is
        consumed: the solve is enqueued on the backup solver's stream and its outcome applied one step later, after the next
        controller step has been enqueued -- in between the instance only must not step (``_pending``), which is all the
        reference's loop needs of it (it either follows the backup trajectory from the next step on or is lost at this one)."""
        import torch
        xp, ctrl, backup, Nb = self._xp, self._ctrl, self._backup, self._Nb
        self._apply_inflight()                               # the previous step's events (their first tracking control goes into u)
        if not bool(self._any_event.item()):                 # (the step's one host synchronisation)
            return
        self._ever_aborted = True
        rows = np.where(xp.host(self.new_abort))[0]
        rows_b = xp.asarray(rows, xp.i64)
        xv_c = ctrl.x_viable[rows_b]
        n_c = len(rows)
        self._abort_events.append((rows + self._first, np.full(n_c, j), xv_c))      # (xv_c goes to the host in results())
        self._pending.copy_(self.new_abort)
        main = torch.cuda.current_stream()
        ev0 = torch.cuda.Event()
        ev0.record(main)
        with torch.cuda.stream(self._side):
            self._side.wait_event(ev0)
            xg_c = xp.repeat_nodes(xv_c, Nb + 1)
            xo_c, uo_c, st_c, _ = backup.ocp_solver.solve(xv_c, xg_c, xp.zeros((n_c, Nb, ctrl.nu)), backup.p[:n_c])
            ev = torch.cuda.Event()
            ev.record(self._side)
        # (These tensors cross streams.  Their lifetimes are ordered by the events, not by the caching allocator: they are released
        #  in _apply_inflight, after the main stream has been told to wait for `ev`, and the side stream's next work waits for an
        #  event recorded on the main stream after that.  torch's record_stream is not usable here -- on this ROCm build it
        #  faults on an ExternalStream.)
        self._inflight = (rows_b, xv_c, xo_c, uo_c, st_c, ev)

    # ---- second half: plant, outcome tests, logs ----------------------------------------------------------------------------------
    def part_b(self):
        xp, ctrl, B = self._xp, self._ctrl, self._B
        solver = ctrl.ocp_solver
        if self._fused:
            solver.loop_post(self, self.u, self._joints_noisy, self._tau_noise)
            return
        # plant (mpc.py:240, env_model.py:192-206).  Logs are written unmasked; rows of an instance after its failure are
        # blanked at the end from the step it died at (mpc.py:114 pre-fills them with NaN)
        xp.put_row(self.u_log, self._jt, self.u)
        x_next, _ = solver.plant_step(self.x_cur, self.u, self._joints_noisy, self._tau_noise)
        # outcome tests on the new state (mpc.py:246-264): one node per instance, so the engine's box + collision test
        # (model bounds widened by tol_x, rows against the check bounds) is exactly checkStateConstraints
        okn = solver.check_trajectory(x_next[:, None, :] if not xp.on_device else x_next[:, None, :].contiguous())
        okn = (okn != 0) if xp.on_device else np.asarray(okn)
        xp.put_row(self.x_log, self._jt, x_next, offset=1)
        bad = self.alive & ~okn
        jj = xp.step_vec(self._jt, B)
        self.last_x, self.last_u = xp.where(bad, jj + 1, self.last_x), xp.where(bad, jj, self.last_u)   # the failing state stays logged
        self.collided = self.collided | bad
        self.alive = self.alive & ~bad
        self.x_cur = xp.where(self.alive[:, None], x_next, self.x_cur)
        xp.step_advance(self._jt)

    def _run_half(self, half, fn, j):
        """eager for the first steps (workspaces get allocated), then captured once per code path and replayed"""
        if getattr(self._ctrl, '_traj_rebound', False):      # setTrajectory bound a new device tensor: captured steps hold the old one
            self._graphs.clear()
            self._ctrl._traj_rebound = False
        key = (half, self._ever_aborted)
        g = self._graphs.get(key)
        if g is None and self._use_graphs and j >= 3 and key not in self._graphs:
            import torch
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=torch.cuda.current_stream(), capture_error_mode='relaxed'):
                    fn()                  # (capturing records the launches without executing them)
                self._graphs[key] = g
            except Exception as e:        # report once, stay eager
                print(f'[run_mpc] hipGraph capture of step half {half} failed ({type(e).__name__}: {e}); eager launches')
                self._graphs[key], self._use_graphs, g = None, False, None
        if g is not None:
            g.replay()
        else:
            fn()

    def _read_times(self, j, final=False):
        """rows of the solves up to step j whose events have finished (device path: never the 32 most recent unless ``final``)"""
        sv = self._ctrl.ocp_solver
        if final:
            sv.sync()                              # end of the run: everything has finished, drain the ring
        while self._time_next <= j:
            back = j - self._time_next
            if back >= 64:                         # the ring has lapped it (the host ran more than 64 solves ahead of this read)
                self._time_lost += 1
                self._time_next += 1
                continue
            if not final and back < 32:
                break
            if not final and back >= 56:
                sv.sync()                          # about to be lapped: wait rather than lose the row
            tm = sv.timing_history(back)
            if tm is None:
                if final:
                    self._time_lost += 1
                    self._time_next += 1
                    continue
                break
            self._time_rows.append(time_row(tm))
            self._time_next += 1

    def run(self):
        """generator: yields once per step, after the first half is enqueued (run_mpc serves the other groups meanwhile)"""
        try:
            yield from self._run()
        finally:
            # whatever path ran (device loop, host state on the HIP engine, an exception in between): the handle stops recording
            # events when the run is over
            sv = self._ctrl.ocp_solver
            if self._collect_times and hasattr(sv, 'enable_timing'):
                try:
                    sv.enable_timing(0)
                except Exception:
                    pass

    def _run(self):
        xp = self._xp
        for j in range(self._n_steps):
            self._run_half('a', self.part_a, j)
            if self._collect_times:
                if xp.on_device:
                    self._read_times(j)
                else:                              # host path: the solve has finished when step() returns
                    self._time_rows.append(self._ctrl.getTime())
            yield j
            self.handle_aborts(j)
            self._run_half('b', self.part_b, j)
            if self._callback and j % 50 == 0:
                print(f'step {j} (instances {self._first}..{self._first + self._B - 1}): alive {int(xp.host(self.alive).sum())}/{self._B}, '
                      f'in abort {int(xp.host(self.sa & self.alive).sum())}, failures {int(xp.host(self.collided).sum())}')
            if not xp.on_device and not self.alive.any():
                break
        if self._fused:
            self._apply_inflight()            # (events of the last step)
        if self._collect_times and xp.on_device:
            self._read_times(self._n_steps - 1, final=True)

    def results(self):
        xp, ctrl, params, B, n_steps = self._xp, self._ctrl, self._params, self._B, self._n_steps
        solver, pr = ctrl.ocp_solver, ctrl.problem
        if xp.on_device:
            solver.sync()
        # convergence at the last step (mpc.py:273): the reference tests x_sim[-1], NaN for instances that broke
        x_sim = np.ascontiguousarray(np.transpose(xp.host(self.x_log), (1, 0, 2)))
        u_sim = np.ascontiguousarray(np.transpose(xp.host(self.u_log), (1, 0, 2)))
        steps = np.arange(n_steps + 1)[None, :]
        x_sim[steps > xp.host(self.last_x)[:, None]] = np.nan
        u_sim[steps[:, :n_steps] > xp.host(self.last_u)[:, None]] = np.nan
        x_last = np.nan_to_num(x_sim[:, -1])
        ev = solver.eval_nodes(xp.repeat_nodes(xp.asarray(x_last, xp.f64), ctrl.N + 1), xp.zeros((B, ctrl.N, ctrl.nu)), ctrl.p)
        ee = xp.host(ev['ee'])[:, 0, :]
        conv = ~np.isnan(x_sim[:, -1]).any(1) & (np.linalg.norm(ee - pr.ee_ref, axis=1) < params.tol_conv)
        if not self._quirks:
            conv &= xp.host(self.alive)        # (the reference tests x_sim[-1] even of an instance it has just recorded as failed)
        return dict(x=x_sim, u=u_sim, r_receding=np.transpose(xp.host(self.r_log), (1, 0))[:, :, None], conv=conv,
                    time_rows=np.array(self._time_rows, float).reshape(-1, len(TIME_FIELDS)), time_lost=self._time_lost,
                    collided=xp.host(self.collided), viable=xp.host(self.viable).astype(np.int64),
                    abort_events=[(e[0], e[1], e[2] if isinstance(e[2], np.ndarray) else xp.host(e[2])) for e in self._abort_events])


TIME_FIELDS = ('time_lin', 'time_sim', 'time_qp', 'time_qp_solver_call', 'time_glob', 'time_reg', 'time_tot')   # controller.py:123-124


def time_row(t):
    """the engine's per-kernel times of one solve (seconds) as the reference's seven acados fields (controller.py:192-193):
    time_lin = linearisation + network pass (acados evaluates the l4casadi row inside its linearisation), time_qp = stage-QP
    set-up + interior point, time_qp_solver_call = the interior point alone; nothing of the kind of time_sim / time_glob /
    time_reg runs here (closed-form dynamics, FIXED_STEP, LM folded into the set-up)"""
    qp_setup, qp_ipm = t.get('time_qp_setup', 0.0), t.get('time_qp_ipm', 0.0)
    qp = t['time_qp'] if 'time_qp' in t else qp_setup + qp_ipm
    return [t.get('time_lin', 0.0) + t.get('time_nn', 0.0), 0.0, qp, qp_ipm, 0.0, 0.0, t.get('time_tot', 0.0)]


def run_mpc(params, cont_name, x_guess, u_guess, noise=0.0, control_noise=0.0, make_controller=None, make_backup=None,
            n_steps=None, callback=False, on_device=False, device=0, timing=None, groups=None, graphs=True, collect_times=False):
    """scripts/mpc.py:102-317 for all instances at once.  Returns the result dict the reference pickles (mpc.py:307-315).

    ``on_device=True``: the whole loop state -- the policy automaton of the controller, the safe-abort automaton of the driver
    (flags, abort clocks, backup trajectories, PD tracking), trajectories and outcome masks -- lives in HBM as torch tensors
    and every engine call takes the device-pointer path.  The only host synchronisation inside a step is one scalar per
    group for the policies that can abort: "did any instance raise abort in this step?" (the backup OCP is solved only then,
    as in the reference, mpc.py:161-190, and only for the aborting instances).  The instances are split into ``groups``
    independent sub-batches (default 2 on the device), each with its own engine handle and HIP stream, advanced in turn:
    while the host waits for one group's flag the others' kernels keep the GPU busy, and the long tail of one group's QP
    launch overlaps the bulk of another's.  With ``graphs`` each half of a step (before / after that flag) is captured as a
    hipGraph after a few eager steps and replayed.  ``timing``: optional dict that receives wall-clock ms per step.
    ``collect_times``: append the controller's solver times at every step like ``stats.append(controller.getTime())`` of
    scripts/mpc.py:239; the result then carries ``'time_stats'`` (one row of ``TIME_FIELDS`` per step and group, seconds) and
    ``'time_q99'`` (the 99 % quantiles mpc.py:300-303 prints).  A captured graph cannot record into the engine's event ring,
    so this runs the steps as eager launches."""
    import time
    B = x_guess.shape[0]
    n_steps = int(n_steps if n_steps is not None else params.n_steps)
    if on_device:
        make_controller = make_controller or (lambda name, batch: get_controller(name, params, batch, device=device, device_state=True))
        make_backup = make_backup or (lambda batch: SafeBackupController(params, batch, device=device, device_state=True))
    else:
        make_controller = make_controller or (lambda name, batch: get_controller(name, params, batch))
        make_backup = make_backup or (lambda batch: SafeBackupController(params, batch))
    if collect_times:
        graphs = False
    if groups is None:
        groups = max(1, min(2, B // 512)) if on_device else 1      # measured at B = 4096 (MI355X, r2): 1: 5.1, 2: 4.55, 3: 5.8 ms/step
    from .sharding import shard_range
    spans = [shard_range(B, groups, g) for g in range(groups)]
    gens, grps, streams = [], [], []
    for lo, hi in spans:
        ctrl = make_controller(cont_name, hi - lo)
        backup = make_backup(hi - lo)
        if on_device:
            import torch
            sv = ctrl.ocp_solver
            streams.append(torch.cuda.ExternalStream(sv.L.smpc_stream(sv.h), device=torch.device('cuda', sv.device)))
            with torch.cuda.stream(streams[-1]):
                grp = _Group(params, x_guess[lo:hi], u_guess[lo:hi], noise, control_noise, ctrl, backup, n_steps, lo, callback, graphs,
                             collect_times)
        else:
            streams.append(None)
            grp = _Group(params, x_guess[lo:hi], u_guess[lo:hi], noise, control_noise, ctrl, backup, n_steps, lo, callback, False,
                         collect_times)
        grps.append(grp)
        gens.append(grp.run())

    def advance(g):
        """runs group g up to its next yield (end of the enqueue phase of a step); False once the group is finished"""
        try:
            if streams[g] is not None:
                import torch
                with torch.cuda.stream(streams[g]):      # torch ops and engine calls of a group share the handle's stream
                    next(gens[g])
            else:
                next(gens[g])
            return True
        except StopIteration:
            return False

    live = [True] * groups
    t_loop, j_loop, j = time.perf_counter(), 0, 0
    warm = min(4, n_steps // 2)
    while any(live):
        for g in range(groups):
            if live[g]:
                live[g] = advance(g)
        j += 1
        if timing is not None and j == warm + 1:          # the first steps allocate workspaces: time the rest
            if on_device:
                import torch
                torch.cuda.synchronize()
            t_loop, j_loop = time.perf_counter(), j
    if timing is not None:
        if on_device:
            import torch
            torch.cuda.synchronize()
        done = max(n_steps + 1 - j_loop, 1)
        timing['ms_per_step'] = 1e3 * (time.perf_counter() - t_loop) / done
        timing['steps'], timing['groups'] = done, groups

    # merge the groups (global instance indices)
    outs = []
    for g, grp in enumerate(grps):
        if streams[g] is not None:
            import torch
            with torch.cuda.stream(streams[g]):
                outs.append(grp.results())
        else:
            outs.append(grp.results())
    x_sim = np.concatenate([o['x'] for o in outs], axis=0)
    conv = np.concatenate([o['conv'] for o in outs])
    collided = np.concatenate([o['collided'] for o in outs])
    viable = np.concatenate([o['viable'] for o in outs])
    conv_idx = np.where(conv)[0].tolist()
    coll_idx = np.where(collided)[0].tolist()
    cs, ks = set(conv_idx), set(coll_idx)
    if getattr(params, 'reference_quirks', True):
        # mpc.py:189 appends an instance to viable_idx at EVERY abort event and :277-278 removes it ONCE when it converges: one that
        # went through two events and converged is in both lists; a failure recorded at the last step does not stop the
        # convergence test of :273 either (results(): conv is taken from x_sim[-1] alone)
        viable_idx = [int(i) for i in np.where(viable - conv.astype(np.int64) > 0)[0] if i not in ks]
    else:
        viable_idx = [int(i) for i in np.where(viable)[0] if i not in cs and i not in ks]
    unconv_idx = sorted(set(range(B)) - cs - ks - set(viable_idx))
    # x_viable: one row per abort event, in the order the reference's loops produce them (instance-major, then time: mpc.py:102,125)
    ev = [e for o in outs for e in o['abort_events']]
    if ev:
        inst, step, xv = np.concatenate([e[0] for e in ev]), np.concatenate([e[1] for e in ev]), np.concatenate([e[2] for e in ev])
        x_viable = xv[np.lexsort((step, inst))]
    else:
        x_viable = np.zeros((0, x_sim.shape[2]))
    # 'r': the reference allocates r_index as NaN and never writes it (mpc.py:116, 281) -- kept NaN for format parity; the
    # receding index actually used at every step is returned next to it as 'r_receding' (-1 where the policy has none)
    extra = {}
    if collect_times:
        ts = np.concatenate([o['time_rows'] for o in outs], axis=0)
        extra = {'time_stats': ts, 'time_fields': list(TIME_FIELDS), 'time_lost': int(sum(o['time_lost'] for o in outs)),
                 'time_q99': np.quantile(ts, 0.99, axis=0) if len(ts) else np.zeros(len(TIME_FIELDS))}
    return {**extra, 'x': x_sim, 'u': np.concatenate([o['u'] for o in outs], axis=0),
            'r': np.full((B, n_steps, 1), np.nan), 'r_receding': np.concatenate([o['r_receding'] for o in outs], axis=0),
            'conv_idx': conv_idx, 'collisions_idx': coll_idx, 'unconv_idx': unconv_idx, 'viable_idx': viable_idx,
            'x_viable': x_viable}


def save_pickle(path, obj):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, 'wb') as f:
        pickle.dump(obj, f)
