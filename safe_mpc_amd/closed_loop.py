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
    for row, seed in enumerate(seeds):
        rng = np.random.default_rng(int(seed))
        links = []
        for l in base.links:
            l2 = copy.copy(l)
            if l.inertial is not None:
                m = l.inertial.mass
                m = m + rng.uniform(-abs(m) * noise_pct / 100, abs(m) * noise_pct / 100) if noise_pct > 0 else m
                I = l.inertial.inertia.copy()
                for (a, b) in [(0, 0), (1, 1), (2, 2), (0, 1), (1, 2), (0, 2)]:       # ixx iyy izz ixy iyz ixz
                    n_ = abs(I[a, b]) * noise_pct / 100
                    v = I[a, b] + (rng.uniform(-n_, n_) if n_ > 0 else 0.0)
                    I[a, b] = I[b, a] = v
                xyz = l.inertial.origin.xyz.copy()
                for k in range(3):
                    n_ = abs(xyz[k] * noise_pct / 100)
                    xyz[k] += rng.uniform(-n_, n_) if n_ > 0 else 0.0
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


# ---- warm-start generation ----------------------------------------------------------------------------------------------------
def generate_guess(params, cont_name, n, make_controller=None, sqp_tol=1e-6, verbose=False):
    """guess_acados.py:98-158: Halton q0 in the joint box, collision filter, constant guess, SQP to convergence, checkGuess.

    The reference runs acados' SQP with merit backtracking (parser.py:117,139); here the SQP loop is the engine's RTI step
    applied repeatedly with full steps until the iterate stops moving (documented deviation: no line search).
    Returns dict(xg [m,N+1,nx], ug [m,N,nu]) of the accepted instances, in sampling order.
    """
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
    for it in range(int(params.nlp_max_iter)):
        st = ctrl.solve(x0)
        step = np.maximum(np.abs(ctrl.x_temp - ctrl.x_guess).reshape(B, -1).max(1),
                          np.abs(ctrl.u_temp - ctrl.u_guess).reshape(B, -1).max(1))
        upd = ~done
        ctrl.x_guess = np.where(upd[:, None, None], ctrl.x_temp, ctrl.x_guess)
        ctrl.u_guess = np.where(upd[:, None, None], ctrl.u_temp, ctrl.u_guess)
        status = np.where(upd, st, status)
        done |= (step < sqp_tol) | (st != 0)
        if verbose:
            print(f'SQP iteration {it}: {done.sum()}/{B} done, max step {step[upd].max() if upd.any() else 0:.2e}')
        if done.all():
            break
    ctrl.x_temp, ctrl.u_temp = ctrl.x_guess.copy(), ctrl.u_guess.copy()
    good = (status == 0) & done & ctrl.checkGuess()
    return {'xg': ctrl.x_guess[good], 'ug': ctrl.u_guess[good]}, good


# ---- closed loop -----------------------------------------------------------------------------------------------------------------
_STATE = ('x_guess', 'u_guess', 'fails', 'current_step', 'x_viable', 'r')


def _masked_step(ctrl, x, active):
    """controller.step on all rows, then roll back the rows that must not have stepped."""
    snap = {k: getattr(ctrl, k).copy() for k in _STATE if hasattr(ctrl, k)}
    u, abort = ctrl.step(x)
    for k, v in snap.items():
        cur = getattr(ctrl, k)
        m = active.reshape((-1,) + (1,) * (cur.ndim - 1))
        setattr(ctrl, k, np.where(m, cur, v))
    return u, abort & active


def run_mpc(params, cont_name, x_guess, u_guess, noise=0.0, control_noise=0.0, make_controller=None, make_backup=None,
            n_steps=None, callback=False):
    """scripts/mpc.py:102-317 for all instances at once.  Returns the result dict the reference pickles (mpc.py:307-315)."""
    B = x_guess.shape[0]
    n_steps = int(n_steps if n_steps is not None else params.n_steps)
    make_controller = make_controller or (lambda name, batch: get_controller(name, params, batch))
    make_backup = make_backup or (lambda batch: SafeBackupController(params, batch))
    ctrl = make_controller(cont_name, B)
    backup = make_backup(B)
    pr, nq, nx, nu = ctrl.problem, ctrl.nq, ctrl.nx, ctrl.nu
    kp, kd = 1.0, 1e2                                                               # mpc.py:97
    Nb = backup.N

    joints_noisy = perturbed_joint_tables(params, nq, noise, np.arange(B)) if noise > 0 else None   # mpc.py:106-107
    # model.reset_seed(i) is called at EVERY step (mpc.py:126): each instance sees the same torque-noise draw each step
    tau_noise = None
    if control_noise > 0:
        tau_noise = np.stack([np.random.default_rng(i).normal(np.zeros(nu), pr.tau_max * control_noise / 100, nu)
                              for i in range(B)])

    x_sim = np.full((B, n_steps + 1, nx), np.nan)
    u_log = np.full((B, n_steps, nu), np.nan)
    x_sim[:, 0] = x_guess[:, 0]
    ctrl.setGuess(x_guess, u_guess)                                                  # mpc.py:119-120
    ctrl.reset_controller()
    alive = np.ones(B, bool)
    sa = np.zeros(B, bool)
    ja = np.zeros(B, np.int64)
    x_abort = np.zeros((B, Nb + 1, nx))
    u_abort = np.zeros((B, Nb, nu))
    collisions, viable, x_viable_log = set(), set(), []
    x_cur = x_sim[:, 0].copy()
    r_log = np.full((B, n_steps, 1), -1, np.int64)

    for j in range(n_steps):
        u = np.zeros((B, nu))
        # --- instances following their safe-abort trajectory (mpc.py:130-146)
        in_abort = sa & alive
        follow = in_abort & (ja < Nb)
        if follow.any():
            idx = np.minimum(ja, Nb - 1)
            xa = x_abort[np.arange(B), idx]
            ua = u_abort[np.arange(B), idx]
            u_f = ua - (kp * (x_cur[:, :nq] - xa[:, :nq]) + kd * (x_cur[:, nq:] - xa[:, nq:]))
            u = np.where(follow[:, None], u_f, u)
        hold = in_abort & (ja >= Nb)
        resume = hold & np.all(x_cur[:, nq:] < 5e-3, axis=1)                        # mpc.py:138
        still = hold & ~resume
        if still.any():
            xe = x_abort[:, -1]
            u_h = -(kp * (x_cur[:, :nq] - xe[:, :nq]) + 3e2 * (x_cur[:, nq:] - xe[:, nq:]))
            u = np.where(still[:, None], u_h, u)
        sa = sa & ~resume
        ja = ja + in_abort
        # --- instances under MPC (mpc.py:151)
        stepping = alive & ~sa
        if stepping.any():
            if hasattr(ctrl, 'r'):
                r_log[:, j, 0] = np.where(stepping, ctrl.r, -1)
            u_m, ab = _masked_step(ctrl, x_cur, stepping)
            u = np.where(stepping[:, None], u_m, u)
            new_abort = ab & stepping
            if new_abort.any():                                                     # mpc.py:161-190
                xv = ctrl.getLastViableState()
                backup.setGuess(np.repeat(xv[:, None, :], Nb + 1, axis=1), np.zeros((B, Nb, nu)))
                st_b = backup.solve(xv)
                failed = new_abort & (st_b != 0)
                okb = new_abort & (st_b == 0)
                for i in np.where(new_abort)[0]:
                    x_viable_log.append(xv[i].copy())
                for i in np.where(failed)[0]:
                    collisions.add(int(i))
                alive &= ~failed
                x_abort = np.where(okb[:, None, None], backup.x_temp, x_abort)
                u_abort = np.where(okb[:, None, None], backup.u_temp, u_abort)
                ja = np.where(okb, 0, ja)
                sa = sa | okb
                for i in np.where(okb)[0]:
                    viable.add(int(i))
        # --- plant (mpc.py:240, env_model.py:192-206)
        u_log[:, j] = np.where(alive[:, None], u, np.nan)
        x_next, _ = ctrl.ocp_solver.plant_step(x_cur, np.where(alive[:, None], u, 0.0), joints_noisy, tau_noise)
        # --- outcome tests on the new state (mpc.py:246-264)
        tol = params.tol_x
        in_box = np.all((x_next >= pr.x_min - tol) & (x_next <= pr.x_max + tol), axis=1)
        free = np.asarray(ctrl.ocp_solver.check_trajectory(x_next[:, None, :], tol_x=1e30))
        bad = alive & ~(in_box & free)
        x_sim[:, j + 1] = np.where(alive[:, None], x_next, np.nan)
        for i in np.where(bad)[0]:
            collisions.add(int(i))
        alive &= ~bad
        x_cur = np.where(alive[:, None], x_next, x_cur)
        if callback and j % 50 == 0:
            print(f'step {j}: alive {alive.sum()}/{B}, in abort {int((sa & alive).sum())}, failures {len(collisions)}')
        if not alive.any():
            break

    # convergence at the last step (mpc.py:273): the reference tests x_sim[-1], NaN for instances that broke
    ev = ctrl.ocp_solver.eval_nodes(np.repeat(np.nan_to_num(x_sim[:, -1])[:, None, :], ctrl.N + 1, 1),
                                    np.zeros((B, ctrl.N, nu)), ctrl.p)
    ee = ev['ee'][:, 0, :]
    conv_mask = alive & ~np.isnan(x_sim[:, -1]).any(1) & (np.linalg.norm(ee - pr.ee_ref, axis=1) < params.tol_conv)
    conv_idx = np.where(conv_mask)[0].tolist()
    viable -= set(conv_idx)
    viable_idx = sorted(i for i in viable if i not in collisions)
    coll_idx = sorted(collisions)
    unconv_idx = sorted(set(range(B)) - set(conv_idx) - set(coll_idx) - set(viable_idx))
    # 'r': the reference allocates r_index as NaN and never writes it (mpc.py:116, 281) -- kept NaN for format parity; the
    # receding index actually used at every step is returned next to it as 'r_receding' (-1 where the policy has none)
    return {'x': x_sim, 'u': u_log, 'r': np.full((B, n_steps, 1), np.nan), 'r_receding': r_log, 'conv_idx': conv_idx,
            'collisions_idx': coll_idx, 'unconv_idx': unconv_idx, 'viable_idx': viable_idx,
            'x_viable': np.asarray(x_viable_log)}


def save_pickle(path, obj):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, 'wb') as f:
        pickle.dump(obj, f)
