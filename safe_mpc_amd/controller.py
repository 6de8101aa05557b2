"""Batched mirrors of the reference's controller classes (src/safe_mpc/controller.py:251-712).

Same class names, same method names (``initialize``, ``step``, ``setGuess``/``getGuess``, ``resetHorizon``,
``reset_controller``, ``getTime``, ``getLastViableState``, ``checkGuess``, ``solve``, ``provideControl``,
``guessCorrection``), same accept / reject / abort automata -- but every piece of per-instance state is an array over
the B instances that the reference walks one at a time (scripts/mpc.py:102).  ``step(x)`` takes ``x[B, nx]`` and returns
``(u[B, nu], abort[B])``.  The numerics all go through ``self.ocp_solver`` (a :class:`BatchedOcpSolver`, i.e. the HIP
engine); this file is pure bookkeeping, so it can be unit-tested with a scripted fake solver.

Instances whose ``step`` returns ``abort=True`` keep their guess unshifted and their step counter unchanged, exactly
like the early ``return self.u_guess[0], True`` of the reference (controller.py:384-385, 483-487).
"""
from __future__ import annotations

import numpy as np

from ._xp import InPlaceState, NumpyOps, TorchOps
from .problem import OcpProblem
from .safe_set import SafeSetNet


class AbstractController(InPlaceState):
    """``device_state=True`` keeps every per-instance array of the policy (guesses, counters, receding indices, viable states,
    per-node parameters) as ROCm torch tensors and drives the engine through its device-pointer path: ``step`` then enqueues
    kernels only -- no array crosses PCIe and nothing synchronises with the host.  The default (numpy) is the host path."""
    cont_name = 'naive'
    can_abort = False            # policies whose step() may return abort=True (the driver then needs one flag per step)
    policy_kind = 0              # SMPC_POLICY_* of include/smpc.h: which automaton smpc_policy_step runs for this class

    def __init__(self, params, batch, cost='ext', N=None, solver=None, net=None, device=0, device_state=False):
        self.params = params
        self.B = int(batch)
        self.xp = TorchOps(device) if device_state else NumpyOps()
        self.problem = OcpProblem(params, self.cont_name, cost, N=N)
        self.model = self.problem                      # x_min/x_max/tau_max/ee_ref live here (AdamModel's role)
        self.N = self.problem.N
        self.nx, self.nu, self.nq = self.problem.nx, self.problem.nu, self.problem.nq
        if net is None and params.use_net:
            net = SafeSetNet.from_params(params, self.problem.x_min, self.problem.x_max)
        self.net = net
        if net is not None:
            self.problem.set_normalisation(net.mean, net.std)
        if solver is None:
            from .solver import BatchedOcpSolver
            solver = BatchedOcpSolver(self.problem, net, device=device)
        self.ocp_solver = solver
        self.time_fields = ['time_lin', 'time_sim', 'time_qp', 'time_qp_solver_call', 'time_glob', 'time_reg', 'time_tot']
        xp = self.xp
        self._x_min, self._x_max = xp.asarray(self.problem.x_min, xp.f64), xp.asarray(self.problem.x_max, xp.f64)
        self._tau_min, self._tau_max = xp.asarray(self.problem.tau_min, xp.f64), xp.asarray(self.problem.tau_max, xp.f64)
        self._lbx_e, self._ubx_e = xp.asarray(self.problem.lbx_e, xp.f64), xp.asarray(self.problem.ubx_e, xp.f64)
        self._alloc()
        self.reset_controller()
        self._inplace = bool(device_state)       # device mode: state arrays keep their addresses (see _xp.InPlaceState)

    # -- state --------------------------------------------------------------------------------------------------------
    def _alloc(self):
        B, N, xp = self.B, self.N, self.xp
        self.x_guess = xp.zeros((B, N + 1, self.nx))
        self.u_guess = xp.zeros((B, N, self.nu))
        self.x_temp, self.u_temp = xp.copy(self.x_guess), xp.copy(self.u_guess)
        # per-node parameters p = [ee_ref(3), alpha, flag] (controller.py:27-37, 153-156)
        self.p = xp.zeros((B, N + 1, 5))
        self.p[:, :, :3] = xp.asarray(self.problem.ee_ref, xp.f64)
        self.p[:, :, 3] = self.params.alpha
        self.p[:, :, 4] = 1.0
        self.traj = getattr(self, 'traj', None)                    # reference trajectory (setTrajectory); None = constant ee_ref
        self.last_status = xp.full((B,), 4, xp.i32)                # controller.py:125
        self.qp_iter = xp.zeros((B,), xp.i32)
        self.x_viable = xp.zeros((B, self.nx))
        if xp.on_device:       # outputs of the fused device step (smpc_policy_step): fixed addresses, so a step can be graph-captured
            self._u_out = xp.zeros((B, self.nu))
            self._abort_out = xp.full((B,), False, xp.bool_)
            self._any_abort = xp.zeros((1,), xp.i32)

    def reset_controller(self):
        self.fails = self.xp.zeros((self.B,), self.xp.i64)
        self.current_step = self.xp.zeros((self.B,), self.xp.i64)

    def setReference(self, ee_ref):
        self.problem.ee_ref = np.asarray(ee_ref, float)
        self.p[:, :, :3] = self.xp.asarray(self.problem.ee_ref, self.xp.f64)

    def setTrajectory(self, traj):
        """``cost.traj`` of the reference (cost_definition.py:30-31, 89): ``[3, n_steps + 1 + N]`` reference points, of which
        ``solve`` hands node i the column ``current_step + i`` (controller.py:153-156).  ``None`` (the default) = the constant
        ``ee_ref`` of the ReachTarget / Zero costs, for which the indexing changes nothing."""
        if traj is None:
            self.traj = None
            self.p[:, :, :3] = self.xp.asarray(self.problem.ee_ref, self.xp.f64)
            return
        traj = np.ascontiguousarray(traj, float)
        if traj.ndim != 2 or traj.shape[0] != 3 or traj.shape[1] < 1:
            raise ValueError('traj must be [3, n_columns]')
        # the reference indexes cost.traj[:, current_step + i] for i <= N over a run of n_steps steps and would raise on a shorter
        # array; here the column index is clamped (the device kernel must not read past the end), so say so instead of silently
        # holding the last point
        need = int(getattr(self.params, 'n_steps', 0)) + 1 + self.N
        if traj.shape[1] < need:
            import warnings
            warnings.warn(f'setTrajectory: {traj.shape[1]} columns < n_steps + 1 + N = {need}: the reference would fail with an '
                          'IndexError past the end; this engine holds the last column', RuntimeWarning, stacklevel=2)
        new = self.xp.asarray(traj, self.xp.f64)
        if self.traj is not None and tuple(self.traj.shape) == tuple(new.shape):
            self.traj[...] = new       # in place: a captured step (hipGraph) keeps the device pointer and length it was captured with
        else:
            self.traj = new
            self._traj_rebound = True  # (closed_loop._Group drops its captured graphs when it sees this)

    def _apply_traj(self, rows=None):
        """p[b, i, 0:3] = traj[:, current_step[b] + i] (controller.py:153-156)"""
        if self.traj is None:
            return
        xp = self.xp
        col = xp.clip_max(self.current_step[:, None] + xp.arange(self.N + 1)[None, :], self.traj.shape[1] - 1)     # [B, N+1]
        ref = xp.swap_last(self.traj[:, col])                  # [3, B, N+1] -> [B, N+1, 3]
        if rows is None:
            self.p[:, :, :3] = ref
        else:
            self.p[:, :, :3] = xp.where(rows[:, None, None], ref, self.p[:, :, :3])

    def setGuess(self, x_guess, u_guess):
        self.x_guess = self.xp.asarray(x_guess, self.xp.f64)
        self.u_guess = self.xp.asarray(u_guess, self.xp.f64)

    def getGuess(self):
        return self.xp.copy(self.x_guess), self.xp.copy(self.u_guess)

    def getLastViableState(self):
        return self.xp.copy(self.x_viable)

    def getTime(self):
        """controller.py:192-193: the seven acados timers of the last solve, in seconds (zeros when timing is off)"""
        from .closed_loop import time_row
        try:
            t = self.ocp_solver.timing_history(0) if hasattr(self.ocp_solver, 'timing_history') else self.ocp_solver.timing()
        except Exception:
            t = None
        return np.array(time_row(t)) if t else np.zeros(len(self.time_fields))

    def resetHorizon(self, N):
        """controller.py:205-214: new horizon without re-creating the solver."""
        self.N = int(N)
        self.params.N = self.N
        self.problem.N = self.N
        self.ocp_solver.set_horizon(self.N)
        self._alloc()

    # -- the solve and its neighbours -----------------------------------------------------------------------------------
    def solve(self, x0):
        """controller.py:136-167 for all instances: returns status[B]; x_temp / u_temp hold the iterate regardless."""
        self.p[:, :, 3] = self.params.alpha
        self._apply_traj()
        if not self.xp.on_device:
            x0 = np.asarray(x0, float)
        if self.xp.on_device:     # the engine writes straight into the controller's persistent buffers
            self.ocp_solver.solve(x0, self.x_guess, self.u_guess, self.p, out=(self.x_temp, self.u_temp, self.last_status, self.qp_iter))
            return self.last_status
        x, u, st, it = self.ocp_solver.solve(x0, self.x_guess, self.u_guess, self.p)
        self.x_temp, self.u_temp = np.asarray(x), np.asarray(u)
        self.last_status = np.asarray(st).copy()
        self.qp_iter = np.asarray(it).copy()
        return self.last_status

    def guessCorrection(self):
        self.x_guess = self.ocp_solver.guess_correction(self.x_guess, self.u_guess)

    def provideControl(self, active=None):
        """controller.py:169-184.  ``active`` masks out instances that returned early (abort)."""
        xp = self.xp
        accept = xp.cast(self.fails == 0, xp.i32)
        keep = active is not None
        # the engine's device path shifts the guess in place: keep the old one for the instances that must not shift
        old_xg = xp.copy(self.x_guess) if keep and xp.on_device else self.x_guess
        old_ug = xp.copy(self.u_guess) if keep and xp.on_device else self.u_guess
        xg, ug, u = self.ocp_solver.provide_control(accept, self.x_temp, self.u_temp, self.x_guess, self.u_guess)
        if not keep:
            self.x_guess, self.u_guess = xg, ug
        else:
            u = xp.where(active[:, None], u, old_ug[:, 0])
            self.x_guess = xp.where(active[:, None, None], xg, old_xg)
            self.u_guess = xp.where(active[:, None, None], ug, old_ug)
        return u, xp.zeros((self.B,), xp.bool_)

    # -- feasibility predicates (env_model.py:170-243, safe_set.py:61-68) ------------------------------------------------
    def _as_state(self, x):
        if not self.xp.on_device:
            x = np.asarray(x, float)
        return x[:, None, :] if x.ndim == 2 else x

    def checkStateConstraints(self, x):
        """env_model.py:170-173.  Reference quirk kept by default (``params.reference_quirks``):
        ``checkCollision`` returns inside its loop after the FIRST row (env_model.py:238-243), so a trajectory is
        collision-checked at its first node only, while the box test covers every node."""
        xp = self.xp
        x = self._as_state(x)
        tol = self.params.tol_x
        in_box = xp.all_tail((x >= self._x_min - tol) & (x <= self._x_max + tol))
        xc = x[:, :1] if getattr(self.params, 'reference_quirks', True) else x
        if xp.on_device:
            xc = xc.contiguous()
            free = self.ocp_solver.check_trajectory(xc, tol_x=1e30) != 0
        else:
            free = np.asarray(self.ocp_solver.check_trajectory(np.ascontiguousarray(xc), tol_x=1e30))
        return in_box & free

    def checkSafeConstraints(self, x):
        """x[B, n_nodes, nx] (or [B, nx]) -> bool per (instance, node)."""
        squeeze = x.ndim == 2
        x = self._as_state(x)
        if self.xp.on_device:
            _, nn = self.ocp_solver.check_trajectory(x.contiguous(), want_nn=True)
            nn = nn != 0
        else:
            _, nn = self.ocp_solver.check_trajectory(x, want_nn=True)
            nn = np.asarray(nn)
        return nn[:, 0] if squeeze else nn

    def checkTorqueConstraints(self, x, u):
        ev = self.ocp_solver.eval_nodes(x, u, self.p)
        tau = ev['tau'][:, :self.N, :self.nq]
        tol = self.params.tol_tau
        return self.xp.all_tail((tau >= self._tau_min - tol) & (tau <= self._tau_max + tol))

    def checkDynamicsConstraints(self, x, u):
        """env_model.py:226-234 with the controller's own model (no torque saturation inside the rollout)."""
        sim = self.ocp_solver.guess_correction(self.xp.copy(x), u)       # (the device path integrates in place)
        n = u.shape[1]
        return self.xp.norm_tail(x - sim) < self.params.tol_dyn * np.sqrt(n + 1)

    def checkGuess(self):
        return (self.checkStateConstraints(self.x_temp) & self.checkTorqueConstraints(self.x_temp, self.u_temp) &
                self.checkDynamicsConstraints(self.x_temp, self.u_temp))

    def initialize(self, x0, u0=None):
        """controller.py:260-272: trivial guess, one solve, keep it where it checks out.  Returns 1/0 per instance."""
        xp = self.xp
        x0 = xp.asarray(x0, xp.f64)
        self.x_guess = xp.repeat_nodes(x0, self.N + 1)
        self.u_guess = xp.zeros((self.B, self.N, self.nu)) if u0 is None else xp.repeat_nodes(xp.asarray(u0, xp.f64), self.N)
        status = self.solve(x0)
        good = (status == 0) & self.checkGuess()
        self.x_guess = xp.where(good[:, None, None], self.x_temp, self.x_guess)
        self.u_guess = xp.where(good[:, None, None], self.u_temp, self.u_guess)
        return xp.cast(good, xp.i64)

    def step(self, x):
        raise NotImplementedError

    def step_on_device(self, x, stepping=None, u_other=None, u_out=None):
        """``step`` with the state in HBM: the class's automaton as engine kernels (smpc_policy_step, kernels_policy.hpp) -- the
        numpy ``step`` of each class is the readable statement of the same thing, and the GPU tests run the two side by side.
        ``stepping`` masks out instances that must not step (they are left untouched and get ``u_other``)."""
        return self.ocp_solver.policy_step(self, x, stepping, u_other, u_out)


class NaiveController(AbstractController):
    cont_name = 'naive'

    def step(self, x):
        """controller.py:274-284"""
        if self.xp.on_device and hasattr(self.ocp_solver, 'policy_step'):
            return self.step_on_device(x)
        self.guessCorrection()
        status = self.solve(x)
        self.fails = (self.fails + 1) * self.xp.cast(status != 0, self.xp.i64)      # 0 on success, fails + 1 otherwise
        self.current_step = self.current_step + 1
        return self.provideControl()


class TerminalZeroVelocity(NaiveController):
    cont_name = 'zerovel'


class STController(NaiveController):
    cont_name = 'st'


class STWAController(STController):
    cont_name = 'stwa'
    can_abort = True
    policy_kind = 2

    def setGuess(self, x_guess, u_guess):
        super().setGuess(x_guess, u_guess)
        self.x_viable = self.xp.copy(self.x_guess[:, -1])

    def checkGuess(self):
        return super().checkGuess() & self.checkSafeConstraints(self.x_temp[:, -1])

    def step(self, x):
        """controller.py:375-388"""
        if self.xp.on_device and hasattr(self.ocp_solver, 'policy_step'):
            return self.step_on_device(x)
        xp = self.xp
        self.guessCorrection()
        status = self.solve(x)
        ok = (status == 0) & self.checkStateConstraints(self.x_temp)
        first_fail = ~ok & (self.fails == 0)
        self.x_viable = xp.where(first_fail[:, None], self.x_guess[:, -2], self.x_viable)
        abort = ~ok & (self.fails == self.N - 1)
        u_abort = xp.copy(self.u_guess[:, 0])
        # ok: 0 ; abort: unchanged ; otherwise fails + 1
        self.fails = xp.where(abort, self.fails, self.fails + 1) * xp.cast(~ok, xp.i64)
        active = ~abort
        self.current_step = self.current_step + xp.cast(active, xp.i64)
        u, _ = self.provideControl(active)
        return xp.where(abort[:, None], u_abort, u), abort


class HTWAController(STWAController):
    cont_name = 'htwa'


class RecedingController(STWAController):
    cont_name = 'receding'
    policy_kind = 3

    def reset_controller(self):
        super().reset_controller()
        self.r = self.xp.full((self.B,), self.N, self.xp.i64)
        self.abort_flag = bool(self.params.abort_flag)

    def resetHorizon(self, N):
        super().resetHorizon(N)
        self.r = self.xp.full((self.B,), self.N, self.xp.i64)

    def _set_flags(self):
        """controller.py:452-469: running nodes off except node r; terminal node always on."""
        xp = self.xp
        k = xp.arange(self.N + 1)[None, :]
        r = self.r[:, None]
        on = (k == self.N) | ((k == r) & (r < self.N)) | (k == 0)   # node 0 keeps the default flag (never carries the row anyway)
        self.p[:, :, 4] = xp.cast(on, xp.f64) * 2.0 - 1.0

    def _post_solve(self, x, status, u_abort):
        xp = self.xp
        if self.abort_flag:
            self.r = self.r - 1
            abort = self.r == 0
        else:
            self.r = self.r - xp.cast(self.r > 0, xp.i64)
            abort = xp.zeros((self.B,), xp.bool_)
        self.x_viable = xp.where(abort[:, None], self.x_guess[:, 1], self.x_viable)
        self.r = self.r + xp.cast(abort, xp.i64) * self.N            # r = 0 -> N for the aborting instances
        ok = (status == 0) & self.checkStateConstraints(self.x_temp) & ~abort
        # r <- largest i-1, i in r+2..N, whose node passes the safe-set test (controller.py:491-494)
        safe = self.checkSafeConstraints(self.x_temp)                      # [B, N+1]
        i = xp.arange(self.N + 1)[None, :]
        best = xp.last_true(safe & (i >= (self.r[:, None] + 2)))           # largest such i, -1 if none
        self.r = xp.where(ok & (best >= 0), best - 1, self.r)
        # abort: unchanged ; ok: 0 ; otherwise fails + 1
        self.fails = xp.where(abort, self.fails, (self.fails + 1) * xp.cast(~ok, xp.i64))
        active = ~abort
        self.current_step = self.current_step + xp.cast(active, xp.i64)
        u, _ = self.provideControl(active)
        return xp.where(abort[:, None], u_abort, u), abort

    def step(self, x):
        """controller.py:448-498"""
        if self.xp.on_device and hasattr(self.ocp_solver, 'policy_step'):
            return self.step_on_device(x)
        self.guessCorrection()
        self._set_flags()
        u_abort = self.xp.copy(self.u_guess[:, 0])
        status = self.solve(x)
        return self._post_solve(x, status, u_abort)


class RealReceding(RecedingController):
    """controller.py:504-565: hard terminal safe set; instead of a running safe-set row, node r of each instance is boxed to
    the previously planned state x_guess[r+1] +- 1e-3 (per-instance stage bounds in the engine).  No guessCorrection."""
    cont_name = 'real_receding'
    policy_kind = 4
    TUBE = 1e-3

    def _alloc(self):
        super()._alloc()
        if self.xp.on_device:      # the bounds away from node r (:534-536): model bounds, the terminal node its own
            xp = self.xp
            self._stage_lo = xp.asarray(np.vstack([np.tile(self.problem.x_min, (self.N, 1)), self.problem.lbx_e[None, :]]), xp.f64)
            self._stage_hi = xp.asarray(np.vstack([np.tile(self.problem.x_max, (self.N, 1)), self.problem.ubx_e[None, :]]), xp.f64)

    def step(self, x):
        if self.xp.on_device and hasattr(self.ocp_solver, 'policy_step'):
            return self.step_on_device(x)
        xp, pr, N = self.xp, self.problem, self.N
        # other nodes: the model bounds (:534-536); the terminal node keeps lbx_e / ubx_e
        k = xp.arange(N + 1)[None, :, None]
        last = k == N
        lo = xp.where(last, self._lbx_e[None, None, :], self._x_min[None, None, :])
        hi = xp.where(last, self._ubx_e[None, None, :], self._x_max[None, None, :])
        sel = self.r < N
        centre = xp.take_rows(self.x_guess, xp.clip_max(self.r, N - 1) + 1)[:, None, :]     # x_guess[b, r_b + 1]
        at_r = (k == self.r[:, None, None]) & sel[:, None, None]
        lo = xp.where(at_r, centre - self.TUBE, lo + 0.0 * centre)      # (+ 0 * centre: broadcast to [B, N+1, nx])
        hi = xp.where(at_r, centre + self.TUBE, hi + 0.0 * centre)
        if xp.on_device:
            lo, hi = lo.contiguous(), hi.contiguous()
        self.ocp_solver.set_instance_bounds(lo, hi)
        u_abort = xp.copy(self.u_guess[:, 0])
        status = self.solve(x)
        return self._post_solve(x, status, u_abort)


class ControllerSafeSetEverywhere(STController):
    cont_name = 'constraint_everywhere'
    policy_kind = 1

    def step(self, x):
        """controller.py:651-661"""
        if self.xp.on_device and hasattr(self.ocp_solver, 'policy_step'):
            return self.step_on_device(x)
        xp = self.xp
        self.guessCorrection()
        status = self.solve(x)
        ok = (status == 0) & self.checkStateConstraints(self.x_temp)
        self.fails = (self.fails + 1) * xp.cast(~ok, xp.i64)
        self.current_step = self.current_step + 1
        return self.provideControl()


class SafeBackupController(AbstractController):
    """controller.py:692-712: zero cost, terminal zero velocity, horizon back_hor; only ``solve`` is used (mpc.py:177)."""
    cont_name = 'backup'

    def __init__(self, params, batch, N=None, **kw):
        super().__init__(params, batch, cost='zero', N=N if N is not None else params.back_hor, **kw)


CONTROLLERS = {'naive': NaiveController, 'zerovel': TerminalZeroVelocity, 'st': STController, 'stwa': STWAController,
               'htwa': HTWAController, 'receding': RecedingController, 'real_receding': RealReceding,
               'constraint_everywhere': ControllerSafeSetEverywhere}


def get_controller(cont_name, params, batch, **kw):
    """utils.py:64-75"""
    if cont_name not in CONTROLLERS:
        raise ValueError(f'Controller {cont_name} not available')
    return CONTROLLERS[cont_name](params, batch, **kw)
