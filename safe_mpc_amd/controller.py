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

from .problem import OcpProblem
from .safe_set import SafeSetNet


class AbstractController:
    cont_name = 'naive'

    def __init__(self, params, batch, cost='ext', N=None, solver=None, net=None, device=0):
        self.params = params
        self.B = int(batch)
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
        # per-node parameters p = [ee_ref(3), alpha, flag] (controller.py:27-37, 153-156)
        self.p = np.zeros((self.B, self.N + 1, 5))
        self._alloc()
        self.reset_controller()

    # -- state --------------------------------------------------------------------------------------------------------
    def _alloc(self):
        B, N = self.B, self.N
        self.x_guess = np.zeros((B, N + 1, self.nx))
        self.u_guess = np.zeros((B, N, self.nu))
        self.x_temp, self.u_temp = self.x_guess.copy(), self.u_guess.copy()
        self.p = np.zeros((B, N + 1, 5))
        self.p[:, :, :3] = self.problem.ee_ref
        self.p[:, :, 3] = self.params.alpha
        self.p[:, :, 4] = 1.0
        self.last_status = np.full(B, 4, np.int32)                 # controller.py:125
        self.qp_iter = np.zeros(B, np.int32)
        self.x_viable = np.zeros((B, self.nx))

    def reset_controller(self):
        self.fails = np.zeros(self.B, np.int64)
        self.current_step = np.zeros(self.B, np.int64)

    def setReference(self, ee_ref):
        self.problem.ee_ref = np.asarray(ee_ref, float)
        self.p[:, :, :3] = self.problem.ee_ref

    def setGuess(self, x_guess, u_guess):
        self.x_guess = np.array(x_guess, float, copy=True)
        self.u_guess = np.array(u_guess, float, copy=True)

    def getGuess(self):
        return np.copy(self.x_guess), np.copy(self.u_guess)

    def getLastViableState(self):
        return np.copy(self.x_viable)

    def getTime(self):
        try:
            t = self.ocp_solver.timing()
        except Exception:
            t = {}
        return np.array([t.get(f, 0.0) for f in self.time_fields])

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
        x, u, st, it = self.ocp_solver.solve(np.asarray(x0, float), self.x_guess, self.u_guess, self.p)
        self.x_temp, self.u_temp = np.asarray(x), np.asarray(u)
        self.last_status = np.asarray(st).copy()
        self.qp_iter = np.asarray(it).copy()
        return self.last_status

    def guessCorrection(self):
        self.x_guess = self.ocp_solver.guess_correction(self.x_guess, self.u_guess)

    def provideControl(self, active=None):
        """controller.py:169-184.  ``active`` masks out instances that returned early (abort)."""
        accept = (self.fails == 0).astype(np.int32)
        xg, ug, u = self.ocp_solver.provide_control(accept, self.x_temp, self.u_temp, self.x_guess, self.u_guess)
        if active is None:
            self.x_guess, self.u_guess = xg, ug
        else:
            u = np.where(active[:, None], u, self.u_guess[:, 0])
            self.x_guess = np.where(active[:, None, None], xg, self.x_guess)
            self.u_guess = np.where(active[:, None, None], ug, self.u_guess)
        return u, np.zeros(self.B, bool)

    # -- feasibility predicates (env_model.py:170-243, safe_set.py:61-68) ------------------------------------------------
    def checkStateConstraints(self, x):
        """env_model.py:170-173.  Reference quirk kept by default (``params.reference_quirks``):
        ``checkCollision`` returns inside its loop after the FIRST row (env_model.py:238-243), so a trajectory is
        collision-checked at its first node only, while the box test covers every node."""
        x = np.asarray(x, float)
        if x.ndim == 2:
            x = x[:, None, :]
        tol = self.params.tol_x
        in_box = np.all((x >= self.problem.x_min - tol) & (x <= self.problem.x_max + tol), axis=(1, 2))
        xc = x[:, :1] if getattr(self.params, 'reference_quirks', True) else x
        free = np.asarray(self.ocp_solver.check_trajectory(np.ascontiguousarray(xc), tol_x=1e30))
        return in_box & free

    def checkSafeConstraints(self, x):
        """x[B, n_nodes, nx] (or [B, nx]) -> bool per (instance, node)."""
        x = np.asarray(x, float)
        squeeze = x.ndim == 2
        if squeeze:
            x = x[:, None, :]
        _, nn = self.ocp_solver.check_trajectory(x, want_nn=True)
        nn = np.asarray(nn)
        return nn[:, 0] if squeeze else nn

    def checkTorqueConstraints(self, x, u):
        ev = self.ocp_solver.eval_nodes(x, u, self.p)
        tau = ev['tau'][:, :self.N, :self.nq]
        tol = self.params.tol_tau
        return np.all((tau >= self.problem.tau_min - tol) & (tau <= self.problem.tau_max + tol), axis=(1, 2))

    def checkDynamicsConstraints(self, x, u):
        """env_model.py:226-234 with the controller's own model (no torque saturation inside the rollout)."""
        sim = self.ocp_solver.guess_correction(x, u)
        n = u.shape[1]
        return np.linalg.norm((x - sim).reshape(self.B, -1), axis=1) < self.params.tol_dyn * np.sqrt(n + 1)

    def checkGuess(self):
        return (self.checkStateConstraints(self.x_temp) & self.checkTorqueConstraints(self.x_temp, self.u_temp) &
                self.checkDynamicsConstraints(self.x_temp, self.u_temp))

    def initialize(self, x0, u0=None):
        """controller.py:260-272: trivial guess, one solve, keep it where it checks out.  Returns 1/0 per instance."""
        x0 = np.asarray(x0, float)
        self.x_guess = np.repeat(x0[:, None, :], self.N + 1, axis=1)
        self.u_guess = np.zeros((self.B, self.N, self.nu)) if u0 is None else np.repeat(np.asarray(u0)[:, None, :], self.N, 1)
        status = self.solve(x0)
        good = (status == 0) & self.checkGuess()
        self.x_guess = np.where(good[:, None, None], self.x_temp, self.x_guess)
        self.u_guess = np.where(good[:, None, None], self.u_temp, self.u_guess)
        return good.astype(int)

    def step(self, x):
        raise NotImplementedError


class NaiveController(AbstractController):
    cont_name = 'naive'

    def step(self, x):
        """controller.py:274-284"""
        self.guessCorrection()
        status = self.solve(x)
        self.fails = np.where(status == 0, 0, self.fails + 1)
        self.current_step += 1
        return self.provideControl()


class TerminalZeroVelocity(NaiveController):
    cont_name = 'zerovel'


class STController(NaiveController):
    cont_name = 'st'


class STWAController(STController):
    cont_name = 'stwa'

    def setGuess(self, x_guess, u_guess):
        super().setGuess(x_guess, u_guess)
        self.x_viable = self.x_guess[:, -1].copy()

    def checkGuess(self):
        return super().checkGuess() & self.checkSafeConstraints(self.x_temp[:, -1])

    def step(self, x):
        """controller.py:375-388"""
        self.guessCorrection()
        status = self.solve(x)
        ok = (status == 0) & self.checkStateConstraints(self.x_temp)
        first_fail = ~ok & (self.fails == 0)
        self.x_viable = np.where(first_fail[:, None], self.x_guess[:, -2], self.x_viable)
        abort = ~ok & (self.fails == self.N - 1)
        u_abort = self.u_guess[:, 0].copy()
        self.fails = np.where(ok, 0, np.where(abort, self.fails, self.fails + 1))
        active = ~abort
        self.current_step += active
        u, _ = self.provideControl(active)
        return np.where(abort[:, None], u_abort, u), abort


class HTWAController(STWAController):
    cont_name = 'htwa'


class RecedingController(STWAController):
    cont_name = 'receding'

    def reset_controller(self):
        super().reset_controller()
        self.r = np.full(self.B, self.N, np.int64)
        self.abort_flag = bool(self.params.abort_flag)

    def resetHorizon(self, N):
        super().resetHorizon(N)
        self.r = np.full(self.B, self.N, np.int64)

    def _set_flags(self):
        """controller.py:452-469: running nodes off except node r; terminal node always on."""
        k = np.arange(self.N + 1)[None, :]
        on = (k == self.N) | ((k == self.r[:, None]) & (self.r[:, None] < self.N))
        on[:, 0] = True                                   # node 0 keeps the default flag (never carries the row anyway)
        self.p[:, :, 4] = np.where(on, 1.0, -1.0)

    def _post_solve(self, x, status, u_abort):
        if self.abort_flag:
            self.r = self.r - 1
        else:
            self.r = np.where(self.r > 0, self.r - 1, self.r)
        abort = (self.r == 0) & self.abort_flag
        self.x_viable = np.where(abort[:, None], self.x_guess[:, 1], self.x_viable)
        self.r = np.where(abort, self.N, self.r)
        ok = (status == 0) & self.checkStateConstraints(self.x_temp) & ~abort
        # r <- largest i-1, i in r+2..N, whose node passes the safe-set test (controller.py:491-494)
        safe = self.checkSafeConstraints(self.x_temp)                      # [B, N+1]
        i = np.arange(self.N + 1)[None, :]
        cand = safe & (i >= (self.r[:, None] + 2))
        best = np.where(cand.any(1), self.N - np.argmax(cand[:, ::-1], axis=1), -1)   # largest such i
        self.r = np.where(ok & (best >= 0), best - 1, self.r)
        self.fails = np.where(abort, self.fails, np.where(ok, 0, self.fails + 1))
        active = ~abort
        self.current_step += active
        u, _ = self.provideControl(active)
        return np.where(abort[:, None], u_abort, u), abort

    def step(self, x):
        """controller.py:448-498"""
        self.guessCorrection()
        self._set_flags()
        u_abort = self.u_guess[:, 0].copy()
        status = self.solve(x)
        return self._post_solve(x, status, u_abort)


class RealReceding(RecedingController):
    """controller.py:504-565: hard terminal safe set; instead of a running safe-set row, node r of each instance is boxed to
    the previously planned state x_guess[r+1] +- 1e-3 (per-instance stage bounds in the engine).  No guessCorrection."""
    cont_name = 'real_receding'
    TUBE = 1e-3

    def step(self, x):
        pr = self.problem
        lo = np.broadcast_to(pr.x_min, (self.B, self.N + 1, self.nx)).copy()      # other nodes: the model bounds (:534-536)
        hi = np.broadcast_to(pr.x_max, (self.B, self.N + 1, self.nx)).copy()
        lo[:, self.N], hi[:, self.N] = pr.lbx_e, pr.ubx_e                         # terminal node keeps lbx_e / ubx_e
        rows = np.where(self.r < self.N)[0]
        if rows.size:
            r = self.r[rows]
            centre = self.x_guess[rows, r + 1]
            lo[rows, r], hi[rows, r] = centre - self.TUBE, centre + self.TUBE
        self.ocp_solver.set_instance_bounds(lo, hi)
        u_abort = self.u_guess[:, 0].copy()
        status = self.solve(x)
        return self._post_solve(x, status, u_abort)


class ControllerSafeSetEverywhere(STController):
    cont_name = 'constraint_everywhere'

    def step(self, x):
        """controller.py:651-661"""
        self.guessCorrection()
        status = self.solve(x)
        ok = (status == 0) & self.checkStateConstraints(self.x_temp)
        self.fails = np.where(ok, 0, self.fails + 1)
        self.current_step += 1
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
