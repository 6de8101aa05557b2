"""Test plumbing: the ``numerics`` object oracle/policy_oracle.py asks for, on top of anything with the BatchedOcpSolver
interface -- the CPU oracle double (tests/fake_solver.py) or the HIP engine's host path -- one instance per call.  Statuses,
safe-set verdicts and backup outcomes can be scripted, so that the automata are driven through branches a real solver
rarely takes."""
import numpy as np


class SolverNumerics:
    def __init__(self, solver, problem, params, backup_solver=None, backup_problem=None, joints_noisy=None, tau_noise=None):
        self.s, self.pr, self.par = solver, problem, params
        self.bs, self.bpr = backup_solver, backup_problem
        self.nq, self.N = problem.nq, problem.N
        self.joints_noisy, self.tau_noise = joints_noisy, tau_noise
        self.status_script = None          # callable(step) -> status or None (None: the solver's own)
        self.safe_script = None            # callable(x_node) -> bool
        self.backup_script = None          # callable(x_viable) -> status or None
        self.step_no = 0

    def on_step(self, j):
        self.step_no = j

    # ---- controller side --------------------------------------------------------------------------------------------------
    def integrate_naively(self, x, u):
        """env_model.py:63-67, 210-212: the double integrator the controller believes in"""
        nq, dt = self.nq, self.par.dt
        x, u = np.asarray(x, float), np.asarray(u, float)
        return np.concatenate([x[:nq] + dt * x[nq:] + 0.5 * dt * dt * u, x[nq:] + dt * u])

    def solve(self, x0, x_guess, u_guess, flags, lo, hi, ref=None):
        N = self.N
        p = np.zeros((1, N + 1, 5))
        p[0, :, :3] = self.pr.ee_ref if ref is None else np.asarray(ref, float)
        p[0, :, 3] = self.par.alpha
        p[0, :, 4] = np.asarray(flags, float)
        if lo is not None:
            self.s.set_instance_bounds(np.asarray(lo, float)[None], np.asarray(hi, float)[None])
        x, u, st, _ = self.s.solve(np.asarray(x0, float)[None], np.asarray(x_guess, float)[None], np.asarray(u_guess, float)[None], p)
        status = int(np.asarray(st)[0])
        if self.status_script is not None:
            forced = self.status_script(self.step_no)
            if forced is not None:
                status = int(forced)
        return status, np.asarray(x)[0], np.asarray(u)[0]

    def model_state_bounds(self):
        return self.pr.x_min, self.pr.x_max

    def default_state_bounds(self, N):
        lo = [np.array(self.pr.x_min, float) for _ in range(N)] + [np.array(self.pr.lbx_e, float)]
        hi = [np.array(self.pr.x_max, float) for _ in range(N)] + [np.array(self.pr.ubx_e, float)]
        return lo, hi

    def check_state_bounds(self, x):
        x = np.asarray(x, float)
        tol = self.par.tol_x
        return bool(np.all(x >= self.pr.x_min - tol) and np.all(x <= self.pr.x_max + tol))

    def check_collision(self, x):
        """env_model.py:238-243: returns after the first row of what it is given"""
        x = np.atleast_2d(np.asarray(x, float))
        return bool(np.asarray(self.s.check_trajectory(np.ascontiguousarray(x[None, :1, :]), tol_x=1e30))[0])

    def check_state_constraints(self, x_traj):
        return self.check_state_bounds(np.asarray(x_traj, float)) and self.check_collision(x_traj)

    def check_safe(self, x_node):
        if self.safe_script is not None:
            return bool(self.safe_script(np.asarray(x_node, float)))
        _, nn = self.s.check_trajectory(np.asarray(x_node, float)[None, None, :], want_nn=True)
        return bool(np.asarray(nn)[0, 0])

    # ---- driver side ------------------------------------------------------------------------------------------------------
    def plant(self, x, u):
        jn = None if self.joints_noisy is None else self.joints_noisy[None]
        tn = None if self.tau_noise is None else self.tau_noise[None]
        xn, _ = self.s.plant_step(np.asarray(x, float)[None], np.asarray(u, float)[None], jn, tn)
        return np.asarray(xn)[0]

    def backup_solve(self, xv):
        Nb = self.bpr.N
        p = np.zeros((1, Nb + 1, 5))
        p[0, :, :3], p[0, :, 3], p[0, :, 4] = self.bpr.ee_ref, self.par.alpha, 1.0
        xv = np.asarray(xv, float)
        x, u, st, _ = self.bs.solve(xv[None], np.repeat(xv[None, None, :], Nb + 1, axis=1), np.zeros((1, Nb, self.pr.nu)), p)
        status = int(np.asarray(st)[0])
        if self.backup_script is not None:
            forced = self.backup_script(xv)
            if forced is not None:
                status = int(forced)
        return status, np.asarray(x)[0], np.asarray(u)[0]

    def converged(self, x_last):
        N = self.N
        p = np.zeros((1, N + 1, 5))
        p[0, :, :3], p[0, :, 3], p[0, :, 4] = self.pr.ee_ref, self.par.alpha, 1.0
        ev = self.s.eval_nodes(np.repeat(np.asarray(x_last, float)[None, None, :], N + 1, axis=1), np.zeros((1, N, self.pr.nu)), p)
        ee = np.asarray(ev['ee'])[0, 0, :]
        return bool(np.linalg.norm(ee - self.pr.ee_ref) < self.par.tol_conv)
