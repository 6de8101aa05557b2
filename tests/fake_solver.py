"""Test double with the BatchedOcpSolver interface, backed by the CPU oracle -- lets the policy layer and the entry
scripts run on a machine without a GPU.  Lives under tests/: the product never imports it."""
import numpy as np

from oracle.oracle import Oracle


class OracleSolver:
    def __init__(self, problem, net=None, device=0):
        self.problem = problem
        self.o = Oracle(problem, (net.weights, net.biases, getattr(net, "act", "gelu")) if net is not None else None)
        self.N, self.nx, self.nu, self.nq = problem.N, problem.nx, problem.nu, problem.nq
        self.scripted_status = None          # optional list of status arrays consumed by successive solves

    def set_horizon(self, N):
        self.o.set_horizon(N)
        self.N = N

    def set_stage_bounds(self, lo=None, hi=None):
        self.o.set_stage_bounds(lo, hi)

    def set_instance_bounds(self, lo=None, hi=None):
        self.o.set_instance_bounds(lo, hi)

    def set_slack_weights(self, zl=None):
        self.o.set_slack_weights(zl)

    def enable_timing(self, on=True):
        pass

    def solve(self, x0, xg, ug, p, out=None):
        import time
        t0 = time.perf_counter()
        x, u, st, it = self.o.solve_batch(x0, xg, ug, p)
        dt = time.perf_counter() - t0
        self._last_times = {'time_lin': 0.0, 'time_nn': 0.0, 'time_qp_setup': 0.0, 'time_qp_ipm': dt, 'time_tot': dt}
        if self.scripted_status:
            st = np.asarray(self.scripted_status.pop(0), np.int32)
        return x, u, st, it

    def eval_nodes(self, xg, ug, p):
        return self.o.eval_nodes(xg, ug, p)

    def guess_correction(self, xg, ug):
        return self.o.guess_correction(xg, ug)

    def provide_control(self, accept, xt, ut, xg, ug):
        return self.o.provide_control(accept, xt, ut, xg, ug)

    def check_trajectory(self, x, x_min=None, x_max=None, tol_x=None, row_lb=None, row_ub=None, alpha=None, tol_safe=None,
                         want_nn=False):
        pr, par = self.problem, self.problem.params
        return self.o.check_trajectory(x, pr.x_min if x_min is None else x_min, pr.x_max if x_max is None else x_max,
                                       par.tol_x if tol_x is None else tol_x,
                                       pr.row_check[:, 0] if row_lb is None else row_lb,
                                       pr.row_check[:, 1] if row_ub is None else row_ub,
                                       par.alpha if alpha is None else alpha,
                                       par.tol_safe_set if tol_safe is None else tol_safe, want_nn=want_nn)

    def plant_step(self, x, u, joints_noisy=None, tau_noise=None):
        return self.o.plant_step(x, u, joints_noisy, tau_noise)

    def timing(self):
        return {}

    def timing_history(self, back=0):
        return getattr(self, '_last_times', None) if back == 0 else None

    def sync(self):
        pass


def make_double_controller(name, params, batch, N=None, cost='ext'):
    """A policy-layer controller (safe_mpc_amd.controller) built around the CPU oracle instead of the HIP engine: what the
    engine-backed controller is compared with in the tests (same class, same automaton, other numerics)."""
    from safe_mpc_amd import controller as C
    from safe_mpc_amd.safe_set import SafeSetNet
    cls = C.SafeBackupController if name == 'backup' else C.CONTROLLERS[name]
    N = int(N if N is not None else (params.back_hor if name == 'backup' else params.N))
    prob = C.OcpProblem(params, cls.cont_name, 'zero' if name == 'backup' else cost, N=N)
    net = SafeSetNet.from_params(params, prob.x_min, prob.x_max) if params.use_net else None
    if net is not None:
        prob.set_normalisation(net.mean, net.std)
    ctrl = cls.__new__(cls)
    C.AbstractController.__init__(ctrl, params, batch, 'zero' if name == 'backup' else cost, N, solver=OracleSolver(prob, net), net=net)
    return ctrl
