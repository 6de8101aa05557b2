"""``HipOcpSolver``: the reference's ONLY seam for this path, method for method.

In idra-lab/safe-mpc everything below ``AbstractController.solve`` is reached through one object, ``self.ocp_solver`` --
an ``acados_template.AcadosOcpSolver`` created at controller.py:247.  This class has the same method set the reference
calls on it (complete list) and forwards to the HIP engine through the C ABI (include/smpc.h); a
scalar-instance view = a batch of one.  A reference ``controller.py`` that replaces line 247 by

    self.ocp_solver = HipOcpSolver.from_params(self.model.params, cont_name, cost)

keeps every other line.  The batched engine (:class:`BatchedOcpSolver`) is the fast path; this facade exists so that the
reference-shaped policies and scripts stay drop-in.

    reference call (controller.py)                          here
    ------------------------------------------------------  ----------------------------------------------------------
    reset()                                          :141   zero the primal iterate
    constraints_set(0, 'lbx'|'ubx', x0)           :144-145  pins x_0 (both must be set to the same vector)
    constraints_set(k, 'lbx'|'ubx', v), k >= 1    :531-536  stage bounds of node k (RealReceding's tube) -> smpc_set_stage_bounds
    set(k, 'x'|'u'|'p', v)                        :148-156  warm start / per-node parameters [ee_ref, alpha, flag]
    solve() -> status                                :158   smpc_solve_batch with B = 1
    get(k, 'x'|'u'|'p')                           :162-164  iterate after the solve
    cost_set(k, 'zl'|'zu', v)                 :455-468,526  L1 weight of the safe-set slack of node k -> smpc_set_slack_weights
    get_stats('time_*'|'sqp_iter'|'qp_iter')         :193   HIP-event times of the last solve (seconds), iteration counts
    get_status(), print_statistics()              mpc.py:178
    set_new_time_steps(arr), update_qp_solver_cond_N(N) :208-209   smpc_set_horizon (uniform steps of dt only)

Errors: like acados, numerical outcomes are integer statuses (0 ok, 1 NaN, 2 max-iter, 3 min-step, 4 QP failure); API
misuse raises (acados raises too: wrong field name, wrong size).
"""
from __future__ import annotations

import numpy as np

from .problem import OcpProblem


class HipOcpSolver:
    TIME_FIELDS = ('time_lin', 'time_sim', 'time_qp', 'time_qp_solver_call', 'time_glob', 'time_reg', 'time_tot')

    def __init__(self, problem: OcpProblem, net=None, device=0, batched=None):
        if batched is None:
            from .solver import BatchedOcpSolver
            batched = BatchedOcpSolver(problem, net, device=device)
        self.batched = batched
        self.problem = problem
        self.nx, self.nu, self.N = problem.nx, problem.nu, problem.N
        self.dt = float(problem.params.dt)
        self._status = 4                                   # controller.py:125 initial last_status
        self._qp_iter = 0
        self._alloc()
        try:
            self.batched.enable_timing(True)
        except Exception:
            pass

    @classmethod
    def from_params(cls, params, controller='naive', cost='ext', N=None, net=None, device=0):
        """What AcadosOcpSolver(ocp, json_file, generate, build) does at controller.py:247: formulation -> loaded solver."""
        from .safe_set import SafeSetNet
        prob = OcpProblem(params, controller, cost, N=N)
        if net is None and prob.desc.nn_mode != 0:
            net = SafeSetNet.from_params(params, prob.x_min, prob.x_max)
        if net is not None:
            prob.set_normalisation(net.mean, net.std)
        return cls(prob, net, device=device)

    # -- state -----------------------------------------------------------------------------------------------------------
    def _alloc(self):
        N, nx, nu, pr = self.N, self.nx, self.nu, self.problem
        self._x = np.zeros((1, N + 1, nx))
        self._u = np.zeros((1, N, nu))
        self._p = np.zeros((1, N + 1, 5))
        self._p[0, :, :3], self._p[0, :, 3], self._p[0, :, 4] = pr.ee_ref, pr.params.alpha, 1.0     # controller.py:37
        self._lbx = np.tile(pr.lbx, (N + 1, 1))
        self._ubx = np.tile(pr.ubx, (N + 1, 1))
        self._lbx[N], self._ubx[N] = pr.lbx_e, pr.ubx_e
        self._lbx[0], self._ubx[0] = np.nan, np.nan          # x_0 has to be given before a solve
        self._bounds_dirty = False                           # stage bounds differ from what the engine holds
        self._bounds_default = True
        self._zl = None                                      # per-node slack weights, None = the formulation's
        self._zl_dirty = False

    def _node(self, k, last_ok=True):
        k = int(k)
        if not (0 <= k <= (self.N if last_ok else self.N - 1)):
            raise ValueError(f'stage {k} outside the horizon 0..{self.N if last_ok else self.N - 1}')
        return k

    @staticmethod
    def _vec(v, n, what):
        a = np.asarray(v, float).reshape(-1)
        if a.size != n:
            raise ValueError(f'{what}: expected {n} values, got {a.size}')
        return a

    # -- AcadosOcpSolver method set ----------------------------------------------------------------------------------------
    def reset(self):
        """controller.py:141: acados zeroes the primal (and dual) iterate; the RTI step starts its duals afresh anyway."""
        self._x[:] = 0.0
        self._u[:] = 0.0

    def constraints_set(self, stage, field, value):
        k = self._node(stage)
        if field not in ('lbx', 'ubx'):
            raise ValueError(f"constraints_set: field '{field}' is not used by the reference on this path (lbx, ubx)")
        v = self._vec(value, self.nx, f'constraints_set({k}, {field})')
        tgt = self._lbx if field == 'lbx' else self._ubx
        if k >= 1 and not np.array_equal(tgt[k], v):
            self._bounds_dirty = True
            self._bounds_default = False
        tgt[k] = v

    def set(self, stage, field, value):
        if field == 'x':
            self._x[0, self._node(stage)] = self._vec(value, self.nx, 'set x')
        elif field == 'u':
            self._u[0, self._node(stage, last_ok=False)] = self._vec(value, self.nu, 'set u')
        elif field == 'p':
            self._p[0, self._node(stage)] = self._vec(value, 5, 'set p')
        else:
            raise ValueError(f"set: field '{field}' is not used by the reference on this path (x, u, p)")

    def get(self, stage, field):
        if field == 'x':
            return self._x[0, self._node(stage)].copy()
        if field == 'u':
            return self._u[0, self._node(stage, last_ok=False)].copy()
        if field == 'p':
            return self._p[0, self._node(stage)].copy()
        raise ValueError(f"get: field '{field}' is not used by the reference on this path (x, u, p)")

    def cost_set(self, stage, field, value):
        """controller.py:455-468, 526-527.  Only the safe-set row can carry a slack (idxsh, controller.py:348-354, 439-442), so
        the arrays have zero or one entry: zero entries (a hard row) is the reference's no-op."""
        k = self._node(stage)
        if field not in ('zl', 'zu', 'Zl', 'Zu'):
            raise ValueError(f"cost_set: field '{field}' is not used by the reference on this path (zl, zu)")
        a = np.asarray(value, float).reshape(-1)
        if a.size == 0:
            return
        if a.size != 1:
            raise ValueError('cost_set: this OCP has at most one soft row per node (the safe-set row)')
        if field in ('Zl', 'Zu'):
            if a[0] != 0.0:
                raise NotImplementedError('quadratic slack penalties are zero in the reference (controller.py:351-354)')
            return
        if field == 'zu':
            return          # the row's upper side is 1e6 = absent (safe_set.py:104): its slack never moves
        if self._zl is None:
            d = self.problem.desc
            self._zl = np.full(self.N + 1, max(d.nn_soft_run, 0.0))
            self._zl[self.N] = max(d.nn_soft_e, 0.0)
        if self._zl[k] != a[0]:
            self._zl[k] = a[0]
            self._zl_dirty = True

    def solve(self):
        """controller.py:158: one SQP-RTI iteration on the loaded iterate; returns the acados status code."""
        if not np.array_equal(self._lbx[0], self._ubx[0]):
            raise ValueError('lbx and ubx of stage 0 must both be set to the measured state before solve() '
                             '(controller.py:144-145)')
        if self._bounds_dirty:
            lo, hi = self._lbx.copy(), self._ubx.copy()
            lo[0], hi[0] = self.problem.lbx, self.problem.ubx           # node 0 is pinned to x0, its box never enters
            self.batched.set_stage_bounds(lo, hi)
            self._bounds_dirty = False
        if self._zl_dirty:
            self.batched.set_slack_weights(self._zl)
            self._zl_dirty = False
        x, u, st, it = self.batched.solve(self._lbx[0][None, :], self._x, self._u, self._p)
        self._x, self._u = np.array(x, float), np.array(u, float)
        self._status, self._qp_iter = int(st[0]), int(it[0])
        return self._status

    def get_status(self):
        return self._status

    def get_stats(self, field):
        """controller.py:193 (time_* in seconds, as acados reports them), plus the iteration counts of test_z1.py:64."""
        if field == 'sqp_iter':
            return 1                                            # SQP_RTI: one iteration per solve (controller.py:102)
        if field == 'qp_iter':
            return np.array([self._qp_iter])
        if field in self.TIME_FIELDS:
            try:
                t = self.batched.timing()
            except Exception:
                return 0.0
            g = lambda k: float(t.get(k, 0.0))
            return {'time_lin': g('time_lin') + g('time_nn'), 'time_qp': g('time_qp'), 'time_qp_solver_call': g('time_qp_ipm'),
                    'time_tot': g('time_tot')}.get(field, 0.0)   # time_sim / time_glob / time_reg: nothing of the kind runs
        raise ValueError(f"get_stats: unknown field '{field}'")

    def print_statistics(self):
        print(f'iter\tqp_stat\tqp_iter\n1\t{self._status}\t{self._qp_iter}')

    def set_new_time_steps(self, new_time_steps):
        """controller.py:208: the reference passes np.full(N, dt); a non-uniform grid is outside this path."""
        ts = np.asarray(new_time_steps, float).reshape(-1)
        if ts.size < 1 or not np.allclose(ts, self.dt, rtol=1e-12, atol=0.0):
            raise NotImplementedError('only uniform steps of params.dt (what controller.py:208 passes)')
        if ts.size != self.N:
            self.N = int(ts.size)
            self.problem.N = self.N
            self.batched.set_horizon(self.N)                     # resets stage bounds and slack weights in the engine
            self._alloc()

    def update_qp_solver_cond_N(self, qp_solver_cond_N):
        """controller.py:209: partial-condensing block count of HPIPM -- the Riccati solver here works on the N stages."""
        if int(qp_solver_cond_N) != self.N:
            raise ValueError(f'qp_solver_cond_N = {qp_solver_cond_N} but the horizon is {self.N}')
