"""ctypes mirror of include/smpc.h and the builder that turns ``Parameters`` + URDF into a ``smpc_problem_desc``.

This is the host-side restatement of the OCP *formulation* layer of the reference (L2 in SURVEY section 1):
``AdamModel.__init__`` / ``generate_NLconstraints_list`` (env_model.py:19-165, 246-319) and
``AbstractController.__init__`` + the ``additionalSetting`` of each policy (controller.py:13-125, 295-306, 332-357,
411-442, 514-515, 663-689, 692-712).  Instead of CasADi expressions it emits plain data: a joint table, robot-attached
points, collision rows with their bounds, and scalar options.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .urdf import SerialChain

ABI_VERSION = 5
MAX_NQ, MAX_NX, MAX_POINTS, MAX_ROWS, MAX_LAYERS, MAX_N, NP = 7, 14, 12, 12, 6, 63, 5
INF = 1.0e5

ROW_SEG_FIXEDSEG, ROW_SEG_SEG, ROW_SEG_POINT, ROW_POINT_POINT, ROW_COORD = 0, 1, 2, 3, 4
COST_ZERO, COST_REACH = 0, 1
HESS_GAUSS_NEWTON, HESS_EXACT = 0, 1
NN_NONE, NN_TERMINAL, NN_ALL = 0, 1, 2

STATUS_NAMES = {0: 'SUCCESS', 1: 'NAN', 2: 'MAXITER', 3: 'MINSTEP', 4: 'QP_FAILURE'}


class Joint(C.Structure):
    _fields_ = [('R0', C.c_double * 9), ('p0', C.c_double * 3), ('axis', C.c_double * 3), ('mass', C.c_double),
                ('com', C.c_double * 3), ('inertia', C.c_double * 6), ('q_min', C.c_double), ('q_max', C.c_double),
                ('v_max', C.c_double), ('tau_max', C.c_double)]


class Point(C.Structure):
    _fields_ = [('link', C.c_int32), ('reserved', C.c_int32), ('local', C.c_double * 3)]


class Row(C.Structure):
    _fields_ = [('kind', C.c_int32), ('pa', C.c_int32), ('pb', C.c_int32), ('pc', C.c_int32), ('pd', C.c_int32),
                ('axis', C.c_int32), ('C', C.c_double * 3), ('D', C.c_double * 3), ('len2', C.c_double),
                ('offset', C.c_double), ('lb', C.c_double), ('ub', C.c_double)]


class ProblemDesc(C.Structure):
    _fields_ = [('abi_version', C.c_int32), ('nq', C.c_int32), ('N', C.c_int32), ('n_points', C.c_int32),
                ('n_rows', C.c_int32), ('ee_point', C.c_int32), ('cost_kind', C.c_int32), ('hessian', C.c_int32),
                ('nn_mode', C.c_int32), ('nn_dof', C.c_int32), ('qp_max_iter', C.c_int32), ('rows_at_node0', C.c_int32),
                ('qp_stall_iters', C.c_int32), ('reserved_i0', C.c_int32),
                ('dt', C.c_double), ('Q', C.c_double), ('R', C.c_double), ('cost_scale_stage', C.c_double),
                ('cost_scale_term', C.c_double), ('lm_stage', C.c_double), ('lm_term', C.c_double),
                ('nn_eps', C.c_double), ('nn_soft_e', C.c_double), ('nn_soft_run', C.c_double),
                ('qp_tol', C.c_double), ('qp_tol_res', C.c_double), ('qp_mu0', C.c_double), ('gravity', C.c_double * 3),
                ('nn_mean', C.c_double * MAX_NQ), ('nn_std', C.c_double * MAX_NQ),
                ('x_lo', C.c_double * MAX_NX), ('x_hi', C.c_double * MAX_NX),
                ('x_lo_e', C.c_double * MAX_NX), ('x_hi_e', C.c_double * MAX_NX),
                ('joints', Joint * MAX_NQ), ('points', Point * MAX_POINTS), ('rows', Row * MAX_ROWS)]


class NodeEval(C.Structure):
    _fields_ = [('tau', C.c_double * MAX_NQ), ('M', C.c_double * (MAX_NQ * MAX_NQ)),
                ('dtau_dq', C.c_double * (MAX_NQ * MAX_NQ)), ('dtau_dv', C.c_double * (MAX_NQ * MAX_NQ)),
                ('ee', C.c_double * 3), ('cost_grad_q', C.c_double * MAX_NQ),
                ('cost_hess_qq', C.c_double * (MAX_NQ * MAX_NQ)), ('row_val', C.c_double * MAX_ROWS),
                ('row_grad', C.c_double * (MAX_ROWS * MAX_NQ)), ('nn_val', C.c_double),
                ('nn_grad', C.c_double * MAX_NX)]


NODE_EVAL_DTYPE = np.dtype([('tau', 'f8', MAX_NQ), ('M', 'f8', MAX_NQ * MAX_NQ), ('dtau_dq', 'f8', MAX_NQ * MAX_NQ),
                            ('dtau_dv', 'f8', MAX_NQ * MAX_NQ), ('ee', 'f8', 3), ('cost_grad_q', 'f8', MAX_NQ),
                            ('cost_hess_qq', 'f8', MAX_NQ * MAX_NQ), ('row_val', 'f8', MAX_ROWS),
                            ('row_grad', 'f8', MAX_ROWS * MAX_NQ), ('nn_val', 'f8'), ('nn_grad', 'f8', MAX_NX)])
assert NODE_EVAL_DTYPE.itemsize == C.sizeof(NodeEval)

JOINT_DTYPE = np.dtype([('R0', 'f8', 9), ('p0', 'f8', 3), ('axis', 'f8', 3), ('mass', 'f8'), ('com', 'f8', 3),
                        ('inertia', 'f8', 6), ('q_min', 'f8'), ('q_max', 'f8'), ('v_max', 'f8'), ('tau_max', 'f8')])
assert JOINT_DTYPE.itemsize == C.sizeof(Joint)


def _rot(axis, th):
    c, s = np.cos(th), np.sin(th)
    if axis == 0:
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
    if axis == 1:
        return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


# controller name -> how the safe-set row enters
CONTROLLER_KINDS = {
    #                  nn_mode      soft_e     soft_run  term_zero_vel  cost      lm
    'naive':           (NN_NONE,     None,      None,     False),
    'zerovel':         (NN_NONE,     None,      None,     True),
    'st':              (NN_TERMINAL, 'ws_r',    None,     False),
    'stwa':            (NN_TERMINAL, 'ws_r',    None,     False),
    'htwa':            (NN_TERMINAL, None,      None,     False),
    'receding':        (NN_ALL,      'ws_t',    None,     False),
    'real_receding':   (NN_TERMINAL, None,      None,     False),
    'constraint_everywhere': (NN_ALL, None,     None,     False),
    'backup':          (NN_NONE,     None,      None,     True),
}


class OcpProblem:
    """Everything the engine needs about one OCP family; ``desc`` is the C struct handed to ``smpc_create``."""

    def __init__(self, params, controller='naive', cost='ext', N=None, chain=None, rows_at_node0=None):
        if controller not in CONTROLLER_KINDS:
            raise ValueError(f'Controller {controller} not available')
        self.params = params
        self.controller = controller
        nq = params.nq
        if nq > MAX_NQ:
            raise ValueError(f'nq={nq} exceeds SMPC_MAX_NQ={MAX_NQ}')
        self.chain = chain or SerialChain(params.robot_descr, nq)
        self.nq, self.nx, self.nu = nq, 2 * nq, nq
        self.N = int(N if N is not None else (params.back_hor if controller == 'backup' else params.N))
        if not (1 <= self.N <= MAX_N):
            raise ValueError(f'horizon {self.N} outside 1..{MAX_N}')
        nn_mode, soft_e, soft_run, zero_vel = CONTROLLER_KINDS[controller]
        if nn_mode != NN_NONE and not params.use_net:
            raise NotImplementedError('AnalyticSafeSet (use_net: false) is out of scope of this engine')

        d = ProblemDesc()
        d.abi_version, d.nq, d.N = ABI_VERSION, nq, self.N
        d.dt = params.dt
        d.gravity[:] = [0.0, 0.0, -9.80665]

        # joints, limits (env_model.py:107-121)
        jl = np.array([[j.q_min, j.q_max, j.v_max, j.tau_max] for j in self.chain.joints])
        for i, j in enumerate(self.chain.joints):
            J = d.joints[i]
            J.R0[:] = j.R0.reshape(-1)
            J.p0[:] = j.p0
            J.axis[:] = j.axis
            J.mass = j.mass
            J.com[:] = j.com
            I = j.inertia
            J.inertia[:] = [I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]]
            J.q_min, J.q_max, J.v_max, J.tau_max = jl[i]
        self.tau_min, self.tau_max = -jl[:, 3].copy(), jl[:, 3].copy()
        x_min_nom = np.hstack([jl[:, 0], -jl[:, 2]])
        x_max_nom = np.hstack([jl[:, 1], jl[:, 2]])
        self.bounds_diff = np.abs(x_max_nom - x_min_nom)
        mg = params.q_margin / 100.0
        # model bounds are widened by the margin, the OCP shrinks them back (env_model.py:118-121, controller.py:49-55)
        self.x_min = x_min_nom - self.bounds_diff * mg
        self.x_max = x_max_nom + self.bounds_diff * mg
        lo = self.x_min + mg * self.bounds_diff
        hi = self.x_max - mg * self.bounds_diff
        lo_e, hi_e = lo.copy(), hi.copy()
        if controller == 'zerovel':                           # controller.py:300-306
            lo_e[nq:], hi_e[nq:] = 0.0, 0.0
        if controller == 'backup':                            # controller.py:701-707 (model bounds, zero velocity)
            lo_e = np.hstack([self.x_min[:nq], np.zeros(nq)])
            hi_e = np.hstack([self.x_max[:nq], np.zeros(nq)])
        d.x_lo[:2 * nq], d.x_hi[:2 * nq] = lo, hi
        d.x_lo_e[:2 * nq], d.x_hi_e[:2 * nq] = lo_e, hi_e
        self.lbx, self.ubx, self.lbx_e, self.ubx_e = lo, hi, lo_e, hi_e

        # robot-attached points
        self._points = []
        fr_idx, fr_R, fr_p = self.chain.frame(params.frame_name)
        d.ee_point = self._add_point(fr_idx, fr_p + fr_R @ params.ee_pos)      # env_model.py:92-95
        self.ee_ref = np.array(params.ee_ref, float)

        # collision rows (env_model.py:263-316)
        self.rows = []
        self.row_names = []
        self.row_check = []            # (lo, hi) used by checkCollision (env_model.py:236-243)
        m2 = 2.0 * params.collision_margin
        tol = params.tol_obs
        for i, pair in enumerate(params.collisions_pairs):
            e0, e1 = pair['elements']
            r = Row()
            if pair['type'] == 'capsule-capsule':
                a, b = self._capsule_points(e0)
                r.pa, r.pb = a, b
                if e1['type'] == 'fixed_capsule':
                    r.kind = ROW_SEG_FIXEDSEG
                    r.C[:], r.D[:] = e1['end_points'][0], e1['end_points'][1]
                else:
                    r.kind = ROW_SEG_SEG
                    r.pc, r.pd = self._capsule_points(e1)
                r.lb, r.ub = (e0['radius'] + e1['radius'] + m2) ** 2, 1e6
                self._push_row(r, f"{e0['name']}-{e1['name']}", (e0['radius'] + e1['radius']) ** 2 - tol, 1e6 + tol)
            elif pair['type'] == 'capsule-sphere':
                r.kind = ROW_SEG_POINT
                r.pa, r.pb = self._capsule_points(e0)
                r.C[:] = e1['position']
                r.len2 = e0['length'] ** 2
                r.lb, r.ub = (e1['radius'] + e0['radius'] + m2) ** 2, 1e6
                self._push_row(r, f"{e0['name']}-{e1['name']}", (e1['radius'] + e0['radius']) ** 2 - tol, 1e6 + tol)
            elif pair['type'] == 'capsule-plane':
                for pt in self._capsule_points(e0):
                    r = Row()
                    r.kind, r.pa, r.axis, r.offset = ROW_COORD, pt, int(e1['perpendicular_axis']), 0.0
                    r.lb = e1['bounds'][0] + e0['radius'] + m2
                    r.ub = e1['bounds'][1] - e0['radius'] - m2
                    self._push_row(r, f"{e0['name']}-{e1['name']}", e1['bounds'][0] + e0['radius'] - tol,
                                   e1['bounds'][1] - e0['radius'] + tol)
            elif pair['type'] == 'sphere-sphere':
                r.kind, r.pa = ROW_POINT_POINT, d.ee_point      # the reference uses t_glob here (env_model.py:300)
                r.C[:] = e1['position']
                r.lb, r.ub = (e0['radius'] + e1['radius'] + m2) ** 2, 1e6
                self._push_row(r, f"{e0['name']}-{e1['name']}", (e0['radius'] + e1['radius']) ** 2 - tol, 1e6 + tol)
            elif pair['type'] == 'sphere-plane':
                li, lR, lp = self.chain.frame(e0['link_name'])
                r.kind = ROW_COORD
                r.pa = self._add_point(li, lp + lR @ np.array(e0['spatial_offset'], float))
                r.axis = int(e1['perpendicular_axis'])
                r.offset = float(e1['bounds'][int(e1['real_bound'])])           # utils.py:123-124
                r.lb = e1['bounds'][0] + e0['radius'] + m2
                r.ub = e1['bounds'][1] - e0['radius'] - m2
                self._push_row(r, f"{e0['name']}-{e1['name']}", e1['bounds'][0] + e0['radius'] - tol,
                               e1['bounds'][1] - e0['radius'] + tol)
            else:
                raise ValueError(f"unsupported collision pair type {pair['type']}")
        if len(self.rows) > MAX_ROWS:
            raise ValueError(f'{len(self.rows)} collision rows exceed SMPC_MAX_ROWS={MAX_ROWS}')
        if len(self._points) > MAX_POINTS:
            raise ValueError(f'{len(self._points)} robot points exceed SMPC_MAX_POINTS={MAX_POINTS}')
        d.n_rows, d.n_points = len(self.rows), len(self._points)
        for i, r in enumerate(self.rows):
            d.rows[i] = r
        for i, (li, loc) in enumerate(self._points):
            d.points[i].link = li
            d.points[i].local[:] = loc
        self.row_lb = np.array([r.lb for r in self.rows])
        self.row_ub = np.array([r.ub for r in self.rows])
        self.row_check = np.array(self.row_check).reshape(-1, 2)

        # cost (cost_definition.py:34-100)
        cost = 'zero' if controller == 'backup' else cost
        self.cost = cost
        d.cost_kind = COST_ZERO if cost == 'zero' else COST_REACH
        d.hessian = HESS_EXACT if cost == 'ext' else HESS_GAUSS_NEWTON
        d.Q, d.R = params.Q_weight, params.R_weight
        # acados: stage costs x dt, terminal x 1 [EXT-UNVERIFIED]; NONLINEAR_LS is 1/2 |y|^2_W (cost_definition.py:61-81:
        # W = diag(Q, R) -> gradient Q J^T delta, R u; Gauss-Newton Hessian Q J^T J, R I), EXTERNAL is the expression itself
        # (cost_definition.py:91-96: Q |delta|^2 + R |u|^2 -> 2Q ..., 2R ...).  The engine differentiates Q|delta|^2 + R|u|^2.
        half = 0.5 if cost == 'nls' else 1.0
        d.cost_scale_stage, d.cost_scale_term = half * params.dt, half
        # controller.py:711 sets LM = 0 for the backup OCP, whose cost is zero as well: the QP Hessian vanishes and the
        # reference relies on HPIPM's internal primal regularisation [EXT-UNVERIFIED].  The engine states it: a 1e-4
        # diagonal makes the QP strictly convex (minimum-norm feasible correction of the guess) and well enough
        # conditioned that its solution is resolved at the IPM's 1e-8 exit tolerance.
        lm = 1e-4 / params.dt if controller == 'backup' else params.levenberg_marquardt
        d.lm_stage, d.lm_term = lm * params.dt, (1e-4 if controller == 'backup' else lm)

        # safe set (safe_set.py:72-104)
        d.nn_mode, d.nn_dof, d.nn_eps = nn_mode, params.n_dof_safe_set, params.eps
        d.nn_soft_e = getattr(params, soft_e) if soft_e else -1.0
        d.nn_soft_run = getattr(params, soft_run) if soft_run else -1.0
        for i in range(MAX_NQ):
            d.nn_mean[i], d.nn_std[i] = 0.0, 1.0

        # collision rows at node 0: kept by the reference unless the run is noisy (controller.py:68-79)
        d.rows_at_node0 = int(not float(getattr(params, 'noise', 0.0) or 0.0) > 0.0) if rows_at_node0 is None \
            else int(bool(rows_at_node0))
        d.qp_max_iter = params.qp_max_iter
        d.qp_tol, d.qp_mu0 = float(getattr(params, 'qp_tol', 1e-8)), 1.0
        # stall exit of the IPM (include/smpc.h): on for RealReceding, whose tubes make ~1 % of its QPs infeasible
        # the stall exit (include/smpc.h) is on where infeasible QPs are part of normal operation: RealReceding's tubes, and the backup
        # OCP (terminal zero velocity from the viable state of an aborting instance: in RealReceding's loop 27 of 29 such solves
        # are infeasible and used to run 34 iterations on average, 52 at worst, until the step length underflowed; feasible ones take 5)
        d.qp_stall_iters = int(getattr(params, 'qp_stall_iters', 24 if controller in ('real_receding', 'backup') else 0))
        d.qp_tol_res = float(getattr(params, 'qp_tol_res', 0.0))     # 0: same as qp_tol
        self.desc = d

    # -- helpers ----------------------------------------------------------------------------------------------------
    def set_normalisation(self, mean, std):
        n = self.params.n_dof_safe_set
        self.desc.nn_mean[:n] = np.asarray(mean, float).reshape(-1)[:n]
        self.desc.nn_std[:n] = np.asarray(std, float).reshape(-1)[:n]

    def _add_point(self, link, local):
        local = np.asarray(local, float)
        for i, (li, loc) in enumerate(self._points):
            if li == link and np.array_equal(loc, local):
                return i
        self._points.append((int(link), local))
        return len(self._points) - 1

    def _capsule_points(self, cap):
        """Two end points of a moving capsule in the frame of the actuated link that carries it
        (env_model.py:131-150: FK_link . Trans(spatial_offset) . Rx Ry Rz . e_i)."""
        cache = self.__dict__.setdefault('_cap_pts', {})
        if cap['name'] not in cache:
            li, lR, lp = self.chain.frame(cap['link_name'])
            R = np.eye(3)
            if cap.get('rotation_offset') is not None:
                th = cap['rotation_offset']
                R = _rot(0, th[0]) @ _rot(1, th[1]) @ _rot(2, th[2])
            off = np.zeros(3) if cap.get('spatial_offset') is None else np.array(cap['spatial_offset'], float)
            pts = []
            for e in cap['end_points']:
                pts.append(self._add_point(li, lp + lR @ (off + R @ np.asarray(e[:3], float))))
            cache[cap['name']] = tuple(pts)
        return cache[cap['name']]

    def _push_row(self, r, name, chk_lo, chk_hi):
        self.rows.append(r)
        self.row_names.append(name)
        self.row_check.append((chk_lo, chk_hi))

    # per-instance perturbed joint tables for the plant (utils.py:126-171 semantics on the lumped inertias is NOT what
    # the reference does -- it perturbs each URDF link before lumping; see noise.py)
    def joint_table(self):
        out = np.zeros(self.nq, JOINT_DTYPE)
        for i in range(self.nq):
            J = self.desc.joints[i]
            for f in ('R0', 'p0', 'axis', 'com', 'inertia'):
                out[i][f] = np.array(getattr(J, f))
            for f in ('mass', 'q_min', 'q_max', 'v_max', 'tau_max'):
                out[i][f] = getattr(J, f)
        return out
