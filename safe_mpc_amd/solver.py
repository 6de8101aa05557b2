"""``BatchedOcpSolver``: Python face of the HIP engine (include/smpc.h) -- what ``self.ocp_solver`` is in the reference
(an ``acados_template.AcadosOcpSolver``, controller.py:247), but for B instances per call.

Inputs may be numpy arrays (host path: copied in and out by the engine, synchronous) or ROCm torch tensors
(device path: the engine only gets ``data_ptr()``s and enqueues on its own stream; call :meth:`sync`).
PyTorch is used for device memory only.
"""
from __future__ import annotations

import contextlib
import ctypes as C

import numpy as np

from . import _lib
from .problem import JOINT_DTYPE, NODE_EVAL_DTYPE, OcpProblem


def _is_torch(a):
    return type(a).__module__.startswith('torch')


class BatchedOcpSolver:
    def __init__(self, problem: OcpProblem, net=None, device=0):
        self.problem = problem
        self.L = _lib.lib()
        self.device = int(device)
        h = C.c_void_p()
        rc = self.L.smpc_create(C.byref(problem.desc), self.device, C.byref(h))
        if rc != 0:
            raise _lib.EngineError(f'smpc_create failed ({rc}): {self.L.smpc_last_error(None).decode()}')
        self.h = h
        self.nq, self.nx, self.nu = problem.nq, problem.nx, problem.nu
        self.N = problem.N
        self.net = None
        if net is not None:
            self.set_mlp(net)

    # -- lifetime ------------------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, 'h', None):
            self.L.smpc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise _lib.EngineError(f'engine error {rc}: {self.L.smpc_last_error(self.h).decode()}')

    @contextlib.contextmanager
    def _ordered(self, dev):
        """Stream contract of the device path ("Streams"): the engine enqueues on its OWN non-blocking
        stream, torch on its current stream.  On entry the engine's stream waits for everything already enqueued on
        torch's current stream (inputs produced / buffers zero-filled by torch kernels); on exit torch's current stream
        waits for the engine's work, so later torch ops -- and the caching allocator's reuse of temporaries freed after
        the call -- are ordered behind it.  No host synchronisation.  A caller that already runs under
        ``torch.cuda.stream(ExternalStream(smpc_stream))`` (bench.py) pays nothing."""
        if not dev:
            yield
            return
        import torch
        if getattr(self, '_ext_stream', None) is None:
            self._ext_stream = torch.cuda.ExternalStream(self.L.smpc_stream(self.h), device=torch.device('cuda', self.device))
        cur = torch.cuda.current_stream(self.device)
        same = cur.cuda_stream == self._ext_stream.cuda_stream
        if not same:
            self._ext_stream.wait_stream(cur)
        try:
            yield
        finally:
            if not same:
                cur.wait_stream(self._ext_stream)

    def set_mlp(self, net):
        """net: SafeSetNet (weights as numpy fp32) -- or anything with .weights/.biases lists of [out, in] / [out]."""
        Ws = [np.ascontiguousarray(w, np.float32) for w in net.weights]
        bs = [np.ascontiguousarray(b, np.float32) for b in net.biases]
        n = len(Ws)
        dims = np.array([Ws[0].shape[1]] + [w.shape[0] for w in Ws], np.int32)
        Wp = (C.c_void_p * n)(*[w.ctypes.data for w in Ws])
        bp = (C.c_void_p * n)(*[b.ctypes.data for b in bs])
        self._chk(self.L.smpc_set_mlp(self.h, n, dims.ctypes.data_as(C.POINTER(C.c_int32)), Wp, bp, 0))
        act = getattr(net, 'act', 'gelu')
        from .safe_set import SafeSetNet
        self._chk(self.L.smpc_set_mlp_activation(self.h, SafeSetNet.ACT_CODES[act]))      # SMPC_ACT_* (parser.py:95-102)
        self.net = net

    QP_MODES = {'auto': -1, 'throughput': 0, 'latency': 1}

    def set_qp_mode(self, mode):
        """Which form of the QP solve this handle launches (smpc_set_qp_mode): 'auto' (by batch size), 'throughput' (k_qp_ipm, a
        wavefront per two instances) or 'latency' (k_qp_ipm_wg, a workgroup per instance).  Same result to rounding."""
        self._chk(self.L.smpc_set_qp_mode(self.h, self.QP_MODES[mode] if isinstance(mode, str) else int(mode)))

    def set_horizon(self, N):
        self._chk(self.L.smpc_set_horizon(self.h, int(N)))
        self.N = int(N)

    def set_stage_bounds(self, lo=None, hi=None):
        if lo is None:
            self._chk(self.L.smpc_set_stage_bounds(self.h, None, None))
            return
        lo = np.ascontiguousarray(lo, np.float64)
        hi = np.ascontiguousarray(hi, np.float64)
        assert lo.shape == (self.N + 1, self.nx) and hi.shape == lo.shape
        self._chk(self.L.smpc_set_stage_bounds(self.h, lo.ctypes.data, hi.ctypes.data))

    def set_slack_weights(self, zl=None):
        """cost_set(k, 'zl', v) for every node at once (controller.py:455-468): zl[N+1], None restores the formulation's."""
        if zl is None:
            self._chk(self.L.smpc_set_slack_weights(self.h, None))
            return
        zl = np.ascontiguousarray(zl, np.float64)
        assert zl.shape == (self.N + 1,)
        self._chk(self.L.smpc_set_slack_weights(self.h, zl.ctypes.data))

    def set_instance_bounds(self, lo=None, hi=None):
        """Per-instance stage bounds [B, N+1, nx] (RealReceding's state tube, controller.py:530-536); None clears."""
        if lo is None:
            self._chk(self.L.smpc_set_instance_bounds(self.h, 0, None, None, 0))
            return
        B = lo.shape[0]
        ptrs, dev, keep = self._prep([lo, hi], [(B, self.N + 1, self.nx)] * 2)
        with self._ordered(dev):
            self._chk(self.L.smpc_set_instance_bounds(self.h, B, ptrs[0], ptrs[1], dev))

    def sync(self):
        self._chk(self.L.smpc_sync(self.h))

    def enable_timing(self, on=True):
        """True / 1: HIP events + the load-balance probe inside k_qp_ipm; 2: events only; False / 0: off"""
        self._chk(self.L.smpc_enable_timing(self.h, int(on)))

    def timing(self):
        ms = (C.c_float * 4)()
        self._chk(self.L.smpc_get_timing(self.h, ms))
        q = (C.c_float * 2)()
        self._chk(self.L.smpc_get_qp_timing(self.h, q))
        w = (C.c_double * 3)()
        self._chk(self.L.smpc_get_qp_wave_stats(self.h, w))
        return {'time_lin': ms[0] * 1e-3, 'time_nn': ms[1] * 1e-3, 'time_qp': ms[2] * 1e-3, 'time_tot': ms[3] * 1e-3,
                'time_qp_setup': q[0] * 1e-3, 'time_qp_ipm': q[1] * 1e-3,
                'qp_wave_busy_mean': w[0] * 1e-6, 'qp_wave_span': w[1] * 1e-6}

    def timing_history(self, back=0):
        """per-kernel times of the solve ``back`` solves before the last one (the engine keeps 64), or None if it has not
        finished / does not exist; never waits"""
        ms = (C.c_float * 6)()
        self._chk(self.L.smpc_get_timing_history(self.h, int(back), ms))
        if ms[5] == 0.0:
            return None
        return {'time_lin': ms[0] * 1e-3, 'time_nn': ms[1] * 1e-3, 'time_qp_setup': ms[2] * 1e-3, 'time_qp_ipm': ms[3] * 1e-3,
                'time_tot': ms[4] * 1e-3}

    def accumulate_stats(self, status, qp_iter, acc):
        """acc (3 x int64 on the device) += [sum of IPM iterations, failed solves, solves] -- no host round trip"""
        with self._ordered(1):
            self._chk(self.L.smpc_accumulate_stats(self.h, int(status.shape[0]), status.data_ptr(), qp_iter.data_ptr() if qp_iter is not None else None,
                                                   acc.data_ptr()))

    # -- argument plumbing -----------------------------------------------------------------------------------------------
    def _prep(self, arrs, shapes, dtypes=None):
        """returns (pointers, on_device, keepalive)"""
        dev = _is_torch(arrs[0])
        ptrs, keep = [], []
        for i, (a, shp) in enumerate(zip(arrs, shapes)):
            if a is None:
                ptrs.append(None)
                continue
            if _is_torch(a) != dev:
                raise TypeError('mix of torch and numpy arguments')
            if dev:
                if not a.is_cuda or not a.is_contiguous():
                    raise ValueError('device path needs contiguous ROCm tensors')
                if a.device.index != self.device:
                    raise ValueError(f'tensor on device {a.device.index}, solver on {self.device}')
                if tuple(a.shape) != tuple(shp):
                    raise ValueError(f'argument {i}: shape {tuple(a.shape)} != {tuple(shp)}')
                ptrs.append(a.data_ptr())
                keep.append(a)
            else:
                dt = np.float64 if dtypes is None else dtypes[i]
                b = np.ascontiguousarray(a, dt)
                if tuple(b.shape) != tuple(shp):
                    raise ValueError(f'argument {i}: shape {tuple(b.shape)} != {tuple(shp)}')
                ptrs.append(b.ctypes.data)
                keep.append(b)
        return ptrs, int(dev), keep

    # -- the hot path ------------------------------------------------------------------------------------------------------
    def solve(self, x0, x_guess, u_guess, p, out=None):
        """One SQP-RTI solve per instance (controller.py:136-167).  Returns (x, u, status, qp_iter)."""
        B = x0.shape[0]
        N, nx, nu = self.N, self.nx, self.nu
        if B == 0:      # an empty batch is a loop over no instances (scripts/mpc.py:102), not an error
            if _is_torch(x0):
                import torch
                kw = dict(device=x0.device)
                return (torch.empty((0, N + 1, nx), dtype=torch.float64, **kw), torch.empty((0, N, nu), dtype=torch.float64, **kw),
                        torch.empty((0,), dtype=torch.int32, **kw), torch.empty((0,), dtype=torch.int32, **kw))
            return np.empty((0, N + 1, nx)), np.empty((0, N, nu)), np.empty(0, np.int32), np.empty(0, np.int32)
        shapes = [(B, nx), (B, N + 1, nx), (B, N, nu), (B, N + 1, 5)]
        ptrs, dev, keep = self._prep([x0, x_guess, u_guess, p], shapes)
        if dev:
            import torch
            if out is None:
                kw = dict(device=x0.device)
                out = (torch.empty((B, N + 1, nx), dtype=torch.float64, **kw),
                       torch.empty((B, N, nu), dtype=torch.float64, **kw),
                       torch.empty((B,), dtype=torch.int32, **kw), torch.empty((B,), dtype=torch.int32, **kw))
            op = [o.data_ptr() for o in out]
        else:
            if out is None:
                out = (np.empty((B, N + 1, nx)), np.empty((B, N, nu)), np.empty(B, np.int32), np.empty(B, np.int32))
            op = [o.ctypes.data for o in out]
        with self._ordered(dev):
            self._chk(self.L.smpc_solve_batch(self.h, B, *ptrs, *op, dev))
        return out

    def eval_nodes(self, x_guess, u_guess, p):
        B = x_guess.shape[0]
        N, nx, nu = self.N, self.nx, self.nu
        ptrs, dev, keep = self._prep([x_guess, u_guess, p], [(B, N + 1, nx), (B, N, nu), (B, N + 1, 5)])
        if dev:
            # device path: a dict of float64 views into one [B, N+1, sizeof(smpc_node_eval)/8] tensor, same field names
            import torch
            nd = NODE_EVAL_DTYPE.itemsize // 8
            raw = torch.zeros((B, N + 1, nd), dtype=torch.float64, device=x_guess.device)
            with self._ordered(1):
                self._chk(self.L.smpc_eval_nodes(self.h, B, *ptrs, raw.data_ptr(), 1))
            out = {}
            for name in NODE_EVAL_DTYPE.names:
                dt, off = NODE_EVAL_DTYPE.fields[name][:2]
                n = dt.itemsize // 8
                out[name] = raw[..., off // 8: off // 8 + n] if dt.shape else raw[..., off // 8]
            return out
        out = np.zeros((B, N + 1), NODE_EVAL_DTYPE)
        self._chk(self.L.smpc_eval_nodes(self.h, B, *ptrs, out.ctypes.data, 0))
        return out

    # -- callers around the solve (a13, a15, a16) ------------------------------------------------------------------------
    def guess_correction(self, x_guess, u_guess):
        """In place on torch tensors; returns a corrected copy for numpy."""
        B = x_guess.shape[0]
        if not _is_torch(x_guess):
            x_guess = np.array(x_guess, np.float64, order='C', copy=True)
        ptrs, dev, keep = self._prep([x_guess, u_guess], [(B, self.N + 1, self.nx), (B, self.N, self.nu)])
        with self._ordered(dev):
            self._chk(self.L.smpc_guess_correction(self.h, B, *ptrs, dev))
        return x_guess

    def provide_control(self, accept, x_temp, u_temp, x_guess, u_guess):
        B = x_temp.shape[0]
        N, nx, nu = self.N, self.nx, self.nu
        if not _is_torch(x_guess):
            x_guess = np.array(x_guess, np.float64, order='C', copy=True)
            u_guess = np.array(u_guess, np.float64, order='C', copy=True)
            u_apply = np.empty((B, nu))
            accept = np.ascontiguousarray(accept, np.int32)
        else:
            import torch
            u_apply = torch.empty((B, nu), dtype=torch.float64, device=x_guess.device)
        ptrs, dev, keep = self._prep([accept, x_temp, u_temp, x_guess, u_guess, u_apply],
                                     [(B,), (B, N + 1, nx), (B, N, nu), (B, N + 1, nx), (B, N, nu), (B, nu)],
                                     [np.int32] + [np.float64] * 5)
        if not dev:
            # _prep may have re-wrapped the arrays; make sure outputs are the ones we return
            ptrs[3], ptrs[4], ptrs[5] = x_guess.ctypes.data, u_guess.ctypes.data, u_apply.ctypes.data
        with self._ordered(dev):
            self._chk(self.L.smpc_provide_control(self.h, B, *ptrs, dev))
        return x_guess, u_guess, u_apply

    def check_trajectory(self, x, x_min=None, x_max=None, tol_x=None, row_lb=None, row_ub=None, alpha=None,
                         tol_safe=None, want_nn=False):
        """checkStateConstraints (env_model.py:170-173) per instance; optionally the safe-set test per node."""
        pr, par = self.problem, self.problem.params
        x_min = pr.x_min if x_min is None else x_min
        x_max = pr.x_max if x_max is None else x_max
        tol_x = par.tol_x if tol_x is None else tol_x
        row_lb = pr.row_check[:, 0] if row_lb is None else row_lb
        row_ub = pr.row_check[:, 1] if row_ub is None else row_ub
        alpha = par.alpha if alpha is None else alpha
        tol_safe = par.tol_safe_set if tol_safe is None else tol_safe
        B, n_nodes = x.shape[0], x.shape[1]
        small = [np.ascontiguousarray(a, np.float64) for a in (x_min, x_max, row_lb, row_ub)]
        if _is_torch(x):
            import torch
            ok = torch.empty((B,), dtype=torch.int32, device=x.device)
            nn = torch.empty((B, n_nodes), dtype=torch.int32, device=x.device) if want_nn else None
            with self._ordered(1):
                self._chk(self.L.smpc_check_trajectory(self.h, B, n_nodes, x.data_ptr(), small[0].ctypes.data,
                                                       small[1].ctypes.data, tol_x, small[2].ctypes.data,
                                                       small[3].ctypes.data, alpha, tol_safe, ok.data_ptr(),
                                                       nn.data_ptr() if want_nn else None, 1))
            return (ok, nn) if want_nn else ok
        xx = np.ascontiguousarray(x, np.float64)
        ok = np.empty(B, np.int32)
        nn = np.empty((B, n_nodes), np.int32) if want_nn else None
        self._chk(self.L.smpc_check_trajectory(self.h, B, n_nodes, xx.ctypes.data, small[0].ctypes.data,
                                               small[1].ctypes.data, tol_x, small[2].ctypes.data, small[3].ctypes.data,
                                               alpha, tol_safe, ok.ctypes.data, nn.ctypes.data if want_nn else None, 0))
        return (ok.astype(bool), nn.astype(bool)) if want_nn else ok.astype(bool)

    def rollout(self, x0, x_guess, u_guess, p, n_steps, joints_noisy=None, tau_noise=None):
        """n_steps of the plain RTI policy + plant without a host round trip per step (smpc_rollout_batch).
        x_guess / u_guess are updated in place (torch) or returned updated (numpy).  Returns step-major
        (x_traj[n+1, B, nx], u_traj[n, B, nu], status[n, B], qp_iter[n, B], x_guess, u_guess)."""
        B, n = x0.shape[0], int(n_steps)
        N, nx, nu = self.N, self.nx, self.nu
        if _is_torch(x0):
            import torch
            kw = dict(device=x0.device)
            xt = torch.empty((n + 1, B, nx), dtype=torch.float64, **kw)
            ut = torch.empty((n, B, nu), dtype=torch.float64, **kw)
            st = torch.empty((n, B), dtype=torch.int32, **kw)
            it = torch.empty((n, B), dtype=torch.int32, **kw)
            jn = joints_noisy.data_ptr() if joints_noisy is not None else None
            tn = tau_noise.data_ptr() if tau_noise is not None else None
            for a, shp in ((x0, (B, nx)), (x_guess, (B, N + 1, nx)), (u_guess, (B, N, nu)), (p, (B, N + 1, 5))):
                if tuple(a.shape) != shp or a.dtype != torch.float64 or not a.is_contiguous():
                    raise ValueError(f'rollout: expected a contiguous float64 tensor of shape {shp}')
            with self._ordered(1):
                self._chk(self.L.smpc_rollout_batch(self.h, B, n, x0.data_ptr(), x_guess.data_ptr(), u_guess.data_ptr(), p.data_ptr(),
                                                    jn, tn, xt.data_ptr(), ut.data_ptr(), st.data_ptr(), it.data_ptr(), 1))
            return xt, ut, st, it, x_guess, u_guess
        x0 = np.ascontiguousarray(x0, np.float64)
        xg = np.array(x_guess, np.float64, order='C', copy=True)
        ug = np.array(u_guess, np.float64, order='C', copy=True)
        pp = np.ascontiguousarray(p, np.float64)
        assert x0.shape == (B, nx) and xg.shape == (B, N + 1, nx) and ug.shape == (B, N, nu) and pp.shape == (B, N + 1, 5)
        xt, ut = np.empty((n + 1, B, nx)), np.empty((n, B, nu))
        st, it = np.empty((n, B), np.int32), np.empty((n, B), np.int32)
        jn = tn = None
        if joints_noisy is not None:
            joints_noisy = np.ascontiguousarray(joints_noisy, JOINT_DTYPE)
            assert joints_noisy.shape == (B, self.nq)
            jn = joints_noisy.ctypes.data
        if tau_noise is not None:
            tau_noise = np.ascontiguousarray(tau_noise, np.float64)
            assert tau_noise.shape == (n, B, nu)
            tn = tau_noise.ctypes.data
        self._chk(self.L.smpc_rollout_batch(self.h, B, n, x0.ctypes.data, xg.ctypes.data, ug.ctypes.data, pp.ctypes.data, jn, tn,
                                            xt.ctypes.data, ut.ctypes.data, st.ctypes.data, it.ctypes.data, 0))
        return xt, ut, st, it, xg, ug

    # -- the policy layer with all state in HBM (smpc_policy_step / smpc_loop_pre / smpc_loop_post) -------------------------------
    def _policy_params(self, kind=0, abort_flag=0, tube=0.0, stage_lo=None, stage_hi=None):
        """smpc_policy_params from the problem's parameters; the small host arrays it points to are kept alive here"""
        pr, par = self.problem, self.problem.params
        small = [np.ascontiguousarray(a, np.float64) for a in (pr.x_min, pr.x_max, pr.row_check[:, 0], pr.row_check[:, 1])]
        pp = _lib.PolicyParams(int(kind), int(bool(abort_flag)), int(bool(getattr(par, 'reference_quirks', True))), 0,
                               float(par.tol_x), float(par.alpha), float(par.tol_safe_set), float(tube),
                               small[0].ctypes.data, small[1].ctypes.data, small[2].ctypes.data, small[3].ctypes.data,
                               stage_lo.data_ptr() if stage_lo is not None else None,
                               stage_hi.data_ptr() if stage_hi is not None else None)
        pp._keep = small
        return pp

    def policy_step(self, ctrl, x, stepping=None, u_other=None, u_out=None):
        """<Controller>.step(x) of ``ctrl`` (device state) as engine kernels.  Returns (u, abort) -- persistent tensors of the
        controller -- and leaves "did any instance abort" in ``ctrl._any_abort`` (one int32 on the device)."""
        ptr = lambda t: t.data_ptr() if t is not None else None
        pp = self._policy_params(ctrl.policy_kind, getattr(ctrl, 'abort_flag', False), getattr(ctrl, 'TUBE', 0.0),
                                 getattr(ctrl, '_stage_lo', None), getattr(ctrl, '_stage_hi', None))
        st = _lib.PolicyState(ptr(ctrl.x_guess), ptr(ctrl.u_guess), ptr(ctrl.x_temp), ptr(ctrl.u_temp), ptr(ctrl.p), ptr(ctrl.x_viable),
                              ptr(ctrl.fails), ptr(ctrl.current_step), ptr(getattr(ctrl, 'r', None)), ptr(ctrl.last_status),
                              ptr(ctrl.qp_iter), ptr(getattr(ctrl, 'traj', None)),
                              int(ctrl.traj.shape[1]) if getattr(ctrl, 'traj', None) is not None else 0)
        u_out = ctrl._u_out if u_out is None else u_out
        with self._ordered(1):
            self._chk(self.L.smpc_policy_step(self.h, ctrl.B, C.byref(pp), C.byref(st), x.data_ptr(), ptr(stepping), ptr(u_other),
                                              u_out.data_ptr(), ctrl._abort_out.data_ptr(), ctrl._any_abort.data_ptr()))
        return u_out, ctrl._abort_out

    def _loop_state(self, g):
        ptr = lambda t: t.data_ptr() if t is not None else None
        return _lib.LoopState(ptr(g.x_cur), ptr(g.alive), ptr(g.sa), ptr(g.collided), ptr(g.ja), ptr(g.last_x), ptr(g.last_u),
                              ptr(g.x_abort), ptr(g.u_abort), ptr(g._jt), ptr(g.x_log), ptr(g.u_log), ptr(g.r_log),
                              ptr(getattr(g, '_resumed', None)))

    def loop_pre(self, g, r, pending, u_other, stepping):
        """scripts/mpc.py:130-151 for the group ``g`` (closed_loop._Group, device state)"""
        ls = self._loop_state(g)
        with self._ordered(1):
            self._chk(self.L.smpc_loop_pre(self.h, g._B, g._Nb, C.byref(ls), r.data_ptr() if r is not None else None,
                                           pending.data_ptr() if pending is not None else None, u_other.data_ptr(), stepping.data_ptr()))

    def loop_classify_aborts(self, g, abort, any_event):
        """scripts/mpc.py:137-141 vs 161-190: which of this step's aborts open an abort event (smpc_loop_classify_aborts)"""
        ls = self._loop_state(g)
        quirks = int(bool(getattr(self.problem.params, 'reference_quirks', True)))
        with self._ordered(1):
            self._chk(self.L.smpc_loop_classify_aborts(self.h, g._B, C.byref(ls), quirks, abort.data_ptr(), any_event.data_ptr()))

    def loop_apply_backup(self, g, rows, status_c, x_c, u_c, viable, u, pending):
        """scripts/mpc.py:161-190 (second half) for the previous step's abort events, see smpc_loop_apply_backup"""
        ls = self._loop_state(g)
        with self._ordered(1):
            self._chk(self.L.smpc_loop_apply_backup(self.h, g._B, g._Nb, C.byref(ls), int(rows.shape[0]), rows.data_ptr(), status_c.data_ptr(),
                                                    x_c.data_ptr(), u_c.data_ptr(), viable.data_ptr(), u.data_ptr(), pending.data_ptr()))

    def loop_post(self, g, u, joints_noisy=None, tau_noise=None):
        """scripts/mpc.py:240-264: plant, outcome tests, logs, j += 1"""
        ls, pp = self._loop_state(g), self._policy_params()
        with self._ordered(1):
            self._chk(self.L.smpc_loop_post(self.h, g._B, C.byref(pp), C.byref(ls), u.data_ptr(),
                                            joints_noisy.data_ptr() if joints_noisy is not None else None,
                                            tau_noise.data_ptr() if tau_noise is not None else None))

    def plant_step(self, x, u, joints_noisy=None, tau_noise=None, out=None):
        """AdamModel.integrate (env_model.py:192-206) for B instances.  ``out`` (device path): (x_next, u_eff) tensors to fill."""
        B = x.shape[0]
        if _is_torch(x):
            import torch
            xn, ue = out if out is not None else (torch.empty_like(x), torch.empty_like(u))
            jn = joints_noisy.data_ptr() if joints_noisy is not None else None
            tn = tau_noise.data_ptr() if tau_noise is not None else None
            with self._ordered(1):
                self._chk(self.L.smpc_plant_step(self.h, B, x.data_ptr(), u.data_ptr(), jn, tn, xn.data_ptr(),
                                                 ue.data_ptr(), 1))
            return xn, ue
        xx, uu = np.ascontiguousarray(x, np.float64), np.ascontiguousarray(u, np.float64)
        xn, ue = np.empty_like(xx), np.empty_like(uu)
        jn = tn = None
        if joints_noisy is not None:
            joints_noisy = np.ascontiguousarray(joints_noisy, JOINT_DTYPE)
            assert joints_noisy.shape == (B, self.nq)
            jn = joints_noisy.ctypes.data
        if tau_noise is not None:
            tau_noise = np.ascontiguousarray(tau_noise, np.float64)
            tn = tau_noise.ctypes.data
        self._chk(self.L.smpc_plant_step(self.h, B, xx.ctypes.data, uu.ctypes.data, jn, tn, xn.ctypes.data,
                                         ue.ctypes.data, 0))
        return xn, ue
