"""Loads the HIP engine (csrc/libsmpc_hip.so) through ctypes and declares the C ABI of include/smpc.h.

There is no CPU fallback: if the library is missing or no MI355X is visible, creating a solver raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

from .problem import Joint, NodeEval, ProblemDesc

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
# SMPC_HIP_LIB selects another build of the SAME engine (diagnostic builds of scripts/qp_phase_profile.py)
LIB_PATH = os.environ.get('SMPC_HIP_LIB') or os.path.join(_CSRC, 'libsmpc_hip.so')

SYMBOLS = ['smpc_create', 'smpc_destroy', 'smpc_abi_version', 'smpc_last_error', 'smpc_set_mlp', 'smpc_set_horizon',
           'smpc_set_stage_bounds', 'smpc_set_slack_weights', 'smpc_set_instance_bounds', 'smpc_solve_batch', 'smpc_eval_nodes', 'smpc_guess_correction',
           'smpc_provide_control', 'smpc_check_trajectory', 'smpc_plant_step', 'smpc_rollout_batch', 'smpc_sync', 'smpc_stream',
           'smpc_enable_timing', 'smpc_get_timing', 'smpc_get_qp_timing', 'smpc_get_qp_wave_stats', 'smpc_policy_step', 'smpc_loop_pre',
           'smpc_loop_post', 'smpc_loop_apply_backup', 'smpc_loop_classify_aborts', 'smpc_get_timing_history',
           'smpc_accumulate_stats', 'smpc_set_mlp_activation', 'smpc_set_qp_mode']


class EngineError(RuntimeError):
    pass


_vp = C.c_void_p


class PolicyParams(C.Structure):
    """smpc_policy_params (include/smpc.h)"""
    _fields_ = [('kind', C.c_int32), ('abort_flag', C.c_int32), ('collision_first_node', C.c_int32), ('reserved0', C.c_int32),
                ('tol_x', C.c_double), ('alpha', C.c_double), ('tol_safe', C.c_double), ('tube', C.c_double),
                ('x_min', _vp), ('x_max', _vp), ('row_lb_chk', _vp), ('row_ub_chk', _vp), ('stage_lo', _vp), ('stage_hi', _vp)]


class PolicyState(C.Structure):
    """smpc_policy_state"""
    _fields_ = [(k, _vp) for k in ('x_guess', 'u_guess', 'x_temp', 'u_temp', 'p', 'x_viable', 'fails', 'current_step', 'r',
                                   'status', 'qp_iter', 'traj')] + [('traj_len', C.c_int64)]


class LoopState(C.Structure):
    """smpc_loop_state"""
    _fields_ = [(k, _vp) for k in ('x_cur', 'alive', 'sa', 'collided', 'ja', 'last_x', 'last_u', 'x_abort', 'u_abort', 'step',
                                   'x_log', 'u_log', 'r_log', 'resumed')]


def build(force=False):
    """Compile the engine for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith(('.hip', '.hpp'))]
    srcs.append(os.path.join(_CSRC, '..', '..', 'include', 'smpc.h'))
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return LIB_PATH
    subprocess.check_call(['make', '-C', _CSRC, '-B', 'libsmpc_hip.so'])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError(f'{LIB_PATH} not found: run `python -c "import __graft_entry__ as g; g.build()"` '
                          f'(or make -C safe_mpc_amd/csrc); the engine has no CPU fallback')
    L = C.CDLL(LIB_PATH)
    vp, i32p, dp = C.c_void_p, C.POINTER(C.c_int32), C.c_void_p
    L.smpc_create.argtypes = [C.POINTER(ProblemDesc), C.c_int, C.POINTER(vp)]
    L.smpc_destroy.argtypes = [vp]
    L.smpc_destroy.restype = None
    L.smpc_last_error.argtypes = [vp]
    L.smpc_last_error.restype = C.c_char_p
    L.smpc_set_mlp.argtypes = [vp, C.c_int, i32p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int]
    L.smpc_set_horizon.argtypes = [vp, C.c_int]
    L.smpc_set_stage_bounds.argtypes = [vp, dp, dp]
    L.smpc_set_slack_weights.argtypes = [vp, dp]
    L.smpc_set_instance_bounds.argtypes = [vp, C.c_int, dp, dp, C.c_int]
    L.smpc_solve_batch.argtypes = [vp, C.c_int, dp, dp, dp, dp, dp, dp, dp, dp, C.c_int]
    L.smpc_eval_nodes.argtypes = [vp, C.c_int, dp, dp, dp, dp, C.c_int]
    L.smpc_guess_correction.argtypes = [vp, C.c_int, dp, dp, C.c_int]
    L.smpc_provide_control.argtypes = [vp, C.c_int, dp, dp, dp, dp, dp, dp, C.c_int]
    L.smpc_check_trajectory.argtypes = [vp, C.c_int, C.c_int, dp, dp, dp, C.c_double, dp, dp, C.c_double, C.c_double,
                                        dp, dp, C.c_int]
    L.smpc_plant_step.argtypes = [vp, C.c_int, dp, dp, dp, dp, dp, dp, C.c_int]
    L.smpc_rollout_batch.argtypes = [vp, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, dp, dp, dp, dp, C.c_int]
    L.smpc_sync.argtypes = [vp]
    L.smpc_stream.argtypes = [vp]
    L.smpc_stream.restype = C.c_void_p
    L.smpc_enable_timing.argtypes = [vp, C.c_int]
    L.smpc_get_timing.argtypes = [vp, C.POINTER(C.c_float)]
    L.smpc_get_qp_timing.argtypes = [vp, C.POINTER(C.c_float)]
    L.smpc_get_qp_wave_stats.argtypes = [vp, C.POINTER(C.c_double)]
    L.smpc_policy_step.argtypes = [vp, C.c_int, C.POINTER(PolicyParams), C.POINTER(PolicyState), dp, dp, dp, dp, dp, dp]
    L.smpc_loop_pre.argtypes = [vp, C.c_int, C.c_int, C.POINTER(LoopState), dp, dp, dp, dp]
    L.smpc_loop_apply_backup.argtypes = [vp, C.c_int, C.c_int, C.POINTER(LoopState), C.c_int, dp, dp, dp, dp, dp, dp, dp]
    L.smpc_loop_post.argtypes = [vp, C.c_int, C.POINTER(PolicyParams), C.POINTER(LoopState), dp, dp, dp]
    L.smpc_loop_classify_aborts.argtypes = [vp, C.c_int, C.POINTER(LoopState), C.c_int, dp, dp]
    L.smpc_get_timing_history.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
    L.smpc_accumulate_stats.argtypes = [vp, C.c_int, dp, dp, dp]
    L.smpc_set_mlp_activation.argtypes = [vp, C.c_int]
    L.smpc_set_qp_mode.argtypes = [vp, C.c_int]
    _lib = L
    return L
