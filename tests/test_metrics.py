"""scripts/metrics_count_fails.py (reference scripts/metrics_count_fails.py:19-28, 53-93): closed-loop cost and fail counts of
a result pickle, against a cost computed by hand."""
import importlib.util
import os
import pickle

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_script():
    spec = importlib.util.spec_from_file_location('metrics_count_fails', os.path.join(ROOT, 'scripts', 'metrics_count_fails.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _two_instance_pickle(tmp_path, params, prob, ee):
    """instance 0 completes (3 steps), instance 1 collides after the first step (NaN-padded like mpc.py:114)."""
    from safe_mpc_amd import closed_loop as cl
    rng = np.random.default_rng(0)
    T = 3
    q = rng.uniform(prob.lbx[:6], prob.ubx[:6], (2, T + 1, 6))
    x = np.concatenate([q, 0.1 * rng.standard_normal((2, T + 1, 6))], axis=2)
    u = rng.uniform(-2, 2, (2, T, 6))
    x[1, 2:], u[1, 1:] = np.nan, np.nan
    res = {'x': x, 'u': u, 'r': np.full((2, T, 1), np.nan), 'conv_idx': [], 'collisions_idx': [1], 'unconv_idx': [0],
           'viable_idx': [], 'x_viable': np.zeros((0, 12))}
    f = cl.result_file(params, 'z1', 'st', params.N, True, 0.0, 0.0, 0.0, 0.0)
    cl.save_pickle(f, res)
    # hand-computed cost of instance 0 (metrics_count_fails.py:19-28): stage terms for i < T plus the terminal EE term
    Q, R = params.Q_weight, params.R_weight
    cost = sum(Q * np.sum((ee(x[0, i, :6]) - prob.ee_ref) ** 2) + R * np.sum(u[0, i] ** 2) for i in range(T))
    cost += Q * np.sum((ee(x[0, T, :6]) - prob.ee_ref) ** 2)
    return cost


def _run(tmp_path, make_solver, ee_of):
    m = _load_script()
    from safe_mpc_amd.parser import Parameters, default_args
    from safe_mpc_amd.problem import OcpProblem
    params = Parameters({**default_args(), 'horizon': 30, 'alpha': 10.0}, 'z1', rti=True)
    params.N = 30
    params.DATA_DIR = os.path.join(str(tmp_path), '')
    prob = OcpProblem(params, 'naive', 'ext', N=2)
    expect = _two_instance_pickle(tmp_path, params, prob, ee_of(prob))
    scores = m.main(['-c', 'st', '-c', 'htwa', '--horizon', '30', '--alpha', '10', '--data_dir', str(tmp_path)], make_solver=make_solver)
    s = scores['st']
    assert s['fails'] == 1 and s['completed_idx'] == [0] and s['score'] == 0
    assert s['costs'][1] == -100.0                                     # metrics_count_fails.py:70-71
    assert abs(s['costs'][0] - expect) < 1e-9 * abs(expect)
    assert scores['htwa']['fails'] == 0 and np.isnan(scores['htwa']['costs'][0])   # missing file (:75-79)
    out = m.scores_file(params, 'z1', 30, 10.0, 0.0, 0.0, 0.0, 0.0)
    assert out.endswith('z1_30hor_10sm_noise0.0_control_noise0.0_q_collision_margins_0.0_0.0_scores.pkl')
    on_disk = pickle.load(open(out, 'rb'))
    assert set(on_disk['st']) >= {'score', 'fails', 'costs', 'completed_idx'}


def _ee_by_oracle(prob):
    from oracle.oracle import Oracle
    o = Oracle(prob)
    return lambda q: o.points(q)[prob.desc.ee_point]


def test_metrics_count_fails_on_oracle_double(tmp_path):
    from fake_solver import OracleSolver
    _run(tmp_path, lambda prob: OracleSolver(prob, None), _ee_by_oracle)


@pytest.mark.gpu
def test_metrics_count_fails_on_engine(tmp_path):
    _run(tmp_path, None, _ee_by_oracle)
