#!/usr/bin/env python3
"""Scores of closed-loop runs -- the live metrics script of the reference (scripts/metrics_count_fails.py:19-28,53-93):
closed-loop cost  sum_j Q |ee(x_j) - ee_ref|^2 + R |u_j|^2  of every completed instance and failure counts, written as
{controller: {'score','fails','costs','completed_idx'}} to *_scores.pkl (consumed by plot_data_noise.py:84,142).

    python scripts/metrics_count_fails.py --horizon 30 --alpha 10 [--noise 5] -c st -c htwa ...
"""
import argparse
import os
import pickle
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_mpc_amd import closed_loop as cl                      # noqa: E402
from safe_mpc_amd.parser import Parameters, default_args        # noqa: E402
from safe_mpc_amd.problem import OcpProblem                     # noqa: E402
from safe_mpc_amd.solver import BatchedOcpSolver                # noqa: E402


def closed_loop_costs(params, prob, solver, x, u):
    """metrics_count_fails.py:19-28: sum_{i<T} (Q |ee(x_i) - ee_ref|^2 + R |u_i|^2) + Q |ee(x_T) - ee_ref|^2 per instance.
    x [n, T+1, nx], u [n, T, nu]; one batched FK evaluation for all n (T+1) states."""
    n, T = u.shape[0], u.shape[1]
    nodes = np.nan_to_num(x[:, :T + 1]).reshape(-1, 1, x.shape[2])
    dummy_u = np.zeros((nodes.shape[0], solver.N, prob.nu))
    p = np.zeros((nodes.shape[0], solver.N + 1, 5))
    ee = np.asarray(solver.eval_nodes(np.repeat(nodes, solver.N + 1, 1), dummy_u, p)['ee'])[:, 0, :].reshape(n, T + 1, 3)
    err = np.sum((ee - prob.ee_ref) ** 2, axis=2)
    return params.Q_weight * err.sum(1) + params.R_weight * np.sum(np.nan_to_num(u) ** 2, axis=(1, 2))


def score_results(params, prob, solver, res):
    """One controller's entry of the scores dict (metrics_count_fails.py:63-93): every instance has a cost, -100 marking
    the failed ones (:70-71); 'score' is written as 0 by the reference (:87) -- the mean over completed runs is added as
    'mean_cost'."""
    n = res['x'].shape[0]
    failed = sorted(int(i) for i in res['collisions_idx'])
    done = sorted(set(range(n)) - set(failed))
    costs = np.full(n, -100.0)
    if done:
        costs[done] = closed_loop_costs(params, prob, solver, res['x'][done], res['u'][done])
    return {'score': 0, 'fails': len(failed), 'costs': costs.tolist(), 'completed_idx': done,
            'mean_cost': float(np.mean(costs[done])) if done else float('nan')}


def scores_file(params, model_name, hor, alpha, noise, control_noise, jm, cm):
    """metrics_count_fails.py:90"""
    return (f'{params.DATA_DIR}{model_name}_{hor}hor_{int(alpha)}sm_noise{noise}_control_noise{control_noise}'
            f'_q_collision_margins_{jm}_{cm}_scores.pkl')


def main(argv=None, make_solver=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('-c', '--controller', action='append', default=None)
    ap.add_argument('--horizon', type=int, default=45)
    ap.add_argument('--alpha', type=float, default=10.0)
    ap.add_argument('--noise', type=float, default=0.0)
    ap.add_argument('--control_noise', type=float, default=0.0)
    ap.add_argument('--joint_bounds_margin', type=float, default=0.0)
    ap.add_argument('--collision_margin', type=float, default=0.0)
    ap.add_argument('--data_dir', default=None, help='directory of the *_mpc.pkl files (default: <root>/data_noise/)')
    a = ap.parse_args(argv)
    args = {**default_args(), 'horizon': a.horizon, 'alpha': a.alpha, 'noise': a.noise, 'control_noise': a.control_noise,
            'joint_bounds_margin': a.joint_bounds_margin, 'collision_margin': a.collision_margin}
    params = Parameters(args, 'z1', rti=True)
    params.alpha, params.N = a.alpha, a.horizon
    if a.data_dir:
        params.DATA_DIR = os.path.join(a.data_dir, '')
    prob = OcpProblem(params, 'naive', 'ext', N=2)
    solver = make_solver(prob) if make_solver else BatchedOcpSolver(prob, None)
    scores = {}
    for cont in a.controller or ['naive', 'zerovel', 'st', 'htwa', 'receding', 'constraint_everywhere']:
        use_net = None if cont in ('naive', 'zerovel') else True
        f = cl.result_file(params, 'z1', cont, params.N, use_net, a.noise, a.control_noise, a.joint_bounds_margin,
                           a.collision_margin)
        if not os.path.exists(f):
            # metrics_count_fails.py:75-79: a missing file counts as "zero fails", costs NaN
            print(f'Controller {cont}, zero fails or file not available')
            scores[cont] = {'score': 0, 'fails': 0, 'costs': [float('nan')] * 100, 'completed_idx': list(range(params.test_num))}
            continue
        res = pickle.load(open(f, 'rb'))
        scores[cont] = score_results(params, prob, solver, res)
        print(f"{cont:24s} fails {scores[cont]['fails']:4d}  mean cost {scores[cont]['mean_cost']:.4f}")
    out = scores_file(params, 'z1', params.N, params.alpha, a.noise, a.control_noise, a.joint_bounds_margin, a.collision_margin)
    cl.save_pickle(out, scores)
    print(out)
    return scores


if __name__ == '__main__':
    main()
