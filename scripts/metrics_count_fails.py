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
    """x [n, T+1, nx], u [n, T, nu] (NaN after a failure) -> cost per instance (NaN rows excluded by the caller)."""
    n, T = u.shape[0], u.shape[1]
    nodes = np.nan_to_num(x[:, :T]).reshape(-1, 1, x.shape[2])
    dummy_u = np.zeros((nodes.shape[0], solver.N, prob.nu))
    p = np.zeros((nodes.shape[0], solver.N + 1, 5))
    ee = solver.eval_nodes(np.repeat(nodes, solver.N + 1, 1), dummy_u, p)['ee'][:, 0, :].reshape(n, T, 3)
    err = np.sum((ee - prob.ee_ref) ** 2, axis=2)
    return params.Q_weight * err.sum(1) + params.R_weight * np.sum(np.nan_to_num(u) ** 2, axis=(1, 2))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('-c', '--controller', action='append', default=None)
    ap.add_argument('--horizon', type=int, default=45)
    ap.add_argument('--alpha', type=float, default=10.0)
    ap.add_argument('--noise', type=float, default=0.0)
    ap.add_argument('--control_noise', type=float, default=0.0)
    a = ap.parse_args(argv)
    args = {**default_args(), 'horizon': a.horizon, 'alpha': a.alpha, 'noise': a.noise, 'control_noise': a.control_noise}
    params = Parameters(args, 'z1', rti=True)
    params.alpha, params.N = a.alpha, a.horizon
    scores = {}
    for cont in a.controller or ['naive', 'zerovel', 'st', 'htwa', 'receding']:
        use_net = None if cont in ('naive', 'zerovel') else True
        f = cl.result_file(params, 'z1', cont, params.N, use_net, a.noise, a.control_noise, 0.0, 0.0)
        if not os.path.exists(f):
            print('missing', f)
            continue
        res = pickle.load(open(f, 'rb'))
        prob = OcpProblem(params, 'naive', 'ext', N=2)
        solver = BatchedOcpSolver(prob, None)
        done = sorted(set(range(res['x'].shape[0])) - set(res['collisions_idx']))
        costs = closed_loop_costs(params, prob, solver, res['x'][done], res['u'][done]) if done else np.zeros(0)
        scores[cont] = {'score': float(np.mean(costs)) if done else float('nan'), 'fails': len(res['collisions_idx']),
                        'costs': costs, 'completed_idx': done}
        print(f"{cont:24s} fails {scores[cont]['fails']:4d}  mean cost {scores[cont]['score']:.4f}")
    out = f'{params.DATA_DIR}z1_{params.N}hor_{int(params.alpha)}sm_noise_{a.noise}_scores.pkl'
    cl.save_pickle(out, scores)
    print(out)


if __name__ == '__main__':
    main()
