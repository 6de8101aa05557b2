#!/usr/bin/env python3
"""Closed-loop MPC run -- drop-in for the reference's scripts/mpc.py (same flags, config.yaml, guess / result pickles),
with every instance of its outer loop (mpc.py:102) solved at once on the MI355X engine.

    python scripts/mpc.py -c st --horizon 30 --alpha 10 [--noise 5 --control_noise 1]
Exit code = number of failed instances, as in the reference (mpc.py:317).
"""
import os
import pickle
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_mpc_amd import closed_loop as cl                      # noqa: E402
from safe_mpc_amd.parser import Parameters, parse_args          # noqa: E402


def main(argv=None):
    args = parse_args(argv)
    model_name = args['system']
    params = Parameters(args, model_name, rti=True)
    params.act, params.alpha, params.N = args['activation'], args['alpha'], args['horizon']
    params.back_hor = args['back_hor']
    cont_name = args['controller']
    use_net = None if cont_name in ('naive', 'zerovel') else True          # controller.py:236 / STController
    gfile = cl.guess_file(params, model_name, cont_name, params.N, use_net)
    print(gfile)
    data = pickle.load(open(gfile, 'rb'))
    x_guess, u_guess = data['xg'][:params.test_num], data['ug'][:params.test_num]
    tm = {}
    # all loop state in HBM (policy automaton, abort handling, logs); SMPC_HOST_STATE=1 keeps it in numpy arrays instead
    res = cl.run_mpc(params, cont_name, x_guess, u_guess, noise=args['noise'], control_noise=args['control_noise'],
                     callback=True, on_device=os.environ.get('SMPC_HOST_STATE', '0') != '1', timing=tm,
                     collect_times=os.environ.get('SMPC_NO_TIME_STATS', '0') != '1')
    with_stats = os.environ.get('SMPC_NO_TIME_STATS', '0') != '1'
    print(f"{tm['ms_per_step']:.3f} ms per closed-loop step of {x_guess.shape[0]} instances"
          + (' (eager launches with per-solve HIP events for the time statistics below; SMPC_NO_TIME_STATS=1 replays the step halves as '
             'hipGraphs without them)' if with_stats else ' (step halves replayed as hipGraphs, no per-solve time statistics)'))
    if 'time_stats' in res:      # the block of the reference's mpc.py:300-303 (here one solve = all instances of a group)
        print('99% quantile of the computation time:')
        for field, t in zip(res['time_fields'], res['time_q99']):
            print(f"{field:<20} -> {t}")
    n = x_guess.shape[0]
    print(f"Completed task: {len(res['conv_idx'])}\nCollisions: {len(res['collisions_idx'])}"
          f"\nViable states: {len(res['viable_idx'])}\nNot converged: {n - len(res['conv_idx']) - len(res['collisions_idx'])}")
    cl.save_pickle(cl.result_file(params, model_name, cont_name, params.N, use_net, args['noise'], args['control_noise'],
                                  args['joint_bounds_margin'], args['collision_margin']), res)
    return len(res['collisions_idx'])


if __name__ == '__main__':
    sys.exit(main())
