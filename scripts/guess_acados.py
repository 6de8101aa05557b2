#!/usr/bin/env python3
"""Warm-start generation -- drop-in for the reference's scripts/guess_acados.py: writes the
{'xg': [n,N+1,nx], 'ug': [n,N,nu]} pickle that scripts/mpc.py loads (mpc.py:79-84), all instances solved at once.

    python scripts/guess_acados.py -c st --horizon 30 --alpha 10
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_mpc_amd import closed_loop as cl                      # noqa: E402
from safe_mpc_amd.parser import Parameters, parse_args          # noqa: E402


def main(argv=None):
    args = parse_args(argv)
    model_name = args['system']
    params = Parameters(args, model_name, rti=False)            # full SQP (parser.py:115-117)
    params.act, params.alpha, params.N = args['activation'], args['alpha'], args['horizon']
    cont_name = args['controller']
    # every safe-set controller name is generated with the hard-terminal OCP (utils.py:46-58)
    gen_name = cont_name if cont_name in ('naive', 'zerovel') else 'htwa'
    t0 = time.time()
    guess, good = cl.generate_guess(params, gen_name, params.test_num, verbose=True)
    print(f'{good.sum()}/{len(good)} guesses accepted in {time.time() - t0:.1f} s')
    use_net = None if cont_name in ('naive', 'zerovel') else True
    out = cl.guess_file(params, model_name, cont_name, params.N, use_net)
    cl.save_pickle(out, guess)
    print(out)
    return 0


if __name__ == '__main__':
    sys.exit(main())
