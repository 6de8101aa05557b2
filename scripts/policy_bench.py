#!/usr/bin/env python3
"""ms per closed-loop step of the full policy layer with all state in HBM (run_mpc(on_device=True)): the controller's
accept / reject / abort automaton, the driver's backup-OCP + PD abort handling, plant and outcome tests -- at the bench's
workload size (B = 4096, Z1, N = 30), next to the plain-policy number bench.py reports.

    python scripts/policy_bench.py [controller ...]      (default: st htwa receding)
SMPC_WARM=1: start from generate_guess warm starts (full SQP with merit backtracking on the hard-terminal OCP, as the reference's
guess_acados.py writes them for every safe-set controller, utils.py:46-58, and scripts/mpc.py:79-84 loads them) instead of the
constant guess -- what the reference times; the share of infeasible QPs is then the policy's, not the cold start's."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                   # noqa: E402
from safe_mpc_amd import closed_loop as cl                     # noqa: E402
from safe_mpc_amd.solver import BatchedOcpSolver               # noqa: E402


def main():
    names = sys.argv[1:] or ['st', 'htwa', 'receding']
    par, prob, net = bench.build_problem()
    par.back_hor = 30
    B, N, steps = int(os.environ.get('SMPC_B', '4096')), prob.N, int(os.environ.get('SMPC_STEPS', '60'))
    s = BatchedOcpSolver(prob, net)
    x0 = bench.initial_states(s, prob, B, 0)
    xg = np.repeat(x0[:, None, :], N + 1, axis=1)
    ug = np.zeros((B, N, prob.nu))
    warm = os.environ.get('SMPC_WARM', '0') == '1'
    if warm:
        import copy
        pg = copy.copy(par)
        pg.nlp_max_iter = int(os.environ.get('SMPC_SQP_ITERS', '60'))
        t0 = time.perf_counter()
        guess, good = cl.generate_guess(pg, 'htwa', B)
        xg, ug = guess['xg'], guess['ug']
        B = len(xg)
        print(f'warm starts: {good.sum()} of {len(good)} Halton starts accepted by checkGuess after <= {pg.nlp_max_iter} SQP iterations '
              f'({time.perf_counter() - t0:.1f} s); running {B} instances', flush=True)
    for name in names:
        for dev in (True, False):
            if not dev and os.environ.get('SMPC_HOST', '0') != '1':
                continue
            tm = {}
            t0 = time.perf_counter()
            res = cl.run_mpc(par, name, xg, ug, n_steps=steps, on_device=dev, timing=tm,
                             groups=int(os.environ['SMPC_GROUPS']) if 'SMPC_GROUPS' in os.environ else None,
                             graphs=os.environ.get('SMPC_GRAPHS', '1') != '0')
            print(f"{name:12s} {'device' if dev else 'host  '} state{' (warm starts)' if warm else ''}: {tm['ms_per_step']:.3f} ms/step over {tm['steps']} steps "
                  f"(B={B}, N={N}, groups {tm.get('groups')}; total {time.perf_counter() - t0:.1f} s incl. set-up) | collisions {len(res['collisions_idx'])} "
                  f"viable {len(res['viable_idx'])} converged {len(res['conv_idx'])} unconverged {len(res['unconv_idx'])} abort events {len(res['x_viable'])}", flush=True)


if __name__ == '__main__':
    main()
