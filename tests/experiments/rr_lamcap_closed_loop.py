"""CPU experiment (needs the oracle patch): real_receding closed loop with the multiplier-cap exit (rr_elastic_replay.py).  Round 5:
  cap 0 (stall exit only): failed solves  843, collisions 11, viable  0, per-step max iterations 32
  cap 1e3 / 3e3 / 1e4:      failed solves 1701 / 1619 / 1389, collisions 14, viable 11 / 10 / 9, per-step max 23-25
A false alarm starts a cascade of its own, so the cap changes the policy's outcomes -- and the launch-bounding maximum hardly moves
(QPs that are infeasible in other rows still run to the stall exit).  Not adopted."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests/experiments'))
import numpy as np
_a = sys.argv[:]; sys.argv = sys.argv[:1]
import rr_infeasible as R
sys.argv = _a
from conftest import make_problem, sample_instances
from safe_mpc_amd import closed_loop as cl
B, STEPS, N = 256, 56, 30
for cap in [0.0] + [float(a) for a in sys.argv[1:]]:
    os.environ['SMPC_ORACLE_X_LAMCAP'] = str(cap)
    par, prob, net = make_problem('real_receding', N=N)
    par.back_hor = 30
    x0 = sample_instances(prob, B, seed=0)
    xg = np.repeat(x0[:, None, :], N + 1, axis=1); ug = np.zeros((B, N, prob.nu))
    R.LOG.clear()
    mk, mkb = R.factories(par)
    res = cl.run_mpc(par, 'real_receding', xg, ug, make_controller=mk, make_backup=mkb, n_steps=STEPS)
    its = np.array([l[1] for l in R.LOG]); sts = np.array([l[0] for l in R.LOG])
    print(f'cap {cap:g}: mean it {its.mean():.2f} mean of per-step max {its.max(1).mean():.1f}; failed solves {(sts != 0).sum()} (its mean {its[sts != 0].mean():.1f}); '
          f'conv {len(res["conv_idx"])} collisions {len(res["collisions_idx"])} viable {len(res["viable_idx"])} unconv {len(res["unconv_idx"])}', flush=True)
