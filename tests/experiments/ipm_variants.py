"""CPU experiment (oracle only, not product): variants of the IPM iteration that move fewer bytes per solve.
Closed loop of the oracle; per variant: mean IPM iterations, iterations whose corrector (sweeps B2 + F2) was skipped, the cost of a
solve in units of one full iteration's workspace traffic (B1 + F1 = 0.64, B2 + F2 = 0.36 of the 1857 doubles per stage and
iteration), the mean of the per-step maxima (what sets a launch's duration), failed solves and the deviation of
the closed-loop trajectory from the baseline's.
Needs the experiment knobs in the oracle:  git apply tests/experiments/ipm_variants_oracle_patch.diff && make -C oracle
(the patch was cut against oracle/smpc_oracle.cpp of commit 671f866; the stall exit's total count, commit 1b9877b, came after it: check that
file out of 671f866 first, or merge hunk 8 by hand -- round 6's tests/experiments/gondzio_oracle_patch.diff applies to the current oracle)
(and `git checkout oracle/smpc_oracle.cpp && make -C oracle` afterwards: the committed oracle restates the engine's algorithm only).
usage: python tests/experiments/ipm_variants.py [problem: st | constraint_everywhere | fr7 | ...] [B] [steps]

Round 5 (128 instances x 100 closed-loop steps, 'st'; cost in units of one full iteration, an iteration without corrector = 0.64):
  variant                | mean it | skipped | cost/solve | vs base | mean step-max it | fails | max |x - x_base|
  baseline               |   7.390 |   0.000 |      7.390 |   1.000 |            11.64 |     0 | 0
  affine exit            |   7.392 |   0.672 |      7.150 |   0.967 |            11.67 |     0 | 1.5e-03
  skip a>=.95 r<=.1      |   7.772 |   1.852 |      7.105 |   0.961 |            12.01 |     0 | 9.8e-03
  skip a>=.9 r<=.15      |   8.194 |   3.200 |      7.042 |   0.953 |            12.61 |     2 | 1.8e-01
  skip a>=.8 r<=.25      |   9.000 |   5.417 |      7.050 |   0.954 |            13.51 |    13 | 1.8e-01
  no 2nd-order term      |  10.661 |   0.000 |     10.661 |   1.442 |            16.70 |     0 | 2.9e-03      (the two-right-hand-side iteration: 0.79 per iteration -> 1.14)
  ftb .999 / .9995       |   7.238 |                7.238 |   0.979
  sigma cap .1 / pow 2 / pow 4: 1.022 / 1.034 / 0.997;  second-order term scaled by a_aff^2 / a_aff always: 1.131 / 1.041
  512 instances: fraction to the boundary 0.9 / 0.95 / 0.98 / 0.99: mean x 1.18 / 1.10 / 1.04 / 1.02, per-step maximum 12.9 / 12.3 / 12.1 / 12.4
  (baseline 12.15); centring with sigma >= 0.6 / 1 after a step shorter than 0.5 / 0.3: mean x 1.00-1.01, per-step maximum 11.9-12.2.
  constraint_everywhere (128 x 100): baseline 6.66 it. / per-step max 11.35; affine exit cost 0.965; skip a>=.95 0.933 (max 11.57); skip a>=.9
  0.919 (max 12.57); no 2nd-order term 1.40; ftb .999 0.964.   7-DoF N=40 (64 x 60): baseline 6.70 / 10.27; 0.968; 0.949 (10.50); 0.930 (11.05);
  1.44; 0.975.
-> a conditional corrector saves at most 4.7 % and costs robustness; without the second-order term +44 % iterations; the per-step
   maximum (what a launch lasts: spikes from collision rows that come active one stage after the other) does not respond to the
   step-length or centring rules."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from conftest import make_problem, make_problem_fr7, sample_instances, constant_guess
from oracle.oracle import Oracle

PROB = sys.argv[1] if len(sys.argv) > 1 else 'st'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 100
C_HALF = float(os.environ.get('C_HALF', 0.64))     # cost of an iteration without corrector


def setup():
    if PROB == 'fr7':
        par, prob, net = make_problem_fr7()
    else:
        par, prob, net = make_problem(PROB, 'ext', N=30)
    return par, prob, net


def run(env):
    for k in list(os.environ):
        if k.startswith('SMPC_ORACLE_X_'):
            del os.environ[k]
    os.environ.update({'SMPC_ORACLE_X_' + k: str(v) for k, v in env.items()})
    par, prob, net = setup()
    o = Oracle(prob, (net.weights, net.biases))
    x = sample_instances(prob, B, seed=0)
    xg, ug, p = constant_guess(prob, x)
    fails = np.zeros(B, int)
    X, IT, SK, nfail = [x.copy()], [], [], 0
    for j in range(STEPS):
        xg = o.guess_correction(xg, ug)
        xt, ut, st, it = o.solve_batch(x, xg, ug, p)
        IT.append(it % 1000)
        SK.append(it // 1000)
        nfail += int((st != 0).sum())
        fails = np.where(st == 0, 0, fails + 1)
        xg, ug, u = o.provide_control((fails == 0).astype(np.int32), xt, ut, xg, ug)
        x, _ = o.plant_step(x, u)
        X.append(x.copy())
    return np.array(X), np.array(IT), np.array(SK), nfail


VARIANTS = [
    ('baseline', {}),
    ('affine exit', {'AFFEXIT': 1}),
    ('skip a>=.95 r<=.1', {'AFFEXIT': 1, 'SKIP_A': 0.95, 'SKIP_R': 0.1}),
    ('skip a>=.9 r<=.15', {'AFFEXIT': 1, 'SKIP_A': 0.9, 'SKIP_R': 0.15}),
    ('skip a>=.8 r<=.25', {'AFFEXIT': 1, 'SKIP_A': 0.8, 'SKIP_R': 0.25}),
    ('no 2nd-order term', {'NO2ND': 1}),
    ('ftb .999', {'FTB': 0.999}),
    ('sigma cap .1', {'SIGCAP': 0.1}),
    ('sigma pow 2', {'SIGPOW': 2}),
    ('sigma pow 4', {'SIGPOW': 4}),
    ('cw a^2', {'CWMODE': 1}),
    ('cw a', {'CWMODE': 2}),
    ('ftb .9995', {'FTB': 0.9995}),
    ('ftb gate .9', {'FTBGATE': 0.9}),
    ('ftb gate .5', {'FTBGATE': 0.5}),
    ('ftb gate .9 ftb .999', {'FTBGATE': 0.9, 'FTB': 0.999}),
    ('ftb hi .999999', {'FTBHI': 0.999999}),
    ('recentre .6 after a<.5', {'RECSIG': 0.6}),
    ('recentre 1 after a<.5', {'RECSIG': 1.0}),
    ('recentre .6 after a<.3', {'RECSIG': 0.6, 'RECALPHA': 0.3}),
    ('recentre 1 after a<.3', {'RECSIG': 1.0, 'RECALPHA': 0.3}),
    ('ftb .9', {'FTB': 0.9}),
    ('ftb .95', {'FTB': 0.95}),
    ('ftb .98', {'FTB': 0.98}),
    ('ftb .99', {'FTB': 0.99}),
]
if len(sys.argv) > 4:
    VARIANTS = [VARIANTS[0]] + [v for v in VARIANTS if any(a in v[0] for a in sys.argv[4:])]

base = None
print(f'problem {PROB}, B = {B}, {STEPS} closed-loop steps; cost of an iteration without corrector = {C_HALF}')
print(f'{"variant":22s} | mean it | skipped | cost/solve | vs base | mean step-max it | step-max cost | fails | max |x - x_base|')
for name, env in VARIANTS:
    X, IT, SK, nf = run(env)
    cost = (IT - SK) + C_HALF * SK
    if base is None:
        base, c0 = X, cost.mean()
    print(f'{name:22s} | {IT.mean():7.3f} | {SK.mean():7.3f} | {cost.mean():10.3f} | {cost.mean() / c0:7.3f} | {IT.max(1).mean():16.2f} | '
          f'{cost.max(1).mean():13.2f} | {nf:5d} | {np.abs(X - base).max():.2e}', flush=True)
