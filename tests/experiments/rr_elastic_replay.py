"""CPU experiment (needs tests/experiments/ipm_variants_oracle_patch.diff applied): the recorded real_receding QPs (rr_record.py)
re-solved (a) with the tube rows ELASTIC (one shared slack per two-sided row, exact L1 penalty w, eliminated in closed form) and
(b) with an exit when a tube row's multiplier exceeds a cap.  Round 5: infeasible tube QPs, hard: 73 iterations on average without
any exit (27 with the stall exit); elastic w = 1e2 / 1e3 / 1e4 / 1e5: 15 / 19 / 23 / 27 iterations, and 134 / 27 / 21 / 7 of 4584
FEASIBLE QPs reported infeasible (the penalty is exact only above the hard problem's multipliers); multiplier cap 1e3 / 1e4 / 1e5:
9 / 16 / 26 iterations, 27 / 21 / 7 false alarms.  The multipliers of an infeasible tube climb by ~x3 per iteration, so any test that
is exact enough to spare the feasible QPs needs as many iterations as the stall exit.  usage: rr_elastic_replay.py <w or cap> ..."""
import os, sys, pickle
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from conftest import make_problem
from oracle.oracle import Oracle
REC = pickle.load(open(os.environ.get('RR_REC', '/tmp/rr_rec.pkl'), 'rb'))
N=30
W = [float(a) for a in sys.argv[1:]] or [1e4]
par, prob, net = make_problem('real_receding', N=N)
prob.desc.qp_stall_iters = 0
o = Oracle(prob, (net.weights, net.biases))
p = np.zeros((256, N + 1, 5)); p[:, :, :3] = prob.ee_ref; p[:, :, 3] = par.alpha; p[:, :, 4] = 1.0
for w in [0.0] + W:
    os.environ["SMPC_ORACLE_X_LAMCAP"] = str(w)
    its_f, its_o, st_all, du = [], [], [], []
    for s_ in range(0, len(REC), 3):
        st, it, r, x0, xg, ug, lo, hi = REC[s_][:8]
        if lo is None: continue
        o.set_instance_bounds(lo, hi)
        xs, us, ss, is_ = o.solve_batch(x0, xg, ug, p)
        if w == 0.0:
            REC[s_] = REC[s_] + (ss.copy(), us.copy())
        else:
            sh, uh = REC[s_][8], REC[s_][9]
            bad = sh != 0
            its_f += is_[bad].tolist(); its_o += is_[~bad].tolist()
            st_all.append((int(((ss != 0) & bad).sum()), int(bad.sum()), int(((ss != 0) & ~bad).sum()), int((~bad).sum())))
            ok = (~bad) & (ss == 0)
            if ok.any(): du.append((np.abs(us[ok] - uh[ok]).max(axis=(1,2)) / (1 + np.abs(uh[ok]).max(axis=(1,2)))).max())
        if w == 0.0:
            bad = ss != 0
            its_f += is_[bad].tolist(); its_o += is_[~bad].tolist()
    f = lambda a: ('mean %.1f max %d' % (np.mean(a), np.max(a))) if len(a) else None
    if w == 0.0:
        print(f'hard (no stall exit): infeasible QPs ({len(its_f)}): its {f(its_f)}; feasible ({len(its_o)}): its {f(its_o)}')
    else:
        sa = np.array(st_all).sum(0)
        print(f'lamcap {w:g}: hard-infeasible QPs: reported failed {sa[0]}/{sa[1]}, its {f(its_f)}; hard-feasible: reported failed {sa[2]}/{sa[3]}, its {f(its_o)}; |u - u_hard| rel max {max(du):.2e}')
