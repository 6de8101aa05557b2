"""CPU experiment (oracle only): records every QP of a real_receding closed loop (256 instances x 56 steps) with its tube node,
status and iteration count into $RR_REC (default /tmp/rr_rec.pkl), and prints where the failures are: a first failure at r = N - 1 (0.3 % of those
solves) starts a cascade -- the tube asks for the SAME absolute state one step earlier than the plan reaches it, so every following
solve of that instance fails too until r = 0 aborts (controller.py:530-553): 843 of 14 336 solves fail, 803 of them in cascades."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests/experiments'))
import numpy as np
import rr_infeasible as R
from conftest import make_problem, sample_instances
from safe_mpc_amd import closed_loop as cl
B, STEPS, N = 256, 56, 30
par, prob, net = make_problem('real_receding', N=N)
par.back_hor = 30
x0 = sample_instances(prob, B, seed=0)
xg = np.repeat(x0[:, None, :], N + 1, axis=1); ug = np.zeros((B, N, prob.nu))
REC = []
class S2(R.LoggingSolver):
    def solve(self, x0, xg, ug, p, out=None):
        x, u, st, it = self.o.solve_batch(x0, xg, ug, p)
        # find tube node per instance: where hi-lo == 2e-3
        if self._lo is not None:
            w = (self._hi - self._lo)[:, :, 0]
            r = np.where((np.abs(w - 2e-3) < 1e-9).any(1), np.argmax(np.abs(w - 2e-3) < 1e-9, 1), -1)
        else:
            r = np.full(len(st), -1)
        REC.append((np.asarray(st).copy(), np.asarray(it).copy(), r.copy(), x0.copy(), xg.copy(), ug.copy(), self._lo.copy() if self._lo is not None else None, self._hi.copy() if self._hi is not None else None))
        return x, u, st, it
R.LoggingSolver = S2
mk, mkb = R.factories(par)
res = cl.run_mpc(par, 'real_receding', xg, ug, make_controller=mk, make_backup=mkb, n_steps=STEPS)
st = np.array([r[0] for r in REC]); it = np.array([r[1] for r in REC]); rr = np.array([r[2] for r in REC])
print('solves', st.shape, 'mean it', it.mean(), 'fails', (st!=0).sum())
bad = st != 0
print('r of failed solves:', np.bincount(rr[bad] + 1)[:32], '(index = r+1)')
print('r of all solves   :', np.bincount(rr.ravel() + 1)[:32])
print('iterations of failed:', np.sort(it[bad])[-20:], 'mean', it[bad].mean())
print('iterations of ok: mean', it[~bad].mean(), 'max', it[~bad].max())
# for failed solves: closed-form r=... test
dt = par.dt
for (s_, j_) in zip(*np.where(bad)):
    pass
import pickle; pickle.dump(REC, open(os.environ.get('RR_REC', '/tmp/rr_rec.pkl'), 'wb'))
