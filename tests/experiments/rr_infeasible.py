"""CPU experiment (oracle only, not product): IPM iteration counts of the real_receding policy's tube QPs in closed loop, and a
dump of the solves that run long (status, iterations, final mu / residual) -- the cases an infeasibility exit has to catch.
usage: python tests/experiments/rr_infeasible.py [B] [steps] [controller]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from conftest import make_problem, sample_instances
from fake_solver import OracleSolver
from safe_mpc_amd import closed_loop as cl
from safe_mpc_amd import controller as C

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 30
CONT = sys.argv[3] if len(sys.argv) > 3 else 'real_receding'
N = 30
LOG = []


class LoggingSolver(OracleSolver):
    def solve(self, x0, xg, ug, p, out=None):
        x, u, st, it = self.o.solve_batch(x0, xg, ug, p, with_res=True)[:4] if False else self.o.solve_batch(x0, xg, ug, p)
        LOG.append((np.asarray(st).copy(), np.asarray(it).copy()))
        if os.environ.get('RR_DUMP') and np.asarray(it).max() >= int(os.environ['RR_DUMP']):
            b = int(np.argmax(it))
            np.savez('/tmp/rr_case.npz', x0=x0[b:b + 1], xg=xg[b:b + 1], ug=ug[b:b + 1], p=p[b:b + 1],
                     lo=self._lo[b:b + 1] if self._lo is not None else np.zeros(0), hi=self._hi[b:b + 1] if self._hi is not None else np.zeros(0))
            print('dumped instance', b, 'it', it[b], 'st', st[b])
            os.environ.pop('RR_DUMP')
        return x, u, st, it

    _lo = _hi = None

    def set_instance_bounds(self, lo=None, hi=None):
        self._lo, self._hi = (None if lo is None else np.array(lo)), (None if hi is None else np.array(hi))
        super().set_instance_bounds(lo, hi)


def factories(par):
    def make_controller(name, batch):
        cls = C.CONTROLLERS[name]
        ctrl = cls.__new__(cls)
        prob = C.OcpProblem(par, cls.cont_name, 'ext', N=N)
        net = C.SafeSetNet.from_params(par, prob.x_min, prob.x_max)
        prob.set_normalisation(net.mean, net.std)
        C.AbstractController.__init__(ctrl, par, batch, 'ext', N, solver=LoggingSolver(prob, net), net=net)
        return ctrl

    def make_backup(batch):
        ctrl = C.SafeBackupController.__new__(C.SafeBackupController)
        prob = C.OcpProblem(par, 'backup', 'zero', N=par.back_hor)
        net = C.SafeSetNet.from_params(par, prob.x_min, prob.x_max)
        C.AbstractController.__init__(ctrl, par, batch, 'zero', par.back_hor, solver=OracleSolver(prob, net), net=net)
        return ctrl
    return make_controller, make_backup


if __name__ == '__main__':
    par, prob, net = make_problem(CONT, N=N)
    par.back_hor = 30
    x0 = sample_instances(prob, B, seed=0)
    xg = np.repeat(x0[:, None, :], N + 1, axis=1)
    ug = np.zeros((B, N, prob.nu))
    mk, mkb = factories(par)
    res = cl.run_mpc(par, CONT, xg, ug, make_controller=mk, make_backup=mkb, n_steps=STEPS)
    its = np.array([l[1] for l in LOG]); sts = np.array([l[0] for l in LOG])
    print('solves', its.shape, 'mean it', its.mean(), 'max', its.max())
    print('histogram of iterations:', np.bincount(np.minimum(its.ravel(), 60) // 5) , '(bins of 5, last = 60+)')
    for s in np.unique(sts):
        m = sts == s
        print(f'status {s}: {m.sum()} solves, iterations mean {its[m].mean():.1f} max {its[m].max()} min {its[m].min()}')
    print('collisions', len(res['collisions_idx']), 'viable', len(res['viable_idx']))
