"""CPU experiment (oracle only): closed loop of the bench's workload with and without a warm start of the IPM's multipliers
from the previous step's solution (oracle hook SMPC_ORACLE_WARM, rules in qp_ipm).  Prints iteration statistics and how far
the closed-loop trajectories drift apart."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np

def run(cont, B, steps, N=30, cost='ext'):
    from conftest import make_problem, sample_instances
    from oracle.oracle import Oracle
    par, prob, net = make_problem(cont, cost, N=N)
    o = Oracle(prob, (net.weights, net.biases))
    x = sample_instances(prob, B, seed=0)
    xg = np.repeat(x[:, None, :], N + 1, axis=1); ug = np.zeros((B, N, 6)); p = np.zeros((B, N + 1, 5))
    p[:, :, :3], p[:, :, 3], p[:, :, 4] = prob.ee_ref, par.alpha, 1.0
    its, sts, xs = [], [], []
    for t in range(steps):
        xo, uo, st, it = o.solve_batch(x, xg, ug, p)
        its.append(it.copy()); sts.append(st.copy())
        xg, ug, ua = o.provide_control((st == 0).astype(np.int32), xo, uo, xg, ug)
        x, _ = o.plant_step(x, ua)
        xg = o.guess_correction(xg, ug)
        xs.append(x.copy())
    return np.array(its), np.array(sts), np.array(xs)

if __name__ == '__main__':
    cont, B, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    its, sts, xs = run(cont, B, steps, cost=sys.argv[4] if len(sys.argv) > 4 else 'ext')
    np.savez(os.environ.get('OUT', '/tmp/ws.npz'), its=its, sts=sts, xs=xs)
    print(cont, 'warm' if os.environ.get('SMPC_ORACLE_WARM') else 'cold', 'mean it %.2f' % its.mean(), 'sum of per-step max %d' % its.max(1).sum(),
          'fails', int((sts != 0).sum()), 'per-step mean', np.round(its.mean(1), 1)[::4])
