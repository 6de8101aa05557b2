"""C4-like timing (7-DoF, N = 40, safe-set row on every node, B = 4096, first step from the constant guess); starts sampled
with the test helpers (tests/conftest.py), hence under tests/."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from conftest import *
from safe_mpc_amd.solver import BatchedOcpSolver
par, prob, net = make_problem_fr7(N=40)
s = BatchedOcpSolver(prob, net)
B = 4096
x0 = sample_instances(prob, B, seed=3, vel_scale=0.1)
xg, ug, p = constant_guess(prob, x0, ee_ref=prob.ee_ref)
dev = torch.device('cuda:0'); t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
xd, xgd, ugd, pd = t(x0), t(xg), t(ug), t(p)
s.enable_timing(True)
res = []
for i in range(6):
    out = s.solve(xd, xgd, ugd, pd); tm = s.timing(); res.append([tm['time_lin'], tm['time_nn'], tm['time_qp_setup'], tm['time_qp_ipm']])
res = np.array(res) * 1e3
it = out[3].cpu().numpy(); st = out[2].cpu().numpy()
print('fr7 C4-like B %d N 40: iters mean %.2f max %d fails %d | ms lin %.3f nn %.3f setup %.3f ipm %.3f' % (B, it.mean(), it.max(), (st != 0).sum(), *res.min(axis=0)))
