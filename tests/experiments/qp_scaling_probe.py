"""Experiment (round 3): how does k_qp_ipm's time per iteration depend on the batch size and on how much of the chip it has?
Same state as scripts/qp_bench.py; every instance held to SMPC_MAXIT iterations so that the launch has no tail."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from safe_mpc_amd.solver import BatchedOcpSolver

par, prob, net = bench.build_problem()
maxit = int(os.environ.get('SMPC_MAXIT', '0'))
if maxit:
    prob.desc.qp_max_iter = maxit
s = BatchedOcpSolver(prob, net)
Bfull = 4096
x0 = bench.initial_states(s, prob, Bfull, 0)
N = prob.N
xg = np.repeat(x0[:, None, :], N + 1, axis=1); ug = np.zeros((Bfull, N, 6)); p = np.zeros((Bfull, N + 1, 5))
p[:, :, :3], p[:, :, 3], p[:, :, 4] = prob.ee_ref, par.alpha, 1.0
x = x0
for i in range(5):
    xo, uo, st, it = s.solve(x, xg, ug, p)
    xg, ug, ua = s.provide_control((st == 0).astype(np.int32), xo, uo, xg, ug)
    x, _ = s.plant_step(x, ua)
    xg = s.guess_correction(xg, ug)
dev = torch.device('cuda:0')
t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
s.enable_timing(True)
for B in [int(v) for v in os.environ.get('SMPC_BS', '512,1024,2048,3072,4096').split(',')]:
    xd, xgd, ugd, pd = t(x[:B]), t(xg[:B]), t(ug[:B]), t(p[:B])
    res = []
    for i in range(6):
        out = s.solve(xd, xgd, ugd, pd)
        tm = s.timing()
        res.append([tm['time_qp_setup'], tm['time_qp_ipm'], tm['qp_wave_busy_mean'], tm['qp_wave_span']])
    res = np.array(res[1:]) * 1e3
    it = out[3].cpu().numpy()
    print('hog %s maxit %d B %5d iters mean %.2f max %2d | setup %.3f ipm min %.3f mean %.3f ms | busy mean %.3f span %.3f | ms per iteration (busy) %.4f' % (
        os.environ.get('SMPC_EXP_HOG', '0'), maxit, B, it.mean(), it.max(), res[:, 0].min(), res[:, 1].min(), res[:, 1].mean(),
        res[:, 2].mean(), res[:, 3].mean(), res[:, 2].mean() / it.mean()), flush=True)
