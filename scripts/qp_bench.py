"""Controlled micro-benchmark of the solve kernels: identical inputs (the state after a few closed-loop steps), repeated."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from safe_mpc_amd.solver import BatchedOcpSolver

# SMPC_QPB_PROBLEM=fr7: BASELINE config 4's problem (7-DoF, N = 40, row on every node) at one sub-batch launch of bench.py --config c4
FR7 = os.environ.get('SMPC_QPB_PROBLEM') == 'fr7'
if FR7:
    from safe_mpc_amd.parser import Parameters
    from safe_mpc_amd.problem import OcpProblem
    from safe_mpc_amd.safe_set import SafeSetNet
    par = Parameters({}, 'fr7', filename=os.path.join(ROOT, 'config_fr7.yaml'))
    par.N = 40
    prob = OcpProblem(par, 'constraint_everywhere', 'ext', N=40)
    net = SafeSetNet.from_params(par, prob.x_min, prob.x_max)
    prob.set_normalisation(net.mean, net.std)
else:
    par, prob, net = bench.build_problem()
if os.environ.get('SMPC_MAXIT'):
    prob.desc.qp_max_iter = int(os.environ['SMPC_MAXIT'])
s = BatchedOcpSolver(prob, net)
B = int(os.environ.get('SMPC_B', '5461' if FR7 else '4096'))
x0 = bench.initial_states(s, prob, B, 0)
if FR7:
    x0[:, prob.nq:] = 0.1 * np.random.default_rng(0).uniform(-1, 1, (B, prob.nq)) * prob.ubx[prob.nq:]      # (as bench.py --config c4)
N = prob.N
xg = np.repeat(x0[:, None, :], N + 1, axis=1); ug = np.zeros((B, N, prob.nq)); p = np.zeros((B, N + 1, 5))
p[:, :, :3], p[:, :, 3], p[:, :, 4] = prob.ee_ref, par.alpha, 1.0
x = x0
for i in range(int(os.environ.get('SMPC_WARM', '5'))):
    xo, uo, st, it = s.solve(x, xg, ug, p)
    xg, ug, ua = s.provide_control((st == 0).astype(np.int32), xo, uo, xg, ug)
    x, _ = s.plant_step(x, ua)
    xg = s.guess_correction(xg, ug)
dev = torch.device('cuda:0')
t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
xd, xgd, ugd, pd = t(x), t(xg), t(ug), t(p)
s.enable_timing(True)
res = []
for i in range(int(os.environ.get('SMPC_REP', '8'))):
    out = s.solve(xd, xgd, ugd, pd)
    tm = s.timing()
    res.append([tm['time_lin'], tm['time_nn'], tm['time_qp']])
res = np.array(res) * 1e3
it = out[3].cpu().numpy()
print('B', B, 'iters mean %.2f max %d' % (it.mean(), it.max()), 'lin ms min %.3f' % res[:, 0].min(), 'nn %.3f' % res[:, 1].min(),
      'QP ms min %.3f mean %.3f' % (res[:, 2].min(), res[:, 2].mean()))
