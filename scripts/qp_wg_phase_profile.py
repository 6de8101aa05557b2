"""Per-phase shader-clock breakdown of k_qp_ipm_wg, the latency form of the QP solve (diagnostic build:
make -C safe_mpc_amd/csrc libsmpc_hip_prof.so).  Clocks of thread 0 of every workgroup, summed over workgroups.

Usage on the GPU box:  SMPC_B=64 python scripts/qp_wg_phase_profile.py
"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['SMPC_HIP_LIB'] = os.environ.get('SMPC_PROF_LIB', os.path.join(ROOT, 'safe_mpc_amd', 'csrc', 'libsmpc_hip_prof.so'))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from safe_mpc_amd import _lib
from safe_mpc_amd.solver import BatchedOcpSolver

par, prob, net = bench.build_problem()
s = BatchedOcpSolver(prob, net)
s.set_qp_mode('latency')
B = int(os.environ.get('SMPC_B', '64'))
x0 = bench.initial_states(s, prob, B, 0)
N = prob.N
xg = np.repeat(x0[:, None, :], N + 1, axis=1); ug = np.zeros((B, N, 6)); p = np.zeros((B, N + 1, 5))
p[:, :, :3], p[:, :, 3], p[:, :, 4] = prob.ee_ref, par.alpha, 1.0
x = x0
for i in range(5):
    xo, uo, st, it = s.solve(x, xg, ug, p)
    xg, ug, ua = s.provide_control((st == 0).astype(np.int32), xo, uo, xg, ug)
    x, _ = s.plant_step(x, ua)
    xg = s.guess_correction(xg, ug)
dev = torch.device('cuda:0')
t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
xd, xgd, ugd, pd = t(x), t(xg), t(ug), t(p)
L = _lib.lib()
buf = (C.c_ulonglong * 16)()
L.smpc_debug_qp_wg_profile(buf)   # reset
s.enable_timing(True)
out = s.solve(xd, xgd, ugd, pd)
qp_ms = s.timing()['time_qp'] * 1e3
L.smpc_debug_qp_wg_profile(buf)
v = np.array(list(buf), dtype=np.float64)
names = {0: 'A rows + H blocks (parallel)', 1: 'B Riccati recursion', 2: 'C predictor roll-out (+ D beside it)', 3: 'D rows left over (parallel)',
         4: 'E corrector costate', 7: '  B: commit H, P b', 8: '  B: Lambda, G, rho', 9: '  B: Cholesky, columns', 10: '  B: P update, p', 5: 'F corrector roll-out (+ G beside it)', 6: 'G rows left over (parallel)', 12: 'epilogue', 13: 'prologue'}
blocks, its = v[14], v[15]
tot = v[:14].sum()
print('B %d  QP (setup + ipm) %.3f ms  workgroups %d  mean iterations %.2f' % (B, qp_ms, blocks, its / max(blocks, 1)))
print('mean clocks per workgroup %.0f  (kernel time = %.0f clocks at 2.4 GHz); per iteration %.0f' % (tot / blocks, qp_ms * 2.4e6, tot / max(its, 1)))
for i, n in names.items():
    print('  %-30s %5.1f %%   %9.0f clocks/iteration  %7.0f per stage' % (n, 100 * v[i] / tot, v[i] / max(its, 1), v[i] / max(its, 1) / (N + 1)))
