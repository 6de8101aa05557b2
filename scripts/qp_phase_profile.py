"""Per-phase shader-clock breakdown of k_qp_ipm (diagnostic build: make -C safe_mpc_amd/csrc libsmpc_hip_prof.so).

Usage on the GPU box:  SMPC_B=4096 python scripts/qp_phase_profile.py
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
if os.environ.get('SMPC_MAXIT'):
    prob.desc.qp_max_iter = int(os.environ['SMPC_MAXIT'])
s = BatchedOcpSolver(prob, net)
s.set_qp_mode('throughput')      # (k_qp_ipm's phases; the latency form, which the engine would pick below 512 instances, has scripts/qp_wg_phase_profile.py)
B = int(os.environ.get('SMPC_B', '4096'))
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
L.smpc_debug_qp_profile(buf)   # reset
s.enable_timing(True)
out = s.solve(xd, xgd, ugd, pd)
qp_ms = s.timing()['time_qp'] * 1e3
L.smpc_debug_qp_profile(buf)
v = np.array(list(buf), dtype=np.float64)
names = ['B1 commit+rows', 'B1 P b, D-scaling', 'B1 assembly+grad', 'B1 chol+columns', 'B1 P update', 'B1 end fence',
         'F1 control', 'F1 rows', 'B2 rows', 'B2 recursion', 'F2 control', 'F2 rows', 'epilogue', 'prologue']
waves, its = v[14], v[15]
tot = v[:14].sum()
print('B %d  QP (setup + ipm) %.3f ms  half-waves %d  mean iterations %.2f' % (B, qp_ms, waves, its / max(waves, 1)))
print('mean clocks per half-wave %.0f  (kernel time = %.0f clocks at 2.4 GHz)' % (tot / waves, qp_ms * 2.4e6))
for n, c in zip(names, v[:14]):
    print('  %-16s %5.1f %%   %9.0f clocks/wave' % (n, 100 * c / tot, c / waves))
