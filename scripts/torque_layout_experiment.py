"""Layout experiment for the thread-per-node linearisation kernel (VERDICT r1, weak point 9): per-node records (AoS, what
the engine ships) against field-major across the batch axis (SoA, the north star's "coalesced HBM across the batch axis")."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from safe_mpc_amd.solver import BatchedOcpSolver
par, prob, net = bench.build_problem()
s = BatchedOcpSolver(prob, net)
B, N = 4096, prob.N
x0 = bench.initial_states(s, prob, B, 0)
xg = np.repeat(x0[:, None, :], N + 1, axis=1) + 0.01 * np.random.default_rng(0).standard_normal((B, N + 1, 12))
ug = np.random.default_rng(1).uniform(-3, 3, (B, N, 6))
dev = torch.device('cuda:0')
xd, ud = torch.tensor(xg, device=dev), torch.tensor(ug, device=dev)
torch.cuda.synchronize()
s.L.smpc_debug_torque_layout.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]
for mode, name in ((0, 'node records (AoS, shipped)'), (1, 'field-major across the batch (SoA)'), (0, 'AoS again'), (1, 'SoA again')):
    ms = C.c_float()
    rc = s.L.smpc_debug_torque_layout(s.h, B, xd.data_ptr(), ud.data_ptr(), mode, 20, C.byref(ms))
    print(f'k_node_torque, B={B} N={N}: {name:38s} {ms.value:.4f} ms (rc {rc})')
