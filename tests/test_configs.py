"""The non-headline BASELINE configurations as parity / property cases (C2, C3, C4)."""
import numpy as np
import pytest

from conftest import constant_guess, make_problem, make_problem_fr7, sample_instances
from oracle.oracle import Oracle


def test_fr7_problem_assembly_and_oracle_rows():
    """C4 on CPU: 7-DoF chain, capsule-sphere / sphere-sphere / sphere-plane rows, 14-input network."""
    par, prob, net = make_problem_fr7(N=6)
    assert prob.nq == 7 and prob.desc.n_rows == 4 and net.dims == [14, 256, 256, 256, 1]
    kinds = [r.kind for r in prob.rows]
    assert kinds == [2, 2, 3, 4]
    o = Oracle(prob, (net.weights, net.biases))
    x0 = sample_instances(prob, 3, seed=2)
    xg, ug, p = constant_guess(prob, x0, ee_ref=prob.ee_ref)
    ev = o.eval_nodes(xg, ug, p)
    q = x0[0, :7]
    pts = o.points(q)
    ball = np.array([0.45, 0.0, 0.35])
    # sphere-sphere row = |ee - ball|^2 (env_model.py:300-301); sphere-plane row = z of the ee sphere minus the floor level
    assert abs(ev[0, 1]['row_val'][2] - np.sum((pts[prob.desc.ee_point] - ball) ** 2)) < 1e-12
    assert abs(ev[0, 1]['row_val'][3] - pts[prob.rows[3].pa][2]) < 1e-12
    # capsule-sphere row: brute-force point-segment distance (utils.py:115-118)
    A, Bp = pts[prob.rows[1].pa], pts[prob.rows[1].pb]
    t = np.clip((ball - A) @ (Bp - A) / prob.rows[1].len2, 0, 1)
    assert abs(ev[0, 1]['row_val'][1] - np.sum((ball - (A + (Bp - A) * t)) ** 2)) < 1e-12
    xo, uo, st, it = o.solve_batch(x0, xg, ug, p)
    assert np.all(st == 0) and np.allclose(xo[:, 0], x0)


@pytest.mark.gpu
def test_c4_fr7_parity_on_engine():
    from safe_mpc_amd.solver import BatchedOcpSolver
    par, prob, net = make_problem_fr7(N=40)
    s, o = BatchedOcpSolver(prob, net), Oracle(prob, (net.weights, net.biases))
    B = 24
    x0 = sample_instances(prob, B, seed=3, vel_scale=0.1)
    xg, ug, p = constant_guess(prob, x0, ee_ref=prob.ee_ref)
    a, b = s.eval_nodes(xg, ug, p), o.eval_nodes(xg, ug, p)
    for f, n in [('tau', 7), ('M', 49), ('dtau_dq', 49), ('dtau_dv', 49), ('row_val', 4), ('row_grad', 28), ('cost_hess_qq', 49)]:
        assert np.abs(a[f][..., :n] - b[f][..., :n]).max() < 1e-9 * (1 + np.abs(b[f][..., :n]).max()), f
    xa, ua, sa, ia = s.solve(x0, xg, ug, p)
    xb, ub, sb, ib = o.solve_batch(x0, xg, ug, p)
    assert np.array_equal(sa, sb)
    ok = sb == 0
    assert ok.sum() >= B - 2 and np.abs(ua[ok] - ub[ok]).max() < 1e-4 * (1 + np.abs(ub[ok]).max())


@pytest.mark.gpu
def test_c2_full_size_model_noise_properties():
    """C2 at its stated size: B = 65 536, N = 30, per-instance perturbed plants (256 distinct draws of
    utils.py:138-166 laid out as one table per instance), torque noise, two closed-loop steps; size-independent properties
    + oracle spot parity on 64 instances + plant parity on per-instance tables."""
    from safe_mpc_amd import closed_loop as cl
    from safe_mpc_amd.solver import BatchedOcpSolver
    par, prob, net = make_problem('st', N=30)
    s = BatchedOcpSolver(prob, net)
    B = 65536
    base = sample_instances(prob, 512, seed=6)
    x0 = base[np.arange(B) % 512]
    xg, ug, p = constant_guess(prob, x0)
    # 65 536 distinct plants, seed = instance id (utils.py:126-171 per model) -- but for one stretch that repeats
    # the first 512 plants, so that "same start + same model = same bits wherever it sits in the batch" can be asserted below
    jt = cl.perturbed_joint_tables_batched(par, 6, 10.0, np.arange(B))
    jt[512 * 64:512 * 65] = jt[:512]
    assert len({jt[i]['mass'].tobytes() for i in range(0, B, 97)}) == len(range(0, B, 97))          # (distinct draws)
    rng = np.random.default_rng(0)
    tn = rng.normal(0, prob.tau_max * 0.01, (B, 6))
    x = x0
    dt = par.dt
    for step in range(2):
        xo, uo, st, it = s.solve(x, xg, ug, p)
        ok = st == 0
        assert ok.mean() > 0.99
        assert np.allclose(xo[:, 0], x, atol=1e-12)
        assert np.allclose(xo[ok, 1:, :6], xo[ok, :-1, :6] + dt * xo[ok, :-1, 6:] + 0.5 * dt * dt * uo[ok], atol=1e-9)
        assert np.all(xo[ok, 1:] >= prob.lbx - 1e-6) and np.all(xo[ok, 1:] <= prob.ubx + 1e-6)
        # instances that share a start and a model are bit-identical wherever they sit in the batch
        if step == 0:
            assert np.array_equal(uo[:512], uo[512 * 64:512 * 65]) and np.array_equal(it[:512], it[512 * 64:512 * 65])
        xg, ug, ua = s.provide_control(ok.astype(np.int32), xo, uo, xg, ug)
        x, _ = s.plant_step(x, ua, jt, tn)
        xg = s.guess_correction(xg, ug)
    assert np.all(np.isfinite(x))
    assert it.max() <= 25
    o = Oracle(prob, (net.weights, net.biases))
    sl = np.r_[0:32, B - 32:B]
    xb, ub, sb, ib = o.solve_batch(x[sl], xg[sl], ug[sl], p[sl])
    xa, ua2, sa, ia = s.solve(x, xg, ug, p)
    assert np.array_equal(sa[sl], sb) and np.abs(ua2[sl] - ub).max() < 1e-4 * (1 + np.abs(ub).max())
    xn_o, _ = o.plant_step(x[sl], ua[sl], jt[sl], tn[sl])
    xn_g, _ = s.plant_step(x[sl], ua[sl], jt[sl], tn[sl])
    assert np.allclose(xn_o, xn_g, atol=1e-9)


@pytest.mark.gpu
def test_c4_per_gpu_size_iteration_bound():
    """C4 at its per-GPU size (131 072 over 8 GPUs = 16 384 each): 7-DoF, N = 40, safe-set row on every node, first step
    from the constant guess.  The interior-point iterations of EVERY instance stay bounded (round 1 had one instance in
    4 096 jam at 66 iterations on a collision boundary and stretch the launch 4x); properties + oracle spot parity."""
    from safe_mpc_amd.solver import BatchedOcpSolver
    par, prob, net = make_problem_fr7(N=40)
    s, o = BatchedOcpSolver(prob, net), Oracle(prob, (net.weights, net.biases))
    B = 16384
    base = sample_instances(prob, 4096, seed=3, vel_scale=0.0)
    x0 = base[np.arange(B) % 4096]
    xg, ug, p = constant_guess(prob, x0, ee_ref=prob.ee_ref)
    xa, ua, sa, ia = s.solve(x0, xg, ug, p)
    ok = sa == 0
    assert ok.mean() > 0.99
    assert ia.max() <= 25, f'IPM straggler: max {ia.max()} iterations (instance {int(ia.argmax())})'
    dt = par.dt
    assert np.allclose(xa[ok, 1:, 7:], xa[ok, :-1, 7:] + dt * ua[ok], atol=1e-9)
    worst = np.argsort(-ia)[:16]                         # the slowest instances are the interesting ones for parity
    sl = np.r_[worst, 0:16]
    xb, ub, sb, ib = o.solve_batch(x0[sl], xg[sl], ug[sl], p[sl])
    assert np.array_equal(sa[sl], sb)
    okb = sb == 0
    assert np.abs(ua[sl][okb] - ub[okb]).max() < 1e-4 * (1 + np.abs(ub[okb]).max())
    assert ib.max() <= 25


@pytest.mark.gpu
def test_c3_horizon_alpha_sweep_grid():
    """C3: the (N, alpha) grid of run_mpc_horizons.sh / run_mpc_alphas.sh as batch axes: one handle, set_horizon per group,
    alpha per instance through p[:, :, 3]."""
    from safe_mpc_amd.sharding import shard_by_horizon
    from safe_mpc_amd.solver import BatchedOcpSolver
    par, prob, net = make_problem('st', N=40)
    s, o = BatchedOcpSolver(prob, net), Oracle(prob, (net.weights, net.biases))
    # 32 768 instances over 8 GPUs = 4 096 per GPU: rank 3's share of the full grid (N x alpha x ~1 638 starts each)
    n_full = 32768
    horizons = np.repeat([20, 25, 30, 35, 40], n_full // 5 + 1)[:n_full]
    alphas = np.tile([20.0, 30.0, 40.0, 50.0], n_full // 4)
    starts = sample_instances(prob, 512, seed=8)
    x_all = starts[np.arange(n_full) % 512]
    owned = shard_by_horizon(horizons, 8, 3)
    assert abs(sum(len(v) for v in owned.values()) - 4096) <= 4      # 4 095 here; ranks 0 and 1 take the remainders
    for N, idx_full in owned.items():
        s.set_horizon(N)
        B = len(idx_full)
        xg = np.repeat(x_all[idx_full][:, None, :], N + 1, axis=1)
        p = np.zeros((B, N + 1, 5))
        p[:, :, :3], p[:, :, 4] = prob.ee_ref, 1.0
        p[:, :, 3] = alphas[idx_full][:, None]
        xf, uf, sf, itf = s.solve(x_all[idx_full], xg, np.zeros((B, N, 6)), p)
        assert xf.shape == (B, N + 1, 12) and (sf == 0).mean() > 0.99 and itf.max() <= 25
        okf = sf == 0
        assert np.allclose(xf[okf, 1:, 6:], xf[okf, :-1, 6:] + par.dt * uf[okf], atol=1e-9)
    # oracle parity on a thinned copy of the same grid
    for N, idx in shard_by_horizon(horizons, 8, 3).items():
        idx = idx[::103]
        s.set_horizon(N)
        o.set_horizon(N)
        B = len(idx)
        xg = np.repeat(x_all[idx][:, None, :], N + 1, axis=1)
        ug = np.zeros((B, N, 6))
        p = np.zeros((B, N + 1, 5))
        p[:, :, :3], p[:, :, 4] = prob.ee_ref, 1.0
        p[:, :, 3] = alphas[idx][:, None]
        xa, ua, sa, ia = s.solve(x_all[idx], xg, ug, p)
        xb, ub, sb, ib = o.solve_batch(x_all[idx], xg, ug, p)
        assert xa.shape == (B, N + 1, 12) and np.array_equal(sa, sb)
        assert np.abs(ua - ub).max() < 1e-4 * (1 + np.abs(ub).max())


@pytest.mark.gpu
def test_c4_degenerate_start_on_a_collision_bound():
    """Round 1's straggler: a 7-DoF start whose constant guess sits ON the capsule-sphere bound (margin 4.7e-6) at all 40
    nodes -- a degenerate QP (the same active row at every stage).  66 interior-point iterations in round 1, 33 with the
    starting point scaled by the rows' gradients; engine and oracle agree on the count and on the solution."""
    import os
    from safe_mpc_amd.solver import BatchedOcpSolver
    par, prob, net = make_problem_fr7(N=40)
    s, o = BatchedOcpSolver(prob, net), Oracle(prob, (net.weights, net.biases))
    x0 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'c4_degenerate_start.npz'))['x0']
    xg, ug, p = constant_guess(prob, x0, ee_ref=prob.ee_ref)
    ev = o.eval_nodes(xg, ug, p)
    assert 0.0 < ev[0, 1]['row_val'][0] - prob.row_lb[0] < 1e-5          # the row is (barely) satisfied at the guess
    xa, ua, sa, ia = s.solve(x0, xg, ug, p)
    xb, ub, sb, ib = o.solve_batch(x0, xg, ug, p)
    assert sa[0] == 0 and sb[0] == 0
    assert ia[0] <= 40 and abs(int(ia[0]) - int(ib[0])) <= 2
    assert np.abs(ua - ub).max() < 1e-4 * (1 + np.abs(ub).max())
    # ... and with the IPM's stall exit switched on at RealReceding's default the degenerate but FEASIBLE start still converges
    par2, prob2, net2 = make_problem_fr7(N=40)
    prob2.desc.qp_stall_iters = 24
    xc, uc, sc, ic = BatchedOcpSolver(prob2, net2).solve(x0, xg, ug, p)
    assert sc[0] == 0 and ic[0] == ia[0] and np.array_equal(uc, ua)
