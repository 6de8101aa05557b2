"""Receiving end for REFERENCE-PRODUCED golden vectors.

The reference's own implementation of this path (acados / HPIPM / CasADi / l4casadi / adam) cannot run in this pipeline and the
reference holds no vectors, so parity is "unpinned" (DESIGN.md section 5).  A maintainer who CAN run the reference dumps one RTI
solve per file into tests/golden/ref_<name>.npz (INTEGRATION.md section 5 has the ten-line dump snippet and says which
AcadosOcpSolver.get(...) call fills which field); this file then holds the oracle (CPU suite) and the HIP engine (-m gpu) against
them.  With no ref_*.npz present the tests skip; the checker itself is exercised on a file in the same layout written from the
oracle into a temporary directory (accepts it, rejects a perturbed one), so the seam is known to work before the first real
vector arrives.

Layout of ref_<name>.npz (arrays as the reference holds them; B = instances in the file, usually 1):
    x0 [B, nx]   xg [B, N+1, nx]   ug [B, N, nu]   p [B, N+1, 5]      inputs of AbstractController.solve (controller.py:141-156)
    x [B, N+1, nx]   u [B, N, nu]   status [B]                         outputs (controller.py:158-165)
    time_lin, time_sim, time_qp, time_qp_solver_call, time_glob, time_reg, time_tot   [B] seconds, optional (controller.py:192-193)
    controller, cost, system   0-d strings: 'st' | 'naive' | ..., 'ext' | 'nls', 'z1' | 'fr7'
    late   0-d bool: the QP comes from a RUNNING closed loop (x0 off the guess, active rows) rather than a cold start
Tolerances = DESIGN.md section 5's honest expectation against an independent solver at HPIPM's exit level: status equal;
objective of the stage QP within 1e-5 relative; controls within 1e-6 (1 + |u|) cold, 1e-2 (1 + |u|) late."""
import glob
import os

import numpy as np
import pytest

from conftest import constant_guess, make_problem, make_problem_fr7, sample_instances
from qp_ref import condense

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
REF_FILES = sorted(glob.glob(os.path.join(GOLD, 'ref_*.npz')))
TIME_FIELDS = ('time_lin', 'time_sim', 'time_qp', 'time_qp_solver_call', 'time_glob', 'time_reg', 'time_tot')
TOL_OBJ, TOL_U_COLD, TOL_U_LATE = 1e-5, 1e-6, 1e-2


def problem_of(g):
    system, cont, cost = str(g['system']), str(g['controller']), str(g['cost'])
    N = g['xg'].shape[1] - 1
    if system == 'fr7':
        return make_problem_fr7(cont, cost, N=N)
    return make_problem(cont, cost, N=N, nq=g['x0'].shape[1] // 2)


def qp_objective(oracle, par, g, b, du):
    """value of instance b's stage QP (soft rows as their L1 penalty) at the control step du, through the dense condensed form"""
    N, nq = g['xg'].shape[1] - 1, g['x0'].shape[1] // 2
    cq = condense(oracle.build_qp(g['x0'][b], g['xg'][b], g['ug'][b], g['p'][b]), N, nq, par.dt)
    if cq['Z'] is not None:
        return None            # (equality rows eliminated through a null space: the controls are compared, not the objective)
    r = cq['G'] @ du - cq['h']
    return 0.5 * du @ cq['H'] @ du + cq['g'] @ du + np.sum(np.where(cq['soft_w'] >= 0, cq['soft_w'] * np.maximum(r, 0), 0.0))


def check_against_ref(g, solve, oracle, par):
    """`solve(x0, xg, ug, p) -> x, u, status, it` of the implementation under test against the vectors of one ref file"""
    for f in ('x0', 'xg', 'ug', 'p', 'x', 'u', 'status', 'controller', 'cost', 'system', 'late'):
        assert f in g, f'ref file lacks "{f}"'
    B, N = g['x0'].shape[0], g['xg'].shape[1] - 1
    assert g['x'].shape == g['xg'].shape and g['u'].shape == g['ug'].shape and g['p'].shape == (B, N + 1, 5)
    for f in TIME_FIELDS:
        if f in g:
            assert g[f].shape == (B,) and (g[f] >= 0).all()
    x, u, st, it = solve(g['x0'], g['xg'], g['ug'], g['p'])
    x, u, st = np.asarray(x), np.asarray(u), np.asarray(st)
    assert np.array_equal(st, g['status']), (st, g['status'])
    tol_u = TOL_U_LATE if bool(g['late']) else TOL_U_COLD
    for b in np.where(g['status'] == 0)[0]:
        scale = 1 + np.abs(g['u'][b]).max()
        assert np.abs(u[b] - g['u'][b]).max() < tol_u * scale, (b, np.abs(u[b] - g['u'][b]).max(), tol_u * scale)
        assert np.abs(x[b] - g['x'][b]).max() < tol_u * (1 + np.abs(g['x'][b]).max())
        f_ref = qp_objective(oracle, par, g, b, (g['u'][b] - g['ug'][b]).reshape(-1))
        f_own = qp_objective(oracle, par, g, b, (u[b] - g['ug'][b]).reshape(-1))
        if f_ref is not None:
            assert abs(f_own - f_ref) < TOL_OBJ * (1 + abs(f_ref)), (b, f_own, f_ref)


@pytest.mark.parametrize('path', REF_FILES or [None])
def test_oracle_against_reference_vectors(path):
    if path is None:
        pytest.skip(f'{len(REF_FILES)} reference-produced vector files (tests/golden/ref_*.npz) found: parity unpinned -- the reference cannot run in this pipeline (DESIGN.md section 5); see INTEGRATION.md section 5')
    from oracle.oracle import Oracle
    g = dict(np.load(path, allow_pickle=False))
    par, prob, net = problem_of(g)
    o = Oracle(prob, (net.weights, net.biases))
    check_against_ref(g, o.solve_batch, o, par)


@pytest.mark.gpu
@pytest.mark.parametrize('path', REF_FILES or [None])
def test_engine_against_reference_vectors(path):
    if path is None:
        pytest.skip(f'{len(REF_FILES)} reference-produced vector files (tests/golden/ref_*.npz) found: parity unpinned (see test_oracle_against_reference_vectors)')
    from oracle.oracle import Oracle
    from safe_mpc_amd.solver import BatchedOcpSolver
    g = dict(np.load(path, allow_pickle=False))
    par, prob, net = problem_of(g)
    s = BatchedOcpSolver(prob, net)
    check_against_ref(g, s.solve, Oracle(prob, (net.weights, net.biases)), par)


def _write_ref_layout(path, cont='st', late=False):
    """a file in the ref layout, filled from the ORACLE (self-test of the seam only: this is not reference data and is never committed)"""
    from oracle.oracle import Oracle
    par, prob, net = make_problem(cont, 'ext', N=8)
    o = Oracle(prob, (net.weights, net.biases))
    x0 = sample_instances(prob, 2, seed=3)
    xg, ug, p = constant_guess(prob, x0)
    if late:     # three closed-loop steps first
        x = x0
        for _ in range(3):
            xg = o.guess_correction(xg, ug)
            xt, ut, st, it = o.solve_batch(x, xg, ug, p)
            xg, ug, ua = o.provide_control((st == 0).astype(np.int32), xt, ut, xg, ug)
            x, _ = o.plant_step(x, ua)
        x0, xg = x, o.guess_correction(xg, ug)
    x, u, st, it = o.solve_batch(x0, xg, ug, p)
    np.savez(path, x0=x0, xg=xg, ug=ug, p=p, x=x, u=u, status=st, controller=cont, cost='ext', system='z1', late=late,
             **{f: np.zeros(2) for f in TIME_FIELDS})
    return par, prob, net, o


@pytest.mark.parametrize('late', [False, True])
def test_checker_accepts_its_layout_and_rejects_a_wrong_vector(tmp_path, late):
    path = str(tmp_path / 'ref_selftest.npz')
    par, prob, net, o = _write_ref_layout(path, late=late)
    g = dict(np.load(path, allow_pickle=False))
    assert problem_of(g)[1].N == 8
    check_against_ref(g, o.solve_batch, o, par)                                    # the layout round-trips
    bad = dict(g)
    bad['u'] = g['u'] + 0.05 * (1 + np.abs(g['u']).max())                          # well outside the late tolerance
    with pytest.raises(AssertionError):
        check_against_ref(bad, o.solve_batch, o, par)
    bad = dict(g)
    bad['status'] = g['status'] + 4
    with pytest.raises(AssertionError):
        check_against_ref(bad, o.solve_batch, o, par)
    del bad['late']
    with pytest.raises(AssertionError):
        check_against_ref(bad, o.solve_batch, o, par)
