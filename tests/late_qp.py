"""Shared by the CPU and GPU suites: the solver-independent cross-check on QPs taken from a RUNNING closed loop.

`run_case(case, make_solver, make_oracle, ...)` drives a closed loop with the implementation under test (`make_solver(prob, net)`:
the HIP engine in the GPU suite, the oracle's test double in the CPU suite), and at the chosen steps exports every instance's stage
QP, solves it with tests/qp_ref.py (condensed dense QP, log-barrier Newton with numpy.linalg, exit 1e-11 -- no Riccati recursion, no
Mehrotra corrector, other variables) and checks, per late QP:
  (a) implementation == oracle at the RTI tolerance (same algorithm, rounding-different paths)           [when the two differ]
  (b) the step is feasible for the dense QP and its objective lies within the duality-gap bound 2 m x qp_tol of the dense optimum
  (c) the gap in the controls at the default exit stays below `gap_default` (the softness of late QPs, stated: DESIGN.md section 5)
  (d) re-solved with qp_tol = 1e-12 the step converges to the dense optimum: below `gap_tight`
Cases (VERDICT r4 item 5 added the last two):
  'st', 'constraint_everywhere'   Z1, N = 30, steps 40 and 100
  'receding'                      Z1, N = 30, the safe-set row switched per node through p[4]: on at the end node (soft) and at ONE
                                  running node r (hard) that recedes with the step as in controller.py:452-469
  'fr7'                           BASELINE config 4's problem: 7-DoF, N = 40, safe-set row on every node, steps 20 and 40
"""
import numpy as np

from conftest import constant_guess, make_problem, make_problem_fr7, sample_instances
from qp_ref import condense, solve_condensed

CASES = {
    #                         steps      nq  N   B
    'st':                    ((40, 100), 6, 30, 12),
    'constraint_everywhere': ((40, 100), 6, 30, 12),
    'receding':              ((40, 100), 6, 30, 12),
    'fr7':                   ((20, 40), 7, 40, 6),
}


def build(case, **over):
    if case == 'fr7':
        par, prob, net = make_problem_fr7(N=40)
        for k, v in over.items():
            setattr(prob.desc, k, v)
        return par, prob, net
    return make_problem(case, 'ext', N=30, **over)


def receding_flags(p, N, j):
    """controller.py:452-469 with a running node that recedes with the step: r = N - (j mod (N - 2)) in 3..N"""
    r = N - (j % (N - 2))
    p[:, :, 4] = -1.0
    p[:, N, 4] = 1.0
    p[:, 0, 4] = 1.0
    if r < N:
        p[:, r, 4] = 1.0
    return r


def run_case(case, make_solver, make_oracle, steps=None, B=None, gap_default=5e-2, gap_tight=2e-3, feas_tol=1e-6, same_as_oracle=True):
    steps_d, nq, N, B_d = CASES[case]
    steps = steps_d if steps is None else steps
    B = B_d if B is None else B
    par, prob, net = build(case)
    par_t, prob_t, _ = build(case, qp_tol=1e-12, qp_tol_res=1e-8)
    s, s_tight, o = make_solver(prob, net), make_solver(prob_t, net), make_oracle(prob, net)
    x = sample_instances(prob, B, seed=0)
    xg, ug, p = constant_guess(prob, x, ee_ref=prob.ee_ref)
    fails = np.zeros(B, int)
    worst = {'gap_default': 0.0, 'gap_tight': 0.0, 'obj': 0.0, 'it': 0.0}
    checked, rows = 0, []
    for j in range(max(steps) + 1):
        xg = s.guess_correction(xg, ug)
        if case == 'receding':
            receding_flags(p, N, j)
        xt, ut, st, it = s.solve(x, xg, ug, p)
        xt, ut, st, it = np.asarray(xt), np.asarray(ut), np.asarray(st), np.asarray(it)
        if j in steps:
            xq, uq, sq, iq = s_tight.solve(x, xg, ug, p)
            uq, sq = np.asarray(uq), np.asarray(sq)
            if same_as_oracle:
                xo, uo, so, io = o.solve_batch(x, xg, ug, p)
                assert np.array_equal(st, so) and np.abs(it - io).max() <= 2, (case, j, st, so, it, io)
                assert np.abs(ut - uo).max() < 1e-4 * (1 + np.abs(uo).max()), (case, j)                          # (a)
            w_step = {'gap_default': 0.0, 'gap_tight': 0.0, 'obj': 0.0}
            for b in np.argsort(-it):                      # slowest first: it is in the set whatever else is
                if st[b] != 0 or sq[b] != 0:
                    continue
                cq = condense(o.build_qp(x[b], xg[b], ug[b], p[b]), N, nq, par.dt)
                v, _, _, nit = solve_condensed(cq)
                assert nit < 150
                sw, G, h = cq['soft_w'], cq['G'], cq['h']

                def obj(w):      # quadratic + the L1 penalty of the soft rows' violation (their slack eliminated)
                    r = G @ w - h
                    return 0.5 * w @ cq['H'] @ w + cq['g'] @ w + np.sum(np.where(sw >= 0, sw * np.maximum(r, 0.0), 0.0))
                du, dq = (ut[b] - ug[b]).reshape(-1), (uq[b] - ug[b]).reshape(-1)
                m = G.shape[0] + int((sw >= 0).sum())
                # (b) feasible (hard rows; feas_tol: the engine's fp32 network row against the oracle's fp64 restatement in the dense
                #     QP's data) and within the duality-gap bound m x qp_tol of the optimum (x2 + the same data noise)
                assert np.max(np.where(sw >= 0, -1.0, G @ du - h)) < feas_tol, (case, j, b)
                gap = obj(du) - obj(v)
                assert -feas_tol * (1 + abs(obj(v))) < gap < 2 * m * 1e-8 + feas_tol * (1 + abs(obj(v))), (case, j, b, gap)
                g0 = np.abs(du - v).max() / (1 + np.abs(v).max())
                g1 = np.abs(dq - v).max() / (1 + np.abs(v).max())
                assert g0 < gap_default, (case, j, b, g0)                                                       # (c)
                assert g1 < gap_tight, (case, j, b, g1)                                                         # (d)
                for k_, v_ in (('gap_default', g0), ('gap_tight', g1), ('obj', gap / (1 + abs(obj(v))))):
                    w_step[k_] = max(w_step[k_], v_)
                    worst[k_] = max(worst[k_], v_)
                checked += 1
            rows.append((j, float(it.mean()), int(it.max()), w_step['gap_default'], w_step['gap_tight'], w_step['obj']))
        fails = np.where(st == 0, 0, fails + 1)
        xg, ug, u = s.provide_control((fails == 0).astype(np.int32), xt, ut, xg, ug)
        x = s.plant_step(x, u)[0]
        x = np.asarray(x)
    return checked, worst, rows
