"""Pins the oracle's rigid-body pieces from first principles (the reference holds no golden vectors).

Covers SURVEY 8(a) rows a1, a2, a4, a10: double integrator, RNEA = M(q)u + h(q,qd) and its Jacobians, FK / EE point,
cost gradient and Hessian.
"""
import numpy as np
import pytest

from conftest import make_problem
from oracle.oracle import Oracle
from safe_mpc_amd.problem import JOINT_DTYPE
from safe_mpc_amd.urdf import _axis_angle

G = 9.80665


def _single_pendulum_joint(m=1.3, l=0.7, izz=0.05):
    """one revolute joint about world y, COM at distance l along x"""
    J = np.zeros(1, JOINT_DTYPE)
    J[0]['R0'] = np.eye(3).reshape(-1)
    J[0]['axis'] = [0, 1, 0]
    J[0]['mass'] = m
    J[0]['com'] = [l, 0, 0]
    J[0]['inertia'] = [0.01, 0, 0, izz, 0, 0.02]
    J[0]['tau_max'] = 100
    return J


def _rnea_joints(J, q, qd, qdd):
    import ctypes as C
    from oracle.oracle import lib, _p
    nq = len(J)
    tau = np.zeros(nq)
    grav = np.array([0, 0, -G])
    lib().orc_rnea_joints(J.ctypes.data_as(C.c_void_p), nq, _p(grav), _p(np.ascontiguousarray(q, float)),
                          _p(np.ascontiguousarray(qd, float)), _p(np.ascontiguousarray(qdd, float)), _p(tau))
    return tau


def test_single_pendulum_closed_form():
    m, l, izz = 1.3, 0.7, 0.05
    J = _single_pendulum_joint(m, l, izz)
    for q, qd, qdd in [(0.3, -1.1, 2.0), (-2.0, 0.4, -0.7), (0.0, 0.0, 0.0)]:
        tau = _rnea_joints(J, [q], [qd], [qdd])
        # rotation about +y by q takes x to (cos q, 0, -sin q): height of the COM is -l sin q
        expect = (izz + m * l * l) * qdd - m * G * l * np.cos(q)
        assert abs(tau[0] - expect) < 1e-12


def test_double_pendulum_closed_form():
    """planar 2R arm in the x-z plane (both axes y), point masses at the link tips: textbook M, C, g."""
    m1, m2, l1, l2 = 0.9, 0.6, 0.5, 0.4
    J = np.zeros(2, JOINT_DTYPE)
    for i, (m, l, p0) in enumerate([(m1, l1, [0, 0, 0]), (m2, l2, [l1, 0, 0])]):
        J[i]['R0'] = np.eye(3).reshape(-1)
        J[i]['p0'] = p0
        J[i]['axis'] = [0, 1, 0]
        J[i]['mass'] = m
        J[i]['com'] = [l, 0, 0]
        J[i]['tau_max'] = 100
    rng = np.random.default_rng(1)
    for _ in range(5):
        q, qd, qdd = rng.uniform(-2, 2, 2), rng.uniform(-2, 2, 2), rng.uniform(-3, 3, 2)
        tau = _rnea_joints(J, q, qd, qdd)
        c2, s2 = np.cos(q[1]), np.sin(q[1])
        M = np.array([[m1 * l1 ** 2 + m2 * (l1 ** 2 + l2 ** 2 + 2 * l1 * l2 * c2), m2 * (l2 ** 2 + l1 * l2 * c2)],
                      [m2 * (l2 ** 2 + l1 * l2 * c2), m2 * l2 ** 2]])
        h = m2 * l1 * l2 * s2
        Cv = np.array([-h * (2 * qd[0] * qd[1] + qd[1] ** 2), h * qd[0] ** 2])
        # a rotation by +q about y lowers the tip: z = -l sin(q)  ->  V = -g (m1 l1 s1 + m2 (l1 s1 + l2 s12))
        g = -G * np.array([(m1 + m2) * l1 * np.cos(q[0]) + m2 * l2 * np.cos(q[0] + q[1]),
                           m2 * l2 * np.cos(q[0] + q[1])])
        assert np.allclose(tau, M @ qdd + Cv + g, atol=1e-12)


@pytest.fixture(scope='module')
def z1():
    par, prob, net = make_problem('naive')
    return par, prob, Oracle(prob, (net.weights, net.biases))


def test_fk_matches_transform_product(z1):
    par, prob, o = z1
    rng = np.random.default_rng(2)
    for _ in range(4):
        q = rng.uniform(prob.lbx[:6], prob.ubx[:6])
        R, p = o.fk(q)
        Rn, pn = np.eye(3), np.zeros(3)
        for i, j in enumerate(prob.chain.joints):
            pn = pn + Rn @ j.p0
            Rn = Rn @ j.R0 @ _axis_angle(j.axis, q[i])
            assert np.allclose(R[i], Rn, atol=1e-13) and np.allclose(p[i], pn, atol=1e-13)


def test_mass_matrix_symmetric_pd_and_linear_in_u(z1):
    par, prob, o = z1
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(prob.lbx[:6], prob.ubx[:6]), rng.uniform(-1, 1, 6)])
    u = rng.uniform(-5, 5, 6)
    N = prob.N
    xg = np.tile(x, (1, N + 1, 1)); ug = np.tile(u, (1, N, 1)); p = np.zeros((1, N + 1, 5))
    ev = o.eval_nodes(xg, ug, p)[0, 0]
    M = ev['M'][:36].reshape(6, 6)
    assert np.allclose(M, M.T, atol=1e-12)
    assert np.all(np.linalg.eigvalsh(M) > 0)
    h = o.rnea(x[:6], x[6:], np.zeros(6))
    assert np.allclose(ev['tau'][:6], M @ u + h, atol=1e-11)
    # energy consistency: qd^T (tau - g) = d/dt kinetic energy  ->  qd^T C(q,qd) qd-terms vanish for Mdot - 2C skew;
    # check through a finite difference of T = 0.5 qd^T M qd along the motion with zero gravity contribution removed
    g = o.rnea(x[:6], np.zeros(6), np.zeros(6))
    eps = 1e-6
    def kinetic(q, v):
        xg2 = np.tile(np.concatenate([q, v]), (1, N + 1, 1))
        Mq = o.eval_nodes(xg2, ug, p)[0, 0]['M'][:36].reshape(6, 6)
        return 0.5 * v @ Mq @ v
    qd, qdd = x[6:], u
    Tdot = (kinetic(x[:6] + eps * qd, qd + eps * qdd) - kinetic(x[:6] - eps * qd, qd - eps * qdd)) / (2 * eps)
    assert abs(Tdot - qd @ (ev['tau'][:6] - g)) < 1e-6


def test_tau_jacobians_finite_difference(z1):
    par, prob, o = z1
    rng = np.random.default_rng(4)
    N = prob.N
    for _ in range(3):
        x = np.concatenate([rng.uniform(prob.lbx[:6], prob.ubx[:6]), rng.uniform(-2, 2, 6)])
        u = rng.uniform(-8, 8, 6)
        xg = np.tile(x, (1, N + 1, 1)); ug = np.tile(u, (1, N, 1)); p = np.zeros((1, N + 1, 5))
        ev = o.eval_nodes(xg, ug, p)[0, 0]
        dq, dv = ev['dtau_dq'][:36].reshape(6, 6), ev['dtau_dv'][:36].reshape(6, 6)
        eps = 1e-6
        for j in range(6):
            e = np.zeros(6); e[j] = eps
            fd_q = (o.rnea(x[:6] + e, x[6:], u) - o.rnea(x[:6] - e, x[6:], u)) / (2 * eps)
            fd_v = (o.rnea(x[:6], x[6:] + e, u) - o.rnea(x[:6], x[6:] - e, u)) / (2 * eps)
            assert np.allclose(dq[:, j], fd_q, atol=2e-7, rtol=1e-7)
            assert np.allclose(dv[:, j], fd_v, atol=2e-7, rtol=1e-7)


@pytest.mark.parametrize('cost', ['ext', 'nls'])
def test_cost_gradient_and_hessian(cost):
    par, prob, net = make_problem('naive', cost)
    o = Oracle(prob)
    rng = np.random.default_rng(5)
    N = prob.N
    ref = np.array(par.ee_ref)
    Q = par.Q_weight

    def ee(q):
        return o.points(q)[prob.desc.ee_point]

    def grad(q):
        xg = np.tile(np.concatenate([q, np.zeros(6)]), (1, N + 1, 1))
        p = np.zeros((1, N + 1, 5)); p[:, :, :3] = ref
        return o.eval_nodes(xg, np.zeros((1, N, 6)), p)[0, 1]

    q = rng.uniform(prob.lbx[:6], prob.ubx[:6])
    ev = grad(q)
    assert np.allclose(ev['ee'], ee(q), atol=1e-14)
    eps = 1e-6
    g_fd = np.zeros(6); J = np.zeros((3, 6))
    for j in range(6):
        e = np.zeros(6); e[j] = eps
        lp, lm = Q * np.sum((ee(q + e) - ref) ** 2), Q * np.sum((ee(q - e) - ref) ** 2)
        g_fd[j] = (lp - lm) / (2 * eps)
        J[:, j] = (ee(q + e) - ee(q - e)) / (2 * eps)
    assert np.allclose(ev['cost_grad_q'][:6], g_fd, atol=1e-6)
    H = ev['cost_hess_qq'][:36].reshape(6, 6)
    assert np.allclose(H, H.T, atol=1e-10)
    if cost == 'nls':
        assert np.allclose(H, 2 * Q * J.T @ J, atol=1e-5)            # Gauss-Newton (cost_definition.py:69-81)
    else:
        H_fd = np.zeros((6, 6))
        for j in range(6):
            e = np.zeros(6); e[j] = eps
            H_fd[:, j] = (grad(q + e)['cost_grad_q'][:6] - grad(q - e)['cost_grad_q'][:6]) / (2 * eps)
        assert np.allclose(H, H_fd, atol=1e-5)                      # exact Hessian of the EXTERNAL cost


def test_double_integrator_closed_form(z1):
    """a1: guessCorrection is N applications of f_disc; closed form q_k = q0 + k dt v0 + sum_j (k-j-1/2) dt^2 u_j."""
    par, prob, o = z1
    rng = np.random.default_rng(6)
    N, dt = prob.N, par.dt
    x0 = rng.uniform(-1, 1, 12)
    ug = rng.uniform(-3, 3, (1, N, 6))
    xg = np.zeros((1, N + 1, 12)); xg[0, 0] = x0
    out = o.guess_correction(xg, ug)[0]
    for k in (1, 7, N):
        v = x0[6:] + dt * ug[0, :k].sum(0)
        q = x0[:6] + k * dt * x0[6:] + dt ** 2 * sum((k - j - 0.5) * ug[0, j] for j in range(k))
        assert np.allclose(out[k, :6], q, atol=1e-13) and np.allclose(out[k, 6:], v, atol=1e-13)


def test_momentum_operator_closed_form_used_by_the_stage_builder():
    """kernel_build.hpp (round 4) replaces the 36-entry momentum operator of rnea_deriv.hpp,
        B m = Y (m x v) + m x* (Y v) + v x* (Y m),
    by its closed form with 12 numbers: with v = (w; u), h = Y v = (n; f), m = (a; l):  B m = (Baa a ; -2 f x a),
    Baa = [w]x I - I [w]x - [n]x - (u mc^T + mc u^T - 2 (mc . u) 1); the blocks acting on l vanish identically.  Checked here
    against the column-by-column definition for random bodies and motions (the GPU test compares the two kernels' outputs)."""
    rng = np.random.default_rng(7)

    def sk(x):
        return np.array([[0, -x[2], x[1]], [x[2], 0, -x[0]], [-x[1], x[0], 0]])
    for _ in range(20):
        m = rng.uniform(0.2, 5.0)
        c = rng.normal(size=3)
        mc = m * c
        A = rng.normal(size=(3, 3))
        Io = A @ A.T + m * (c @ c * np.eye(3) - np.outer(c, c))          # rotational inertia about the origin

        def Y(v):
            a, l = v[:3], v[3:]
            return np.concatenate([Io @ a + np.cross(mc, l), m * l + np.cross(a, mc)])

        def mxm(m1, m2):
            return np.concatenate([np.cross(m1[:3], m2[:3]), np.cross(m1[:3], m2[3:]) + np.cross(m1[3:], m2[:3])])

        def mxf(mv, f):
            return np.concatenate([np.cross(mv[:3], f[:3]) + np.cross(mv[3:], f[3:]), np.cross(mv[:3], f[3:])])
        v = rng.normal(size=6)
        h = Y(v)
        B = np.zeros((6, 6))
        for k in range(6):
            e = np.zeros(6)
            e[k] = 1.0
            B[:, k] = Y(mxm(e, v)) + mxf(e, h) + mxf(v, Y(e))
        w, u, n, f = v[:3], v[3:], h[:3], h[3:]
        Baa = sk(w) @ Io - Io @ sk(w) - sk(n) - (np.outer(u, mc) + np.outer(mc, u) - 2.0 * (mc @ u) * np.eye(3))
        scale = np.abs(B).max()
        assert np.abs(B[:3, :3] - Baa).max() < 1e-13 * scale
        assert np.abs(B[:3, 3:]).max() < 1e-13 * scale and np.abs(B[3:, 3:]).max() < 1e-13 * scale
        assert np.abs(B[3:, :3] + 2.0 * sk(f)).max() < 1e-13 * scale
