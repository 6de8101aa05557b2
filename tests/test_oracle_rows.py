"""Pins the oracle's inequality rows: capsule distances (a5), the MLP and the safe-set row (a6-a8)."""
import numpy as np
import pytest
import torch

from conftest import make_problem
from oracle.oracle import Oracle


@pytest.fixture(scope='module')
def setup():
    par, prob, net = make_problem('st')
    return par, prob, net, Oracle(prob, (net.weights, net.biases))


def _seg_brute(A, B, C, D, n=400):
    t = np.linspace(0, 1, n)
    P = A[None] + t[:, None] * (B - A)[None]
    Qp = C[None] + t[:, None] * (D - C)[None]
    return np.min(np.sum((P[:, None, :] - Qp[None, :, :]) ** 2, axis=-1))


def test_segment_distance_vs_brute_force(setup):
    _, _, _, o = setup
    rng = np.random.default_rng(0)
    for _ in range(30):
        A, B, C, D = (rng.uniform(-1, 1, 3) for _ in range(4))
        d = o.segment_dist2(A, B, C, D)
        # the reference's formula (utils.py:94-113) carries a 1e-5 regulariser and a fixed clamp order: it is the true
        # distance up to that regularisation, never below it
        bf = _seg_brute(A, B, C, D)
        assert d >= bf - 1e-4 and d <= bf + 5e-3


def test_segment_distance_special_cases(setup):
    _, _, _, o = setup
    # crossing segments -> 0, parallel offset segments -> offset^2, endpoint-to-endpoint
    assert o.segment_dist2([-1, 0, 0], [1, 0, 0], [0, -1, 0], [0, 1, 0]) < 1e-12
    assert abs(o.segment_dist2([0, 0, 0], [1, 0, 0], [0, 0.3, 0], [1, 0.3, 0]) - 0.09) < 1e-9
    assert abs(o.segment_dist2([0, 0, 0], [1, 0, 0], [2, 0, 0], [3, 0, 0]) - 1.0) < 1e-9
    # closed form of the reference expression at one generic point, evaluated here in numpy
    A, B, C, D = np.array([0.1, 0.2, 0.3]), np.array([0.5, -0.1, 0.4]), np.array([0.5, 0.2, 0.0]), np.array([0.5, 0.2, 0.25])
    R = (B - A) @ (D - C); S1 = (B - A) @ (C - A); D1 = (B - A) @ (B - A); S2 = (D - C) @ (C - A); D2 = (D - C) @ (D - C)
    t = np.clip((S1 * D2 - S2 * R) / (D1 * D2 - (R ** 2 + 1e-5)), 0, 1)
    u = np.clip((t * R - S2) / D2, 0, 1)
    t = np.clip((u * R + S1) / D1, 0, 1)
    w = (B - A) * t - (D - C) * u - (C - A)
    assert abs(o.segment_dist2(A, B, C, D) - w @ w) < 1e-15


def test_row_values_and_gradients(setup):
    par, prob, net, o = setup
    rng = np.random.default_rng(1)
    N = prob.N
    nrows = prob.desc.n_rows
    assert nrows == 6                                     # config.yaml:205-216
    assert np.allclose(prob.row_lb, [(0.055 + 0.05) ** 2] * 3 + [(0.05 + 0.05) ** 2] * 3)

    def rows(q):
        xg = np.tile(np.concatenate([q, np.zeros(6)]), (1, N + 1, 1))
        return o.eval_nodes(xg, np.zeros((1, N, 6)), np.zeros((1, N + 1, 5)))[0, 1]

    checked = 0
    for _ in range(10):
        q = rng.uniform(prob.lbx[:6], prob.ubx[:6])
        ev = rows(q)
        pts = o.points(q)
        for r in range(nrows):
            row = prob.rows[r]
            d = o.segment_dist2(pts[row.pa], pts[row.pb], np.array(row.C), np.array(row.D))
            assert abs(ev['row_val'][r] - d) < 1e-14
        eps = 1e-7
        G = ev['row_grad'][:nrows * 6].reshape(nrows, 6)
        fd = np.zeros((nrows, 6))
        for j in range(6):
            e = np.zeros(6); e[j] = eps
            fd[:, j] = (rows(q + e)['row_val'][:nrows] - rows(q - e)['row_val'][:nrows]) / (2 * eps)
        # away from clamp switches the FD matches; count how often (kinks are measure-zero but FD straddles them)
        good = np.isclose(G, fd, atol=1e-6).all(axis=1)
        checked += good.sum()
        assert good.sum() >= nrows - 1
        # joint 6 never moves either capsule; forearm (link03) only sees joints 1..3
        assert np.all(G[:, 5] == 0) and np.all(G[3:, 3:] == 0)
    assert checked >= 50


def test_mlp_matches_torch(setup):
    par, prob, net, o = setup
    rng = np.random.default_rng(2)
    S = rng.uniform(-1.5, 1.5, (64, 12)).astype(np.float32)
    y_t, g_t = net.torch_value_and_grad(S)
    for i in range(64):
        y, g = o.mlp(S[i])
        assert abs(y - y_t[i]) <= 1e-5 * max(1.0, abs(y_t[i]))
        assert np.allclose(g, g_t[i], atol=2e-6, rtol=1e-4)
    assert net.dims == [12, 256, 256, 256, 1]
    assert sum(w.size for w in net.weights) + sum(b.size for b in net.biases) == 135169


@pytest.mark.parametrize('act', ['relu', 'elu', 'tanh', 'silu', 'gelu'])
def test_mlp_activations_match_torch(act):
    """parser.py:95-102: every activation the reference's parser offers, oracle (fp32 loops) vs torch autograd."""
    from conftest import make_problem
    from oracle.oracle import Oracle
    par, prob, net = make_problem('st', act=act)
    assert net.act == act
    o = Oracle(prob, (net.weights, net.biases, net.act))
    S = np.random.default_rng(4).uniform(-1.5, 1.5, (32, 12)).astype(np.float32)
    y_t, g_t = net.torch_value_and_grad(S)
    for i in range(32):
        y, g = o.mlp(S[i])
        assert abs(y - y_t[i]) <= 2e-5 * max(1.0, abs(y_t[i]))
        assert np.allclose(g, g_t[i], atol=5e-6, rtol=2e-4)


def test_nn_row_formula_and_gradient(setup):
    par, prob, net, o = setup
    rng = np.random.default_rng(3)
    alpha = 10.0
    for _ in range(5):
        x = np.concatenate([rng.uniform(prob.lbx[:6], prob.ubx[:6]), rng.uniform(-1, 1, 6)])
        g, dg = o.nn_row(x, alpha)
        # restate safe_set.py:82-94 in numpy + torch
        xc = x.copy(); xc[6] += par.eps
        vn = np.linalg.norm(xc[6:])
        s = np.concatenate([(xc[:6] - net.mean) / net.std, xc[6:] / vn]).astype(np.float32)
        y = net.model(torch.tensor(s)).item()
        assert abs(g - (y * (100 - alpha) / 100 - vn)) < 2e-5
        # gradient: FD through the oracle itself in float64 around the fp32 network is noisy -> compare with torch
        xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
        net64 = net.model.double()
        xcp = xt + torch.nn.functional.one_hot(torch.tensor(6), 12).double() * par.eps
        vnt = torch.linalg.norm(xcp[6:])
        st = torch.cat([(xcp[:6] - torch.tensor(net.mean)) / torch.tensor(net.std), xcp[6:] / vnt])
        gt = net64(st)[0] * (100 - alpha) / 100 - vnt
        gt.backward()
        net.model.float()
        assert np.allclose(dg, xt.grad.numpy(), atol=1e-4, rtol=1e-4)


def test_nn_row_at_rest_uses_eps(setup):
    """qd = 0: |v| = eps, direction = e_0 (safe_set.py:83): finite value, huge but finite velocity gradient."""
    par, prob, net, o = setup
    x = np.concatenate([0.5 * (prob.lbx[:6] + prob.ubx[:6]), np.zeros(6)])
    g, dg = o.nn_row(x, 10.0)
    assert np.isfinite(g) and np.all(np.isfinite(dg))
    assert abs(dg[6] + 1.0) < 1e-6          # d(-|v|)/dv_0 = -1, the direction term vanishes along v
