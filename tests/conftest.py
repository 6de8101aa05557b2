import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def make_problem(controller='naive', cost='ext', N=30, nq=6, **over):
    from safe_mpc_amd.parser import Parameters
    from safe_mpc_amd.problem import OcpProblem
    from safe_mpc_amd.safe_set import SafeSetNet
    par = Parameters({}, 'z1')
    par.nq = nq
    par.n_dof_safe_set = over.pop('n_dof_safe_set', nq)
    par.net_size = [2 * nq, over.pop('hidden', 256), 1]
    par.N = N
    for k, v in over.items():
        setattr(par, k, v)
    prob = OcpProblem(par, controller, cost, N=N)
    net = SafeSetNet.from_params(par, prob.x_min, prob.x_max)
    prob.set_normalisation(net.mean, net.std)
    return par, prob, net


def halton(n, dim, skip=1):
    """Unscrambled Halton points in [0,1)^dim (guess_acados.py:79 uses scipy's qmc.Halton(scramble=False))."""
    primes = [2, 3, 5, 7, 11, 13, 17, 19, 23, 29]
    out = np.zeros((n, dim))
    for d in range(dim):
        b = primes[d]
        for i in range(n):
            f, r, k = 1.0, 0.0, i + skip
            while k > 0:
                f /= b
                r += f * (k % b)
                k //= b
            out[i, d] = r
    return out


def sample_instances(prob, B, seed=0, vel_scale=0.0, margin=0.05):
    """B collision-free initial states: Halton q0 inside the joint box (guess_acados.py:100,109), qd0 = vel_scale*U."""
    from oracle.oracle import Oracle
    nq = prob.nq
    o = Oracle(prob)
    pts = halton(4 * B + 16, nq, skip=1 + 7 * seed)
    lo, hi = prob.lbx[:nq] + margin, prob.ubx[:nq] - margin
    rng = np.random.default_rng(seed)
    xs = []
    for u in pts:
        q = lo + u * (hi - lo)
        x = np.concatenate([q, vel_scale * rng.uniform(-1, 1, nq) * prob.ubx[nq:]])
        ok = o.check_trajectory(x[None, None, :], prob.x_min, prob.x_max, 0.0, prob.row_lb, prob.row_ub)
        if ok[0]:
            xs.append(x)
        if len(xs) == B:
            break
    assert len(xs) == B, 'not enough collision-free samples'
    return np.array(xs)


def constant_guess(prob, x0, alpha=10.0, flag=1.0, ee_ref=None):
    B, N = x0.shape[0], prob.N
    xg = np.repeat(x0[:, None, :], N + 1, axis=1).copy()
    ug = np.zeros((B, N, prob.nu))
    p = np.zeros((B, N + 1, 5))
    p[:, :, :3] = prob.ee_ref if ee_ref is None else ee_ref
    p[:, :, 3] = alpha
    p[:, :, 4] = flag
    return xg, ug, p


@pytest.fixture(scope='session')
def z1_naive():
    return make_problem('naive')


@pytest.fixture(scope='session')
def z1_st():
    return make_problem('st')


def make_problem_fr7(controller='constraint_everywhere', cost='ext', N=40):
    """BASELINE config 4: 7-DoF Franka-class arm (config_fr7.yaml), sphere obstacle + floor, NN row on every node."""
    from safe_mpc_amd.parser import Parameters
    from safe_mpc_amd.problem import OcpProblem
    from safe_mpc_amd.safe_set import SafeSetNet
    par = Parameters({}, 'fr7', filename=os.path.join(ROOT, 'config_fr7.yaml'))
    par.N = N
    prob = OcpProblem(par, controller, cost, N=N)
    net = SafeSetNet.from_params(par, prob.x_min, prob.x_max)
    prob.set_normalisation(net.mean, net.std)
    return par, prob, net
