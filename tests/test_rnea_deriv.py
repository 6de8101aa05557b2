"""Closed-form RNEA derivatives of the product (safe_mpc_amd/csrc/rnea_deriv.hpp, compiled here for the host with g++) against
the oracle's dual-number differentiation of a different (link-frame) recursion: tau, M = dtau/du, dtau/dq, dtau/dqd
(env_model.py:80-83 and its Jacobian).  Two independent derivations agreeing to 1e-10 is the point of the test."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import make_problem, make_problem_fr7
from oracle.oracle import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def host_lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp('native') / 'libhost_rnea.so')
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-Wno-unknown-pragmas',
                           os.path.join(ROOT, 'tests', 'native', 'host_rnea_deriv.cpp'), '-o', out])
    return C.CDLL(out)


@pytest.mark.parametrize('which', ['z1_6', 'z1_5', 'fr7'])
def test_closed_form_derivatives_match_dual_numbers(host_lib, which):
    if which == 'fr7':
        par, prob, net = make_problem_fr7(N=3)
    else:
        par, prob, net = make_problem('naive', 'ext', N=3, nq=6 if which == 'z1_6' else 5)
    nq = prob.nq
    o = Oracle(prob)
    rng = np.random.default_rng(1)
    B = 12
    q = rng.uniform(prob.lbx[:nq], prob.ubx[:nq], (B, nq))
    qd = rng.uniform(-1, 1, (B, nq)) * prob.ubx[nq:]
    u = rng.uniform(-8, 8, (B, nq))
    xg = np.repeat(np.hstack([q, qd])[:, None, :], 4, axis=1)
    ug = np.repeat(u[:, None, :], 3, axis=1)
    ev = o.eval_nodes(xg, ug, np.zeros((B, 4, 5)))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    grav = np.array(prob.desc.gravity[:], float)
    for b in range(B):
        tau, M, dq, dv = np.zeros(nq), np.zeros((nq, nq)), np.zeros((nq, nq)), np.zeros((nq, nq))
        rc = host_lib.host_rnea_with_derivatives(nq, C.byref(prob.desc.joints), dp(grav), dp(q[b]), dp(qd[b]), dp(u[b]),
                                                 dp(tau), dp(M), dp(dq), dp(dv))
        assert rc == 0
        e = ev[b, 0]
        ref = {'tau': e['tau'][:nq], 'M': e['M'][:nq * nq].reshape(nq, nq), 'dq': e['dtau_dq'][:nq * nq].reshape(nq, nq),
               'dv': e['dtau_dv'][:nq * nq].reshape(nq, nq)}
        for name, got in (('tau', tau), ('M', M), ('dq', dq), ('dv', dv)):
            scale = 1.0 + np.abs(ref[name]).max()
            assert np.abs(got - ref[name]).max() < 1e-10 * scale, (which, b, name, np.abs(got - ref[name]).max())
        assert np.abs(M - M.T).max() < 1e-12 * (1 + np.abs(M).max())
        assert np.all(np.linalg.eigvalsh(M) > 0)
