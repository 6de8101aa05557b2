"""The C-ABI library loads without a GPU and exports every symbol include/smpc.h declares; struct mirrors match."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from safe_mpc_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    L = C.CDLL(_lib.LIB_PATH)
    hdr = open(os.path.join(ROOT, 'include', 'smpc.h')).read()
    declared = set(re.findall(r'^(?:int|void|void\*|const char\*)\s+(smpc_[a-z_]+)\s*\(', hdr, re.M))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name
    L.smpc_abi_version.restype = C.c_int
    assert L.smpc_abi_version() == 5


def test_struct_mirrors_match_header_sizes():
    from safe_mpc_amd.problem import Joint, NodeEval, Point, ProblemDesc, Row
    assert C.sizeof(Joint) == 29 * 8
    assert C.sizeof(Point) == 8 + 24
    assert C.sizeof(Row) == 6 * 4 + 10 * 8
    assert C.sizeof(NodeEval) == 8 * (7 + 49 * 3 + 3 + 7 + 49 + 12 + 84 + 1 + 14)
    assert C.sizeof(ProblemDesc) == 14 * 4 + 13 * 8 + 3 * 8 + 2 * 7 * 8 + 4 * 14 * 8 + 7 * C.sizeof(Joint) + \
        12 * C.sizeof(Point) + 12 * C.sizeof(Row)


def test_policy_structs_match_the_header(tmp_path):
    """smpc_policy_params / smpc_policy_state / smpc_loop_state: the ctypes mirrors against sizeof / offsetof from the header itself
    (compiled with gcc), and the policy kinds the controller classes name against the header's enum."""
    import subprocess
    from safe_mpc_amd import _lib
    from safe_mpc_amd import controller as Cn
    src = tmp_path / 'sz.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "smpc.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu %zu %d %d %d %d %d\\n", sizeof(smpc_policy_params), sizeof(smpc_policy_state), '
                   'sizeof(smpc_loop_state), offsetof(smpc_policy_params, tol_x), offsetof(smpc_policy_params, x_min), '
                   'offsetof(smpc_policy_params, stage_lo), SMPC_POLICY_NAIVE, SMPC_POLICY_STATE_CHECK, SMPC_POLICY_STWA, '
                   'SMPC_POLICY_RECEDING, SMPC_POLICY_REAL_RECEDING); return 0; }\n')
    exe = tmp_path / 'sz'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    v = [int(t) for t in subprocess.check_output([str(exe)]).split()]
    P = _lib.PolicyParams
    assert v[:6] == [C.sizeof(P), C.sizeof(_lib.PolicyState), C.sizeof(_lib.LoopState), P.tol_x.offset, P.x_min.offset, P.stage_lo.offset]
    kinds = dict(zip(('naive', 'constraint_everywhere', 'stwa', 'receding', 'real_receding'), v[6:]))
    assert kinds == {k: Cn.CONTROLLERS[k].policy_kind for k in kinds}
    assert Cn.CONTROLLERS['st'].policy_kind == Cn.CONTROLLERS['zerovel'].policy_kind == kinds['naive']
    assert Cn.CONTROLLERS['htwa'].policy_kind == kinds['stwa']


def test_create_fails_loudly_without_gpu():
    """No CPU fallback: on a machine without a HIP device smpc_create must fail, not silently compute elsewhere."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from conftest import make_problem
    from safe_mpc_amd._lib import EngineError
    from safe_mpc_amd.solver import BatchedOcpSolver
    par, prob, net = make_problem('naive', N=5)
    with pytest.raises(EngineError):
        BatchedOcpSolver(prob, net)
