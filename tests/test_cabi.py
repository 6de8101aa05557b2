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
    assert L.smpc_abi_version() == 3


def test_struct_mirrors_match_header_sizes():
    from safe_mpc_amd.problem import Joint, NodeEval, Point, ProblemDesc, Row
    assert C.sizeof(Joint) == 29 * 8
    assert C.sizeof(Point) == 8 + 24
    assert C.sizeof(Row) == 6 * 4 + 10 * 8
    assert C.sizeof(NodeEval) == 8 * (7 + 49 * 3 + 3 + 7 + 49 + 12 + 84 + 1 + 14)
    assert C.sizeof(ProblemDesc) == 12 * 4 + 13 * 8 + 3 * 8 + 2 * 7 * 8 + 4 * 14 * 8 + 7 * C.sizeof(Joint) + \
        12 * C.sizeof(Point) + 12 * C.sizeof(Row)


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
