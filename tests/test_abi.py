"""CPU-only: the C-ABI library builds, loads and exports every symbol include/*.h declares.
No compute call is made (there is no GPU here)."""
import ctypes
import os
import re

import pytest

from coldrec_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "coldrec_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(crh_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_header_symbols():
    path = _lib.build()
    assert os.path.exists(path)
    handle = ctypes.CDLL(path)
    syms = _header_symbols()
    assert len(syms) >= 8
    for s in syms:
        assert hasattr(handle, s), f"{s} declared in include/coldrec_hip.h but not exported"
    # the Python binding table covers exactly the header
    assert sorted(_lib.SIGNATURES) == syms


def test_argument_errors_are_reported_without_gpu():
    L = _lib.lib()
    assert L.crh_version() >= 100
    assert L.crh_score_topk_supports_dim(128) == 1 and L.crh_score_topk_supports_dim(100) == 0
    rc = L.crh_merge_topk(None, None, 2, 4, 20, 20, None, None, None)
    assert rc == -1 and b"NULL" in L.crh_last_error()
    with pytest.raises(RuntimeError, match="NULL"):
        _lib.check(rc, "crh_merge_topk")
    assert L.crh_score_topk_min_workspace_bytes(4096, 20) == 64 * 4096 * 20 * 8
    # full workspace = partial lists + the fragment-ordered copy of the shard (whole 32-row tiles)
    assert L.crh_score_topk_workspace_bytes(4096, 10_000_001, 128, 20) >= 64 * 4096 * 20 * 8 + 10_000_032 * 128 * 4


def test_ops_refuse_cpu_tensors():
    import torch
    from coldrec_amd import ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.score_topk(torch.zeros(4, 8), None, torch.zeros(9, 8), 2)
