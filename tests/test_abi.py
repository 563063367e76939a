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


def test_workspace_query_follows_the_route_of_the_call():
    """the full-workspace query charges the seeded route's prefix block only to shapes that take that route,
    with the prefix the dispatcher clamps to (4 096 items below 2 M items, fp32)."""
    L = _lib.lib()
    GiB = float(1 << 30)
    lists = 64 * 131072 * 20 * 8
    # headline: fused, never seeded -> partial lists + packed copy (+ lockstep counters), nothing else
    head = L.crh_score_topk_workspace_bytes(131072, 10_000_000, 128, 20)
    assert lists + 10_000_000 * 128 * 4 <= head <= lists + 10_000_032 * 128 * 4 + (4 << 20)
    # 131 072 x 262 144: seeded with the clamped prefix: 131 072 x 4 096 x 4 B = 2 GiB of stage-1 block, not 8 GiB
    assert L.crh_score_topk_workspace_bytes(131072, 262144, 128, 20) <= 2.1 * GiB
    assert L.crh_score_topk_workspace_bytes(131072, 1_250_000, 128, 20) <= 2.1 * GiB
    # fp16 (config 5 shape): the user block fills the chip, no cuts, no small-catalogue rule -> no seed term
    f16 = L.crh_score_topk_f16_workspace_bytes(131072, 50_000_000, 256, 20)
    assert f16 <= lists + 50_000_032 * 256 * 2 + (16 << 20)


def test_ops_refuse_cpu_tensors():
    import torch
    from coldrec_amd import ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.score_topk(torch.zeros(4, 8), None, torch.zeros(9, 8), 2)
