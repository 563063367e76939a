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
    # headline: fused, lists seeded from an 8 192-item prefix whose 4 GiB score block fits where the packed copy goes afterwards
    # -> partial lists + packed copy (+ lockstep counters, tile bits) + the seed lists, nothing else
    head = L.crh_score_topk_workspace_bytes(131072, 10_000_000, 128, 20)
    assert lists + 10_000_000 * 128 * 4 <= head <= lists + 10_000_032 * 128 * 4 + (4 << 20) + 131072 * 20 * 8 + 256
    # 131 072 x 262 144: seeded with the clamped prefix: 131 072 x 4 096 x 4 B = 2 GiB of stage-1 block, not 8 GiB
    assert L.crh_score_topk_workspace_bytes(131072, 262144, 128, 20) <= 2.1 * GiB
    assert L.crh_score_topk_workspace_bytes(131072, 1_250_000, 128, 20) <= 2.1 * GiB
    # few users: the longer prefixes (32 768 / 65 536 items) stop at 1 GiB of stage-1 block: 16 384 users keep 16 384 items
    GiB1 = 1 << 30
    assert L.crh_score_topk_workspace_bytes(16384, 1_000_000, 128, 20) <= GiB1 + 1_000_032 * 128 * 4 + 64 * 16384 * 20 * 8 + (8 << 20)
    # fp16 (config 5 shape): the user block fills the chip and nothing is cut, but at 16x the MFMA rate the slow-path events
    # are worth removing: a 4 096-item prefix seeds the lists (round 5) -> the prefix's 2 GiB score block, not more
    f16 = L.crh_score_topk_f16_workspace_bytes(131072, 50_000_000, 256, 20)
    assert f16 <= lists + 50_000_032 * 256 * 2 + (32 << 20) + 131072 * 20 * 8 + 1
    # ... and a user block too small for the workgroup kernels keeps the old rule
    assert L.crh_score_topk_f16_workspace_bytes(16384, 50_000_000, 256, 20) <= 64 * 16384 * 20 * 8 + 50_000_032 * 256 * 2 + 2.2 * GiB


def test_ops_refuse_cpu_tensors():
    import torch
    from coldrec_amd import ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.score_topk(torch.zeros(4, 8), None, torch.zeros(9, 8), 2)


def test_the_library_reports_its_route():
    """bench.py and the profiles name the kernel the LIBRARY says a shape takes (crh_score_topk_route = the
    dispatcher's own predicates), instead of re-deriving the dispatcher in Python.  Pinned here, on the CPU: the shapes of the
    bench line."""
    from coldrec_amd import ops
    head = ops.score_topk_route(131072, 10_000_000, 128, 20)                 # the headline: 4-wave workgroups fed by LDS-DMA
    assert head["route"] == "fused-dma" and head["kernel"] == "score_topk_dma_kernel" and head["n_splits"] == 1
    assert head["seeded"] and head["prefix_items"] == 8192                   # (lists seeded from an 8 192-item prefix: round 5)
    shard = ops.score_topk_route(131072, 1_250_000, 128, 20)                 # one rank's shard of the 8-GPU split: the same kernel
    assert shard["route"] == "fused-dma" and shard["seeded"] and shard["prefix_items"] == 4096    # (its flag form below 6 M items)
    assert shard["dma_form"] == "flags" and head["dma_form"] == "barrier" and shard["code"] == head["code"] | 32
    below = ops.score_topk_route(131072, 1_000_000, 128, 20)                 # below the 1.2 M-item gate: per-wave kernel, seeded
    assert below["route"] == "fused-wave" and below["seeded"] and below["prefix_items"] == 4096
    # the reference's default width (main.py:97 --emb_size 64, BASELINE configs[0]): one wave of 128 users per SIMD up to 7.5 M items,
    # the LDS-DMA kernel beyond; lists of up to 28 entries fit beside its four 8 KiB slots
    d64 = ops.score_topk_route(131072, 10_000_000, 64, 20)
    assert d64["route"] == "fused-dma" and d64["kernel"] == "score_topk_dma_kernel" and d64["seeded"] and d64["n_splits"] == 1
    assert ops.score_topk_route(131072, 10_000_000, 64, 28)["route"] == "fused-dma"
    assert ops.score_topk_route(131072, 10_000_000, 64, 29)["route"] == "fused-wave"
    mid64 = ops.score_topk_route(131072, 1_250_000, 64, 20)
    assert mid64["route"] == "fused-wave" and mid64["seeded"] and mid64["prefix_items"] == 4096 and mid64["n_splits"] == 1
    assert ops.score_topk_route(6040, 3706, 64, 20)["route"] == "dense"
    assert below["kernel"] == "score_topk_kernel" and below["code"] != head["code"]
    f16 = ops.score_topk_route(131072, 50_000_000, 256, 20, half=True)       # configs[4]: the LDS-DMA workgroup kernel
    assert f16["route"] == "fused-dma" and f16["kernel"] == "score_topk_dma_kernel" and f16["seeded"] and f16["prefix_items"] == 4096
    f16_shard = ops.score_topk_route(131072, 6_250_000, 256, 20, half=True)
    assert f16_shard["route"] == "fused-dma"
    assert ops.score_topk_route(131072, 50_000_000, 256, 21, half=True)["route"] == "fused-wg"   # k > 20: the lists leave no room for 4 slots
    assert ops.score_topk_route(131072, 10_000_000, 128, 21)["route"] == "fused-wg"
    assert ops.score_topk_route(131072, 50_000_000, 256, 50, half=True)["route"] == "fused-wave"  # ... k = 50: nor for a workgroup's lists
    val = ops.score_topk_route(6040, 3706, 128, 20)                          # a trainer's validation block: dense block + ranking
    assert val["route"] == "dense" and val["n_splits"] == 1
    mid = ops.score_topk_route(8192, 262144, 128, 20)
    assert mid["seeded"] and mid["prefix_items"] == 32768 and mid["route"] == "fused-wave" and mid["n_splits"] > 1
    # a caller that withholds the packed copy stays on the per-wave row-major kernel; a caller that names a split count is not seeded
    assert ops.score_topk_route(131072, 10_000_000, 128, 20, pack=False)["route"] == "fused-wave"
    assert not ops.score_topk_route(8192, 262144, 128, 20, n_splits=4)["seeded"]
    L = _lib.lib()
    assert L.crh_score_topk_route(3, 10, 10, 128, 20, 0, 0, 0, None, None) == -1 and b"elem_bytes" in L.crh_last_error()
