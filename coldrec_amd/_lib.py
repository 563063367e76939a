"""ctypes binding of libcoldrec_hip.so (the C ABI declared in include/coldrec_hip.h).

There is deliberately NO fallback: if the library is missing or a call fails, a
RuntimeError is raised.  Tensors cross the boundary as raw device pointers
(``tensor.data_ptr()``) plus sizes, the stream as ``torch.cuda.current_stream().cuda_stream``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# CRH_LIB: the measurement build (make -C coldrec_amd/csrc profile), loaded by tools/profile_*.sh only
LIB_PATH = os.environ.get("CRH_LIB") or os.path.join(_HERE, "lib", "libcoldrec_hip.so")
CSRC = os.path.join(_HERE, "csrc")

PAD_IDX = 0x7FFFFFFF
MASKED_SCORE = -1.0e9
MAX_K = 128

_lib = None

_vp, _i32, _i64, _sz, _f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t, ctypes.c_float
_f64 = ctypes.c_double

# name -> (restype, argtypes); kept in one table so tests can check every header symbol is exported
SIGNATURES = {
    "crh_last_error": (ctypes.c_char_p, []),
    "crh_version": (_i32, []),
    "crh_score_topk_supports_dim": (_i32, [_i32]),
    "crh_score_topk_workspace_bytes": (_sz, [_i64, _i64, _i32, _i32]),
    "crh_score_topk_min_workspace_bytes": (_sz, [_i64, _i32]),
    "crh_score_topk_route": (_i32, [_i32, _i64, _i64, _i32, _i32, _sz, _i32, _i32, _vp, _vp]),
    "crh_score_topk_route_kernel": (ctypes.c_char_p, [_i32]),
    "crh_score_topk_f32": (_i32, [_vp, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp, _i32, _i64, _vp, _vp,
                                  _vp, _sz, _vp]),
    "crh_score_topk_f32_ex": (_i32, [_vp, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp, _i32, _i64, _vp, _vp,
                                     _vp, _sz, _vp, _i32, _vp, _vp]),
    "crh_score_topk_f16_supports_dim": (_i32, [_i32]),
    "crh_score_topk_f16_workspace_bytes": (_sz, [_i64, _i64, _i32, _i32]),
    "crh_score_topk_f16_ex": (_i32, [_vp, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp, _i32, _i64, _vp, _vp,
                                     _vp, _sz, _vp, _i32, _vp, _vp]),
    "crh_mask_topk_f32": (_i32, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _i32, _i64, _i32, _vp, _vp, _vp]),
    "crh_merge_topk": (_i32, [_vp, _vp, _i32, _i64, _i32, _i32, _vp, _vp, _vp]),
    "crh_bpr_workspace_bytes": (_sz, [_i64]),
    "crh_bpr_plan_ints": (_i64, [_i64]),
    "crh_bpr_heavy_threshold": (_i32, []),
    "crh_bpr_plan_build_host": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp]),
    "crh_bpr_plan_build": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _vp]),
    "crh_bpr_plan_build_large_workspace_bytes": (_sz, [_i64, _i64]),
    "crh_bpr_plan_build_large": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _sz, _vp]),
    "crh_bpr_fwd_bwd_f32": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _i64, _f32, _vp, _vp, _vp, _vp, _vp, _vp,
                                   _sz, _vp]),
    "crh_bpr_fwd_f32": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _i64, _vp, _vp, _sz, _vp]),
    "crh_bpr_bwd_f32": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _i64, _i64, _f32, _vp, _vp, _vp, _vp, _vp, _vp,
                               _vp, _sz, _vp]),
    "crh_bpr_bwd_owned_f32": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp, _i64, _f32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _sz, _vp]),
    "crh_rows_pack_cap": (_i64, [_i64, _i32]),
    "crh_rows_pack_f32": (_i32, [_vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _vp]),
    "crh_rows_unpack_f32": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "crh_adam_dense_f32": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _f64, _f64, _f64, _f64,
                                  _i64, _i32, _vp, _vp]),
    "crh_adam_rows_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _i64, _i64, _i64, _vp, _f64, _f64, _f64,
                                 _i32, _vp]),
    "crh_adam_step_scalars_host": (None, [_f64, _f64, _f64, _i64, _vp]),
    "crh_adam_step_scalars_range_host": (None, [_f64, _f64, _f64, _i64, _i64, _vp]),
    "crh_spmm_segment_edges": (_i32, []),
    "crh_spmm_lane_group": (_i32, [_i64, _i32, _i64]),
    "crh_spmm_workspace_bytes": (_sz, [_vp, _i32]),
    "crh_spmm_csr_adam_f32": (_i32, [_vp, _vp, _vp, _i64, _vp, _i32, _vp, _f32, _vp, _f32, _vp, _vp, _vp, _vp,
                                     _f64, _f64, _f64, _f64, _i64, _vp, _i32, _vp]),
    "crh_spmm_csr_f32": (_i32, [_vp, _vp, _vp, _i64, _vp, _i32, _vp, _vp, _f32, _vp, _f32, _vp, _vp, _sz, _vp]),
    "crh_bpr_fwd_parts": (_i32, [_i64, _i32]),
    "crh_mf_step_parts": (_i32, [_i64, _i32]),
    "crh_mf_step_tables": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "crh_mf_step_f32": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _i64, _f32, _vp, _vp, _vp, _vp,
                               _vp, _i32, _vp, _vp, _i64, _vp, _f64, _f64, _f64, _vp, _vp]),
    "crh_mf_step_finish": (_i32, [_vp, _i32, _i64, _vp, _vp]),
    "crh_mf_step_sgd_f32": (_i32, [_vp, _vp, _i64, _i64, _i32, _i64, _f32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp,
                                   _i64, _vp, _f64, _vp]),
    "crh_sgd_dense_f32": (_i32, [_vp, _vp, _i64, _f64, _i32, _vp]),
    "crh_sgd_rows_f32": (_i32, [_vp, _vp, _i32, _vp, _i64, _i64, _f64, _vp]),
    "crh_spmm_csr_sgd_f32": (_i32, [_vp, _vp, _vp, _i64, _vp, _i32, _vp, _f32, _vp, _f32, _vp, _vp, _f64, _i32, _vp]),
    "crh_l2_workspace_bytes": (_sz, []),
    "crh_l2_norm_f32": (_i32, [_vp, _i64, _vp, _vp, _sz, _vp]),
    "crh_l2_reg_bwd_f32": (_i32, [_vp, _i64, _i64, _f32, _vp, _vp, _vp, _i32, _vp]),
    "crh_comm_unique_id": (_i32, [_vp]),
    "crh_comm_init": (_vp, [_i32, _i32, _vp]),
    "crh_comm_destroy": (_i32, [_vp]),
    "crh_comm_rank": (_i32, [_vp]),
    "crh_comm_world": (_i32, [_vp]),
    "crh_comm_allreduce_f32": (_i32, [_vp, _vp, _i64, _vp]),
    "crh_comm_allgather_rows": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp]),
    "crh_comm_allgather_topk_workspace_bytes": (_sz, [_i32, _i64, _i32]),
    "crh_comm_allgather_topk": (_i32, [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _sz, _vp]),
    "crh_sampler_create": (_vp, [_vp, _vp, _i64, _i32, _i32]),
    "crh_sampler_destroy": (None, [_vp]),
    "crh_sampler_seed": (_i32, [_vp, ctypes.c_uint32]),
    "crh_sampler_set_state": (_i32, [_vp, _vp, _i32]),
    "crh_sampler_get_state": (_i32, [_vp, _vp, _vp]),
    "crh_sampler_snapshot": (_i32, [_vp]),
    "crh_sampler_restore": (_i32, [_vp]),
    "crh_sampler_epoch": (_i32, [_vp, _i64, _vp, _vp, _vp]),
    "crh_sampler_epoch_async": (_i32, [_vp, _i64, _vp, _vp, _vp, _i32]),
    "crh_sampler_epoch_wait": (_i32, [_vp]),
    "crh_sampler_set_catalogue": (_i32, [_vp, _i32, _vp]),
    "crh_sampler_set_py_state": (_i32, [_vp, _vp, _i32]),
    "crh_sampler_get_py_state": (_i32, [_vp, _vp, _vp]),
    "crh_sampler_min_candidates": (_i64, [_vp]),
    "crh_sampler_epoch_lara": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp]),
    "crh_sampler_epoch_clcrec": (_i32, [_vp, _i32, _i64, _vp, _vp]),
    "crh_sampler_epoch_ccfcrec": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "crh_sampler_epoch_cgrc": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _i64]),
}


class SpmmSched(ctypes.Structure):
    """crh_spmm_sched of include/coldrec_hip.h."""
    _fields_ = [("seg_row", _vp), ("seg_ptr", _vp), ("seg_slot", _vp), ("n_seg", _i64),
                ("multi_row", _vp), ("multi_first", _vp), ("multi_count", _vp), ("n_multi", _i32),
                ("n_partial", _i64), ("nnz", _i64), ("seg_desc", _vp),
                ("slab", _vp), ("slab_lanes", _i32), ("slab_buckets", _i32), ("n_slab", _i64),
                ("slab_first", _i32 * 12), ("slab_units", _i32 * 12), ("slab_base", _i64 * 12), ("version", _i32)]


SPMM_SCHED_VERSION = 4       # CRH_SPMM_SCHED_VERSION of include/coldrec_hip.h


def build(force: bool = False) -> str:
    """Compile every HIP source for gfx950 into coldrec_amd/lib/ (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")) or f == "Makefile"]   # (flags live there)
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "coldrec_hip.h"))
    stale = not os.path.exists(LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-j8", "-C", CSRC, "all"])
    return LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'`). "
                "coldrec_amd has no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().crh_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def ptr(t) -> int:
    """Device pointer of a torch tensor (or None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
