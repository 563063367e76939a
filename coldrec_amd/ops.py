"""Torch-tensor front ends of the C ABI (include/coldrec_hip.h).  Device memory and streams
come from PyTorch-ROCm; all arithmetic happens in libcoldrec_hip.so.  No CPU fallback."""
from __future__ import annotations

import ctypes
import weakref
import os
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib

SUPPORTED_DIMS = (8, 16, 32, 64, 128, 256)
SUPPORTED_DIMS_F16 = (16, 32, 64, 128, 256)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("coldrec_amd ops need tensors on the MI355X (cuda device); there is no CPU path")


def pad_dim(table: torch.Tensor) -> torch.Tensor:
    """Zero-pad the embedding width to the next width the MFMA kernel is built for.
    Trailing zeros are exact no-ops of the fp32 fma chain (fma(0,0,s) == s, s never -0)."""
    d = table.shape[1]
    dims = SUPPORTED_DIMS_F16 if table.dtype == torch.float16 else SUPPORTED_DIMS
    if d in dims:
        return table
    for w in dims:
        if w > d:
            out = torch.zeros((table.shape[0], w), dtype=table.dtype, device=table.device)
            out[:, :d] = table
            return out
    raise RuntimeError(f"embedding width {d} > 256 is not supported by score_topk")


def make_bitmap(n_items_global: int, masked_ids, device) -> Optional[torch.Tensor]:
    """uint32 words (stored as int32), bit (gi & 31) of word gi >> 5 set => item gi is masked."""
    if masked_ids is None or len(masked_ids) == 0:
        return None
    words = np.zeros((n_items_global + 31) // 32 + 1, dtype=np.uint32)
    ids = np.asarray(masked_ids, dtype=np.int64)
    np.bitwise_or.at(words, ids >> 5, np.uint32(1) << (ids & 31).astype(np.uint32))
    return torch.from_numpy(words.view(np.int32)).to(device)


def rated_csr(rated_lists, device) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    """Per-slot ascending int32 global item ids + int64 row offsets from a list of id arrays."""
    lens = np.fromiter((0 if r is None else len(r) for r in rated_lists), dtype=np.int64,
                       count=len(rated_lists))
    if lens.sum() == 0:
        return None, None
    rowptr = np.zeros(len(rated_lists) + 1, np.int64)
    np.cumsum(lens, out=rowptr[1:])
    col = np.empty(int(rowptr[-1]), np.int32)
    for r, ids in enumerate(rated_lists):
        if lens[r]:
            col[rowptr[r]:rowptr[r + 1]] = np.sort(np.asarray(ids, dtype=np.int64))
    return torch.from_numpy(rowptr).to(device), torch.from_numpy(col).to(device)


_ws_cache = {}


def _workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only scratch, ONE PER (device, stream): calls on different streams never alias, and calls on one stream
    are ordered by it.  Only the eager ops (scoring, dense ranking) use it; everything that may be captured into a
    hipGraph (the engines' BPR steps) owns a private buffer (``bpr_workspace``), because a captured graph keeps the raw
    pointer and a regrown shared buffer would hand that memory back to the caching allocator."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def score_topk(user_emb: torch.Tensor, users: Optional[torch.Tensor], item_emb: torch.Tensor, k: int,
               rated_rowptr: Optional[torch.Tensor] = None, rated_col: Optional[torch.Tensor] = None,
               cand_bitmap: Optional[torch.Tensor] = None, item_base: int = 0, n_splits: int = 0,
               out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, kernel_events=None, pack: bool = True):
    """Fused ``user_emb[users] @ item_emb.T`` -> masks -> top-k (model/MF.py:58-63 +
    model/BaseRecommender.py:175-182).  Returns (scores fp32, global item ids int32), each
    (n_users, k), canonical order.  fp32 tables: exact fp32 MFMA, bit-identical to the oracle's fma chain;
    fp16 tables (both): fp16 MFMA with fp32 accumulation (config 5, generated embeddings).  ``users`` int32 rows of user_emb or None for all rows.
    ``kernel_events``: optional (hipEvent_t, hipEvent_t) raw handles recorded around the scoring
    kernel alone (bench.py's roofline measurement).  ``pack=False`` withholds the workspace for the
    fragment-ordered copy of the item shard (row-major kernel: same results, less memory, ~0.9x speed)."""
    _need_cuda(user_emb, users, item_emb, rated_rowptr, rated_col, cand_bitmap)
    if user_emb.dtype != item_emb.dtype or user_emb.dtype not in (torch.float32, torch.float16):
        raise RuntimeError("score_topk: both tables fp32 (exact, canonical) or both fp16 (fp32 accumulate)")
    half = user_emb.dtype == torch.float16
    if user_emb.shape[1] != item_emb.shape[1]:
        raise RuntimeError("score_topk: user/item embedding widths differ")
    user_emb, item_emb = pad_dim(user_emb.contiguous()), pad_dim(item_emb.contiguous())
    if users is not None:
        users = users.to(torch.int32).contiguous()
    n_users = user_emb.shape[0] if users is None else users.shape[0]
    n_items, d = item_emb.shape
    if rated_rowptr is not None:
        assert rated_rowptr.dtype == torch.int64 and rated_col.dtype == torch.int32
        assert rated_rowptr.shape[0] == n_users + 1
    if cand_bitmap is not None:
        assert cand_bitmap.dtype == torch.int32
    dev = user_emb.device
    if out is None:
        out = (torch.empty((n_users, k), dtype=torch.float32, device=dev),
               torch.empty((n_users, k), dtype=torch.int32, device=dev))
    L = _lib.lib()
    full = L.crh_score_topk_f16_workspace_bytes if half else L.crh_score_topk_workspace_bytes
    ws_bytes = full(n_users, n_items, d, k) if pack else L.crh_score_topk_min_workspace_bytes(n_users, k)
    ws = _workspace(ws_bytes, dev)
    rc = (L.crh_score_topk_f16_ex if half else L.crh_score_topk_f32_ex)(_lib.ptr(user_emb), _lib.ptr(users), n_users, _lib.ptr(item_emb), n_items, d,
                                 _lib.ptr(rated_rowptr), _lib.ptr(rated_col), _lib.ptr(cand_bitmap), k,
                                 item_base, _lib.ptr(out[0]), _lib.ptr(out[1]), _lib.ptr(ws), ws_bytes,
                                 _lib.current_stream(), n_splits,
                                 kernel_events[0] if kernel_events else None,
                                 kernel_events[1] if kernel_events else None)
    _lib.check(rc, "crh_score_topk_f32")
    return out


ROUTE_NAMES = {1: "dense", 2: "fused-wave", 3: "fused-wg", 4: "fused-dma"}


def score_topk_route(n_users: int, n_items: int, d: int, k: int, half: bool = False, has_bitmap: bool = True,
                     n_splits: int = 0, pack: bool = True) -> dict:
    """Which kernels ``score_topk`` runs for a block of this shape: the library's own answer (crh_score_topk_route evaluates
    the dispatcher's predicates under the current environment switches; no GPU needed), with the workspace ``score_topk``
    would pass.  ``{"route", "seeded", "dma_form", "prefix_items", "n_splits", "kernel", "code"}``; ``kernel`` is the scoring kernel's
    name as rocprofv3 prints it."""
    import ctypes
    L = _lib.lib()
    d = d if d % 8 == 0 and (L.crh_score_topk_f16_supports_dim(d) if half else L.crh_score_topk_supports_dim(d)) else \
        next(w for w in (8, 16, 32, 64, 128, 256) if w >= d and (w >= 16 or not half))
    full = L.crh_score_topk_f16_workspace_bytes if half else L.crh_score_topk_workspace_bytes
    ws_bytes = full(n_users, n_items, d, k) if pack else L.crh_score_topk_min_workspace_bytes(n_users, k)
    prefix, splits = ctypes.c_int64(0), ctypes.c_int(0)
    code = L.crh_score_topk_route(2 if half else 4, n_users, n_items, d, k, ws_bytes, 1 if has_bitmap else 0, n_splits,
                                  ctypes.byref(prefix), ctypes.byref(splits))
    _lib.check(code if code < 0 else 0, "crh_score_topk_route")
    return {"route": ROUTE_NAMES[code & 15], "seeded": bool(code & 16), "dma_form": ("flags" if code & 32 else "barrier") if (code & 15) == 4 else None,
            "prefix_items": int(prefix.value),
            "n_splits": int(splits.value), "kernel": L.crh_score_topk_route_kernel(code).decode(), "code": int(code)}


def mask_topk(scores: torch.Tensor, k: int, rated_rowptr=None, rated_col=None, cand_bitmap=None,
              item_base: int = 0, write_back: bool = True):
    """Masks + top-k over a dense (n_users, n_items) fp32 block (model/BaseRecommender.py:175-183)."""
    _need_cuda(scores, rated_rowptr, rated_col, cand_bitmap)
    assert scores.dtype == torch.float32 and scores.dim() == 2 and scores.stride(1) == 1
    n_users, n_items = scores.shape
    dev = scores.device
    out = (torch.empty((n_users, k), dtype=torch.float32, device=dev),
           torch.empty((n_users, k), dtype=torch.int32, device=dev))
    rc = _lib.lib().crh_mask_topk_f32(_lib.ptr(scores), n_users, n_items, scores.stride(0),
                                      _lib.ptr(rated_rowptr), _lib.ptr(rated_col), _lib.ptr(cand_bitmap), k,
                                      item_base, 1 if write_back else 0, _lib.ptr(out[0]), _lib.ptr(out[1]),
                                      _lib.current_stream())
    _lib.check(rc, "crh_mask_topk_f32")
    return out


def merge_topk(scores: torch.Tensor, idx: torch.Tensor, k_out: int):
    """Canonical merge of (n_lists, n_users, k_in) partial lists -> (n_users, k_out)."""
    _need_cuda(scores, idx)
    assert scores.dtype == torch.float32 and idx.dtype == torch.int32
    scores, idx = scores.contiguous(), idx.contiguous()
    n_lists, n_users, k_in = scores.shape
    dev = scores.device
    out = (torch.empty((n_users, k_out), dtype=torch.float32, device=dev),
           torch.empty((n_users, k_out), dtype=torch.int32, device=dev))
    rc = _lib.lib().crh_merge_topk(_lib.ptr(scores), _lib.ptr(idx), n_lists, n_users, k_in, k_out,
                                   _lib.ptr(out[0]), _lib.ptr(out[1]), _lib.current_stream())
    _lib.check(rc, "crh_merge_topk")
    return out


_warmed = set()


def warm_up(device) -> None:
    """Load the library's device code on ``device`` with one tiny launch per source file of the .so (HIP loads a code
    object lazily, at the first launch of any kernel in it: ~0.3 s for this library, which would otherwise land in a
    trainer's first epoch -- where the reference's own timer would count it, main.py:203-205).  The trainers call this
    when they are constructed, like the reference moves its model to the device before its timer starts
    (model/MF.py:13-16).  Idempotent per device; nothing is computed that anyone reads."""
    device = torch.device(device)
    if device.type != 'cuda' or (device.index, ) in _warmed:
        return
    _warmed.add((device.index, ))
    with torch.cuda.device(device):
        f32 = dict(dtype=torch.float32, device=device)
        u, v = torch.zeros((64, 8), **f32), torch.zeros((96, 8), **f32)
        score_topk(u, None, v, 4)                                            # score_topk.hip (+ merge: n_splits)
        score_topk(u, None, v, 4, n_splits=2)                                # merge_topk.hip
        mask_topk(torch.zeros((4, 64), **f32), 4, write_back=False)
        idx = torch.zeros(4, dtype=torch.int32, device=device)
        g, gv = torch.zeros_like(u), torch.zeros_like(v)
        bpr_fwd_bwd(u, v, v, idx, idx, idx, 0.0, g, gv, gv, plan=build_plans_device(idx, idx, idx, 4)[0])   # bpr_adam.hip
        adam_dense(u.clone(), g, torch.zeros_like(u), torch.zeros_like(u), 1)
        l2_norm(u)                                                           # l2_reg.hip
        rp = torch.arange(0, 65, dtype=torch.int64, device=device).clamp_(max=1)
        spmm_csr(rp, torch.zeros(1, dtype=torch.int32, device=device), torch.ones(1, **f32), u, y=torch.empty_like(u))
        # ... and the handful of ATen kernels the trainers' epoch loop touches (PyTorch loads its code lazily too): the
        # membership gather of the validation metrics, the loss read-back, the epoch upload
        dense = torch.zeros((4, 8), dtype=torch.bool, device=device)
        ids = torch.zeros((4, 3), dtype=torch.int32, device=device).long()
        ok = (ids >= 0) & (ids < 8)
        (dense[torch.arange(4, device=device).unsqueeze(1), ids.clamp(0, 7)] & ok).cpu()
        torch.zeros((3, 2), **f32).sum(dim=1).cpu()
        torch.empty(4, dtype=torch.int32, device=device).copy_(torch.zeros(4, dtype=torch.int32).pin_memory(), non_blocking=True)
        torch.cuda.synchronize(device)


# ------------------------------------------------------------------------------ training ops
def bpr_fwd_bwd(user_table, pos_table, neg_table, user_idx, pos_idx, neg_idx, reg: float,
                grad_user=None, grad_pos=None, grad_neg=None, loss_out: Optional[torch.Tensor] = None,
                plan: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None):
    """bpr_loss + l2_reg_loss and their dense table gradients for one batch of triples
    (util/utils.py:25-29,44-48 + autograd at model/MF.py:22-26).  ``*_idx`` int32 device tensors or
    None (tables are already gathered).  Gradients are ACCUMULATED into ``grad_*`` (all three or
    none).  Returns ``loss_out`` = device tensor [bpr, l2]."""
    _need_cuda(user_table, pos_table, neg_table, user_idx, pos_idx, neg_idx, grad_user, grad_pos, grad_neg)
    d = user_table.shape[1]
    batch = user_idx.shape[0] if user_idx is not None else user_table.shape[0]
    for t in (user_table, pos_table, neg_table):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.shape[1] == d
    for ix in (user_idx, pos_idx, neg_idx):
        assert ix is None or (ix.dtype == torch.int32 and ix.is_contiguous() and ix.shape[0] == batch)
    dev = user_table.device
    if loss_out is None:
        loss_out = torch.empty(2, dtype=torch.float32, device=dev)
    L = _lib.lib()
    need = L.crh_bpr_workspace_bytes(batch)
    ws = workspace if workspace is not None else _workspace(need, dev)
    assert ws.numel() >= need, "bpr_fwd_bwd: private workspace too small for this batch"
    rc = L.crh_bpr_fwd_bwd_f32(_lib.ptr(user_table), _lib.ptr(pos_table), _lib.ptr(neg_table), d,
                               _lib.ptr(user_idx), _lib.ptr(pos_idx), _lib.ptr(neg_idx), batch, float(reg),
                               _lib.ptr(grad_user), _lib.ptr(grad_pos), _lib.ptr(grad_neg), _lib.ptr(loss_out),
                               _lib.ptr(plan), _lib.ptr(ws), ws.numel(), _lib.current_stream())
    _lib.check(rc, "crh_bpr_fwd_bwd_f32")
    return loss_out


def _bpr_checks(user_table, pos_table, neg_table, user_idx, pos_idx, neg_idx):
    d = user_table.shape[1]
    batch = user_idx.shape[0] if user_idx is not None else user_table.shape[0]
    for t in (user_table, pos_table, neg_table):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.shape[1] == d
    for ix in (user_idx, pos_idx, neg_idx):
        assert ix is None or (ix.dtype == torch.int32 and ix.is_contiguous() and ix.shape[0] == batch)
    return d, batch


def bpr_fwd(user_table, pos_table, neg_table, user_idx, pos_idx, neg_idx, sums_out: torch.Tensor,
            workspace: torch.Tensor):
    """Forward of one rank's slice of the batch (data-parallel split): ``sums_out`` (4 floats, device)
    = [sum u^2, sum p^2, sum n^2, sum -log(1e-5+sigmoid(x))] of the slice.  ``workspace``: uint8 tensor of
    ``bpr_workspace_bytes(batch)`` that must be handed to ``bpr_bwd`` unchanged."""
    _need_cuda(user_table, pos_table, neg_table, user_idx, pos_idx, neg_idx, sums_out, workspace)
    d, batch = _bpr_checks(user_table, pos_table, neg_table, user_idx, pos_idx, neg_idx)
    assert sums_out.dtype == torch.float32 and sums_out.numel() >= 4
    rc = _lib.lib().crh_bpr_fwd_f32(_lib.ptr(user_table), _lib.ptr(pos_table), _lib.ptr(neg_table), d,
                                    _lib.ptr(user_idx), _lib.ptr(pos_idx), _lib.ptr(neg_idx), batch,
                                    _lib.ptr(sums_out), _lib.ptr(workspace), workspace.numel(),
                                    _lib.current_stream())
    _lib.check(rc, "crh_bpr_fwd_f32")
    return sums_out


def bpr_bwd(user_table, pos_table, neg_table, user_idx, pos_idx, neg_idx, global_batch: int, reg: float,
            sums: torch.Tensor, grad_user, grad_pos, grad_neg, loss_out: Optional[torch.Tensor],
            workspace: torch.Tensor, plan: Optional[torch.Tensor] = None):
    """Backward of one rank's slice with the all-reduced ``sums`` and the global batch size."""
    _need_cuda(user_table, pos_table, neg_table, user_idx, pos_idx, neg_idx, sums, grad_user, grad_pos, grad_neg,
               workspace)
    d, batch = _bpr_checks(user_table, pos_table, neg_table, user_idx, pos_idx, neg_idx)
    rc = _lib.lib().crh_bpr_bwd_f32(_lib.ptr(user_table), _lib.ptr(pos_table), _lib.ptr(neg_table), d,
                                    _lib.ptr(user_idx), _lib.ptr(pos_idx), _lib.ptr(neg_idx), batch,
                                    int(global_batch), float(reg), _lib.ptr(sums), _lib.ptr(grad_user),
                                    _lib.ptr(grad_pos), _lib.ptr(grad_neg), _lib.ptr(loss_out), _lib.ptr(plan),
                                    _lib.ptr(workspace), workspace.numel(), _lib.current_stream())
    _lib.check(rc, "crh_bpr_bwd_f32")
    return loss_out


def bpr_bwd_owned(user_table, item_table, user_idx, pos_idx, neg_idx, reg: float, sums: torch.Tensor, grad_user, grad_item,
                  loss_out: Optional[torch.Tensor], workspace: torch.Tensor, plan: torch.Tensor, own_mod: int, own_rem: int):
    """Deterministic backward over the plan's row slots ``w % own_mod == own_rem`` only (crh_bpr_bwd_owned_f32): the
    row-ownership split of the data-parallel touched-rows step; ``workspace`` carries a forward over the WHOLE batch."""
    _need_cuda(user_table, item_table, user_idx, pos_idx, neg_idx, sums, grad_user, grad_item, workspace, plan)
    d, batch = _bpr_checks(user_table, item_table, item_table, user_idx, pos_idx, neg_idx)
    rc = _lib.lib().crh_bpr_bwd_owned_f32(_lib.ptr(user_table), _lib.ptr(item_table), d, _lib.ptr(user_idx), _lib.ptr(pos_idx),
                                          _lib.ptr(neg_idx), batch, float(reg), _lib.ptr(sums), _lib.ptr(grad_user),
                                          _lib.ptr(grad_item), _lib.ptr(loss_out), _lib.ptr(plan), int(own_mod), int(own_rem),
                                          _lib.ptr(workspace), workspace.numel(), _lib.current_stream())
    _lib.check(rc, "crh_bpr_bwd_owned_f32")
    return loss_out


def rows_pack_cap(batch: int, own_mod: int) -> int:
    return int(_lib.lib().crh_rows_pack_cap(int(batch), int(own_mod)))


def rows_pack(table, plan, batch: int, user_rows: int, own_mod: int, own_rem: int, out_ids, out_rows) -> None:
    """(row id, row) of every plan slot this rank owns, from ``table`` (crh_rows_pack_f32); ids past the plan's rows are -1."""
    _need_cuda(table, plan, out_ids, out_rows)
    assert out_ids.dtype == torch.int32 and out_rows.dtype == torch.float32 and out_rows.is_contiguous() and table.is_contiguous()
    assert out_ids.numel() >= rows_pack_cap(batch, own_mod) and out_rows.shape[0] >= out_ids.numel()
    _lib.check(_lib.lib().crh_rows_pack_f32(_lib.ptr(table), _lib.ptr(plan), int(batch), int(user_rows), table.shape[1],
                                            int(own_mod), int(own_rem), _lib.ptr(out_ids), _lib.ptr(out_rows),
                                            _lib.current_stream()), "crh_rows_pack_f32")


def rows_unpack(table, ids, rows) -> None:
    """table[ids[e]] = rows[e] for ids[e] >= 0 (crh_rows_unpack_f32)."""
    _need_cuda(table, ids, rows)
    assert ids.dtype == torch.int32 and rows.dtype == torch.float32 and rows.is_contiguous() and ids.is_contiguous()
    _lib.check(_lib.lib().crh_rows_unpack_f32(_lib.ptr(table), _lib.ptr(ids), _lib.ptr(rows), ids.numel(), table.shape[1],
                                              _lib.current_stream()), "crh_rows_unpack_f32")


def bpr_workspace(batch: int, device) -> torch.Tensor:
    """A private scratch buffer for a bpr_fwd / bpr_bwd pair (the shared grow-only workspace may be
    reused by other ops between the two calls)."""
    return torch.empty(int(_lib.lib().crh_bpr_workspace_bytes(int(batch))), dtype=torch.uint8, device=device)


def build_plans(user_idx, pos_idx, neg_idx, batch_size: int) -> np.ndarray:
    """Host: reverse indices ("plans") for every batch of an epoch of triples (int32 numpy arrays as the
    sampler returns them).  Row b of the result is the plan of batch b (last batch: its own size)."""
    L = _lib.lib()
    u, p, n = (np.ascontiguousarray(x, dtype=np.int32) for x in (user_idx, pos_idx, neg_idx))
    n_rec = u.shape[0]
    n_batches = (n_rec + batch_size - 1) // batch_size
    stride = int(L.crh_bpr_plan_ints(batch_size))
    out = np.zeros((n_batches, stride), np.int32)
    for b in range(n_batches):
        lo, hi = b * batch_size, min((b + 1) * batch_size, n_rec)
        rc = L.crh_bpr_plan_build_host(u[lo:hi].ctypes.data, p[lo:hi].ctypes.data, n[lo:hi].ctypes.data,
                                       hi - lo, batch_size, out[b].ctypes.data)
        _lib.check(rc, "crh_bpr_plan_build_host")
    return out


def plan_shape(n_records: int, batch_size: int):
    """(n_batches, ints per plan) of ``build_plans_device``'s result."""
    return (n_records + batch_size - 1) // batch_size, int(_lib.lib().crh_bpr_plan_ints(int(batch_size)))


def build_plans_device(user_idx: torch.Tensor, pos_idx: torch.Tensor, neg_idx: torch.Tensor,
                       batch_size: int, lds_max_batch: int = 8192, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Plans of every batch of an epoch, (n_batches, stride) int32 on the GPU: one launch of crh_bpr_plan_build (LDS
    sort per batch) for batch_size <= 8192, crh_bpr_plan_build_large (several workgroups per batch) above that.
    ``lds_max_batch`` below 8192 sends smaller batches through the large builder too (tests); ``out``: build in place."""
    _need_cuda(user_idx, pos_idx, neg_idx)
    L = _lib.lib()
    n_rec = user_idx.numel()
    nb = (n_rec + batch_size - 1) // batch_size
    stride = int(L.crh_bpr_plan_ints(batch_size))
    u, p, n = (x.to(torch.int32).contiguous() for x in (user_idx, pos_idx, neg_idx))
    if out is not None:
        assert out.shape == (nb, stride) and out.dtype == torch.int32 and out.is_contiguous()
    if batch_size <= min(8192, lds_max_batch):
        plans = out if out is not None else torch.empty((nb, stride), dtype=torch.int32, device=user_idx.device)
        _lib.check(L.crh_bpr_plan_build(_lib.ptr(u), _lib.ptr(p), _lib.ptr(n), n_rec, int(batch_size),
                                        _lib.ptr(plans), _lib.current_stream()), "crh_bpr_plan_build")
        return plans
    plans = out if out is not None else torch.empty((nb, stride), dtype=torch.int32, device=user_idx.device)
    group = max(1, min(nb, (256 << 20) // (48 * int(batch_size) + 4096)))     # batches per call: workspace <= ~256 MB
    for b0 in range(0, nb, group):
        b1 = min(nb, b0 + group)
        lo, hi = b0 * batch_size, min(n_rec, b1 * batch_size)
        ws_bytes = int(L.crh_bpr_plan_build_large_workspace_bytes(hi - lo, int(batch_size)))
        ws = _workspace(ws_bytes, user_idx.device)
        _lib.check(L.crh_bpr_plan_build_large(_lib.ptr(u[lo:hi]), _lib.ptr(p[lo:hi]), _lib.ptr(n[lo:hi]), hi - lo,
                                              int(batch_size), _lib.ptr(plans[b0:b1]), _lib.ptr(ws), ws_bytes,
                                              _lib.current_stream()), "crh_bpr_plan_build_large")
    return plans


# ----------------------------------------------------------------------------- MF step in one launch
def mf_step_parts(n_rows: int, d: int) -> int:
    return int(_lib.lib().crh_mf_step_parts(int(n_rows), int(d)))


def bpr_fwd_parts(batch: int, d: int) -> int:
    return int(_lib.lib().crh_bpr_fwd_parts(int(batch), int(d)))


def mf_step_tables(plans: torch.Tensor, user_idx, pos_idx, neg_idx, batch_size: int, user_rows: int, item_rows: int,
                   out=None):
    """Flattened per-epoch tables of crh_mf_step_f32: (range (nb, rows, 2), mult (nb, rows), entries (nb, 3B, 2))."""
    _need_cuda(plans, user_idx, pos_idx, neg_idx)
    nb, R, n_rec = plans.shape[0], int(user_rows) + int(item_rows), user_idx.numel()
    assert nb == (n_rec + batch_size - 1) // batch_size
    if out is None:
        i32 = dict(dtype=torch.int32, device=plans.device)
        out = (torch.empty((nb, R, 2), **i32), torch.empty((nb, R), **i32), torch.zeros((nb, 3 * batch_size, 2), **i32))
    rng, mult, ent = out
    assert rng.shape == (nb, R, 2) and mult.shape == (nb, R) and ent.shape == (nb, 3 * batch_size, 2)
    _lib.check(_lib.lib().crh_mf_step_tables(_lib.ptr(plans), _lib.ptr(user_idx), _lib.ptr(pos_idx), _lib.ptr(neg_idx),
                                             n_rec, int(batch_size), int(user_rows), int(item_rows), _lib.ptr(rng),
                                             _lib.ptr(mult), _lib.ptr(ent), _lib.current_stream()),
               "crh_mf_step_tables")
    return out


def mf_step(table_in, table_out, m, v, user_rows: int, batch: int, reg: float, plan, rng, entries, mult_next,
            part_in, n_parts_in: int, part_out, loss_prev, batch_prev: int, loss_out, step_scalars,
            beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, sgd_lr: Optional[float] = None) -> None:
    """One whole optimiser step of model/MF.py:19-27 in one launch (crh_mf_step_f32, see include/coldrec_hip.h);
    ``sgd_lr`` selects torch.optim.SGD(lr) instead of Adam (crh_mf_step_sgd_f32: m, v, step_scalars unused)."""
    _need_cuda(table_in, table_out, m, v, plan, rng, entries, part_in, part_out, step_scalars)
    R, d = table_in.shape
    for t in (table_in, table_out) + (() if sgd_lr is not None else (m, v)):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.shape == (R, d)
    for t in (plan, rng, entries):
        assert t.dtype == torch.int32 and t.is_contiguous()
    assert rng.shape == (R, 2) and (mult_next is None or (mult_next.shape == (R,) and mult_next.is_contiguous()))
    if sgd_lr is not None:
        rc = _lib.lib().crh_mf_step_sgd_f32(
            _lib.ptr(table_in), _lib.ptr(table_out), int(user_rows), R - int(user_rows), d, int(batch), float(reg),
            _lib.ptr(plan), _lib.ptr(rng), _lib.ptr(entries), _lib.ptr(mult_next), _lib.ptr(part_in), int(n_parts_in),
            _lib.ptr(part_out), _lib.ptr(loss_prev), int(batch_prev), _lib.ptr(loss_out), float(sgd_lr),
            _lib.current_stream())
        _lib.check(rc, "crh_mf_step_sgd_f32")
        return
    rc = _lib.lib().crh_mf_step_f32(
        _lib.ptr(table_in), _lib.ptr(table_out), _lib.ptr(m), _lib.ptr(v), int(user_rows), R - int(user_rows), d,
        int(batch), float(reg), _lib.ptr(plan), _lib.ptr(rng), _lib.ptr(entries), _lib.ptr(mult_next),
        _lib.ptr(part_in), int(n_parts_in), _lib.ptr(part_out), _lib.ptr(loss_prev), int(batch_prev),
        _lib.ptr(loss_out), beta1, beta2, eps, _lib.ptr(step_scalars), _lib.current_stream())
    _lib.check(rc, "crh_mf_step_f32")


def mf_step_finish(part_in, n_parts_in: int, batch: int, loss_out) -> None:
    _need_cuda(part_in, loss_out)
    _lib.check(_lib.lib().crh_mf_step_finish(_lib.ptr(part_in), int(n_parts_in), int(batch), _lib.ptr(loss_out),
                                             _lib.current_stream()), "crh_mf_step_finish")


def adam_step_scalars(first_step: int, n_steps: int, lr: float = 1e-3, betas=(0.9, 0.999)) -> np.ndarray:
    """(n_steps, 2) float32 host array of the step-dependent Adam factors for steps first_step.."""
    out = np.empty((n_steps, 2), np.float32)
    _lib.lib().crh_adam_step_scalars_range_host(float(lr), float(betas[0]), float(betas[1]), int(first_step),
                                                int(n_steps), out.ctypes.data)
    return out


def adam_dense(p, g, m, v, step: int, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
               zero_grad: bool = True, second=None, step_scalars: Optional[torch.Tensor] = None):
    """In-place dense Adam on (p, g, m, v) [and ``second`` = another such 4-tuple] -- model/MF.py:14,27."""
    ts = (p, g, m, v) + (tuple(second) if second else ())
    _need_cuda(*ts)
    for t in ts:
        assert t.dtype == torch.float32 and t.is_contiguous()
    n0 = p.numel()
    s = second if second else (None, None, None, None)
    n1 = s[0].numel() if second else 0
    rc = _lib.lib().crh_adam_dense_f32(_lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), n0,
                                       _lib.ptr(s[0]), _lib.ptr(s[1]), _lib.ptr(s[2]), _lib.ptr(s[3]), n1,
                                       float(lr), float(betas[0]), float(betas[1]), float(eps), int(step),
                                       1 if zero_grad else 0, _lib.ptr(step_scalars), _lib.current_stream())
    _lib.check(rc, "crh_adam_dense_f32")


def adam_rows(p, g, m, v, last_step, plan, batch: int, user_rows: int, step: int, scalar_table, mode: int,
              betas=(0.9, 0.999), eps: float = 1e-8):
    """Touched-rows replay of dense Adam (crh_adam_rows_f32): mode 0 catch-up / 1 step / 2 flush."""
    _need_cuda(p, g, m, v, last_step, plan, scalar_table)
    assert last_step.dtype == torch.int32 and scalar_table.dtype == torch.float32
    assert scalar_table.numel() >= 2 * (step + 1), "scalar table too short for this step"
    rc = _lib.lib().crh_adam_rows_f32(_lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), _lib.ptr(last_step),
                                      p.shape[0], p.shape[1], _lib.ptr(plan), int(batch), int(user_rows), int(step),
                                      _lib.ptr(scalar_table), float(betas[0]), float(betas[1]), float(eps), int(mode),
                                      _lib.current_stream())
    _lib.check(rc, "crh_adam_rows_f32")


def sgd_dense(p, g, lr: float, zero_grad: bool = True):
    """In-place torch.optim.SGD(lr) defaults on a dense tensor: p <- fma(-lr, g, p) (crh_sgd_dense_f32)."""
    _need_cuda(p, g)
    assert p.dtype == torch.float32 and g.dtype == torch.float32 and p.is_contiguous() and g.is_contiguous()
    _lib.check(_lib.lib().crh_sgd_dense_f32(_lib.ptr(p), _lib.ptr(g), p.numel(), float(lr), 1 if zero_grad else 0,
                                            _lib.current_stream()), "crh_sgd_dense_f32")


def sgd_rows(p, g, plan, batch: int, user_rows: int, lr: float):
    """The same update on the rows of one batch's plan only (crh_sgd_rows_f32); clears the consumed gradient rows."""
    _need_cuda(p, g, plan)
    assert p.dtype == torch.float32 and p.is_contiguous() and g.is_contiguous() and plan.dtype == torch.int32
    _lib.check(_lib.lib().crh_sgd_rows_f32(_lib.ptr(p), _lib.ptr(g), p.shape[1], _lib.ptr(plan), int(batch),
                                           int(user_rows), float(lr), _lib.current_stream()), "crh_sgd_rows_f32")


def spmm_csr_sgd(rowptr, col, val, x, acc_in, s_in: float, acc_out, s_out: float, sched, p, lr: float,
                 zero_acc_in: bool = False) -> None:
    """g = (acc_in*s_in + A @ x)*s_out -> p <- fma(-lr, g, p) in the SpMM's epilogue (crh_spmm_csr_sgd_f32)."""
    _need_cuda(rowptr, col, val, x, acc_in, acc_out, p)
    n_rows, d = rowptr.shape[0] - 1, x.shape[1]
    assert x.dtype == torch.float32 and x.is_contiguous()          # x may hold more rows than A (row-sharded propagation)
    assert p.dtype == torch.float32 and p.is_contiguous() and p.shape == (n_rows, d)
    rc = _lib.lib().crh_spmm_csr_sgd_f32(_lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(val), n_rows, _lib.ptr(x), d,
                                         _lib.ptr(acc_in), float(s_in), _lib.ptr(acc_out), float(s_out),
                                         ctypes.byref(sched.for_launch(n_rows, d, col, val)) if sched is not None else None, _lib.ptr(p), float(lr),
                                         int(bool(zero_acc_in)), _lib.current_stream())
    _lib.check(rc, "crh_spmm_csr_sgd_f32")


def l2_norm(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """|x|_F as a device scalar (crh_l2_norm_f32; deterministic two-stage reduction)."""
    _need_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    L = _lib.lib()
    if out is None:
        out = torch.empty(1, dtype=torch.float32, device=x.device)
    ws = torch.empty(int(L.crh_l2_workspace_bytes()), dtype=torch.uint8, device=x.device)
    _lib.check(L.crh_l2_norm_f32(_lib.ptr(x), x.numel(), _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.current_stream()),
               "crh_l2_norm_f32")
    return out


def l2_reg_bwd(x: torch.Tensor, reg: float, norm: torch.Tensor, grad_out: Optional[torch.Tensor]) -> torch.Tensor:
    """d(reg * |x|_F / rows)/dx * grad_out (crh_l2_reg_bwd_f32)."""
    _need_cuda(x, norm, grad_out)
    gx = torch.empty_like(x)
    go = None if grad_out is None else grad_out.reshape(1).float().contiguous()
    _lib.check(_lib.lib().crh_l2_reg_bwd_f32(_lib.ptr(x), x.numel(), x.shape[0], float(reg), _lib.ptr(norm), _lib.ptr(go),
                                             _lib.ptr(gx), 0, _lib.current_stream()), "crh_l2_reg_bwd_f32")
    return gx


def spmm_csr_adam(rowptr, col, val, x, acc_in, s_in: float, acc_out, s_out: float, sched, p, m, v, step: int,
                  lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, step_scalars=None,
                  zero_acc_in: bool = False) -> None:
    """g = (acc_in*s_in + A @ x)*s_out -> one Adam step on (p, m, v) in the SpMM's epilogue (crh_spmm_csr_adam_f32);
    ``zero_acc_in`` clears acc_in row by row once consumed.  Same bits as spmm_csr(acc_out=g) + adam_dense."""
    _need_cuda(rowptr, col, val, x, acc_in, acc_out, p, m, v, step_scalars)
    assert rowptr.dtype == torch.int64 and col.dtype == torch.int32 and val.dtype == torch.float32
    n_rows, d = rowptr.shape[0] - 1, x.shape[1]
    assert x.dtype == torch.float32 and x.is_contiguous()          # x may hold more rows than A (row-sharded propagation)
    for t in (p, m, v):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.shape == (n_rows, d)
    rc = _lib.lib().crh_spmm_csr_adam_f32(_lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(val), n_rows, _lib.ptr(x), d,
                                          _lib.ptr(acc_in), float(s_in), _lib.ptr(acc_out), float(s_out),
                                          ctypes.byref(sched.for_launch(n_rows, d, col, val)) if sched is not None else None, _lib.ptr(p),
                                          _lib.ptr(m), _lib.ptr(v), float(lr), float(betas[0]), float(betas[1]),
                                          float(eps), int(step), _lib.ptr(step_scalars), int(bool(zero_acc_in)),
                                          _lib.current_stream())
    _lib.check(rc, "crh_spmm_csr_adam_f32")


class SpmmSchedule:
    """Load-balancing schedule of one CSR matrix (see crh_spmm_sched), built once per graph on the host with numpy.
    Work items are ROWS, ordered by descending length: the lane groups of a wave (8 rows at d=128 with 4 column
    slices) then walk rows of about the same length instead of waiting for the longest of a Zipf-shaped sample, and
    the longest rows start first.  Rows with more than crh_spmm_segment_edges() edges are "heavy" (a workgroup
    each); they sit at the end of the list with seg_slot >= 0 and the light path skips them.  Rows are independent,
    so the order does not change any result."""

    SLAB_MAX_EDGES = 1 << 26          # the record stream is built for graphs up to this many stored edges
    GIANT = 1024                      # heavy rows above this many edges are cut into 2 (above 4x: 4) column ranges; 0 = never
    SLAB_BUCKETS = 12                 # CRH_SPMM_SLAB_BUCKETS of include/coldrec_hip.h

    def __init__(self, rowptr, device, seg: Optional[int] = None, col=None, val=None):
        """``col`` / ``val`` (the matrix's edge arrays, host or device): with them the schedule can also lay the light rows
        out as a record stream (crh_spmm_sched::slab) for the lane-group width of a launch -- see ``for_launch``."""
        rp = rowptr.cpu().numpy() if torch.is_tensor(rowptr) else np.asarray(rowptr)
        rp = rp.astype(np.int64)
        self._rp, self._col, self._val, self._device, self._slabs = rp, col, val, device, {}
        self._checked, self._baked = None, None
        deg = np.diff(rp)
        if seg is None:
            # heavy threshold: crh_spmm_segment_edges() (64) for sparse graphs; for dense ones (MovieLens shape: mean
            # degree 133, nine rows in ten above 64) a workgroup per row is the costlier path: 256 there
            # (SpMM 59.0 -> 54.7 us; CiteULike shape, mean 11.5: 20 us at 64, 32 us at 256)
            seg = int(_lib.lib().crh_spmm_segment_edges())
            if "CRH_SPMM_SEG" not in os.environ and len(deg) and float(deg.mean()) > 48.0:
                seg *= 4
        self.seg = int(seg)
        heavy = deg > seg
        order = np.argsort(-deg, kind="stable")
        order = np.concatenate([order[~heavy[order]], order[heavy[order]]])
        seg_row = order.astype(np.int32)
        seg_slot = np.where(heavy[order], 0, -1).astype(np.int32)
        seg_ptr = np.zeros(len(seg_row) + 1, np.int64)              # not read any more (kept for the ABI struct)
        # heavy rows: one entry per heavy BLOCK, longest rows first (they lead the grid: the longest chains start with the
        # first wave of workgroups).  Rows above GIANT edges are cut into 2 (4 above 4 * GIANT) column ranges of the slice,
        # one workgroup each with half (a quarter) of the lanes per lane group: multi_count = n_sub | sub << 8.
        giant, giant4 = self.GIANT, 4 * self.GIANT
        hrows = np.nonzero(heavy)[0]
        hrows = hrows[np.argsort(-deg[hrows], kind="stable")]
        n_sub = np.where(deg[hrows] > giant4, 4, np.where(deg[hrows] > giant, 2, 1)) if giant > 0 else np.ones(len(hrows), np.int64)
        multi_row = np.repeat(hrows, n_sub).astype(np.int32)
        sub = (np.arange(len(multi_row)) - np.repeat(np.cumsum(n_sub) - n_sub, n_sub)).astype(np.int64)
        multi_count = (np.repeat(n_sub, n_sub) | (sub << 8)).astype(np.int32)
        assert np.isin(multi_count & 0xff, (1, 2, 4)).all() and ((multi_count >> 8) < (multi_count & 0xff)).all()
        multi_first = np.zeros(len(multi_row) + 1, np.int64)          # not read any more (kept for the ABI struct)
        # one 16-byte descriptor per work item: {row, first edge, edges, slot}
        desc = None
        if len(deg) and int(rp[-1]) < (1 << 31):
            desc = np.stack([seg_row, rp[:-1][order].astype(np.int32), deg[order].astype(np.int32), seg_slot], 1)
            desc = np.ascontiguousarray(desc.astype(np.int32).reshape(-1))
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        self.t = (to(seg_row), to(seg_ptr), to(seg_slot), to(multi_row), to(multi_first[:-1].astype(np.int32)),
                  to(multi_count))
        self.desc = to(desc) if desc is not None else None
        self.c = _lib.SpmmSched(_lib.ptr(self.t[0]), _lib.ptr(self.t[1]), _lib.ptr(self.t[2]), len(seg_row),
                                _lib.ptr(self.t[3]) if len(multi_row) else None,
                                _lib.ptr(self.t[4]) if len(multi_row) else None,
                                _lib.ptr(self.t[5]) if len(multi_row) else None, len(multi_row),
                                int(multi_first[-1]), int(rp[-1]), _lib.ptr(self.desc))
        self.c.version = _lib.SPMM_SCHED_VERSION
        self.n_partial = int(multi_first[-1])
        self.n_seg = len(seg_row)
        self._ws = {}
        self._light = order[~heavy[order]].astype(np.int64)        # light rows in work-item order (descending length)

    def _build_slab(self, G: int):
        """The light rows as a stream of (col, val-bits) pairs for lane groups of G lanes: record = header pair {row, cnt}
        + the edges in order, padded to whole units of G pairs; rows are in descending length, so equal unit counts are
        contiguous buckets.  Returns (stream tensor, first, units, base) or None when it does not apply."""
        if self._col is None or self._val is None or G != 8 or int(self._rp[-1]) > self.SLAB_MAX_EDGES:    # (kernel: G == 8 only)
            return None
        rows = self._light
        deg = (self._rp[rows + 1] - self._rp[rows]).astype(np.int64)
        units = np.where(deg <= G - 1, 1, 1 + (deg - (G - 1) + G - 1) // G)
        cuts = np.nonzero(np.diff(units))[0] + 1 if len(units) else np.zeros(0, np.int64)
        first = np.concatenate([[0], cuts]).astype(np.int64) if len(units) else np.zeros(0, np.int64)
        if len(first) > self.SLAB_BUCKETS or len(rows) >= (1 << 31) - 1:
            return None
        start = np.zeros(len(rows) + 1, np.int64)
        np.cumsum(units * G, out=start[1:])
        n_pairs = int(start[-1]) + 2 * G                                   # tail: a one-unit record's second-unit read
        col = self._col.cpu().numpy() if torch.is_tensor(self._col) else np.asarray(self._col)
        val = self._val.cpu().numpy() if torch.is_tensor(self._val) else np.asarray(self._val)
        # what exactly is being baked in: version counters of the source tensors and two checksums of the host copies taken
        # NOW (a later in-place ``val.mul_(c)`` must not pass for "the same edges" because the pointer is the same).
        # Position-weighted (sum of (i + 1) x_i in int64, wrapping on both sides): a permutation of the edges does not pass either
        wts = np.arange(1, len(col) + 1, dtype=np.int64)
        with np.errstate(over="ignore"):
            self._baked = (self._col._version if torch.is_tensor(self._col) else None,
                           self._val._version if torch.is_tensor(self._val) else None, len(col),
                           int((col.astype(np.int64) * wts).sum()),
                           int((np.ascontiguousarray(val, np.float32).view(np.int32).astype(np.int64) * wts).sum()))
        stream = np.zeros((n_pairs, 2), np.uint32)
        stream[start[:-1], 0] = rows.astype(np.uint32)
        stream[start[:-1], 1] = deg.astype(np.uint32)
        tot = int(deg.sum())
        if tot:
            within = np.arange(tot, dtype=np.int64) - np.repeat(np.cumsum(deg) - deg, deg)
            dst = np.repeat(start[:-1] + 1, deg) + within
            src = np.repeat(self._rp[rows], deg) + within
            stream[dst, 0] = col[src].astype(np.uint32)
            stream[dst, 1] = np.ascontiguousarray(val[src], np.float32).view(np.uint32)
        return (torch.from_numpy(stream.view(np.int32)).to(self._device), first, units[first] if len(first) else first,
                start[first] if len(first) else first)

    def _same_edges(self, col, val) -> bool:
        """Are the launch's edge arrays the ones the record stream was built from?  The stream bakes (col, val) of the LIGHT
        rows in while the heavy rows read the launch's arrays: a caller that hands a rescaled ``val`` (or another matrix) to
        a reused schedule would get a mix of both.  The schedule's own tensors pass by identity + version counter;
        any other pair is checked by two position-weighted 64-bit checksums against the bake-time sums (one reduction each) --
        remembered only for the same tensor objects -- and a pair that does not match, or cannot be checked because the stream
        is being captured, takes the descriptor path, which reads the launch's arrays."""
        if col is None or val is None:
            return True
        baked = getattr(self, "_baked", None)
        if baked is None:
            return False                                      # no stream was built: nothing to vouch for
        # (1) the very tensor OBJECTS the schedule holds (kept alive by it), untouched since the bake: no device work
        if col is self._col and val is self._val and (col._version, val._version) == baked[:2]:
            return True
        # (2) the pair checked last, still the same objects at the same versions.  Keyed by object identity through weak
        # references, never by address: a fresh ``val`` every step is usually recycled at the address of the last one
        # (an address-keyed entry then vouched for contents it never saw)
        last = self._checked
        if last is not None and last[0]() is col and last[1]() is val and last[2:4] == (col._version, val._version):
            return last[4]
        if col.is_cuda and torch.cuda.is_current_stream_capturing():
            if not getattr(self, "_warned_capture", False):
                self._warned_capture = True
                import warnings
                warnings.warn("SpmmSchedule: an edge-array pair met inside a stream capture has not been checked against the "
                              "record stream; this launch (and the captured graph) takes the slower descriptor path -- bind the "
                              "pair once eagerly (sched.for_launch(n, d, col, val)) before capturing")
            return False                                      # not remembered: an eager launch decides
        # (3) position-weighted checksums of the launch's arrays against the BAKE-TIME sums (a permutation changes them)
        ok = col.numel() == baked[2] and val.numel() == baked[2]
        if ok:
            w = torch.arange(1, col.numel() + 1, dtype=torch.int64, device=col.device)
            ok = int((col.to(torch.int64) * w).sum()) == baked[3] and \
                int((val.view(torch.int32).to(torch.int64) * w).sum()) == baked[4]
        self._checked = (weakref.ref(col), weakref.ref(val), col._version, val._version, ok)
        return ok

    def for_launch(self, n_rows: int, d: int, col=None, val=None):
        """Point the C struct at the record stream laid out for THIS launch's lane-group width (built once per width);
        without edge arrays, where the stream does not apply, or when the launch's (col, val) are not the arrays the stream
        was built from (``_same_edges``), the launch takes the descriptor path.  A schedule is bound to ONE (col, val)."""
        c = self.c
        G = int(_lib.lib().crh_spmm_lane_group(int(n_rows), int(d), int(self._rp[-1]))) if self._col is not None else 0
        if G not in self._slabs:
            self._slabs[G] = self._build_slab(G) if G else None
        slab = self._slabs[G]
        if slab is not None and not self._same_edges(col, val):
            slab = None
        if slab is None:
            c.slab, c.slab_lanes, c.slab_buckets, c.n_slab = None, 0, 0, 0
            return c
        stream, first, units, base = slab
        c.slab, c.slab_lanes, c.slab_buckets, c.n_slab = _lib.ptr(stream), G, len(first), len(self._light)
        for b in range(self.SLAB_BUCKETS):
            c.slab_first[b] = int(first[b]) if b < len(first) else 0x7fffffff
            c.slab_units[b] = int(units[b]) if b < len(first) else 1
            c.slab_base[b] = int(base[b]) if b < len(first) else 0
        return c

    def workspace(self, d: int, device) -> Optional[torch.Tensor]:
        """Heavy rows are combined on chip: crh_spmm_workspace_bytes() is 0 and no scratch is needed."""
        return None


def spmm_csr(rowptr, col, val, x, y=None, acc_in=None, s_in: float = 1.0, acc_out=None, s_out: float = 1.0,
             sched: Optional[SpmmSchedule] = None):
    """P = A @ x; y = P; acc_out = (acc_in*s_in + P)*s_out  (model/LightGCN.py:88-93, fused layer sum)."""
    _need_cuda(rowptr, col, val, x, y, acc_in, acc_out)
    assert rowptr.dtype == torch.int64 and col.dtype == torch.int32 and val.dtype == torch.float32
    assert x.dtype == torch.float32 and x.is_contiguous()
    n_rows, d = rowptr.shape[0] - 1, x.shape[1]
    ws = sched.workspace(d, x.device) if sched is not None else None
    rc = _lib.lib().crh_spmm_csr_f32(_lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(val), n_rows, _lib.ptr(x), d,
                                     _lib.ptr(y), _lib.ptr(acc_in), float(s_in), _lib.ptr(acc_out), float(s_out),
                                     ctypes.byref(sched.for_launch(n_rows, d, col, val)) if sched is not None else None,
                                     _lib.ptr(ws), ws.numel() * 4 if ws is not None else 0,
                                     _lib.current_stream())
    _lib.check(rc, "crh_spmm_csr_f32")
