"""Operator API the model plugins import (reference: util/utils.py), backed by the HIP library.

* ``next_batch_pairwise(data, batch_size, n_negs=1)`` -- same generator contract and the same
  NumPy-global-RNG stream as util/utils.py:123-157, produced by the C++ host sampler.
* ``next_batch_pairwise_LARA`` / ``_CLCRec`` / ``_CCFCRec`` / ``next_batch_cgrc`` -- the other samplers
  (util/utils.py:160-336), same generator contracts, CPython ``random`` / NumPy streams restated in C++.
* ``bpr_loss(u, p, n)`` / ``l2_reg_loss(reg, *embs)`` -- differentiable scalars
  (util/utils.py:25-29, 44-48) whose forward and backward run in crh_bpr_fwd_bwd_f32.
* ``set_seed(seed, cuda)`` -- seeds the same three generators (util/utils.py:339-348).

Tensors must live on the GPU: there is no CPU implementation behind these functions.
"""
from __future__ import annotations

import os
import random

import numpy as np
import torch

from .. import ops


def set_seed(seed, cuda):
    print('Set Seed: ', seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    if cuda and torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


def epoch_triples(data, batch_size):
    """One epoch of (u, i, j) int32 arrays from the NumPy global stream (fast path used by the
    built-in trainers; ``next_batch_pairwise`` slices the same arrays into lists)."""
    s = data.sampler
    s.pull_numpy_state()
    u, i, j = s.epoch(batch_size)
    s.push_numpy_state()
    return u, i, j


def next_batch_pairwise(data, batch_size, n_negs=1):
    """Yields (user_idx, pos_idx, neg_idx) Python lists of internal ids per batch; the last batch is
    short; ``n_negs`` is accepted and ignored exactly as in the reference."""
    u, i, j = epoch_triples(data, batch_size)
    for lo in range(0, u.shape[0], batch_size):
        hi = min(lo + batch_size, u.shape[0])
        yield u[lo:hi].tolist(), i[lo:hi].tolist(), j[lo:hi].tolist()


def _batches(n, batch_size):
    return ((lo, min(lo + batch_size, n)) for lo in range(0, n, batch_size))


def next_batch_pairwise_LARA(data, batch_size, n_negs=1):
    """util/utils.py:160-188: per record ``n_negs`` x (a non-rated item, a user that has not rated the positive),
    drawn from CPython's global ``random`` stream.  Yields (u_idx, i_idx, u_neg_idx, i_neg_idx) lists."""
    s = data.sampler
    s.pull_python_state()
    u, i, nu, ni = s.epoch_lara(n_negs)
    s.push_python_state()
    for lo, hi in _batches(u.shape[0], batch_size):
        yield u[lo:hi].tolist(), i[lo:hi].tolist(), nu[lo:hi].reshape(-1).tolist(), ni[lo:hi].reshape(-1).tolist()


def next_batch_pairwise_CLCRec(data, batch_size, n_negs=1):
    """util/utils.py:191-233: per record the positive plus ``random.sample`` of ``n_negs`` warm items the user
    has not rated.  Yields (u_idx, i_idx), both [batch, 1 + n_negs] nested lists."""
    s = data.sampler
    s.pull_python_state()
    u, it = s.epoch_clcrec(n_negs)
    s.push_python_state()
    for lo, hi in _batches(u.shape[0], batch_size):
        yield np.repeat(u[lo:hi, None], 1 + n_negs, 1).tolist(), it[lo:hi].tolist()


def next_batch_pairwise_CCFCRec(data, batch_size, positive_number, negative_number, self_neg_number):
    """util/utils.py:237-300.  Yields (u_idx, i_idx, neg_u_idx, pos_i_list [B][P], neg_i_list [B][P][N],
    self_neg_list [B][S]); positives come from NumPy's global stream, everything else from ``random``."""
    s = data.sampler
    s.pull_python_state()
    s.pull_numpy_state()
    u, i, nu, pos, neg, sneg = s.epoch_ccfcrec(positive_number, negative_number, self_neg_number)
    s.push_numpy_state()
    s.push_python_state()
    for lo, hi in _batches(u.shape[0], batch_size):
        yield (u[lo:hi].tolist(), i[lo:hi].tolist(), nu[lo:hi].tolist(), pos[lo:hi].tolist(),
               neg[lo:hi].tolist() if positive_number > 0 else [],       # the reference only appends inside the loops
               sneg[lo:hi].tolist() if self_neg_number > 0 else [])


def next_batch_cgrc(data, batch_size, ranking_neg_per_user=32):
    """util/utils.py:303-336.  Yields (u_idx, i_idx, B_list): B_list = the batch's positives plus up to
    ``ranking_neg_per_user`` non-rated draws per record, in the order ``list(set)`` gives on CPython."""
    if len(data.item) == 0:
        raise ValueError('next_batch_cgrc: empty item set')
    s = data.sampler
    s.pull_numpy_state()
    u, i, ptr, bset = s.epoch_cgrc(batch_size, ranking_neg_per_user)
    s.push_numpy_state()
    for b, (lo, hi) in enumerate(_batches(u.shape[0], batch_size)):
        yield u[lo:hi].tolist(), i[lo:hi].tolist(), bset[ptr[b]:ptr[b + 1]].tolist()


def _pad4(t):
    """Zero columns up to the next multiple of 4 (the kernels move 16 B per lane): exact for dots and norms."""
    t = t.contiguous().float()
    r = (-t.shape[1]) % 4
    return t if r == 0 else torch.nn.functional.pad(t, (0, r))


class _BprFn(torch.autograd.Function):
    """mean(-log(1e-5 + sigmoid(u.p - u.n))) on gathered (B, d) tensors (util/utils.py:25-29).  The forward
    (crh_bpr_fwd_f32) leaves the per-triple score differences in a private workspace that the backward
    (crh_bpr_bwd_f32) reads: the forward pass is not repeated."""

    @staticmethod
    def forward(ctx, u, p, n):
        d = u.shape[1]
        u, p, n = _pad4(u), _pad4(p), _pad4(n)
        B = u.shape[0]
        ws = ops.bpr_workspace(B, u.device)
        sums = torch.empty(4, dtype=torch.float32, device=u.device)
        ops.bpr_fwd(u, p, n, None, None, None, sums, ws)
        ctx.save_for_backward(u, p, n, sums, ws)
        ctx.d = d
        return sums[3] / B

    @staticmethod
    def backward(ctx, grad_out):
        u, p, n, sums, ws = ctx.saved_tensors
        gu, gp, gn = torch.zeros_like(u), torch.zeros_like(p), torch.zeros_like(n)
        # reg = 0: only the BPR term; identity indices -> every gradient row is written by exactly one triple
        ops.bpr_bwd(u, p, n, None, None, None, u.shape[0], 0.0, sums, gu, gp, gn, None, ws)
        d = ctx.d
        return (gu[:, :d] * grad_out, gp[:, :d] * grad_out, gn[:, :d] * grad_out)


class _L2Fn(torch.autograd.Function):
    """reg * |x|_F / rows(x) for ONE embedding tensor (a term of util/utils.py:44-48) over crh_l2_norm_f32 /
    crh_l2_reg_bwd_f32."""

    @staticmethod
    def forward(ctx, x, reg):
        xc = x.contiguous().float()
        norm = ops.l2_norm(xc)
        ctx.save_for_backward(xc, norm)
        ctx.reg = float(reg)
        return norm[0] * (ctx.reg / xc.shape[0])

    @staticmethod
    def backward(ctx, grad_out):
        xc, norm = ctx.saved_tensors
        return ops.l2_reg_bwd(xc, ctx.reg, norm, grad_out), None


def _check_gpu(*ts):
    for t in ts:
        if not (torch.is_tensor(t) and t.is_cuda):
            raise RuntimeError("coldrec_amd.util.utils: embeddings must be CUDA (MI355X) tensors; no CPU path")


def bpr_loss(user_emb, pos_item_emb, neg_item_emb):
    _check_gpu(user_emb, pos_item_emb, neg_item_emb)
    return _BprFn.apply(user_emb, pos_item_emb, neg_item_emb)


def l2_reg_loss(reg, *args):
    """reg * sum_e |e|_F / rows(e).  Any number of embeddings (other models pass 2..6), any shapes; every term runs
    forward and backward in the HIP library.  Same association as the reference: the norms are summed first, the
    product with reg comes last."""
    _check_gpu(*args)
    emb_loss = 0
    for emb in args:
        emb_loss = emb_loss + _L2Fn.apply(emb.reshape(emb.shape[0], -1) if emb.dim() != 2 else emb, 1.0)
    return emb_loss * reg
