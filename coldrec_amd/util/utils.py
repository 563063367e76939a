"""Operator API the model plugins import (reference: util/utils.py), backed by the HIP library.

* ``next_batch_pairwise(data, batch_size, n_negs=1)`` -- same generator contract and the same
  NumPy-global-RNG stream as util/utils.py:123-157, produced by the C++ host sampler.
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


class _BprFn(torch.autograd.Function):
    """mean(-log(1e-5 + sigmoid(u.p - u.n))) on gathered (B, d) tensors."""

    @staticmethod
    def forward(ctx, u, p, n):
        u, p, n = (t.contiguous().float() for t in (u, p, n))
        ctx.save_for_backward(u, p, n)
        return ops.bpr_fwd_bwd(u, p, n, None, None, None, 0.0)[0].clone()

    @staticmethod
    def backward(ctx, grad_out):
        u, p, n = ctx.saved_tensors
        gu, gp, gn = torch.zeros_like(u), torch.zeros_like(p), torch.zeros_like(n)
        ops.bpr_fwd_bwd(u, p, n, None, None, None, 0.0, gu, gp, gn)
        return gu * grad_out, gp * grad_out, gn * grad_out


def _check_gpu(*ts):
    for t in ts:
        if not (torch.is_tensor(t) and t.is_cuda):
            raise RuntimeError("coldrec_amd.util.utils: embeddings must be CUDA (MI355X) tensors; no CPU path")


def bpr_loss(user_emb, pos_item_emb, neg_item_emb):
    _check_gpu(user_emb, pos_item_emb, neg_item_emb)
    if user_emb.shape[1] % 4:
        raise RuntimeError("bpr_loss: embedding width must be a multiple of 4 for the HIP kernel")
    return _BprFn.apply(user_emb, pos_item_emb, neg_item_emb)


def l2_reg_loss(reg, *args):
    """reg * sum_e |e|_F / rows(e).  Any number of embeddings (other models pass 2..6)."""
    _check_gpu(*args)
    emb_loss = 0
    for emb in args:
        emb_loss = emb_loss + torch.norm(emb, p=2) / emb.shape[0]
    return emb_loss * reg
