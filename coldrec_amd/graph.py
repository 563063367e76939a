"""Sparse adjacency handle that lets unmodified model files hit the HIP SpMM.

Model plugins do ``adj = TorchGraphInterface.convert_sparse_mat_to_tensor(norm_adj).to(device)`` once
and ``torch.sparse.mm(adj, dense)`` per layer (model/LightGCN.py:76,90; also NGCF, SimGCL, CGRC,
FSGNN).  ``HipSparseAdj`` is a ``torch.Tensor`` subclass around the same coalesced COO tensor the
reference builds (util/databuilder.py:953-962); through ``__torch_function__`` it answers
``torch.sparse.mm`` with crh_spmm_csr_f32 when the dense operand lives on the GPU (autograd
included: d/dX (A X) = A^T G, CSR of A^T kept alongside), and forwards everything else to the
COO tensor untouched.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import torch

from . import ops


def _pad4(x: torch.Tensor) -> torch.Tensor:
    """The kernels move 16 bytes per lane: a width that is not a multiple of 4 (the reference accepts any --emb_size)
    gets zero columns up to the next one.  A zero column of the dense operand gives a zero column of the product (every
    term is val * 0 = 0 added to 0), so the real columns are bit for bit what the unpadded product would be -- the same
    rule the built-in engines apply to their tables (train._TableState)."""
    x = x.contiguous().float()
    d = x.shape[1]
    if d % 4 == 0:
        return x
    out = torch.zeros((x.shape[0], (d + 3) // 4 * 4), dtype=torch.float32, device=x.device)
    out[:, :d] = x
    return out


class _SpmmFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, adj: "HipSparseAdj", dense: torch.Tensor):
        ctx.adj = adj
        d = dense.shape[1]
        x = _pad4(dense)
        y = torch.empty((adj.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
        rp, col, val, heavy = adj.csr(x.device)
        ops.spmm_csr(rp, col, val, x, y=y, sched=heavy)
        return y if x.shape[1] == d else y[:, :d].contiguous()

    @staticmethod
    def backward(ctx, grad_out):
        d = grad_out.shape[1]
        g = _pad4(grad_out)
        rp, col, val, heavy = ctx.adj.csr(g.device, transposed=True)
        gx = torch.empty((ctx.adj.shape[1], g.shape[1]), dtype=torch.float32, device=g.device)
        ops.spmm_csr(rp, col, val, g, y=gx, sched=heavy)
        return None, (gx if g.shape[1] == d else gx[:, :d].contiguous())


_KEEP_WRAPPED = (torch.Tensor.to, torch.Tensor.cuda, torch.Tensor.cpu, torch.Tensor.float, torch.Tensor.detach,
                 torch.Tensor.clone, torch.Tensor.coalesce)


class HipSparseAdj(torch.Tensor):
    @staticmethod
    def from_scipy(X) -> "HipSparseAdj":
        coo = sp.coo_matrix(X)
        idx = torch.from_numpy(np.vstack((coo.row, coo.col)).astype(np.int64))
        coo_t = torch.sparse_coo_tensor(idx, torch.from_numpy(coo.data.astype(np.float32)), coo.shape).coalesce()
        return HipSparseAdj._wrap(coo_t, sp.csr_matrix(X, dtype=np.float32))

    @staticmethod
    def _wrap(coo_t: torch.Tensor, csr_host) -> "HipSparseAdj":
        out = torch.Tensor._make_subclass(HipSparseAdj, coo_t)
        out._coo = coo_t
        out._csr_host = csr_host
        out._dev = {}
        return out

    def csr(self, device, transposed: bool = False):
        key = (str(device), transposed)
        if key not in self._dev:
            m = self._csr_host.T.tocsr() if transposed else self._csr_host
            m.sort_indices()
            self._dev[key] = (torch.from_numpy(m.indptr.astype(np.int64)).to(device),
                              torch.from_numpy(m.indices.astype(np.int32)).to(device),
                              torch.from_numpy(m.data.astype(np.float32)).to(device),
                              ops.SpmmSchedule(m.indptr, device, col=m.indices.astype(np.int32),
                                               val=m.data.astype(np.float32)))
        return self._dev[key]

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is torch.sparse.mm and len(args) == 2 and isinstance(args[0], HipSparseAdj) \
                and isinstance(args[1], torch.Tensor) and args[1].is_cuda and args[1].dim() == 2 \
                and args[1].layout == torch.strided:
            return _SpmmFn.apply(args[0], args[1])             # ANY width: _pad4 (no silent hipSPARSE route)
        if func in _KEEP_WRAPPED and isinstance(args[0], HipSparseAdj):
            # device moves, and what nn.Module._apply does to a registered buffer (model/FSGNN.py:249-272)
            moved = func(args[0]._coo, *args[1:], **kwargs)
            if moved.layout == torch.sparse_coo and moved.dtype == torch.float32:
                return HipSparseAdj._wrap(moved, args[0]._csr_host)
            return moved
        plain = [a._coo if isinstance(a, HipSparseAdj) else a for a in args]
        with torch._C.DisableTorchFunctionSubclass():
            return func(*plain, **kwargs)
