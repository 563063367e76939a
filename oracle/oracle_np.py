"""ORACLE -- TEST INFRASTRUCTURE ONLY (numpy / C restatement of the reference hot path).

Nothing under ``coldrec_amd/`` may import this module; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and only as the
checker.  Each function names the reference lines (relative to the upstream repository
YuanchenBei/ColdRec) it restates.  The restatement is pinned against outputs of the
reference itself run in the build container: ``tests/golden/*.npz`` produced by
``tests/golden/make_golden.py`` (checked by ``tests/test_oracle_golden.py``).

The arithmetic of the reference lives in PyTorch / NumPy library calls (torch >= 1.11 and
numpy >= 1.24.4 per the reference README; oracle pinned with torch 2.10.0, numpy 2.2.6).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MASKED = np.float32(-1.0e9)        # -10e8, model/BaseRecommender.py:177,180
PAD_IDX = np.int32(0x7FFFFFFF)


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "_build", "liboracle.so")
    src = os.path.join(_HERE, "topk_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ct))


# ----------------------------------------------------------------------------- A7 / A8
def make_bitmap(n_items: int, masked_ids: Optional[np.ndarray]) -> Optional[np.ndarray]:
    if masked_ids is None or len(masked_ids) == 0:
        return None
    words = np.zeros((n_items + 31) // 32, dtype=np.uint32)
    ids = np.asarray(masked_ids, dtype=np.int64)
    np.bitwise_or.at(words, ids >> 5, (np.uint32(1) << (ids & 31).astype(np.uint32)))
    return words


def sort_rated(rowptr: np.ndarray, col: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Per-row ascending int32 copy of a rated CSR (the C code walks it monotonically)."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    out = np.empty(len(col), dtype=np.int32)
    for r in range(len(rowptr) - 1):
        out[rowptr[r]:rowptr[r + 1]] = np.sort(col[rowptr[r]:rowptr[r + 1]])
    return rowptr, out


def score_topk(U, users, V, k, rated_rowptr=None, rated_col=None, cand_bitmap=None,
               item_base: int = 0):
    """model/MF.py:58-63 + model/BaseRecommender.py:175-182 in canonical order (C)."""
    U = np.ascontiguousarray(U, np.float32)
    V = np.ascontiguousarray(V, np.float32)
    users = None if users is None else np.ascontiguousarray(users, np.int64)
    n = U.shape[0] if users is None else users.shape[0]
    if rated_rowptr is not None:
        rated_rowptr, rated_col = sort_rated(rated_rowptr, rated_col)
    sc = np.empty((n, k), np.float32)
    ix = np.empty((n, k), np.int32)
    _lib().orc_score_topk(_p(U, ctypes.c_float), _p(users, ctypes.c_int64), ctypes.c_int64(n),
                          _p(V, ctypes.c_float), ctypes.c_int64(V.shape[0]), ctypes.c_int(U.shape[1]),
                          _p(rated_rowptr, ctypes.c_int64), _p(rated_col, ctypes.c_int32),
                          _p(cand_bitmap, ctypes.c_uint32), ctypes.c_int(k), ctypes.c_int64(item_base),
                          _p(sc, ctypes.c_float), _p(ix, ctypes.c_int32))
    return sc, ix


def mask_topk(S, k, rated_rowptr=None, rated_col=None, cand_bitmap=None, item_base: int = 0,
              write_back: bool = False):
    S = np.ascontiguousarray(S, np.float32) if not write_back else S
    assert S.dtype == np.float32 and S.flags.c_contiguous
    n, m = S.shape
    if rated_rowptr is not None:
        rated_rowptr, rated_col = sort_rated(rated_rowptr, rated_col)
    sc = np.empty((n, k), np.float32)
    ix = np.empty((n, k), np.int32)
    work = S if write_back else S.copy()
    _lib().orc_mask_topk(_p(work, ctypes.c_float), ctypes.c_int64(n), ctypes.c_int64(m),
                         _p(rated_rowptr, ctypes.c_int64), _p(rated_col, ctypes.c_int32),
                         _p(cand_bitmap, ctypes.c_uint32), ctypes.c_int(k), ctypes.c_int64(item_base),
                         ctypes.c_int(1 if write_back else 0), _p(sc, ctypes.c_float),
                         _p(ix, ctypes.c_int32))
    return sc, ix


def merge_topk(scores, idx, k_out):
    """scores/idx: (n_lists, n_users, k_in)."""
    scores = np.ascontiguousarray(scores, np.float32)
    idx = np.ascontiguousarray(idx, np.int32)
    L, n, k_in = scores.shape
    sc = np.empty((n, k_out), np.float32)
    ix = np.empty((n, k_out), np.int32)
    _lib().orc_merge_topk(_p(scores, ctypes.c_float), _p(idx, ctypes.c_int32), ctypes.c_int(L),
                          ctypes.c_int64(n), ctypes.c_int(k_in), ctypes.c_int(k_out),
                          _p(sc, ctypes.c_float), _p(ix, ctypes.c_int32))
    return sc, ix


def scores_dense(U, users, V):
    """Canonical (fmaf-chain) dense score matrix, shape (n_users, n_items)."""
    U = np.ascontiguousarray(U, np.float32)
    V = np.ascontiguousarray(V, np.float32)
    users = None if users is None else np.ascontiguousarray(users, np.int64)
    n = U.shape[0] if users is None else users.shape[0]
    S = np.empty((n, V.shape[0]), np.float32)
    _lib().orc_scores_dense(_p(U, ctypes.c_float), _p(users, ctypes.c_int64), ctypes.c_int64(n),
                            _p(V, ctypes.c_float), ctypes.c_int64(V.shape[0]), ctypes.c_int(U.shape[1]),
                            _p(S, ctypes.c_float))
    return S


def dot_chain(a, b) -> np.float32:
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    f = _lib().orc_dot_chain
    f.restype = ctypes.c_float
    return np.float32(f(_p(a, ctypes.c_float), _p(b, ctypes.c_float), ctypes.c_int(a.shape[0])))


# ----------------------------------------------------------------------------- A1
class PairwiseSampler:
    """util/utils.py:123-157 ``next_batch_pairwise`` on internal ids, NumPy legacy global RNG.

    ``rec_u/rec_i`` are the training records (internal ids) in file order; the reference
    shuffles ``data.training_data`` in place each epoch, so the permutation is cumulative
    (``self.order``).  ``item_list = list(data.item.keys())`` is in id-table order, so
    ``np.random.choice(item_list, n)`` (== ``item_list[np.random.randint(0, len, n)]``) draws
    INTERNAL ids directly; rejection tests membership in the user's training items.
    """

    def __init__(self, rec_u, rec_i, n_items_seen: int, n_users: int):
        self.rec_u = np.asarray(rec_u, np.int64)
        self.rec_i = np.asarray(rec_i, np.int64)
        self.n_items = int(n_items_seen)
        self.order = np.arange(self.rec_u.shape[0])
        self.rated = [set() for _ in range(n_users)]
        for u, i in zip(self.rec_u.tolist(), self.rec_i.tolist()):
            self.rated[u].add(i)

    def epoch(self, batch_size: int):
        np.random.shuffle(self.order)                                  # utils.py:125
        n = self.order.shape[0]
        for lo in range(0, n, batch_size):
            sel = self.order[lo:min(lo + batch_size, n)]
            u = self.rec_u[sel]
            i = self.rec_i[sel]
            j = np.zeros(u.shape[0], np.int64)
            check = np.arange(u.shape[0])
            while check.size:                                           # utils.py:141-153
                j[check] = np.random.randint(0, self.n_items, size=check.size)
                check = np.array([c for c in check.tolist() if int(j[c]) in self.rated[int(u[c])]],
                                 dtype=np.int64)
            yield u.copy(), i.copy(), j


class OtherSamplers:
    """util/utils.py:160-336 on internal ids, drawing from CPython's global ``random`` and
    NumPy's legacy global RNG exactly where the reference does.  Plain Python loops: small cases only.

    ``data.item`` / ``data.user`` are insertion-ordered dicts original id -> internal id, so
    ``list(data.item.keys())[k]`` has internal id k and ``choice(item_list)`` draws an internal id directly;
    ``training_set_u[user]`` is a dict in first-appearance order of the unshuffled records.
    """

    def __init__(self, rec_u, rec_i, n_users_seen: int, n_items_seen: int, cold_items=()):
        self.rec = list(zip(np.asarray(rec_u).tolist(), np.asarray(rec_i).tolist()))   # shuffled in place, cumulative
        self.n_users, self.n_items = int(n_users_seen), int(n_items_seen)
        self.by_user, self.by_item = {}, {}
        for u, i in self.rec:
            self.by_user.setdefault(u, {})[i] = 1.0
            self.by_item.setdefault(i, {})[u] = 1.0
        cold = frozenset(int(c) for c in cold_items)
        self.pool = [k for k in range(self.n_items) if k not in cold]                 # utils.py:198-199, 245-246
        self._cand = {}

    def _candidates(self, u):
        if u not in self._cand:                                                       # utils.py:205-208, 252-255
            self._cand[u] = [k for k in self.pool if k not in self.by_user.get(u, {})]
        return self._cand[u]

    def lara_epoch(self, n_negs=1):                                                   # utils.py:160-188
        import random
        random.shuffle(self.rec)
        users, items = list(range(self.n_users)), list(range(self.n_items))
        out = ([], [], [], [])
        for u, i in self.rec:
            out[0].append(u); out[1].append(i)
            for _ in range(n_negs):
                j = random.choice(items)
                while j in self.by_user[u]:
                    j = random.choice(items)
                out[3].append(j)
                v = random.choice(users)
                while v in self.by_item[i]:
                    v = random.choice(users)
                out[2].append(v)
        return out

    def clcrec_epoch(self, n_negs=1):                                                 # utils.py:191-233
        import random
        random.shuffle(self.rec)
        if not self.pool:
            raise ValueError("empty warm pool")
        us, rows = [], []
        for u, i in self.rec:
            cand = self._candidates(u)
            if len(cand) < n_negs:
                raise ValueError("too few warm negatives")
            us.append(u)
            rows.append([i] + random.sample(cand, n_negs))
        return us, rows

    def ccfcrec_epoch(self, P, N, S):                                                 # utils.py:237-300
        import random
        random.shuffle(self.rec)
        users = list(range(self.n_users))
        out = ([], [], [], [], [], [])
        for u, i in self.rec:
            out[0].append(u); out[1].append(i)
            v = random.choice(users)
            while v in self.by_item[i]:
                v = random.choice(users)
            out[2].append(v)
            out[3].append(np.random.choice(list(self.by_user[u]), P, replace=True).tolist())
            cand = self._candidates(u)
            if not cand:
                raise ValueError("no warm negatives")
            flat = [random.choice(cand) for _ in range(P * N)]
            out[4].append([flat[m * N:(m + 1) * N] for m in range(P)])
            out[5].append([random.choice(cand) for _ in range(S)])
        return out

    def cgrc_epoch(self, batch_size, ranking_neg_per_user=32):                        # utils.py:303-336
        order = np.arange(len(self.rec))
        np.random.shuffle(order)                          # == np.random.shuffle(list): same draws, same permutation
        self.rec = [self.rec[k] for k in order.tolist()]
        keys = list(range(self.n_items))
        for lo in range(0, len(self.rec), batch_size):
            part = self.rec[lo:lo + batch_size]
            bset = set(i for _, i in part)
            for u, _ in part:
                added = tries = 0
                while added < ranking_neg_per_user and tries < ranking_neg_per_user * 50:
                    tries += 1
                    j = int(np.random.choice(keys))
                    if j not in self.by_user[u]:
                        bset.add(j)
                        added += 1
            yield [u for u, _ in part], [i for _, i in part], list(bset)


# ----------------------------------------------------------------------------- A2
def bpr_l2_fwd_bwd(U, V, ui, pi, ni, reg: float):
    """util/utils.py:25-29 + :44-48 and the gradients autograd produces at MF.py:22-27.

    Returns (bpr, l2, dense gradU, dense gradV) -- gradients accumulate over duplicate rows
    (index backward = index_put_(accumulate=True)).  Evaluated in float64 from fp32 inputs.
    """
    U64 = np.asarray(U, np.float64)
    V64 = np.asarray(V, np.float64)
    u, p, n = U64[ui], V64[pi], V64[ni]
    B = u.shape[0]
    x = (u * p).sum(1) - (u * n).sum(1)
    with np.errstate(over="ignore"):
        sig = 1.0 / (1.0 + np.exp(-x))
    bpr = np.mean(-np.log(1e-5 + sig))                                  # 10e-6 literal
    nu, npp, nn = np.sqrt((u * u).sum()), np.sqrt((p * p).sum()), np.sqrt((n * n).sum())
    l2 = reg * (nu + npp + nn) / B
    g = -(1.0 / B) * sig * (1.0 - sig) / (1e-5 + sig)
    gu = g[:, None] * (p - n) + (reg / (B * nu)) * u
    gp = g[:, None] * u + (reg / (B * npp)) * p
    gn = -g[:, None] * u + (reg / (B * nn)) * n
    gU = np.zeros_like(U64)
    gV = np.zeros_like(V64)
    np.add.at(gU, ui, gu)
    np.add.at(gV, pi, gp)
    np.add.at(gV, ni, gn)
    return np.float64(bpr), np.float64(l2), gU, gV, (gu, gp, gn)


# ----------------------------------------------------------------------------- A3
def adam_dense(p, g, m, v, step: int, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults, single-tensor path (torch/optim/adam.py), as used at
    model/MF.py:14,25-27: every element of the table moves every step.  fp32 state, scalar
    factors in Python floats then cast like torch does."""
    p = p.astype(np.float32, copy=True)
    m = m.astype(np.float32, copy=True)
    v = v.astype(np.float32, copy=True)
    g = np.asarray(g, np.float32)
    m += (g - m) * np.float32(1 - b1)                                   # lerp_
    v *= np.float32(b2)
    v += np.float32(1 - b2) * g * g                                     # addcmul_
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = np.sqrt(v) / np.float32(np.sqrt(bc2)) + np.float32(eps)
    p += np.float32(-lr / bc1) * (m / denom)                            # addcdiv_
    return p, m, v


def sgd_dense(p, g, lr=1e-3):
    """torch.optim.SGD(lr) defaults (no momentum, no weight decay): p <- p + (-lr) * g.  Canonical form of this
    build: ONE fused multiply-add per element, fma(-lr, g, p), evaluated here in float64 and rounded once (the
    product of two fp32 values is exact in float64, so this is the correctly rounded fma).  ATen's CPU kernel for
    param.add_(grad, alpha=-lr) uses a vector fma too; tests/test_oracle_golden.py pins this function to
    torch.optim.SGD within one fp32 ulp.  (north_star: "BPR loss + SGD update"; SURVEY.md F3: the reference's own
    trainers use Adam -- this is the extra mode.)"""
    p64 = np.asarray(p, np.float32).astype(np.float64)
    g64 = np.asarray(g, np.float32).astype(np.float64)
    return (p64 + np.float64(np.float32(-lr)) * g64).astype(np.float32)


def l2_reg(reg, *embs):
    """util/utils.py:44-48: reg * sum_e |e|_F / rows(e) and d/de = reg * e / (rows * |e|_F), in float64."""
    loss, grads = 0.0, []
    for e in embs:
        e64 = np.asarray(e, np.float64)
        nrm = np.sqrt((e64 * e64).sum())
        loss += nrm / e64.shape[0]
        grads.append(reg * e64 / (e64.shape[0] * nrm) if nrm > 0 else np.zeros_like(e64))
    return reg * loss, grads


# ----------------------------------------------------------------------------- A5
def norm_adj_csr(rec_u, rec_i, user_num: int, item_num: int):
    """util/databuilder.py:220-254: A=[[0,R],[R^T,0]], A_hat = D^-1/2 A D^-1/2 in fp32 with
    zero-degree rows -> 0; duplicates in the training pairs sum (csr_matrix semantics).
    Returns (rowptr int64, col int32, val fp32) with columns ascending per row."""
    n = user_num + item_num
    r = np.concatenate([np.asarray(rec_u, np.int64), np.asarray(rec_i, np.int64) + user_num])
    c = np.concatenate([np.asarray(rec_i, np.int64) + user_num, np.asarray(rec_u, np.int64)])
    key = r * n + c
    uk, cnt = np.unique(key, return_counts=True)
    rr, cc = uk // n, uk % n
    w = cnt.astype(np.float32)
    rowsum = np.zeros(n, np.float32)
    np.add.at(rowsum, rr, w)
    d_inv = np.zeros(n, np.float32)
    nz = rowsum != 0
    d_inv[nz] = np.power(rowsum[nz], np.float32(-0.5)).astype(np.float32)
    val = (d_inv[rr] * w).astype(np.float32) * d_inv[cc]
    rowptr = np.zeros(n + 1, np.int64)
    np.add.at(rowptr, rr + 1, 1)
    return np.cumsum(rowptr), cc.astype(np.int32), val.astype(np.float32)


def spmm(rowptr, col, val, X, alpha=1.0, beta=0.0, Z=None):
    X = np.ascontiguousarray(X, np.float32)
    Y = np.empty_like(X[: len(rowptr) - 1]) if X.shape[0] == len(rowptr) - 1 else np.empty((len(rowptr) - 1, X.shape[1]), np.float32)
    Zc = None if Z is None else np.ascontiguousarray(Z, np.float32)
    _lib().orc_spmm_csr(_p(np.ascontiguousarray(rowptr, np.int64), ctypes.c_int64),
                        _p(np.ascontiguousarray(col, np.int32), ctypes.c_int32),
                        _p(np.ascontiguousarray(val, np.float32), ctypes.c_float),
                        ctypes.c_int64(len(rowptr) - 1), _p(X, ctypes.c_float), ctypes.c_int(X.shape[1]),
                        ctypes.c_float(alpha), ctypes.c_float(beta), _p(Zc, ctypes.c_float),
                        _p(Y, ctypes.c_float))
    return Y


# ----------------------------------------------------------------------------- A6
def lgcn_forward(rowptr, col, val, U, V, n_layers: int):
    """model/LightGCN.py:86-96: E0 = cat(U,V); E_{k+1} = A_hat E_k; out = mean(E0..EL)."""
    E = np.concatenate([U, V], 0).astype(np.float32)
    acc = E.astype(np.float64)
    for _ in range(n_layers):
        E = spmm(rowptr, col, val, E)
        acc += E
    out = (acc / (n_layers + 1)).astype(np.float32)
    return out[: U.shape[0]], out[U.shape[0]:]


def lgcn_backward(rowptr, col, val, gOutU, gOutV, n_layers: int):
    """Autograd of lgcn_forward: dE0 = 1/(L+1) * sum_k A_hat^k dOut (A_hat symmetric),
    evaluated by Horner's rule with L SpMMs."""
    g0 = (np.concatenate([gOutU, gOutV], 0) / (n_layers + 1)).astype(np.float32)
    G = g0.copy()
    for _ in range(n_layers):
        G = spmm(rowptr, col, val, G, 1.0, 1.0, g0)
    return G[: gOutU.shape[0]], G[gOutU.shape[0]:]


# ----------------------------------------------------------------------------- metrics (row f1)
def ranking_metrics(gt_rowptr, gt_items, pred, topn: Sequence[int]) -> List[List[float]]:
    """util/evaluator.py:4-115,153-187 on arrays: per cut-off [hit ratio, precision, recall,
    NDCG], each rounded to 5 decimals (Python round on float64)."""
    import math
    res = []
    n_user = len(gt_rowptr) - 1
    for n in topn:
        hits_total = 0
        gt_total = 0
        recalls = []
        ndcgs = []
        for r in range(n_user):
            truth = gt_items[gt_rowptr[r]:gt_rowptr[r + 1]]
            tset = set(truth.tolist())
            p = pred[r, :n].tolist()
            h = len(tset.intersection(p))
            hits_total += h
            gt_total += len(truth)
            if len(truth):
                recalls.append(h / len(truth))
            dcg = sum(1.0 / math.log(q + 2, 2) for q, it in enumerate(p) if it in tset)
            idcg = sum(1.0 / math.log(q + 2, 2) for q in range(min(len(truth), n)))
            if idcg > 0:
                ndcgs.append(dcg / idcg)
        hr = round(hits_total / gt_total, 5) if gt_total else 0.0
        prec = round(hits_total / (n_user * n), 5) if n_user and n else 0.0
        rec = round(sum(recalls) / len(recalls), 5) if recalls else 0.0
        ndcg = round(sum(ndcgs) / len(ndcgs), 5) if ndcgs else 0.0
        res.append([hr, prec, rec, ndcg])
    return res
