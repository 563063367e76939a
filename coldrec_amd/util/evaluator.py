"""Ranking metrics on arrays (consumer of the top-k output).

Same definitions, rounding and return format as the reference's util/evaluator.py:4-187:
  Hit Ratio = sum_u hits_u / sum_u |truth_u|                (interaction level)
  Precision = sum_u hits_u / (|users| * n)
  Recall    = mean over users with non-empty truth of hits_u / |truth_u|
  NDCG      = mean over users with IDCG > 0 of DCG_u / IDCG_u,  DCG weights 1/log2(rank + 2)
each rounded to 5 decimals -- but evaluated on a (users, k) index array and a CSR ground truth, so
the per-epoch validation costs milliseconds instead of the dict/set loops of the reference.
``ranking_evaluation(origin, res, N)`` keeps the dict-based signature for plugins that call it.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np


def truth_csr(origin: Dict, item_of=None) -> Tuple[List, np.ndarray, np.ndarray]:
    """users (dict order), rowptr, item ids of the ground truth {user: {item: 1.0}}."""
    users = list(origin.keys())
    lens = np.fromiter((len(origin[u]) for u in users), dtype=np.int64, count=len(users))
    rowptr = np.zeros(len(users) + 1, np.int64)
    np.cumsum(lens, out=rowptr[1:])
    items = np.fromiter((it if item_of is None else item_of[it] for u in users for it in origin[u].keys()),
                        dtype=np.int64, count=int(rowptr[-1]))
    return users, rowptr, items


def truth_dense(gt_rowptr: np.ndarray, gt_items: np.ndarray, n_items: int, max_cells: int = 1 << 26):
    """bool (users, n_items) membership table of a ground truth, or None when it would exceed ``max_cells``:
    the per-epoch validation then tests its (users, k) predictions with one gather instead of a sort."""
    n_user = len(gt_rowptr) - 1
    if n_user * max(int(n_items), 1) > max_cells or (len(gt_items) and int(gt_items.max()) >= n_items):
        return None
    dense = np.zeros((n_user, int(n_items)), dtype=bool)
    rows = np.repeat(np.arange(n_user, dtype=np.int64), np.diff(gt_rowptr))
    dense[rows, np.asarray(gt_items, np.int64)] = True
    return dense


def hit_matrix(gt_rowptr: np.ndarray, gt_items: np.ndarray, pred: np.ndarray, dense=None) -> np.ndarray:
    """bool (users, k): pred[r, q] is in the ground truth of row r."""
    n_user, k = pred.shape
    if dense is not None:
        p = np.asarray(pred, np.int64)
        ok = (p >= 0) & (p < dense.shape[1])
        return dense[np.arange(n_user)[:, None], np.where(ok, p, 0)] & ok
    base = int(max(gt_items.max(initial=0), pred.max(initial=0))) + 1
    rows = np.repeat(np.arange(n_user, dtype=np.int64), np.diff(gt_rowptr))
    gt_keys = rows * base + gt_items.astype(np.int64)
    pr_keys = np.arange(n_user, dtype=np.int64)[:, None] * base + pred.astype(np.int64)
    return np.isin(pr_keys, gt_keys)


def _seq_sum(x: np.ndarray) -> float:
    """Left-to-right float64 sum, as Python's sum() over the reference's per-user lists (np.cumsum adds strictly in
    order; np.sum is pairwise and rounds differently)."""
    return float(np.cumsum(np.asarray(x, np.float64))[-1]) if len(x) else 0.0


def ranking_metrics(gt_rowptr, gt_items, pred, topn: Sequence[int], dense=None, hit=None) -> List[List[float]]:
    """[[hit ratio, precision, recall, ndcg] for n in topn]; pred is (users, >= max(topn)); ``dense``: optional
    ``truth_dense`` table of the same ground truth; ``hit``: the bool (users, k) membership matrix when the caller
    already has it (the trainers test membership on the GPU and bring back one bit per prediction)."""
    gt_rowptr = np.asarray(gt_rowptr, np.int64)
    n_user = gt_rowptr.shape[0] - 1
    tlen = np.diff(gt_rowptr)
    if hit is None:
        hit = hit_matrix(gt_rowptr, np.asarray(gt_items), np.asarray(pred), dense)
    out = []
    hit_t = np.ascontiguousarray(np.asarray(hit, bool).T)     # (k, users): every rank's column is one contiguous pass
    for n in topn:
        hits = hit_t[:n].sum(0, dtype=np.int64)
        total_truth = int(tlen.sum())
        hr = round(int(hits.sum()) / total_truth, 5) if total_truth else 0.0
        prec = round(int(hits.sum()) / (n_user * n), 5) if n_user and n else 0.0
        has = tlen > 0
        recall = round(_seq_sum(hits[has] / tlen[has]) / int(has.sum()), 5) if has.any() else 0.0
        w = np.array([1.0 / math.log(q + 2, 2) for q in range(n)])
        cw = np.concatenate([[0.0], np.cumsum(w)])
        dcg = np.zeros(n_user)
        for q in range(n):                               # rank-ascending accumulation, as the reference (x + 0.0 == x:
            np.add(dcg, w[q], out=dcg, where=hit_t[q])   # adding only where the rank hit gives the same bits)
        idcg = cw[np.minimum(tlen, n)]
        ok = idcg > 0
        ndcg = round(_seq_sum(dcg[ok] / idcg[ok]) / int(ok.sum()), 5) if ok.any() else 0.0
        out.append([hr, prec, recall, ndcg])
    return out


def format_measure(perf: List[List[float]], topn: Sequence[int]) -> List[str]:
    """The flat list of strings the reference returns next to the numbers (evaluator.py:153-187)."""
    measure = []
    for (hr, prec, recall, ndcg), n in zip(perf, topn):
        measure += ['Top ' + str(n) + '\n', 'Hit Ratio:' + str(hr) + '\n', 'Precision:' + str(prec) + '\n',
                    'Recall:' + str(recall) + '\n', 'NDCG:' + str(ndcg) + '\n']
    return measure


def ranking_evaluation(origin, res, N):
    """Dict-based entry point: origin {user: {item: 1.0}}, res {user: [(item, score), ...]}."""
    if len(origin) != len(res):
        print(f"ground-truth set size: {len(origin)}, predicted set size: {len(res)}")
        print('The Lengths of ground-truth set and predicted set do not match!')
        exit(-1)
    ids: Dict = {}

    def key(x):
        return ids.setdefault(x, len(ids))

    users, rowptr, items = truth_csr(origin, item_of=None)
    items = np.fromiter((key(it) for it in items.tolist()), dtype=np.int64, count=items.shape[0]) \
        if items.size else items
    width = max(N)
    pred = np.full((len(users), width), -1, np.int64)
    for r, u in enumerate(users):
        row = [key(it) for it, _ in res[u][:width]]
        pred[r, :len(row)] = row
    perf = ranking_metrics(rowptr, items, pred, N)
    return format_measure(perf, N), [list(p) for p in perf]
