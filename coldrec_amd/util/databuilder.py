"""Data side of the hot path: id tables, training sets, the normalised bipartite adjacency.

Mirrors the attribute surface of the reference's ``ColdStartDataBuilder`` / ``TorchGraphInterface``
(util/databuilder.py:6-385, 953-962) that model plugins read, built with
array code instead of per-record Python loops, plus the array views the HIP path consumes:

* ``train_u`` / ``train_i``   internal ids of the training records in file order (sampler, graph)
* ``rated_rowptr`` / ``rated_col``  per internal user, ascending internal training-item ids
* ``sampler``                 the MT19937-exact host sampler (coldrec_amd/sampler.py)

Internal ids are assigned in first-appearance order over train, warm val, warm test, cold val,
cold test, overall val, overall test (util/databuilder.py:90-216).
"""
from __future__ import annotations

from collections import defaultdict
from typing import Dict

import numpy as np
import scipy.sparse as sp


def _as_pairs(records) -> np.ndarray:
    if isinstance(records, np.ndarray):
        return records[:, :2].astype(np.int64, copy=False).reshape(-1, 2)
    if len(records) == 0:
        return np.zeros((0, 2), np.int64)
    return np.asarray([(r[0], r[1]) for r in records], dtype=np.int64)


def _first_seen(stacked: np.ndarray):
    """ids in order of first appearance + dict id -> rank."""
    if stacked.size == 0:
        return np.zeros(0, np.int64), {}
    uniq, first = np.unique(stacked, return_index=True)
    keys = uniq[np.argsort(first, kind="stable")]
    return keys, {int(k): i for i, k in enumerate(keys.tolist())}


def _nested(pairs: np.ndarray) -> Dict[int, Dict[int, float]]:
    out: Dict[int, Dict[int, float]] = defaultdict(dict)
    for u, i in pairs.tolist():
        out[u][i] = 1.0
    return out


def truth_pairs(data_set: Dict) -> int:
    """(user, item) pairs of a ground-truth dict {user: {item: 1.0}} -- the cheap fingerprint the caches are checked with."""
    return sum(map(len, data_set.values()))


class ColdStartDataBuilder(object):
    def __init__(self, training_data, warm_valid_data, cold_valid_data, overall_valid_data,
                 warm_test_data, cold_test_data, overall_test_data, user_num, item_num,
                 warm_user_idx, warm_item_idx, cold_user_idx, cold_item_idx,
                 user_content=None, item_content=None):
        self.user_num, self.item_num = int(user_num), int(item_num)
        self.training_data = training_data
        self.warm_valid_data, self.warm_test_data = warm_valid_data, warm_test_data
        self.cold_valid_data, self.cold_test_data = cold_valid_data, cold_test_data
        self.overall_valid_data, self.overall_test_data = overall_valid_data, overall_test_data

        parts = [_as_pairs(x) for x in (training_data, warm_valid_data, warm_test_data, cold_valid_data,
                                        cold_test_data, overall_valid_data, overall_test_data)]
        train = parts[0]
        allp = np.concatenate(parts, 0)
        self.user_keys, self.user = _first_seen(allp[:, 0])
        self.item_keys, self.item = _first_seen(allp[:, 1])
        self.id2user = {i: int(k) for i, k in enumerate(self.user_keys.tolist())}
        self.id2item = {i: int(k) for i, k in enumerate(self.item_keys.tolist())}

        # dict views the trainer API exposes (original ids, insertion order)
        self.training_set_u = _nested(train)
        self.training_set_i = _nested(train[:, ::-1])
        names = ("warm_valid", "warm_test", "cold_valid", "cold_test", "overall_valid", "overall_test")
        self._truth_csr = {}
        for name, p in zip(names, parts[1:]):
            setattr(self, name + "_set", _nested(p))
            setattr(self, name + "_set_item", set(p[:, 1].tolist()))
            # the same ground truth as arrays (users in dict order, CSR of internal item ids), built here -- vectorised --
            # instead of walking the nested dict inside the first timed validation (evaluator.truth_csr: 30 ms at
            # MovieLens size); keyed by the dict object the trainers pass around
            dset = getattr(self, name + "_set")
            csr = self._pairs_truth_csr(p)
            self._truth_csr[id(dset)] = (dset, csr, int(csr[1][-1]))     # the dict itself is kept: ids can be reused

        # array views for the HIP path
        self.train_u = self.map_users(train[:, 0]).astype(np.int32)
        self.train_i = self.map_items(train[:, 1]).astype(np.int32)
        order = np.lexsort((self.train_i, self.train_u))
        su, si = self.train_u[order], self.train_i[order]
        keep = np.ones(su.shape[0], bool)
        keep[1:] = (su[1:] != su[:-1]) | (si[1:] != si[:-1])
        su, si = su[keep], si[keep]
        self.rated_rowptr = np.zeros(self.user_num + 1, np.int64)
        np.add.at(self.rated_rowptr, su.astype(np.int64) + 1, 1)
        np.cumsum(self.rated_rowptr, out=self.rated_rowptr)
        self.rated_col = si.astype(np.int32)
        # the reference keeps a numpy object array of sets of ORIGINAL item ids per internal uid
        self.training_set_uid = np.array([set() for _ in range(self.user_num)])
        for u, items in self.training_set_u.items():
            self.training_set_uid[self.user[u]] = set(items)

        self.source_user_content, self.source_item_content = user_content, item_content
        self.mapped_user_content = self._map_content(user_content, self.user_keys, self.user_num, "user")
        self.mapped_item_content = self._map_content(item_content, self.item_keys, self.item_num, "item")

        self.source_warm_user_idx, self.source_warm_item_idx = warm_user_idx, warm_item_idx
        self.source_cold_user_idx, self.source_cold_item_idx = cold_user_idx, cold_item_idx
        self.mapped_warm_user_idx = self.get_user_id_list(warm_user_idx)
        self.mapped_warm_item_idx = self.get_item_id_list(warm_item_idx)
        self.mapped_cold_user_idx = self.get_user_id_list(cold_user_idx)
        self.mapped_cold_item_idx = self.get_item_id_list(cold_item_idx)

        self.ui_adj = self.create_sparse_complete_bipartite_adjacency()
        self.norm_adj = self.normalize_graph_mat(self.ui_adj)
        self.interaction_mat = self.create_sparse_interaction_matrix()
        self._sampler = None

    def _pairs_truth_csr(self, p: np.ndarray):
        """evaluator.truth_csr(_nested(p), item_of=self.item) without the dict walk: users in order of first appearance,
        each user's DISTINCT items in order of first appearance, as (users, rowptr, internal item ids)."""
        if p.shape[0] == 0:
            return [], np.zeros(1, np.int64), np.zeros(0, np.int64)
        _, first = np.unique(p, axis=0, return_index=True)           # one entry per distinct (user, item): its first row
        first = np.sort(first)
        pu, pi = p[first, 0], p[first, 1]
        ukeys, urank = np.unique(pu, return_inverse=True)
        ufirst = np.full(len(ukeys), len(pu), np.int64)
        np.minimum.at(ufirst, urank, np.arange(len(pu)))
        uorder = np.argsort(ufirst, kind="stable")                    # users by first appearance
        pos = np.empty(len(ukeys), np.int64)
        pos[uorder] = np.arange(len(ukeys))
        order = np.argsort(pos[urank], kind="stable")                 # group by user, first-appearance order inside
        rowptr = np.zeros(len(ukeys) + 1, np.int64)
        np.cumsum(np.bincount(pos[urank], minlength=len(ukeys)), out=rowptr[1:])
        sorter = np.argsort(self.item_keys, kind="stable")
        items = sorter[np.searchsorted(self.item_keys, pi[order], sorter=sorter)].astype(np.int64)
        return ukeys[uorder].tolist(), rowptr, items

    def truth_csr_cached(self, data_set):
        """(users, rowptr, internal item ids) of one of this builder's own valid / test sets, or None for any other dict."""
        hit = self._truth_csr.get(id(data_set))
        if hit is None or hit[0] is not data_set:
            return None
        # a plugin may have filtered or extended the dict in place since: the arrays stand only while the user count and
        # the (user, item) pair count are what they were; otherwise the caller walks the dict it was given
        if len(data_set) != len(hit[1][0]) or truth_pairs(data_set) != hit[2]:
            return None
        return hit[1]

    def truth_csr_invalidate(self):
        """Forget the arrays of this builder's valid / test dicts (BaseColdStartTrainer.invalidate_eval_cache)."""
        self._truth_csr.clear()

    # ------------------------------------------------------------------ id mapping
    def _map(self, table: Dict[int, int], ids, what: str) -> np.ndarray:
        ids = np.asarray(ids).reshape(-1)
        try:
            return np.fromiter((table[int(x)] for x in ids.tolist()), dtype=np.int64, count=ids.shape[0])
        except KeyError as e:
            raise Exception(f"{what} {e.args[0]} not in current id table") from None

    def map_users(self, ids) -> np.ndarray:
        return self._map(self.user, ids, "user")

    def map_items(self, ids) -> np.ndarray:
        return self._map(self.item, ids, "item")

    def get_user_id(self, u):
        if u in self.user:
            return self.user[u]
        raise Exception(f"user {u} not in current id table")

    def get_item_id(self, i):
        if i in self.item:
            return self.item[i]
        raise Exception(f"item {i} not in current id table")

    def get_user_id_list(self, u_list):
        return self.map_users(u_list)

    def get_item_id_list(self, i_list):
        return self.map_items(i_list)

    def _map_content(self, content, keys, n_rows, what):
        if content is None:
            return None
        setattr(self, f"{what}_content_dim", content.shape[-1])
        rows = max(int(n_rows), int(content.shape[0]), len(keys))
        out = np.empty((rows, content.shape[1]))
        out[: len(keys)] = content[keys]
        return out

    # ------------------------------------------------------------------ graph (A5)
    def create_sparse_complete_bipartite_adjacency(self, self_connection=False):
        """A = [[0, R], [R^T, 0]] over user_num + item_num nodes (util/databuilder.py:220-234)."""
        n = self.user_num + self.item_num
        r = self.train_u.astype(np.int64)
        c = self.train_i.astype(np.int64) + self.user_num
        ones = np.ones(r.shape[0], np.float32)
        half = sp.csr_matrix((ones, (r, c)), shape=(n, n), dtype=np.float32)
        adj = half + half.T
        if self_connection:
            adj = adj + sp.eye(n)
        return adj

    def normalize_graph_mat(self, adj_mat):
        """D^-1/2 A D^-1/2 for square matrices, D^-1 A otherwise; zero-degree rows stay zero
        (util/databuilder.py:236-254).  fp32 throughout, like the reference."""
        adj_mat = sp.csr_matrix(adj_mat)
        rowsum = np.asarray(adj_mat.sum(1)).reshape(-1)
        power = -0.5 if adj_mat.shape[0] == adj_mat.shape[1] else -1.0
        d_inv = np.zeros_like(rowsum, dtype=np.float32)
        np.power(rowsum, power, out=d_inv, where=rowsum != 0)
        left = sp.diags(d_inv).dot(adj_mat)
        return left.dot(sp.diags(d_inv)) if power == -0.5 else left

    def convert_to_laplacian_mat(self, adj_mat):
        rows, cols = adj_mat.nonzero()
        n = adj_mat.shape[0] + adj_mat.shape[1]
        half = sp.csr_matrix((adj_mat.data, (rows, cols + adj_mat.shape[0])), shape=(n, n), dtype=np.float32)
        return self.normalize_graph_mat(half + half.T)

    def create_sparse_interaction_matrix(self):
        ones = np.ones(self.train_u.shape[0], np.float32)
        return sp.csr_matrix((ones, (self.train_u.astype(np.int64), self.train_i.astype(np.int64))),
                             shape=(self.user_num, self.item_num), dtype=np.float32)

    def norm_adj_csr(self):
        """(rowptr int64, col int32 ascending, val fp32) of ``norm_adj`` for crh_spmm_csr_f32."""
        m = sp.csr_matrix(self.norm_adj)
        m.sort_indices()
        return m.indptr.astype(np.int64), m.indices.astype(np.int32), m.data.astype(np.float32)

    # ------------------------------------------------------------------ sampler (A1)
    @property
    def sampler(self):
        if self._sampler is None:
            from ..sampler import PairwiseSampler
            self._sampler = PairwiseSampler(self.train_u, self.train_i, self.user_num, len(self.item))
            self._sampler.set_catalogue(len(self.user), self.mapped_cold_item_idx)
        return self._sampler

    # ------------------------------------------------------------------ small accessors of the reference
    def training_size(self):
        return len(self.user), len(self.item), len(self.training_data)

    def contain(self, u, i):
        return u in self.user and i in self.training_set_u[u]

    def contain_user(self, u):
        return u in self.user

    def contain_item(self, i):
        return i in self.item

    def user_rated(self, u):
        return list(self.training_set_u[u].keys()), list(self.training_set_u[u].values())

    def item_rated(self, i):
        return list(self.training_set_i[i].keys()), list(self.training_set_i[i].values())


def bipartite_norm_adj_csr(train_u, train_i, user_num: int, item_num: int):
    """``norm_adj`` straight from internal-id training pairs (what ``ColdStartDataBuilder`` computes at
    util/databuilder.py:86-87,220-254), as the (rowptr int64, col int32 ascending, val fp32) arrays
    crh_spmm_csr_f32 takes.  Same SciPy products as ``normalize_graph_mat`` above."""
    n = int(user_num) + int(item_num)
    r = np.asarray(train_u, np.int64)
    c = np.asarray(train_i, np.int64) + int(user_num)
    half = sp.csr_matrix((np.ones(r.shape[0], np.float32), (r, c)), shape=(n, n), dtype=np.float32)
    adj = sp.csr_matrix(half + half.T)
    rowsum = np.asarray(adj.sum(1)).reshape(-1)
    d_inv = np.zeros_like(rowsum, dtype=np.float32)
    np.power(rowsum, -0.5, out=d_inv, where=rowsum != 0)
    m = sp.csr_matrix(sp.diags(d_inv).dot(adj).dot(sp.diags(d_inv)))
    m.sort_indices()
    return m.indptr.astype(np.int64), m.indices.astype(np.int32), m.data.astype(np.float32)


class TorchGraphInterface(object):
    """``convert_sparse_mat_to_tensor`` keeps returning what model files pass to
    ``torch.sparse.mm(adj, dense)`` (model/LightGCN.py:76,90) -- a coalesced fp32 COO tensor --
    wrapped so that, on the GPU, the product runs in crh_spmm_csr_f32 (coldrec_amd/graph.py)."""

    @staticmethod
    def convert_sparse_mat_to_tensor(X):
        from ..graph import HipSparseAdj
        return HipSparseAdj.from_scipy(X)
