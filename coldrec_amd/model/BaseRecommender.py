"""Trainer base class with the reference's plugin contract (model/BaseRecommender.py:13-370) and a
fused GPU evaluation path.

Subclasses implement ``train / predict / batch_predict / save`` and publish ``self.user_emb`` /
``self.item_emb``; they inherit ``fast_evaluation`` (per-epoch validation, strict-NDCG early stop),
``valid / test / full_evaluation / run`` and the attribute names listed in SURVEY.md Appendix B.

Evaluation (``_evaluate``, reference lines 153-188) never materialises the (users x items) score
block when the plugin's ``batch_predict`` is the stock ``user_emb[users] @ item_emb.T``: users,
rated-item CSR and the warm/cold candidate bitmap are cached on the GPU per (set, type) and one
call of crh_score_topk_f32 returns the masked top-k.  Any other ``batch_predict`` (VBPR, ALDI, ...)
is honoured: its dense block goes through crh_mask_topk_f32.  Either way the order is the canonical
(score desc, index asc) one, and the metrics are computed on arrays (util/evaluator.py).
"""
from __future__ import annotations

import dis
import inspect
import math
import time
from abc import ABC, abstractmethod
from collections.abc import ItemsView, KeysView, ValuesView
from typing import Any, Dict, List, Tuple

import numpy as np
import torch

from .. import _lib, ops
from ..eval import ShardedTopK, shard_bounds
from ..train import dp_from_env
from ..util.evaluator import format_measure, ranking_metrics, truth_csr, truth_dense


def _truth_pairs(data_set: Dict) -> int:
    fast = getattr(data_set, 'n_pairs', None)         # array-backed ground truths (bench.py's S-EVAL leg) know their size
    return int(fast) if fast is not None else sum(map(len, data_set.values()))


def _structural(name):
    """a dict method that changes the key set (or reads all of it): RecList builds every list first"""
    def method(self, *a, **kw):
        self._materialise()
        return getattr(dict, name)(self, *a, **kw)
    method.__name__ = name
    return method


class RecList(dict):
    """What ``_evaluate`` / ``valid`` / ``test`` return: ``{user: [(original item id, np.float32 score), ...]}`` exactly as
    model/BaseRecommender.py:185-187 builds it -- a ``dict`` with the same keys in the same order and the same lists -- but
    held as the ``(users, k)`` score / id arrays the ranking kernel wrote.  A user's list is built when somebody first looks
    at it and KEPT (``__missing__``), so the object handed out is the object looked up later: a plugin that edits a list in
    place (``rec[u].sort()``, ``rec[u][:] = ...``, ``.pop()``) or assigns one (``rec[u] = [...]``) is scored on its edit, as
    with the reference's plain dict.  The eager dict was 2e6 Python tuples per 1e5 users (0.55 s beside 1.95 s of ranking);
    callers that only hand the result to ``full_evaluation`` (every reference model does) never pay for it: the metrics take
    the arrays, and walk only the lists that were handed out.  A structural edit (``del``, ``pop``, ``update``, a new key)
    turns it into the plain dict it imitates (every list built, ``_full``)."""

    def __init__(self, users, scores: np.ndarray, idx: np.ndarray, item_keys: np.ndarray):
        super().__init__()
        self.users, self.scores, self.idx, self._item_keys = users, scores, idx, item_keys
        self._row = None            # user -> row, built on first keyed access
        self._full = False          # every list is in the dict storage: an ordinary dict from then on

    def _rows(self):
        if self._row is None:
            self._row = {u: r for r, u in enumerate(self.users)}
        return self._row

    def _list(self, r: int) -> list:
        names = self._item_keys[np.minimum(self.idx[r], len(self._item_keys) - 1)].tolist()
        return list(zip(names, self.scores[r]))

    def __missing__(self, user):    # dict.__getitem__ lands here for a user whose list has not been handed out yet
        if self._full:
            raise KeyError(user)
        lst = self._list(self._rows()[user])
        dict.__setitem__(self, user, lst)
        return lst

    def _materialise(self):
        """every list into the dict storage, in the users' order; plain-dict behaviour afterwards"""
        if self._full:
            return
        handed = dict(dict.items(self))
        dict.clear(self)
        for r, u in enumerate(self.users):
            dict.__setitem__(self, u, handed[u] if u in handed else self._list(r))
        self._full = True

    def __setitem__(self, user, value):
        if not self._full and user not in self._rows():
            self._materialise()
        dict.__setitem__(self, user, value)

    def __iter__(self):
        return dict.__iter__(self) if self._full else iter(self.users)

    def __len__(self):
        return dict.__len__(self) if self._full else len(self.users)

    def __contains__(self, user):
        return dict.__contains__(self, user) if self._full else user in self._rows()

    def get(self, user, default=None):
        return self[user] if user in self else default

    def keys(self):
        return dict.keys(self) if self._full else KeysView(self)

    def items(self):
        return dict.items(self) if self._full else ItemsView(self)

    def values(self):
        return dict.values(self) if self._full else ValuesView(self)

    def __eq__(self, other):
        self._materialise()
        return dict.__eq__(self, other)

    __hash__ = None

    def __repr__(self):
        return dict.__repr__(self) if self._full else f"RecList({len(self.users)} users x {self.idx.shape[1] if self.idx.ndim == 2 else 0})"

    def copy(self):
        self._materialise()
        return dict(self)

    __delitem__, pop, popitem, clear = _structural('__delitem__'), _structural('pop'), _structural('popitem'), _structural('clear')
    update, setdefault, __ior__ = _structural('update'), _structural('setdefault'), _structural('__ior__')
    __or__, __ror__, __reversed__ = _structural('__or__'), _structural('__ror__'), _structural('__reversed__')

    @property
    def untouched(self) -> bool:
        """no list was handed out, replaced or removed: the arrays ARE the lists"""
        return not self._full and dict.__len__(self) == 0

    def handed_out(self):
        """(user, list) of every list somebody holds a reference to (and may have edited in place) or assigned"""
        return list(dict.items(self))


# ---- is a plugin's batch_predict the stock one (model/MF.py:58-63, copied 22-fold across model/*.py)?
# Decided on the function's BYTECODE against templates compiled here in the running interpreter: formatting, comments,
# docstrings and local names do not matter, a plugin shipped without source (.pyc) is recognised too, and anything that
# computes something else (a second product, a scale, a sign) differs in its instruction stream.
_STOCK_ITEM_T = ("self.item_emb.transpose(0, 1)", "self.item_emb.transpose(1, 0)", "self.item_emb.T", "self.item_emb.t()",
                 "self.item_emb.mT", "self.item_emb.permute(1, 0)", "torch.transpose(self.item_emb, 0, 1)",
                 "torch.t(self.item_emb)")
_STOCK_PRODUCT = ("torch.matmul({a}, {b})", "torch.mm({a}, {b})", "{a} @ {b}", "{a}.matmul({b})", "{a}.mm({b})")
_stock_codes = None


def _norm_code(code) -> tuple:
    """instruction stream with local names numbered by first use"""
    local: Dict[str, int] = {}
    out = []
    for ins in dis.get_instructions(code):
        arg = ins.argval
        if ins.opname in ('LOAD_FAST', 'STORE_FAST', 'DELETE_FAST'):
            arg = local.setdefault(arg, len(local))
        elif ins.opname.startswith(('JUMP', 'POP_JUMP', 'SETUP', 'FOR_ITER')):
            arg = ins.arg
        out.append((ins.opname, arg if isinstance(arg, (str, int, float, tuple, type(None))) else repr(arg)))
    return tuple(out)


def _stock_templates():
    global _stock_codes
    if _stock_codes is None:
        codes = set()
        for prod in _STOCK_PRODUCT:
            for bt in _STOCK_ITEM_T:
                expr = prod.format(a="self.user_emb[users]", b=bt)
                for tail in (f"score = {expr}\n{{i}}return score", f"return {expr}"):
                    for grad in (True, False):
                        ind = "        " if grad else "    "
                        for head in ("users = self.data.get_user_id_list(users)\n{i}users = torch.tensor(users, device=self.device)",
                                     "users = torch.as_tensor(self.data.get_user_id_list(users), device=self.device)"):
                            src = "def batch_predict(self, users):\n" + ("    with torch.no_grad():\n" if grad else "") + \
                                  ind + head.format(i=ind) + "\n" + ind + tail.format(i=ind) + "\n"
                            ns: Dict[str, Any] = {}
                            exec(compile(src, "<stock batch_predict>", "exec"), ns)
                            codes.add(_norm_code(ns["batch_predict"].__code__))
        _stock_codes = codes
    return _stock_codes


def _is_stock_batch_predict(fn) -> bool:
    """True when ``batch_predict`` is the 22-fold copy of model/MF.py:58-63, in any spelling of the same product
    (``torch.matmul`` / ``torch.mm`` / ``@``; ``.transpose(0, 1)`` / ``.T`` / ``.t()`` ...), with or without source."""
    try:
        code = inspect.unwrap(fn).__code__
    except (AttributeError, ValueError):
        return False
    return _norm_code(code) in _stock_templates()


class BaseColdStartTrainer(ABC):
    fused_eval = None     # subclasses may force True / False; None = detect from batch_predict's source
    EVAL_USER_BLOCK = 131072   # users per fused scoring call (see _topk_device)
    eval_timing = None    # a dict here makes _metrics record its phases (synchronising at each boundary): bench.py's eval_e2e

    def __init__(self, config):
        self.config = config
        self.args = config.args
        self.data = config.data
        self.device = config.device
        a = self.args
        self.bestPerformance = []
        self.topN = [int(x) for x in a.topN.split(',')]
        self.max_N = max(self.topN)
        if self.max_N > _lib.MAX_K or min(self.topN) < 1:
            raise ValueError(f'--topN {a.topN}: the HIP ranking kernels keep lists of 1..{_lib.MAX_K} entries '
                             f'(CRH_MAX_K in include/coldrec_hip.h)')
        self.model_name, self.dataset_name = a.model, a.dataset
        self.emb_size, self.maxEpoch, self.batch_size = a.emb_size, a.epochs, a.bs
        self.lr, self.reg = a.lr, a.reg
        self.result = []
        self.early_stop_flag = a.early_stop != 0
        if self.early_stop_flag:
            self.early_stop_patience = self.max_early_stop_patience = a.early_stop
        self.epochs_ran = 0
        self.eval_every = max(1, int(getattr(a, 'eval_every', 1)))
        self._eval_cache: Dict[Any, Dict[str, Any]] = {}
        ops.warm_up(self.device)          # the library's device code is loaded here, not inside the first timed epoch

    # ------------------------------------------------------------------ plugin contract
    @abstractmethod
    def train(self) -> None: ...

    @abstractmethod
    def predict(self, u): ...

    @abstractmethod
    def batch_predict(self, users): ...

    @abstractmethod
    def save(self) -> None: ...

    def print_basic_info(self):
        print('*' * 80)
        for label, v in (('Model: ', self.model_name), ('Dataset: ', self.dataset_name),
                         ('Embedding Dimension:', self.emb_size), ('Maximum Epoch:', self.maxEpoch),
                         ('Learning Rate:', self.lr), ('Batch Size:', self.batch_size)):
            print(label, v)
        print('*' * 80)

    def timer(self, start=True):
        if start:
            self.train_start_time = time.time()
        else:
            self.train_end_time = time.time()

    # ------------------------------------------------------------------ evaluation
    def _sets(self, kind: str, which: str) -> Dict:
        table = {'warm': f'warm_{kind}_set', 'cold': f'cold_{kind}_set', 'all': f'overall_{kind}_set'}
        if which not in table:
            raise ValueError(f'Invalid {"valid" if kind == "valid" else kind} type!')
        return getattr(self.data, table[which])

    def _get_eval_cache(self, data_set: Dict, data_type: str) -> Dict[str, Any]:
        key = (id(data_set), data_type, str(self.device), self.args.cold_object)
        hit = self._eval_cache.get(key)
        # the reference caches the user list and the masks per dataset object but reads the ground truth from the dict at
        # every evaluation (util/evaluator.py:153-187): a dict a plugin changed in place must not be scored against the
        # arrays of its old contents -- the entry stands while user and pair counts are unchanged
        if hit is not None and hit['fingerprint'] == (len(data_set), _truth_pairs(data_set)):
            return hit
        d = self.data
        cached = d.truth_csr_cached(data_set) if hasattr(d, 'truth_csr_cached') else None
        users, gt_rowptr, gt_items = cached if cached is not None else truth_csr(data_set, item_of=d.item)
        uint = d.get_user_id_list(users)
        lens = d.rated_rowptr[uint + 1] - d.rated_rowptr[uint]
        rowptr = np.zeros(len(users) + 1, np.int64)
        np.cumsum(lens, out=rowptr[1:])
        take = np.repeat(d.rated_rowptr[uint] - rowptr[:-1], lens) + np.arange(int(rowptr[-1]))
        col = d.rated_col[take]
        masked = None
        if self.args.cold_object == 'item':           # reference lines 130-143
            if data_type == 'warm':
                masked = np.asarray(d.mapped_cold_item_idx)
            elif data_type == 'cold':
                masked = np.asarray(d.mapped_warm_item_idx)
        dev = self.device
        hit = {
            'users': users, 'users_int': torch.from_numpy(uint.astype(np.int32)).to(dev),
            'rated_rowptr': torch.from_numpy(rowptr).to(dev) if rowptr[-1] else None, 'rated_rowptr_host': rowptr,
            'rated_col': torch.from_numpy(np.ascontiguousarray(col)).to(dev) if rowptr[-1] else None,
            'bitmap': ops.make_bitmap(d.item_num, masked, dev),
            'gt_rowptr': gt_rowptr, 'gt_items': gt_items, 'fingerprint': (len(users), int(gt_rowptr[-1])),
            'gt_dense': truth_dense(gt_rowptr, gt_items, len(d.item)),
        }
        self._eval_cache[key] = hit
        return hit

    def _topk_arrays(self, data_set: Dict, data_type: str):
        """(users, scores (n,k) float32, internal item ids (n,k)) on the host, canonical order."""
        c, s, i = self._topk_device(data_set, data_type)
        if s is None:
            return c, np.zeros((0, self.max_N), np.float32), np.zeros((0, self.max_N), np.int32)
        return c, s.cpu().numpy(), i.cpu().numpy()

    def invalidate_eval_cache(self):
        """Drop the per-dataset evaluation arrays.  The cache entry of a ground-truth dict stands while its user and pair COUNTS
        are unchanged (the reference re-reads the dict at every evaluation); a plugin that swaps users or items in place with
        the same counts calls this."""
        self._eval_cache.clear()
        if hasattr(self.data, 'truth_csr_invalidate'):
            self.data.truth_csr_invalidate()

    def _topk_device(self, data_set: Dict, data_type: str, c=None):
        """(cache, scores (n,k), internal item ids (n,k)) as device tensors; (cache, None, None) for an empty split.
        c: the cache entry when the caller already looked it up (the lookup walks the dict: O(users) of Python)."""
        if c is None:
            c = self._get_eval_cache(data_set, data_type)
        if len(c['users']) == 0:        # an empty warm / cold / valid split: the reference reports zeros, no kernel runs
            return c, None, None
        fused = self.fused_eval
        if fused is None:
            fused = _is_stock_batch_predict(type(self).batch_predict)
        ue, ie = getattr(self, 'user_emb', None), getattr(self, 'item_emb', None)
        stock = bool(fused)
        fused = fused and torch.is_tensor(ue) and torch.is_tensor(ie) and ue.is_cuda and ue.dim() == 2
        if not getattr(self, '_route_told', False):
            self._route_told = True
            self._tell_route(fused, stock, c)
        if fused:
            dp = dp_from_env()
            tdt = torch.float16 if getattr(self.args, 'score_dtype', 'fp32') == 'fp16' else torch.float32
            ue, ie = ue.detach().to(tdt), ie.detach().to(tdt)
            if dp is None:
                def rank(users, rp, rc, out=None):
                    return ops.score_topk(ue, users, ie, self.max_N, rp, rc, c['bitmap'], out=out)
            else:                     # item rows sharded over the ranks, all-gather + canonical merge
                lo, hi = shard_bounds(ie.shape[0], dp.world, dp.rank)
                eng = ShardedTopK(ie[lo:hi].contiguous(), lo, ie.shape[0], self.max_N, dp.world, dp.rank)

                def rank(users, rp, rc, out=None):
                    return eng.topk(ue, users, rp, rc, c['bitmap'])
            n, blk = len(c['users']), self.EVAL_USER_BLOCK
            if n <= blk:
                s, i = rank(c['users_int'], c['rated_rowptr'], c['rated_col'])
            else:
                # catalogue-scale evaluations (S-EVAL: 1e6 users) go in blocks of EVAL_USER_BLOCK users: the kernel's
                # partial-list workspace is 64 x users x k x 8 bytes (10 GB for 1e6 users at once), and 2048 groups of 64
                # users already fill the chip -- one block is one launch at the headline's shape
                s = torch.empty((n, self.max_N), dtype=torch.float32, device=ue.device)
                i = torch.empty((n, self.max_N), dtype=torch.int32, device=ue.device)
                rp_all, rc_all, rp_host = c['rated_rowptr'], c['rated_col'], c['rated_rowptr_host']
                for b0 in range(0, n, blk):
                    b1 = min(n, b0 + blk)
                    rp = rc = None
                    if rp_all is not None:
                        rp = (rp_all[b0:b1 + 1] - rp_all[b0]).contiguous()
                        rc = rc_all[int(rp_host[b0]):int(rp_host[b1])]
                    bs_, bi_ = rank(c['users_int'][b0:b1], rp, rc, out=(s[b0:b1], i[b0:b1]))
                    if bs_.data_ptr() != s[b0:b1].data_ptr():          # the sharded engine returns its own (merged) block
                        s[b0:b1], i[b0:b1] = bs_, bi_
        else:
            parts_s, parts_i = [], []
            for lo in range(0, len(c['users']), self.batch_size):
                hi = min(lo + self.batch_size, len(c['users']))
                block = self.batch_predict(c['users'][lo:hi])
                block = torch.as_tensor(block, device=self.device).float().contiguous()
                rp = rc = None
                if c['rated_rowptr'] is not None:
                    rp = (c['rated_rowptr'][lo:hi + 1] - c['rated_rowptr'][lo]).contiguous()
                    rc = c['rated_col'][int(c['rated_rowptr'][lo]):int(c['rated_rowptr'][hi])].contiguous()
                s, i = ops.mask_topk(block, self.max_N, rp, rc, c['bitmap'], write_back=True)
                parts_s.append(s)
                parts_i.append(i)
            s, i = torch.cat(parts_s), torch.cat(parts_i)
        return c, s, i

    DENSE_BLOCK_WARN_BYTES = 4 << 30

    def _tell_route(self, fused: bool, stock: bool, c) -> None:
        """One line at a trainer's first evaluation: which route ranks -- the fused kernel the LIBRARY names for the shape
        (crh_score_topk_route) or batch_predict's dense block -- and a warning when that block is catalogue-sized."""
        n_items = int(self.data.item_num) if hasattr(self.data, 'item_num') else len(self.data.item)
        if fused:
            ie = self.item_emb
            half = getattr(self.args, 'score_dtype', 'fp32') == 'fp16'
            r = ops.score_topk_route(min(len(c['users']), self.EVAL_USER_BLOCK), int(ie.shape[0]), int(ie.shape[1]), self.max_N,
                                     half=half, has_bitmap=c['bitmap'] is not None)
            form = f", {r['dma_form']} form" if r.get('dma_form') else ''
            print(f"Evaluation route: fused HIP scoring + masks + top-{self.max_N} ({r['kernel']}, {r['route']}{form}"
                  f"{', seeded from a %d-item prefix' % r['prefix_items'] if r['seeded'] else ''}); no score block is written")
            return
        block = int(self.batch_size) * n_items * 4
        why = ("user_emb / item_emb are not 2-D device tensors" if stock else
               "batch_predict is not the stock user_emb[users] @ item_emb.T (set fused_eval = True on the trainer if it is)")
        print(f"Evaluation route: batch_predict -> ({self.batch_size} x {n_items}) fp32 score block "
              f"({block / 2 ** 30:.2f} GiB) -> crh_mask_topk_f32; {why}")
        if block > self.DENSE_BLOCK_WARN_BYTES:
            import warnings
            warnings.warn(f"evaluation materialises a {block / 2 ** 30:.1f} GiB score block per {self.batch_size} users: "
                          f"lower --bs, or set fused_eval = True if batch_predict is user_emb[users] @ item_emb.T", RuntimeWarning)

    def _evaluate(self, data_set: Dict, data_type: str = 'all') -> Dict[Any, List[Tuple[Any, float]]]:
        t0 = self._tick('_before', time.perf_counter())
        c, s, i = self._topk_arrays(data_set, data_type)
        t0 = self._tick('evaluate_rank_s', t0)
        # {user: [(original item id, np.float32 score), ...]} as model/BaseRecommender.py:185-187 builds it, as a Mapping over
        # the arrays that materialises a user's list on access (RecList)
        out = RecList(c['users'], s, i, self.data.item_keys)
        self._tick('evaluate_dict_s', t0)
        return out

    def valid(self, valid_type: str = 'all'):
        return self._evaluate(self._sets('valid', valid_type), valid_type)

    def test(self, test_type: str = 'all'):
        return self._evaluate(self._sets('test', test_type), test_type)

    def _tick(self, name: str, t0: float) -> float:
        """eval_timing hook: seconds since t0 under ``name`` (device drained first); no-op without the hook."""
        if self.eval_timing is None:
            return t0
        torch.cuda.synchronize()
        now = time.perf_counter()
        self.eval_timing[name] = self.eval_timing.get(name, 0.0) + now - t0
        return now

    def _metrics(self, data_set: Dict, data_type: str, topn):
        t0 = self._tick('_before', time.perf_counter())
        c = self._get_eval_cache(data_set, data_type)
        t0 = self._tick('cache_s', t0)
        c, _s, i = self._topk_device(data_set, data_type, c)
        t0 = self._tick('rank_s', t0)
        if self.eval_timing is not None:
            self.eval_timing['last_topk'] = (_s, i)
        try:
            return self._metrics_of(c, i, topn)
        finally:
            self._tick('metrics_s', t0)

    def _metrics_of(self, c, i, topn):
        if i is None:
            return ranking_metrics(c['gt_rowptr'], c['gt_items'], np.zeros((0, self.max_N), np.int32), topn)
        t0 = time.perf_counter()
        hit = self._membership(c, i)
        t0 = self._tick('membership_s', t0)
        try:
            return ranking_metrics(c['gt_rowptr'], c['gt_items'], None, topn, hit=hit)
        finally:
            self._tick('host_metrics_s', t0)

    def _membership(self, c, i) -> np.ndarray:
        """bool (users, k) on the host: is prediction (r, q) in the ground truth of user r -- tested on the GPU, one bit per
        prediction comes back instead of scores + ids and a host gather (the host part of a MovieLens-size validation was
        5-6 ms per epoch, twice the epoch's training time)."""
        if c['gt_dense'] is not None:
            # small catalogues: a dense (users, items) truth table on the device, one gather
            if c.get('gt_dense_dev') is None:
                c['gt_dense_dev'] = torch.from_numpy(c['gt_dense']).to(i.device)
                c['row_ids_dev'] = torch.arange(c['gt_dense'].shape[0], device=i.device).unsqueeze(1)
            idx = i.long()
            ok = (idx >= 0) & (idx < c['gt_dense_dev'].shape[1])
            return (c['gt_dense_dev'][c['row_ids_dev'], idx.clamp(0, c['gt_dense_dev'].shape[1] - 1)] & ok).cpu().numpy()
        # ground truth too large for a dense table (S-EVAL: 1e6 users x 1e7 items): sorted (row, item) keys on the device,
        # one binary search per prediction, in blocks of users (the int64 key arrays of 2e7 predictions at once are 0.5 GB)
        if c.get('gt_keys_dev') is None:
            n_items = len(self.data.item)
            rows = np.repeat(np.arange(len(c['users']), dtype=np.int64), np.diff(c['gt_rowptr']))
            gi = np.asarray(c['gt_items'], np.int64)
            keys = (rows * n_items + np.clip(gi, 0, n_items - 1))[gi < n_items]
            c['gt_keys_dev'] = torch.sort(torch.from_numpy(keys).to(i.device))[0]
            c['gt_base'] = n_items
        keys, base = c['gt_keys_dev'], c['gt_base']
        n = i.shape[0]
        hit = torch.zeros(i.shape, dtype=torch.bool, device=i.device)
        if keys.numel():
            blk = 1 << 18
            for lo in range(0, n, blk):
                idx = i[lo:lo + blk].long()
                ok = (idx >= 0) & (idx < base)
                rows = torch.arange(lo, lo + idx.shape[0], device=i.device).unsqueeze(1)
                pk = (rows * base + idx.clamp(0, base - 1)).reshape(-1)
                pos = torch.searchsorted(keys, pk).clamp_(max=keys.numel() - 1)
                hit[lo:lo + blk] = (keys[pos] == pk).reshape(idx.shape) & ok
        return hit.cpu().numpy()

    def _metrics_from_rec_list(self, data_set: Dict, data_type: str, rec_list: Dict, topn):
        """Metrics of a caller-supplied ``{user: [(item, score), ...]}`` (what ``test()`` returns, possibly
        post-processed by a plugin) -- the reference's ranking_evaluation(test_set, rec_list, topN) on arrays."""
        c = self._get_eval_cache(data_set, data_type)
        item_id = self.data.item
        pad = np.iinfo(np.int32).max
        if isinstance(rec_list, RecList) and not rec_list._full and rec_list.idx.shape[0] == len(c['users']) and \
                rec_list.idx.shape[1] >= max(topn) and list(rec_list.users) == list(c['users']):
            # the lists are the ranking kernel's own arrays (internal ids): no Python tuple is built on the way to the metrics.
            # Lists somebody was handed (and may have edited in place) or assigned are read back from the lists themselves
            pred = rec_list.idx.astype(np.int64)
            if not rec_list.untouched:
                rows = rec_list._rows()
                for u, lst in rec_list.handed_out():
                    row = [item_id.get(it, pad) for it, _ in lst[: pred.shape[1]]]
                    pred[rows[u]] = row + [pad] * (pred.shape[1] - len(row))
            return ranking_metrics(c['gt_rowptr'], c['gt_items'], pred, topn, dense=c['gt_dense'])
        pred = np.full((len(c['users']), max(topn)), pad, np.int64)
        for r, u in enumerate(c['users']):
            row = [item_id.get(it, pad) for it, _ in rec_list[u][: max(topn)]]
            pred[r, : len(row)] = row
        return ranking_metrics(c['gt_rowptr'], c['gt_items'], pred, topn, dense=c['gt_dense'])

    def full_evaluation(self, rec_list=None, test_type: str = 'warm') -> None:
        """Prints and stores the test metrics (model/BaseRecommender.py:230-254).  With a ``rec_list`` (the dict
        ``test()`` returns, or a plugin's post-processed version of it) the numbers are computed FROM it, as the
        reference does; ``rec_list=None`` (the built-in ``run()``) ranks once on the GPU and scores the arrays."""
        test_set = self._sets('test', test_type)
        if rec_list is not None and len(rec_list) != len(test_set):
            print(f"ground-truth set size: {len(test_set)}, predicted set size: {len(rec_list)}")
            print('The Lengths of ground-truth set and predicted set do not match!')
            exit(-1)
        if rec_list is not None:
            perf = self._metrics_from_rec_list(test_set, test_type, rec_list, self.topN)
        else:
            perf = self._metrics(test_set, test_type, self.topN)
        self.result = format_measure(perf, self.topN)
        setattr(self, {'warm': 'warm_test_results', 'cold': 'cold_test_results',
                       'all': 'overall_test_results'}[test_type], perf)
        print('*' * 80)
        print(f'[{test_type} setting] The result of %s:\n%s' % (self.model_name, ''.join(self.result)))

    @staticmethod
    def _metrics_dict_from_measure(measure: List[str]) -> Dict[str, float]:
        return {k: float(v) for k, v in (m.strip().split(':') for m in measure[1:])}

    @staticmethod
    def _metrics_all_finite(performance: Dict[str, float]) -> bool:
        return all(math.isfinite(v) for v in performance.values())

    def fast_evaluation(self, epoch: int, valid_type: str = 'all') -> List[str]:
        """Validation at max(topN); strict NDCG improvement saves and resets patience, anything else
        (equal, worse, non-finite) costs one unit of patience (reference lines 268-351)."""
        valid_set = self._sets('valid', valid_type)
        print(f'Evaluating the model under the {valid_type} setting...')
        measure = format_measure(self._metrics(valid_set, valid_type, [self.max_N]), [self.max_N])
        performance = self._metrics_dict_from_measure(measure)
        finite = self._metrics_all_finite(performance)
        improved = False
        if not self.bestPerformance:
            if finite:
                self.bestPerformance = [epoch + 1, performance]
                self.save()
                improved = None                       # first checkpoint: patience untouched
            else:
                print('Warning: first validation has non-finite metrics; best checkpoint not initialized yet.')
        elif not finite:
            print('Warning: validation metrics are non-finite; early-stop patience decreased, '
                  'best checkpoint unchanged.')
        elif performance['NDCG'] > self.bestPerformance[1]['NDCG']:
            self.bestPerformance = [epoch + 1, performance]
            self.save()
            improved = True
        if self.early_stop_flag and improved is not None:
            if improved:
                self.early_stop_patience = self.max_early_stop_patience
            else:
                self.early_stop_patience -= 1

        print('-' * 120)
        print('Performance ' + ' (Top-' + str(self.max_N) + ' Recommendation)')
        measure_lines = [m.strip() for m in measure[1:]]
        print('*Current Performance*')
        print('Epoch:', str(epoch + 1) + ',', '  |  '.join(measure_lines))
        if self.bestPerformance:
            bp = '  |  '.join(f'{k}:{self.bestPerformance[1][k]}' for k in ('Hit Ratio', 'Precision', 'Recall', 'NDCG'))
            print(f'*Best {valid_type} Performance* ')
            print('Epoch:', str(self.bestPerformance[0]) + ',', bp)
        else:
            print(f'*Best {valid_type} Performance* not initialized (waiting for finite validation).')
        if self.early_stop_flag:
            if self.early_stop_patience <= 0:
                print(f"Stopping early at epoch {epoch + 1}.")
            else:
                print(f"Early stopping patience left: {self.early_stop_patience}.")
        print('-' * 120)
        return measure_lines

    def run(self) -> None:
        self.print_basic_info()
        print('Training Model...')
        self.train()
        if getattr(self, 'epochs_ran', 0) == 0 and self.maxEpoch > 0:
            self.epochs_ran = self.maxEpoch
        for test_type in ['all', 'cold', 'warm']:
            print('*' * 80)
            print(f'Testing under [{test_type}] setting...')
            print(f'Evaluating under [{test_type}] setting...')
            self.full_evaluation(None, test_type=test_type)
