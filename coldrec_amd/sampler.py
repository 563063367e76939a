"""Host sampler front end (C++ MT19937 restatement in csrc/sampler.hip, no GPU involved).

``PairwiseSampler`` produces the (user, positive, negative) triples of one epoch exactly as
util/utils.py:123-157 would -- same NumPy legacy stream, same cumulative shuffle -- in
milliseconds instead of seconds, as three int32 arrays ready for one host-to-device copy.
"""
from __future__ import annotations

import ctypes
import math
import os
import random
import threading

import numpy as np

from . import _lib


class PairwiseSampler:
    def __init__(self, rec_user, rec_item, n_users: int, n_items_seen: int):
        ru = np.ascontiguousarray(rec_user, dtype=np.int32)
        ri = np.ascontiguousarray(rec_item, dtype=np.int32)
        if ru.shape != ri.shape or ru.ndim != 1:
            raise ValueError("rec_user / rec_item must be 1-D arrays of equal length")
        self._L = _lib.lib()
        self._h = self._L.crh_sampler_create(ru.ctypes.data, ri.ctypes.data, ru.shape[0], int(n_users),
                                             int(n_items_seen))
        if not self._h:
            raise RuntimeError("crh_sampler_create failed: " + self._L.crh_last_error().decode())
        self.n_records = int(ru.shape[0])
        self._n_items = int(n_items_seen)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.crh_sampler_destroy(h)

    def seed(self, seed: int) -> None:
        """Same stream as ``np.random.seed(seed)`` (util/utils.py:342)."""
        _lib.check(self._L.crh_sampler_seed(self._h, int(seed) & 0xFFFFFFFF), "crh_sampler_seed")

    def pull_numpy_state(self) -> None:
        """Adopt NumPy's global legacy RNG state (the reference samples from that stream)."""
        name, key, pos, _hg, _cg = np.random.get_state()
        assert name == "MT19937"
        key = np.ascontiguousarray(key, dtype=np.uint32)
        _lib.check(self._L.crh_sampler_set_state(self._h, key.ctypes.data, int(pos)), "crh_sampler_set_state")

    def push_numpy_state(self) -> None:
        """Write the advanced state back so later np.random users see the reference's stream."""
        key = np.empty(624, dtype=np.uint32)
        pos = ctypes.c_int(0)
        _lib.check(self._L.crh_sampler_get_state(self._h, key.ctypes.data, ctypes.addressof(pos)),
                   "crh_sampler_get_state")
        st = np.random.get_state()
        np.random.set_state((st[0], key, int(pos.value), st[3], st[4]))

    def epoch(self, batch_size: int):
        """One epoch of triples: three int32 arrays of n_records (batches concatenated)."""
        u = np.empty(self.n_records, np.int32)
        i = np.empty(self.n_records, np.int32)
        j = np.empty(self.n_records, np.int32)
        _lib.check(self._L.crh_sampler_epoch(self._h, int(batch_size), u.ctypes.data, i.ctypes.data,
                                             j.ctypes.data), "crh_sampler_epoch")
        return u, i, j

    # ------------------------------------------------------------------ the other samplers
    def set_catalogue(self, n_users_seen: int, cold_item_idx=None) -> None:
        """``len(data.user)`` and ``data.mapped_cold_item_idx`` (util/utils.py:176, 198-199, 243-246)."""
        flags = np.zeros(self._n_items, np.uint8)
        if cold_item_idx is not None and len(cold_item_idx):
            flags[np.asarray(cold_item_idx, dtype=np.int64)] = 1
        _lib.check(self._L.crh_sampler_set_catalogue(self._h, int(n_users_seen), flags.ctypes.data),
                   "crh_sampler_set_catalogue")
        self._has_catalogue = True

    def pull_python_state(self) -> None:
        """Adopt the state of CPython's global ``random`` module (version-3 state: 624 words + position)."""
        ver, internal, _gauss = random.getstate()
        assert ver == 3 and len(internal) == 625
        key = np.array(internal[:624], dtype=np.uint32)
        _lib.check(self._L.crh_sampler_set_py_state(self._h, key.ctypes.data, int(internal[624])),
                   "crh_sampler_set_py_state")

    def push_python_state(self) -> None:
        key = np.empty(624, dtype=np.uint32)
        pos = ctypes.c_int(0)
        _lib.check(self._L.crh_sampler_get_py_state(self._h, key.ctypes.data, ctypes.addressof(pos)),
                   "crh_sampler_get_py_state")
        st = random.getstate()
        random.setstate((st[0], tuple(int(x) for x in key) + (int(pos.value),), st[2]))

    def _need_catalogue(self):
        if not getattr(self, "_has_catalogue", False):
            raise RuntimeError("PairwiseSampler.set_catalogue(n_users_seen, cold_item_idx) has not been called")

    def _check_value(self, rc: int, what: str) -> None:
        """The reference raises ValueError for empty / too small candidate pools (util/utils.py:201-204, 223-227)."""
        if rc != 0:
            msg = self._L.crh_last_error().decode("utf-8", "replace")
            if msg.startswith("next_batch_pairwise_"):
                raise ValueError(msg)
            raise RuntimeError(f"{what} failed (code {rc}): {msg}")

    def epoch_lara(self, n_negs: int = 1):
        """util/utils.py:160-188 -> user (n,), item (n,), neg_user (n, n_negs), neg_item (n, n_negs)."""
        self._need_catalogue()
        n = self.n_records
        u, i = np.empty(n, np.int32), np.empty(n, np.int32)
        nu, ni = np.empty((n, n_negs), np.int32), np.empty((n, n_negs), np.int32)
        self._check_value(self._L.crh_sampler_epoch_lara(self._h, int(n_negs), u.ctypes.data, i.ctypes.data,
                                                         nu.ctypes.data, ni.ctypes.data), "crh_sampler_epoch_lara")
        return u, i, nu, ni

    @staticmethod
    def sample_setsize(k: int) -> int:
        """The population size up to which random.sample copies the population (CPython Lib/random.py)."""
        setsize = 21
        if k > 5:
            setsize += 4 ** math.ceil(math.log(k * 3, 4))
        return setsize

    def epoch_clcrec(self, n_negs: int = 1):
        """util/utils.py:191-233 -> user (n,), item (n, 1 + n_negs): positive first, then the sampled negatives."""
        self._need_catalogue()
        n = self.n_records
        u, it = np.empty(n, np.int32), np.empty((n, 1 + n_negs), np.int32)
        self._check_value(self._L.crh_sampler_epoch_clcrec(self._h, int(n_negs), self.sample_setsize(int(n_negs)),
                                                           u.ctypes.data, it.ctypes.data), "crh_sampler_epoch_clcrec")
        return u, it

    def epoch_ccfcrec(self, positive_number: int, negative_number: int, self_neg_number: int):
        """util/utils.py:237-300 -> user, item, neg_user (n,), pos_items (n, P), neg_items (n, P, N), self_neg (n, S)."""
        self._need_catalogue()
        n, P, N, S = self.n_records, int(positive_number), int(negative_number), int(self_neg_number)
        u, i, nu = (np.empty(n, np.int32) for _ in range(3))
        pos, neg, sneg = np.empty((n, P), np.int32), np.empty((n, P, N), np.int32), np.empty((n, S), np.int32)
        self._check_value(self._L.crh_sampler_epoch_ccfcrec(self._h, P, N, S, u.ctypes.data, i.ctypes.data,
                                                            nu.ctypes.data, pos.ctypes.data, neg.ctypes.data,
                                                            sneg.ctypes.data), "crh_sampler_epoch_ccfcrec")
        return u, i, nu, pos, neg, sneg

    def epoch_cgrc(self, batch_size: int, ranking_neg_per_user: int = 32):
        """util/utils.py:303-336 -> user (n,), item (n,), bset_ptr (n_batches + 1,), bset (concatenated item sets)."""
        n, bs, R = self.n_records, int(batch_size), int(ranking_neg_per_user)
        nb = (n + bs - 1) // bs
        cap = nb * min(self._n_items, bs * (1 + R))
        u, i = np.empty(n, np.int32), np.empty(n, np.int32)
        ptr, bset = np.empty(nb + 1, np.int64), np.empty(max(cap, 1), np.int32)
        self._check_value(self._L.crh_sampler_epoch_cgrc(self._h, bs, R, u.ctypes.data, i.ctypes.data, ptr.ctypes.data,
                                                         bset.ctypes.data, cap), "crh_sampler_epoch_cgrc")
        return u, i, ptr, bset[:ptr[-1]]


class EpochPrefetcher:
    """Samples the NEXT epoch of ``next_batch_pairwise`` triples on a worker thread while the GPU trains and
    evaluates the current one (the C++ sampler runs outside the GIL), without changing the stream: the reference
    draws epoch e+1 from NumPy's global generator right after epoch e, and nothing in the BPR-MF / LightGCN trainers
    touches that generator in between.  Guarded: ``get()`` publishes the advanced generator state on the calling
    thread (as if the epoch had been sampled there and then); if somebody did use ``np.random`` since the last
    ``get()``, the speculative epoch is discarded and redrawn from the current state; ``close()`` takes an unused
    speculative epoch back (sampler snapshot), so early stopping leaves generator and permutation where the
    reference leaves them.  ``CRH_SAMPLER_PREFETCH=0`` samples on the calling thread."""

    def __init__(self, sampler: PairwiseSampler, batch_size: int, enabled=None):
        self.s, self.B = sampler, int(batch_size)
        self.enabled = (os.environ.get("CRH_SAMPLER_PREFETCH", "1") != "0") if enabled is None else bool(enabled)
        self._thread = self._result = self._base = None

    @staticmethod
    def _np_state():
        st = np.random.get_state()
        return st[1].copy(), int(st[2])

    def _start(self) -> None:
        """Begin sampling one epoch from NumPy's CURRENT state (which is left untouched until get())."""
        self._base = self._np_state()
        _lib.check(self.s._L.crh_sampler_snapshot(self.s._h), "crh_sampler_snapshot")
        self.s.pull_numpy_state()

        def work():
            try:
                out = self.s.epoch(self.B)
                key = np.empty(624, dtype=np.uint32)
                pos = ctypes.c_int(0)
                _lib.check(self.s._L.crh_sampler_get_state(self.s._h, key.ctypes.data, ctypes.addressof(pos)),
                           "crh_sampler_get_state")
                self._result = (out, key, int(pos.value))
            except BaseException as e:          # re-raised on the calling thread by _finish()
                self._result = e

        if self.enabled:
            self._thread = threading.Thread(target=work, daemon=True)
            self._thread.start()
        else:
            self._thread = None
            work()

    def _finish(self):
        if self._thread is not None:
            self._thread.join()
            self._thread = None
        res, self._result = self._result, None
        if isinstance(res, BaseException) or res is None:
            # the worker failed: put permutation and generator back where the epoch started, then report
            _lib.check(self.s._L.crh_sampler_restore(self.s._h), "crh_sampler_restore")
            self._base = None
            raise RuntimeError("EpochPrefetcher: sampling the epoch failed") from (res if res is not None else None)
        return res

    def get(self):
        """The next epoch's (u, i, j); NumPy's global generator advances exactly as ``epoch_triples`` would."""
        if self._base is not None:
            key, pos = self._np_state()
            if pos != self._base[1] or not np.array_equal(key, self._base[0]):     # np.random was used meanwhile
                self._finish()
                _lib.check(self.s._L.crh_sampler_restore(self.s._h), "crh_sampler_restore")
                self._base = None
        if self._base is None:
            self._start()
        out, key, pos = self._finish()
        st = np.random.get_state()
        np.random.set_state((st[0], key, pos, st[3], st[4]))
        self._base = None
        if self.enabled:
            self._start()                      # speculate on the following epoch
        return out

    def close(self) -> None:
        """Take back a speculative epoch nobody asked for."""
        if self._base is not None:
            self._finish()
            _lib.check(self.s._L.crh_sampler_restore(self.s._h), "crh_sampler_restore")
            self._base = None
