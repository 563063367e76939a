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
import time

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
    """Samples the NEXT epoch of ``next_batch_pairwise`` triples in the background while the GPU trains and evaluates
    the current one, without changing the stream: the reference draws epoch e+1 from NumPy's global generator right
    after epoch e, and nothing in the BPR-MF / LightGCN trainers touches that generator in between.  The work runs on
    the sampler's persistent C++ worker thread (crh_sampler_epoch_async: a thread created per epoch would start on a
    sleeping core and run at half speed) and writes into PINNED host buffers; with ``device`` given, ``get()`` uploads
    them asynchronously and returns device tensors (a pageable upload would block the host behind the previous
    epoch's kernels).  Guarded: ``get()`` publishes the advanced generator state on the calling thread (as if the
    epoch had been sampled there and then); if somebody did use ``np.random`` since the last ``get()``, the speculative
    epoch is discarded and redrawn from the current state; ``close()`` takes an unused speculative epoch back (sampler
    snapshot), so early stopping leaves generator and permutation where the reference leaves them.
    ``CRH_SAMPLER_PREFETCH=0`` samples on the calling thread."""

    def __init__(self, sampler: PairwiseSampler, batch_size: int, enabled=None, device=None):
        self.s, self.B = sampler, int(batch_size)
        self.enabled = (os.environ.get("CRH_SAMPLER_PREFETCH", "1") != "0") if enabled is None else bool(enabled)
        self._base = None          # NumPy state the outstanding epoch started from
        self._slot = None          # buffer slot of the outstanding epoch
        self._async = False
        self.timing, self._t_sync = None, 0.0        # set to [] to record (epoch, seconds waited for the sampler) pairs
        self._k = 0
        n = self.s.n_records
        self.device = None
        if device is not None:
            import torch
            self._torch = torch
            self.device = torch.device(device)
            self._host = [tuple(torch.empty(n, dtype=torch.int32).pin_memory() for _ in range(3)) for _ in range(2)]
            self._dev = [tuple(torch.empty(n, dtype=torch.int32, device=self.device) for _ in range(3)) for _ in range(2)]
            self._uploaded = [None, None]      # event: the last upload out of host slot k has executed
            # uploads run on a stream of their own, beside the previous epoch's kernels: queued on the compute stream they sat
            # between two epochs' graphs (3 x 2.6 MB over PCIe = 0.16 ms of a 2.8 ms MovieLens epoch during which the GPU did
            # nothing else).  _mark = everything the compute stream had been given when the previous get() began, i.e. the last
            # reader of the device slot that is about to be overwritten.
            self._copy_stream = torch.cuda.Stream(self.device)
            self._mark = None
        else:
            self._host = [None, None]          # host mode: fresh arrays per epoch (the caller may keep them)

    @staticmethod
    def _np_state():
        st = np.random.get_state()
        return st[1].copy(), int(st[2])

    def _ptrs(self, k):
        if self.device is not None:
            return [t.data_ptr() for t in self._host[k]]
        return [a.ctypes.data for a in self._host[k]]

    def _start(self) -> None:
        """Begin sampling one epoch from NumPy's CURRENT state (which is left untouched until get())."""
        self._base = self._np_state()
        if not self.enabled:       # (the background job takes the snapshot itself, on the worker's core)
            _lib.check(self.s._L.crh_sampler_snapshot(self.s._h), "crh_sampler_snapshot")
        self.s.pull_numpy_state()
        k = self._k
        self._k ^= 1
        if self.device is None:
            self._host[k] = tuple(np.empty(self.s.n_records, np.int32) for _ in range(3))
        elif self._uploaded[k] is not None:
            _t = time.perf_counter()
            self._uploaded[k].synchronize()          # the upload that last read this slot (two epochs ago) is done
            self._t_sync = time.perf_counter() - _t
        pu, pi, pj = self._ptrs(k)
        self._slot, self._async = k, self.enabled
        if self.enabled:
            try:
                _lib.check(self.s._L.crh_sampler_epoch_async(self.s._h, self.B, pu, pi, pj, 1), "crh_sampler_epoch_async")
            except BaseException:
                # the job was never queued, so the worker took no snapshot of THIS epoch and nothing has moved:
                # restoring here would roll permutation and generator back to the previous epoch's snapshot
                self._base = self._slot = None
                raise
            return
        try:
            _lib.check(self.s._L.crh_sampler_epoch(self.s._h, self.B, pu, pi, pj), "crh_sampler_epoch")
        except BaseException:
            self._abort()      # the snapshot taken above belongs to this epoch
            raise

    def _abort(self) -> None:
        """Put permutation and generator back where the outstanding epoch started."""
        _lib.check(self.s._L.crh_sampler_restore(self.s._h), "crh_sampler_restore")
        self._base = self._slot = None

    def _finish(self):
        """Wait for the outstanding epoch; returns (slot, key, pos).  A failed epoch is rolled back and reported."""
        if self._async:
            try:
                _lib.check(self.s._L.crh_sampler_epoch_wait(self.s._h), "crh_sampler_epoch (background)")
            except BaseException:
                self._abort()
                raise
        key = np.empty(624, dtype=np.uint32)
        pos = ctypes.c_int(0)
        _lib.check(self.s._L.crh_sampler_get_state(self.s._h, key.ctypes.data, ctypes.addressof(pos)),
                   "crh_sampler_get_state")
        return self._slot, key, int(pos.value)

    def get(self):
        """The next epoch's (u, i, j) -- int32 device tensors when a device was given (valid until the next-but-one
        ``get()``: two slots), else fresh host arrays; NumPy's global generator advances exactly as ``epoch_triples``
        would."""
        if self.device is not None:
            cur = self._torch.cuda.current_stream(self.device)
            older, self._mark = self._mark, self._torch.cuda.Event()
            self._mark.record(cur)
        if self._base is not None:
            key, pos = self._np_state()
            if pos != self._base[1] or not np.array_equal(key, self._base[0]):     # np.random was used meanwhile
                self._finish()
                self._abort()
        if self._base is None:
            self._start()
        _t0 = time.perf_counter()
        k, key, pos = self._finish()
        _t1 = time.perf_counter()
        st = np.random.get_state()
        np.random.set_state((st[0], key, pos, st[3], st[4]))
        self._base = self._slot = None
        _t2 = time.perf_counter()
        if self.enabled:
            self._start()                      # speculate on the following epoch FIRST (its buffers are the other slot's): the
        _t3 = time.perf_counter()              # worker samples while this thread queues the upload and the epoch's launches
        out = self._host[k]
        if self.device is not None:
            torch = self._torch
            ev = torch.cuda.Event()
            with torch.cuda.stream(self._copy_stream):
                if older is not None:
                    self._copy_stream.wait_event(older)      # the epoch before last has read this device slot
                for dst, src in zip(self._dev[k], self._host[k]):
                    dst.copy_(src, non_blocking=True)
                ev.record(self._copy_stream)
            cur.wait_event(ev)                               # whatever the caller launches next sees the triples
            self._uploaded[k] = ev
            out = self._dev[k]
        if self.timing is not None:
            self.timing.append((_t1 - _t0, (_t2 - _t1) + (time.perf_counter() - _t3), _t3 - _t2, self._t_sync))
        return out

    def close(self) -> None:
        """Take back a speculative epoch nobody asked for."""
        if self._base is not None:
            self._finish()
            self._abort()
        if self.device is not None:
            self._copy_stream.synchronize()
