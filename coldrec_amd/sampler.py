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
        self.timing, self._t_sync = ([] if os.environ.get("CRH_PREFETCH_TIMING") else None), 0.0
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
        out = self._host[k]
        if self.device is not None:
            torch = self._torch
            for dst, src in zip(self._dev[k], self._host[k]):
                dst.copy_(src, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._uploaded[k] = ev
            out = self._dev[k]
        _t2 = time.perf_counter()
        if self.enabled:
            self._start()                      # speculate on the following epoch
        if self.timing is not None:
            self.timing.append((_t1 - _t0, _t2 - _t1, time.perf_counter() - _t2, self._t_sync))
        return out

    def close(self) -> None:
        """Take back a speculative epoch nobody asked for."""
        if self._base is not None:
            self._finish()
            self._abort()


class DeviceSampler:
    """next_batch_pairwise on the GPU (csrc/sampler_dev.hip, crh_dsampler_epoch): the same NumPy stream, the triples
    are written to device memory by a chain of kernels on ``stream``.  Holds the device copies of the records, of the
    rated-item table (users x items bitmap while it fits ``bitmap_bytes``, else the sorted CSR) and of the cumulative
    permutation; the 624-word generator state travels host <-> device as 2.5 KB copies."""

    def __init__(self, rec_user, rec_item, n_users: int, n_items_seen: int, device, bitmap_bytes: int = 1 << 29):
        import torch
        self._torch = torch
        ru = np.ascontiguousarray(rec_user, dtype=np.int32)
        ri = np.ascontiguousarray(rec_item, dtype=np.int32)
        if ru.shape != ri.shape or ru.ndim != 1 or ru.shape[0] < 1:
            raise ValueError("rec_user / rec_item must be non-empty 1-D arrays of equal length")
        if n_items_seen < 2:
            raise ValueError("the device sampler needs at least two items")
        self._L = _lib.lib()
        self.device = torch.device(device)
        self.n, self.n_users, self.n_items = int(ru.shape[0]), int(n_users), int(n_items_seen)
        dev = self.device
        self.rec_u, self.rec_i = torch.from_numpy(ru).to(dev), torch.from_numpy(ri).to(dev)
        wpu = (self.n_items + 31) // 32
        self.bits = self.rowptr = self.col = None
        if wpu * 4 * self.n_users <= bitmap_bytes:
            words = np.zeros(self.n_users * wpu + 1, np.uint32)
            np.bitwise_or.at(words, ru.astype(np.int64) * wpu + (ri >> 5), np.uint32(1) << (ri & 31).astype(np.uint32))
            self.bits, self.wpu = torch.from_numpy(words.view(np.int32)).to(dev), wpu
        else:
            key = np.unique(ru.astype(np.int64) << 32 | ri.astype(np.int64))
            rowptr = np.zeros(self.n_users + 1, np.int64)
            np.cumsum(np.bincount((key >> 32).astype(np.int64), minlength=self.n_users), out=rowptr[1:])
            self.rowptr = torch.from_numpy(rowptr).to(dev)
            self.col = torch.from_numpy((key & 0xFFFFFFFF).astype(np.int32)).to(dev)
            self.wpu = 0
        cnt = np.bincount(ru, minlength=self.n_users).astype(np.float64)
        self.reject_rate = float((cnt * cnt).sum() / (self.n * float(self.n_items)))   # P(a uniform item is rated by the slot's user)
        self.order = torch.arange(self.n, dtype=torch.int32, device=dev)
        self.state = torch.zeros(626, dtype=torch.int32, device=dev)
        self._snap_order, self._snap_state = torch.empty_like(self.order), torch.empty_like(self.state)
        self._pin = torch.empty(626, dtype=torch.int32).pin_memory()
        self.n_blocks = int(self._L.crh_dsampler_blocks_hint(self.n, self.n_items, self.reject_rate))
        self._ws = None
        self.max_batch = int(self._L.crh_dsampler_max_batch())
        self.seed(5489)

    # ---- generator state (np.random.get_state()[1:3])
    def set_state(self, key, pos: int) -> None:
        torch = self._torch
        self._pin[:624] = torch.from_numpy(np.ascontiguousarray(key, dtype=np.uint32).view(np.int32))
        self._pin[624], self._pin[625] = int(pos), 0
        self.state.copy_(self._pin, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()          # the pinned buffer is reused by the next call

    def get_state(self):
        """(key uint32[624], pos, status) -- synchronises with the stream the last epoch() ran on."""
        self._pin.copy_(self.state, non_blocking=True)
        self._torch.cuda.current_stream(self.device).synchronize()
        a = self._pin.numpy().view(np.uint32)
        return a[:624].copy(), int(a[624]), int(a[625])

    def seed(self, seed: int) -> None:
        rs = np.random.RandomState(int(seed) & 0xFFFFFFFF)
        st = rs.get_state()
        self.set_state(st[1], st[2])

    def pull_numpy_state(self) -> None:
        st = np.random.get_state()
        assert st[0] == "MT19937"
        self.set_state(st[1], st[2])

    def push_numpy_state(self) -> None:
        key, pos, status = self.get_state()
        if status != 0:
            raise RuntimeError("device sampler: generator state is invalid (epoch status %d)" % status)
        st = np.random.get_state()
        np.random.set_state((st[0], key, pos, st[3], st[4]))

    # ---- one epoch
    def _io(self, out):
        p = _lib.ptr
        return _lib.DSamplerIO(p(self.rec_u), p(self.rec_i), self.n, self.n_users, self.n_items, p(self.bits), self.wpu,
                               p(self.rowptr), p(self.col), p(self.order), p(self.state), p(out[0]), p(out[1]), p(out[2]))

    def snapshot(self) -> None:
        self._snap_order.copy_(self.order)
        self._snap_state.copy_(self.state)

    def restore(self) -> None:
        self.order.copy_(self._snap_order)
        self.state.copy_(self._snap_state)

    def launch(self, batch_size: int, out=None):
        """Enqueue one epoch on the CURRENT stream (asynchronous).  Returns the three int32 device tensors."""
        torch = self._torch
        if not 1 <= int(batch_size) <= self.max_batch:
            raise ValueError("device sampler: batch_size %d outside 1..%d" % (batch_size, self.max_batch))
        if out is None:
            out = tuple(torch.empty(self.n, dtype=torch.int32, device=self.device) for _ in range(3))
        need = int(self._L.crh_dsampler_workspace_bytes(self.n, self.n_blocks))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        io = self._io(out)
        _lib.check(self._L.crh_dsampler_epoch(ctypes.byref(io), int(batch_size), self.n_blocks, _lib.ptr(self._ws),
                                              self._ws.numel(), _lib.current_stream()), "crh_dsampler_epoch")
        return out

    def epoch(self, batch_size: int, out=None):
        """One epoch, checked: if the generated stream turned out too short (status != 0) the epoch is repeated from
        the snapshot with twice the key blocks.  Synchronises (reads the status back)."""
        for _ in range(6):
            self.snapshot()
            out = self.launch(batch_size, out)
            _key, _pos, status = self.get_state()
            if status == 0:
                return out
            self.restore()
            self.n_blocks *= 2
        raise RuntimeError("device sampler: could not generate enough MT19937 words")


class DevicePrefetcher:
    """EpochPrefetcher's contract with the sampler on the GPU: ``get()`` returns the next epoch's (u, i, j) as DEVICE
    tensors and advances NumPy's global generator exactly as ``epoch_triples`` would; the following epoch is sampled
    speculatively on a side stream while the caller trains (taken back by ``close()`` or when somebody used
    ``np.random`` in between).  The cumulative permutation is taken from / returned to the host sampler, so host and
    device epochs can be mixed on one dataset."""

    def __init__(self, host_sampler: PairwiseSampler, dsampler: DeviceSampler, batch_size: int, enabled=None):
        import torch
        self._torch = torch
        self.hs, self.ds, self.B = host_sampler, dsampler, int(batch_size)
        self.enabled = (os.environ.get("CRH_SAMPLER_PREFETCH", "1") != "0") if enabled is None else bool(enabled)
        self.side = torch.cuda.Stream(self.ds.device)
        self._bufs = [tuple(torch.empty(self.ds.n, dtype=torch.int32, device=self.ds.device) for _ in range(3))
                      for _ in range(2)]
        self._k = 0
        self._base = None          # NumPy state the outstanding (speculative) epoch started from
        self._done = None          # event: the outstanding epoch's kernels
        self._out = None
        order = np.empty(self.ds.n, np.int32)
        _lib.check(self.hs._L.crh_sampler_get_order(self.hs._h, order.ctypes.data), "crh_sampler_get_order")
        self.ds.order.copy_(torch.from_numpy(order))
        self._dev_state_is = None  # NumPy state the device generator is known to equal

    @staticmethod
    def _np_state():
        st = np.random.get_state()
        return st[1].copy(), int(st[2])

    def _start(self) -> None:
        torch = self._torch
        self._base = self._np_state()
        main = torch.cuda.current_stream(self.ds.device)
        self.side.wait_stream(main)            # the buffers of this slot were last read by copies enqueued on `main`
        with torch.cuda.stream(self.side):
            known = self._dev_state_is
            if known is None or known[1] != self._base[1] or not np.array_equal(known[0], self._base[0]):
                self.ds.set_state(*self._base)
            self.ds.snapshot()
            self._out = self.ds.launch(self.B, self._bufs[self._k])
            self._done = torch.cuda.Event()
            self._done.record(self.side)
        self._k ^= 1

    def _take_back(self) -> None:
        torch = self._torch
        with torch.cuda.stream(self.side):
            self.ds.restore()
        self.side.synchronize()
        self._dev_state_is = None
        self._base = self._out = self._done = None

    def get(self):
        torch = self._torch
        if self._base is not None:
            key, pos = self._np_state()
            if pos != self._base[1] or not np.array_equal(key, self._base[0]):     # np.random was used meanwhile
                self._take_back()
        for _ in range(6):
            if self._base is None:
                self._start()
            with torch.cuda.stream(self.side):
                key, pos, status = self.ds.get_state()                             # waits for the epoch's kernels
            if status == 0:
                break
            self._take_back()                                                      # stream too short: more key blocks
            self.ds.n_blocks *= 2
        else:
            raise RuntimeError("device sampler: could not generate enough MT19937 words")
        st = np.random.get_state()
        np.random.set_state((st[0], key, pos, st[3], st[4]))
        self._dev_state_is = (key, pos)
        out, done = self._out, self._done
        self._base = self._out = self._done = None
        torch.cuda.current_stream(self.ds.device).wait_event(done)
        if self.enabled:
            self._start()                      # speculate on the following epoch
        return out

    def close(self) -> None:
        """Take back a speculative epoch nobody asked for and hand the permutation back to the host sampler."""
        if self._base is not None:
            self._take_back()
        self.side.synchronize()
        order = self.ds.order.cpu().numpy()
        _lib.check(self.hs._L.crh_sampler_set_order(self.hs._h, order.ctypes.data), "crh_sampler_set_order")
