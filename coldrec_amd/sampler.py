"""Host sampler front end (C++ MT19937 restatement in csrc/sampler.hip, no GPU involved).

``PairwiseSampler`` produces the (user, positive, negative) triples of one epoch exactly as
util/utils.py:123-157 would -- same NumPy legacy stream, same cumulative shuffle -- in
milliseconds instead of seconds, as three int32 arrays ready for one host-to-device copy.
"""
from __future__ import annotations

import ctypes

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
