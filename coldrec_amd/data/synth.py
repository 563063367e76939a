"""Seeded synthetic interaction data in ColdRec's shapes and on-disk format.

No dataset ships with the reference (Google-Drive links only, data/README.md:13), so
every benchmark and test here runs on data produced by this generator:

* ``make_interactions`` draws unique (user, item) pairs, users uniform and items
  Zipf-distributed, with ``numpy.random.default_rng`` (PCG64).
* ``split_cold`` applies the reference's warm/cold rule (data/split.py:65-160 and
  data/convert.py:60-143): 80 % of the cold-object groups are warm; warm records are
  split 80/10/10 with val/test records whose user or item never occurs in train moved
  back into train; cold groups are split 50/50 into val/test; the overall val/test
  sets keep the warm-side ids present in both the warm and the cold part.
* ``write_dataset`` emits the seven CSV files, ``info_dict.pkl`` and the content
  ``.npy`` a stock ColdRec ``main.py`` reads (main.py:28-53).

The split is an independent vectorised implementation of the same rule; it is not
required to reproduce the reference's pandas/``random`` stream.
"""
from __future__ import annotations

import os
import pickle
from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np

SHAPES = {
    # name: (users, items, interactions, content_dim)  -- data/README.md:8-11
    "movielens": (6040, 3706, 1_000_209, 206),
    "citeulike": (5551, 16980, 204_986, 300),
    "toy": (300, 500, 6000, 16),
}


def make_interactions(n_user: int, n_item: int, n_pairs: int, zipf_a: float = 0.8,
                      seed: int = 1) -> np.ndarray:
    """Unique (user, item) int64 pairs, users uniform, items ~ rank^-zipf_a."""
    if n_pairs > n_user * n_item:
        raise ValueError("more pairs requested than the user x item grid holds")
    rng = np.random.default_rng(seed)
    w = 1.0 / np.power(np.arange(1, n_item + 1, dtype=np.float64), zipf_a)
    cdf = np.cumsum(w / w.sum())
    item_of_rank = rng.permutation(n_item)
    # every user and every item occurs at least once (the reference indexes id2item with
    # every row of the item table, BaseRecommender.py:186, so unseen ids would crash it)
    cover_u = np.concatenate([np.arange(n_user), rng.integers(0, n_user, size=n_item)])
    cover_i = np.concatenate([rng.integers(0, n_item, size=n_user), np.arange(n_item)])
    seen = np.unique(cover_u.astype(np.int64) * n_item + cover_i)
    seen = seen[rng.permutation(seen.shape[0])]
    if seen.shape[0] > n_pairs:
        raise ValueError("n_pairs too small to cover every user and item once")
    need = n_pairs - seen.shape[0]
    while need > 0:
        m = int(need * 1.3) + 64
        u = rng.integers(0, n_user, size=m, dtype=np.int64)
        i = item_of_rank[np.minimum(np.searchsorted(cdf, rng.random(m)), n_item - 1)]
        key = np.concatenate([seen, u * n_item + i])
        # keep first occurrence order so the result does not depend on batch size quirks
        _, first = np.unique(key, return_index=True)
        seen = key[np.sort(first)]
        need = n_pairs - seen.shape[0]
    seen = seen[:n_pairs]
    return np.stack([seen // n_item, seen % n_item], axis=1)


@dataclass
class ColdSplit:
    cold_object: str
    user_num: int
    item_num: int
    warm_train: np.ndarray
    warm_val: np.ndarray
    warm_test: np.ndarray
    cold_val: np.ndarray
    cold_test: np.ndarray
    overall_val: np.ndarray
    overall_test: np.ndarray
    info: Dict[str, object] = field(default_factory=dict)
    content: Optional[np.ndarray] = None

    def as_lists(self, name: str):
        """``[[u, i, 1.0], ...]`` exactly as util/loader.py:21-33 returns it."""
        arr = getattr(self, name)
        return [[int(u), int(i), 1.0] for u, i in arr]


def _isin(a: np.ndarray, pool: np.ndarray) -> np.ndarray:
    return np.isin(a, pool, assume_unique=False)


def split_cold(pairs: np.ndarray, cold_object: str = "item", warm_ratio: float = 0.8,
               warm_split=(0.8, 0.1, 0.1), cold_split=(0.5, 0.5), seed: int = 42,
               content_dim: int = 0) -> ColdSplit:
    assert cold_object in ("user", "item")
    rng = np.random.default_rng(seed)
    col = 0 if cold_object == "user" else 1
    wcol = 1 - col
    user_num = int(pairs[:, 0].max()) + 1
    item_num = int(pairs[:, 1].max()) + 1

    groups = np.unique(pairs[:, col])
    rng.shuffle(groups)
    n_warm_group = int(warm_ratio * groups.shape[0])
    warm_groups, cold_groups = groups[:n_warm_group], groups[n_warm_group:]
    is_warm = _isin(pairs[:, col], warm_groups)
    warm, cold = pairs[is_warm], pairs[~is_warm]

    warm = warm[rng.permutation(warm.shape[0])]
    n_val = int(warm_split[1] * warm.shape[0])
    n_test = int(warm_split[2] * warm.shape[0])
    n_train = warm.shape[0] - n_val - n_test
    train, val, test = warm[:n_train], warm[n_train:n_train + n_val], warm[n_train + n_val:]

    def pull_back(train, part):
        for c in (0, 1):  # user first, then item, as the reference does
            lost = ~_isin(part[:, c], train[:, c])
            train = np.concatenate([train, part[lost]], axis=0)
            part = part[~lost]
        return train, part

    train, val = pull_back(train, val)
    train, test = pull_back(train, test)

    cg = cold_groups.copy()
    rng.shuffle(cg)
    n_cv = int(cold_split[0] * cg.shape[0])
    cold_val = cold[_isin(cold[:, col], cg[:n_cv])]
    cold_test = cold[_isin(cold[:, col], cg[n_cv:])]

    def overall(cpart, wpart):
        both = np.intersect1d(cpart[:, wcol], wpart[:, wcol])
        cat = np.concatenate([cpart, wpart], axis=0)
        return cat[_isin(cat[:, wcol], both)]

    info = {
        "user_num": user_num,
        "item_num": item_num,
        "user_array": np.arange(user_num, dtype=np.int32),
        "item_array": np.arange(item_num, dtype=np.int32),
        "warm_user": np.unique(train[:, 0]).astype(np.int32),
        "warm_item": np.unique(train[:, 1]).astype(np.int32),
        "cold_user": np.unique(cold[:, 0]).astype(np.int32),
        "cold_item": np.unique(cold[:, 1]).astype(np.int32),
    }
    content = None
    if content_dim > 0:
        n = item_num if cold_object == "item" else user_num
        content = np.random.default_rng(seed + 1).standard_normal((n, content_dim)).astype(np.float32)
    return ColdSplit(cold_object, user_num, item_num, train, val, test, cold_val, cold_test,
                     overall(cold_val, val), overall(cold_test, test), info, content)


def make_dataset(name: str = "toy", cold_object: str = "item", seed: int = 1,
                 with_content: bool = True) -> ColdSplit:
    n_user, n_item, n_pairs, cdim = SHAPES[name]
    pairs = make_interactions(n_user, n_item, n_pairs, seed=seed)
    return split_cold(pairs, cold_object, seed=seed + 41, content_dim=cdim if with_content else 0)


def write_dataset(split: ColdSplit, root: str, dataset: str) -> str:
    """Write ``root/dataset/cold_<obj>/*.csv`` + ``info_dict.pkl`` + content ``.npy``."""
    out = os.path.join(root, dataset, f"cold_{split.cold_object}")
    os.makedirs(out, exist_ok=True)
    files = {
        "warm_train.csv": split.warm_train,
        "warm_val.csv": split.warm_val,
        "warm_test.csv": split.warm_test,
        f"cold_{split.cold_object}_val.csv": split.cold_val,
        f"cold_{split.cold_object}_test.csv": split.cold_test,
        "overall_val.csv": split.overall_val,
        "overall_test.csv": split.overall_test,
    }
    for fname, arr in files.items():
        np.savetxt(os.path.join(out, fname), arr, fmt="%d", delimiter=",", header="user,item",
                   comments="")
    with open(os.path.join(out, "info_dict.pkl"), "wb") as f:
        pickle.dump(split.info, f, protocol=4)
    if split.content is not None:
        np.save(os.path.join(root, dataset, f"{dataset}_{split.cold_object}_content.npy"),
                split.content)
    return out
