"""G13: the reference's own ``_evaluate`` (model/BaseRecommender.py:109-188: torch.matmul on the host -> per-user rated masks ->
candidate mask -> torch.topk) at a catalogue that is not toy-sized -- 512 users x 100 000 items, d = 128, the three test
settings of a cold_object=item split -- against the fused HIP ranking through the product's trainer API.

The fixture holds only the reference's OUTPUT (top-20 ids and scores per setting, 190 KB; tests/golden/make_golden.py g13);
interactions, split, id tables and embedding tables are regenerated here from the same seeds (numpy PCG64 and the product's
split_cold are deterministic; checksums of the tables and of both id tables are compared first).  north_star: "top-k indices
bit-exact (ties broken identically)": a user's list must be IDENTICAL wherever its ranking is determined -- every adjacent gap
of the fp64 top-21 above twice the fp32 dot-product error bound -- and equal as a multiset of (id within tolerance of score)
elsewhere; scores within the fp32 bound of MKL's summation order."""
import argparse
import types
import zlib

import numpy as np
import pytest
import torch

from tests.conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def g13_pairs(n_user=512, n_item=100_000, extra=60_000, seed=13):
    """The twin of tests/golden/make_golden.py:g13_pairs (same generator calls)."""
    rng = np.random.default_rng(seed)
    u = np.concatenate([rng.integers(0, n_user, n_item), rng.integers(0, n_user, extra)])
    i = np.concatenate([np.arange(n_item), rng.integers(0, n_item, extra)])
    key = np.unique(u.astype(np.int64) * n_item + i)
    return np.stack([key // n_item, key % n_item], axis=1)


def _crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes(), 0)


def test_g13_reference_evaluate_at_100k_items():
    from coldrec_amd.data.synth import split_cold
    from coldrec_amd.model.BaseRecommender import BaseColdStartTrainer
    from coldrec_amd.util.databuilder import ColdStartDataBuilder
    g = load_golden("g13_eval_100k.npz")
    split = split_cold(g13_pairs(seed=int(g["pairs_seed"])), "item", seed=int(g["split_seed"]))
    info = split.info
    data = ColdStartDataBuilder(split.warm_train, split.warm_val, split.cold_val, split.overall_val, split.warm_test,
                                split.cold_test, split.overall_test, info["user_num"], info["item_num"], info["warm_user"],
                                info["warm_item"], info["cold_user"], info["cold_item"], None, None)
    assert data.user_num == int(g["n_user"]) and data.item_num == int(g["n_item"])
    assert _crc(np.asarray(data.user_keys, np.int64)) == int(g["user_keys_crc"])       # same internal ids as the reference's builder
    assert _crc(np.asarray(data.item_keys, np.int64)) == int(g["item_keys_crc"])
    d = int(g["d"])
    rng = np.random.default_rng(int(g["table_seed"]))
    a_u, a_i = np.sqrt(6.0 / (data.user_num + d)), np.sqrt(6.0 / (data.item_num + d))
    U = ((rng.random((data.user_num, d)) * 2 - 1) * a_u * 8).astype(np.float32)
    V = ((rng.random((data.item_num, d)) * 2 - 1) * a_i * 8).astype(np.float32)
    assert _crc(U) == int(g["U_crc"]) and _crc(V) == int(g["V_crc"])

    class Stock(BaseColdStartTrainer):
        fused_eval = True

        def train(self): ...
        def predict(self, u): ...
        def save(self): ...
        def batch_predict(self, users): ...

    a = dict(dataset="g13", model="MF", epochs=0, layers=2, topN="10,20", bs=256, emb_size=d, lr=1e-3, reg=1e-4, runs=1,
             seed=2024, use_gpu=True, save_emb=False, gpu_id=0, cold_object="item", backbone="MF", early_stop=0, eval_every=1)
    tr = Stock(types.SimpleNamespace(args=argparse.Namespace(**a), data=data, device=DEV))
    tr.user_emb, tr.item_emb = torch.from_numpy(U).to(DEV), torch.from_numpy(V).to(DEV)
    gam = d * 2.0 ** -24 / (1 - d * 2.0 ** -24)
    V64, aV = V.astype(np.float64), np.abs(V).astype(np.float64)
    total = same = det = 0
    for t in ("all", "warm", "cold"):
        c, s, i = tr._topk_arrays(tr._sets("test", t), t)
        users = g[t + "_users_int"].astype(np.int64)
        assert np.array_equal(c["users_int"].cpu().numpy().astype(np.int64), users), t       # same users in the same order
        want_i, want_s = g[t + "_idx"].astype(np.int64), g[t + "_score"]
        k = want_i.shape[1]
        rp = c["rated_rowptr_host"]
        rc = c["rated_col"].cpu().numpy() if c["rated_col"] is not None else np.zeros(0, np.int32)
        cand = {"warm": np.asarray(data.mapped_cold_item_idx), "cold": np.asarray(data.mapped_warm_item_idx), "all": None}[t]
        S = U[users].astype(np.float64) @ V64.T
        for r in range(len(users)):
            S[r, rc[rp[r]:rp[r + 1]]] = -1e9
        if cand is not None and len(cand):
            S[:, cand] = -1e9
        err = gam * (np.abs(U[users]).astype(np.float64) @ aV.T).max(axis=1)             # fp32 dot-product bound, any order
        top = -np.sort(-S, axis=1)[:, :k + 1]
        gaps = np.where(top[:, 1:] > -1e8, top[:, :-1] - top[:, 1:], np.inf)
        determined = gaps.min(axis=1) > 2.0 * err
        real = want_s > -1e8
        equal = np.array([np.array_equal(i[r][real[r]], want_i[r][real[r]]) for r in range(len(users))])
        assert equal[determined].all(), (t, "determined rankings that differ:", np.nonzero(determined & ~equal)[0][:8])
        # scores: ours (canonical fp32 fma chain) and the reference's (MKL) both within the bound of the fp64 value
        for r in range(len(users)):
            m = real[r]
            assert np.all(np.abs(s[r][m] - S[r, i[r][m]]) <= err[r] + 1e-12)
            assert np.all(np.abs(want_s[r][m] - S[r, want_i[r][m]]) <= err[r] + 1e-12)
            if not equal[r]:      # an undetermined user: the two lists may swap near-ties, never more -- same score multiset
                assert np.all(np.abs(np.sort(s[r][m]) - np.sort(want_s[r][m])) <= 2 * err[r] + 1e-12)
        total, same, det = total + len(users), same + int(equal.sum()), det + int(determined.sum())
    assert det >= 0.85 * total, f"only {det} of {total} rankings are determined"      # (88 % at this shape: the worst-case bound is ~30x the typical error)
    print(f"g13: {same} of {total} top-20 lists identical to the reference's at 100 000 items ({det} with a determined ranking)")
