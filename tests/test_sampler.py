"""CPU: the product's C++ host sampler against the reference's own triples (golden g1) and
against the numpy-RNG oracle on other seeds / batch sizes."""
import os
import time

import numpy as np
import pytest

from coldrec_amd.sampler import PairwiseSampler
from oracle import oracle_np as orc
from tests.conftest import load_golden


def _toy():
    g = load_golden("toy_item.npz")
    umap = {int(k): i for i, k in enumerate(g["user_keys"])}
    imap = {int(k): i for i, k in enumerate(g["item_keys"])}
    ru = np.array([umap[int(u)] for u in g["warm_train"][:, 0]])
    ri = np.array([imap[int(i)] for i in g["warm_train"][:, 1]])
    return g, ru, ri


def test_matches_reference_stream_golden_g1():
    g1 = load_golden("g1_sampler.npz")
    g, ru, ri = _toy()
    s = PairwiseSampler(ru, ri, int(g["user_num"]), len(g["item_keys"]))
    s.seed(int(g1["seed"]))
    u, i, j = [], [], []
    for _ in range(int(g1["epochs"])):
        a, b, c = s.epoch(int(g1["batch_size"]))
        u.append(a); i.append(b); j.append(c)
    assert np.array_equal(np.concatenate(u), g1["u"])
    assert np.array_equal(np.concatenate(i), g1["i"])
    assert np.array_equal(np.concatenate(j), g1["j"])
    # the generator state after three epochs is NumPy's: continue the global stream from it
    np.random.seed(0)
    s.push_numpy_state()
    assert np.array_equal(np.random.randint(0, 1 << 30, size=4), g1["rng_tail"])


def test_two_runs_share_the_shuffled_training_list_g16():
    """``--runs 2`` of the reference's main.py (main.py:160-205; tests/golden/g16_runs2.json was written by main.py itself):
    round r is seeded with r and BOTH rounds sample from the one data object, so round 2 shuffles the training list as round
    1 left it (util/utils.py:125 shuffles in place).  One PairwiseSampler reseeded between the rounds -- what
    coldrec_amd.main does through set_seed -- must reproduce the checksum of each round's first batch and of all its batches."""
    import json
    import zlib
    from tests.conftest import GOLDEN
    g16 = json.load(open(os.path.join(GOLDEN, "g16_runs2.json")))
    g, ru, ri = _toy()
    s = PairwiseSampler(ru, ri, int(g["user_num"]), len(g["item_keys"]))
    B, epochs = 512, 3

    def crc3(a, b, c):
        x = 0
        for arr in (a, b, c):
            x = zlib.crc32(np.ascontiguousarray(arr, dtype=np.int32).tobytes(), x)
        return x

    first, every, counts = [], [], []
    for rnd in range(2):
        s.seed(rnd)
        acc, n = 0, 0
        for _ in range(epochs):
            u, i, j = s.epoch(B)
            for lo in range(0, len(u), B):
                c = crc3(u[lo:lo + B], i[lo:lo + B], j[lo:lo + B])
                if n == 0:
                    first.append(c)
                acc = c ^ (acc * 31 & 0xFFFFFFFF)
                n += 1
        every.append(acc)
        counts.append(n)
    assert counts == g16["n_batches"]
    assert first == g16["first_batch_crc"] and every == g16["all_triples_crc"]
    # a FRESH sampler seeded with 1 gives a different first batch: the carried-over order is what the fixture pins
    t = PairwiseSampler(ru, ri, int(g["user_num"]), len(g["item_keys"]))
    t.seed(1)
    u, i, j = t.epoch(B)
    assert crc3(u[:B], i[:B], j[:B]) != g16["first_batch_crc"][1]


@pytest.mark.parametrize("seed,bs", [(0, 7), (2024, 4096), (123456789, 100), (4294967295, 333)])
def test_matches_numpy_oracle_other_seeds(seed, bs):
    g, ru, ri = _toy()
    n_u, n_i = int(g["user_num"]), len(g["item_keys"])
    s = PairwiseSampler(ru, ri, n_u, n_i)
    o = orc.PairwiseSampler(ru, ri, n_i, n_u)
    np.random.seed(seed)
    s.pull_numpy_state()          # adopt numpy's state instead of seeding: same thing
    for _ in range(2):
        want = [np.concatenate(x) for x in zip(*o.epoch(bs))]
        got = s.epoch(bs)
        for a, b in zip(got, want):
            assert np.array_equal(a, b)


def test_large_record_set_matches_numpy_oracle():
    """Round 6: the gather of the shuffled records rides in the shuffle loop (csrc/sampler.hip: a slot is final once its swap is
    done; slot 0 is fetched after the loop).  Two epochs of 230 000 records against the NumPy oracle, triple for triple, and the
    generator state after them."""
    rng = np.random.default_rng(3)
    n_u, n_i = 1500, 2500
    key = np.unique(rng.integers(0, n_u * n_i, 245_000))[:230_000]
    rng.shuffle(key)
    ru, ri = (key // n_i).astype(np.int32), (key % n_i).astype(np.int32)
    s = PairwiseSampler(ru, ri, n_u, n_i)
    o = orc.PairwiseSampler(ru, ri, n_i, n_u)
    np.random.seed(77)
    s.pull_numpy_state()
    for _ in range(2):
        want = [np.concatenate(x) for x in zip(*o.epoch(4096))]
        got = s.epoch(4096)
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
    key_np, pos_np = np.random.get_state()[1], np.random.get_state()[2]
    import ctypes
    k = np.empty(624, np.uint32)
    pos = ctypes.c_int(0)
    s._L.crh_sampler_get_state(s._h, k.ctypes.data, ctypes.addressof(pos))
    assert np.array_equal(k, key_np) and int(pos.value) == int(pos_np)


def test_fuzz_small_and_odd_record_sets_against_numpy_oracle():
    """Round 6's shuffle (two phases per window, AVX-512 or scalar accept list, swap log) and draws on the shapes the real-size
    tests never see: one to a few hundred records, batches of 1 to 4 096, id spaces on both sides of the 16-bit limit of the
    record-carrying permutation -- three epochs each, triple for triple against the NumPy oracle."""
    rng = np.random.default_rng(20)
    done = 0
    for case in range(160):
        n_u, n_i = int(rng.integers(1, 60)), int(rng.integers(2, 90))
        if case % 7 == 0:
            n_u = int(rng.integers(60000, 70000))
        if case % 11 == 0:
            n_i = int(rng.integers(65000, 66000))
        n = int(rng.integers(1, 700))
        ru, ri = rng.integers(0, min(n_u, 50), n), rng.integers(0, max(1, min(n_i, 80) - 1), n)
        key = np.unique(ru.astype(np.int64) * n_i + ri)
        rng.shuffle(key)
        ru, ri = (key // n_i).astype(np.int32), (key % n_i).astype(np.int32)
        B = int(rng.choice([1, 3, 16, 64, 512, 4096]))
        s = PairwiseSampler(ru, ri, n_u, n_i)
        o = orc.PairwiseSampler(ru, ri, n_i, n_u)
        np.random.seed(int(rng.integers(0, 2 ** 31)))
        s.pull_numpy_state()
        for _ in range(3):
            want = [np.concatenate(x) for x in zip(*o.epoch(B))]
            for a, b in zip(s.epoch(B), want):
                assert np.array_equal(a, b), (case, n_u, n_i, len(ru), B)
        done += 1
    assert done == 160


def test_negatives_never_rated_and_speed():
    rng = np.random.default_rng(0)
    n_u, n_i, n = 2000, 3000, 300_000
    key = np.unique(rng.integers(0, n_u * n_i, n))
    ru, ri = (key // n_i).astype(np.int32), (key % n_i).astype(np.int32)
    s = PairwiseSampler(ru, ri, n_u, n_i)
    s.seed(1)
    t0 = time.perf_counter()
    u, i, j = s.epoch(4096)
    dt = time.perf_counter() - t0
    rated = set((ru.astype(np.int64) * n_i + ri).tolist())
    assert not (set((u.astype(np.int64) * n_i + j).tolist()) & rated)
    assert sorted((u.astype(np.int64) * n_i + i).tolist()) == sorted(rated)   # a permutation of the records
    assert dt < 2.0, dt


def test_bad_records_rejected():
    with pytest.raises(RuntimeError, match="out of range"):
        PairwiseSampler(np.array([0, 5]), np.array([0, 1]), 3, 4)


def test_host_plan_builder_reverse_index():
    from coldrec_amd import ops
    rng = np.random.default_rng(3)
    B, L = 500, 512
    u, p, n = (rng.integers(0, 60, B).astype(np.int32) for _ in range(3))
    plan = ops.build_plans(u, p, n, L)[0]
    nu, ni, lay = plan[:3]
    assert lay == L
    urow, uptr, ulist = plan[3:3 + L], plan[3 + L:4 + 2 * L], plan[4 + 2 * L:4 + 3 * L]
    assert np.array_equal(urow[:nu], np.unique(u)) and uptr[nu] == B
    for s in range(nu):
        seg = ulist[uptr[s]:uptr[s + 1]]
        assert (np.diff(seg) > 0).all() and (u[seg] == urow[s]).all()
    o = 4 + 3 * L
    irow, iptr, ilist = plan[o:o + 2 * L], plan[o + 2 * L:o + 4 * L + 1], plan[o + 4 * L + 1:o + 6 * L + 1]
    assert np.array_equal(irow[:ni], np.unique(np.concatenate([p, n]))) and iptr[ni] == 2 * B
    seen = 0
    for s in range(ni):
        for ent in ilist[iptr[s]:iptr[s + 1]]:
            b, role = ent & 0x3FFFFFFF, ent >> 30
            assert (n if role else p)[b] == irow[s]
            seen += 1
    assert seen == 2 * B


# ----------------------------------------------------------------------------- the other samplers
def _toy_sampler():
    g, ru, ri = _toy()
    s = PairwiseSampler(ru, ri, int(g["user_num"]), len(g["item_keys"]))
    s.set_catalogue(len(g["user_keys"]), g["mapped_cold_item_idx"])
    return g, ru, ri, s


def _seed_both(seed):
    import random
    random.seed(seed)
    np.random.seed(seed)


def _tails():
    import random
    return [random.getrandbits(30), int(np.random.randint(0, 1 << 30))]


def _run_product(s, fn, epochs=2):
    s.pull_python_state(); s.pull_numpy_state()
    outs = [fn() for _ in range(epochs)]
    s.push_numpy_state(); s.push_python_state()
    return outs


def test_other_samplers_match_reference_golden_g10():
    """Product (C++) and oracle (Python restatement) against what the reference's own functions returned."""
    g10 = load_golden("g10_samplers.npz")
    bs = int(g10["batch_size"])
    cat = lambda outs, k: np.concatenate([np.asarray(o[k]).reshape((len(o[0]),) + np.asarray(o[k]).shape[1:])
                                          if np.asarray(o[k]).ndim > 1 else np.asarray(o[k]) for o in outs], 0)
    for tag, n_negs in (("lara", 1), ("lara3", 3)):
        g, ru, ri, s = _toy_sampler()
        _seed_both(int(g10["seed"]))
        outs = _run_product(s, lambda: s.epoch_lara(n_negs))
        assert _tails() == g10[tag + "_tail"].tolist()
        for k, name in enumerate(("u", "i", "nu", "ni")):
            assert np.array_equal(cat(outs, k).reshape(-1), g10[f"{tag}_{name}"]), (tag, name)
        o = orc.OtherSamplers(ru, ri, len(g["user_keys"]), len(g["item_keys"]), g["mapped_cold_item_idx"])
        _seed_both(int(g10["seed"]))
        want = [o.lara_epoch(n_negs) for _ in range(2)]
        for k, name in enumerate(("u", "i", "nu", "ni")):
            assert np.array_equal(np.concatenate([w[k] for w in want]), g10[f"{tag}_{name}"]), ("oracle", tag, name)
    for tag, n_negs in (("clc1", 1), ("clc8", 8)):
        g, ru, ri, s = _toy_sampler()
        _seed_both(int(g10["seed"]))
        outs = _run_product(s, lambda: s.epoch_clcrec(n_negs))
        assert _tails() == g10[tag + "_tail"].tolist()
        assert np.array_equal(np.repeat(cat(outs, 0)[:, None], 1 + n_negs, 1), g10[tag + "_u"])
        assert np.array_equal(cat(outs, 1), g10[tag + "_i"])
        o = orc.OtherSamplers(ru, ri, len(g["user_keys"]), len(g["item_keys"]), g["mapped_cold_item_idx"])
        _seed_both(int(g10["seed"]))
        want = [o.clcrec_epoch(n_negs) for _ in range(2)]
        assert np.array_equal(np.concatenate([np.asarray(w[1]) for w in want]), g10[tag + "_i"])
    g, ru, ri, s = _toy_sampler()
    _seed_both(int(g10["seed"]))
    outs = _run_product(s, lambda: s.epoch_ccfcrec(3, 4, 5))
    assert _tails() == g10["ccf_tail"].tolist()
    o = orc.OtherSamplers(ru, ri, len(g["user_keys"]), len(g["item_keys"]), g["mapped_cold_item_idx"])
    _seed_both(int(g10["seed"]))
    want = [o.ccfcrec_epoch(3, 4, 5) for _ in range(2)]
    for k, name in enumerate(("u", "i", "nu", "pos", "neg", "sneg")):
        assert np.array_equal(cat(outs, k), g10["ccf_" + name]), name
        assert np.array_equal(np.concatenate([np.asarray(w[k]) for w in want]), g10["ccf_" + name]), ("oracle", name)
    for tag, r in (("cgrc32", 32), ("cgrc2", 2)):
        g, ru, ri, s = _toy_sampler()
        _seed_both(int(g10["seed"]))
        outs = _run_product(s, lambda: s.epoch_cgrc(bs, r))
        assert _tails() == g10[tag + "_tail"].tolist()
        assert np.array_equal(cat(outs, 0), g10[tag + "_u"]) and np.array_equal(cat(outs, 1), g10[tag + "_i"])
        assert np.array_equal(np.concatenate([o_[3] for o_ in outs]), g10[tag + "_bset"])   # list(set) order included
        ptr = np.concatenate([outs[0][2], outs[0][2][-1] + outs[1][2][1:]])
        assert np.array_equal(ptr, g10[tag + "_bptr"])
        o = orc.OtherSamplers(ru, ri, len(g["user_keys"]), len(g["item_keys"]), g["mapped_cold_item_idx"])
        _seed_both(int(g10["seed"]))
        want = [b for _ in range(2) for b in o.cgrc_epoch(bs, r)]
        assert np.array_equal(np.concatenate([w[2] for w in want]), g10[tag + "_bset"])


def test_generators_keep_the_reference_contract_g10():
    """util.utils generators: list shapes per batch, short last batch, global RNG states advanced."""
    from coldrec_amd.util import utils as U

    class Data:
        pass
    g, ru, ri, s = _toy_sampler()
    d = Data()
    d.sampler, d.item = s, dict.fromkeys(range(len(g["item_keys"])))
    g10 = load_golden("g10_samplers.npz")
    bs = int(g10["batch_size"])
    _seed_both(int(g10["seed"]))
    b = [x for _ in range(2) for x in U.next_batch_pairwise_CCFCRec(d, bs, 3, 4, 5)]
    assert _tails() == g10["ccf_tail"].tolist()
    assert [len(x[0]) for x in b] == g10["ccf_sizes"].tolist()
    assert isinstance(b[0][4], list) and np.asarray(b[0][4]).shape == (bs, 3, 4) and np.asarray(b[0][5]).shape == (bs, 5)
    assert np.array_equal(np.concatenate([np.asarray(x[4]) for x in b]), g10["ccf_neg"])
    g, ru, ri, s = _toy_sampler()
    d.sampler = s
    _seed_both(int(g10["seed"]))
    b = [x for _ in range(2) for x in U.next_batch_pairwise_CLCRec(d, bs, 8)]
    assert np.array_equal(np.concatenate([np.asarray(x[0]) for x in b]), g10["clc8_u"])
    assert np.array_equal(np.concatenate([np.asarray(x[1]) for x in b]), g10["clc8_i"])
    g, ru, ri, s = _toy_sampler()
    d.sampler = s
    _seed_both(int(g10["seed"]))
    b = [x for _ in range(2) for x in U.next_batch_cgrc(d, bs, 2)]
    assert [len(x[2]) for x in b] == np.diff(g10["cgrc2_bptr"]).tolist()
    assert sum((x[2] for x in b), []) == g10["cgrc2_bset"].tolist()
    g, ru, ri, s = _toy_sampler()
    d.sampler = s
    _seed_both(int(g10["seed"]))
    b = [x for _ in range(2) for x in U.next_batch_pairwise_LARA(d, bs, 3)]
    assert sum((x[2] for x in b), []) == g10["lara3_nu"].tolist() and len(b[0][3]) == 3 * bs


@pytest.mark.parametrize("seed", [0, 11, 987654321])
def test_other_samplers_vs_oracle_small_pools(seed):
    """Users that rated almost every warm item: random.sample takes its copy-the-pool branch, _randbelow(1) and
    two-candidate pools appear; duplicate records; a user and items seen only outside training."""
    rng = np.random.default_rng(seed)
    n_u, n_i = 12, 40
    dense = [(u, i) for u in range(3) for i in rng.permutation(30)[:26 + u].tolist()]      # 26..28 of 30 warm items
    sparse = [(int(u), int(i)) for u, i in zip(rng.integers(3, n_u - 1, 120), rng.integers(0, 36, 120))]
    rec = dense + sparse + dense[:5]
    rec = [rec[k] for k in rng.permutation(len(rec)).tolist()]
    ru, ri = np.array([r[0] for r in rec], np.int32), np.array([r[1] for r in rec], np.int32)
    cold = np.arange(30, 40)
    for what, args in (("lara", (2,)), ("clcrec", (1,)), ("clcrec", (2,)), ("ccfcrec", (2, 3, 4)), ("cgrc", (16, 5))):
        s = PairwiseSampler(ru, ri, n_u, n_i)
        s.set_catalogue(n_u, cold)
        o = orc.OtherSamplers(ru, ri, n_u, n_i, cold)
        _seed_both(seed)
        want = [getattr(o, what + "_epoch")(*args) for _ in range(3)]
        if what == "cgrc":
            want = [[sum((b[k] for b in ep), []) for k in range(3)] for ep in (list(w) for w in want)]
        tail_o = _tails()
        _seed_both(seed)
        got = _run_product(s, lambda: getattr(s, "epoch_" + what)(*args), epochs=3)
        assert _tails() == tail_o, what
        for ep in range(3):
            if what == "cgrc":
                gu, gi, _ptr, gb = got[ep]
                assert gu.tolist() == want[ep][0] and gi.tolist() == want[ep][1] and gb.tolist() == want[ep][2]
            elif what == "clcrec":
                assert got[ep][0].tolist() == want[ep][0] and got[ep][1].tolist() == want[ep][1]
            else:
                for k in range(len(want[ep])):
                    assert np.asarray(got[ep][k]).reshape(-1).tolist() == np.asarray(want[ep][k]).reshape(-1).tolist(), (what, k)


def test_other_samplers_error_behaviour():
    ru, ri = np.array([0, 0, 1], np.int32), np.array([0, 1, 0], np.int32)
    s = PairwiseSampler(ru, ri, 2, 3)
    with pytest.raises(RuntimeError, match="set_catalogue"):
        s.epoch_lara(1)
    s.set_catalogue(3, np.array([2]))                    # warm pool {0, 1}; user 0 rated both; user 2 rated nothing
    assert s._L.crh_sampler_min_candidates(s._h) == 0
    with pytest.raises(ValueError, match="only 0 warm negatives"):
        s.epoch_clcrec(1)
    with pytest.raises(ValueError, match="no warm negative items"):
        s.epoch_ccfcrec(1, 1, 1)
    s.set_catalogue(3, np.array([0, 1, 2]))
    with pytest.raises(ValueError, match="pool is empty"):
        s.epoch_clcrec(1)


def test_item_set_order_is_cpythons():
    """crh_sampler_epoch_cgrc returns list(set(...)) order: check the restated set table against this interpreter
    over sizes that cross several table rebuilds (8 -> 32 -> 128 -> ... and the 2x growth beyond 50000)."""
    rng = np.random.default_rng(0)
    for n_items, n_rec in ((50, 30), (3000, 700), (70_000, 9_000), (400_000, 80_000)):
        ri = rng.integers(0, n_items, n_rec).astype(np.int32)
        ru = np.zeros(n_rec, np.int32)
        s = PairwiseSampler(ru, ri, 1, n_items)
        np.random.seed(5)
        s.pull_numpy_state()
        u, i, ptr, bset = s.epoch_cgrc(n_rec, 0)          # no negatives: B = set(positives)
        assert bset.tolist() == list(set(i.tolist()))


def test_epoch_prefetcher_keeps_the_stream_and_takes_back_unused_epochs():
    """Sampling the next epoch on a worker thread must be invisible: same triples, same NumPy generator state at
    every point the main thread can observe, also after an early stop and when somebody else draws from np.random."""
    from coldrec_amd.sampler import EpochPrefetcher
    g, ru, ri = _toy()
    n_u, n_i, B = int(g["user_num"]), len(g["item_keys"]), 512

    def sequential(n_epochs, disturb_after=None):
        s = PairwiseSampler(ru, ri, n_u, n_i)
        np.random.seed(5)
        out = []
        for e in range(n_epochs):
            s.pull_numpy_state()
            out.append(s.epoch(B))
            s.push_numpy_state()
            if disturb_after == e:
                np.random.rand(3)
        return out, np.random.randint(0, 1 << 30, 4)

    for threaded in (True, False):
        want, tail = sequential(4)
        s = PairwiseSampler(ru, ri, n_u, n_i)
        np.random.seed(5)
        pf = EpochPrefetcher(s, B, enabled=threaded)
        got = [pf.get(), pf.get()]
        pf.close()                                               # "early stop": the speculative third epoch goes back
        probe = np.random.get_state()
        pf2 = EpochPrefetcher(s, B, enabled=threaded)
        got += [pf2.get(), pf2.get()]
        pf2.close()
        for a, b in zip(got, want):
            assert all(np.array_equal(x, y) for x, y in zip(a, b))
        assert np.array_equal(np.random.randint(0, 1 << 30, 4), tail)
        _, tail2 = sequential(2)
        np.random.set_state(probe)
        assert np.array_equal(np.random.randint(0, 1 << 30, 4), tail2)   # state after close() == after 2 epochs
        # somebody draws from np.random between two epochs: the speculative epoch is dropped and redrawn
        want, tail = sequential(3, disturb_after=0)
        s = PairwiseSampler(ru, ri, n_u, n_i)
        np.random.seed(5)
        pf = EpochPrefetcher(s, B, enabled=threaded)
        got = [pf.get()]
        np.random.rand(3)
        got += [pf.get(), pf.get()]
        pf.close()
        for a, b in zip(got, want):
            assert all(np.array_equal(x, y) for x, y in zip(a, b))
        assert np.array_equal(np.random.randint(0, 1 << 30, 4), tail)


def test_snapshot_restore_through_every_path():
    """crh_sampler_snapshot copies nothing: the pairwise epoch after it logs its swaps and _restore replays them backwards;
    a second epoch or another sampler's shuffle under the same snapshot turns it into a copy first.  Whatever ran in
    between, the epochs after _restore are the epochs a sampler that never speculated draws."""
    from coldrec_amd import _lib
    g, ru, ri = _toy()
    n_u, n_i, B = int(g["user_num"]), len(g["item_keys"]), 512

    def fresh(wide=True):
        # (ids beyond 16 bits: the permutation stays in its narrow form)
        s = PairwiseSampler(ru, ri, n_u if wide else 70000, n_i)
        s.set_catalogue(len(g["user_keys"]), g["mapped_cold_item_idx"])
        s.seed(11)
        import random
        random.seed(11)
        s.pull_python_state()
        return s

    def snap(s):
        _lib.check(s._L.crh_sampler_snapshot(s._h), "snapshot")

    def back(s):
        _lib.check(s._L.crh_sampler_restore(s._h), "restore")

    for wide in (True, False):
        ref = fresh(wide)
        e1 = ref.epoch(B)
        want = [ref.epoch(B), ref.epoch(B)]
        want_lara = ref.epoch_lara(1)
        for between in ("nothing", "one epoch", "two epochs", "epoch + lara", "lara", "restore twice"):
            s = fresh(wide)
            assert all(np.array_equal(a, b) for a, b in zip(s.epoch(B), e1))
            snap(s)
            if between in ("one epoch", "two epochs", "epoch + lara", "restore twice"):
                s.epoch(B)
            if between == "two epochs":
                s.epoch(B)
            if between in ("epoch + lara", "lara"):
                s.epoch_lara(1)
            back(s)
            if between == "restore twice":
                assert all(np.array_equal(a, b) for a, b in zip(s.epoch(B), want[0]))
                back(s)
            got = [s.epoch(B), s.epoch(B)]
            for a, b in zip(got, want):
                assert all(np.array_equal(x, y) for x, y in zip(a, b)), (wide, between)
            assert all(np.array_equal(x, y) for x, y in zip(s.epoch_lara(1), want_lara)), (wide, between)


def test_scalar_compactions_when_avx512_is_switched_off():
    """On hosts with AVX-512 (the build container and the GPU boxes both) the two stream compactions of an epoch run 16 draws
    per step; the scalar forms are what every other host runs.  The switch is read once per process, so the golden-stream
    tests are repeated in a child with CRH_SAMPLER_NO_AVX512=1."""
    import subprocess
    import sys
    env = dict(os.environ, CRH_SAMPLER_NO_AVX512="1")
    here = os.path.abspath(__file__)
    r = subprocess.run([sys.executable, "-m", "pytest", here, "-x", "-q", "-p", "no:cacheprovider", "-k",
                        "golden_g1 or large_record_set or snapshot_restore or prefetcher or fuzz_small"],
                       env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(here)))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "deselected" in r.stdout


def test_cgrc_try_limit_and_lara_guard():
    """next_batch_cgrc stops drawing for a user after 50 x ranking_neg tries (a user that rated everything consumes
    exactly that many draws, util/utils.py:321-334); next_batch_pairwise_LARA would loop for ever on such a user in
    the reference -- the C++ sampler reports it instead of hanging the host."""
    n_u, n_i = 4, 6
    rec = [(0, i) for i in range(n_i)] + [(1, 0), (2, 3), (3, 5)]          # user 0 rated every item
    ru, ri = np.array([r[0] for r in rec], np.int32), np.array([r[1] for r in rec], np.int32)
    s = PairwiseSampler(ru, ri, n_u, n_i)
    s.set_catalogue(n_u + 1, None)
    o = orc.OtherSamplers(ru, ri, n_u + 1, n_i, ())
    _seed_both(3)
    want = [[sum((b[k] for b in ep), []) for k in range(3)] for ep in (list(o.cgrc_epoch(4, 2)) for _ in range(2))]
    tail = _tails()
    _seed_both(3)
    got = _run_product(s, lambda: s.epoch_cgrc(4, 2))
    assert _tails() == tail
    for ep in range(2):
        gu, gi, _ptr, gb = got[ep]
        assert gu.tolist() == want[ep][0] and gi.tolist() == want[ep][1] and gb.tolist() == want[ep][2]
    with pytest.raises(RuntimeError, match="rated every item"):
        s.epoch_lara(1)


def test_destroy_while_a_background_epoch_is_running_returns():
    """ADVICE.md (round 2): crh_sampler_destroy during a RUNNING background epoch used to hang -- the finishing job
    overwrote the shutdown flag.  Destroy now lets the epoch finish (the worker owns the output arrays until then),
    then stops the worker.  Run in a child process so a regression is a timeout, not a hung suite."""
    import subprocess
    import sys
    code = r"""
import time, numpy as np
from coldrec_amd.sampler import PairwiseSampler
rng = np.random.default_rng(0)
n_u, n_i, n = 20000, 30000, 3000000
s = PairwiseSampler(rng.integers(0, n_u, n), rng.integers(0, n_i, n), n_u, n_i)
s.seed(1)
out = [np.empty(n, np.int32) for _ in range(3)]
for delay, wait in ((0.0, True), (0.05, True), (0.003, False)):
    out[0][:] = -1
    rc = s._L.crh_sampler_epoch_async(s._h, 4096, out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data, 1)
    assert rc == 0
    time.sleep(delay)
    if wait:
        assert s._L.crh_sampler_epoch_wait(s._h) == 0
assert (out[0] < 0).any()            # 3 ms into a 3 M-record epoch: the worker is still writing
h, s._h = s._h, None
t0 = time.perf_counter()
s._L.crh_sampler_destroy(h)          # job == 2 (running) at this point
assert (out[0] >= 0).all() and (out[0] < n_u).all()      # the epoch was completed before the worker stopped
print("destroyed in %.3f s" % (time.perf_counter() - t0))
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "destroyed in" in r.stdout
