"""GPU: the device sampler (csrc/sampler_dev.hip, crh_dsampler_epoch) against the reference's own triples (golden g1,
util/utils.py:123-157 run by the reference) and against the C++ host sampler -- itself pinned to the reference and to
the NumPy oracle in tests/test_sampler.py -- on other shapes, seeds, generator positions and batch sizes.  Everything
is integer work: bit-exact triples, permutation and generator state."""
import numpy as np
import pytest
import torch

from coldrec_amd.sampler import DevicePrefetcher, DeviceSampler, EpochPrefetcher, PairwiseSampler
from tests.conftest import load_golden
from tests.test_sampler import _toy

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _np(ts):
    torch.cuda.synchronize()
    return [t.cpu().numpy() for t in ts]


def test_device_sampler_matches_reference_stream_golden_g1():
    g1 = load_golden("g1_sampler.npz")
    g, ru, ri = _toy()
    s = DeviceSampler(ru, ri, int(g["user_num"]), len(g["item_keys"]), DEV)
    s.seed(int(g1["seed"]))
    u, i, j = [], [], []
    for _ in range(int(g1["epochs"])):
        a, b, c = _np(s.epoch(int(g1["batch_size"])))
        u.append(a); i.append(b); j.append(c)
    assert np.array_equal(np.concatenate(u), g1["u"])
    assert np.array_equal(np.concatenate(i), g1["i"])
    assert np.array_equal(np.concatenate(j), g1["j"])
    np.random.seed(0)
    s.push_numpy_state()                    # the generator after three epochs is NumPy's: continue the global stream
    assert np.array_equal(np.random.randint(0, 1 << 30, size=4), g1["rng_tail"])


def _random_records(rng, n_u, n_i, n):
    key = np.unique(rng.integers(0, n_u * n_i, n))
    rng.shuffle(key)
    return (key // n_i).astype(np.int32), (key % n_i).astype(np.int32)


@pytest.mark.parametrize("n_u,n_i,n,bs,skip,csr", [
    (1, 2, 1, 1, 0, False), (3, 5, 2, 8, 1, False), (4, 9, 3, 2, 623, True), (10, 40, 63, 64, 624, False),
    (10, 40, 64, 7, 5, True), (12, 33, 65, 65, 100, False), (50, 64, 1000, 128, 611, False),
    (300, 500, 9000, 1024, 7, True), (2000, 3000, 300000, 4096, 333, False), (700, 100, 30000, 8192, 0, False),
    (40, 30, 100, 16, 17, False)])
def test_device_sampler_equals_host_sampler(n_u, n_i, n, bs, skip, csr):
    """Shapes from one record to 300 K, dense and sparse users, batch sizes up to the 8192 limit, the generator started
    at arbitrary positions inside a key block (skip draws first; 624 = exactly at a block end), bitmap and CSR tests."""
    rng = np.random.default_rng(n * 31 + bs)
    ru, ri = _random_records(rng, n_u, n_i, n)
    hs = PairwiseSampler(ru, ri, n_u, n_i)
    ds = DeviceSampler(ru, ri, n_u, n_i, DEV, bitmap_bytes=0 if csr else 1 << 29)
    assert (ds.bits is None) == csr
    np.random.seed(n + bs)
    if skip:
        np.random.random_sample(1)          # leave the freshly seeded pos = 624
        np.random.randint(0, 1 << 30, size=skip - 1) if skip > 1 else None
    hs.pull_numpy_state()
    ds.pull_numpy_state()
    for ep in range(3):
        want = hs.epoch(bs)
        got = _np(ds.epoch(bs))
        for a, b, name in zip(got, want, "uij"):
            assert np.array_equal(a, b), (name, ep, np.flatnonzero(a != b)[:5])
        hk, hp = np.empty(624, np.uint32), None
        st = np.random.get_state()
        hs.push_numpy_state()
        hk, hp = np.random.get_state()[1].copy(), np.random.get_state()[2]
        np.random.set_state(st)
        dk, dp, status = ds.get_state()
        assert status == 0 and dp == hp and np.array_equal(dk, hk), ("generator state", ep)


def test_device_sampler_recovers_from_a_short_stream():
    rng = np.random.default_rng(3)
    ru, ri = _random_records(rng, 200, 300, 20000)
    hs, ds = PairwiseSampler(ru, ri, 200, 300), DeviceSampler(ru, ri, 200, 300, DEV)
    hs.seed(9); ds.seed(9)
    full = ds.n_blocks
    ds.n_blocks = max(3, full // 5)          # far too few key blocks: status != 0 -> snapshot restored, blocks doubled
    for _ in range(2):
        for a, b in zip(_np(ds.epoch(512)), hs.epoch(512)):
            assert np.array_equal(a, b)
    assert ds.n_blocks >= full // 2


def test_movielens_shape_two_epochs_and_timing():
    from coldrec_amd.data.synth import make_dataset
    split = make_dataset("movielens", "item", seed=1, with_content=False)
    tr = split.warm_train
    _, ru = np.unique(tr[:, 0], return_inverse=True)
    _, ri = np.unique(tr[:, 1], return_inverse=True)
    hs = PairwiseSampler(ru, ri, split.user_num, split.item_num)
    ds = DeviceSampler(ru, ri, split.user_num, split.item_num, DEV)
    hs.seed(2024); ds.seed(2024)
    for _ in range(2):
        for a, b in zip(_np(ds.epoch(4096)), hs.epoch(4096)):
            assert np.array_equal(a, b)
    out = ds.launch(4096)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ds.launch(4096, out)
    e1.record()
    torch.cuda.synchronize()
    assert ds.get_state()[2] == 0
    assert e0.elapsed_time(e1) < 20.0, e0.elapsed_time(e1)     # ms per epoch of 650 161 triples (bench.py reports it)


def test_device_prefetcher_follows_the_host_prefetcher_contract():
    """get() sequence == EpochPrefetcher's (triples and np.random state after every epoch); an np.random user between
    two epochs invalidates the speculative one; close() takes the unused epoch back and returns the permutation to the
    host sampler, which then continues the SAME epoch sequence."""
    rng = np.random.default_rng(5)
    ru, ri = _random_records(rng, 150, 400, 12000)
    ref_s = PairwiseSampler(ru, ri, 150, 400)
    hs = PairwiseSampler(ru, ri, 150, 400)
    ds = DeviceSampler(ru, ri, 150, 400, DEV)
    np.random.seed(77)
    ref = EpochPrefetcher(ref_s, 256, enabled=False)
    want, states = [], []
    for e in range(6):
        if e == 3:
            np.random.random_sample(3)
        want.append(ref.get())
        states.append(np.random.get_state())
    ref.close()
    np.random.seed(77)
    hp = EpochPrefetcher(PairwiseSampler(ru, ri, 150, 400), 256, device=DEV)     # pinned buffers + async upload
    np.random.seed(77)
    for e in range(3):
        got = _np(hp.get())
        for a, b in zip(got, want[e]):
            assert np.array_equal(a, b), ("host prefetcher, device output", e)
    hp.close()
    np.random.seed(77)
    pf = DevicePrefetcher(hs, ds, 256)
    for e in range(4):
        if e == 3:
            np.random.random_sample(3)      # somebody else draws: the speculative epoch 3 must be discarded
        got = _np(pf.get())
        for a, b in zip(got, want[e]):
            assert np.array_equal(a, b), e
        st = np.random.get_state()
        assert st[2] == states[e][2] and np.array_equal(st[1], states[e][1]), e
    pf.close()                              # epoch 4 was sampled speculatively: taken back
    st = np.random.get_state()
    assert st[2] == states[3][2] and np.array_equal(st[1], states[3][1])
    host = EpochPrefetcher(hs, 256, enabled=False)       # the host sampler continues where the device one stopped
    for e in (4, 5):
        for a, b in zip(host.get(), want[e]):
            assert np.array_equal(a, b), e
    host.close()
