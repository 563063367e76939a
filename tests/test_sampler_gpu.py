"""GPU: the host sampler's path onto the device (EpochPrefetcher with pinned buffers + asynchronous upload, what the trainers
use) against the reference's own triples (golden g1, util/utils.py:123-157 run by the reference) and against the plain host
sampler on other shapes.  Everything is integer work: bit-exact triples, permutation and generator state.
(The device-side sampler of rounds 3-4 -- the same stream as a chain of kernels, 3x slower than this path at MovieLens size --
left the library in round 5: HISTORY.md.)"""
import numpy as np
import pytest
import torch

from coldrec_amd.sampler import EpochPrefetcher, PairwiseSampler
from tests.conftest import load_golden
from tests.test_sampler import _toy

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _np(ts):
    torch.cuda.synchronize()
    return [t.cpu().numpy() for t in ts]


def _random_records(rng, n_u, n_i, n):
    key = np.unique(rng.integers(0, n_u * n_i, n))
    rng.shuffle(key)
    return (key // n_i).astype(np.int32), (key % n_i).astype(np.int32)


def test_prefetched_epochs_on_the_device_match_reference_stream_golden_g1():
    g1 = load_golden("g1_sampler.npz")
    g, ru, ri = _toy()
    s = PairwiseSampler(ru, ri, int(g["user_num"]), len(g["item_keys"]))
    np.random.seed(int(g1["seed"]))
    pf = EpochPrefetcher(s, int(g1["batch_size"]), device=DEV)
    u, i, j = [], [], []
    for e in range(int(g1["epochs"])):
        if e == int(g1["epochs"]) - 1:
            pf.enabled = False                  # nothing follows the last epoch (what the trainers do)
        a, b, c = _np(pf.get())
        u.append(a); i.append(b); j.append(c)
    pf.close()
    assert np.array_equal(np.concatenate(u), g1["u"])
    assert np.array_equal(np.concatenate(i), g1["i"])
    assert np.array_equal(np.concatenate(j), g1["j"])
    assert np.array_equal(np.random.randint(0, 1 << 30, size=4), g1["rng_tail"])     # NumPy's global stream continues


def test_movielens_shape_two_epochs_through_the_prefetcher():
    from coldrec_amd.data.synth import make_dataset
    split = make_dataset("movielens", "item", seed=1, with_content=False)
    tr = split.warm_train
    _, ru = np.unique(tr[:, 0], return_inverse=True)
    _, ri = np.unique(tr[:, 1], return_inverse=True)
    plain = PairwiseSampler(ru, ri, split.user_num, split.item_num)
    plain.seed(2024)
    np.random.seed(2024)
    pf = EpochPrefetcher(PairwiseSampler(ru, ri, split.user_num, split.item_num), 4096, device=DEV)
    for _ in range(2):
        for a, b in zip(_np(pf.get()), plain.epoch(4096)):
            assert np.array_equal(a, b)
    pf.close()


def test_device_output_prefetcher_follows_the_prefetcher_contract():
    """get() sequence == the unprefetched sampler's (triples and np.random state after every epoch); an np.random user
    between two epochs invalidates the speculative one; close() takes the unused epoch back, and a host prefetcher over
    the same sampler then continues the SAME epoch sequence."""
    rng = np.random.default_rng(5)
    ru, ri = _random_records(rng, 150, 400, 12000)
    np.random.seed(77)
    ref = EpochPrefetcher(PairwiseSampler(ru, ri, 150, 400), 256, enabled=False)
    want, states = [], []
    for e in range(6):
        if e == 3:
            np.random.random_sample(3)
        want.append(ref.get())
        states.append(np.random.get_state())
    ref.close()
    hs = PairwiseSampler(ru, ri, 150, 400)
    np.random.seed(77)
    pf = EpochPrefetcher(hs, 256, device=DEV)                # pinned buffers + async upload, speculation on
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
    host = EpochPrefetcher(hs, 256, enabled=False)
    for e in (4, 5):
        for a, b in zip(host.get(), want[e]):
            assert np.array_equal(a, b), e
    host.close()
