#!/usr/bin/env python3
"""Debug aid for csrc/sampler_dev.hip: runs one configuration epoch by epoch and compares the intermediate arrays of
the device pipeline (J of the shuffle scan, the permutation, the accepted draws) with a numpy restatement."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coldrec_amd.sampler import DeviceSampler, PairwiseSampler

MTN = 624


def pow2mask(x):
    m = int(x)
    for s in (1, 2, 4, 8, 16):
        m |= m >> s
    return m


def layout(n, nb):
    words = nb * MTN
    al = lambda x: (x + 255) & ~255
    o, L = 0, {}
    for name, size in (("W", words * 4), ("Kraw", words * 4), ("J", n * 4), ("cnt", (n + 1) * 4), ("off", (n + 1) * 4),
                       ("cursor", (n + 1) * 4), ("bins", n * 4), ("new_order", n * 4), ("flag", words * 4),
                       ("foff", (words + 1) * 4), ("bsum", (max(words, n + 1) // 1024 + 2) * 4), ("V", words * 4),
                       ("Vraw", words * 4), ("ctrl", 16)):
        L[name] = (o, size)
        o += al(size)
    return L


def serial_J(raw, n):
    J = np.zeros(n, np.int64)
    q, i = 0, n - 1
    while i >= 1:
        mask = pow2mask(i)
        while True:
            v = int(raw[q]) & mask
            q += 1
            if v <= i:
                break
        J[i] = v
        i -= 1
    return J, q


def main():
    n_u, n_i, n, bs = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (700, 100, 30000, 8192)))
    rng = np.random.default_rng(n * 31 + bs)
    key = np.unique(rng.integers(0, n_u * n_i, n)); rng.shuffle(key)
    ru, ri = (key // n_i).astype(np.int32), (key % n_i).astype(np.int32)
    n = len(ru)
    hs, ds = PairwiseSampler(ru, ri, n_u, n_i), DeviceSampler(ru, ri, n_u, n_i, "cuda:0")
    np.random.seed(n + bs if len(sys.argv) < 6 else int(sys.argv[5]))
    hs.pull_numpy_state(); ds.pull_numpy_state()
    for ep in range(4):
        key0, pos0, _ = ds.get_state()
        order0 = ds.order.cpu().numpy().copy()
        want = hs.epoch(bs)
        got = [t.cpu().numpy() for t in ds.epoch(bs)]
        torch.cuda.synchronize()
        ok = [np.array_equal(a, b) for a, b in zip(got, want)]
        L = layout(n, ds.n_blocks)
        ws = ds._ws.cpu().numpy()
        take = lambda name, dt=np.int32: ws[L[name][0]: L[name][0] + L[name][1]].view(dt)
        J_dev, ctrl = take("J").astype(np.int64), take("ctrl")
        rs = np.random.RandomState(); rs.set_state(("MT19937", key0, pos0, 0, 0.0))
        raw = rs._bit_generator.random_raw(ds.n_blocks * MTN).astype(np.uint32)
        W_dev = take("W", np.uint32)
        w_ok = np.array_equal(W_dev[pos0: pos0 + 4000], raw[:4000]) and np.array_equal(W_dev[pos0:], raw[: len(W_dev) - pos0])
        J_ref, q_ref = serial_J(raw, n)
        bad = np.flatnonzero(J_dev[1:] != J_ref[1:]) + 1
        print(f"epoch {ep}: triples ok {ok}  blocks {ds.n_blocks} pos0 {pos0}  W ok {w_ok}  q_end dev {ctrl[0]} ref {q_ref}  "
              f"J mismatches {len(bad)} first {bad[:8]}  n_acc {ctrl[1]} consumed {ctrl[2]} status {ctrl[3]}")
        if len(bad):
            i = bad[-1]
            print("   highest bad i", i, "dev", J_dev[i], "ref", J_ref[i], " lowest bad", bad[0], J_dev[bad[0]], J_ref[bad[0]])
        # permutation check given the reference J
        x = order0.copy()
        for i in range(n - 1, 0, -1):
            j = J_ref[i]; x[i], x[j] = x[j], x[i]
        print("   order ok", np.array_equal(x, ds.order.cpu().numpy()))


main()
