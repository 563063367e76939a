"""ORACLE -- TEST INFRASTRUCTURE ONLY: a numpy restatement of the ALGORITHM of the device sampler
(coldrec_amd/csrc/sampler_dev.hip), step for step -- groups of 64 draws classified at once, Fisher-Yates resolved through
previous-occurrence chains, accepted item draws as a compaction of the stream, rejection rounds per batch -- so that its
logic can be checked on the CPU against ``np.random`` itself (tests/test_sampler.py), i.e. against the library the
reference samples with (util/utils.py:123-157).  Nothing under coldrec_amd/ imports it."""
import numpy as np


def pow2mask(x: int) -> int:
    m = int(x)
    for s in (1, 2, 4, 8, 16):
        m |= m >> s
    return m


def shuffle_scan(W, n):
    """np.random.shuffle's draws (for i = n-1..1: j = masked rejection in [0, i]) from the raw 32-bit stream W, 64 draws
    per step as ds_shuffle_scan_kernel does.  Returns (J, raw words consumed, serial-fallback groups, groups)."""
    J = np.zeros(n, np.int64)
    i, q = n - 1, 0
    lanes = np.arange(64)
    n_fb = n_gr = 0
    while i >= 1:
        mask = pow2mask(i)
        stop = mask >> 1
        cap = i - stop                                   # accepts this mask still allows
        v = (W[q:q + 64] & np.uint32(mask)).astype(np.int64)
        n_gr += 1
        acc = v <= i - lanes                             # accepted whatever the earlier lanes did
        rej = v > i
        if (~(acc | rej)).any():                         # an ambiguous draw: walk the group serially
            n_fb += 1
            a = c = 0
            for lane in range(64):
                c += 1
                if v[lane] <= i - a:
                    J[i - a] = v[lane]
                    a += 1
                    if a == cap:
                        break
            i -= a
            q += c
            continue
        pref = np.cumsum(acc) - acc
        A = int(acc.sum())
        if A >= cap:                                     # the mask changes inside the group: cut behind the cap-th accept
            L = int(np.nonzero(acc & (pref == cap - 1))[0][0])
            valid, A, c = acc & (lanes <= L), cap, L + 1
        else:
            valid, c = acc, 64
        J[i - pref[valid]] = v[valid]
        i -= A
        q += c
    return J, q, n_fb, n_gr


def fisher_yates_chains(x, J):
    """x after ``for i = n-1..1: swap(x[i], x[J[i]])`` WITHOUT running the swaps in order: out[i] = what position J[i]
    held just before step i, found by following "the most recent earlier step that targeted this position"."""
    n = len(x)
    steps = np.arange(n)
    valid = (steps >= 1) & (J != steps)
    tgt, st = J[valid], steps[valid]
    o = np.argsort(tgt, kind="stable")
    tgt, st = tgt[o], st[o]
    off = np.concatenate([[0], np.cumsum(np.bincount(tgt, minlength=n))])
    out = np.empty_like(x)
    for i in range(n):
        p, t = (int(J[i]) if i >= 1 else 0), i
        while True:
            b = st[off[p]:off[p + 1]]
            k = np.searchsorted(b, t, side="right")
            if k == len(b):
                break
            p = t = int(b[k])
        out[i] = x[p]
    return out


def epoch(W, order, rec_u, rec_i, rated, n_items, batch_size):
    """One epoch from the raw stream W.  Returns (new order, u, i, j, raw words consumed)."""
    n = len(order)
    J, q, _, _ = shuffle_scan(W, n)
    order = fisher_yates_chains(order, J)
    imax = n_items - 1
    rest = (W[q:] & np.uint32(pow2mask(imax))).astype(np.int64)
    keep = rest <= imax
    V, Vraw = rest[keep], np.nonzero(keep)[0] + q
    u, p, neg = rec_u[order], rec_i[order], np.zeros(n, np.int64)
    o = 0
    for lo in range(0, n, batch_size):
        chk = np.arange(lo, min(lo + batch_size, n))
        while len(chk):
            neg[chk] = V[o:o + len(chk)]
            o += len(chk)
            chk = chk[np.array([int(neg[t]) in rated[int(u[t])] for t in chk], bool)]
    return order, u, p, neg, int(Vraw[o - 1]) + 1
