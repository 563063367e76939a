#!/usr/bin/env python3
"""Latency of the fused ranking at trainer-validation sizes (a few thousand users x a few thousand items)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coldrec_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for n_users, n_items, masks in ((5400, 3706, "both"), (5400, 3706, "none"), (5400, 3706, "bitmap"), (5400, 3706, "rated"),
                                (5551, 16980, "both"), (1000, 3706, "both"), (64, 3706, "both")):
    d, k = 128, 20
    U = torch.randn(6040, d, device=dev) * 0.1
    V = torch.randn(n_items, d, device=dev) * 0.1
    users = torch.from_numpy(rng.permutation(6040)[:n_users].astype(np.int32)).to(dev)
    rated = [np.unique(rng.integers(0, n_items, 100)) for _ in range(n_users)]
    rp, rc = ops.rated_csr(rated, dev) if masks in ("both", "rated") else (None, None)
    bm = ops.make_bitmap(n_items, np.arange(0, n_items, 5), dev) if masks in ("both", "bitmap") else None
    for _ in range(3):
        s, i = ops.score_topk(U, users, V, k, rp, rc, bm)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        s, i = ops.score_topk(U, users, V, k, rp, rc, bm)
    torch.cuda.synchronize()
    print(f"{n_users} x {n_items} masks={masks}: score_topk {(time.perf_counter() - t) / 20 * 1e3:.3f} ms", flush=True)
