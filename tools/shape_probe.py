#!/usr/bin/env python3
"""One ranking shape, repeated (for rocprofv3 --kernel-trace --stats):  python tools/shape_probe.py USERS ITEMS [D] [REPS]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from coldrec_amd import ops
n_users, n_items = int(sys.argv[1]), int(sys.argv[2])
d = int(sys.argv[3]) if len(sys.argv) > 3 else 128
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device("cuda:0")
U = bench.xavier_(n_users, d, 31, dev, n_items)
V = bench.item_shard(n_items, d, 0, n_items, dev)
rowptr, col = bench.rated_lists(n_users, n_items, 50, seed=4)
cold = np.where(np.random.default_rng(5).random(n_items) < 0.2)[0]
bm = ops.make_bitmap(n_items, cold, dev)
rp, rc = torch.from_numpy(rowptr).to(dev), torch.from_numpy(col).to(dev)
for _ in range(2):
    ops.score_topk(U, None, V, 20, rp, rc, bm)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    ops.score_topk(U, None, V, 20, rp, rc, bm)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / reps * 1e3
print("%d x %d d=%d: %.2f ms, %.3f of the fp32 MFMA peak" % (n_users, n_items, d, ms, 2.0 * d * n_users * n_items / ms / 1e9 / 157.3))
