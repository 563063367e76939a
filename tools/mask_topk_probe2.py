#!/usr/bin/env python3
"""crh_mask_topk_f32 on a 4096 x 1M block: which mask costs what."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from coldrec_amd import ops
dev = torch.device("cuda:0")
n_users, n_items = 4096, 1_000_000
S = torch.randn(n_users, n_items, device=dev)
rowptr, col = bench.rated_lists(n_users, n_items, 50, seed=4)
rp, rc = torch.from_numpy(rowptr).to(dev), torch.from_numpy(col).to(dev)
bm = ops.make_bitmap(n_items, np.where(np.random.default_rng(5).random(n_items) < 0.2)[0], dev)
x = torch.empty(n_users * n_items, device=dev)
def t(fn, reps=6):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
for name, args in (("no masks", (None, None, None)), ("rated lists only", (rp, rc, None)), ("bitmap only", (None, None, bm)), ("both", (rp, rc, bm))):
    ms = t(lambda: ops.mask_topk(S, 20, *args, write_back=False))
    print(f"{name:18s} {ms:.2f} ms  {n_users * n_items * 4 / ms / 1e6:.0f} GB/s")
ms = t(lambda: x.copy_(S.view(-1)))
print(f"torch copy (read + write) {ms:.2f} ms -> {2 * n_users * n_items * 4 / ms / 1e6:.0f} GB/s total")
ms = t(lambda: S.sum())
print(f"torch sum (read only) {ms:.2f} ms -> {n_users * n_items * 4 / ms / 1e6:.0f} GB/s")
