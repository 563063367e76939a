#!/usr/bin/env python3
"""HBM roofline of crh_mask_topk_f32 (dense-block fallback of _evaluate): one streaming read of the block."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from coldrec_amd import ops

dev = torch.device("cuda:0")
for n_users, n_items in ((4096, 1_000_000), (1024, 10_000_000)):
    S = torch.randn(n_users, n_items, device=dev)
    rowptr, col = bench.rated_lists(n_users, n_items, 50, seed=4)
    rp, rc = torch.from_numpy(rowptr).to(dev), torch.from_numpy(col).to(dev)
    bm = ops.make_bitmap(n_items, np.where(np.random.default_rng(5).random(n_items) < 0.2)[0], dev)
    for wb in (False, True):
        for _ in range(2):
            ops.mask_topk(S, 20, rp, rc, bm, write_back=wb)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.mask_topk(S, 20, rp, rc, bm, write_back=wb)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"mask_topk {n_users} x {n_items} write_back={wb}: {ms:.2f} ms, {n_users * n_items * 4 / ms / 1e6:.0f} GB/s read "
              f"({n_users * n_items / ms / 1e6:.2f}e9 ranked items/s)")
    del S
