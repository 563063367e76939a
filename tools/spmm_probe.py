#!/usr/bin/env python3
"""Time crh_spmm_csr_f32 alone on the CiteULike-shaped and MovieLens-shaped synthetic graphs (d=128)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from coldrec_amd import ops
from coldrec_amd.data.synth import make_dataset
from coldrec_amd.util.databuilder import bipartite_norm_adj_csr

dev = torch.device("cuda:0")
for shape, seed in (() if "--xl" in sys.argv else (("citeulike", 2), ("movielens", 1))):
    split = make_dataset(shape, "item", seed=seed, with_content=False)
    tr = split.warm_train
    _, ru = np.unique(tr[:, 0], return_inverse=True)
    _, ri = np.unique(tr[:, 1], return_inverse=True)
    rowptr, col, val = bipartite_norm_adj_csr(ru, ri, split.user_num, split.item_num)
    n = len(rowptr) - 1
    deg = np.diff(rowptr)
    X = torch.randn(n, 128, device=dev)
    Y = torch.empty_like(X)
    rp, cl, vl = (torch.from_numpy(a).to(dev) for a in (rowptr, col, val))
    sched = ops.SpmmSchedule(rowptr, dev)
    for name, sc in (("sched", sched), ("rows", None)):
        for _ in range(5):
            ops.spmm_csr(rp, cl, vl, X, y=Y, sched=sc)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            ops.spmm_csr(rp, cl, vl, X, y=Y, sched=sc)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 200 * 1e3
        byts = len(col) * 8 + (n + 1) * 8 + 2 * n * 128 * 4
        print(f"{shape:10s} {name:5s} N={n} nnz={len(col)} maxdeg={deg.max()} heavy={int((deg > 64).sum())}: "
              f"{us:.1f} us/SpMM, algorithmic {byts / us / 1e3:.0f} GB/s, gathers {len(col) * 512 / us / 1e3:.0f} GB/s")


def xl(n=4_000_000, avg_deg=50, d=128):
    """HBM-bound case: random graph far beyond every cache (n rows, n*avg_deg edges), rows in one piece."""
    torch.manual_seed(0)
    deg = torch.randint(avg_deg // 2, avg_deg * 3 // 2 + 1, (n,), device=dev)
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(deg, 0)
    nnz = int(rowptr[-1])
    col = torch.randint(0, n, (nnz,), device=dev, dtype=torch.int32)
    val = torch.rand(nnz, device=dev)
    X = torch.randn(n, d, device=dev)
    Y = torch.empty_like(X)
    for _ in range(2):
        ops.spmm_csr(rowptr, col, val, X, y=Y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.spmm_csr(rowptr, col, val, X, y=Y)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    byts = nnz * 8 + (n + 1) * 8 + 2 * n * d * 4
    print(f"XL random graph N={n} nnz={nnz} d={d}: {ms:.2f} ms/SpMM, algorithmic {byts / ms / 1e6:.0f} GB/s "
          f"(SURVEY 8(d) formula), gathered rows {nnz * d * 4 / ms / 1e6:.0f} GB/s")


if len(sys.argv) > 1 and sys.argv[1] == "--xl":
    xl()
