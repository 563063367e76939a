#!/usr/bin/env python3
"""S-TRAIN-XL SpMM, where do the 30 ms go?  Times crh_spmm_csr_f32 over the USER-row half and the ITEM-row half of the XL
graph separately (row-block sub-matrices, x = the whole table), for user tables of different sizes at the same edge count:
an item row gathers USER rows, so a 250 K-user table (128 MB of 512-byte rows) is Infinity-Cache resident while the 1 M-user
table (512 MB) is not -- the best case a column-blocked pass over the item rows could reach, without its partial-sum traffic.

    python tools/xl_spmm_probe.py [n_inter]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from coldrec_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
n_inter = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
d = 128


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for n_u, n_i in ((1_000_000, 10_000_000), (250_000, 10_000_000), (1_000_000, 2_500_000)):
    rowptr, col, val, deg = bench.xl_graph(dev, n_u, n_i, n_inter, 7)
    N = n_u + n_i
    X = bench.xavier_(N, d, 1, dev, n_i)
    Y = torch.empty_like(X)
    A = torch.empty_like(X)
    sched = ops.SpmmSchedule(rowptr.cpu().numpy(), dev, col=col, val=val)
    whole = timed(lambda: ops.spmm_csr(rowptr, col, val, X, y=Y, acc_in=X, s_in=1.0, acc_out=A, s_out=1.0, sched=sched))
    out = {"n_u": n_u, "n_i": n_i, "E2": int(col.numel()), "whole_ms": round(whole, 2)}
    for name, r0, r1 in (("user_rows", 0, n_u), ("item_rows", n_u, N)):
        e0, e1 = int(rowptr[r0]), int(rowptr[r1])
        rp = (rowptr[r0:r1 + 1] - e0).contiguous()
        c, v = col[e0:e1].contiguous(), val[e0:e1].contiguous()
        sc = ops.SpmmSchedule(rp.cpu().numpy(), dev, seg=sched.seg, col=c, val=v)
        ms = timed(lambda: ops.spmm_csr(rp, c, v, X, y=Y[r0:r1], acc_in=X[r0:r1], s_in=1.0, acc_out=A[r0:r1], s_out=1.0, sched=sc))
        gathered = (e1 - e0) * d * 4
        out[name] = {"rows": r1 - r0, "edges": e1 - e0, "ms": round(ms, 2), "gather_TBps": round(gathered / ms / 1e9, 2),
                     "gathered_table_MB": round((n_i if name == "user_rows" else n_u) * d * 4 / 1e6),
                     "heavy_wgs": int(sc.c.n_multi)}
    print(out, flush=True)
    del rowptr, col, val, X, Y, A, sched
    torch.cuda.empty_cache()
