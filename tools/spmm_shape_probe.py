#!/usr/bin/env python3
"""What bounds the CSR SpMM at CiteULike size: the same N and nnz with (a) the Zipf-shaped degrees of the catalogue,
(b) every row at exactly 12 edges (no imbalance, no heavy rows), each with and without the XCD-pinned column slices."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from coldrec_amd import ops
from coldrec_amd.data.synth import make_dataset
from coldrec_amd.util.databuilder import bipartite_norm_adj_csr
dev = torch.device("cuda:0")
split = make_dataset("citeulike", "item", seed=2, with_content=False)
tr = split.warm_train
_, ru = np.unique(tr[:, 0], return_inverse=True); _, ri = np.unique(tr[:, 1], return_inverse=True)
rowptr, col, val = bipartite_norm_adj_csr(ru, ri, split.user_num, split.item_num)
n = len(rowptr) - 1
rng = np.random.default_rng(0)
graphs = {"zipf (catalogue)": (rowptr, col, val)}
k = 12
rp2 = np.arange(n + 1, dtype=np.int64) * k
graphs["regular 12 edges/row"] = (rp2, np.sort(rng.integers(0, n, (n, k)), 1).reshape(-1).astype(np.int32), rng.random(n * k).astype(np.float32))
for name, (rp_h, cl_h, vl_h) in graphs.items():
    X = torch.randn(n, 128, device=dev); Y = torch.empty_like(X); ACC = torch.zeros_like(X)
    rp, cl, vl = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (rp_h, cl_h, vl_h))
    for tag, kw in (("plain Y = A X", dict(y=Y)), ("Y and acc (layer sum)", dict(y=Y, acc_in=X, acc_out=ACC))):
        sched = ops.SpmmSchedule(rp_h, dev)
        for _ in range(5): ops.spmm_csr(rp, cl, vl, X, sched=sched, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): ops.spmm_csr(rp, cl, vl, X, sched=sched, **kw)
        e1.record(); torch.cuda.synchronize()
        print(f"{name:24s} {tag:24s} nnz={len(cl_h)} heavy rows={int((np.diff(rp_h) > sched.seg).sum())}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us", flush=True)
