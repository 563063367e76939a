#!/usr/bin/env python3
"""The reference's own library calls (model/MF.py:12-29, model/LightGCN.py:14-29,86-96, model/BaseRecommender.py:172-183)
run on the SAME MI355X through stock PyTorch-ROCm -- what a ColdRec checkout does with --use_gpu -- next to the HIP path.
Plain torch only (no oracle import): synthetic shapes of bench.py's training / headline legs."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from coldrec_amd.util.databuilder import bipartite_norm_adj_csr  # noqa: E402

dev = torch.device("cuda:0")


def bpr_loss(u, p, n):                      # util/utils.py:25-29
    pos = (u * p).sum(1)
    neg = (u * n).sum(1)
    return torch.mean(-torch.log(10e-6 + torch.sigmoid(pos - neg)))


def l2_reg(reg, *embs):                     # util/utils.py:44-48
    loss = 0
    for e in embs:
        loss = loss + torch.norm(e, p=2) / e.shape[0]
    return loss * reg


def train(name, n_u, n_i, n_pairs, d, B, layers, steps=60):
    rng = np.random.default_rng(1)
    U = torch.nn.Parameter(torch.nn.init.xavier_uniform_(torch.empty(n_u, d, device=dev)))
    V = torch.nn.Parameter(torch.nn.init.xavier_uniform_(torch.empty(n_i, d, device=dev)))
    opt = torch.optim.Adam([U, V], lr=1e-3)
    adj = None
    if layers:
        pairs = np.unique(np.stack([rng.integers(0, n_u, n_pairs), rng.integers(0, n_i, n_pairs)], 1), axis=0)
        rowptr, col, val = bipartite_norm_adj_csr(pairs[:, 0], pairs[:, 1], n_u, n_i)
        rows = np.repeat(np.arange(n_u + n_i), np.diff(rowptr))
        adj = torch.sparse_coo_tensor(np.stack([rows, col]), val, (n_u + n_i, n_u + n_i)).coalesce().to(dev)
    tri = [tuple(torch.from_numpy(rng.integers(0, n, B)).to(dev) for n in (n_u, n_i, n_i)) for _ in range(8)]

    def step(s):
        u, i, j = tri[s % 8]
        if layers:                          # model/LightGCN.py:86-96
            ego = torch.cat([U, V], 0)
            outs = [ego]
            for _ in range(layers):
                ego = torch.sparse.mm(adj, ego)
                outs.append(ego)
            out = torch.mean(torch.stack(outs, dim=1), dim=1)
            ue, ie = out[:n_u], out[n_u:]
        else:
            ue, ie = U, V
        a, b, c = ue[u], ie[i], ie[j]
        loss = bpr_loss(a, b, c) + l2_reg(1e-4, a, b, c)
        opt.zero_grad()
        loss.backward()
        opt.step()

    for s in range(5):
        step(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        step(s)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"torch-rocm {name}: {ms:.3f} ms per optimiser step = {B / ms * 1e3:.3g} triples/s", flush=True)


def evaluate(n_users=1024, n_items=10_000_000, d=128, k=20):
    V = bench.item_shard(n_items, d, 0, n_items, dev)
    U = bench.xavier_(n_users, d, 17, dev, 1_000_000)
    rowptr, col = bench.rated_lists(n_users, n_items, 50, seed=4)
    cold = torch.from_numpy(np.where(np.random.default_rng(5).random(n_items) < 0.2)[0]).to(dev)
    rated = [torch.from_numpy(col[rowptr[r]:rowptr[r + 1]].astype(np.int64)).to(dev) for r in range(n_users)]

    def block():                            # model/BaseRecommender.py:172-183 for one user block
        S = U @ V.T
        for r in range(n_users):
            S[r, rated[r]] = -10e8
        S[:, cold] = -10e8
        return torch.topk(S, k, dim=1, largest=True, sorted=True)

    block()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        block()
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / 2
    print(f"torch-rocm eval block {n_users} users x {n_items} items: {sec * 1e3:.1f} ms = {n_users * n_items / sec:.3g} ranked items/s "
          f"(per-user mask loop included, as the reference runs it)", flush=True)

    def block_vec():                        # the same with the rated mask applied in one indexed write
        S = U @ V.T
        S[rows_all, cols_all] = -10e8
        S[:, cold] = -10e8
        return torch.topk(S, k, dim=1, largest=True, sorted=True)

    rows_all = torch.from_numpy(np.repeat(np.arange(n_users), np.diff(rowptr[:n_users + 1]))).to(dev)
    cols_all = torch.from_numpy(col[:rowptr[n_users]].astype(np.int64)).to(dev)
    block_vec()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        block_vec()
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / 3
    print(f"torch-rocm eval block, vectorised masks: {sec * 1e3:.1f} ms = {n_users * n_items / sec:.3g} ranked items/s", flush=True)


if __name__ == "__main__":
    train("BPR-MF S-ML (6040 x 3706, d=128, B=4096)", 6040, 3706, 0, 128, 4096, 0)
    train("LightGCN L=3 S-CUL (5551 x 16980, d=128, B=4096)", 5551, 16980, 131000, 128, 4096, 3)
    evaluate()
