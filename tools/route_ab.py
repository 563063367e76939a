#!/usr/bin/env python3
"""Interleaved A/B of scoring ROUTES in one process: every arm is a set of per-call environment switches
(CRH_SCORE_WG / CRH_SCORE_DMA / CRH_SCORE_SEED are read per call), e.g.

    python tools/route_ab.py --users 131072 --items 1250000 --arms "default;CRH_SCORE_WG=2;CRH_SCORE_WG=0"

Prints the library's route per arm, median kernel ms (HIP events around the scoring launches) and the fraction of the MFMA
peak, and checks that all arms return identical lists."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from coldrec_amd import ops  # noqa: E402

SWITCHES = ("CRH_SCORE_WG", "CRH_SCORE_DMA", "CRH_SCORE_SEED", "CRH_SCORE_SEED_MAX_ITEMS")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=131072)
    ap.add_argument("--items", type=int, default=1_250_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--dtype", choices=["f16", "f32"], default="f32")
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--arms", default="default;CRH_SCORE_WG=2")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    f16 = args.dtype == "f16"
    d, I, Bu, k = args.dim, args.items, args.users, args.k
    peak = 2500e12 if f16 else 157.3e12
    V = bench.item_shard(I, d, 0, I, dev, torch.float16 if f16 else torch.float32)
    U = bench.xavier_(Bu, d, 17, dev, 1_000_000)
    if f16:
        U = U.to(torch.float16)
    rowptr, col = bench.rated_lists(Bu, I, 50, seed=4)
    cold = np.where(np.random.default_rng(5).random(I) < 0.2)[0]
    bitmap = ops.make_bitmap(I, cold, dev)
    rp, rc = torch.from_numpy(rowptr).to(dev), torch.from_numpy(col).to(dev)
    arms = [a.strip() for a in args.arms.split(";")]

    def setenv(arm):
        for s in SWITCHES:
            os.environ.pop(s, None)
        if arm != "default":
            for kv in arm.split("+"):
                key, val = kv.split("=")
                os.environ[key] = val

    outs, res, routes = {}, {a: [] for a in arms}, {}
    for a in arms:
        setenv(a)
        routes[a] = ops.score_topk_route(Bu, I, d, k, half=f16)
        s, i = ops.score_topk(U, None, V, k, rp, rc, bitmap)
        torch.cuda.synchronize()
        outs[a] = (s.clone(), i.clone())
    same = all(torch.equal(outs[a][1], outs[arms[0]][1]) and
               torch.equal(outs[a][0].view(torch.int32), outs[arms[0]][0].view(torch.int32)) for a in arms)
    for _ in range(args.rounds):
        for a in arms:
            setenv(a)
            ev = bench.HipEvents(1)
            ops.score_topk(U, None, V, k, rp, rc, bitmap, kernel_events=ev.pairs[0])
            torch.cuda.synchronize()
            res[a].append(ev.elapsed_ms()[0])
    flops = 2.0 * d * Bu * I
    for a in arms:
        med = float(np.median(res[a]))
        r = routes[a]
        print(f"{args.dtype} d={d} {Bu} x {I}  [{a}]  route {r['route']}{' seeded ' + str(r['prefix_items']) if r['seeded'] else ''} "
              f"cuts {r['n_splits']}: median {med:.2f} ms -> {flops / (med * 1e-3) / peak:.4f} of peak", flush=True)
    print("arms identical:", same, flush=True)


if __name__ == "__main__":
    main()
