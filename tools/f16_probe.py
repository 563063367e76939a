#!/usr/bin/env python3
"""fp16 score_topk probe: kernel ms (HIP events around the scoring kernel) and fraction of the 2.5 PF dense fp16
peak.  Ablations come from the profile build (CRH_LIB=coldrec_amd/lib/libcoldrec_hip_profile.so CRH_SCORE_ABLATE=n),
one process per setting because the switches are read once.

    python tools/f16_probe.py [--users 131072] [--items 10000000] [--dim 256] [--reps 3] [--nomask]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from coldrec_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=131072)
    ap.add_argument("--items", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=256)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--nomask", action="store_true")
    ap.add_argument("--tag", default="")
    ap.add_argument("--splits", type=int, default=0)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    I, d, Bu, k = args.items, args.dim, args.users, args.k
    V = bench.item_shard(I, d, 0, I, dev, torch.float16)
    U = bench.xavier_(Bu, d, 17, dev, 1_000_000).to(torch.float16)
    rp = rc = bitmap = None
    if not args.nomask:
        rowptr, col = bench.rated_lists(Bu, I, 50, seed=4)
        cold = np.where(np.random.default_rng(5).random(I) < 0.2)[0]
        bitmap = ops.make_bitmap(I, cold, dev)
        rp, rc = torch.from_numpy(rowptr).to(dev), torch.from_numpy(col).to(dev)
    ev = bench.HipEvents(args.reps)
    ops.score_topk(U, None, V, k, rp, rc, bitmap, n_splits=args.splits)
    for r in range(args.reps):
        ops.score_topk(U, None, V, k, rp, rc, bitmap, n_splits=args.splits, kernel_events=ev.pairs[r])
    torch.cuda.synchronize()
    ms = ev.elapsed_ms()
    flops = 2.0 * d * Bu * I
    abl = os.environ.get("CRH_SCORE_ABLATE", "0")
    print(f"f16 {args.tag} users={Bu} items={I} d={d} k={k} mask={not args.nomask} splits={args.splits} ablate={abl}: kernel ms "
          f"{['%.1f' % x for x in ms]} best {min(ms):.1f} -> {flops / (min(ms) * 1e-3) / 1e12:.0f} TF = "
          f"{flops / (min(ms) * 1e-3) / 1e12 / 2500:.3f} of 2.5 PF", flush=True)


if __name__ == "__main__":
    main()
