#!/usr/bin/env python3
"""Measurement probe for score_topk variants inside ONE process (interleaved repetitions, so DVFS /
thermal drift hits every variant alike).  Prints kernel ms (HIP events around the scoring kernel) and
the fraction of the fp32-MFMA peak per variant.

    python tools/score_probe.py [--users 32768] [--items 10000000] [--reps 3]
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
    ap.add_argument("--users", type=int, default=32768)
    ap.add_argument("--items", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--variants", default="pack+mask,row+mask,pack,row")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    I, d, Bu, k = args.items, args.dim, args.users, 20
    V = bench.item_shard(I, d, 0, I, dev)
    U = bench.xavier_(Bu, d, 17, dev, 1_000_000)
    rowptr, col = bench.rated_lists(Bu, I, 50, seed=4)
    cold = np.where(np.random.default_rng(5).random(I) < 0.2)[0]
    bitmap = ops.make_bitmap(I, cold, dev)
    rp, rc = torch.from_numpy(rowptr).to(dev), torch.from_numpy(col).to(dev)
    variants = args.variants.split(",")
    ev = bench.HipEvents(len(variants) * args.reps)
    n = 0
    times = {v: [] for v in variants}
    order = []
    for r in range(args.reps + 1):
        for v in variants:
            pack = v.startswith("pack")
            mask = v.endswith("+mask")
            pair = ev.pairs[n] if r else None
            ops.score_topk(U, None, V, k, rp if mask else None, rc if mask else None, bitmap if mask else None,
                           kernel_events=pair, pack=pack)
            if r:
                order.append(v)
                n += 1
    torch.cuda.synchronize()
    for v, ms in zip(order, ev.elapsed_ms()):
        times[v].append(ms)
    flops = 2.0 * d * Bu * I
    for v in variants:
        ms = times[v]
        print(f"{v:10s} kernel ms {['%.1f' % x for x in ms]}  best {min(ms):.1f} -> "
              f"{flops / (min(ms) * 1e-3) / 1e12 / bench.MFMA_F32_PEAK_TFLOPS:.3f} of fp32-MFMA peak", flush=True)


if __name__ == "__main__":
    main()
