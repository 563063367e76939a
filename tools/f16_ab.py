#!/usr/bin/env python3
"""Interleaved A/B of the fp16 (or, with --dtype f32, the fp32) workgroup scoring kernels in ONE process: the LDS-DMA
kernel (CRH_SCORE_DMA=2) against the register-staged ring kernel (CRH_SCORE_DMA=0); both switches are read per call.
Prints per-round kernel ms (HIP events around the scoring launch) and the fraction of the MFMA peak, and checks that
both arms return identical lists.

    python tools/f16_ab.py [--users 131072] [--items 10000000] [--dim 256] [--rounds 4] [--dtype f16]
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
    ap.add_argument("--dim", type=int, default=0)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--dtype", choices=["f16", "f32"], default="f16")
    ap.add_argument("--nomask", action="store_true")
    ap.add_argument("--arms", default="0,2", help="CRH_SCORE_DMA values to interleave; 'V/noseed' also sets CRH_SCORE_SEED=0")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    f16 = args.dtype == "f16"
    d = args.dim or (256 if f16 else 128)
    peak = 2500e12 if f16 else 157.3e12
    I, Bu, k = args.items, args.users, args.k
    V = bench.item_shard(I, d, 0, I, dev, torch.float16 if f16 else torch.float32)
    U = bench.xavier_(Bu, d, 17, dev, 1_000_000)
    if f16:
        U = U.to(torch.float16)
    rp = rc = bitmap = None
    if not args.nomask:
        rowptr, col = bench.rated_lists(Bu, I, 50, seed=4)
        cold = np.where(np.random.default_rng(5).random(I) < 0.2)[0]
        bitmap = ops.make_bitmap(I, cold, dev)
        rp, rc = torch.from_numpy(rowptr).to(dev), torch.from_numpy(col).to(dev)
    arms = [a.strip() for a in args.arms.split(",")]
    os.environ["CRH_SCORE_WG"] = "2" if not f16 else os.environ.get("CRH_SCORE_WG", "1")
    flops = 2.0 * d * Bu * I
    res = {a: [] for a in arms}
    outs = {}
    def setenv(a):
        os.environ["CRH_SCORE_DMA"] = a.split("/")[0]
        os.environ["CRH_SCORE_SEED"] = "0" if a.endswith("/noseed") else "1"

    for a in arms:                      # warm-up + result capture
        setenv(a)
        s, i = ops.score_topk(U, None, V, k, rp, rc, bitmap)
        torch.cuda.synchronize()
        outs[a] = (s.clone(), i.clone())
    same = all(torch.equal(outs[a][1], outs[arms[0]][1]) and
               torch.equal(outs[a][0].view(torch.int32), outs[arms[0]][0].view(torch.int32)) for a in arms)
    for r in range(args.rounds):
        for a in arms:
            setenv(a)
            ev = bench.HipEvents(1)
            ops.score_topk(U, None, V, k, rp, rc, bitmap, kernel_events=ev.pairs[0])
            torch.cuda.synchronize()
            res[a].append(ev.elapsed_ms()[0])
    for a in arms:
        ms = res[a]
        med = float(np.median(ms))
        print(f"{args.dtype} d={d} users={Bu} items={I} mask={not args.nomask} CRH_SCORE_DMA={a}: ms {['%.1f' % x for x in ms]} "
              f"median {med:.1f} -> {flops / (med * 1e-3) / peak:.4f} of peak (best {flops / (min(ms) * 1e-3) / peak:.4f})", flush=True)
    print("arms identical:", same, flush=True)


if __name__ == "__main__":
    main()
