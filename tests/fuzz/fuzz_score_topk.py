#!/usr/bin/env python3
"""Randomised parity fuzzing of the scoring kernels against the canonical oracle (tests-style tool: it imports
oracle/).  Draws shapes, widths, k, mask densities, split counts, layouts and dtypes for --minutes, checks
bit-exact scores + indices (fp32, and fp16 on exact-arithmetic tables) and prints a summary line.

    python tests/fuzz/fuzz_score_topk.py --minutes 5 [--seed 0]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from coldrec_amd import ops  # noqa: E402
from oracle import oracle_np as orc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=5.0)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.minutes * 60
    n_cases = n_big = n_seeded = n_many = 0
    while time.time() < t_end:
        half = rng.random() < 0.35
        d = int(rng.choice([16, 32, 64, 128, 256] if half else [8, 16, 32, 64, 128, 256, 24, 100]))
        k = int(rng.choice([1, 5, 10, 20, 20, 20, 33, 64, 100, 128]))
        big = rng.random() < 0.15                       # workgroup-kernel territory
        n_users = int(rng.integers(32768, 34000)) if big else int(rng.integers(1, 700))
        n_items = int(rng.integers(1, 3000)) if big else int(rng.integers(1, 40000))
        # seeded-route territory (catalogues of >= 65 536 items: score_topk_any ranks a prefix by the dense route and seeds the
        # fused selection with it): few users (the item range is cut; prefix = 1/16 of the catalogue) or, fp32, users that fill
        # the chip on their own (no cuts; 4 096-item prefix)
        seeded = (not big) and rng.random() < 0.05
        many = seeded and rng.random() < 0.4
        if seeded:
            n_items = int(rng.integers(65536, 80000))
            if many:
                half, n_users = False, 131072 + int(rng.integers(0, 200))
            d = 16 if half else int(rng.choice([8, 16]))
        quant = half or rng.random() < 0.5              # exact arithmetic / heavy ties
        if quant:
            q = int(rng.choice([2, 4, 8]))
            U = (rng.integers(-q, q + 1, (n_users, d)) / q).astype(np.float32)
            V = (rng.integers(-q, q + 1, (n_items, d)) / q).astype(np.float32)
        else:
            U = (rng.standard_normal((n_users, d)) * 0.3).astype(np.float32)
            V = (rng.standard_normal((n_items, d)) * 0.3).astype(np.float32)
        base = int(rng.choice([0, 0, 31, 4096, 100003]))
        n_glob = base + n_items + int(rng.integers(0, 100))
        mean_r = int(rng.choice([0, 3, 30, 200]))
        if many:
            mean_r = min(mean_r, 3)
        rated = [np.unique(rng.integers(base, base + n_items, rng.poisson(mean_r))) if mean_r else np.zeros(0, np.int64)
                 for _ in range(n_users)]
        rowptr = np.concatenate([[0], np.cumsum([len(r) for r in rated])]).astype(np.int64)
        col = np.concatenate(rated).astype(np.int64) if rowptr[-1] else np.zeros(0, np.int64)
        frac = float(rng.choice([0.0, 0.05, 0.2, 0.9, 1.0]))
        bm_ids = np.where(rng.random(n_glob) < frac)[0] if frac else None
        use_idx = rng.random() < 0.5 and not big and not many
        users = rng.permutation(n_users)[: max(1, n_users // 2)].astype(np.int64) if use_idx else None
        splits = int(rng.choice([0, 0, 1, 2, 7]))
        if seeded:
            splits = 0                                   # a caller that names a split count gets the plain fused selection
        # fp32 launches take the workgroup kernel from 2 M items only; CRH_SCORE_WG=2 (read per call) forces it here
        if big and rng.random() < 0.6:
            os.environ["CRH_SCORE_WG"] = "2"
        else:
            os.environ.pop("CRH_SCORE_WG", None)
        pack = bool(rng.random() < 0.7)
        sel = slice(None) if users is None else users
        nq = n_users if users is None else len(users)
        if users is not None:
            rp_u = np.concatenate([[0], np.cumsum([len(rated[u]) for u in users])]).astype(np.int64)
            col_u = np.concatenate([rated[u] for u in users]).astype(np.int64) if rp_u[-1] else np.zeros(0, np.int64)
        else:
            rp_u, col_u = rowptr, col
        tdt = torch.float16 if half else torch.float32
        tU, tV = torch.from_numpy(U).to(dev).to(tdt), torch.from_numpy(V).to(dev).to(tdt)
        if not half and d in (24, 100):
            pass                                         # ops pads the width (exact)
        srp, src = orc.sort_rated(rp_u, col_u)
        rp_t = torch.from_numpy(srp).to(dev) if rp_u[-1] else None
        rc_t = torch.from_numpy(src).to(dev) if rp_u[-1] else None
        bm_t = ops.make_bitmap(n_glob, bm_ids, dev)
        tu = None if users is None else torch.from_numpy(users.astype(np.int32)).to(dev)
        s, i = ops.score_topk(tU, tu, tV, k, rp_t, rc_t, bm_t, item_base=base, n_splits=splits, pack=pack)
        torch.cuda.synchronize()
        # oracle on a sample of the queried users (all of them when small)
        pick = np.arange(nq) if nq <= 96 else np.sort(rng.choice(nq, 64, replace=False))
        q_users = pick if users is None else users[pick]
        prp = np.concatenate([[0], np.cumsum([len(rated[u]) for u in q_users])]).astype(np.int64)
        pcol = np.concatenate([rated[u] for u in q_users]).astype(np.int64) if prp[-1] else np.zeros(0, np.int64)
        bm_o = orc.make_bitmap(n_glob, bm_ids) if bm_ids is not None and len(bm_ids) else None
        ws, wi = orc.score_topk(U, q_users.astype(np.int64), V, k, prp if prp[-1] else None, pcol if prp[-1] else None,
                                bm_o, item_base=base)
        gs, gi = s.cpu().numpy()[pick], i.cpu().numpy()[pick]
        ok = np.array_equal(gi, wi) and np.array_equal(gs.view(np.uint32), ws.view(np.uint32))
        if not ok:
            print("MISMATCH", dict(half=half, d=d, k=k, n_users=n_users, n_items=n_items, quant=quant, base=base, seeded=seeded,
                                   mean_r=mean_r, frac=frac, use_idx=use_idx, splits=splits, pack=pack, seed=args.seed,
                                   case=n_cases), flush=True)
            sys.exit(1)
        n_cases += 1
        n_big += big
        n_seeded += seeded
        n_many += many
    print(f"fuzz ok: {n_cases} random cases ({n_big} in workgroup-kernel territory, {n_seeded} in seeded-route territory of which "
          f"{n_many} with users that fill the chip) bit-exact vs the oracle, seed {args.seed}")


if __name__ == "__main__":
    main()
