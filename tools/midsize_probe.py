#!/usr/bin/env python3
"""Mid-size catalogues (65 K - 1 M items): fused selection vs the dense-block route of crh_score_topk_f32, masks on.
One child process per route (the switches are read once).  Prints ms per call and the fraction of the fp32 MFMA peak.

    python tools/midsize_probe.py            # parent: runs the grid through child processes
"""
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES_A = [(8192, 65536), (8192, 262144), (65536, 65536), (65536, 131072), (65536, 262144), (131072, 131072),
          (131072, 262144), (131072, 524288), (131072, 1048576)]


SHAPES_B = [(16384, 131072), (16384, 524288), (32768, 131072), (32768, 262144), (32768, 1048576), (131072, 2097152),
            (131072, 4194304)]
SHAPES_C = [(8192, 262144), (16384, 131072), (65536, 65536), (32768, 131072), (6040, 3706), (262144, 16384)]
SHAPES_D = [(131072, 65536), (65536, 131072), (32768, 262144), (262144, 32768), (131072, 131072), (524288, 16384),
            (524288, 32768)]
SHAPES_E = [(65536, 131072), (65536, 131072), (131072, 262144)]
SHAPES = {"B": SHAPES_B, "C": SHAPES_C, "D": SHAPES_D, "E": SHAPES_E}.get(os.environ.get("SHAPES"), SHAPES_A)


def child():
    import bench
    from coldrec_amd import ops
    dev = torch.device("cuda:0")
    for n_users, n_items in SHAPES:
        U = bench.xavier_(n_users, 128, 17, dev, 1_000_000)
        V = bench.item_shard(n_items, 128, 0, n_items, dev)
        rowptr, col = bench.rated_lists(n_users, n_items, 50, seed=4)
        cold = np.where(np.random.default_rng(5).random(n_items) < 0.2)[0]
        bitmap = ops.make_bitmap(n_items, cold, dev)
        rp, rc = torch.from_numpy(rowptr).to(dev), torch.from_numpy(col).to(dev)
        for _ in range(2):
            ops.score_topk(U, None, V, 20, rp, rc, bitmap)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            ops.score_topk(U, None, V, 20, rp, rc, bitmap)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / 3 * 1e3
        tf = 2.0 * 128 * n_users * n_items / (ms * 1e-3) / 1e12
        print(f"{os.environ.get('ROUTE'):6s} {n_users:7d} x {n_items:8d}: {ms:9.3f} ms  {tf:6.1f} TF = {tf / 157.3:.3f} of fp32 MFMA peak",
              flush=True)


if __name__ == "__main__":
    if os.environ.get("ROUTE"):
        child()
    else:
        routes = (("fused", {"CRH_SCORE_DENSE": "0"}),
                  ("perwav", {"CRH_SCORE_DENSE": "0", "CRH_SCORE_WG": "0"}),
                  ("dense", {"CRH_SCORE_DENSE_MAX_ITEMS": "2000000", "CRH_SCORE_DENSE_MAX_PAIRS": "1e18"}),
                  ("auto", {}),
                  ("d128", {"CRH_SCORE_DENSE_BLOCK_MB": "128"}), ("d256", {"CRH_SCORE_DENSE_BLOCK_MB": "256"}),
                  ("d512", {"CRH_SCORE_DENSE_BLOCK_MB": "512"}), ("d2048", {"CRH_SCORE_DENSE_BLOCK_MB": "2048"}),
                  ("d4096", {"CRH_SCORE_DENSE_BLOCK_MB": "4096"}), ("d8192", {"CRH_SCORE_DENSE_BLOCK_MB": "8192"}),
                  ("d32768", {"CRH_SCORE_DENSE_BLOCK_MB": "32768"}))
        only = sys.argv[1].split(",") if len(sys.argv) > 1 else None
        for route, env in routes:
            if only and route not in only:
                continue
            e = dict(os.environ, ROUTE=route, **env)
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=e, check=False)
