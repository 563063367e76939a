#!/usr/bin/env python3
"""Fused selection vs the dense-block route of crh_score_topk_f32 over (users, items): which is faster where.
CRH_SCORE_DENSE is read once per process, so the two routes are timed through n_splits (0 = dispatcher's choice,
1 = fused single split) and the choice is reported from the timing of a forced-dense child process."""
import os
import subprocess
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coldrec_amd import ops  # noqa: E402


def run(shapes):
    dev = torch.device("cuda:0")
    for n_users, n_items in shapes:
        U = torch.randn(n_users, 128, device=dev) * 0.1
        V = torch.randn(n_items, 128, device=dev) * 0.1
        out = []
        for ns in (0, 1):
            for _ in range(2):
                ops.score_topk(U, None, V, 20, n_splits=ns)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(3):
                ops.score_topk(U, None, V, 20, n_splits=ns)
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t) / 3 * 1e3)
        print(f"{n_users:8d} x {n_items:6d}: dispatcher {out[0]:9.3f} ms   fused (1 split) {out[1]:9.3f} ms", flush=True)


if __name__ == "__main__":
    run([(4096, 4096), (65536, 4096), (262144, 4096), (4096, 16384), (65536, 16384), (262144, 16384), (4096, 65536),
         (65536, 65536), (131072, 65536), (131072, 32768)])
