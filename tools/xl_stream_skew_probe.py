#!/usr/bin/env python3
"""VERDICT r5 #3, second question: the S-TRAIN-XL dense-Adam pass (crh_adam_dense_f32: four 5.6 GB tables read, four
written) runs 8.0 ms on some boxes and 9.4 ms on others while a two-stream copy of the same table runs the same on both.
Is it the RELATIVE PLACEMENT of the four tables (eight concurrent streams at equal offsets meeting in the same HBM
channels)?  One buffer, the tables carved out of it with a chosen skew between consecutive tables; ms per pass per skew.

    python tools/xl_stream_skew_probe.py [rows]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from coldrec_amd import ops  # noqa: E402


def main():
    rows, d = int(sys.argv[1]) if len(sys.argv) > 1 else 11_000_000, 128
    dev = torch.device("cuda:0")
    n = rows * d
    skews = [0, 256, 4096 + 256, 65536 + 4096 + 256, (1 << 20) + 65536 + 4096 + 256, 3 * 4096, 7 * 256]
    pad = max(skews) * 4 + 4096
    buf = torch.zeros(4 * n + pad // 4 + 1024, dtype=torch.float32, device=dev)
    base = (-buf.data_ptr()) % 4096 // 4            # 4 KiB-aligned start
    for skew in skews:
        t = [buf[base + q * (n + skew // 4): base + q * (n + skew // 4) + n].view(rows, d) for q in range(4)]
        p, g, m, v = t
        p.uniform_(-0.01, 0.01)
        g.uniform_(-0.001, 0.001)
        m.zero_()
        v.zero_()
        for s in range(3):
            ops.adam_dense(p, g, m, v, s + 1, lr=1e-3, zero_grad=False)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
        ev[0].record()
        for s in range(10):
            ops.adam_dense(p, g, m, v, s + 4, lr=1e-3, zero_grad=False)
            ev[s + 1].record()
        torch.cuda.synchronize()
        ms = np.array([ev[s].elapsed_time(ev[s + 1]) for s in range(10)])
        print("skew %8d B between consecutive tables (first table at %#x): median %.3f ms  min %.3f  max %.3f  -> %.0f GB/s"
              % (skew, p.data_ptr(), float(np.median(ms)), ms.min(), ms.max(), 28.0 * n / (np.median(ms) * 1e-3) / 1e9), flush=True)
    # the engine's own allocation (four separate torch tensors), for reference
    del buf
    torch.cuda.empty_cache()
    t = [torch.zeros((rows, d), dtype=torch.float32, device=dev) for _ in range(4)]
    p, g, m, v = t
    p.uniform_(-0.01, 0.01)
    g.uniform_(-0.001, 0.001)
    for s in range(3):
        ops.adam_dense(p, g, m, v, s + 1, lr=1e-3, zero_grad=False)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
    ev[0].record()
    for s in range(10):
        ops.adam_dense(p, g, m, v, s + 4, lr=1e-3, zero_grad=False)
        ev[s + 1].record()
    torch.cuda.synchronize()
    ms = np.array([ev[s].elapsed_time(ev[s + 1]) for s in range(10)])
    print("four torch.zeros tables at %s: median %.3f ms" % ([hex(x.data_ptr()) for x in t], float(np.median(ms))), flush=True)


if __name__ == "__main__":
    main()
