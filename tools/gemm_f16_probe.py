#!/usr/bin/env python3
"""What does the vendor GEMM (hipBLASLt through torch.matmul) sustain in fp16 on this chip with random operands?
Context for the fp16 scoring leg: the 2.5 PF nominal peak is not reachable on random data (power limit)."""
import torch

dev = torch.device("cuda:0")
for (m, n, k) in ((16384, 65536, 1024), (16384, 16384, 8192), (8192, 8192, 8192), (32768, 262144, 256)):
    a = (torch.rand((m, k), device=dev) - 0.5).half()
    b = (torch.rand((n, k), device=dev) - 0.5).half()
    for _ in range(3):
        c = a @ b.T
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        c = a @ b.T
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    tf = 2.0 * m * n * k / (ms * 1e-3) / 1e12
    print(f"torch.matmul fp16 {m} x {k} @ {k} x {n}: {ms:.3f} ms  {tf:.0f} TFLOP/s = {tf / 2500:.3f} of 2.5 PF", flush=True)
    del a, b, c
