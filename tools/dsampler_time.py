#!/usr/bin/env python3
"""Per-kernel time of one device-sampler epoch at the MovieLens shape (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coldrec_amd.data.synth import make_dataset
from coldrec_amd.sampler import DeviceSampler
shape = sys.argv[1] if len(sys.argv) > 1 else "movielens"
split = make_dataset(shape, "item", seed=1, with_content=False)
tr = split.warm_train
_, ru = np.unique(tr[:, 0], return_inverse=True)
_, ri = np.unique(tr[:, 1], return_inverse=True)
ds = DeviceSampler(ru, ri, split.user_num, split.item_num, "cuda:0")
ds.seed(2024)
out = ds.epoch(4096)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    ds.launch(4096, out)
torch.cuda.synchronize()
print("ms per epoch (10 launches back to back):", (time.perf_counter() - t0) * 100, "blocks", ds.n_blocks, "status", ds.get_state()[2])
