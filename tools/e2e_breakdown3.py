#!/usr/bin/env python3
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coldrec_amd import _lib
from coldrec_amd.data.synth import make_dataset
from coldrec_amd.sampler import EpochPrefetcher, PairwiseSampler
from coldrec_amd.train import EpochRunner, MFEngine
dev = torch.device("cuda:0")
split = make_dataset("movielens", "item", seed=1, with_content=False)
tr = split.warm_train
_, ru = np.unique(tr[:, 0], return_inverse=True)
_, ri = np.unique(tr[:, 1], return_inverse=True)
n_u, n_i, n, B, d = split.user_num, split.item_num, tr.shape[0], 4096, 128
smp = PairwiseSampler(ru, ri, n_u, n_i)
L = smp._L
bufs = [torch.empty(n, dtype=torch.int32).pin_memory() for _ in range(3)]
# 1. the worker thread alone, back to back
for _ in range(3):
    L.crh_sampler_epoch_async(smp._h, B, *[b.data_ptr() for b in bufs], 0); L.crh_sampler_epoch_wait(smp._h)
ts = []
for _ in range(10):
    t = time.perf_counter(); L.crh_sampler_epoch_async(smp._h, B, *[b.data_ptr() for b in bufs], 0); L.crh_sampler_epoch_wait(smp._h); ts.append(time.perf_counter() - t)
print("worker thread alone, back to back: median %.2f ms" % (np.median(ts) * 1e3))
ts = []
for _ in range(10):
    time.sleep(0.003)
    t = time.perf_counter(); L.crh_sampler_epoch_async(smp._h, B, *[b.data_ptr() for b in bufs], 0); L.crh_sampler_epoch_wait(smp._h); ts.append(time.perf_counter() - t)
print("worker thread with 3 ms gaps: median %.2f ms" % (np.median(ts) * 1e3))
ts = []
for _ in range(10):
    t = time.perf_counter(); smp.epoch(B); ts.append(time.perf_counter() - t)
print("calling thread: median %.2f ms" % (np.median(ts) * 1e3))
g = torch.Generator().manual_seed(2024)
eng = MFEngine(torch.nn.init.xavier_uniform_(torch.empty(n_u, d), generator=g), torch.nn.init.xavier_uniform_(torch.empty(n_i, d), generator=g), 1e-3, 1e-4, dev)
runner = EpochRunner(eng, n, B)
np.random.seed(2024)
pref = EpochPrefetcher(smp, B, device=dev)
for _ in range(3):
    runner.run(*pref.get())
torch.cuda.synchronize()
os.environ["CRH_PREFETCH_TIMING"] = "1"
pref.close()
def loop(tag, fn, dev_out=True, n_it=20):
    pf = EpochPrefetcher(smp, B, device=dev if dev_out else None)
    if not dev_out: pf.timing = []
    for _ in range(3): fn(pf.get())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for e in range(n_it): fn(pf.get())
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / n_it * 1e3
    print("%-46s %.2f ms per epoch; sampler (finish wait) median %.2f ms" % (tag, tot, np.median([r[0] for r in pf.timing[-10:]]) * 1e3), flush=True)
    pf.close()
loop("full: upload + run()", lambda tri: runner.run(*tri))
loop("upload only, no run()", lambda tri: None)
loop("no upload (host arrays), no GPU work", lambda tri: None, dev_out=False)
loop("no upload, graph replay only", lambda tri: runner.graph.replay(), dev_out=False)
x = torch.empty(64 << 20, device=dev)
loop("no upload, one 20 us kernel per epoch", lambda tri: x[:1024].zero_(), dev_out=False)
loop("no upload, 200 tiny kernels per epoch", lambda tri: [x[:1024].zero_() for _ in range(200)], dev_out=False)
