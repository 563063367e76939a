#!/usr/bin/env python3
"""Where an end-to-end BPR-MF epoch goes (MovieLens shape): every phase of EpochRunner.run timed with a device sync,
the prefetcher wait, and the unsynchronised loop beside it."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coldrec_amd import ops
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
g = torch.Generator().manual_seed(2024)
U0 = torch.nn.init.xavier_uniform_(torch.empty(n_u, d), generator=g)
V0 = torch.nn.init.xavier_uniform_(torch.empty(n_i, d), generator=g)
eng = MFEngine(U0, V0, 1e-3, 1e-4, dev)
runner = EpochRunner(eng, n, B)
np.random.seed(2024)
pref = EpochPrefetcher(smp, B)
for _ in range(3):
    runner.run(*pref.get())
torch.cuda.synchronize()

T = {}
def tick(label, fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    T.setdefault(label, []).append(time.perf_counter() - t); return r

for _ in range(8):
    u, i, j = tick("prefetch.get (wait for the sampler thread)", pref.get)
    tick("h2d triples", lambda: [dst.copy_(torch.as_tensor(src), non_blocking=True) for dst, src in ((runner.u, u), (runner.i, i), (runner.j, j))])
    plans = tick("plans kernel", lambda: ops.build_plans_device(runner.u, runner.i, runner.j, B))
    tick("plans copy", lambda: runner.plans.copy_(plans))
    tick("mf tables", lambda: ops.mf_step_tables(runner.plans, runner.u, runner.i, runner.j, B, eng.user_num, eng.item_num, out=runner.tables))
    sc = tick("scalars host", lambda: ops.adam_step_scalars(eng.step_count + 1, len(runner.steps), eng.lr))
    tick("scalars h2d", lambda: runner.scalars.copy_(torch.from_numpy(sc), non_blocking=True))
    tick("graph replay", lambda: runner.graph.replay())
    eng.step_count += len(runner.steps)
for k, v in T.items():
    print(f"{k:50s} median {np.median(v) * 1e3:7.3f} ms")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    runner.run(*pref.get())
torch.cuda.synchronize()
print("unsynchronised loop: %.3f ms per epoch" % ((time.perf_counter() - t0) * 100), flush=True)
pref.close()
t0 = time.perf_counter()
for _ in range(10):
    smp.epoch(B)
print("sampler alone: %.3f ms per epoch" % ((time.perf_counter() - t0) * 100))
