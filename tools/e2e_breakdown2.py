#!/usr/bin/env python3
import os, sys, time, threading
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coldrec_amd import ops
from coldrec_amd.data.synth import make_dataset
from coldrec_amd.sampler import EpochPrefetcher, PairwiseSampler
from coldrec_amd.train import EpochRunner, MFEngine
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
print("affinity", len(os.sched_getaffinity(0)))
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
u, i, j = smp.epoch(B)
for _ in range(3):
    runner.run(u, i, j)
torch.cuda.synchronize()
# host cost of the pieces WITHOUT device syncs
def host_ms(fn, reps=10):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    dt = (time.perf_counter() - t) / reps; torch.cuda.synchronize(); return dt * 1e3
print("host ms graph.replay() call", host_ms(lambda: runner.graph.replay()))
print("host ms runner.run (device-resident triples)", host_ms(lambda: runner.run(runner.u, runner.i, runner.j)))
print("host ms runner.run (host triples)", host_ms(lambda: runner.run(u, i, j)))
pu, pi, pj = (torch.from_numpy(x).pin_memory() for x in (u, i, j))
print("host ms runner.run (pinned host triples)", host_ms(lambda: runner.run(pu, pi, pj)))
def sampler_ms():
    t = time.perf_counter(); smp.epoch(B); return (time.perf_counter() - t) * 1e3
print("sampler alone ms", np.median([sampler_ms() for _ in range(5)]))
# sampler on a thread while the main thread (a) sleeps (b) spins in synchronize behind a replay (c) enqueues run()
for mode in ("sleep", "sync", "run"):
    res = []
    for _ in range(5):
        out = {}
        th = threading.Thread(target=lambda: out.setdefault("ms", sampler_ms()))
        th.start()
        if mode == "sleep": time.sleep(0.004)
        elif mode == "sync": runner.graph.replay(); torch.cuda.synchronize()
        else: runner.run(pu, pi, pj)
        th.join(); torch.cuda.synchronize()
        res.append(out["ms"])
    print("sampler thread ms while main does", mode, np.median(res))
