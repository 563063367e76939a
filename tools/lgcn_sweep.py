#!/usr/bin/env python3
"""LightGCN step time (CiteULike shape, hipGraph epoch) under the SpMM tuning switches given on the command line as
NAME=VALUE pairs (each combination in a fresh process: the switches are read once)."""
import itertools, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ["CR_ROOT"])
from coldrec_amd.data.synth import make_dataset
from coldrec_amd.sampler import PairwiseSampler
from coldrec_amd.train import EpochRunner, LGCNEngine
from coldrec_amd.util.databuilder import bipartite_norm_adj_csr
dev = torch.device("cuda:0")
split = make_dataset("citeulike", "item", seed=2, with_content=False)
tr = split.warm_train
_, ru = np.unique(tr[:, 0], return_inverse=True); _, ri = np.unique(tr[:, 1], return_inverse=True)
n_u, n_i, n, B, d = split.user_num, split.item_num, tr.shape[0], 4096, 128
smp = PairwiseSampler(ru, ri, n_u, n_i); smp.seed(2024); u, i, j = smp.epoch(B)
g = torch.Generator().manual_seed(2024)
U0 = torch.nn.init.xavier_uniform_(torch.empty(n_u, d), generator=g); V0 = torch.nn.init.xavier_uniform_(torch.empty(n_i, d), generator=g)
rowptr, col, val = bipartite_norm_adj_csr(ru, ri, n_u, n_i)
eng = LGCNEngine(U0, V0, rowptr, col, val, 3, 1e-3, 1e-4, dev)
runner = EpochRunner(eng, n, B)
tu, ti, tj = (torch.from_numpy(x).to(dev) for x in (u, i, j))
for _ in range(3): runner.run(tu, ti, tj)
torch.cuda.synchronize(); ts = []
for _ in range(5):
    t0 = time.perf_counter(); runner.run(tu, ti, tj); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / len(runner.steps))
print("RESULT %.2f us/step (min %.2f) loss %.6f" % (np.median(ts) * 1e6, min(ts) * 1e6, float(runner.losses[-1].sum())))
'''
axes = [a.split("=") for a in sys.argv[1:]]
names = [a[0] for a in axes]
for combo in itertools.product(*[a[1].split(",") for a in axes]):
    env = dict(os.environ, CR_ROOT=ROOT, **dict(zip(names, combo)))
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    res = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    print(dict(zip(names, combo)), res[0] if res else out.stderr[-400:], flush=True)
