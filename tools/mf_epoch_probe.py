#!/usr/bin/env python3
"""Phase timeline of one step (step 10) of the one-launch BPR-MF epoch (profile build: make -C coldrec_amd/csrc profile;
CRH_LIB=coldrec_amd/lib/libcoldrec_hip_profile.so python tools/mf_epoch_probe.py)."""
import ctypes, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from coldrec_amd import _lib, ops
from coldrec_amd.data.synth import make_dataset
from coldrec_amd.sampler import PairwiseSampler
from coldrec_amd.train import EpochRunner, MFEngine
dev = torch.device("cuda:0")
split = make_dataset("movielens", "item", seed=1, with_content=False)
tr = split.warm_train
_, ru = np.unique(tr[:, 0], return_inverse=True); _, ri = np.unique(tr[:, 1], return_inverse=True)
n_u, n_i, n, B, d = split.user_num, split.item_num, tr.shape[0], 4096, 128
smp = PairwiseSampler(ru, ri, n_u, n_i); smp.seed(2024); u, i, j = smp.epoch(B)
g = torch.Generator().manual_seed(2024)
U0 = torch.nn.init.xavier_uniform_(torch.empty(n_u, d), generator=g); V0 = torch.nn.init.xavier_uniform_(torch.empty(n_i, d), generator=g)
eng = MFEngine(U0, V0, 1e-3, 1e-4, dev)
runner = EpochRunner(eng, n, B)
tu, ti, tj = (torch.from_numpy(x).to(dev) for x in (u, i, j))
for _ in range(3): runner.run(tu, ti, tj)
torch.cuda.synchronize(); ts = []
for _ in range(5):
    t0 = time.perf_counter(); runner.run(tu, ti, tj); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / len(runner.steps))
print("step %.2f us (min %.2f), %d steps per epoch, epoch launch blocks %d" % (np.median(ts) * 1e6, min(ts) * 1e6, len(runner.steps), eng._eblocks))
L = _lib.lib()
if hasattr(L, "crh_profile_mf_clocks") and eng._eblocks:
    nb = eng._eblocks
    buf = np.zeros(8 * nb, np.uint64)
    assert L.crh_profile_mf_clocks(ctypes.c_void_p(buf.ctypes.data), 4 * nb) == 0
    t = buf.reshape(nb, 8)[:, :6].astype(np.int64)
    t0 = t[:, 0].min()
    names = ["batch sums", "light rows", "heavy rows", "update + sums", "barrier"]
    d_ = (t[:, 1:] - t[:, :-1]) * 0.01
    print("phase (us): mean / max over workgroups")
    for k, nm in enumerate(names):
        print("  %-14s %6.2f / %6.2f" % (nm, d_[:, k].mean(), d_[:, k].max()))
    print("  step start spread %.2f us; arrival at the barrier: first %.2f, median %.2f, last %.2f us after the step's first start; "
          "release: first %.2f, last %.2f" % ((t[:, 0].max() - t0) * 0.01, (t[:, 4].min() - t0) * 0.01, (np.median(t[:, 4]) - t0) * 0.01,
                                              (t[:, 4].max() - t0) * 0.01, (t[:, 5].min() - t0) * 0.01, (t[:, 5].max() - t0) * 0.01))
