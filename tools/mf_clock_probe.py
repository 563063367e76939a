#!/usr/bin/env python3
"""Per-workgroup timeline of the one-launch BPR-MF step (profile build: make -C coldrec_amd/csrc profile;
CRH_LIB=coldrec_amd/lib/libcoldrec_hip_profile.so CRH_MF_ABLATE=16[+bits] python tools/mf_clock_probe.py)."""
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
print("step %.2f us (min %.2f), %d steps per epoch" % (np.median(ts) * 1e6, min(ts) * 1e6, len(runner.steps)))
L = _lib.lib()
if hasattr(L, "crh_profile_mf_clocks"):
    nb = int(ops.mf_step_parts(n_u + n_i, d))
    buf = np.zeros(2 * nb, np.uint64)
    assert L.crh_profile_mf_clocks(ctypes.c_void_p(buf.ctypes.data), nb) == 0
    st, en = buf[0::2].astype(np.int64), buf[1::2].astype(np.int64)
    ok = st > 0
    t0 = st[ok].min()
    print("blocks %d; first start -> last end %.2f us; block life mean %.2f us, max %.2f us" % (
        ok.sum(), (en[ok].max() - t0) * 0.01, (en[ok] - st[ok]).mean() * 0.01, (en[ok] - st[ok]).max() * 0.01))
    edges = np.arange(0, (en[ok].max() - t0) + 50, 50)
    hs, _ = np.histogram(st[ok] - t0, edges); he, _ = np.histogram(en[ok] - t0, edges)
    print("per 0.5 us [started/finished]:", " ".join("%d/%d" % (a, b) for a, b in zip(hs, he)))
    order = np.argsort(en[ok])[-8:]
    idx = np.nonzero(ok)[0][order]
    print("last blocks to finish (block id: start, end us):", [(int(b), round((st[b] - t0) * 0.01, 2), round((en[b] - t0) * 0.01, 2)) for b in idx])
