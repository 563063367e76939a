import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from coldrec_amd.data.synth import make_dataset
from coldrec_amd.sampler import PairwiseSampler
from coldrec_amd.train import EpochRunner, MFEngine
from coldrec_amd import ops
dev = torch.device("cuda:0")
B = 4096
d = int(sys.argv[1]) if len(sys.argv) > 1 else 128
optim = sys.argv[2] if len(sys.argv) > 2 else "adam"
split = make_dataset("movielens", "item", seed=1, with_content=False)
tr = split.warm_train
_, ru = np.unique(tr[:, 0], return_inverse=True)
_, ri = np.unique(tr[:, 1], return_inverse=True)
n_u, n_i, n = split.user_num, split.item_num, tr.shape[0]
smp = PairwiseSampler(ru, ri, n_u, n_i); smp.seed(2024)
u, i, j = smp.epoch(B)
g = torch.Generator().manual_seed(2024)
U0 = torch.nn.init.xavier_uniform_(torch.empty(n_u, d), generator=g)
V0 = torch.nn.init.xavier_uniform_(torch.empty(n_i, d), generator=g)
eng = MFEngine(U0, V0, 1e-3, 1e-4, dev, optimizer=optim)
tu, ti, tj = (torch.from_numpy(x).to(dev) for x in (u, i, j))
r = EpochRunner(eng, n, B)
for _ in range(4): r.run(tu, ti, tj)
torch.cuda.synchronize()
N = 40
def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / N * 1e3
print("d", d, optim)
print("full run()            %.3f ms" % timed(lambda: r.run(tu, ti, tj)))
print("graph replay only     %.3f ms" % timed(lambda: r.graph.replay()))
# preamble alone
u2, i2, j2 = (torch.empty_like(x) for x in (tu, ti, tj))
plans2 = ops.build_plans_device(tu, ti, tj, B)
tables2 = ops.mf_step_tables(plans2, tu, ti, tj, B, eng.user_num, eng.item_num)
def pre():
    for dst, src in ((u2, tu), (i2, ti), (j2, tj)): dst.copy_(src, non_blocking=True)
    ops.build_plans_device(u2, i2, j2, B, out=plans2)
    ops.mf_step_tables(plans2, u2, i2, j2, B, eng.user_num, eng.item_num, out=tables2)
print("preamble alone        %.3f ms" % timed(pre))
side = torch.cuda.Stream(dev)
def both():
    with torch.cuda.stream(side):
        pre()
    r.graph.replay()
print("graph + preamble on a side stream  %.3f ms" % timed(both))
