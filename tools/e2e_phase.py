import sys, time, os, types, argparse
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from coldrec_amd.data.synth import make_dataset
from coldrec_amd.util.databuilder import ColdStartDataBuilder
from coldrec_amd.model import AVAILABLE_MODELS
import coldrec_amd.model.BaseRecommender as BR
split = make_dataset("movielens", "item", seed=1, with_content=False)
info = split.info
data = ColdStartDataBuilder(split.as_lists("warm_train"), split.as_lists("warm_val"), split.as_lists("cold_val"),
    split.as_lists("overall_val"), split.as_lists("warm_test"), split.as_lists("cold_test"), split.as_lists("overall_test"),
    info["user_num"], info["item_num"], info["warm_user"], info["warm_item"], info["cold_user"], info["cold_item"], None, None)
a = dict(dataset="ml", model="MF", epochs=25, layers=2, topN="10,20", bs=4096, emb_size=128, lr=0.001, reg=0.0001, runs=1,
         seed=2024, use_gpu=True, save_emb=False, gpu_id=0, cold_object="item", backbone="MF", early_stop=100, eval_every=1)
cfg = types.SimpleNamespace(args=argparse.Namespace(**a), data=data, device=torch.device("cuda:0"))
tr = AVAILABLE_MODELS["MF"](cfg)
T = {}
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*x, **k):
        torch.cuda.synchronize(); t = time.perf_counter(); r = f(*x, **k); torch.cuda.synchronize()
        T.setdefault(label, []).append(time.perf_counter() - t); return r
    setattr(obj, name, g)
from coldrec_amd import train, sampler, ops
wrap(train.EpochRunner, "run", "runner.run (copy+plans+graph)")
wrap(sampler.EpochPrefetcher, "get", "prefetch.get (wait for sampler)")
wrap(BR.BaseColdStartTrainer, "_topk_arrays", "eval: score_topk + D2H")
wrap(BR, "ranking_metrics", "metrics")
wrap(BR.BaseColdStartTrainer, "fast_evaluation", "fast_evaluation total")
t0 = time.perf_counter(); tr.train(); torch.cuda.synchronize(); tot = time.perf_counter() - t0
print("epoch ms", tot / 25 * 1e3)
for k, v in T.items():
    print(f"{k:40s} n={len(v):3d} median {np.median(v)*1e3:7.2f} ms  last-10 mean {np.mean(v[-10:])*1e3:7.2f} ms")
