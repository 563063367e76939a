#!/usr/bin/env python3
"""Where a CLI training epoch's wall time goes (MovieLens-shaped synthetic, BPR-MF): the trainer's own methods are
wrapped with timers (perf_counter around each call, GPU synchronised at the phase ends the trainer itself synchronises).

    python -m coldrec_amd.main --make_synthetic movielens --dataset movielens --data_root /tmp/crdata
    python tools/cli_epoch_breakdown.py /tmp/crdata
"""
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from coldrec_amd import main as cli  # noqa: E402
from coldrec_amd.model import BaseRecommender as base  # noqa: E402
from coldrec_amd import train, sampler  # noqa: E402
from coldrec_amd.util import evaluator  # noqa: E402

T = collections.defaultdict(float)
N = collections.defaultdict(int)


def wrap(obj, name, label=None, sync=False):
    f = getattr(obj, name)
    label = label or name

    def g(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        if sync:
            torch.cuda.synchronize()
        T[label] += time.perf_counter() - t
        N[label] += 1
        return r
    setattr(obj, name, g)


def run(data_root):
    wrap(sampler.EpochPrefetcher, "get", "prefetcher.get")
    wrap(train.EpochRunner, "run", "runner.run (enqueue)")
    wrap(base.BaseColdStartTrainer, "fast_evaluation", "fast_evaluation (total)")
    wrap(base.BaseColdStartTrainer, "_metrics", "  _metrics")
    wrap(base.BaseColdStartTrainer, "_topk_device", "    _topk_device (enqueue)")
    wrap(base, "ranking_metrics", "    ranking_metrics (host)")
    wrap(base.BaseColdStartTrainer, "save", "  save (best checkpoint)")
    args = cli.parse_args(["--dataset", "movielens", "--data_root", data_root, "--model", "MF", "--emb_size", "128",
                           "--epochs", "40", "--early_stop", "1000", "--save_emb", "false", "--result_dir", "/tmp/crres"])
    config = cli.Config(args, args.data_root)
    cli.set_seed(args.seed, True)
    model = cli.model_factory(config)
    t = time.perf_counter()
    model.train()
    total = time.perf_counter() - t
    ep = model.epochs_ran
    print(f"train(): {total / ep * 1e3:.2f} ms per epoch over {ep} epochs")
    for k, v in T.items():
        print(f"  {k:32s} {v / ep * 1e3:7.3f} ms per epoch ({N[k]} calls)")


if __name__ == "__main__":
    run(sys.argv[1] if len(sys.argv) > 1 else "/tmp/crdata")
