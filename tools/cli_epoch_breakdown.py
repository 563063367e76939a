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


FIRST = {}


SYNC_ALL = os.environ.get("CLI_BREAKDOWN_SYNC", "0") == "1"      # synchronise after every wrapped call (first-call costs incl. the GPU's)


def wrap(obj, name, label=None, sync=False):
    sync = sync or SYNC_ALL
    f = getattr(obj, name)
    label = label or name

    def g(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        if sync:
            torch.cuda.synchronize()
        dt = time.perf_counter() - t
        T[label] += dt
        N[label] += 1
        FIRST.setdefault(label, []).append(dt)
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
    from coldrec_amd.model import MF as mfmod
    wrap(mfmod.MF, "_make_engine", "one-off: engine (tables to the GPU)", sync=True)
    wrap(sampler.EpochPrefetcher, "__init__", "one-off: prefetcher (pinned buffers)")
    wrap(train.EpochRunner, "__init__", "one-off: runner")
    from coldrec_amd import ops as opsmod
    wrap(opsmod, "build_plans_device", "  plans (enqueue)")
    wrap(opsmod, "mf_step_tables", "  step tables (enqueue)")
    args = cli.parse_args(["--dataset", "movielens", "--data_root", data_root, "--model", "MF", "--emb_size", "128",
                           "--epochs", "40", "--early_stop", "1000", "--save_emb", "false", "--result_dir", "/tmp/crres"])
    config = cli.Config(args, args.data_root)
    cli.set_seed(args.seed, True)
    t_build = time.perf_counter()
    model = cli.model_factory(config)
    t_build = time.perf_counter() - t_build
    # wall clock at the end of every epoch (fast_evaluation is the last thing an epoch does): the first epochs carry the
    # one-off costs -- kernel loading, allocations, the eager epoch, the hipGraph capture -- that "seconds per completed
    # training epoch" (main.py:203-205 divides the whole train() by the epochs) spreads over the run
    marks = []
    fe = base.BaseColdStartTrainer.fast_evaluation

    def fe_marked(self, *a, **k):
        r = fe(self, *a, **k)
        marks.append(time.perf_counter())
        return r
    base.BaseColdStartTrainer.fast_evaluation = fe_marked
    t = time.perf_counter()
    model.train()
    total = time.perf_counter() - t
    ep = model.epochs_ran
    per = [marks[0] - t] + [b - a for a, b in zip(marks, marks[1:])]
    steady = sorted(per[3:])[len(per[3:]) // 2]
    print(f"trainer construction {t_build * 1e3:.1f} ms; train(): {total / ep * 1e3:.2f} ms per epoch over {ep} epochs")
    print("  epoch 1 (engine, sampler, eager steps) %.1f ms, epoch 2 (capture) %.1f ms, epoch 3 %.1f ms, steady state (median) "
          "%.2f ms, after the last epoch %.1f ms" % (per[0] * 1e3, per[1] * 1e3, per[2] * 1e3, steady * 1e3, (t + total - marks[-1]) * 1e3))
    print("  one-off costs = %.1f ms = %.2f ms per epoch of this %d-epoch run" % ((total - steady * ep) * 1e3, (total - steady * ep) / ep * 1e3, ep))
    print("  first calls (ms):", {k.strip(): [round(x * 1e3, 1) for x in v[:3]] for k, v in FIRST.items()})
    attributed = sum(v for k, v in T.items() if not k.startswith(" "))
    for k, v in T.items():
        print(f"  {k:32s} {v / ep * 1e3:7.3f} ms per epoch ({N[k]} calls)")
    print(f"  {'losses to the host (GPU wait)':32s} {T.get('wait', 0.0) / ep * 1e3:7.3f} ms per epoch")
    print(f"  unattributed: {(total - attributed - T.get('wait', 0.0)) / ep * 1e3:.3f} ms per epoch (prints, Python between the calls, one-off costs)")


if __name__ == "__main__":
    run(sys.argv[1] if len(sys.argv) > 1 else "/tmp/crdata")
