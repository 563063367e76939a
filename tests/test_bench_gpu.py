"""bench.py as the driver runs it, on the GPU box: the N > 1 path must start by itself (``python bench.py --gpus N`` with no
launcher around it), give ONE JSON line, and rank the same lists as N = 1 (item-row shards + one
all-gather + canonical merge are independent of N).  One GPU here, so the ranks share it and exchange over gloo
(CRH_BENCH_BACKEND=gloo); the driver's runs use RCCL with one GPU per rank."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--items", "200003", "--users", "4096", "--users-per-step", "2048", "--steps", "2", "--warmup", "1",
         "--no-cpu-baseline", "--no-train", "--legs", "none"]


def _bench(flags, **env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + flags, env=env, capture_output=True, text=True,
                         timeout=1200)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.returncode, out.stdout[-2000:], out.stderr[-3000:])
    return json.loads(lines[0])


def test_gpus_2_self_launched_equals_gpus_1():
    one = _bench(["--gpus", "1"] + SMALL)
    two = _bench(["--gpus", "2"] + SMALL, CRH_BENCH_BACKEND="gloo")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert one["verified_users"] == 64 and two["verified_users"] == 64        # both re-checked by the oracle, bit for bit
    assert one["result_crc32"] == two["result_crc32"]                          # the last step's (scores, ids), byte for byte
    assert "all_gather" in two["config"]["parallelism"] and two["scaling"] == "strong"
    assert list(two)[-1] == "legs_summary" and "headline" in two["legs_summary"]


def test_gpus_4_user_shards_self_launched_equals_gpus_1():
    one = _bench(["--gpus", "1"] + SMALL)
    four = _bench(["--gpus", "4", "--shard", "users"] + SMALL, CRH_BENCH_BACKEND="gloo")
    assert four["n_gpus"] == 4 and one["result_crc32"] == four["result_crc32"]


def test_gpus_8_full_headline_size_self_launched_equals_gpus_1():
    """The driver's N = 8 command shape at the FULL headline size (1 M-row user table x 10 M items, 131 072 users per step): eight
    self-launched ranks with 1.25 M-item shards on the one GPU, one all-gather of 8 x 131 072 x 40 words, the canonical merge,
    rank 0's oracle check against the whole table rebuilt from the shards' seeds, and the data-parallel train leg -- the last
    step's (scores, ids) are byte for byte those of the one-rank run (tools/n8_full_size_one_gpu.sh is the same as a script)."""
    full = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--legs", "none"]
    one = _bench(["--gpus", "1"] + full)
    eight = _bench(["--gpus", "8"] + full, CRH_BENCH_BACKEND="gloo")
    assert eight["n_gpus"] == 8 and eight["config"]["items"] == 10_000_000 and eight["config"]["users_per_step"] == 131072
    assert one["verified_users"] == 64 and eight["verified_users"] == 64
    assert one["result_crc32"] == eight["result_crc32"]
    assert eight["train_mf_dp"]["replicas_identical"] is True


def test_config5_generator_fp16_three_uneven_shards_equal_one_rank():
    """BASELINE configs[4]'s command shape (--dtype f16 --generator) at a small size: the DropoutNet item tower generates each
    rank's rows of the fp16 table from inputs drawn per GLOBAL block of 250 000 items, so three ranks whose shard bounds fall
    inside blocks (700 003 items) rank the same table as one rank: the last step's lists are equal byte for byte."""
    small = ["--dtype", "f16", "--generator", "--dim", "64", "--items", "700003", "--users", "4096", "--users-per-step", "2048",
             "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-train", "--legs", "none"]
    one = _bench(["--gpus", "1"] + small)
    three = _bench(["--gpus", "3"] + small, CRH_BENCH_BACKEND="gloo")
    assert three["n_gpus"] == 3 and three["dtype"] == "f16" and "dropoutnet_generator" in three
    assert one["result_crc32"] == three["result_crc32"]
