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
