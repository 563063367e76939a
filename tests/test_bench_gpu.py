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
    # round 6: the touched-rows step over the ranks at S-TRAIN-XL (an eighth of it when eight ranks share one GPU): replicas bit-equal,
    # one all-gather of ceil(3 B / 8) (row id, row) slots per rank and step instead of the dense gradient all-reduce
    xl = eight["train_xl_dp"]
    assert xl["replicas_identical"] is True and xl["exchange_bytes_per_step"] == 8 * (-(-3 * 65536 // 8)) * (128 + 4) * 4
    assert xl["exchange_bytes_per_step"] * 5 < xl["dense_gradient_allreduce_bytes_it_replaces"]


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


def test_config5_full_size_fp16_properties_in_the_suite():
    """BASELINE configs[4] AT ITS OWN SIZE inside the GPU suite (it used to run only as bench.py's eval_f16 leg):
    131 072 users x 50 M fp16 items, d=256, rated CSR + 20 % cold bitmap, through the C ABI.  Size-independent properties --
    (1) two item shards (25 M each, global ids) + the canonical merge == the whole-table call, byte for byte;
    (2) a caller-named item-range cut (n_splits = 3) == the dispatcher's own route, byte for byte;
    and 16 users re-scored by a plain fp32 matmul over the same fp16 tables (the float-kernel reference of this tier: 1e-3
    relative + 1e-5), every returned id present in the reference's top-(k+8) or tied with its k-th score, no masked id returned."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from bench_legs.common import item_shard, rated_lists, xavier_
    from coldrec_amd import ops
    dev = torch.device("cuda:0")
    n_items, d, Bu, k = 50_000_000, 256, 131072, 20
    V = item_shard(n_items, d, 0, n_items, dev, torch.float16)
    U = xavier_(Bu, d, 17, dev, 1_000_000).to(torch.float16)
    rowptr, col = rated_lists(Bu, n_items, 50, seed=4)
    cold = np.where(np.random.default_rng(5).random(n_items) < 0.2)[0]
    bm = ops.make_bitmap(n_items, cold, dev)
    rp, rc = torch.from_numpy(rowptr).to(dev), torch.from_numpy(col).to(dev)
    assert ops.score_topk_route(Bu, n_items, d, k, half=True)["route"] == "fused-dma"
    s, i = ops.score_topk(U, None, V, k, rp, rc, bm)
    half = n_items // 2
    parts = [ops.score_topk(U, None, V[lo:hi], k, rp, rc, bm, item_base=lo) for lo, hi in ((0, half), (half, n_items))]
    ms, mi = ops.merge_topk(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]), k)
    assert torch.equal(mi, i) and torch.equal(ms.view(torch.int32), s.view(torch.int32)), "two shards + merge != whole table"
    del parts, ms, mi
    s3, i3 = ops.score_topk(U, None, V, k, rp, rc, bm, n_splits=3)
    assert torch.equal(i3, i) and torch.equal(s3.view(torch.int32), s.view(torch.int32)), "named cuts != the dispatcher's route"
    del s3, i3
    rng = np.random.default_rng(9)
    slots = np.sort(rng.choice(Bu, 16, replace=False))
    uu = U[torch.from_numpy(slots).to(dev)].float()
    best_s = torch.full((16, k + 8), -float("inf"), device=dev)
    best_i = torch.zeros((16, k + 8), dtype=torch.int64, device=dev)
    cold_t = torch.from_numpy(cold).to(dev)
    for lo in range(0, n_items, 2_500_000):
        hi = min(lo + 2_500_000, n_items)
        S = uu @ V[lo:hi].float().T
        S[:, cold_t[(cold_t >= lo) & (cold_t < hi)] - lo] = -1e9
        for q, sl in enumerate(slots.tolist()):
            ids = col[rowptr[sl]:rowptr[sl + 1]]
            ids = ids[(ids >= lo) & (ids < hi)] - lo
            if len(ids):
                S[q, torch.from_numpy(ids.astype(np.int64)).to(dev)] = -1e9
        cs, ci = torch.topk(S, k + 8, dim=1)
        m_s, m_i = torch.topk(torch.cat([best_s, cs], 1), k + 8, dim=1)
        best_i = torch.gather(torch.cat([best_i, ci + lo], 1), 1, m_i)
        best_s = m_s
        del S
    gs, gi = s[torch.from_numpy(slots).to(dev)].cpu().numpy(), i[torch.from_numpy(slots).to(dev)].cpu().numpy()
    rs, ri = best_s.cpu().numpy(), best_i.cpu().numpy()
    cold_set = set(cold[:0].tolist())
    for q, sl in enumerate(slots.tolist()):
        ref_of = dict(zip(ri[q].tolist(), rs[q].tolist()))
        rated_u = set(col[rowptr[sl]:rowptr[sl + 1]].tolist())
        for g, sg in zip(gi[q].tolist(), gs[q].tolist()):
            assert g not in rated_u and sg > -1e8, (sl, g, sg)                     # nothing masked is returned
            assert g in ref_of and abs(ref_of[g] - sg) <= 1e-3 * abs(sg) + 1e-5, (sl, g, sg, ref_of.get(g))
        assert np.all(np.abs(np.sort(gs[q])[::-1] - rs[q, :k]) <= 1e-3 * np.abs(rs[q, :k]) + 1e-5), sl
    assert not np.isin(gi, cold).any()
