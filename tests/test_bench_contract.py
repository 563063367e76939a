"""bench.py's host-side contract pieces that need no GPU: the committed-profile traffic lookup (kernel instantiation +
grid must match, newest record wins) and the argument defaults the driver relies on."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_traffic_lookup_matches_instantiation_and_grid():
    import bench
    head = bench.measured_traffic("score_topk_wg_kernel<float, 128, 2, 8,", 131072.0)
    assert head is not None and head[1].startswith("r0") and 3.5e10 < head[0] < 6e10       # 8 XCDs x the 5.12 GB table
    assert bench.measured_traffic("score_topk_wg_kernel<float, 128, 2, 8,", 12345.0) is None   # another grid: no record
    f16 = bench.measured_traffic(("score_topk_wg_kernel<_Float16, 256", "score_topk_wg_kernelIDF16_Li256E"), 131072.0)
    assert f16 is not None and 1.5e11 < f16[0] < 3e11                                          # 8 x the 25.6 GB table
    mk = bench.measured_traffic("mask_topk_kernel<1", 4096 * 64.0)
    assert mk is not None and 1.6e10 <= mk[0] < 3.5e10                                         # the 16.4 GB block (+ write-back share)
    assert bench.measured_traffic("no_such_kernel", None) is None


def test_default_command_line_and_legs():
    import bench
    src = open(os.path.join(ROOT, "bench.py")).read()
    for leg in ("eval_f16", "mask_topk", "train_xl", "train_xl_lightgcn", "train", "eval_validation", "eval_midsize", "eval_e2e",
                "torch_rocm"):
        assert leg in src
    rec = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_f.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in rec, key
    assert rec["vs_baseline"] is None and rec["roofline"]["bound"] == "mfma" and rec["cpu_baseline"]["kind"] == "port"
    assert set(rec["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert abs(rec["roofline"]["frac"] - rec["roofline"]["achieved"] / rec["roofline"]["peak"]) < 1e-9
    assert bench.MFMA_F32_PEAK_TFLOPS == 157.3 and bench.HBM_PEAK_GBS == 8000.0
    xl = rec["train_xl_lightgcn"]
    assert xl["roofline"]["bound"] == "hbm" and xl["spmm"]["gather_GBps"] > 0 and xl["spmm"]["traffic"] is not None
    assert {"4096x10000000", "131072x1250000"} <= set(rec["eval_midsize"]) and rec["verified_users"] >= 64
    assert rec["eval_f16"]["verified_users"] >= 16 and rec["mask_topk"]["verified_users"] >= 32
    assert all(v["verified_users"] >= 16 for v in rec["eval_midsize"].values())
    assert 6.0 < rec["predicted_scaling_8gpu"]["value"] <= 8.0
    # round 4: every training leg is the MEDIAN of 60 individually timed hipGraph epochs, with its spread
    for leg in ("train_mf", "train_mf_sgd", "train_lightgcn"):
        sp = rec[leg]["ms_per_step_spread"]
        assert rec[leg]["timed_epochs"] == 60 and abs(rec[leg]["ms_per_step"] - sp["median"]) < 1e-12
        assert sp["min"] <= sp["median"] <= sp["max"] and sp["stalled_epoch_seen"] == (sp["max"] > 2.0 * sp["median"])
    # (this record's LightGCN leg did catch a stalled epoch -- max 0.754 ms against a median of 0.1143: the flag is set and the
    # median, which is what the line reports, does not move; round 3's five-epoch block average would have read 0.54)
    assert rec["train_lightgcn"]["ms_per_step_spread"]["p90"] <= 0.12
    assert rec["train_lightgcn"]["ms_per_step"] <= 0.12                                  # the bar of VERDICT r2 1b / r3 2
    # ... the legs VERDICT r3 asked for: S-EVAL through the trainer API with a time split (metrics < 5 % of the ranking), the
    # configs[4] shard shape with its own scaling prediction, and a compact summary as the LAST key of the line
    e2e = rec["eval_e2e"]
    assert e2e["metrics_share_of_ranking"] < 0.05 and e2e["verified_users"] >= 32 and e2e["metrics"]["top20"][0] > 0
    assert set(e2e["seconds"]) >= {"rank", "membership_gpu", "host_metrics", "eval_cache_build_once", "total"}
    assert rec["eval_f16"]["shard_8gpu"]["items"] == 6_250_000 and 6.0 < rec["eval_f16"]["predicted_scaling_8gpu"]["value"] <= 8.0
    assert list(rec)[-1] == "legs_summary" and {"headline", "eval_f16", "mask_topk", "train_lightgcn"} <= set(rec["legs_summary"])
    assert isinstance(rec["result_crc32"], int)
    assert rec["wall_s"]["total"] < 360 and set(rec["wall_s"]) >= {"eval_f16", "train", "eval_e2e", "total"}   # "finishes within minutes"
    # the XL SpMM's traffic comes from a profile of its own instantiation
    tr = bench.measured_traffic("spmm_csr_kernel<32>", None)
    assert tr is not None and tr[1].startswith("r0") and 1.5e11 < tr[0] < 3e11


def test_gpus_n_without_a_launcher_starts_its_own_ranks():
    """``python bench.py --gpus N`` with WORLD_SIZE unset launches ``torch.distributed.run`` itself, as
    a CHILD process, before anything touches the GPU, and relays the child's exit code.  Here (no GPU) the dry hook shows
    the command; the real launch reaches both ranks' "needs an MI355X" exit and the parent returns non-zero."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    dry = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1"],
                         env=dict(env, CRH_BENCH_DRY_LAUNCH="1"), capture_output=True, text=True, timeout=300)
    assert dry.returncode == 0, dry.stderr[-2000:]
    cmd = json.loads(dry.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-7:] == [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1"]
    # the source decides before the first torch.cuda call of main()
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def main():"):]
    assert body.index("self_launch(") < body.index("torch.cuda.")
    assert "os.exec" not in src
    import torch
    if not torch.cuda.is_available():
        real = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                              env=env, capture_output=True, text=True, timeout=600)
        assert real.returncode != 0                              # relayed from the launcher (both ranks refuse a CPU box)
        assert "needs an MI355X" in real.stderr + real.stdout    # ... i.e. the ranks DID start and parse their flags
