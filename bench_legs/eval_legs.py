"""Evaluation legs of bench.py beside the headline: the trainers' validation shapes, mid-size catalogues, configs[4] in fp16,
the dense-block ranking kernel, S-EVAL through the trainer API, the DropoutNet generator."""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from .common import *  # noqa: F401,F403
from .common import _median_ms, _time_steps, _time_steps_each  # noqa: F401


def validation_eval_leg(dev):
    """The ranking the trainers run after every epoch, at the reference's own dataset sizes (configs[1] / [2]:
    MovieLens- and CiteULike-shaped validation: every user against the whole catalogue, rated lists + cold-item bitmap,
    k=20, fp32 d=128).  At these sizes the library scores into a dense block and ranks it with one wave per user
    (DESIGN.md 4.1); the fused selection of the headline kernel (forced single item range) is timed beside it."""
    from coldrec_amd import ops
    out = {}
    rng = np.random.default_rng(11)
    for name, n_users, n_items, mean_rated, d in (("movielens", 6040, 3706, 108, 128), ("citeulike", 5551, 16980, 23, 128),
                                                  ("movielens.d64", 6040, 3706, 108, 64), ("citeulike.d64", 5551, 16980, 23, 64)):
        U = xavier_(n_users, d, 21, dev, n_items)
        V = xavier_(n_items, d, 22, dev, n_users)
        rated = [np.unique(rng.integers(0, n_items, mean_rated)) for _ in range(n_users)]
        rp, rc = ops.rated_csr(rated, dev)
        bm = ops.make_bitmap(n_items, np.where(rng.random(n_items) < 0.2)[0], dev)
        ms = {}
        spread = {}
        for tag, ns, reps in (("library", 0, 50), ("fused_selection", 1, 5)):
            ms[tag], spread[tag] = _median_ms(lambda: ops.score_topk(U, None, V, 20, rp, rc, bm, n_splits=ns), reps)
        out[name] = {"users": n_users, "items": n_items, "d": d, "ms": ms["library"], "items_per_s": n_users * n_items / ms["library"] * 1e3,
                     "ms_min_max": [spread["library"]["min"], spread["library"]["max"]], "timed_calls": 50,
                     "ms_fused_selection": ms["fused_selection"]}
    return {"eval_validation": out}


def eval_d64_leg(dev, steps=3, warmup=1, n_items=10_000_000, d=64, Bu=131072, k=20):
    """the headline's shape at the REFERENCE'S DEFAULT WIDTH (main.py:97 --emb_size 64; BASELINE configs[0] is
    d=64): 131 072 users x 10 M items, fp32 d=64, rated CSR + 20 % cold bitmap, k=20.  Half the MFMA work per tile for the
    same selection, so every non-MFMA cycle weighs twice; 64 users of the last step re-ranked by the oracle, bit for bit."""
    from coldrec_amd import ops
    V = item_shard(n_items, d, 0, n_items, dev)
    n_blocks = 2
    U = xavier_(Bu * n_blocks, d, 17, dev, 1_000_000)
    rowptr, col = rated_lists(Bu * n_blocks, n_items, 50, seed=4)
    cold = np.where(np.random.default_rng(5).random(n_items) < 0.2)[0]
    bitmap = ops.make_bitmap(n_items, cold, dev)
    blocks = []
    for b in range(n_blocks):
        u0 = b * Bu
        blocks.append((torch.arange(u0, u0 + Bu, dtype=torch.int32, device=dev),
                       torch.from_numpy(rowptr[u0:u0 + Bu + 1] - rowptr[u0]).to(dev),
                       torch.from_numpy(col[rowptr[u0]:rowptr[u0 + Bu]]).to(dev)))
    events = HipEvents(steps)
    for w in range(warmup):
        ops.score_topk(U, *blocks[w % n_blocks][:1], V, k, *blocks[w % n_blocks][1:], bitmap)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s_ in range(steps):
        users, rp, rc = blocks[(warmup + s_) % n_blocks]
        out = ops.score_topk(U, users, V, k, rp, rc, bitmap, kernel_events=events.pairs[s_])
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / steps
    kern_ms = float(np.mean(events.elapsed_ms()))
    b_last = (warmup + steps - 1) % n_blocks
    u0 = b_last * Bu
    verified = verify_users("eval_d64", out[0].cpu().numpy(), out[1].cpu().numpy(), np.arange(u0, u0 + Bu, dtype=np.int64),
                            U.cpu().numpy(), V.cpu().numpy(), rowptr[u0:u0 + Bu + 1] - rowptr[u0],
                            col[rowptr[u0]:rowptr[u0 + Bu]], cold, k, n_check=64, seed=31)
    flops = 2.0 * d * Bu * n_items
    rt = route_of(Bu, n_items, d, k)
    leg = {"metric": "ranked items/sec (full-catalogue eval)", "value": Bu * n_items / sec, "unit": "items/s",
           "ms_per_step": sec * 1e3, "steps": steps, "dtype": "f32", "verified_users": verified,
           "config": {"workload": "the headline's shape at the reference's default width: %d users x %d items per step, d=%d, "
                                  "k=%d, fp32, rated CSR + 20%% cold-item bitmap" % (Bu, n_items, d, k)},
           "roofline": {"bound": "mfma", "kernel": rt["label"], "route": rt["route"], "seeded": rt["seeded"],
                        "achieved": flops / (kern_ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": flops / (kern_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, "kernel_ms": kern_ms,
                        "flops_per_launch": flops, "traffic": None,
                        "note": "v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate on the vector lanes: every VALU instruction of "
                                "the selection costs its full issue time beside it (tools/probes/dma_stream_probe_f32.hip: the "
                                "kernel's bare stream reaches 0.943 at d=64, 0.963 at d=128; DESIGN.md 4.1.3)"}}
    tr = measured_traffic(rt["profile_patterns"], rt["grid_threads"])
    if tr:
        leg["roofline"].update({"traffic": tr[0], "traffic_source": "committed profile " + tr[1]})
    del V, U, out
    return {"eval_d64": leg}


def midsize_eval_leg(dev):
    """Mid-size catalogues (65 K - 1 M items, fp32 d=128, masks, k=20): the shapes between the trainers' validation and
    the headline, where the fused selection's slow path, not MFMA, sets the time.  Whatever route the
    library picks (dense block + wave-per-user ranking, per-wave kernel, workgroup kernel); 2 users per shape are
    re-checked against the canonical oracle, bit for bit.  Round 6: two shapes at the reference's default width (d=64) and
    one at k=50 (lists too long for the LDS-DMA kernel's ring: the fallback route gets a number)."""
    from coldrec_amd import ops
    from oracle import oracle_np as orc
    out = {}
    for n_users, n_items, d, k in ((8192, 262144, 128, 20), (65536, 131072, 128, 20), (131072, 262144, 128, 20),
                                   (131072, 1048576, 128, 20),
                                   (4096, 10_000_000, 128, 20),        # the reference's own user block (--bs 4096) on the S-EVAL catalogue
                                   (131072, 1_250_000, 128, 20),       # one rank's item shard of the 8-GPU run
                                   (8192, 262144, 64, 20), (131072, 1_250_000, 64, 20),   # main.py:97 --emb_size 64
                                   (65536, 1048576, 128, 50),          # main.py:95 --topN is free-form
                                   (65536, 524288, 256, 20)):          # fp32 d=256: 32 users per wave, one wave per SIMD (never timed before)
        U = xavier_(n_users, d, 31, dev, n_items)
        V = item_shard(n_items, d, 0, n_items, dev)
        rowptr, col = rated_lists(n_users, n_items, 50, seed=4)
        cold = np.where(np.random.default_rng(5).random(n_items) < 0.2)[0]
        bm = ops.make_bitmap(n_items, cold, dev)
        rp, rc = torch.from_numpy(rowptr).to(dev), torch.from_numpy(col).to(dev)
        hold = {}

        def call():
            hold["res"] = ops.score_topk(U, None, V, k, rp, rc, bm)

        ms, sp = _median_ms(call, 5 if n_users * n_items < 4e10 else 3)
        res = hold["res"]
        pick = np.unique(np.concatenate([[0, n_users - 1], np.random.default_rng(9).integers(0, n_users, 14)])).astype(np.int64)
        sub_rp = np.concatenate([[0], np.cumsum([rowptr[u + 1] - rowptr[u] for u in pick])]).astype(np.int64)
        sub_col = np.concatenate([col[rowptr[u]:rowptr[u + 1]] for u in pick]).astype(np.int64)
        ws, wi = orc.score_topk(U[torch.from_numpy(pick).to(dev)].cpu().numpy(), np.arange(len(pick), dtype=np.int64),
                                V.cpu().numpy(), k, sub_rp, sub_col, orc.make_bitmap(n_items, cold))
        gs, gi = res[0][torch.from_numpy(pick).to(dev)].cpu().numpy(), res[1][torch.from_numpy(pick).to(dev)].cpu().numpy()
        if not (np.array_equal(gi, wi) and np.array_equal(gs.view(np.uint32), ws.view(np.uint32))):
            print(json.dumps({"error": "eval_midsize %d x %d differs from the oracle" % (n_users, n_items)}), flush=True)
            raise SystemExit(3)
        tf = 2.0 * d * n_users * n_items / (ms * 1e-3) / 1e12
        rt = route_of(n_users, n_items, d, k)
        tag = "%dx%d" % (n_users, n_items) + ("" if d == 128 else ".d%d" % d) + ("" if k == 20 else ".k%d" % k)
        out[tag] = {"ms": ms, "items_per_s": n_users * n_items / ms * 1e3, "d": d, "k": k,
                                             "frac_of_fp32_mfma_peak": tf / MFMA_F32_PEAK_TFLOPS, "verified_users": int(len(pick)),
                                             "route": {q: rt[q] for q in ("route", "seeded", "prefix_items", "n_splits", "kernel")}}
        del U, V, res
    return {"eval_midsize": out}


def eval_f16_leg(dev, steps=3, warmup=1, n_items=50_000_000, d=256, Bu=131072, k=20):
    """BASELINE.json configs[4] at its largest single-GPU shape: 131 072 users ranked against 50 M generated-style fp16
    item embeddings, d=256 (crh_score_topk_f16_ex: v_mfma_f32_32x32x16_f16, fp32 accumulate), masks as in the headline.
    Roofline vs the dense fp16 MFMA peak (2.5 PF).  Self-check: 4 users re-scored by a plain PyTorch fp32 matmul over
    the same fp16 tables (the float-kernel reference), scores within 1e-3 relative + 1e-5 and every returned id in the
    reference list or tied with its k-th score within that tolerance."""
    from coldrec_amd import ops
    V = item_shard(n_items, d, 0, n_items, dev, torch.float16)
    n_blocks = 2
    U = xavier_(Bu * n_blocks, d, 17, dev, 1_000_000).to(torch.float16)
    rowptr, col = rated_lists(Bu * n_blocks, n_items, 50, seed=4)
    cold = np.where(np.random.default_rng(5).random(n_items) < 0.2)[0]
    bitmap = ops.make_bitmap(n_items, cold, dev)
    blocks = []
    for b in range(n_blocks):
        u0 = b * Bu
        blocks.append((torch.arange(u0, u0 + Bu, dtype=torch.int32, device=dev),
                       torch.from_numpy(rowptr[u0:u0 + Bu + 1] - rowptr[u0]).to(dev),
                       torch.from_numpy(col[rowptr[u0]:rowptr[u0 + Bu]]).to(dev)))
    events = HipEvents(steps)
    for w in range(warmup):
        ops.score_topk(U, *blocks[w % n_blocks][:1], V, k, *blocks[w % n_blocks][1:], bitmap)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s_ in range(steps):
        users, rp, rc = blocks[(warmup + s_) % n_blocks]
        out = ops.score_topk(U, users, V, k, rp, rc, bitmap, kernel_events=events.pairs[s_])
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / steps
    kern_ms = float(np.mean(events.elapsed_ms()))
    flops = 2.0 * d * Bu * n_items
    # ---- self-check on 16 users of the last block against torch fp32 over the same fp16 inputs
    # (profiling aid: CRH_SCORE_ABLATE switches the selection off in the -DCRH_PROFILE build -- its results are invalid by
    # design and the leg then reports them as unverified)
    n_chk = 0 if os.environ.get("CRH_SCORE_ABLATE", "0") not in ("", "0") and os.environ.get("CRH_LIB") else 16
    b_last = (warmup + steps - 1) % n_blocks
    users, rp, rc = blocks[b_last]
    rng = np.random.default_rng(9)
    slots = np.sort(rng.choice(Bu, max(n_chk, 1), replace=False))[:n_chk]
    uu = U[users[torch.from_numpy(slots).to(dev)].long()].float()
    best_s = torch.full((n_chk, k + 8), -float("inf"), device=dev)
    best_i = torch.zeros((n_chk, k + 8), dtype=torch.int64, device=dev)
    cold_t = torch.from_numpy(cold).to(dev)
    rp_h, rc_h = rp.cpu().numpy(), rc.cpu().numpy()
    for lo in range(0, n_items if n_chk else 0, 2_500_000):
        hi = min(lo + 2_500_000, n_items)
        S = uu @ V[lo:hi].float().T
        cm = cold_t[(cold_t >= lo) & (cold_t < hi)] - lo
        S[:, cm] = -1e9
        for q, sl in enumerate(slots.tolist()):
            ids = rc_h[rp_h[sl]:rp_h[sl + 1]]
            ids = ids[(ids >= lo) & (ids < hi)] - lo
            if len(ids):
                S[q, torch.from_numpy(ids.astype(np.int64)).to(dev)] = -1e9
        cs, ci = torch.topk(S, k + 8, dim=1)
        ms, mi = torch.topk(torch.cat([best_s, cs], 1), k + 8, dim=1)
        best_i = torch.gather(torch.cat([best_i, ci + lo], 1), 1, mi)
        best_s = ms
        del S
    gs, gi = out[0][torch.from_numpy(slots).to(dev)].cpu().numpy(), out[1][torch.from_numpy(slots).to(dev)].cpu().numpy()
    rs, ri = best_s.cpu().numpy(), best_i.cpu().numpy()
    for q in range(n_chk):
        tol = 1e-3 * np.abs(rs[q, :k]) + 1e-5
        ref_of = dict(zip(ri[q].tolist(), rs[q].tolist()))
        ok = all((int(g) in ref_of and abs(ref_of[int(g)] - float(sg)) <= 1e-3 * abs(float(sg)) + 1e-5)
                 for g, sg in zip(gi[q], gs[q]))
        ok = ok and np.all(np.abs(np.sort(gs[q])[::-1] - rs[q, :k]) <= tol)
        if not ok:
            print(json.dumps({"error": "eval_f16: kernel result outside tolerance of the fp32 reference, slot %d" % slots[q],
                              "got": gi[q].tolist(), "ref": ri[q, :k].tolist()}), flush=True)
            raise SystemExit(3)
    rt16 = route_of(Bu, n_items, d, k, "f16")
    leg = {"metric": "ranked items/sec (full-catalogue eval)", "value": Bu * n_items / sec, "unit": "items/s",
           "ms_per_step": sec * 1e3, "steps": steps, "dtype": "f16", "verified_users": n_chk,
           "config": {"workload": "configs[4] shape on one GPU: %d users x %d items per step, d=%d, k=%d, fp16 tables / fp32 "
                                  "accumulate, rated CSR + 20%% cold-item bitmap" % (Bu, n_items, d, k)},
           "roofline": {"bound": "mfma", "kernel": rt16["label"], "route": rt16["route"], "achieved": flops / (kern_ms * 1e-3) / 1e12,
                        "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": flops / (kern_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, "kernel_ms": kern_ms,
                        "flops_per_launch": flops, "traffic": None,
                        "note": "peak is the nominal dense fp16 figure; the chip is power-limited under random fp16 operands "
                                "(a bare MFMA stream sustains 0.60-0.70 of it depending on the box): bare_loop_frac is that "
                                "stream measured in THIS run (tools/probes/mfma_energy_probe.hip), frac_of_bare = frac / it"}}
    bare = bare_mfma_loops()
    if bare:
        leg["roofline"]["bare_loop_frac"] = bare["register_fed"]
        leg["roofline"]["frac_of_bare"] = leg["roofline"]["frac"] / bare["register_fed"]
        leg["roofline"]["frac_of_bare_lds_fed"] = leg["roofline"]["frac"] / bare["lds_fed_same_shape"]
        leg["roofline"]["bare_loops"] = bare
    tr = measured_traffic(rt16["profile_patterns"], rt16["grid_threads"])
    if tr:
        leg["roofline"].update({"traffic": tr[0], "traffic_source": "committed profile " + tr[1]})
    # ---- one rank's launch of the 8-GPU run of configs[4]: the same user block against rows [0, I/8) of the same table
    # (global rated CSR and bitmap, ids outside the shard are skipped by the kernel exactly as on a rank)
    n_shard = n_items // 8
    ev_sh = HipEvents(steps)
    users, rp, rc = blocks[0]
    out_sh = ops.score_topk(U, users, V[:n_shard], k, rp, rc, bitmap)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s_ in range(steps):
        out_sh = ops.score_topk(U, users, V[:n_shard], k, rp, rc, bitmap, kernel_events=ev_sh.pairs[s_])
    torch.cuda.synchronize()
    sec_sh = (time.perf_counter() - t0) / steps
    kern_sh = float(np.mean(ev_sh.elapsed_ms()))
    tf_sh = 2.0 * d * Bu * n_shard / (kern_sh * 1e-3) / 1e12
    leg["shard_8gpu"] = {"users": Bu, "items": n_shard, "ms_per_step": sec_sh * 1e3, "kernel_ms": kern_sh,
                         "route": route_of(Bu, n_shard, d, k, "f16")["route"],
                         "items_per_s": Bu * n_shard / sec_sh, "frac_of_fp16_mfma_peak": tf_sh / MFMA_F16_PEAK_TFLOPS}
    leg["predicted_scaling_8gpu"] = {
        "value": 8.0 * (Bu * n_shard / sec_sh) / (Bu * n_items / sec),
        "note": "8 x rate(one rank's %d-item shard of configs[4]) / rate(the whole %d-item table), both on one GPU; the "
                "exchange (8 k bytes per user and rank, one all-gather) and the 160-candidate merge are < 1 %% of the step; "
                "no measured 8-GPU number exists" % (n_shard, n_items)}
    del V, U, out, out_sh
    return {"eval_f16": leg}


def mask_topk_leg(dev, n_users=4096, n_items=1_000_000, k=20, reps=5):
    """crh_mask_topk_f32 on a dense (4096 x 1 M) fp32 score block: the ranking path of every plugin whose batch_predict is
    not the stock matmul (model/VBPR.py:68-75, ALDI.py:149-160).  HBM-bound: 4 bytes per pair (one streaming read);
    with write-back (the reference mutates the block) the masked 16-byte vectors are stored too."""
    from coldrec_amd import ops
    from oracle import oracle_np as orc
    S = torch.randn(n_users, n_items, device=dev)
    rowptr, col = rated_lists(n_users, n_items, 50, seed=4)
    rp, rc = torch.from_numpy(rowptr).to(dev), torch.from_numpy(col).to(dev)
    cold = np.where(np.random.default_rng(5).random(n_items) < 0.2)[0]
    bm = ops.make_bitmap(n_items, cold, dev)
    ms = {}
    for wb in (False, True):
        hold = {}

        def call():
            hold["out"] = ops.mask_topk(S, k, rp, rc, bm, write_back=wb)

        ms[wb], _ = _median_ms(call, 3 * reps)
        out = hold["out"]
        if not wb:                                           # self-check before the block is mutated: 32 rows vs the oracle
            rows = sorted(set(int(x) for x in np.linspace(0, n_users - 1, 32)))
            for r in rows:
                ws, wi = orc.mask_topk(S[r:r + 1].cpu().numpy(), k, np.array([0, rowptr[r + 1] - rowptr[r]], np.int64),
                                       col[rowptr[r]:rowptr[r + 1]], orc.make_bitmap(n_items, cold))
                if not (np.array_equal(out[1][r].cpu().numpy(), wi[0]) and
                        np.array_equal(out[0][r].cpu().numpy().view(np.uint32), ws[0].view(np.uint32))):
                    print(json.dumps({"error": "mask_topk: row %d differs from the oracle" % r}), flush=True)
                    raise SystemExit(3)
    byts = n_users * n_items * 4.0
    leg = {"metric": "ranked items/sec (dense score block)", "value": n_users * n_items / (ms[False] * 1e-3), "unit": "items/s",
           "ms": ms[False], "ms_with_write_back": ms[True], "verified_users": len(rows),
           "config": {"workload": "crh_mask_topk_f32: %d x %d fp32 score block, k=%d, rated CSR + 20%% bitmap" % (n_users, n_items, k)},
           "roofline": {"bound": "hbm", "kernel": "mask_topk_kernel<1>", "achieved": byts / (ms[False] * 1e-3) / 1e9,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": byts / (ms[False] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "bytes_per_launch": byts, "traffic": None}}
    tr = measured_traffic("mask_topk_kernel<1", float(n_users * 64))
    if tr:
        leg["roofline"].update({"traffic": tr[0], "traffic_source": "committed profile " + tr[1],
                                "traffic_note": "FETCH_SIZE x2 + WRITE_SIZE per launch, mean over the launches of the "
                                                "profiled run (with and without write-back)"})
    del S
    return {"mask_topk": leg}


class ArrayTruth(dict):
    """A ground truth {user: {item: 1.0}} held as arrays (users, CSR of internal item ids): what
    ColdStartDataBuilder.truth_csr_cached hands the trainers for its own sets, without 1e6 nested Python dicts."""

    def __init__(self, users, rowptr, items):
        super().__init__()
        self.csr = (users, rowptr, items)
        self.n_pairs = int(rowptr[-1])

    def __len__(self):
        return len(self.csr[0])


class SyntheticEvalData:
    """The attributes of util/databuilder.ColdStartDataBuilder that BaseColdStartTrainer's evaluation reads (internal
    ids == original ids), over arrays generated for S-EVAL; nothing else of the builder is needed to rank and score."""

    def __init__(self, n_users, n_items, rated_rowptr, rated_col, cold_ids):
        self.user_num, self.item_num = n_users, n_items
        self.item = range(n_items)
        self.item_keys = np.arange(n_items, dtype=np.int64)
        self.rated_rowptr, self.rated_col = rated_rowptr, rated_col
        self.mapped_cold_item_idx = cold_ids
        self.mapped_warm_item_idx = np.zeros(0, np.int64)

    def truth_csr_cached(self, data_set):
        return data_set.csr

    def get_user_id_list(self, users):
        return np.asarray(users, np.int64)


def eval_e2e_leg(dev, n_users=1_000_000, n_items=10_000_000, d=128, truth_per_user=5, n_dict_users=100_000):
    """S-EVAL end to end THROUGH THE TRAINER API -- BaseColdStartTrainer._metrics (model/BaseRecommender.py:
    153-188 + util/evaluator.py:153-187 of the reference): 1e6 users x 1e7 items ranked (131 072-user blocks, 'warm' masks),
    the membership of the 2e7 predictions in a synthetic ground truth (~5 items per user, 2 of them planted among the
    user's actual top-20 for one user in 16) tested on the GPU, hits / precision / recall / NDCG at 10 and 20 on the host
    -- with a time split; and ``_evaluate`` (the {user: [(item, score)]} dict the plugin API returns) for 1e5 users.
    SURVEY.md 8(f)1's claim is that the consumer side must not dwarf the ranking: ``metrics_share_of_ranking``."""
    import argparse
    import types
    from coldrec_amd import ops
    from coldrec_amd.model.BaseRecommender import BaseColdStartTrainer
    from coldrec_amd.util.evaluator import ranking_metrics

    class EvalOnly(BaseColdStartTrainer):
        fused_eval = True

        def train(self): ...
        def predict(self, u): ...
        def batch_predict(self, users): ...
        def save(self): ...

    t0 = time.perf_counter()
    V = item_shard(n_items, d, 0, n_items, dev)
    U = xavier_(n_users, d, 17, dev, n_users)
    rowptr, col = rated_lists(n_users, n_items, 50, seed=4)
    cold = np.where(np.random.default_rng(5).random(n_items) < 0.2)[0]
    data = SyntheticEvalData(n_users, n_items, rowptr, col, cold)
    args = argparse.Namespace(dataset="s-eval", model="MF", epochs=0, layers=2, topN="10,20", bs=4096, emb_size=d, lr=1e-3,
                              reg=1e-4, runs=1, seed=2024, use_gpu=True, save_emb=False, gpu_id=0, cold_object="item",
                              backbone="MF", early_stop=0, eval_every=1)
    tr = EvalOnly(types.SimpleNamespace(args=args, data=data, device=dev))
    tr.user_emb, tr.item_emb = U, V
    # ground truth: truth_per_user uniform items per user; every 16th user gets two of its REAL top-20 items planted (found
    # by one ranking call over those users), so that the metrics are not all zero and their arithmetic is exercised
    rng = np.random.default_rng(21)
    gt = rng.integers(0, n_items, (n_users, truth_per_user), dtype=np.int64)
    planted = np.arange(0, n_users, 16)
    pu = torch.from_numpy(planted.astype(np.int32)).to(dev)
    p_rp = np.zeros(len(planted) + 1, np.int64)
    np.cumsum(rowptr[planted + 1] - rowptr[planted], out=p_rp[1:])
    p_rc = np.concatenate([col[rowptr[u]:rowptr[u + 1]] for u in planted]).astype(np.int32)
    _, top = ops.score_topk(U, pu, V, 20, torch.from_numpy(p_rp).to(dev), torch.from_numpy(p_rc).to(dev),
                            ops.make_bitmap(n_items, cold, dev))
    top = top.cpu().numpy().astype(np.int64)
    gt[planted, 0], gt[planted, 1] = top[:, 3], top[:, 14]
    gt_rowptr = np.arange(0, (n_users + 1) * truth_per_user, truth_per_user, dtype=np.int64)
    users = list(range(n_users))
    truth = ArrayTruth(users, gt_rowptr, gt.reshape(-1))
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t0
    tr.eval_timing = {}
    t0 = time.perf_counter()
    perf = tr._metrics(truth, "warm", [10, 20])
    torch.cuda.synchronize()
    t_total = time.perf_counter() - t0
    tm = dict(tr.eval_timing)
    s_all, i_all = tm.pop("last_topk")
    # ---- self-checks: (1) 32 users' lists against the CPU oracle, bit for bit; (2) the metrics of the first 131 072 users
    # recomputed on the host from the returned ids (numpy set membership, no GPU) == the trainer's GPU-membership route
    nchk = 131072
    sub = ArrayTruth(users[:nchk], gt_rowptr[:nchk + 1], gt[:nchk].reshape(-1))
    want = ranking_metrics(sub.csr[1], sub.csr[2], i_all[:nchk].cpu().numpy().astype(np.int64), [10, 20])
    hit = tr._membership({"gt_dense": None, "users": sub.csr[0], "gt_rowptr": sub.csr[1], "gt_items": sub.csr[2]}, i_all[:nchk])
    got = ranking_metrics(sub.csr[1], sub.csr[2], None, [10, 20], hit=hit)
    if got != want or perf[1][0] <= 0.0:
        print(json.dumps({"error": "eval_e2e: GPU membership metrics differ from the host recompute", "got": got, "want": want,
                          "all": perf}), flush=True)
        raise SystemExit(3)
    blk0 = slice(0, 131072)
    verified = verify_users("eval_e2e", s_all[blk0].cpu().numpy(), i_all[blk0].cpu().numpy(), np.arange(131072, dtype=np.int64),
                            U[blk0].cpu().numpy(), V.cpu().numpy(), rowptr[:131073], col, cold, 20, n_check=32, seed=77)
    del s_all, i_all
    # ---- the plugin-facing dict for 1e5 users (valid() / test() of the reference's API)
    sub_d = ArrayTruth(users[:n_dict_users], gt_rowptr[:n_dict_users + 1], gt[:n_dict_users].reshape(-1))
    tr.eval_timing = {}
    t0 = time.perf_counter()
    rec = tr._evaluate(sub_d, "warm")
    t_eval = time.perf_counter() - t0
    te = dict(tr.eval_timing)
    assert len(rec) == n_dict_users and len(rec[0]) == 20 and isinstance(rec[0][0][1], np.float32)
    rank_s = tm["rank_s"]
    consumer = tm["membership_s"] + tm["host_metrics_s"]
    leg = {"metric": "ranked items/sec (full-catalogue eval, ranking + metrics through the trainer API)",
           "value": n_users * n_items / (rank_s + consumer), "unit": "items/s",
           "config": {"workload": "S-EVAL through BaseColdStartTrainer._metrics: %d users x %d items, d=%d, k=20, 'warm' masks "
                                  "(rated CSR mean ~50 + 20%% cold-item bitmap), ground truth %d items per user, topN 10,20; "
                                  "ranking in %d-user blocks" % (n_users, n_items, d, truth_per_user, tr.EVAL_USER_BLOCK)},
           "seconds": {"total": t_total, "eval_cache_build_once": tm["cache_s"], "rank": rank_s, "membership_gpu": tm["membership_s"],
                       "host_metrics": tm["host_metrics_s"], "setup_untimed": t_setup},
           "metrics_share_of_ranking": consumer / rank_s,
           "cache_share_of_ranking": tm["cache_s"] / rank_s,
           "metrics": {"top10": perf[0], "top20": perf[1]}, "verified_users": verified, "metrics_recomputed_on_host_users": nchk,
           "evaluate_dict": {"users": n_dict_users, "seconds_total": t_eval, "rank_and_copy": te["evaluate_rank_s"],
                             "dict_build": te["evaluate_dict_s"], "dict_share_of_ranking": te["evaluate_dict_s"] / te["evaluate_rank_s"],
                             "note": "{user: [(item id, np.float32 score) x 20]} as model/BaseRecommender.py:185-187 returns it: "
                                     "2e6 Python tuples; run() itself never builds it (it scores the arrays)"},
           "roofline": {"bound": "mfma", "achieved": 2.0 * d * n_users * n_items / rank_s / 1e12, "peak": MFMA_F32_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": 2.0 * d * n_users * n_items / rank_s / 1e12 / MFMA_F32_PEAK_TFLOPS,
                        "traffic": None, "note": "ranking part only (the eight score_topk launches + their slicing)"}}
    del U, V, tr, rec
    return {"eval_e2e": leg}


def dropoutnet_generator(dev, n_items, d, content_dim=300, chunk=1_000_000, item_lo=0, block=250_000):
    """BASELINE.json configs[4], generator half (model/DropoutNet.py:126-135): every item goes through the item tower
    of DeepCF -- [warm embedding ; content] (d + content_dim) -> 200 -> 100 -> d, Linear + eval-mode BatchNorm + tanh --
    as stock PyTorch-ROCm modules (rocBLAS / hipBLASLt GEMMs), chunk by chunk, and lands as the fp16 item table the
    scoring kernel ranks.  Inputs are generated per chunk on the device (a 50 M x 300 content matrix is 60 GB)."""
    from coldrec_amd.model.DropoutNet import get_model
    torch.manual_seed(0)
    net = get_model(d, 0, content_dim, [200, 100], d).to(dev).eval()
    out = torch.empty((n_items, d), dtype=torch.float16, device=dev)
    g = torch.Generator(device=dev).manual_seed(11)
    warm = torch.randn((chunk, d), generator=g, device=dev) * 0.1
    content = torch.randn((chunk, content_dim), generator=g, device=dev)
    users_dummy = torch.zeros((1, d), device=dev)
    flops_item = 2.0 * ((d + content_dim) * 200 + 200 * 100 + 100 * d)

    def run():
        with torch.no_grad():
            for lo in range(0, n_items, chunk):
                hi = min(lo + chunk, n_items)
                _, v = net.encode(users_dummy, warm[: hi - lo], None, content[: hi - lo])
                out[lo:hi] = v.to(torch.float16)

    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run()
    torch.cuda.synchronize()
    sec = time.perf_counter() - t0
    # The table that is RANKED: the same tower over inputs drawn per GLOBAL block of `block` items (seed = block index), so a
    # rank's shard [item_lo, item_lo + n_items) holds exactly the rows the one-GPU run generates there -- the lists are then
    # independent of the number of ranks, as in the fp32 headline (untimed: the rate above is the tower's, on resident inputs)
    with torch.no_grad():
        for b in range(item_lo // block, (item_lo + n_items + block - 1) // block):
            gb = torch.Generator(device=dev).manual_seed(1000 + b)
            w_b = torch.randn((block, d), generator=gb, device=dev) * 0.1
            c_b = torch.randn((block, content_dim), generator=gb, device=dev)
            _, v = net.encode(users_dummy, w_b, None, c_b)
            g_lo, g_hi = max(b * block, item_lo), min((b + 1) * block, item_lo + n_items)
            out[g_lo - item_lo:g_hi - item_lo] = v[g_lo - b * block:g_hi - b * block].to(torch.float16)
    torch.cuda.synchronize()
    return out, {"metric": "items generated/sec (DropoutNet item tower)", "value": n_items / sec, "unit": "items/s",
                 "seconds": sec, "tflops": flops_item * n_items / sec / 1e12,
                 "config": {"workload": "DeepCF item tower %d -> 200 -> 100 -> %d (fp32 GEMMs via PyTorch-ROCm, eval-mode "
                                        "BatchNorm, tanh), %d items in chunks of %d, output cast to fp16"
                                        % (d + content_dim, d, n_items, chunk)}}
