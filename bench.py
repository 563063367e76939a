#!/usr/bin/env python3
"""Headline benchmark: ranked items/s of the fused full-catalogue evaluation (BASELINE.json
config 4: 1M users x 10M items, d=128, k=20) on N MI355X.

One "step" = one block of ``--users-per-step`` users scored and ranked against the WHOLE
catalogue: crh_score_topk_f32 over the rank's item shard (rows [r*I/N, (r+1)*I/N)), then for
N > 1 an RCCL all-gather of the per-shard top-k and the canonical merge.
Total work per step is fixed as N grows ("scaling": "strong", the north_star's ">= 6x further
at 8 GPUs").  Inputs are resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 4 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 4 --warmup 1

The default N=1 run also carries every other leg the design claims, in the same JSON line (each with its own
``roofline``): ``eval_f16`` (configs[4] shape, fp16 MFMA), ``mask_topk`` (dense-block ranking, HBM-bound), ``train_xl``
(S-TRAIN-XL dense Adam, HBM-bound), ``train_mf`` / ``train_mf_sgd`` / ``train_lightgcn`` (configs[1] / [2], with
``value_end_to_end`` = triples per wall second over whole epochs WITH the sampler running) and the per-epoch validation
ranking; after the timed region 8 users of the last headline step are re-ranked by the CPU oracle and must match bit
for bit (``verified_users``; a mismatch ends the run with a non-zero exit code).

Rank 0 prints ONE JSON line.  ``roofline`` is for the dominant kernel (score_topk_kernel):
achieved = 2*d flop per (user,item) pair x pairs per launch / average kernel time measured with
HIP events recorded around that kernel on its stream.  ``cpu_baseline`` is the reference path
restated with the same library calls (oracle/ref_port.py: torch.matmul -> masks -> torch.topk),
timed on the host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: fp32-input MFMA, dense
MFMA_F16_PEAK_TFLOPS = 2500.0  # same guide: BF16/F16 MFMA ~2.5 PF dense (not the 2:1-sparsity headline)
CHUNK_ROWS = 1_250_000         # item table is generated in chunks so shards agree for N = 1,2,4,8


class HipEvents:
    """Raw hipEvent_t pairs (libamdhip64 via ctypes) recorded by the C ABI around the kernel."""

    def __init__(self, n):
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.pairs = []
        for _ in range(n):
            a, b = ctypes.c_void_p(), ctypes.c_void_p()
            assert self.hip.hipEventCreate(ctypes.byref(a)) == 0
            assert self.hip.hipEventCreate(ctypes.byref(b)) == 0
            self.pairs.append((a, b))

    def elapsed_ms(self):
        out = []
        for a, b in self.pairs:
            ms = ctypes.c_float()
            assert self.hip.hipEventElapsedTime(ctypes.byref(ms), a, b) == 0
            out.append(ms.value)
        return out


def xavier_(rows, d, seed, device, fan_rows):
    g = torch.Generator(device=device).manual_seed(seed)
    a = (6.0 / (fan_rows + d)) ** 0.5
    return (torch.rand((rows, d), generator=g, device=device, dtype=torch.float32) * 2 - 1) * a


def item_shard(n_items, d, lo, hi, device, dtype=torch.float32):
    """Rows [lo, hi) of the synthetic item table U(-a, a) (seed 3 + chunk), xavier-like."""
    out = torch.empty((hi - lo, d), dtype=dtype, device=device)
    for c in range(lo // CHUNK_ROWS, (hi + CHUNK_ROWS - 1) // CHUNK_ROWS):
        c_lo, c_hi = c * CHUNK_ROWS, min((c + 1) * CHUNK_ROWS, n_items)
        chunk = xavier_(c_hi - c_lo, d, 3000 + c, device, n_items)
        a, b = max(lo, c_lo), min(hi, c_hi)
        out[a - lo: b - lo] = chunk[a - c_lo: b - c_lo].to(dtype)
    return out


def rated_lists(n_users, n_items, mean_len, seed):
    """Per-user training items (SURVEY.md 8(d) S-EVAL): Zipf-truncated list lengths with mean ~mean_len
    (zipf(2.5) * 0.54 mean_len, capped at 40*mean_len), uniform item ids, ascending within a user."""
    rng = np.random.default_rng(seed)
    lens = np.minimum(rng.zipf(2.5, n_users) * max(int(round(mean_len * 0.54)), 1), 40 * mean_len).astype(np.int64)
    rowptr = np.zeros(n_users + 1, np.int64)
    np.cumsum(lens, out=rowptr[1:])
    col = rng.integers(0, n_items, int(rowptr[-1]), dtype=np.int64)
    key = np.repeat(np.arange(n_users, dtype=np.int64), lens) << 32 | col
    key.sort()
    return rowptr, (key & 0xFFFFFFFF).astype(np.int32)


def cpu_baseline(U_cpu, V_cpu, rowptr, col, cold_ids, k, block=256, budget_s=75.0):
    """oracle/ref_port.eval_block (the reference's own library calls: torch.matmul -> masks -> torch.topk) on the host
    cores: user blocks of ``block`` (the reference's 4096 would need a 164 GB score block at 10 M items, SURVEY.md
    8(d)) against the WHOLE item table, until every sampled user is ranked or the time budget is spent.
    Returns (items/s, users ranked, seconds)."""
    from oracle import ref_port
    torch.set_num_threads(os.cpu_count())
    cand = torch.from_numpy(cold_ids[cold_ids < V_cpu.shape[0]].astype(np.int64))

    def rated_of(lo, hi):
        out = []
        for r in range(lo, hi):
            ids = col[rowptr[r]:rowptr[r + 1]]
            ids = ids[ids < V_cpu.shape[0]]
            out.append(torch.from_numpy(ids.astype(np.int64)) if len(ids) else None)
        return out

    ref_port.eval_block(U_cpu[:8], V_cpu, torch.arange(8), rated_of(0, 8), cand, k)   # touch pages / warm MKL
    done, t0 = 0, time.perf_counter()
    while done < U_cpu.shape[0]:
        hi = min(done + block, U_cpu.shape[0])
        ref_port.eval_block(U_cpu, V_cpu, torch.arange(done, hi), rated_of(done, hi), cand, k)
        done = hi
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return done * V_cpu.shape[0] / dt, done, dt


def verify_users(tag, got_s, got_i, users_rows, U_cpu, V_cpu, rowptr_blk, col_blk, cold_ids, k, n_check=8, seed=123):
    """Self-check of a timed step: ``n_check`` users of the block re-ranked by the CPU oracle (oracle/topk_oracle.c,
    the canonical fma chain and order) must equal what the kernel returned, scores and indices, bit for bit.
    got_s / got_i: (block, k) host arrays; users_rows: table rows of the block's slots; rowptr_blk / col_blk: the
    block's rated CSR.  Raises SystemExit(3) on a mismatch -- a fast wrong kernel must not produce a number."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle_np as orc
    rng = np.random.default_rng(seed)
    slots = np.sort(rng.choice(got_i.shape[0], size=min(n_check, got_i.shape[0]), replace=False))
    bm = orc.make_bitmap(V_cpu.shape[0], cold_ids) if cold_ids is not None and len(cold_ids) else None

    def one(sl):
        lo, hi = int(rowptr_blk[sl]), int(rowptr_blk[sl + 1])
        rp = np.array([0, hi - lo], np.int64)
        return orc.score_topk(U_cpu[users_rows[sl]:users_rows[sl] + 1], None, V_cpu, k, rp, col_blk[lo:hi], bm)

    with ThreadPoolExecutor(max_workers=min(len(slots), os.cpu_count() or 1)) as ex:       # ctypes releases the GIL
        want = list(ex.map(one, slots.tolist()))
    for sl, (ws, wi) in zip(slots.tolist(), want):
        if not (np.array_equal(got_i[sl], wi[0]) and np.array_equal(got_s[sl].view(np.uint32), ws[0].view(np.uint32))):
            print(json.dumps({"error": "%s: kernel result differs from the oracle for block slot %d" % (tag, sl),
                              "got_idx": got_i[sl].tolist(), "want_idx": wi[0].tolist()}), flush=True)
            raise SystemExit(3)
    return len(slots)


HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def measured_traffic(kernel_prefix, grid_threads, prefer=None):
    """HBM-side bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC summary
    (profiles/*_pmc.json, written by tools/profile_round.sh + tools/prof_summary.py in separate --pmc passes;
    FETCH_SIZE x 1024 x 2 as MI355X_MICROARCH.md prescribes for 16-B/lane streams on gfx950, + WRITE_SIZE x
    1024).  Only a record of the SAME kernel instantiation and grid counts; otherwise None."""
    import glob
    best = None
    # newest record by name; among a round's passes the one taken for this leg (``prefer``) wins
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")),
                    key=lambda x: (os.path.basename(x).split("_")[0], bool(prefer) and prefer in os.path.basename(x), os.path.basename(x))):
        try:
            rec = json.load(open(f))
        except (OSError, ValueError):
            continue
        prefixes = kernel_prefix if isinstance(kernel_prefix, (tuple, list)) else (kernel_prefix,)
        for name, v in rec.items():
            if any(p in name for p in prefixes) and grid_threads in (None, v.get("_Grid_Size")) and \
                    "hbm_read_bytes_corrected" in v:
                best = (v["hbm_read_bytes_corrected"] + v.get("hbm_write_bytes", 0.0), os.path.basename(f))
    return best


def route_of(n_users, n_items, d, k, dtype="f32", masks=True, n_splits=0):
    """The scoring route of a block of this shape AS THE LIBRARY REPORTS IT (crh_score_topk_route: the dispatcher's own
    predicates, no Python re-implementation), with the kernel's label, its grid in threads and the name patterns under which
    a profile record of that instantiation is filed (rocprofv3 prints demangled or mangled names, build by build)."""
    from coldrec_amd import ops
    r = ops.score_topk_route(n_users, n_items, d, k, half=(dtype == "f16"), has_bitmap=masks, n_splits=n_splits)
    upw, waves = {"fused-dma": (128, 4), "fused-wg": (64, 8)}.get(r["route"], (None, 1))
    if upw is None:            # per-wave kernel: users per wave by row width (score_topk.hip users_per_wave)
        upw = (32 if d >= 256 else 64 if d >= 128 else 128) if dtype == "f32" else (64 if d >= 256 else 128)
    groups = -(-n_users // upw)
    r["grid_threads"] = float(64 * waves * -(-groups // waves) * max(1, r["n_splits"])) if r["route"] != "dense" else None
    r["label"] = "%s<%s,%d>%s" % (r["kernel"], dtype, d, " + mask_topk_kernel" if r["route"] == "dense" else "")
    ctype, mangled = ("float", "If") if dtype == "f32" else ("_Float16", "IDF16_")
    r["profile_patterns"] = ("%s<%s, %d" % (r["kernel"], ctype, d), "%s%sLi%dE" % (r["kernel"], mangled, d))
    return r


def _time_steps_each(fn, n_steps, warm):
    """Every step between its own pair of events (the host does not wait in between): (median seconds, spread dict).  For legs
    whose whole timed region is tens of milliseconds, where one hiccup would own a block average."""
    for s in range(warm):
        fn(s)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_steps + 1)]
    ev[0].record()
    for s in range(n_steps):
        fn(warm + s)
        ev[s + 1].record()
    torch.cuda.synchronize()
    ms = np.array([ev[s].elapsed_time(ev[s + 1]) for s in range(n_steps)])
    med = float(np.median(ms))
    return med * 1e-3, {"median": med, "min": float(ms.min()), "max": float(ms.max()), "mean": float(ms.mean()),
                        "stalled_step_seen": bool(ms.max() > 2 * med), "how": "each of %d steps timed event to event" % n_steps}


def _median_ms(fn, reps, warm=2):
    """median milliseconds of ``reps`` calls, each between its own pair of events (+ min / max): the short secondary legs"""
    sec, sp = _time_steps_each(lambda s: fn(), reps, warm)
    return sp["median"], sp


def _time_steps(fn, n_steps, warm):
    for s in range(warm):
        fn(s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for s in range(n_steps):
        fn(warm + s)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n_steps


def validation_eval_leg(dev):
    """The ranking the trainers run after every epoch, at the reference's own dataset sizes (configs[1] / [2]:
    MovieLens- and CiteULike-shaped validation: every user against the whole catalogue, rated lists + cold-item bitmap,
    k=20, fp32 d=128).  At these sizes the library scores into a dense block and ranks it with one wave per user
    (DESIGN.md 4.1); the fused selection of the headline kernel (forced single item range) is timed beside it."""
    from coldrec_amd import ops
    out = {}
    rng = np.random.default_rng(11)
    for name, n_users, n_items, mean_rated in (("movielens", 6040, 3706, 108), ("citeulike", 5551, 16980, 23)):
        U = xavier_(n_users, 128, 21, dev, n_items)
        V = xavier_(n_items, 128, 22, dev, n_users)
        rated = [np.unique(rng.integers(0, n_items, mean_rated)) for _ in range(n_users)]
        rp, rc = ops.rated_csr(rated, dev)
        bm = ops.make_bitmap(n_items, np.where(rng.random(n_items) < 0.2)[0], dev)
        ms = {}
        spread = {}
        for tag, ns, reps in (("library", 0, 50), ("fused_selection", 1, 5)):
            ms[tag], spread[tag] = _median_ms(lambda: ops.score_topk(U, None, V, 20, rp, rc, bm, n_splits=ns), reps)
        out[name] = {"users": n_users, "items": n_items, "ms": ms["library"], "items_per_s": n_users * n_items / ms["library"] * 1e3,
                     "ms_min_max": [spread["library"]["min"], spread["library"]["max"]], "timed_calls": 50,
                     "ms_fused_selection": ms["fused_selection"]}
    return {"eval_validation": out}


def midsize_eval_leg(dev):
    """Mid-size catalogues (65 K - 1 M items, fp32 d=128, masks, k=20): the shapes between the trainers' validation and
    the headline, where the fused selection's slow path, not MFMA, sets the time.  Whatever route the
    library picks (dense block + wave-per-user ranking, per-wave kernel, workgroup kernel); 2 users per shape are
    re-checked against the canonical oracle, bit for bit."""
    from coldrec_amd import ops
    from oracle import oracle_np as orc
    out = {}
    for n_users, n_items in ((8192, 262144), (65536, 131072), (131072, 262144), (131072, 1048576),
                             (4096, 10_000_000),        # the reference's own user block (--bs 4096) on the S-EVAL catalogue
                             (131072, 1_250_000)):      # one rank's item shard of the 8-GPU run

        U = xavier_(n_users, 128, 31, dev, n_items)
        V = item_shard(n_items, 128, 0, n_items, dev)
        rowptr, col = rated_lists(n_users, n_items, 50, seed=4)
        cold = np.where(np.random.default_rng(5).random(n_items) < 0.2)[0]
        bm = ops.make_bitmap(n_items, cold, dev)
        rp, rc = torch.from_numpy(rowptr).to(dev), torch.from_numpy(col).to(dev)
        hold = {}

        def call():
            hold["res"] = ops.score_topk(U, None, V, 20, rp, rc, bm)

        ms, sp = _median_ms(call, 5 if n_users * n_items < 4e10 else 3)
        res = hold["res"]
        pick = np.unique(np.concatenate([[0, n_users - 1], np.random.default_rng(9).integers(0, n_users, 14)])).astype(np.int64)
        sub_rp = np.concatenate([[0], np.cumsum([rowptr[u + 1] - rowptr[u] for u in pick])]).astype(np.int64)
        sub_col = np.concatenate([col[rowptr[u]:rowptr[u + 1]] for u in pick]).astype(np.int64)
        ws, wi = orc.score_topk(U[torch.from_numpy(pick).to(dev)].cpu().numpy(), np.arange(len(pick), dtype=np.int64),
                                V.cpu().numpy(), 20, sub_rp, sub_col, orc.make_bitmap(n_items, cold))
        gs, gi = res[0][torch.from_numpy(pick).to(dev)].cpu().numpy(), res[1][torch.from_numpy(pick).to(dev)].cpu().numpy()
        if not (np.array_equal(gi, wi) and np.array_equal(gs.view(np.uint32), ws.view(np.uint32))):
            print(json.dumps({"error": "eval_midsize %d x %d differs from the oracle" % (n_users, n_items)}), flush=True)
            raise SystemExit(3)
        tf = 2.0 * 128 * n_users * n_items / (ms * 1e-3) / 1e12
        rt = route_of(n_users, n_items, 128, 20)
        out["%dx%d" % (n_users, n_items)] = {"ms": ms, "items_per_s": n_users * n_items / ms * 1e3,
                                             "frac_of_fp32_mfma_peak": tf / MFMA_F32_PEAK_TFLOPS, "verified_users": int(len(pick)),
                                             "route": {q: rt[q] for q in ("route", "seeded", "prefix_items", "n_splits", "kernel")}}
        del U, V, res
    return {"eval_midsize": out}


def torch_rocm_leg(dev):
    """The reference's own library calls (model/MF.py:12-29, model/LightGCN.py:14-29,86-96,
    model/BaseRecommender.py:172-183) on the SAME GPU through stock PyTorch-ROCm -- what a ColdRec checkout does with
    --use_gpu -- at the shapes of the train legs and on a 1024-user block of the headline.  Context only (plain torch,
    no oracle): not a target and not the CPU baseline."""
    from coldrec_amd.util.databuilder import bipartite_norm_adj_csr

    def bpr_loss(u, p, n):                      # util/utils.py:25-29
        return torch.mean(-torch.log(10e-6 + torch.sigmoid((u * p).sum(1) - (u * n).sum(1))))

    def l2_reg(reg, *embs):                     # util/utils.py:44-48
        loss = 0
        for e in embs:
            loss = loss + torch.norm(e, p=2) / e.shape[0]
        return loss * reg

    def train(n_u, n_i, n_pairs, d, B, layers, steps=60):
        rng = np.random.default_rng(1)
        U = torch.nn.Parameter(torch.nn.init.xavier_uniform_(torch.empty(n_u, d, device=dev)))
        V = torch.nn.Parameter(torch.nn.init.xavier_uniform_(torch.empty(n_i, d, device=dev)))
        opt = torch.optim.Adam([U, V], lr=1e-3)
        adj = None
        if layers:
            pairs = np.unique(np.stack([rng.integers(0, n_u, n_pairs), rng.integers(0, n_i, n_pairs)], 1), axis=0)
            rowptr, col, val = bipartite_norm_adj_csr(pairs[:, 0], pairs[:, 1], n_u, n_i)
            rows = np.repeat(np.arange(n_u + n_i), np.diff(rowptr))
            adj = torch.sparse_coo_tensor(np.stack([rows, col]), val, (n_u + n_i, n_u + n_i)).coalesce().to(dev)
        tri = [tuple(torch.from_numpy(rng.integers(0, n, B)).to(dev) for n in (n_u, n_i, n_i)) for _ in range(8)]

        def step(s):
            u, i, j = tri[s % 8]
            if layers:                          # model/LightGCN.py:86-96
                ego = torch.cat([U, V], 0)
                outs = [ego]
                for _ in range(layers):
                    ego = torch.sparse.mm(adj, ego)
                    outs.append(ego)
                out = torch.mean(torch.stack(outs, dim=1), dim=1)
                ue, ie = out[:n_u], out[n_u:]
            else:
                ue, ie = U, V
            a, b, c = ue[u], ie[i], ie[j]
            loss = bpr_loss(a, b, c) + l2_reg(1e-4, a, b, c)
            opt.zero_grad()
            loss.backward()
            opt.step()

        for s in range(5):
            step(s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(steps):
            step(s)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        return {"ms_per_step": ms, "triples_per_s": B / ms * 1e3}

    out = {"note": "stock PyTorch-ROCm (%s) running the reference's calls on this GPU; eager, as the reference" % torch.__version__,
           "train_mf": train(6040, 3706, 0, 128, 4096, 0),
           "train_lightgcn": train(5551, 16980, 131000, 128, 4096, 3)}
    n_users, n_items, d, k = 1024, 10_000_000, 128, 20
    V = item_shard(n_items, d, 0, n_items, dev)
    U = xavier_(n_users, d, 17, dev, 1_000_000)
    rowptr, col = rated_lists(n_users, n_items, 50, seed=4)
    cold = torch.from_numpy(np.where(np.random.default_rng(5).random(n_items) < 0.2)[0]).to(dev)
    rated = [torch.from_numpy(col[rowptr[r]:rowptr[r + 1]].astype(np.int64)).to(dev) for r in range(n_users)]

    def block():                                # model/BaseRecommender.py:172-183 for one user block
        S = U @ V.T
        for r in range(n_users):
            S[r, rated[r]] = -10e8
        S[:, cold] = -10e8
        return torch.topk(S, k, dim=1, largest=True, sorted=True)

    block()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        block()
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / 2
    out["eval"] = {"users_per_block": n_users, "items": n_items, "ms_per_block": sec * 1e3,
                   "items_per_s": n_users * n_items / sec}
    del V, U
    return {"torch_rocm_same_gpu": out}


def train_legs(dev, with_cpu, e2e_epochs=30, timed_epochs=60):
    """Secondary metric of BASELINE.json: BPR triples/s (train): configs[1] (BPR-MF, MovieLens shape, d=128) with
    Adam as the reference and with plain SGD (the north_star's "BPR loss + SGD update"), configs[2] (LightGCN L=3,
    CiteULike shape, d=128).  Two numbers per leg:
      value             one epoch of optimiser steps with the triples already in HBM (kernel-side rate; the roofline
                        refers to it)
      value_end_to_end  whole epochs as the trainers run them -- NumPy-stream-exact sampler (A1)
                        producing epoch e+1 while epoch e trains, upload, reverse index, optimiser steps -- triples
                        per wall second over ``e2e_epochs`` epochs (the reference's timing point, main.py:179-187,
                        without the validation pass)."""
    from coldrec_amd.data.synth import make_dataset
    from coldrec_amd.sampler import EpochPrefetcher, PairwiseSampler
    from coldrec_amd.train import EpochRunner, LGCNEngine, MFEngine
    from coldrec_amd.ops import mf_step_parts as ops_parts
    from coldrec_amd import ops as _ops
    from coldrec_amd.util.databuilder import bipartite_norm_adj_csr
    out = {}
    B, d = 4096, 128
    data_cache = {}
    for name, shape, layers, optim in (("train_mf", "movielens", 0, "adam"), ("train_mf_sgd", "movielens", 0, "sgd"),
                                       ("train_lightgcn", "citeulike", 3, "adam")):
        if shape not in data_cache:
            split = make_dataset(shape, "item", seed=1 if layers == 0 else 2, with_content=False)
            tr = split.warm_train
            _, ru = np.unique(tr[:, 0], return_inverse=True)
            _, ri = np.unique(tr[:, 1], return_inverse=True)
            data_cache[shape] = (split.user_num, split.item_num, tr.shape[0], ru, ri)
        n_u, n_i, n, ru, ri = data_cache[shape]
        smp = PairwiseSampler(ru, ri, n_u, n_i)
        smp.seed(2024)
        u, i, j = smp.epoch(B)
        ts = []
        for _ in range(7):                                         # one epoch of triples per host call, on its own:
            t0 = time.perf_counter()                               # median of 7 (the first calls run on a cold core)
            smp.epoch(B)
            ts.append(time.perf_counter() - t0)
        t_sample = float(np.median(ts))
        g = torch.Generator().manual_seed(2024)
        U0 = torch.nn.init.xavier_uniform_(torch.empty(n_u, d), generator=g)
        V0 = torch.nn.init.xavier_uniform_(torch.empty(n_i, d), generator=g)
        if layers:
            rowptr, col, val = bipartite_norm_adj_csr(ru, ri, n_u, n_i)
            eng = LGCNEngine(U0, V0, rowptr, col, val, layers, 1e-3, 1e-4, dev, optimizer=optim)
        else:
            eng = MFEngine(U0, V0, 1e-3, 1e-4, dev, optimizer=optim)
        tu, ti, tj = (torch.from_numpy(x).to(dev) for x in (u, i, j))
        steps = [(lo, min(lo + B, n)) for lo in range(0, n, B)]
        runner = EpochRunner(eng, n, B)
        runner.run(tu, ti, tj)            # eager warm-up epoch
        runner.run(tu, ti, tj)            # captured into a hipGraph (and replayed once)
        runner.run(tu, ti, tj)            # one more untimed replay: a freshly instantiated graph's first launches, the
        torch.cuda.synchronize()          # allocator's last growth and the clocks' ramp stay outside the timed region
        # timed: EVERY epoch on its own (per epoch: the plans kernel + per-step factors + one graph replay), event to
        # event on the stream the epochs run on, host never waiting in between; the MEDIAN epoch is the leg's number and
        # the spread is reported -- one stalled epoch (a box hiccup) must not own a 20 ms window
        n_ep = timed_epochs
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(n_ep + 1)]
        marks[0].record()
        for e in range(n_ep):
            runner.run(tu, ti, tj)
            marks[e + 1].record()
        torch.cuda.synchronize()
        ep_ms = np.array([marks[e].elapsed_time(marks[e + 1]) for e in range(n_ep)])
        sec = float(np.median(ep_ms)) * 1e-3 / len(steps)
        t0 = time.perf_counter()
        _ops.build_plans_device(tu, ti, tj, B)
        torch.cuda.synchronize()
        t_plans = time.perf_counter() - t0
        # ---- end to end: sampler + prefetch + upload + plans + steps, as model/MF.py's epoch loop runs them
        np.random.seed(2024)
        pref = EpochPrefetcher(smp, B, device=dev)
        for _ in range(3):                                            # warm: speculation running, worker core at speed
            runner.run(*pref.get())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(e2e_epochs):
            runner.run(*pref.get())
        torch.cuda.synchronize()
        sec_e2e = (time.perf_counter() - t0) / e2e_epochs
        pref.close()
        N, nnz = n_u + n_i, (len(val) if layers else 0)
        opt_bytes = 8 if optim == "sgd" else 32                        # dense Adam moves 32 B per element;
        bytes_step = 24 * d * B + opt_bytes * N * d                    # plain SGD reads and writes the parameter only
        if layers:                                                     # + 2L SpMM + layer mean fwd/bwd + dOUT zero
            bytes_step += 2 * layers * (nnz * 8 + (N + 1) * 8 + 2 * N * d * 4) + 2 * (layers + 2) * N * d * 4
        leg = {"metric": "BPR triples/sec (train)", "value": B / sec * (n / (len(steps) * B)), "unit": "triples/s",
               "value_end_to_end": n / sec_e2e, "ms_per_epoch_end_to_end": sec_e2e * 1e3, "end_to_end_epochs": e2e_epochs,
               "ms_per_step": sec * 1e3, "steps_per_epoch": len(steps), "timed_epochs": n_ep,
               "ms_per_step_spread": {"median": float(np.median(ep_ms)) / len(steps), "min": float(ep_ms.min()) / len(steps),
                                      "max": float(ep_ms.max()) / len(steps), "mean": float(ep_ms.mean()) / len(steps),
                                      "p90": float(np.percentile(ep_ms, 90)) / len(steps),
                                      "stalled_epoch_seen": bool(ep_ms.max() > 2.0 * np.median(ep_ms)),
                                      "how": "each of %d hipGraph epochs timed event to event; ms_per_step = median epoch "
                                             "/ steps per epoch" % n_ep},
               "config": {"workload": "configs[%d] %s, %s-shaped synthetic (%d users x %d items, %d train triples), "
                                      "d=%d, B=%d, %s" % (2 if layers else 1, "LightGCN L=3" if layers else "BPR-MF",
                                                          shape, n_u, n_i, n, d, B,
                                                          "plain SGD (torch.optim.SGD defaults)" if optim == "sgd" else "dense Adam")},
               "sampler": "host (csrc/sampler.hip, persistent worker thread, pinned async upload)",
               "host_sampler_s_per_epoch": t_sample, "device_plan_s_per_epoch": t_plans,
               "roofline": {"bound": "hbm", "achieved": bytes_step / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": bytes_step / sec / 1e9 / HBM_PEAK_GBS, "bytes_per_step": bytes_step,
                            "traffic": None, "note": "whole step (all kernels of one optimiser step)"}}
        # fabric-side bytes per launch of the step's dominant kernel from the committed PMC record (same kernel, same grid)
        tr = None
        if layers:
            tr = measured_traffic("spmm_csr_kernel<8>", None)
            what = "spmm_csr_kernel<8> (FETCH_SIZE x2 + WRITE_SIZE) per launch; a step has %d such launches" % (2 * layers)
        elif getattr(eng, "fused", False):
            tr = measured_traffic("mf_step_kernel<32, %d>" % (1 if optim == "sgd" else 0), float(ops_parts(n_u + n_i, d) * 256))
            what = "mf_step_kernel<32> = the whole step (FETCH_SIZE x2 + WRITE_SIZE); it keeps no gradient table"
        if tr:
            leg["roofline"].update({"traffic": tr[0], "traffic_source": "committed profile " + tr[1], "traffic_note": what})
        if with_cpu:
            from oracle import ref_port

            def make_port():
                if layers:
                    return ref_port.LGCNPort(U0.numpy(), V0.numpy(), ref_port.coo_adj(rowptr, col, val), layers, 1e-3, 1e-4,
                                             optimizer=optim)
                return ref_port.MFPort(U0.numpy(), V0.numpy(), 1e-3, 1e-4, optimizer=optim)

            def cpu_steps(port, count):
                t0 = time.perf_counter()
                for s in range(count):
                    lo, hi = steps[s % len(steps)]
                    port.step(u[lo:hi], i[lo:hi], j[lo:hi])
                return (time.perf_counter() - t0) / count

            # tiny ATen ops do not scale to every core: take the best of a few thread counts, bounded time
            best = None
            for th in sorted({os.cpu_count(), min(32, os.cpu_count()), min(8, os.cpu_count())}):
                torch.set_num_threads(th)
                port = make_port()
                cpu_steps(port, 1)
                dt = cpu_steps(port, 2)
                if best is None or dt < best[0]:
                    best = (dt, th)
            torch.set_num_threads(best[1])
            port = make_port()
            cpu_steps(port, 1)
            n_cpu = int(max(2, min(60, 6.0 / best[0])))
            dt = cpu_steps(port, n_cpu)
            torch.set_num_threads(os.cpu_count())
            leg["cpu_baseline"] = {"value": B / dt, "unit": "triples/s", "cores": best[1], "kind": "port",
                                   "sample": "%d optimiser steps of the same epoch (torch autograd + torch.optim.%s%s) on %d "
                                             "threads (best of 8/32/all), sampler excluded"
                                             % (n_cpu, "SGD" if optim == "sgd" else "Adam",
                                                ", torch.sparse.mm COO" if layers else "", best[1])}
        out[name] = leg
        del eng, runner
    return out


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
                        "note": "peak is the nominal dense fp16 figure; a bare v_mfma_f32_32x32x16_f16 stream with random "
                                "operands sustains 0.65-0.68 of it on this chip (power-limited, ~1.57 GHz; 0.60 once the A "
                                "fragments come from LDS at one read per two MFMAs): tools/probes/mfma_energy_probe.hip, "
                                "profiles/r02_mfma_energy_probe.log; hipBLASLt fp16 GEMMs reach 0.60 at best and 0.25 at this "
                                "K = 256 shape: profiles/r02_gemm_f16_probe.log"}}
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


def train_dp_leg(dev, world, rank):
    """N > 1 only: the data-parallel BPR-MF step of SURVEY.md 8(e) on the MovieLens-shaped config (tables and
    Adam state replicated, batch sharded, RCCL all-reduce of the 4 batch sums and of the dense gradient), eager
    launches.  At this size the 5 MB gradient all-reduce is latency-bound: reported as measured."""
    import torch.distributed as dist
    from coldrec_amd.data.synth import make_dataset
    from coldrec_amd.sampler import PairwiseSampler
    from coldrec_amd.train import DPContext, MFEngine
    B, d = 4096, 128
    split = make_dataset("movielens", "item", seed=1, with_content=False)
    tr = split.warm_train
    _, ru = np.unique(tr[:, 0], return_inverse=True)
    _, ri = np.unique(tr[:, 1], return_inverse=True)
    n_u, n_i, n = split.user_num, split.item_num, tr.shape[0]
    smp = PairwiseSampler(ru, ri, n_u, n_i)
    smp.seed(2024)                                   # same stream on every rank: replicated sampler
    u, i, j = (torch.from_numpy(x).to(dev) for x in smp.epoch(B))
    g = torch.Generator().manual_seed(2024)
    U0 = torch.nn.init.xavier_uniform_(torch.empty(n_u, d), generator=g)
    V0 = torch.nn.init.xavier_uniform_(torch.empty(n_i, d), generator=g)
    eng = MFEngine(U0, V0, 1e-3, 1e-4, dev)
    eng.enable_data_parallel(DPContext(world, rank))
    steps = [(lo, min(lo + B, n)) for lo in range(0, n, B)]
    for lo, hi in steps[:8]:
        eng.step(u[lo:hi], i[lo:hi], j[lo:hi])
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for lo, hi in steps:
        eng.step(u[lo:hi], i[lo:hi], j[lo:hi])
    torch.cuda.synchronize()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    sec = float(dt.item()) / len(steps)
    chk = eng.E.double().sum().reshape(1)
    lo_, hi_ = chk.clone(), chk.clone()
    dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
    return {"metric": "BPR triples/sec (train)", "value": n / (sec * len(steps)), "unit": "triples/s",
            "ms_per_step": sec * 1e3, "replicas_identical": bool(lo_.item() == hi_.item()),
            "config": {"workload": "configs[1] BPR-MF, movielens-shaped synthetic, d=%d, global B=%d sharded over %d GPUs, "
                                   "dense Adam, all-reduce(4 sums) + all-reduce(gradient %d bytes) per step"
                                   % (d, B, world, (n_u + n_i) * d * 4), "parallelism": "dp%d" % world}}


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


def train_xl(dev, steps, warm, lazy=False):
    """HBM-roofline case for the training kernels: tables far beyond every cache.  ``lazy``: the touched-rows
    replay of dense Adam (same bits, crh_adam_rows_f32) instead of the dense pass; the per-batch reverse index
    is built inside the timed step and the final flush of all rows is timed as well."""
    from coldrec_amd import ops
    from coldrec_amd.train import MFEngine
    n_u, n_i, d, B = 1_000_000, 10_000_000, 128, 65536
    eng = MFEngine.from_table(xavier_(n_u + n_i, d, 1, dev, n_i), n_u, 1e-3, 1e-4)
    if lazy:
        eng.enable_lazy_adam()
    g = torch.Generator(device=dev).manual_seed(3)
    tri = [(torch.randint(0, n_u, (B,), generator=g, device=dev, dtype=torch.int32),
            torch.randint(0, n_i, (B,), generator=g, device=dev, dtype=torch.int32),
            torch.randint(0, n_i, (B,), generator=g, device=dev, dtype=torch.int32)) for _ in range(8)]
    sec, spread = _time_steps_each(lambda s: eng.step(*tri[s % 8]), steps, warm)
    out = {"metric": "BPR triples/sec (train)", "value": B / sec, "unit": "triples/s", "ms_per_step": sec * 1e3,
           "ms_per_step_spread": spread,
           "config": {"workload": "S-TRAIN-XL: BPR-MF, 1M users x 10M items, d=128, B=65536, %s"
                                  % ("dense Adam replayed on touched rows (bit-identical)" if lazy else "dense Adam")}}
    if lazy:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.sync_tables()
        torch.cuda.synchronize()
        flush = time.perf_counter() - t0
        rows = 3 * B
        # compulsory bytes of a lazy step: gather + gradient rows as before, p/m/v of the touched rows read and
        # written twice (catch-up, step), their gradient rows read and cleared
        bytes_step = 24 * d * B + rows * d * 4 * (6 + 6 + 2)
        out.update({"flush_all_rows_ms": flush * 1e3, "steps_since_flush": steps + warm,
                    "value_with_flush": B * steps / (sec * steps + flush),
                    "roofline": {"bound": "hbm", "achieved": bytes_step / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": bytes_step / sec / 1e9 / HBM_PEAK_GBS, "bytes_per_step": bytes_step,
                                 "traffic": None, "note": "bytes of the touched rows only; the dense formulation "
                                 "would move %d bytes per step" % (24 * d * B + 32 * (n_u + n_i) * d)}})
    else:
        bytes_step = 24 * d * B + 32 * (n_u + n_i) * d
        tr = measured_traffic("adam_dense_kernel", float(16384 * 256))      # dominant kernel of the step
        out["roofline"] = {"bound": "hbm", "achieved": bytes_step / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": bytes_step / sec / 1e9 / HBM_PEAK_GBS, "bytes_per_step": bytes_step,
                           "traffic": tr[0] if tr else None,
                           "traffic_source": ("committed profile " + tr[1]) if tr else None,
                           "traffic_note": "adam_dense_kernel FETCH_SIZE x2 + WRITE_SIZE per launch" if tr else None}
    return out


def xl_graph(dev, n_u, n_i, n_inter, seed):
    """S-TRAIN-XL interaction graph built ON the GPU: users uniform, items Zipf(0.8) with shuffled ids,
    distinct pairs, then the symmetric bipartite adjacency D^-1/2 A D^-1/2 over n_u + n_i nodes as CSR (int64 rowptr,
    ascending int32 col, fp32 val = d_inv[row] * d_inv[col] -- util/databuilder.py:220-254 restated with torch ops; the
    CiteULike-sized legs use the product's own host builder)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    N = n_u + n_i
    u = torch.randint(0, n_u, (n_inter,), generator=g, device=dev)
    cdf = torch.cumsum(torch.arange(1, n_i + 1, device=dev, dtype=torch.float64).pow_(-0.8), 0)
    cdf /= cdf[-1].clone()
    it = torch.searchsorted(cdf, torch.rand(n_inter, generator=g, device=dev, dtype=torch.float64)).clamp_(max=n_i - 1)
    del cdf
    it = torch.randperm(n_i, generator=g, device=dev)[it]
    key = torch.unique(u * n_i + it)
    del u, it
    u, it = key // n_i, key % n_i + n_u
    del key
    rows, cols = torch.cat([u, it]), torch.cat([it, u])
    del u, it
    deg = torch.bincount(rows, minlength=N)
    order = torch.argsort(rows * N + cols)
    rows, cols = rows[order], cols[order]
    del order
    rowptr = torch.zeros(N + 1, dtype=torch.int64, device=dev)
    torch.cumsum(deg, 0, out=rowptr[1:])
    d_inv = torch.where(deg > 0, deg.to(torch.float32).pow(-0.5), torch.zeros((), device=dev))
    val = d_inv[rows] * d_inv[cols]
    return rowptr, cols.to(torch.int32), val, deg


def train_xl_lightgcn(dev, steps, warm, n_u=1_000_000, n_i=10_000_000, n_inter=200_000_000, d=128, L=3, B=65536):
    """the LightGCN step where HBM is the bound -- SURVEY.md 8(d)'s S-TRAIN-XL WITH its graph
    (N = 1.1e7 nodes, E2 ~ 4e8 stored edges, d = 128, L = 3, B = 65 536).  Every layer state is 5.6 GB, far beyond L2
    and Infinity Cache, so every gathered neighbour row is an HBM access: the honest figures are the gather rate
    (E2 x d x 4 bytes per SpMM over its time) against the HBM peak, and SURVEY's formula (which counts the dense operand
    once) beside it; their ratio is the gathered-row re-read factor the formula leaves out."""
    from coldrec_amd import ops
    from coldrec_amd.train import LGCNEngine
    t0 = time.perf_counter()
    rowptr, col, val, deg = xl_graph(dev, n_u, n_i, n_inter, 7)
    torch.cuda.synchronize()
    t_graph = time.perf_counter() - t0
    N, E2 = n_u + n_i, int(col.numel())
    eng = LGCNEngine.from_device(xavier_(N, d, 1, dev, n_i), n_u, rowptr, col, val, L, 1e-3, 1e-4)
    g = torch.Generator(device=dev).manual_seed(3)
    # triples from the graph itself: a stored (user, item) edge as the positive, a uniform item as the negative
    tri = []
    for _ in range(4):
        e = torch.randint(0, int(rowptr[n_u]), (B,), generator=g, device=dev)
        uu = (torch.searchsorted(rowptr[:n_u + 1], e, right=True) - 1).to(torch.int32)
        tri.append((uu, (col[e] - n_u).to(torch.int32), torch.randint(0, n_i, (B,), generator=g, device=dev, dtype=torch.int32)))

    def step(s):
        u, i, j = tri[s % 4]
        eng.step(u, i, j, plan=ops.build_plans_device(u, i, j, B)[0])

    sec = _time_steps(step, steps, warm)
    # the SpMM alone (forward layer 1 of the step: gathers E, reads E as acc_in, writes the next layer's input and the sum)
    def one_spmm(_s):
        ops.spmm_csr(eng.rowptr, eng.col, eng.val, eng.E, y=eng.X[0], acc_in=eng.E, s_in=1.0, acc_out=eng.OUT, s_out=1.0,
                     sched=eng.sched)
    spmm_sec = _time_steps(one_spmm, 3, 1)
    spmm_formula = E2 * 8 + (N + 1) * 4 + 2 * N * d * 4
    layer_mean = 2 * (L + 2) * N * d * 4
    bytes_step = 2 * L * spmm_formula + layer_mean + 24 * d * B + 32 * N * d
    gathered = E2 * d * 4
    G = 1
    while G < d // 4 and G < 64:
        G <<= 1
    tr = measured_traffic("spmm_csr_kernel<%d>" % G, None)
    return {"metric": "BPR triples/sec (train)", "value": B / sec, "unit": "triples/s", "ms_per_step": sec * 1e3,
            "config": {"workload": "S-TRAIN-XL with its graph: LightGCN L=%d, %d users + %d items, %d stored edges "
                                   "(mean degree %.1f, max %d), d=%d, B=%d, Adam in the last SpMM's epilogue"
                                   % (L, n_u, n_i, E2, E2 / N, int(deg.max()), d, B),
                       "graph_build_s": t_graph, "heavy_workgroups": int(eng.sched.c.n_multi), "segment": eng.sched.seg},
            "roofline": {"bound": "hbm", "achieved": bytes_step / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": bytes_step / sec / 1e9 / HBM_PEAK_GBS, "bytes_per_step": bytes_step,
                         "formula": "SURVEY.md 8(d): 2L (E2*8 + (N+1)*4 + 2*N*d*4) + 2 (L+2) N d 4 + 24 d B + 32 N d",
                         "traffic": None},
            "spmm": {"ms": spmm_sec * 1e3, "formula_bytes": spmm_formula,
                     "formula_GBps": spmm_formula / spmm_sec / 1e9, "formula_frac": spmm_formula / spmm_sec / 1e9 / HBM_PEAK_GBS,
                     "gathered_row_bytes": gathered, "gather_GBps": gathered / spmm_sec / 1e9,
                     "gather_frac_of_hbm_peak": gathered / spmm_sec / 1e9 / HBM_PEAK_GBS,
                     "reread_factor_of_the_dense_operand": gathered / (N * d * 4.0),
                     "traffic": tr[0] if tr else None, "traffic_source": ("committed profile " + tr[1]) if tr else None,
                     "note": "every gathered neighbour row (512 B) is an HBM access at this size: the kernel's bound is "
                             "the random-row gather rate, not SURVEY's formula, which counts the dense operand once"}}


def legs_summary(result):
    """leg -> [ms per step (or per launch), fraction of its roofline] for every leg of the line, compact, printed as the
    LAST key so a reader that keeps only the tail of the line still sees every leg."""
    out = {"headline": [round(result["ms_per_step"], 3), round(result["roofline"]["frac"], 4)]}
    for name, leg in result.items():
        if not isinstance(leg, dict):
            continue
        if "roofline" in leg and isinstance(leg["roofline"], dict) and "frac" in leg["roofline"]:
            ms = leg.get("ms_per_step", leg.get("ms"))
            out[name] = [None if ms is None else round(ms, 4), round(leg["roofline"]["frac"], 4)]
            if isinstance(leg.get("shard_8gpu"), dict):
                out[name + ".shard_8gpu"] = [round(leg["shard_8gpu"]["ms_per_step"], 3),
                                             round(leg["shard_8gpu"]["frac_of_fp16_mfma_peak"], 4)]
        elif name == "eval_midsize":
            for shape, v in leg.items():
                out["eval_midsize." + shape] = [round(v["ms"], 3), round(v["frac_of_fp32_mfma_peak"], 4)]
        elif name == "eval_validation":
            for shape, v in leg.items():
                out["eval_validation." + shape] = [round(v["ms"], 4), round(
                    2.0 * 128 * v["users"] * v["items"] / (v["ms"] * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)]
    if "eval_e2e" in result:
        e = result["eval_e2e"]
        out["eval_e2e.metrics_share_of_ranking"] = [round((e["seconds"]["membership_gpu"] + e["seconds"]["host_metrics"]) * 1e3, 1),
                                                     round(e["metrics_share_of_ranking"], 4)]
    if "train_xl_lightgcn" in result and "spmm" in result["train_xl_lightgcn"]:
        sp = result["train_xl_lightgcn"]["spmm"]
        out["train_xl_lightgcn.spmm"] = [round(sp["ms"], 3), round(sp["formula_frac"], 4)]
    return out


def self_launch(n_gpus, argv):
    """``python bench.py --gpus N`` (N > 1) without a launcher around it: run
    ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py
    <same flags>`` as a child process -- one rank per GPU, RCCL over xGMI -- with stdout / stderr inherited (rank 0's JSON
    line reaches the caller unchanged) and return the child's exit code.  The parent never initialises the GPU."""
    import socket
    import subprocess
    with socket.socket() as s:                    # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL needs it across processes)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n_gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    if os.environ.get("CRH_BENCH_DRY_LAUNCH") == "1":     # tests: show the command, start nothing
        print(json.dumps({"launch": cmd}), flush=True)
        return 0
    sys.stdout.flush()
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dp-leg-timeout", type=int, default=240, help="N > 1: seconds the data-parallel training leg may take")
    ap.add_argument("--items", type=int, default=10_000_000)
    ap.add_argument("--users", type=int, default=1_000_000, help="rows of the user table")
    ap.add_argument("--users-per-step", type=int, default=131072,
                    help="users scored per step; 2048 wave-groups of 64 users fill the 256 CUs without cutting the "
                         "item range (every extra cut repeats the top-k warm-up of each user)")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--dtype", choices=["f32", "f16"], default="f32",
                    help="f32 = exact fp32 MFMA, bit-exact parity mode (headline); f16 = fp16 tables with fp32 "
                         "accumulation (BASELINE.json configs[4]: --dtype f16 --items 50000000 --dim 256 --users 100000)")
    ap.add_argument("--generator", action="store_true",
                    help="configs[4]: build the (fp16) item table with the DropoutNet item tower first and report its rate")
    ap.add_argument("--shard", choices=["items", "users"], default="items",
                    help="N > 1: 'items' = item table row-sharded + all-gather(top-k) + merge (north_star); 'users' = item "
                         "table replicated, user block cut across the ranks (zero-exchange validation mode)")
    ap.add_argument("--n-splits", type=int, default=0)
    ap.add_argument("--masks", choices=["warm", "none"], default="warm",
                    help="'warm' = rated CSR + 20%% cold-item bitmap (default); 'none' = diagnostic run without masks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the secondary train legs (N=1 only)")
    ap.add_argument("--lazy-adam", action="store_true", help="with --train-xl: touched-rows replay of dense Adam")
    ap.add_argument("--train-only", action="store_true", help="only the two secondary train legs (profiling aid)")
    ap.add_argument("--train-xl", action="store_true",
                    help="only run the S-TRAIN-XL roofline case of SURVEY.md 8(d) (1M users x 10M items, d=128, "
                         "B=65536 MF steps; 22.5 GB of state) and print its JSON line")
    ap.add_argument("--train-xl-lightgcn", action="store_true",
                    help="only run the S-TRAIN-XL LightGCN leg (1.1e7 nodes, ~4e8 stored edges, d=128, L=3)")
    ap.add_argument("--cpu-sample-users", type=int, default=2048,
                    help="users of the CPU baseline (blocks of 256 against the WHOLE item table, SURVEY.md 8(d))")
    ap.add_argument("--cpu-budget-s", type=float, default=30.0, help="the CPU baseline stops after this many seconds")
    ap.add_argument("--legs", default="eval_f16,mask_topk,train_xl,train_xl_lightgcn,train,eval_validation,eval_midsize,eval_e2e,torch_rocm",
                    help="N=1: secondary legs carried in the same JSON line (comma separated; 'none' = headline only)")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle self-check of the last timed step")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks ourselves.  Decided BEFORE anything touches the GPU
        # (no torch.cuda call has run in this process), and the launcher is a CHILD process whose output and exit code
        # are relayed -- a process that has initialised the GPU must never be replaced by another program.
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    t_main = time.perf_counter()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus must agree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: coldrec_amd has no CPU path")
    # test hook (CRH_BENCH_BACKEND=gloo): several ranks on ONE GPU over gloo, to exercise the N > 1 control flow on a
    # 1-GPU box; the driver's runs use RCCL ("nccl") with one GPU per rank
    backend = os.environ.get("CRH_BENCH_BACKEND", "nccl")
    local_dev = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from coldrec_amd import ops
    from coldrec_amd.eval import ShardedTopK

    if args.train_xl:
        print(json.dumps(train_xl(dev, args.steps, args.warmup, lazy=args.lazy_adam)), flush=True)
        return
    if args.train_xl_lightgcn:
        print(json.dumps(train_xl_lightgcn(dev, args.steps, args.warmup)), flush=True)
        return
    if args.train_only:
        legs = train_legs(dev, not args.no_cpu_baseline)
        legs.update(validation_eval_leg(dev))
        print(json.dumps(legs), flush=True)
        return

    I, d, k, Bu = args.items, args.dim, args.k, args.users_per_step
    lo, hi = rank * I // world, (rank + 1) * I // world
    if args.shard == "users":
        lo, hi = 0, I
    tdtype = torch.float16 if args.dtype == "f16" else torch.float32
    Bu = min(Bu, args.users)
    gen_leg = None
    if args.generator:           # configs[4]: the item table is GENERATED by the DropoutNet tower, then ranked in fp16
        assert args.dtype == "f16", "--generator produces the fp16 table of configs[4]: use --dtype f16"
        V, gen_leg = dropoutnet_generator(dev, hi - lo, d, item_lo=lo)
    else:
        V = item_shard(I, d, lo, hi, dev, tdtype)
    n_blocks = args.warmup + args.steps
    n_user_rows = min(args.users, Bu * n_blocks)
    U = xavier_(n_user_rows, d, 17, dev, args.users).to(tdtype)
    rowptr, col = rated_lists(n_user_rows, I, 50, seed=4)
    cold = np.where(np.random.default_rng(5).random(I) < 0.2)[0]       # 'warm' setting: cold items masked
    bitmap = ops.make_bitmap(I, cold, dev)
    blocks = []
    for b in range(n_blocks):
        u0 = (b * Bu) % max(n_user_rows - Bu + 1, 1)
        rp = torch.from_numpy(rowptr[u0:u0 + Bu + 1] - rowptr[u0]).to(dev)
        rc = torch.from_numpy(col[rowptr[u0]:rowptr[u0 + Bu]]).to(dev)
        blocks.append((torch.arange(u0, u0 + Bu, dtype=torch.int32, device=dev), rp, rc))

    if args.shard == "users":
        from coldrec_amd.eval import UserShardedTopK
        engine = UserShardedTopK(V, k, world, rank)
    else:
        engine = ShardedTopK(V, item_base=lo, n_items_global=I, k=k, world=world, rank=rank)
    events = HipEvents(args.steps)

    def step(b, ev=None):
        users, rp, rc = blocks[b]
        m = (None, None, None) if args.masks == "none" else (rp, rc, bitmap)
        if args.shard == "users":
            return engine.topk(U, users, *m)
        return engine.topk(U, users, *m, n_splits=args.n_splits, kernel_events=ev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(args.warmup):
        step(w)
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        out = step(args.warmup + s, events.pairs[s])
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    ms_per_step = dt / args.steps * 1e3
    value = Bu * I / (ms_per_step * 1e-3)
    if args.shard == "users":                      # no kernel events in this mode: the step time stands in
        kern_ms = ms_per_step
        flops_per_launch = 2.0 * d * ((rank + 1) * Bu // world - rank * Bu // world) * I
    else:
        kern_ms = float(np.mean(events.elapsed_ms()))
        flops_per_launch = 2.0 * d * Bu * (hi - lo)
    achieved = flops_per_launch / (kern_ms * 1e-3) / 1e12
    peak_tf = MFMA_F16_PEAK_TFLOPS if args.dtype == "f16" else MFMA_F32_PEAK_TFLOPS

    import zlib
    result_crc = zlib.crc32(out[1].cpu().numpy().tobytes(), zlib.crc32(out[0].cpu().numpy().tobytes()))
    # what THIS rank launches, as the library reports it: Bu users x its item shard (item shards), or its share of the user
    # block x the whole table (user shards)
    users_here = Bu if args.shard == "items" else (rank + 1) * Bu // world - rank * Bu // world
    rt_head = route_of(users_here, hi - lo, d, k, args.dtype, masks=args.masks == "warm", n_splits=args.n_splits)
    rt_full = route_of(Bu, I, d, k, args.dtype, masks=args.masks == "warm", n_splits=args.n_splits)
    result = {
        "metric": "ranked items/sec (full-catalogue eval)", "value": value, "unit": "items/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype,
        "data": "synthetic",
        "result_crc32": result_crc,      # of the last timed step's (scores, ids): independent of N by construction
        "config": {"workload": "configs[%d] full-catalogue eval: %d-row user table x %d items, d=%d, k=%d, %s tables, "
                               "user block %d per step, rated CSR (mean ~50) + 20%% cold-item bitmap ('warm' setting), "
                               "%s over %d GPU(s)"
                               % (4 if args.dtype == "f16" else 3, args.users, I, d, k, args.dtype, Bu,
                                  "item table row-sharded" if args.shard == "items" else "user block sharded, items replicated",
                                  world),
                   "users_per_step": Bu, "items": I, "dim": d, "k": k,
                   "parallelism": ("single GPU" if world == 1 else
                                   "item-row-shard x%d + all_gather(top-k) + canonical merge" % world if args.shard == "items"
                                   else "user-block-shard x%d (items replicated) + all_gather" % world)},
        "roofline": {"bound": "mfma", "kernel": rt_head["label"],
                     "route": {q: rt_head[q] for q in ("route", "seeded", "prefix_items", "n_splits")},
                     "achieved": achieved,
                     "peak": peak_tf, "unit": "TFLOP/s", "frac": achieved / peak_tf,
                     "kernel_ms": kern_ms, "flops_per_launch": flops_per_launch, "traffic": None},
    }
    esz = 2 if args.dtype == "f16" else 4
    if world > 1 and (rt_head["kernel"], rt_head["seeded"]) != (rt_full["kernel"], rt_full["seeded"]):
        # e.g. 8 ranks x 1.25 M items: the shard takes the per-wave kernel, the one-GPU profile is of the workgroup kernel over
        # the whole table -- its counters say nothing about this launch
        result["roofline"]["traffic_note"] = ("no committed counter record of this rank's route (%s; the one-GPU launch over the "
                                              "whole table runs %s)" % (rt_head["label"], rt_full["label"]))
    else:
        # the committed PMC record of THE KERNEL THE LIBRARY NAMED for this launch (same instantiation, same grid)
        tr = measured_traffic(rt_head["profile_patterns"], rt_head["grid_threads"],
                              prefer="_eval_pmc" if args.dtype == "f32" else "_f16")
        if tr:
            # the profile is of the one-GPU launch over the whole table; a rank's launch streams its shard only
            # (the traffic is 8 XCD L2s x the table streamed, so it scales with the shard)
            frac = (hi - lo) / float(args.items)
            result["roofline"]["traffic"] = tr[0] * frac
            result["roofline"]["traffic_source"] = "committed profile " + tr[1] + (
                "" if world == 1 else " (one-GPU launch) scaled by the shard's share of the table")
            result["roofline"]["traffic_note"] = ("L2-miss (fabric-side) bytes per launch from %s; mostly served by the "
                                                  "256 MB Infinity Cache, compulsory HBM bytes are %d" % (
                                                      tr[1], (hi - lo) * d * esz + Bu * d * esz))

    if gen_leg is not None and rank == 0:
        result["dropoutnet_generator"] = gen_leg
    # ---- self-check of the LAST timed step + CPU baseline: both need the tables on the host (rank 0)
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    want_verify = rank == 0 and not args.no_verify and args.shard == "items" and not args.generator
    U_cpu = V_cpu = None
    want_verify = want_verify and args.dtype == "f32"          # the fp16 table is checked on the device by the eval_f16 leg
    if (want_cpu or want_verify) and float(I) * d * 4 > 24e9:
        # e.g. configs[4] as a command (--dtype f16 --items 50000000 --dim 256): an fp32 host copy of the whole table would be
        # 51 GB; the CPU legs are for tables a host comfortably holds
        result["host_legs_skipped"] = "fp32 host copy of the %d x %d table = %.0f GB: no CPU baseline / host oracle check" % (
            I, d, float(I) * d * 4 / 1e9)
        want_cpu = want_verify = False
    if want_cpu or want_verify:
        U_cpu = U.float().cpu().numpy()
        if world == 1:
            V_cpu = V.float().cpu().numpy()
        else:   # rank 0 holds one shard: rebuild the whole table chunk by chunk (same seeds as every rank used)
            V_cpu = np.concatenate([item_shard(I, d, c, min(c + CHUNK_ROWS, I), dev, tdtype).float().cpu().numpy()
                                    for c in range(0, I, CHUNK_ROWS)])
    if want_verify:
        b_last = args.warmup + args.steps - 1
        users_last, rp_last, rc_last = blocks[b_last]
        result["verified_users"] = verify_users(
            "headline", out[0].cpu().numpy(), out[1].cpu().numpy(), users_last.cpu().numpy().astype(np.int64), U_cpu, V_cpu,
            rp_last.cpu().numpy() if args.masks != "none" else np.zeros(Bu + 1, np.int64), rc_last.cpu().numpy(),
            None if args.masks == "none" else cold, k, n_check=64)
        result["verified_against"] = "oracle/topk_oracle.c (canonical fp32 fma chain, score desc / index asc), bit-exact"
    if want_cpu:
        nu = min(args.cpu_sample_users, n_user_rows)
        rate, done, secs = cpu_baseline(torch.from_numpy(U_cpu[:nu]), torch.from_numpy(V_cpu), rowptr[:nu + 1], col, cold, k,
                                        block=256, budget_s=args.cpu_budget_s)
        result["cpu_baseline"] = {
            "value": rate, "unit": "items/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "%d users (of %d sampled; blocks of 256, stopped at the %.0f s budget) x the WHOLE %d-item table, same "
                      "masks, torch %s matmul+mask+topk on %d threads, %.1f s of CPU work"
                      % (done, nu, args.cpu_budget_s, V_cpu.shape[0], torch.__version__, os.cpu_count(), secs)}
    del U_cpu, V_cpu
    legs = [] if args.legs == "none" else [x.strip() for x in args.legs.split(",") if x.strip()]
    wall = {"headline_incl_setup_checks_cpu_baseline": round(time.perf_counter() - t_main, 1)}   # where the run's minutes go
    if world > 1 and not args.no_train and args.dtype == "f32":
        # Secondary leg.  The headline line must survive it: an exception is caught below, and if a rank gets stuck
        # in a collective (the others would wait for ever) a watchdog on every rank prints what it has (rank 0) and
        # ends the process with a NON-ZERO code, so a stuck RCCL run is never reported as a success.
        import threading
        done = threading.Lock()

        def bail():
            if not done.acquire(blocking=False):
                return
            if rank == 0:
                result["train_mf_dp"] = {"error": "no result within %d s (watchdog): a rank is stuck in a collective"
                                                  % args.dp_leg_timeout}
                print(json.dumps(result), flush=True)
            os._exit(4)

        dog = threading.Timer(args.dp_leg_timeout, bail)
        dog.daemon = True
        dog.start()
        try:                                   # every rank takes part; rank 0 reports
            del V, U, engine
            torch.cuda.empty_cache()
            leg = train_dp_leg(dev, world, rank)
            if rank == 0:
                result["train_mf_dp"] = leg
        except Exception as e:
            if rank == 0:
                result["train_mf_dp"] = {"error": repr(e)[:300]}
        if not done.acquire(blocking=False):   # the watchdog fired while the leg was finishing: it reports and exits
            os._exit(4)
        dog.cancel()
    if rank == 0 and world == 1 and not args.no_train and args.dtype == "f32" and legs:
        del V, U, engine, blocks, out
        torch.cuda.empty_cache()
        for leg_name, fn in (("eval_f16", lambda: eval_f16_leg(dev)), ("mask_topk", lambda: mask_topk_leg(dev)),
                             ("train_xl", lambda: {"train_xl": train_xl(dev, 9, 2)}),
                             ("train_xl_lightgcn", lambda: {"train_xl_lightgcn": train_xl_lightgcn(dev, 2, 1)}),
                             ("train", lambda: train_legs(dev, not args.no_cpu_baseline)),
                             ("eval_validation", lambda: validation_eval_leg(dev)),
                             ("eval_midsize", lambda: midsize_eval_leg(dev)),
                             ("eval_e2e", lambda: eval_e2e_leg(dev)),
                             ("torch_rocm", lambda: torch_rocm_leg(dev))):
            if leg_name in legs:
                t_leg = time.perf_counter()
                result.update(fn())
                torch.cuda.empty_cache()
                wall[leg_name] = round(time.perf_counter() - t_leg, 1)
    shard_leg = result.get("eval_midsize", {}).get("%dx%d" % (Bu, I // 8)) if rank == 0 and world == 1 else None
    if shard_leg:
        # what the 8-GPU run is expected to give -- every rank ranks the same user block against its
        # eighth of the catalogue (measured above as a one-GPU launch of exactly that shape, whatever kernel the dispatcher
        # picks for it), then one all-gather of 8 k bytes per user and a 160-candidate merge (< 1 % of the step)
        result["predicted_scaling_8gpu"] = {
            "value": 8.0 * shard_leg["items_per_s"] / value,
            "shard_items_per_s": shard_leg["items_per_s"], "whole_items_per_s": value,
            "note": "8 x rate(one rank's %d-item shard) / rate(the whole %d-item table) on one GPU; the shard runs the per-wave "
                    "kernel (%.3f of the fp32 MFMA peak), the whole table the workgroup kernel (%.3f): no measured 8-GPU "
                    "number exists, the driver's SCALE run is the only one" % (
                        I // 8, I, shard_leg["frac_of_fp32_mfma_peak"], result["roofline"]["frac"])}
    if rank == 0:
        wall["total"] = round(time.perf_counter() - t_main, 1)
        result["wall_s"] = wall
        result["legs_summary"] = legs_summary(result)          # LAST key: survives a truncated tail of the line
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
