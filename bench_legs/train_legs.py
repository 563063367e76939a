"""Training legs of bench.py: configs[1] / [2] through the engines (with their CPU baselines), the same steps written with stock
torch-ROCm ops, the data-parallel step, and the S-TRAIN-XL roofline runs."""
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
    B = 4096
    data_cache = {}
    cpu_jobs = []
    # the last two: the REFERENCE'S DEFAULT width and depth (main.py:97 --emb_size 64, :94 --layers 2; config/model_param.py),
    # VERDICT r5 #1 -- kernel-side and end-to-end rates only (no second CPU baseline, half the epochs)
    for name, shape, layers, optim, d, cpu_ok, n_timed, n_e2e in (
            ("train_mf", "movielens", 0, "adam", 128, True, timed_epochs, e2e_epochs),
            ("train_mf_sgd", "movielens", 0, "sgd", 128, True, timed_epochs, e2e_epochs),
            ("train_lightgcn", "citeulike", 3, "adam", 128, True, timed_epochs, e2e_epochs),
            ("train_mf_d64", "movielens", 0, "adam", 64, False, timed_epochs // 2, e2e_epochs // 2),
            ("train_lightgcn_L2_d64", "citeulike", 2, "adam", 64, False, timed_epochs // 2, e2e_epochs // 2)):
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
        n_ep = n_timed
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
        for _ in range(12):                                           # warm: speculation running, worker core at speed -- the
            runner.run(*pref.get())                                   # hosts' cores idle at a quarter of their clock and take
        # ~5 epochs of the worker's sampling (25 ms) to ramp: with 3 warm epochs the first timed ones ran 4-9 ms and cost a
        # 15-epoch leg a quarter of its mean (`epoch_interval_ms.longest` used to name epochs 2-6; profiles/r06_e2e_gaps.log)
        torch.cuda.synchronize()
        pref.timing, t_run, t_mark = [], [], []
        t0 = time.perf_counter()
        for _ in range(n_e2e):
            tri_e = pref.get()
            t_r = time.perf_counter()
            runner.run(*tri_e)
            t_mark.append(time.perf_counter())
            t_run.append(t_mark[-1] - t_r)
        torch.cuda.synchronize()
        sec_e2e = (time.perf_counter() - t0) / n_e2e
        gaps = np.diff(np.array([t0] + t_mark))                       # host-side epoch intervals (the host is at most two epochs ahead)
        tm = np.array(pref.timing) if pref.timing else np.zeros((1, 4))
        e2e_host = {"wait_for_sampler_ms": float(np.median(tm[:, 0])) * 1e3, "publish_state_and_upload_ms": float(np.median(tm[:, 1])) * 1e3,
                    "start_next_epoch_ms": float(np.median(tm[:, 2])) * 1e3, "runner_launch_ms": float(np.median(t_run)) * 1e3,
                    "epoch_interval_ms": {"median": float(np.median(gaps)) * 1e3, "max": float(gaps.max()) * 1e3,
                                          "mean": float(gaps.mean()) * 1e3,
                                          "longest": [{"epoch": int(q), "ms": round(float(gaps[q]) * 1e3, 3),
                                                       "wait_publish_start_sync_run_ms": [round(float(x) * 1e3, 3) for x in
                                                                                         list(tm[q]) + [t_run[q]]]}
                                                      for q in np.argsort(-gaps)[:3]] if len(tm) == len(gaps) else None},
                    "note": "medians per epoch on the calling thread: get() = wait + publish/upload + start of the next epoch's "
                            "sampling; run() = copies, plans, per-step factors, graph replay (all asynchronous launches)"}
        pref.timing = None
        pref.close()
        N, nnz = n_u + n_i, (len(val) if layers else 0)
        opt_bytes = 8 if optim == "sgd" else 32                        # dense Adam moves 32 B per element;
        bytes_step = 24 * d * B + opt_bytes * N * d                    # plain SGD reads and writes the parameter only
        if layers:                                                     # + 2L SpMM + layer mean fwd/bwd + dOUT zero
            bytes_step += 2 * layers * (nnz * 8 + (N + 1) * 8 + 2 * N * d * 4) + 2 * (layers + 2) * N * d * 4
        leg = {"metric": "BPR triples/sec (train)", "value": B / sec * (n / (len(steps) * B)), "unit": "triples/s",
               "value_end_to_end": n / sec_e2e, "ms_per_epoch_end_to_end": sec_e2e * 1e3, "end_to_end_epochs": n_e2e,
               "end_to_end_over_kernel_side": (n / sec_e2e) / (B / sec * (n / (len(steps) * B))),
               "ms_per_step": sec * 1e3, "steps_per_epoch": len(steps), "timed_epochs": n_ep,
               "ms_per_step_spread": {"median": float(np.median(ep_ms)) / len(steps), "min": float(ep_ms.min()) / len(steps),
                                      "max": float(ep_ms.max()) / len(steps), "mean": float(ep_ms.mean()) / len(steps),
                                      "p90": float(np.percentile(ep_ms, 90)) / len(steps),
                                      "stalled_epoch_seen": bool(ep_ms.max() > 2.0 * np.median(ep_ms)),
                                      "how": "each of %d hipGraph epochs timed event to event; ms_per_step = median epoch "
                                             "/ steps per epoch" % n_ep},
               "config": {"workload": "configs[%d] %s, %s-shaped synthetic (%d users x %d items, %d train triples), "
                                      "d=%d, B=%d, %s" % (2 if layers else 1, "LightGCN L=%d" % layers if layers else "BPR-MF",
                                                          shape, n_u, n_i, n, d, B,
                                                          "plain SGD (torch.optim.SGD defaults)" if optim == "sgd" else "dense Adam")},
               "sampler": "host (csrc/sampler.hip, persistent worker thread, pinned async upload)",
               "host_sampler_s_per_epoch": t_sample, "device_plan_s_per_epoch": t_plans, "end_to_end_host_ms": e2e_host,
               "roofline": {"bound": "hbm", "achieved": bytes_step / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": bytes_step / sec / 1e9 / HBM_PEAK_GBS, "bytes_per_step": bytes_step,
                            "traffic": None, "note": "whole step (all kernels of one optimiser step)"}}
        # fabric-side bytes per launch of the step's dominant kernel from the committed PMC record (same kernel, same grid)
        tr = None
        if layers:
            tr = measured_traffic("spmm_csr_kernel<8>", None)
            what = ("spmm_csr_kernel<8> (FETCH_SIZE x2 + WRITE_SIZE) per launch; a step has %d such launches + one "
                    "spmm_csr_opt_kernel<8> (the last backward product, optimiser in its epilogue: its own profile row)" % (2 * layers - 1))
        elif getattr(eng, "fused", False):
            tr = measured_traffic("mf_step_kernel<32, %d>" % (1 if optim == "sgd" else 0), float(ops_parts(n_u + n_i, d) * 256))
            what = "mf_step_kernel<32> = the whole step (FETCH_SIZE x2 + WRITE_SIZE); it keeps no gradient table"
        if tr:
            leg["roofline"].update({"traffic": tr[0], "traffic_source": "committed profile " + tr[1], "traffic_note": what})
        if with_cpu and cpu_ok:
            # the CPU port of this leg runs AFTER every GPU leg of this function (below): its hundreds of ATen worker threads keep
            # spinning for a while after a parallel region, and the NEXT leg's end-to-end epochs -- a host thread sampling beside
            # the GPU -- then showed 20-70 ms hiccups (train_mf_sgd 0.46 of its kernel-side rate behind train_mf's CPU baseline,
            # 0.82-0.84 without one before it)
            cpu_jobs.append((name, leg, U0, V0, (rowptr, col, val) if layers else None, layers, optim, steps, (u, i, j)))
        out[name] = leg
        del eng, runner
    if cpu_jobs:
        from oracle import ref_port
    for name, leg, U0, V0, graph, layers, optim, steps, (u, i, j) in cpu_jobs:
        if graph is not None:
            rowptr, col, val = graph

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
    return out


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


def train_xl_dp_leg(dev, world, rank, steps=12, warm=3):
    """N > 1 only: data-parallel training at the size SURVEY.md 8(e) names for it (S-TRAIN-XL: 1 M users x 10 M items, d=128,
    global B = 65 536) with the touched-rows optimiser: row-ownership split of the backward, ONE all-gather of (row id, row)
    slots per step (MFEngine._lazy_step_dp) instead of the dense split's 5.6 GB gradient all-reduce.  Tables, moments and
    step counters are replicated (22.5 GB per rank); the plan, the catch-up, the forward and the optimiser run on every
    replica, the backward is cut G ways.  When several ranks share one GPU (the gloo test hook) beyond two, the tables
    are an eighth of the size -- named in the workload."""
    import torch.distributed as dist
    from coldrec_amd.train import DPContext, MFEngine
    shared_gpu = os.environ.get("CRH_BENCH_BACKEND", "nccl") != "nccl"
    scale = 8 if (shared_gpu and world > 2) else 1
    n_u, n_i, d, B = 1_000_000 // scale, 10_000_000 // scale, 128, 65536
    eng = MFEngine.from_table(xavier_(n_u + n_i, d, 1, dev, n_i), n_u, 1e-3, 1e-4)
    eng.enable_data_parallel(DPContext(world, rank))
    eng.enable_lazy_adam()
    g = torch.Generator(device=dev).manual_seed(3)                   # same stream on every rank: replicated sampler
    tri = [(torch.randint(0, n_u, (B,), generator=g, device=dev, dtype=torch.int32),
            torch.randint(0, n_i, (B,), generator=g, device=dev, dtype=torch.int32),
            torch.randint(0, n_i, (B,), generator=g, device=dev, dtype=torch.int32)) for _ in range(8)]
    for s_ in range(warm):
        eng.step(*tri[s_ % 8])
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for s_ in range(steps):
        eng.step(*tri[(warm + s_) % 8])
    torch.cuda.synchronize()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    sec = float(dt.item()) / steps
    eng.sync_tables()
    chk = eng.E.view(torch.int32).to(torch.int64).sum().reshape(1)          # bit-level checksum of the flushed parameters
    lo_, hi_ = chk.clone(), chk.clone()
    dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
    return {"metric": "BPR triples/sec (train)", "value": B / sec, "unit": "triples/s", "ms_per_step": sec * 1e3, "steps": steps,
            "scaling": "strong", "replicas_identical": bool(lo_.item() == hi_.item()),
            "exchange_bytes_per_step": eng.exchange_bytes_per_step,
            "dense_gradient_allreduce_bytes_it_replaces": (n_u + n_i) * d * 4,
            "config": {"workload": "S-TRAIN-XL%s BPR-MF: %d users x %d items, d=%d, global B=%d, dense Adam replayed on touched "
                                   "rows; backward split by row ownership over %d ranks, one all-gather of (row id, row) slots "
                                   "per step" % (" / %d (ranks share one GPU)" % scale if scale > 1 else "", n_u, n_i, d, B, world),
                       "parallelism": "dp%d" % world}}


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
    clocks0 = gpu_clocks()
    sec, spread = _time_steps_each(lambda s: eng.step(*tri[s % 8]), steps, warm)
    clocks1 = gpu_clocks()
    out = {"metric": "BPR triples/sec (train)", "value": B / sec, "unit": "triples/s", "ms_per_step": sec * 1e3,
           "ms_per_step_spread": spread, "gpu_clocks_before_after": [clocks0, clocks1],
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
        # in-run normaliser (this leg reads 0.60 or 0.71 of the 8 TB/s peak depending on the box and on what ran
        # before it): a plain device-to-device copy of one 5.6 GB table, same process, same moment -- read + write bytes per second
        # -- and the step's traffic as a fraction of it
        cp_ms, _ = _median_ms(lambda: eng.G.copy_(eng.M), 7, warm=2)
        copy_gbs = 2.0 * eng.M.numel() * 4 / (cp_ms * 1e-3) / 1e9
        eng.G.zero_()
        tr = measured_traffic("adam_dense_kernel", float(16384 * 256))      # dominant kernel of the step
        out["roofline"] = {"bound": "hbm", "achieved": bytes_step / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": bytes_step / sec / 1e9 / HBM_PEAK_GBS, "bytes_per_step": bytes_step,
                           "traffic": tr[0] if tr else None,
                           "traffic_source": ("committed profile " + tr[1]) if tr else None,
                           "traffic_note": "adam_dense_kernel FETCH_SIZE x2 + WRITE_SIZE per launch" if tr else None,
                           "copy_GBps_same_run": copy_gbs, "frac_of_copy": bytes_step / sec / 1e9 / copy_gbs,
                           "copy_note": "torch device-to-device copy of one 5.6 GB table (read + write bytes / s), median of 7, "
                                        "right after the timed steps"}
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
