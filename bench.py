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

from bench_legs.common import *  # noqa: E402,F401,F403
from bench_legs.common import _median_ms, _time_steps, _time_steps_each  # noqa: E402,F401
from bench_legs.eval_legs import (ArrayTruth, SyntheticEvalData, dropoutnet_generator, eval_d64_leg, eval_e2e_leg, eval_f16_leg,  # noqa: E402,F401
                                  mask_topk_leg, midsize_eval_leg, validation_eval_leg)
from bench_legs.train_legs import (torch_rocm_leg, train_dp_leg, train_legs, train_xl, train_xl_dp_leg, train_xl_lightgcn, xl_graph)  # noqa: E402,F401


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
                    2.0 * v.get("d", 128) * v["users"] * v["items"] / (v["ms"] * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)]
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
    ap.add_argument("--legs", default="eval_d64,eval_f16,mask_topk,train_xl,train_xl_lightgcn,train,eval_validation,eval_midsize,eval_e2e,torch_rocm",
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
    if not want_cpu:
        del U_cpu, V_cpu
    # (the CPU baseline itself runs AFTER the legs, below: half a minute of every host core at full tilt leaves worker threads
    # spinning and clocks ramped, and the legs that time host-side work beside the GPU -- the trainers' end-to-end epochs --
    # must not run in its wake)
    legs = [] if args.legs == "none" else [x.strip() for x in args.legs.split(",") if x.strip()]
    wall = {"headline_incl_setup_and_checks": round(time.perf_counter() - t_main, 1)}   # where the run's minutes go
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
            torch.cuda.empty_cache()
            leg = train_xl_dp_leg(dev, world, rank)       # the size SURVEY.md 8(e) names for the scaling figure
            if rank == 0:
                result["train_xl_dp"] = leg
        except Exception as e:
            if rank == 0:
                result["train_mf_dp"] = {"error": repr(e)[:300]}
        if not done.acquire(blocking=False):   # the watchdog fired while the leg was finishing: it reports and exits
            os._exit(4)
        dog.cancel()
    if rank == 0 and world == 1 and not args.no_train and args.dtype == "f32" and legs:
        del V, U, engine, blocks, out
        torch.cuda.empty_cache()
        leg_fns = {"eval_d64": lambda: eval_d64_leg(dev), "eval_f16": lambda: eval_f16_leg(dev), "mask_topk": lambda: mask_topk_leg(dev),
                   "train_xl": lambda: {"train_xl": train_xl(dev, 30, 3)},
                   "train_xl_lightgcn": lambda: {"train_xl_lightgcn": train_xl_lightgcn(dev, 2, 1)},
                   "train": lambda: train_legs(dev, not args.no_cpu_baseline),
                   "eval_validation": lambda: validation_eval_leg(dev), "eval_midsize": lambda: midsize_eval_leg(dev),
                   "eval_e2e": lambda: eval_e2e_leg(dev), "torch_rocm": lambda: torch_rocm_leg(dev)}
        for leg_name in legs:                  # in the order --legs names them
            if leg_name in leg_fns:
                t_leg = time.perf_counter()
                result.update(leg_fns[leg_name]())
                torch.cuda.empty_cache()
                wall[leg_name] = round(time.perf_counter() - t_leg, 1)
    if want_cpu:
        t_cpu = time.perf_counter()
        nu = min(args.cpu_sample_users, n_user_rows)
        rate, done, secs = cpu_baseline(torch.from_numpy(U_cpu[:nu]), torch.from_numpy(V_cpu), rowptr[:nu + 1], col, cold, k,
                                        block=256, budget_s=args.cpu_budget_s)
        result["cpu_baseline"] = {
            "value": rate, "unit": "items/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "%d users (of %d sampled; blocks of 256, stopped at the %.0f s budget) x the WHOLE %d-item table, same "
                      "masks, torch %s matmul+mask+topk on %d threads, %.1f s of CPU work, after every GPU leg of the run"
                      % (done, nu, args.cpu_budget_s, V_cpu.shape[0], torch.__version__, os.cpu_count(), secs)}
        del U_cpu, V_cpu
        wall["cpu_baseline"] = round(time.perf_counter() - t_cpu, 1)
    shard_leg = result.get("eval_midsize", {}).get("%dx%d" % (Bu, I // 8)) if rank == 0 and world == 1 else None
    if shard_leg:
        # what the 8-GPU run is expected to give -- every rank ranks the same user block against its
        # eighth of the catalogue (measured above as a one-GPU launch of exactly that shape, whatever kernel the dispatcher
        # picks for it), then one all-gather of 8 k bytes per user and a 160-candidate merge (< 1 % of the step)
        result["predicted_scaling_8gpu"] = {
            "value": 8.0 * shard_leg["items_per_s"] / value,
            "shard_items_per_s": shard_leg["items_per_s"], "whole_items_per_s": value,
            "note": "8 x rate(one rank's %d-item shard) / rate(the whole %d-item table) on one GPU; the shard's route is %s (%.3f "
                    "of the fp32 MFMA peak), the whole table's %s (%.3f): no measured 8-GPU number exists, the driver's SCALE "
                    "run is the only one" % (I // 8, I, shard_leg["route"]["route"], shard_leg["frac_of_fp32_mfma_peak"],
                                             result["roofline"]["route"]["route"], result["roofline"]["frac"])}
    if rank == 0:
        wall["total"] = round(time.perf_counter() - t_main, 1)
        result["wall_s"] = wall
        result["legs_summary"] = legs_summary(result)          # LAST key: survives a truncated tail of the line
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

