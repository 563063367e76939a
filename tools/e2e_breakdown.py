#!/usr/bin/env python3
"""Where an end-to-end BPR-MF epoch goes (MovieLens shape).  One tool, three views (round 2's e2e_breakdown{,2,3}.py):

    python tools/e2e_breakdown.py phases   every phase of EpochRunner.run timed with a device sync, the prefetcher wait,
                                           and the unsynchronised loop beside it (default)
    python tools/e2e_breakdown.py host     host cost of the pieces WITHOUT device syncs; CPU quota / affinity of the box
    python tools/e2e_breakdown.py worker   the sampler's worker thread alone: back to back, with gaps, beside a busy GPU
"""
import os, sys, time, threading
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coldrec_amd import _lib, ops
from coldrec_amd.data.synth import make_dataset
from coldrec_amd.sampler import EpochPrefetcher, PairwiseSampler
from coldrec_amd.train import EpochRunner, MFEngine

mode = sys.argv[1] if len(sys.argv) > 1 else "phases"
assert mode in ("phases", "host", "worker"), __doc__
dev = torch.device("cuda:0")
split = make_dataset("movielens", "item", seed=1, with_content=False)
tr = split.warm_train
_, ru = np.unique(tr[:, 0], return_inverse=True)
_, ri = np.unique(tr[:, 1], return_inverse=True)
n_u, n_i, n, B, d = split.user_num, split.item_num, tr.shape[0], 4096, 128
smp = PairwiseSampler(ru, ri, n_u, n_i)
if mode in ("phases", "host"):
    g = torch.Generator().manual_seed(2024)
    U0 = torch.nn.init.xavier_uniform_(torch.empty(n_u, d), generator=g)
    V0 = torch.nn.init.xavier_uniform_(torch.empty(n_i, d), generator=g)
    eng = MFEngine(U0, V0, 1e-3, 1e-4, dev)
    runner = EpochRunner(eng, n, B)

if mode == "phases":
    np.random.seed(2024)
    pref = EpochPrefetcher(smp, B)
    for _ in range(3):
        runner.run(*pref.get())
    torch.cuda.synchronize()

    T = {}
    def tick(label, fn):
        torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        T.setdefault(label, []).append(time.perf_counter() - t); return r

    for _ in range(8):
        u, i, j = tick("prefetch.get (wait for the sampler thread)", pref.get)
        tick("h2d triples", lambda: [dst.copy_(torch.as_tensor(src), non_blocking=True) for dst, src in ((runner.u, u), (runner.i, i), (runner.j, j))])
        plans = tick("plans kernel", lambda: ops.build_plans_device(runner.u, runner.i, runner.j, B))
        tick("plans copy", lambda: runner.plans.copy_(plans))
        tick("mf tables", lambda: ops.mf_step_tables(runner.plans, runner.u, runner.i, runner.j, B, eng.user_num, eng.item_num, out=runner.tables))
        sc = tick("scalars host", lambda: ops.adam_step_scalars(eng.step_count + 1, len(runner.steps), eng.lr))
        tick("scalars h2d", lambda: runner.scalars.copy_(torch.from_numpy(sc), non_blocking=True))
        tick("graph replay", lambda: runner.graph.replay())
        eng.step_count += len(runner.steps)
    for k, v in T.items():
        print(f"{k:50s} median {np.median(v) * 1e3:7.3f} ms")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        runner.run(*pref.get())
    torch.cuda.synchronize()
    print("unsynchronised loop: %.3f ms per epoch" % ((time.perf_counter() - t0) * 100), flush=True)
    pref.close()
    t0 = time.perf_counter()
    for _ in range(10):
        smp.epoch(B)
    print("sampler alone: %.3f ms per epoch" % ((time.perf_counter() - t0) * 100))

if mode == "host":
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
        try: print(f, open(f).read().strip())
        except Exception as e: print(f, "n/a")
    print("affinity", len(os.sched_getaffinity(0)))
    u, i, j = smp.epoch(B)
    for _ in range(3):
        runner.run(u, i, j)
    torch.cuda.synchronize()
    # host cost of the pieces WITHOUT device syncs
    def host_ms(fn, reps=10):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps): fn()
        dt = (time.perf_counter() - t) / reps; torch.cuda.synchronize(); return dt * 1e3
    print("host ms graph.replay() call", host_ms(lambda: runner.graph.replay()))
    print("host ms runner.run (device-resident triples)", host_ms(lambda: runner.run(runner.u, runner.i, runner.j)))
    print("host ms runner.run (host triples)", host_ms(lambda: runner.run(u, i, j)))
    pu, pi, pj = (torch.from_numpy(x).pin_memory() for x in (u, i, j))
    print("host ms runner.run (pinned host triples)", host_ms(lambda: runner.run(pu, pi, pj)))
    def sampler_ms():
        t = time.perf_counter(); smp.epoch(B); return (time.perf_counter() - t) * 1e3
    print("sampler alone ms", np.median([sampler_ms() for _ in range(5)]))
    # sampler on a thread while the main thread (a) sleeps (b) spins in synchronize behind a replay (c) enqueues run()
    for mode in ("sleep", "sync", "run"):
        res = []
        for _ in range(5):
            out = {}
            th = threading.Thread(target=lambda: out.setdefault("ms", sampler_ms()))
            th.start()
            if mode == "sleep": time.sleep(0.004)
            elif mode == "sync": runner.graph.replay(); torch.cuda.synchronize()
            else: runner.run(pu, pi, pj)
            th.join(); torch.cuda.synchronize()
            res.append(out["ms"])
        print("sampler thread ms while main does", mode, np.median(res))

if mode == "worker":
    L = smp._L
    bufs = [torch.empty(n, dtype=torch.int32).pin_memory() for _ in range(3)]
    # 1. the worker thread alone, back to back
    for _ in range(3):
        L.crh_sampler_epoch_async(smp._h, B, *[b.data_ptr() for b in bufs], 0); L.crh_sampler_epoch_wait(smp._h)
    ts = []
    for _ in range(10):
        t = time.perf_counter(); L.crh_sampler_epoch_async(smp._h, B, *[b.data_ptr() for b in bufs], 0); L.crh_sampler_epoch_wait(smp._h); ts.append(time.perf_counter() - t)
    print("worker thread alone, back to back: median %.2f ms" % (np.median(ts) * 1e3))
    ts = []
    for _ in range(10):
        time.sleep(0.003)
        t = time.perf_counter(); L.crh_sampler_epoch_async(smp._h, B, *[b.data_ptr() for b in bufs], 0); L.crh_sampler_epoch_wait(smp._h); ts.append(time.perf_counter() - t)
    print("worker thread with 3 ms gaps: median %.2f ms" % (np.median(ts) * 1e3))
    ts = []
    for _ in range(10):
        t = time.perf_counter(); smp.epoch(B); ts.append(time.perf_counter() - t)
    print("calling thread: median %.2f ms" % (np.median(ts) * 1e3))
    g = torch.Generator().manual_seed(2024)
    eng = MFEngine(torch.nn.init.xavier_uniform_(torch.empty(n_u, d), generator=g), torch.nn.init.xavier_uniform_(torch.empty(n_i, d), generator=g), 1e-3, 1e-4, dev)
    runner = EpochRunner(eng, n, B)
    np.random.seed(2024)
    pref = EpochPrefetcher(smp, B, device=dev)
    for _ in range(3):
        runner.run(*pref.get())
    torch.cuda.synchronize()
    os.environ["CRH_PREFETCH_TIMING"] = "1"
    pref.close()
    def loop(tag, fn, dev_out=True, n_it=20):
        pf = EpochPrefetcher(smp, B, device=dev if dev_out else None)
        if not dev_out: pf.timing = []
        for _ in range(3): fn(pf.get())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for e in range(n_it): fn(pf.get())
        torch.cuda.synchronize()
        tot = (time.perf_counter() - t0) / n_it * 1e3
        print("%-46s %.2f ms per epoch; sampler (finish wait) median %.2f ms" % (tag, tot, np.median([r[0] for r in pf.timing[-10:]]) * 1e3), flush=True)
        pf.close()
    loop("full: upload + run()", lambda tri: runner.run(*tri))
    loop("upload only, no run()", lambda tri: None)
    loop("no upload (host arrays), no GPU work", lambda tri: None, dev_out=False)
    loop("no upload, graph replay only", lambda tri: runner.graph.replay(), dev_out=False)
    x = torch.empty(64 << 20, device=dev)
    loop("no upload, one 20 us kernel per epoch", lambda tri: x[:1024].zero_(), dev_out=False)
    loop("no upload, 200 tiny kernels per epoch", lambda tri: [x[:1024].zero_() for _ in range(200)], dev_out=False)
