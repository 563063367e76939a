"""Shared pieces of the bench legs: synthetic inputs, HIP-event timing, the CPU baseline, the oracle self-check, the lookup of
committed counter records and the library's own route report (bench.py re-exports everything here)."""
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


def gpu_clocks():
    """{"sclk_mhz", "mclk_mhz"} of GPU 0 as rocm-smi reports them right now (an ordinary child process; None when the tool
    is missing or says nothing parseable): what a bandwidth-bound leg ran at, beside its number."""
    import re
    import subprocess
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF"))}
    try:
        txt = subprocess.run(["rocm-smi", "-d", "0", "--showclocks"], capture_output=True, text=True, timeout=20, env=env).stdout
    except (OSError, subprocess.SubprocessError):
        return None
    out = {}
    for key, name in (("sclk", "sclk_mhz"), ("mclk", "mclk_mhz"), ("fclk", "fclk_mhz")):
        m = re.search(r"%s clock level:?\s*\S*:?\s*\(?(\d+)\s*Mhz" % key, txt, flags=re.I)
        if m:
            out[name] = int(m.group(1))
    return out or None


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


def bare_mfma_loops():
    """What a bare v_mfma_f32_32x32x16_f16 stream sustains ON THIS BOX IN THIS RUN (tools/probes/mfma_energy_probe, ~1 s, run
    as a child process): the chip is power-limited under random fp16 operands and boxes differ by +-8 %, so the fp16 leg
    reports its kernel against these in-run figures beside the nominal 2.5 PF.  None when the probe binary is absent or fails."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "tools", "probes", "mfma_energy_probe")
    if not os.path.exists(exe):        # built by __graft_entry__.build() / tools/profile_round.sh, never from inside a leg: under
        return None                    # rocprofv3 a compiler started here would be a GPU-initialised process that execs
    # the probe is an ordinary child; a profiler's preload (LD_PRELOAD / ROCP*) is not handed down to it
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF"))}
    try:
        out = subprocess.run([exe], check=True, timeout=120, capture_output=True, text=True, env=env).stdout
    except (OSError, subprocess.SubprocessError):
        return None
    best = {}
    for name, frac in re.findall(r"^(.+?)\s+[\d.]+ ms\s+\d+ TFLOP/s \(([\d.]+) of 2.5 PF\)", out, flags=re.M):
        best[name.strip()] = max(best.get(name.strip(), 0.0), float(frac))
    reg, lds = best.get("4 waves, 4 acc, A in registers"), best.get("4 waves, 4 acc, A from LDS")
    if reg is None or lds is None:
        return None
    return {"register_fed": reg, "lds_fed_same_shape": lds, "unit": "fraction of 2.5 PF",
            "what": "bare v_mfma_f32_32x32x16_f16 loops, random operands, 4 waves x 4 accumulators per CU (the DMA kernel's shape): "
                    "A fragments from registers / from LDS (one ds_read_b128 per four MFMAs); best of 3 interleaved passes"}


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
    if r["route"] == "fused-dma":     # <T, D, ring slots, flag form>: the two forms of the fp32 kernel are separate profile rows
        fl = r.get("dma_form") == "flags"
        r["label"] = "%s<%s,%d,%s>" % (r["kernel"], dtype, d, "flags" if fl else "barrier")
        r["profile_patterns"] = ("%s<%s, %d, 4, %s>" % (r["kernel"], ctype, d, "true" if fl else "false"),
                                 "%s%sLi%dELi4ELb%dE" % (r["kernel"], mangled, d, 1 if fl else 0))
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
