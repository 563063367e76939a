#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/...) into small summaries under profiles/.

    python tools/prof_summary.py <tag> <stats_dir> [<pmc_dir> ...]

Writes profiles/<tag>_kernel_stats.csv (the --stats table, our kernels first) and
profiles/<tag>_pmc.json (per kernel: mean of every collected counter per launch).
Counter corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE/WRITE_SIZE are KiB; on gfx950
FETCH_SIZE reads half of a 16-B/lane stream, so hbm_read_bytes = FETCH_SIZE*1024*2.
"""
import collections
import csv
import glob
import json
import os
import sys


def _db_stats(stats_dir, tag):
    """rocprofv3's default output here is a rocpd sqlite file: rebuild the --stats kernel table from it."""
    import sqlite3
    for f in glob.glob(os.path.join(stats_dir, "**", "*_results.db"), recursive=True):
        c = sqlite3.connect(f)
        per = collections.defaultdict(list)
        for name, dur in c.execute("select name, duration from kernels"):
            per[name].append(float(dur))
        total = sum(sum(v) for v in per.values()) or 1.0
        rows = []
        for name, v in per.items():
            mean = sum(v) / len(v)
            sd = (sum((x - mean) ** 2 for x in v) / len(v)) ** 0.5
            rows.append([name, len(v), int(sum(v)), "%.6f" % mean, "%.4f" % (100 * sum(v) / total), int(min(v)),
                         int(max(v)), "%.6f" % sd])
        ours = lambda n: 0 if ("anonymous namespace)::" in n and "at::native" not in n) else 1
        rows.sort(key=lambda r: (ours(r[0]), -r[2]))
        with open(f"profiles/{tag}_kernel_stats.csv", "w", newline="") as o:
            w = csv.writer(o)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            w.writerows(rows[:40])


def _long_cluster(v):
    """One kernel instantiation launched on very different amounts of work (bench.py's fp16 leg: the 50 M-item launches and
    the 6.25 M-item shard launches have the same name and grid): a mean over all of them describes neither.  Keep the
    launches within 2x of the largest value (the long launches, which the roofline lines refer to); return (kept, dropped)."""
    if len(v) < 2 or min(v) <= 0 or max(v) <= 2.0 * min(v):
        return v, 0
    top = max(v)
    kept = [x for x in v if x >= top / 2.0]
    return kept, len(v) - len(kept)


def _db_pmc(d, out):
    import sqlite3
    for f in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
        c = sqlite3.connect(f)
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        # one row per (dispatch, counter, hardware instance): sum the instances of a dispatch first
        for name, counter, value in c.execute(
                "select name, counter_name, sum(counter_value) from pmc_events group by dispatch_id, name, counter_name"):
            if "at::native" in name or "rocprim" in name:
                continue
            agg[name][counter].append(float(value))
        meta = {}
        for row in c.execute("select name, duration, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, "
                             "grid_x, workgroup_x from kernels"):
            if row[0] in agg:
                m = meta.setdefault(row[0], collections.defaultdict(list))
                m["_duration_ns"].append(float(row[1]))
                for key, v in zip(("_VGPR_Count", "_Accum_VGPR_Count", "_SGPR_Count", "_LDS_Block_Size", "_Grid_Size",
                                   "_Workgroup_Size"), row[2:]):
                    m[key] = [float(v)]
        for name, cs in agg.items():
            o = out.setdefault(name, {})
            for cname, v in list(cs.items()) + list(meta.get(name, {}).items()):
                v, dropped = _long_cluster(v) if (not cname.startswith("_") or cname == "_duration_ns") else (v, 0)
                o[cname] = sum(v) / len(v)
                if not cname.startswith("_"):
                    o[cname + "_launches"] = len(v)
                    if dropped:
                        o[cname + "_other_launches"] = dropped


def main():
    tag, stats_dir, pmc_dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    os.makedirs("profiles", exist_ok=True)
    _db_stats(stats_dir, tag)
    for f in glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.reader(open(f)))
        head, body = rows[0], rows[1:]
        body.sort(key=lambda r: (0 if ("score_topk" in r[0] or "crh" in r[0] or "anonymous namespace)::" in r[0][:40]) else 1,
                                 -float(r[2])))
        with open(f"profiles/{tag}_kernel_stats.csv", "w", newline="") as o:
            w = csv.writer(o)
            w.writerow(head)
            w.writerows(body[:25])
    out = {}
    for d in pmc_dirs:
        _db_pmc(d, out)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"]
                if "at::native" in name or "rocprim" in name:
                    continue
                agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
                agg[name]["_duration_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
                for key in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Grid_Size", "Workgroup_Size"):
                    agg[name]["_" + key] = [float(r[key])]
            for name, cs in agg.items():
                o = out.setdefault(name, {})
                for c, v in cs.items():
                    o[c] = sum(v) / len(v)
                    if not c.startswith("_"):
                        o[c + "_launches"] = len(v)
    for name, o in out.items():
        if "FETCH_SIZE" in o:
            o["hbm_read_bytes_corrected"] = o["FETCH_SIZE"] * 1024 * 2
        if "WRITE_SIZE" in o:
            o["hbm_write_bytes"] = o["WRITE_SIZE"] * 1024
    with open(f"profiles/{tag}_pmc.json", "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True)[:3000])


if __name__ == "__main__":
    main()
