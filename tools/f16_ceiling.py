#!/usr/bin/env python3
"""is 0.60 of the nominal 2.5 PF within reach of the fp16 scoring kernel on this chip?  One record
(profiles/<tag>_f16_ceiling.json) from one box and one run of tools/profile_round4.sh:
  bare      the bare v_mfma_f32_32x32x16_f16 loops (tools/probes/mfma_energy_probe): what an MFMA stream of the kernel's shape
            sustains on random operands -- the chip is power-limited, the clock follows the load;
  shipped   the timed kernel: fraction of the nominal peak (un-profiled run), and from a separate counter pass
            SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) = matrix-pipe occupancy, GRBM_GUI_ACTIVE / 8 /
            duration = effective clock; their product ("busy GHz") x 1024 SIMDs x 1024 flop/clk is the achieved rate;
  ablated   the SAME kernel (-DCRH_PROFILE build, CRH_SCORE_ABLATE=1) with the selection epilogue switched off.
"""
import json
import os
import re
import sys

tag, out = sys.argv[1], sys.argv[2]
SIMDS, XCDS, NOMINAL_GHZ = 1024, 8, 2.4


def leg(path):
    try:
        line = [ln for ln in open(path) if ln.startswith("{")][-1]
        return json.loads(line)["eval_f16"]
    except (OSError, IndexError, KeyError, ValueError):
        return None


def pmc(path):
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None
    for name, v in rec.items():
        if ("score_topk_dma_kernel" in name or "score_topk_wg_kernel" in name) and ("F16" in name or "Float16" in name) \
                and v.get("_duration_ns", 0) > 1e9:                  # the 50 M-item launch, not the shard's
            cyc = v["GRBM_GUI_ACTIVE"] / XCDS
            busy = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * cyc)
            ghz = cyc / v["_duration_ns"]
            return {"kernel": name, "duration_ms_profiled": v["_duration_ns"] / 1e6, "mfma_busy_cycles": v["SQ_VALU_MFMA_BUSY_CYCLES"],
                    "gui_active_cycles_per_xcd": cyc, "matrix_pipe_occupancy": busy, "effective_clock_GHz": ghz,
                    "busy_GHz": busy * ghz, "frac_of_nominal_from_counters": busy * ghz / NOMINAL_GHZ,
                    "vgprs": v.get("_VGPR_Count"), "lds_bytes": v.get("_LDS_Block_Size")}
    return None


bare = []
try:
    for ln in open(os.path.join(out, "f16_bare_loops.log")):
        m = re.match(r"(.+?)\s+([\d.]+) ms\s+(\d+) TFLOP/s \(([\d.]+) of 2.5 PF\)\s+clock ([\d.]+) GHz\s+([\d.]+) cycles per MFMA", ln)
        if m:
            bare.append({"variant": m.group(1).strip(), "tflops": float(m.group(3)), "frac": float(m.group(4)),
                         "clock_GHz_per_wave_counter": float(m.group(5)), "cycles_per_mfma_of_one_wave": float(m.group(6))})
except OSError:
    pass
best = {}
for b in bare:                                             # the probe interleaves its variants three times: keep the best of each
    if b["variant"] not in best or b["frac"] > best[b["variant"]]["frac"]:
        best[b["variant"]] = b
res = {"nominal_peak_TFLOPs": 2500.0, "bare_loops": list(best.values())}
for key, f_leg, f_pmc in (("shipped", "f16_shipped.json", "profiles/%s_f16_pmc.json" % tag),
                          ("ablated_selection", "f16_ablated.json", "profiles/%s_f16_ablated_pmc.json" % tag),
                          ("profile_build_selection_on", "f16_profile_build.json", None)):
    lg = leg(os.path.join(out, f_leg))
    entry = {}
    if lg:
        entry.update({"frac_of_nominal_unprofiled": lg["roofline"]["frac"], "kernel_ms_unprofiled": lg["roofline"]["kernel_ms"],
                      "tflops": lg["roofline"]["achieved"], "verified_users": lg.get("verified_users")})
    if f_pmc:
        p = pmc(f_pmc)
        if p:
            entry["counters"] = p
    res[key] = entry
ceil_same = max([b["frac"] for b in best.values() if b["variant"] == "4 waves, 4 acc, A from LDS"] or [None])
ceil_128 = max([b["frac"] for b in best.values() if b["variant"] == "4 waves, 4 acc, A in registers"] or [None])
res["reading"] = {
    "bare_loop_of_this_kernels_shape (4 waves, 4 accumulators, A from LDS)": ceil_same,
    "bare_loop_register_fed (4 waves, 4 accumulators, A in registers)": ceil_128,
    "shipped_over_register_fed_bare_loop": (res["shipped"].get("frac_of_nominal_unprofiled") or 0) / ceil_128 if ceil_128 else None,
    "shipped_over_its_bare_loop": (res["shipped"].get("frac_of_nominal_unprofiled") or 0) / ceil_same if ceil_same else None,
    "ablated_over_its_bare_loop": (res["ablated_selection"].get("frac_of_nominal_unprofiled") or 0) / ceil_same if ceil_same else None,
}
os.makedirs("profiles", exist_ok=True)
json.dump(res, open("profiles/%s_f16_ceiling.json" % tag, "w"), indent=1)
print(json.dumps(res, indent=1))
