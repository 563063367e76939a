#!/bin/bash
# Round profile: rocprofv3 kernel stats of the default bench command + PMC passes for the dominant kernel.
# Usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
set -u
TAG=${1:-r01}
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -- python3 bench.py --no-cpu-baseline > "$OUT/bench_under_stats.json" 2> "$OUT/stats.err"
ARGS="--no-cpu-baseline --no-train --steps 2 --warmup 1"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d "$OUT/pmc_sq" -- python3 bench.py $ARGS > /dev/null 2> "$OUT/pmc_sq.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -- python3 bench.py $ARGS > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -- python3 bench.py $ARGS > /dev/null 2> "$OUT/pmc_write.err"
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -d "$OUT/pmc_l2" -- python3 bench.py $ARGS > /dev/null 2> "$OUT/pmc_l2.err"
python3 tools/prof_summary.py "$TAG" "$OUT/stats" "$OUT/pmc_sq" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_l2" > "$OUT/summary.txt" 2>&1
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* gpurun_out/profiles_$TAG/ 2>/dev/null
cp "$OUT/bench_under_stats.json" gpurun_out/profiles_$TAG/${TAG}_bench_under_rocprof.json
tail -5 "$OUT/stats.err"; ls gpurun_out/profiles_$TAG
