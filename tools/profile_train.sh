#!/bin/bash
# rocprofv3 record of the training legs (bench.py --train-only: BPR-MF on the MovieLens shape, LightGCN L=3 on the
# CiteULike shape): kernel stats + HBM-side counters of every kernel (separate --pmc passes).
# Usage (on the GPU box, from the repo root): bash tools/profile_train.sh <tag>
set -u
TAG=${1:-r01_train}
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
ARGS="--train-only --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -- python3 bench.py $ARGS > "$OUT/bench_under_stats.json" 2> "$OUT/stats.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -- python3 bench.py $ARGS > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -- python3 bench.py $ARGS > /dev/null 2> "$OUT/pmc_write.err"
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT/pmc_l2" -- python3 bench.py $ARGS > /dev/null 2> "$OUT/pmc_l2.err"
python3 tools/prof_summary.py "$TAG" "$OUT/stats" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_l2" > "$OUT/summary.txt" 2>&1
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* gpurun_out/profiles_$TAG/
cp "$OUT/bench_under_stats.json" gpurun_out/profiles_$TAG/${TAG}_bench_under_rocprof.json
tail -3 "$OUT/stats.err"; ls gpurun_out/profiles_$TAG
