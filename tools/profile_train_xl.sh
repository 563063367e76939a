#!/bin/bash
# rocprofv3 record of the S-TRAIN-XL MF step (dense Adam and the touched-rows replay): kernel stats + HBM counters.
set -u
TAG=${1:-r01_train_xl}
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
ARGS="--train-xl --steps 6 --warmup 2"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -- python3 bench.py $ARGS > "$OUT/bench_dense.json" 2> "$OUT/stats.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -- python3 bench.py $ARGS > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -- python3 bench.py $ARGS > /dev/null 2> "$OUT/pmc_write.err"
python3 tools/prof_summary.py "$TAG" "$OUT/stats" "$OUT/pmc_fetch" "$OUT/pmc_write" > "$OUT/summary.txt" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/stats_lazy" -- python3 bench.py $ARGS --lazy-adam > "$OUT/bench_lazy.json" 2> "$OUT/stats_lazy.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch_lazy" -- python3 bench.py $ARGS --lazy-adam > /dev/null 2> "$OUT/pmc_fetch_lazy.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write_lazy" -- python3 bench.py $ARGS --lazy-adam > /dev/null 2> "$OUT/pmc_write_lazy.err"
python3 tools/prof_summary.py "${TAG}_lazy" "$OUT/stats_lazy" "$OUT/pmc_fetch_lazy" "$OUT/pmc_write_lazy" >> "$OUT/summary.txt" 2>&1
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* gpurun_out/profiles_$TAG/
ls gpurun_out/profiles_$TAG
