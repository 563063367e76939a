#!/bin/bash
# Kernel table of chosen bench legs: bash tools/prof_legs.sh <tag> <legs> (on the GPU box, from the repo root)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
TAG=${1:-tmp_legs}; LEGS=${2:-eval_validation}
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/pl -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify --legs "$LEGS" > gpurun_out/pl.json 2> gpurun_out/pl.err
python3 tools/prof_summary.py "$TAG" gpurun_out/pl < /dev/null
rm -rf gpurun_out/pl
cp "profiles/${TAG}_kernel_stats.csv" gpurun_out/ && head -12 "profiles/${TAG}_kernel_stats.csv" | cut -c1-180
