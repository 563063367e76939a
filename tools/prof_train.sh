#!/bin/bash
# Kernel table of the training legs only (rocprofv3 --stats through tools/prof_summary.py); run on the GPU box from the repo root.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
TAG=${1:-tmp_train}
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/ptr -- python3 bench.py --train-only --no-cpu-baseline > gpurun_out/ptr.json 2> gpurun_out/ptr.err
python3 tools/prof_summary.py "$TAG" gpurun_out/ptr < /dev/null
rm -rf gpurun_out/ptr
cp "profiles/${TAG}_kernel_stats.csv" gpurun_out/ && head -14 "profiles/${TAG}_kernel_stats.csv" | cut -c1-200
