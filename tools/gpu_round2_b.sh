#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 1500 python bench.py > gpurun_out/r02_c_bench.json 2> gpurun_out/r02_c_bench.err
echo "bench rc=$? after $SECONDS s"
tail -c 600 gpurun_out/r02_c_bench.json
