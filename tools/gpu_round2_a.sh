#!/bin/bash
# first GPU pass of round 2: parity tests, then the default bench line
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest_a.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02_pytest_a.log
tail -15 gpurun_out/r02_pytest_a.log
timeout 1500 python bench.py > gpurun_out/r02_bench_a.json 2> gpurun_out/r02_bench_a.err
echo "bench rc=$?"
tail -c 3000 gpurun_out/r02_bench_a.json
tail -5 gpurun_out/r02_bench_a.err
