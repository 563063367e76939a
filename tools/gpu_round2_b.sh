#!/bin/bash
# scratch runner of round 2 (rewritten per experiment during the round); last form: GPU tests + a short default bench
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | grep -v -i -E "rccl|amdgpu|^$" | tail -4
timeout 1500 python bench.py --steps 2 --warmup 1 > gpurun_out/r02_quick_bench.json 2> gpurun_out/r02_quick_bench.err
echo "bench rc=$?"; tail -c 400 gpurun_out/r02_quick_bench.json
