#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 600 python tools/fuzz_train_ops.py --minutes 6 --seed 42 2>&1 | grep -v amdgpu | tail -6 | tee gpurun_out/fuzz_repro.log
