#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_round2_gpu.py tests/test_train_gpu.py -x -q 2>&1 | grep -v -i -E "rccl|amdgpu|^$" | tail -30 | tee gpurun_out/r02_rowshard.log
