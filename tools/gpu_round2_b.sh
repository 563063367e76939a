#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
python tools/mask_topk_probe2.py 2>&1 | grep -v amdgpu | tee gpurun_out/r02_mask_probe2.log
python tools/mask_topk_probe.py 2>&1 | grep mask_topk | tee -a gpurun_out/r02_mask_probe2.log
timeout 900 python -m pytest tests/test_score_topk_gpu.py tests/test_round2_gpu.py tests/test_e2e_gpu.py -x -q 2>&1 | grep -v -i -E "rccl|amdgpu|^$" | tail -5
