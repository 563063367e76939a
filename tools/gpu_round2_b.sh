#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
L=gpurun_out/r02_slowpath.log
: > $L
timeout 900 python -m pytest tests/test_score_topk_gpu.py tests/test_round2_gpu.py -x -q 2>&1 | grep -v -i -E "rccl|amdgpu|^$" | tail -5 | tee -a $L
python tools/midsize_probe.py perwav,auto 2>&1 | grep -E "^(fused|dense|auto|perwav)" | tee -a $L
for d in 256 128; do
  python tools/f16_probe.py --dim $d --reps 2 2>&1 | grep "^f16" | tee -a $L
done
python tools/score_probe.py --users 131072 --reps 2 --variants pack+mask 2>&1 | grep "kernel ms" | tee -a $L
