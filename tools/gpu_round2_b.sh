#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
L=gpurun_out/r02_f16_probe_d.log
: > $L
timeout 900 python -m pytest tests/test_score_topk_gpu.py -x -q 2>&1 | grep -v -i -E "rccl|amdgpu|^$" | tail -5 | tee -a $L
python tools/f16_probe.py --dim 64 --reps 2 2>&1 | grep "^f16" | tee -a $L
for u in 100000 65536 50000 160000 200000; do
  python tools/f16_probe.py --dim 256 --users $u --reps 2 2>&1 | grep "^f16" | tee -a $L
done
python tools/f16_probe.py --dim 128 --users 100000 --reps 2 2>&1 | grep "^f16" | tee -a $L
