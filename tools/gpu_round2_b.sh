#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
L=gpurun_out/r02_fuzz.log
: > $L
timeout 600 python tools/fuzz_score_topk.py --minutes 6 --seed 21 2>&1 | grep -v amdgpu | tail -3 | tee -a $L
timeout 600 python tools/fuzz_train_ops.py --minutes 5 --seed 22 2>&1 | grep -v amdgpu | tail -3 | tee -a $L
timeout 600 python bench.py --train-xl --lazy-adam --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -2 | cut -c1-1500 | tee -a $L
