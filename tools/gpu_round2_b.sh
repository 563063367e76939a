#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for i in 1 2; do
python bench.py --train-only --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); v=d['train_lightgcn']; print('occ7 lightgcn ms/step', v['ms_per_step'], 'e2e', v.get('value_end_to_end'))"
done
timeout 600 python -m pytest tests/test_train_gpu.py -x -q -k "spmm or lgcn or lightgcn" 2>&1 | grep -v -i -E "rccl|amdgpu|^$" | tail -3
