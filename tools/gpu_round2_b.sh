#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
L=gpurun_out/r02_rowmask.log
: > $L
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_round2_gpu.py tests/test_e2e_gpu.py -x -q 2>&1 | grep -v -i -E "rccl|amdgpu|^$" | tail -25 | tee -a $L
for f in 1 0; do
  CRH_LGCN_ROWMASK=$f python bench.py --train-only --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); v=d['train_lightgcn']; print('rowmask=$f lightgcn ms/step', v['ms_per_step'], 'e2e', v.get('value_end_to_end')); v=d['train_mf']; print('   mf ms/step', v['ms_per_step'], 'e2e', v.get('value_end_to_end'))" | tee -a $L
done
CRH_SPMM_ROWS=4 python bench.py --train-only --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); v=d['train_lightgcn']; print('rows=4 lightgcn ms/step', v['ms_per_step'])" | tee -a $L
CRH_SPMM_ROWS=2 python bench.py --train-only --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); v=d['train_lightgcn']; print('rows=2 lightgcn ms/step', v['ms_per_step'])" | tee -a $L
