#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
time timeout 900 python bench.py --legs torch_rocm,eval_validation --no-cpu-baseline --no-verify --steps 1 --warmup 0 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d.get('torch_rocm_same_gpu'), indent=1)); print(d.get('eval_validation'))" | tee gpurun_out/r02_torch_leg.log
