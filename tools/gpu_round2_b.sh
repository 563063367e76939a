#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
time timeout 900 python bench.py --legs eval_midsize --no-cpu-baseline --no-verify --steps 1 --warmup 0 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d.get('eval_midsize'), indent=1))" | tee gpurun_out/r02_midsize_leg.log
