#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
SHAPES=E python tools/midsize_probe.py auto,perwav,dense 2>&1 | grep -E "^(auto|dense|perwav)"
python bench.py --legs eval_midsize --no-cpu-baseline --no-verify --steps 1 --warmup 0 2>&1 | grep -v amdgpu | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:round(v['ms'],2) for k,v in d['eval_midsize'].items()})"
