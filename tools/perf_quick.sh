#!/bin/bash
# quick interleaved perf check of the scoring kernels (fp32 S-EVAL step, fp16 d=256 / d=128)
python tools/score_probe.py --variants pack+mask,pack --users 131072 --reps 2 2>&1 | grep -v amdgpu
python bench.py --dtype f16 --items 20000000 --dim 256 --users 131072 --users-per-step 131072 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('f16 d256', d['ms_per_step'], d['roofline']['frac'])"
python bench.py --dtype f16 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('f16 d128', d['ms_per_step'], d['roofline']['frac'])"
