#!/bin/bash
# same-box A/B of the fp16 workgroup kernel's variants at the S-DN shape (interleaved, un-profiled):
#   CRH_SCORE_WG unset  8 waves x 64 users per workgroup (shipped)     CRH_SCORE_WG=5  4 waves x 128 users, one wave per SIMD
cd "$(dirname "$0")/.." || exit 1
F16="--no-cpu-baseline --no-verify --steps 1 --warmup 0 --legs eval_f16"
for rep in 1 2; do
  for wg in "" 5; do
    CRH_SCORE_WG=$wg python3 bench.py $F16 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['eval_f16']
print('CRH_SCORE_WG=%-2s  frac %.4f  kernel %.1f ms  shard frac %.4f  verified %s' % ('$wg' or '-', d['roofline']['frac'], d['roofline']['kernel_ms'], d['shard_8gpu']['frac_of_fp16_mfma_peak'], d['verified_users']))"
  done
done
CRH_SCORE_WG=5 python3 -m pytest tests/test_score_topk_gpu.py -x -q -m gpu -k "f16 or fp16 or half" 2>&1 | tail -2
