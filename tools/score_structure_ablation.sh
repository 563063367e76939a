#!/bin/bash
# structure ablations of the workgroup scoring kernel (profile build): 1 = no selection, 4 = no workgroup barriers (results invalid)
cd "$(dirname "$0")/.." || exit 1
export CRH_LIB=$PWD/coldrec_amd/lib/libcoldrec_hip_profile.so
for ab in ${ABLATIONS:-0 1 5}; do
  CRH_SCORE_ABLATE=$ab python3 bench.py --no-cpu-baseline --no-verify --steps 2 --warmup 1 --legs eval_f16 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('CRH_SCORE_ABLATE=%-3s fp32 headline frac %.4f (%.1f ms)   fp16 frac %.4f (%.1f ms)' % ('$ab', d['roofline']['frac'], d['roofline']['kernel_ms'], d['eval_f16']['roofline']['frac'], d['eval_f16']['roofline']['kernel_ms']))"
done
