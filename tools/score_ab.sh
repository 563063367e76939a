#!/bin/bash
# same-box A/B of builds of the library: headline (fp32 workgroup kernel) + fp16 leg, interleaved, un-profiled.
#   VARIANTS: names; "new" = the tree's library, anything else = coldrec_amd/lib/libcoldrec_hip_<name>.so
cd "$(dirname "$0")/.." || exit 1
for rep in 1 2 3; do
  for which in ${VARIANTS:-base new}; do
    if [ $which = new ]; then unset CRH_LIB; else export CRH_LIB=$PWD/coldrec_amd/lib/libcoldrec_hip_$which.so; fi
    python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --legs eval_f16 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-5s fp32 headline frac %.4f (%.1f ms)   fp16 frac %.4f (%.1f ms)  shard %.4f' % ('$which', d['roofline']['frac'], d['roofline']['kernel_ms'], d['eval_f16']['roofline']['frac'], d['eval_f16']['roofline']['kernel_ms'], d['eval_f16']['shard_8gpu']['frac_of_fp16_mfma_peak']))"
  done
done
