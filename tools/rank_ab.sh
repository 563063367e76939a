#!/bin/bash
# same-box A/B of builds of the library on the ranking shapes that go through the dense block (validation + mid-size legs).
#   VARIANTS: names; "new" = the tree's library, anything else = coldrec_amd/lib/libcoldrec_hip_<name>.so
cd "$(dirname "$0")/.." || exit 1
for rep in $(seq 1 ${REPS:-2}); do
  for which in ${VARIANTS:-base new}; do
    if [ $which = new ]; then unset CRH_LIB; else export CRH_LIB=$PWD/coldrec_amd/lib/libcoldrec_hip_$which.so; fi
    python3 bench.py --no-cpu-baseline --no-verify --steps 1 --warmup 1 --users-per-step 8192 --items 200000 --legs ${LEGS:-eval_validation,eval_midsize} 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=d['legs_summary']
print('%-5s' % '$which', ' '.join('%s %.4f' % (k.split('.')[-1], v[0]) for k, v in s.items() if k != 'headline'))"
  done
done
