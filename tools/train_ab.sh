#!/bin/bash
# same-box A/B of builds of the library on the training legs (BPR-MF step, LightGCN step; medians of 60 hipGraph epochs).
#   VARIANTS: names; "new" = the tree's library, anything else = coldrec_amd/lib/libcoldrec_hip_<name>.so
cd "$(dirname "$0")/.." || exit 1
for rep in 1 2 3; do
  for which in ${VARIANTS:-new}; do
    if [ $which = new ]; then unset CRH_LIB; else export CRH_LIB=$PWD/coldrec_amd/lib/libcoldrec_hip_$which.so; fi
    python3 bench.py --no-cpu-baseline --no-verify --steps 1 --warmup 1 --users-per-step 8192 --items 200000 --legs train 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-5s' % '$which', ' '.join('%s %.5f' % (k, d[k]['ms_per_step']) for k in ('train_mf', 'train_mf_sgd', 'train_lightgcn')))"
  done
done
