#!/bin/bash
# prefix length of the seeded route (CRH_SCORE_SEED_ITEMS) on the mid-size shapes with item-range cuts; default = n_items/16 in 4096..16384
cd "$(dirname "$0")/.." || exit 1
for shape in "8192 262144" "65536 131072" "4096 10000000" "16384 1048576"; do
  for p in default 4096 8192 16384 32768 65536; do
    if [ $p = default ]; then unset CRH_SCORE_SEED_ITEMS; else export CRH_SCORE_SEED_ITEMS=$p; fi
    echo "P=$p: $(python3 tools/shape_probe.py $shape 128 5 2>/dev/null | tail -1)"
  done
done
