#!/bin/bash
# per-kernel split of the mid-size ranking shapes (seeded route: dense prefix + fused selection over item cuts + merge)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=${OUT:-gpurun_out/midsize_split}
mkdir -p "$OUT"
for shape in ${SHAPES:-"8192 262144" "65536 131072"}; do
  tag=$(echo $shape | tr ' ' x)
  rocprofv3 --kernel-trace -d "$OUT/p_$tag" -- python3 tools/shape_probe.py $shape 128 10 > "$OUT/run_$tag.log" 2>&1
  echo "== $shape: $(grep MFMA "$OUT/run_$tag.log" | tail -1)"
  python3 tools/kstat.py "$OUT/p_$tag" 8
done
