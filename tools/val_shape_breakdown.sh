#!/bin/bash
# per-kernel split of the per-epoch validation ranking (dense route: score block + wave-per-user ranking) at the MovieLens /
# CiteULike shapes, and the score kernel with its stores ablated (profile build, CRH_SCORE_ABLATE=1: results invalid)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=${OUT:-gpurun_out/val_shape}
mkdir -p "$OUT"
for shape in "6040 3706" "5551 16980"; do
  tag=$(echo $shape | tr ' ' x)
  for ab in 0 1; do
    if [ $ab = 1 ]; then export CRH_LIB=$PWD/coldrec_amd/lib/libcoldrec_hip_profile.so CRH_SCORE_ABLATE=1; else unset CRH_LIB CRH_SCORE_ABLATE; fi
    rocprofv3 --kernel-trace -d "$OUT/p_${tag}_$ab" -- python3 tools/shape_probe.py $shape 128 50 > "$OUT/run_${tag}_$ab.log" 2>&1
    echo "== $shape  ablate=$ab: $(tail -1 "$OUT/run_${tag}_$ab.log")"
    python3 tools/kstat.py "$OUT/p_${tag}_$ab" 6
  done
done
