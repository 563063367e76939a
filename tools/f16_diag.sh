#!/bin/bash
# fp16 scoring kernels, one box, one run (from the repo root on the GPU box): bash tools/f16_diag.sh <tag> [items]
#   1. the bare MFMA(+LDS) loops of tools/probes/mfma_energy_probe (what the silicon sustains on random operands)
#   2. interleaved A/B of the LDS-DMA kernel against the register-staged ring kernel (tools/route_ab.py), shipped build
#   3. the same A/B on the -DCRH_PROFILE build with parts switched off: CRH_SCORE_ABLATE = 1 (no selection), 2 (every fetch
#      hits one cached tile), 4 (no workgroup barrier), 5 (1 + 4)
#   4. counters of both kernels, separate passes: matrix-pipe occupancy / clock, then wave-cycle split
set -u
TAG=${1:-r05_f16}
ITEMS=${2:-10000000}
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/diag_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
T="timeout 600"
[ -x tools/probes/mfma_energy_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_energy_probe tools/probes/mfma_energy_probe.hip
$T ./tools/probes/mfma_energy_probe > "$OUT/bare_loops.log" 2>&1
echo "bare loops done after $SECONDS s"
$T python3 tools/route_ab.py --dtype f16 --dim 256 --users 131072 --items $ITEMS --arms "default;CRH_SCORE_DMA=0" --rounds 4 2>&1 | grep -v amdgpu.ids > "$OUT/ab_shipped.log"
PLIB=$PWD/coldrec_amd/lib/libcoldrec_hip_profile.so
for abl in 0 1 2 4 5; do
  CRH_LIB=$PLIB CRH_SCORE_ABLATE=$abl $T python3 tools/route_ab.py --dtype f16 --dim 256 --users 131072 --items $ITEMS --arms "default;CRH_SCORE_DMA=0" --rounds 3 2>&1 | grep -v amdgpu.ids | sed "s/^/ablate=$abl /" >> "$OUT/ab_ablations.log"
done
echo "A/B done after $SECONDS s"
$T rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d "$OUT/pmc_a" -- python3 tools/route_ab.py --dtype f16 --dim 256 --users 131072 --items $ITEMS --arms "default;CRH_SCORE_DMA=0" --rounds 1 > /dev/null 2> "$OUT/pmc_a.err"
$T rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS -d "$OUT/pmc_b" -- python3 tools/route_ab.py --dtype f16 --dim 256 --users 131072 --items $ITEMS --arms "default;CRH_SCORE_DMA=0" --rounds 1 > /dev/null 2> "$OUT/pmc_b.err"
python3 tools/prof_summary.py "${TAG}_ab" "$OUT/pmc_a" "$OUT/pmc_a" "$OUT/pmc_b" > "$OUT/summary.txt" 2>&1
rm -rf "$OUT/pmc_a" "$OUT/pmc_b"
mkdir -p gpurun_out/profiles_$TAG; cp profiles/${TAG}_ab* gpurun_out/profiles_$TAG/ 2>/dev/null
cp "$OUT"/*.log gpurun_out/profiles_$TAG/
for f in "$OUT"/*.err; do echo "== $f"; grep -v -E "simple_timer|generateRocpd|tool.cpp|amdgpu.ids" "$f" | tail -n 3; done
cat "$OUT/bare_loops.log" | tail -12; cat "$OUT/ab_shipped.log" "$OUT/ab_ablations.log"
echo "all done after $SECONDS s"
