#!/bin/bash
# One round's profile (on the GPU box, from the repo root): bash tools/profile_round.sh <tag> [parts]      e.g. r05_z "eval f16"
#   parts: any of  eval f16 train xl mx  (default: all)
#   eval   rocprofv3 kernel stats of the DEFAULT bench command (every leg) + separate PMC passes of the headline kernel
#   f16    the ceiling record of the fp16 scoring kernel (configs[4]): the bare MFMA(+LDS) loops on THIS box, the shipped
#          kernel and the SAME kernel with the selection ablated (-DCRH_PROFILE build), each with
#          SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE (matrix-pipe occupancy and effective clock); + the feature-ladder
#          probe of the kernel's MFMA stream (tools/probes/dma_stream_probe.hip)
#   train  kernel stats + FETCH / WRITE of the train legs;  xl / mx: S-TRAIN-XL LightGCN / BPR-MF steps likewise
set -u
TAG=${1:-r05_a}
PARTS=${2:-"eval f16 train xl mx"}
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT" gpurun_out/profiles_$TAG
T="timeout 1200"
has() { [[ " $PARTS " == *" $1 "* ]]; }
# probes are built HERE, before any rocprofv3 pass: a bench leg never compiles (a compiler started under the profiler's preload
# would be a GPU-initialised process that execs)
[ -x tools/probes/mfma_energy_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_energy_probe tools/probes/mfma_energy_probe.hip
[ -x tools/probes/dma_stream_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -mllvm -amdgpu-mfma-vgpr-form -o tools/probes/dma_stream_probe tools/probes/dma_stream_probe.hip

if has eval; then
$T rocprofv3 --kernel-trace --stats -d "$OUT/stats" -- python3 bench.py --no-cpu-baseline > "$OUT/bench_under_stats.json" 2> "$OUT/stats.err"
EV="--no-cpu-baseline --no-verify --legs eval_d64 --steps 2 --warmup 1"      # the headline's launches + the d=64 leg's (its own kernel rows)
$T rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d "$OUT/pmc_sq" -- python3 bench.py $EV > /dev/null 2> "$OUT/pmc_sq.err"
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -- python3 bench.py $EV > /dev/null 2> "$OUT/pmc_fetch.err"
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -- python3 bench.py $EV > /dev/null 2> "$OUT/pmc_write.err"
python3 tools/prof_summary.py "${TAG}_eval" "$OUT/stats" "$OUT/pmc_sq" "$OUT/pmc_fetch" "$OUT/pmc_write" > "$OUT/summary_eval.txt" 2>&1
rm -rf "$OUT/stats" "$OUT/pmc_sq" "$OUT/pmc_fetch" "$OUT/pmc_write"
cp "$OUT/bench_under_stats.json" gpurun_out/profiles_$TAG/${TAG}_eval_bench_under_rocprof.json
echo "eval passes done after $SECONDS s"
fi

if has f16; then
# (a) bare loops on this box (tools/probes/mfma_energy_probe.hip)
$T ./tools/probes/mfma_energy_probe > "$OUT/f16_bare_loops.log" 2>&1
$T ./tools/probes/dma_stream_probe > "$OUT/f16_stream_ladder.log" 2>&1
# (b) the leg as shipped and with the selection ablated, un-profiled (wall numbers must not come from a profiled pass)
F16="--no-cpu-baseline --no-verify --steps 1 --warmup 0 --legs eval_f16"
PLIB=$PWD/coldrec_amd/lib/libcoldrec_hip_profile.so
[ -f "$PLIB" ] || make -s -C coldrec_amd/csrc profile
$T python3 bench.py $F16 > "$OUT/f16_shipped.json" 2> "$OUT/f16_shipped.err"
CRH_LIB=$PLIB CRH_SCORE_ABLATE=1 $T python3 bench.py $F16 > "$OUT/f16_ablated.json" 2> "$OUT/f16_ablated.err"
CRH_LIB=$PLIB $T python3 bench.py $F16 > "$OUT/f16_profile_build.json" 2> "$OUT/f16_profile_build.err"
# (c) counters, one pass each (no other trace domain beside --pmc)
$T rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d "$OUT/f16_sq" -- python3 bench.py $F16 > /dev/null 2> "$OUT/f16_sq.err"
export CRH_LIB=$PLIB CRH_SCORE_ABLATE=1
$T rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d "$OUT/f16_sq_abl" -- python3 bench.py $F16 > /dev/null 2> "$OUT/f16_sq_abl.err"
unset CRH_LIB CRH_SCORE_ABLATE
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/f16_fetch" -- python3 bench.py $F16 > /dev/null 2> "$OUT/f16_fetch.err"
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/f16_write" -- python3 bench.py $F16 > /dev/null 2> "$OUT/f16_write.err"
python3 tools/prof_summary.py "${TAG}_f16" "$OUT/f16_sq" "$OUT/f16_sq" "$OUT/f16_fetch" "$OUT/f16_write" > "$OUT/summary_f16.txt" 2>&1
python3 tools/prof_summary.py "${TAG}_f16_ablated" "$OUT/f16_sq_abl" "$OUT/f16_sq_abl" > "$OUT/summary_f16_abl.txt" 2>&1
rm -rf "$OUT/f16_sq" "$OUT/f16_sq_abl" "$OUT/f16_fetch" "$OUT/f16_write"
python3 tools/f16_ceiling.py "$TAG" "$OUT" > "$OUT/f16_ceiling.txt" 2>&1
echo "f16 passes done after $SECONDS s"
fi

if has train; then
TR="--train-only --no-cpu-baseline"
$T rocprofv3 --kernel-trace --stats -d "$OUT/tr_stats" -- python3 bench.py $TR > "$OUT/train_under_stats.json" 2> "$OUT/tr_stats.err"
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/tr_fetch" -- python3 bench.py $TR > /dev/null 2> "$OUT/tr_fetch.err"
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/tr_write" -- python3 bench.py $TR > /dev/null 2> "$OUT/tr_write.err"
python3 tools/prof_summary.py "${TAG}_train" "$OUT/tr_stats" "$OUT/tr_fetch" "$OUT/tr_write" > "$OUT/summary_train.txt" 2>&1
rm -rf "$OUT/tr_stats" "$OUT/tr_fetch" "$OUT/tr_write"
cp "$OUT/train_under_stats.json" gpurun_out/profiles_$TAG/${TAG}_train_bench_under_rocprof.json
echo "train passes done after $SECONDS s"
fi

if has xl; then
XL="--train-xl-lightgcn --steps 2 --warmup 1"
$T rocprofv3 --kernel-trace --stats -d "$OUT/xl_stats" -- python3 bench.py $XL > "$OUT/xl_under_stats.json" 2> "$OUT/xl_stats.err"
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/xl_fetch" -- python3 bench.py $XL > /dev/null 2> "$OUT/xl_fetch.err"
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/xl_write" -- python3 bench.py $XL > /dev/null 2> "$OUT/xl_write.err"
$T rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT/xl_l2" -- python3 bench.py $XL > /dev/null 2> "$OUT/xl_l2.err"
python3 tools/prof_summary.py "${TAG}_xl_lightgcn" "$OUT/xl_stats" "$OUT/xl_fetch" "$OUT/xl_write" "$OUT/xl_l2" > "$OUT/summary_xl.txt" 2>&1
rm -rf "$OUT/xl_stats" "$OUT/xl_fetch" "$OUT/xl_write" "$OUT/xl_l2"
cp "$OUT/xl_under_stats.json" gpurun_out/profiles_$TAG/${TAG}_xl_lightgcn_bench_under_rocprof.json
$T python3 tools/xl_spmm_probe.py 2>&1 | grep -v amdgpu > gpurun_out/profiles_$TAG/${TAG}_xl_spmm_halves.log
echo "xl passes done after $SECONDS s"
fi

if has mx; then
MX="--train-xl --steps 4 --warmup 1"
$T rocprofv3 --kernel-trace --stats -d "$OUT/mx_stats" -- python3 bench.py $MX > "$OUT/mx_under_stats.json" 2> "$OUT/mx_stats.err"
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/mx_fetch" -- python3 bench.py $MX > /dev/null 2> "$OUT/mx_fetch.err"
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/mx_write" -- python3 bench.py $MX > /dev/null 2> "$OUT/mx_write.err"
python3 tools/prof_summary.py "${TAG}_train_xl" "$OUT/mx_stats" "$OUT/mx_fetch" "$OUT/mx_write" > "$OUT/summary_mx.txt" 2>&1
rm -rf "$OUT/mx_stats" "$OUT/mx_fetch" "$OUT/mx_write"
echo "train-xl passes done after $SECONDS s"
fi

cp profiles/${TAG}_* gpurun_out/profiles_$TAG/ 2>/dev/null
# (the line above also carries the snapshot's OLD copies of this tag's bench lines: this run's own go on top again)
[ -f "$OUT/bench_under_stats.json" ] && cp "$OUT/bench_under_stats.json" gpurun_out/profiles_$TAG/${TAG}_eval_bench_under_rocprof.json
[ -f "$OUT/train_under_stats.json" ] && cp "$OUT/train_under_stats.json" gpurun_out/profiles_$TAG/${TAG}_train_bench_under_rocprof.json
[ -f "$OUT/xl_under_stats.json" ] && cp "$OUT/xl_under_stats.json" gpurun_out/profiles_$TAG/${TAG}_xl_lightgcn_bench_under_rocprof.json
cp "$OUT"/f16_*.json "$OUT"/f16_*.log "$OUT"/f16_ceiling.txt gpurun_out/profiles_$TAG/ 2>/dev/null
for f in "$OUT"/*.err; do echo "== $f"; grep -v -E "simple_timer|generateRocpd|tool.cpp|amdgpu.ids" "$f" | tail -n 3; done; du -sh gpurun_out; ls -la gpurun_out/profiles_$TAG
