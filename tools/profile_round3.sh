#!/bin/bash
# Round-3 profile: rocprofv3 kernel stats of the DEFAULT bench command (all legs) + separate PMC passes for the
# scoring kernel (headline), the fp16 / mask_topk legs and the training legs.
# Usage (on the GPU box, from the repo root): bash tools/profile_round3.sh <tag>
set -u
TAG=${1:-r03_z}
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
T="timeout 900"
$T rocprofv3 --kernel-trace --stats -d "$OUT/stats" -- python3 bench.py --no-cpu-baseline > "$OUT/bench_under_stats.json" 2> "$OUT/stats.err"
EV="--no-cpu-baseline --no-verify --legs none --steps 2 --warmup 1"
$T rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d "$OUT/pmc_sq" -- python3 bench.py $EV > /dev/null 2> "$OUT/pmc_sq.err"
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -- python3 bench.py $EV > /dev/null 2> "$OUT/pmc_fetch.err"
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -- python3 bench.py $EV > /dev/null 2> "$OUT/pmc_write.err"
$T rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -d "$OUT/pmc_l2" -- python3 bench.py $EV > /dev/null 2> "$OUT/pmc_l2.err"
python3 tools/prof_summary.py "${TAG}_eval" "$OUT/stats" "$OUT/pmc_sq" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_l2" > "$OUT/summary_eval.txt" 2>&1
rm -rf "$OUT/stats" "$OUT/pmc_sq" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_l2"   # raw rocpd databases: only the summaries travel back (64 MiB cap)
echo "eval passes done after $SECONDS s"
# fp16 scoring + dense-block ranking legs (the headline runs once beside them)
LG="--no-cpu-baseline --no-verify --steps 1 --warmup 0 --legs eval_f16,mask_topk"
$T rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d "$OUT/pmc_lg_sq" -- python3 bench.py $LG > /dev/null 2> "$OUT/pmc_lg_sq.err"
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_lg_fetch" -- python3 bench.py $LG > /dev/null 2> "$OUT/pmc_lg_fetch.err"
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_lg_write" -- python3 bench.py $LG > /dev/null 2> "$OUT/pmc_lg_write.err"
python3 tools/prof_summary.py "${TAG}_legs" "$OUT/pmc_lg_sq" "$OUT/pmc_lg_sq" "$OUT/pmc_lg_fetch" "$OUT/pmc_lg_write" > "$OUT/summary_legs.txt" 2>&1
rm -rf "$OUT/pmc_lg_sq" "$OUT/pmc_lg_fetch" "$OUT/pmc_lg_write"
echo "leg passes done after $SECONDS s"
# training legs
TR="--train-only --no-cpu-baseline"
$T rocprofv3 --kernel-trace --stats -d "$OUT/tr_stats" -- python3 bench.py $TR > "$OUT/train_under_stats.json" 2> "$OUT/tr_stats.err"
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/tr_fetch" -- python3 bench.py $TR > /dev/null 2> "$OUT/tr_fetch.err"
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/tr_write" -- python3 bench.py $TR > /dev/null 2> "$OUT/tr_write.err"
$T rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT/tr_l2" -- python3 bench.py $TR > /dev/null 2> "$OUT/tr_l2.err"
python3 tools/prof_summary.py "${TAG}_train" "$OUT/tr_stats" "$OUT/tr_fetch" "$OUT/tr_write" "$OUT/tr_l2" > "$OUT/summary_train.txt" 2>&1
rm -rf "$OUT/tr_stats" "$OUT/tr_fetch" "$OUT/tr_write" "$OUT/tr_l2"
echo "train passes done after $SECONDS s"
# S-TRAIN-XL LightGCN: the SpMM where every gathered row is an HBM access (FETCH_SIZE / WRITE_SIZE of spmm_csr_kernel<32>)
XL="--train-xl-lightgcn --steps 2 --warmup 1"
$T rocprofv3 --kernel-trace --stats -d "$OUT/xl_stats" -- python3 bench.py $XL > "$OUT/xl_under_stats.json" 2> "$OUT/xl_stats.err"
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/xl_fetch" -- python3 bench.py $XL > /dev/null 2> "$OUT/xl_fetch.err"
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/xl_write" -- python3 bench.py $XL > /dev/null 2> "$OUT/xl_write.err"
python3 tools/prof_summary.py "${TAG}_xl_lightgcn" "$OUT/xl_stats" "$OUT/xl_fetch" "$OUT/xl_write" > "$OUT/summary_xl.txt" 2>&1
rm -rf "$OUT/xl_stats" "$OUT/xl_fetch" "$OUT/xl_write"
echo "xl passes done after $SECONDS s"
# S-TRAIN-XL BPR-MF step (dense Adam): the traffic figure of the `train_xl` leg on THIS build
MX="--train-xl --steps 4 --warmup 1"
$T rocprofv3 --kernel-trace --stats -d "$OUT/mx_stats" -- python3 bench.py $MX > "$OUT/mx_under_stats.json" 2> "$OUT/mx_stats.err"
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/mx_fetch" -- python3 bench.py $MX > /dev/null 2> "$OUT/mx_fetch.err"
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/mx_write" -- python3 bench.py $MX > /dev/null 2> "$OUT/mx_write.err"
python3 tools/prof_summary.py "${TAG}_train_xl" "$OUT/mx_stats" "$OUT/mx_fetch" "$OUT/mx_write" > "$OUT/summary_mx.txt" 2>&1
rm -rf "$OUT/mx_stats" "$OUT/mx_fetch" "$OUT/mx_write"
echo "train-xl passes done after $SECONDS s"
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* gpurun_out/profiles_$TAG/ 2>/dev/null
cp "$OUT/xl_under_stats.json" gpurun_out/profiles_$TAG/${TAG}_xl_lightgcn_bench_under_rocprof.json
cp "$OUT/bench_under_stats.json" gpurun_out/profiles_$TAG/${TAG}_eval_bench_under_rocprof.json
cp "$OUT/train_under_stats.json" gpurun_out/profiles_$TAG/${TAG}_train_bench_under_rocprof.json
for f in "$OUT"/*.err; do echo "== $f"; grep -v -E "simple_timer|generateRocpd|tool.cpp|amdgpu.ids" "$f" | tail -n 4; done; du -sh gpurun_out; ls -la gpurun_out/profiles_$TAG
