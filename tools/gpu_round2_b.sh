#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r02_wgsync.log
: > $L
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_score_topk_gpu.py -x -q 2>&1 | grep -v -i -E "rccl|amdgpu|^$" | tail -8 | tee -a $L
python tools/f16_probe.py --dim 256 --reps 2 2>&1 | grep "^f16" | tee -a $L
CRH_SCORE_SYNC_WINDOW=0 python tools/f16_probe.py --dim 256 --reps 2 --tag nosync 2>&1 | grep "^f16" | tee -a $L
python tools/f16_probe.py --dim 128 --reps 2 2>&1 | grep "^f16" | tee -a $L
python tools/f16_probe.py --dim 256 --items 50000000 --reps 1 2>&1 | grep "^f16" | tee -a $L
python tools/score_probe.py --users 131072 --reps 2 --variants pack+mask 2>&1 | grep "kernel ms" | tee -a $L
CRH_SCORE_SYNC_WINDOW=0 python tools/score_probe.py --users 131072 --reps 2 --variants pack+mask 2>&1 | grep "kernel ms" | tee -a $L
rm -rf gpurun_out/pf; timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pf -- python3 tools/f16_probe.py --dim 256 --reps 1 > /dev/null 2>&1
python3 - <<'PY' | tee -a $L
import glob, sqlite3
for f in glob.glob("gpurun_out/pf/**/*_results.db", recursive=True):
    c = sqlite3.connect(f)
    for name, val, n in c.execute("select name, sum(counter_value), count(distinct dispatch_id) from pmc_events where counter_name='FETCH_SIZE' group by name"):
        if "score_topk" in name:
            print("FETCH per launch GB (x2 corrected):", name[:60], val / n * 1024 * 2 / 1e9, "launches", n)
PY
rm -rf gpurun_out/pf
