#!/bin/bash
# why does the S-TRAIN-XL dense-Adam leg read 0.60 on some runs and 0.71 on others?  A/B on ONE box:
#   a  fresh process, nothing before it            b  after the fp16 leg (power-limited MFMA run) in the same process
#   c  before the fp16 leg in the same process     d  a again (drift of the box over the script)
# each with the in-run copy bandwidth and rocm-smi clocks (bench_legs/train_legs.py train_xl).  bash tools/xl_bimodal_ab.sh <tag>
TAG=${1:-r06}
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/xl_ab_$TAG; mkdir -p "$OUT"
C="--no-cpu-baseline --no-verify --steps 1 --warmup 0"
pick() { python3 - "$1" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        r = json.loads(ln)
        x = r.get("train_xl", r)
        print(json.dumps({"ms_per_step": x.get("ms_per_step"), "spread": x.get("ms_per_step_spread"), "frac": x.get("roofline", {}).get("frac"),
                          "copy_GBps": x.get("roofline", {}).get("copy_GBps_same_run"), "frac_of_copy": x.get("roofline", {}).get("frac_of_copy"),
                          "clocks": x.get("gpu_clocks_before_after")}))
PY
}
timeout 600 python3 bench.py --train-xl --steps 30 --warmup 3 > "$OUT/a.json" 2> "$OUT/a.err"; echo "a fresh:        $(pick $OUT/a.json)"
timeout 900 python3 bench.py $C --legs eval_f16,train_xl > "$OUT/b.json" 2> "$OUT/b.err"; echo "b after fp16:   $(pick $OUT/b.json)"
timeout 900 python3 bench.py $C --legs train_xl,eval_f16 > "$OUT/c.json" 2> "$OUT/c.err"; echo "c before fp16:  $(pick $OUT/c.json)"
timeout 900 python3 bench.py $C --legs eval_d64,eval_f16,mask_topk,train_xl > "$OUT/e.json" 2> "$OUT/e.err"; echo "e default order: $(pick $OUT/e.json)"
timeout 600 python3 bench.py --train-xl --steps 30 --warmup 3 > "$OUT/d.json" 2> "$OUT/d.err"; echo "d fresh again:  $(pick $OUT/d.json)"
rocm-smi -d 0 --showclocks > "$OUT/clocks_idle.txt" 2>&1
