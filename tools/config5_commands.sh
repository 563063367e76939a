#!/bin/bash
# BASELINE.json configs[4] as commands: the DropoutNet item tower generates the 50 M x 256 fp16 item table, then the full-rank
# evaluation -- on one GPU, and in the 8-rank form (item rows sharded, ranks sharing the one GPU over gloo: control flow only)
cd "$(dirname "$0")/.." || exit 1
OUT=${OUT:-gpurun_out/config5}
mkdir -p "$OUT"
F="--dtype f16 --items 50000000 --dim 256 --users 100000 --generator --steps 2 --warmup 1 --no-cpu-baseline --legs none"
( time python3 bench.py --gpus 1 $F ) > "$OUT/n1.json" 2> "$OUT/n1.err"; echo "n1 rc=$?"
( time CRH_BENCH_BACKEND=gloo python3 bench.py --gpus 8 $F --no-train ) > "$OUT/n8.json" 2> "$OUT/n8.err"; echo "n8 rc=$?"
python3 - "$OUT" <<'PY'
import json, sys
o = sys.argv[1]
crc = []
for f in ("n1", "n8"):
    try:
        d = json.loads([l for l in open(o + "/" + f + ".json") if l.startswith("{")][-1])
        crc.append(d["result_crc32"])
        print(f, "crc", d["result_crc32"], "ms/step %.1f" % d["ms_per_step"], "frac %.3f" % d["roofline"]["frac"], d["config"]["parallelism"],
              "generator %.2e items/s" % d["dropoutnet_generator"]["value"], d.get("host_legs_skipped", ""))
    except Exception as e:
        print(f, "no line:", repr(e)); print(open(o + "/" + f + ".err").read()[-1500:])
print("last step's (scores, ids) of the two runs:", "EQUAL" if len(crc) == 2 and crc[0] == crc[1] else "DIFFERENT")
PY
