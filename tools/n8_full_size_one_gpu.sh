#!/bin/bash
# The driver's N = 8 command shape at the FULL headline size (1 M users x 10 M items, 131 072-user steps), on ONE GPU: the
# eight ranks share it and exchange over gloo (CRH_BENCH_BACKEND=gloo).  Says nothing about xGMI; it shows that the
# self-launched 8-rank control flow (1.25 M-item shards, one all-gather of 8 x 131 072 x 40 words, canonical merge, rank 0's
# oracle check against the rebuilt whole table, the data-parallel train leg) completes at full size and that the last step's
# (scores, ids) are byte for byte those of the one-rank run (result_crc32).
cd "$(dirname "$0")/.." || exit 1
OUT=${OUT:-gpurun_out/n8_full}
mkdir -p "$OUT"
python3 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --legs none > "$OUT/n1.json" 2> "$OUT/n1.err"
echo "n1 rc=$?"
( time CRH_BENCH_BACKEND=gloo python3 bench.py --gpus 8 --steps 2 --warmup 1 --no-cpu-baseline --legs none ) > "$OUT/n8.json" 2> "$OUT/n8.err"
echo "n8 rc=$?"
python3 - "$OUT" <<'PY'
import json, sys
o = sys.argv[1]
rd = lambda f: json.loads([l for l in open(f) if l.startswith("{")][-1])
a, b = rd(o + "/n1.json"), rd(o + "/n8.json")
print("N=1: crc %d, %d users oracle-checked, %.1f ms per step" % (a["result_crc32"], a.get("verified_users", -1), a["ms_per_step"]))
print("N=8 (one GPU, gloo): crc %d, %d users oracle-checked, %.1f ms per step, parallelism %s, train_mf_dp %s" % (
    b["result_crc32"], b.get("verified_users", -1), b["ms_per_step"], b["config"]["parallelism"],
    json.dumps({k: v for k, v in b.get("train_mf_dp", {}).items() if not isinstance(v, (dict, list))})[:400]))
print("EQUAL" if a["result_crc32"] == b["result_crc32"] else "DIFFERENT")
PY
tail -3 "$OUT/n8.err"
