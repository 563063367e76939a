#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest_b.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02_pytest_b.log
tail -4 gpurun_out/r02_pytest_b.log
for k in 1 2; do
timeout 300 python bench.py --train-only --no-cpu-baseline > gpurun_out/r02_train_host.json 2> gpurun_out/r02_train_host.err
python3 - <<'PY'
import json
for f in ("gpurun_out/r02_train_host.json",):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(" | ".join("%s e2e %.3e %.2f ms/ep smp %.2f" % (k, d[k]["value_end_to_end"], d[k]["ms_per_epoch_end_to_end"], d[k]["host_sampler_s_per_epoch"]*1e3) for k in ("train_mf", "train_mf_sgd", "train_lightgcn")))
    except Exception as e:
        print(f, "failed", e); print(open(f.replace(".json", ".err")).read()[-1500:])
PY
done
