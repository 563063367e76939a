cd /root/repo
for l in libcoldrec_hip.so libcoldrec_hip_h4_96.so libcoldrec_hip_h4_192.so libcoldrec_hip_h6_96.so libcoldrec_hip_h12_96.so libcoldrec_hip.so; do
CRH_LIB=coldrec_amd/lib/$l timeout 300 python bench.py --train-only --no-cpu-baseline > gpurun_out/hv.json 2>gpurun_out/hv.err
python - <<PY
import json
d=json.loads(open("gpurun_out/hv.json").read().strip().splitlines()[-1])
print("$l", " ".join("%s %.2f" % (k, d[k]["ms_per_step"]*1e3) for k in ("train_mf","train_mf_sgd","train_lightgcn")))
PY
done
