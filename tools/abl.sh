run() { python bench.py --dtype f16 --items 20000000 --dim $1 --users 131072 --users-per-step 131072 --steps 2 --warmup 1 --no-cpu-baseline $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])"; }
for wg in 8 1; do for dd in 256 128; do echo "wg=$wg d=$dd"; CRH_SCORE_WG=$wg run $dd; done; done
python tools/score_probe.py --variants pack+mask,pack --users 131072 --reps 2 2>&1 | grep -v amdgpu.ids
