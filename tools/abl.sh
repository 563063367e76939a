python tools/score_probe.py --variants pack+mask,pack 2>&1 | grep -v amdgpu.ids
python tools/score_probe.py --variants pack+mask,pack --users 131072 --reps 2 2>&1 | grep -v amdgpu.ids
