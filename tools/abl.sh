for w in 0 64 128 256; do echo "== sync window $w"; CRH_SCORE_SYNC_WINDOW=$w python tools/score_probe.py --variants pack+mask --users 131072 --reps 2 2>&1 | grep -v amdgpu.ids | tail -1; done
