#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
D=/tmp/crdata
python -m coldrec_amd.main --make_synthetic movielens --dataset movielens --data_root $D > /dev/null 2>&1
python tools/cli_epoch_breakdown.py $D 2>&1 | grep -v -E "^training:|amdgpu" | tail -14 | tee gpurun_out/r02_cli_breakdown.log
