#!/bin/bash
# CLI epoch time (what main.py prints) + where an epoch goes, MovieLens shape, BPR-MF d=128
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
python -m coldrec_amd.main --make_synthetic movielens --dataset movielens --data_root /tmp/crdata > /dev/null 2>&1
python tools/cli_epoch_breakdown.py /tmp/crdata 2>&1 | grep -v amdgpu | tail -22 | tee gpurun_out/${1:-r03_cli_breakdown}.log
python -m coldrec_amd.main --dataset movielens --data_root /tmp/crdata --model MF --emb_size 128 --epochs 30 --early_stop 1000 --save_emb false --result_dir /tmp/crres 2>&1 | grep -E "Time:" | tee -a gpurun_out/${1:-r03_cli_breakdown}.log
