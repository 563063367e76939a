#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_train_gpu.py tests/test_e2e_gpu.py -x -q -k "spmm or lgcn or lightgcn or fsgnn or plugin" > gpurun_out/r02_spmm_b.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02_spmm_b.log
tail -5 gpurun_out/r02_spmm_b.log
python tools/lgcn_sweep.py CRH_SPMM_DESC=0,1 CRH_SPMM_WAVE=0,256,512,1024 2>&1 | tee gpurun_out/r02_lgcn_sweep.log
python tools/spmm_shape_probe.py 2>&1 | tee -a gpurun_out/r02_lgcn_sweep.log
