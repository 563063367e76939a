#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
python tools/gemm_f16_probe.py 2>&1 | grep "torch.matmul" | tee gpurun_out/r02_gemm_f16_probe.log
