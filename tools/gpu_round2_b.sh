#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
SHAPES=C python tools/midsize_probe.py d2048,d4096,d8192,d32768 2>&1 | grep -E "^(auto|d[0-9]+)" | tee gpurun_out/r02_dense_block_probe.log
