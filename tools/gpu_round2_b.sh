#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
SHAPES=D python tools/midsize_probe.py perwav,dense 2>&1 | grep -E "^(auto|dense|perwav|d[0-9]+)" | tee gpurun_out/r02_dense_boundary_probe.log
