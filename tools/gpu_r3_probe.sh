#!/bin/bash
# round-3 probe runner: $1 = probe binary under tools/probes, rest = its arguments; log -> gpurun_out/$LOG
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
P=$1; shift
timeout 300 tools/probes/$P "$@" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${LOG:-probe.log}
