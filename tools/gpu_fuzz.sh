#!/bin/bash
# randomised parity of the current build: scoring kernels, then training ops (minutes as arguments)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
A=${1:-8}; B=${2:-5}
timeout $((A*60+120)) python tests/fuzz/fuzz_score_topk.py --minutes $A --seed ${3:-41} 2>&1 | grep -v amdgpu | tail -2 | tee gpurun_out/fuzz_final.log
timeout $((B*60+120)) python tests/fuzz/fuzz_train_ops.py --minutes $B --seed ${4:-42} 2>&1 | grep -v amdgpu | tail -2 | tee -a gpurun_out/fuzz_final.log
