#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export CRH_BENCH_BACKEND=gloo
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_2rank.json 2> gpurun_out/r02_bench_2rank.err
echo "rc=$? after $SECONDS s"
tail -c 1500 gpurun_out/r02_bench_2rank.json; grep -v -E "amdgpu|RCCL|HIP version|ROCm version|Hostname|Librccl" gpurun_out/r02_bench_2rank.err | tail -5
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29656 bench.py --gpus 2 --steps 1 --warmup 0 --no-cpu-baseline --shard users > gpurun_out/r02_bench_2rank_users.json 2> gpurun_out/r02_bench_2rank_users.err
echo "rc=$? after $SECONDS s"
head -c 600 gpurun_out/r02_bench_2rank_users.json
