#!/bin/bash
# final pass of a build: round profile, then the default bench line, then the whole GPU test suite
cd "$(dirname "$0")/.." || exit 1
TAG=${1:-r02_b}
mkdir -p gpurun_out
bash tools/profile_round2.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1
tail -5 gpurun_out/${TAG}_profile.log
timeout 1500 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench rc=$? after $SECONDS s"
python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest rc=$? after $SECONDS s"
grep -v -i -E "rccl|amdgpu|^$" gpurun_out/${TAG}_pytest.log | tail -4
