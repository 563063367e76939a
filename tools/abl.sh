export TMPDIR=/tmp
rm -rf gpurun_out/prof_train
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -- python3 bench.py --train-only --no-cpu-baseline > gpurun_out/prof_train.log 2>&1
tail -1 gpurun_out/prof_train.log | cut -c1-200
