export TMPDIR=/tmp
rm -rf gpurun_out/prof_lazy
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_lazy -- python3 bench.py --train-xl --lazy-adam --steps 20 --warmup 4 > gpurun_out/prof_lazy.log 2>&1
tail -1 gpurun_out/prof_lazy.log | cut -c1-200
