#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
L=gpurun_out/r02_slowpath2_ab.log
: > $L
for rep in 1 2; do
for lib in old new; do
  if [ $lib = old ]; then export CRH_LIB=$PWD/coldrec_amd/lib/libcoldrec_hip_old.so; else unset CRH_LIB; fi
  python tools/f16_probe.py --dim 256 --reps 2 --tag $lib 2>&1 | grep "^f16" | tee -a $L
  SHAPES=E python tools/midsize_probe.py perwav 2>&1 | grep -E "^perwav" | sed "s/^/$lib /" | tee -a $L
done
done
