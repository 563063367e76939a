#!/bin/bash
# Same-box A/B of the shipped library against lib/libcoldrec_hip_variant.so (make -C coldrec_amd/csrc variant VFLAGS=-D...),
# alternating processes: bash tools/ab_lib.sh <rounds> <command that prints lines containing "of peak">
R=${1:-3}; shift
V=$PWD/coldrec_amd/lib/libcoldrec_hip_variant.so
for r in $(seq $R); do
  "$@" 2>&1 | grep "of peak" | sed "s/^/shipped  /"
  CRH_LIB=$V "$@" 2>&1 | grep "of peak" | sed "s/^/variant  /"
done
