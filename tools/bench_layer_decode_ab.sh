#!/bin/bash
# layer decode step (NSAAttention module, one native call per step): five launches (NSA_HIP_DECODE_BAND=0) against the default three, same box
for sb in "65536 1" "16384 1" "16384 4" "4096 8" "8192 16" "8192 32" "8192 64" "4096 256"; do
  set -- $sb
  a=$(NSA_HIP_DECODE_BAND=0 python3 tools/bench_module.py $1 $2 72 2>/dev/null | tail -1 | sed 's/.*: \([0-9.]*\) us.*/\1/')
  b=$(python3 tools/bench_module.py $1 $2 72 2>/dev/null | tail -1 | sed 's/.*: \([0-9.]*\) us.*/\1/')
  echo "ctx=$1 B=$2: own launch $a us/step -> default $b us/step"
done
