#!/bin/bash
# kernel-trace stats of a prof_hot.py workload (run on the GPU box): tools/ktrace.sh <tag> <prof_hot args...>  -> gpurun_out/ktrace_<tag>.csv
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/ktrace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/prof_hot.py "$@" > $OUT/log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/ktrace_$TAG.csv
grep -v "at::native\|rocclr\|Cijk" $GRAFT_REPO_ROOT/gpurun_out/ktrace_$TAG.csv | cut -c1-150 | head -12
