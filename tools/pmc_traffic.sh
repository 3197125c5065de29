#!/bin/bash
# HBM traffic of a workload by PMC counters (run on the GPU box):  tools/pmc_traffic.sh <tag> <kernel-name-filter> <prof_hot.py args...>
# FETCH_SIZE and WRITE_SIZE in SEPARATE passes with --kernel-trace only (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2 of 4), then
# tools/traffic_summary.py applies the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE reads 1/2 of a wide streaming read)
# and writes profiles/<round>/traffic_<tag>.json (per launch; round = $NSA_PROFILE_ROUND, default r04).
set -e
TAG=$1; FILT=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/traffic_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $GRAFT_REPO_ROOT/tools/prof_hot.py "$@" > $OUT/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $GRAFT_REPO_ROOT/tools/prof_hot.py "$@" > $OUT/write.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/traffic_summary.py $TAG "$FILT" $OUT "$*"
