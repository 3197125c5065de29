#!/bin/bash
# time of the fused decode kernel truncated after phase N (measurement aid): tools/ktrace_decode_phases.sh <B> <S>
set -e
for stop in 1 2 3 0; do
  export NSA_HIP_DECODE_STOP=$stop
  echo "stop=$stop"
  $GRAFT_REPO_ROOT/tools/ktrace.sh decode_B$1_S$2_stop$stop decode $1 $2 20 | grep decode_score
done
