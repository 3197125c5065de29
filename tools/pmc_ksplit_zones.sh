#!/bin/bash
# L2 hit rate / memory-side requests / wave wait shares of the key-split attention launch for several zone layouts (run on the GPU box):
#   tools/pmc_ksplit_zones.sh <S> <B> "T1,T2" ["T1,T2" ...]      -> gpurun_out/pmc_ksplit_zones_S<S>_B<B>.txt
set -e
S=$1; B=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_ksz
RES=$GRAFT_REPO_ROOT/gpurun_out/pmc_ksplit_zones_S${S}_B${B}.txt
: > $RES
cd /tmp && export TMPDIR=/tmp
for z in "$@"; do
  export NSA_HIP_SEL_KSPLIT=1 NSA_HIP_SEL_KSPLIT_T1=${z%,*} NSA_HIP_SEL_KSPLIT_T2=${z#*,}
  rm -rf $OUT; mkdir -p $OUT
  i=0
  for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/prof_hot.py prefill $S $B 3 attn > $OUT/log$i 2>&1 || echo "pass $i failed: $set"
  done
  echo "== zones T1,T2 = $z" >> $RES
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT sel_attn >> $RES
  python3 - $OUT >> $RES <<'PY'
import csv, glob, os, sys
from collections import defaultdict
t = defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "p1", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "sel_attn" in r["Kernel_Name"]:
            t[r["Kernel_Name"].split("(")[0][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in t.items():
    print(f"   duration_us {k}: n={len(v)} mean={sum(v) / len(v):.1f}")
PY
done
cat $RES
