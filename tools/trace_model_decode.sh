#!/bin/bash
# kernel trace of the whole-model decode loop (run on the GPU box): tools/trace_model_decode.sh S B -> per-kernel medians over the last tokens
set -e
S=$1; B=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/mtrace_${S}_${B}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/bench_model.py $S $B 40 > $OUT/log 2>&1
t=$(find $OUT -name "*kernel_trace.csv" | head -1)
tail -2 $OUT/log
python3 - "$t" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "embed_rows" in n][-21:]
agg = collections.OrderedDict()
tot = []
for a, b in zip(idx[:-1], idx[1:]):
    tot.append(int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]))
    for r in rows[a:b]:
        nm = r["Kernel_Name"].split("(")[0][:70]
        agg.setdefault(nm, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
ntok = len(idx) - 1
print(f"{ntok} tokens, median token {sorted(tot)[len(tot)//2]/1e3:.1f} us")
for nm, v in agg.items():
    print(f"{nm:70s} {len(v)/ntok:5.1f} launches/token  median {sorted(v)[len(v)//2]/1e3:6.2f} us  total/token {sum(v)/ntok/1e3:7.1f} us")
PY
