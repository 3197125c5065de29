#!/bin/bash
# kernel trace of the layer decode loop (run on the GPU box): tools/trace_layer_decode.sh S B  -> gpurun_out/layer_decode_S_B_{stats,trace}.csv
set -e
S=$1; B=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/ltrace_${S}_${B}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/bench_module.py $S $B 72 > $OUT/log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1); cp "$f" $GRAFT_REPO_ROOT/gpurun_out/layer_decode_${S}_${B}_stats.csv
t=$(find $OUT -name "*kernel_trace.csv" | head -1); cp "$t" $GRAFT_REPO_ROOT/gpurun_out/layer_decode_${S}_${B}_trace.csv
tail -2 $OUT/log
python3 - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 40 decode steps: find the repeating pattern by the decode step kernel name
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "decode_step" in n]
idx = idx[-41:]
per = {}
gaps = {}
for a, b in zip(idx[:-1], idx[1:]):
    seq = rows[a:b]
    for k, r in enumerate(seq):
        nm = r["Kernel_Name"].split("(")[0][:60]
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        per.setdefault((k, nm), []).append(d)
        nxt = rows[a + k + 1]
        gaps.setdefault((k, nm), []).append(int(nxt["Start_Timestamp"]) - int(r["End_Timestamp"]))
tot = 0
for key in sorted(per):
    d = sorted(per[key])[len(per[key]) // 2]; g = sorted(gaps[key])[len(gaps[key]) // 2]
    tot += d + g
    print(f"{key[0]:2d} {key[1]:60s} kernel {d/1e3:6.2f} us   gap to next {g/1e3:6.2f} us   (n={len(per[key])})")
print(f"sum of medians: {tot/1e3:.1f} us per step")
PY
