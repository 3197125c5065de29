#!/bin/bash
# kernel trace of the layer prefill (run on the GPU box): tools/trace_layer_prefill.sh S B -> kernels of the LAST prefill, in order
set -e
S=$1; B=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/ptrace_${S}_${B}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/bench_module.py $S $B 16 > $OUT/log 2>&1
t=$(find $OUT -name "*kernel_trace.csv" | head -1)
head -1 $OUT/log
python3 - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the last prefill = from the last rope_cache_append_kernel launch back to the GEMM before it, up to the first decode kernel after it
ia = max(i for i, n in enumerate(names) if "rope_cache_append_kernel" in n)
ib = next(i for i in range(ia, len(rows)) if "qkv_rope_append" in names[i] or "linear_mfma" in names[i])
seq = rows[ia - 3: ib]
t0 = int(seq[0]["Start_Timestamp"])
for r in seq:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(a - t0)/1e3:9.1f} us  +{(b - a)/1e3:8.1f} us  {r['Kernel_Name'].split('(')[0][:90]}")
print(f"span {(int(seq[-1]['End_Timestamp']) - t0)/1e3:.1f} us")
PY
