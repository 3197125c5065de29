#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic.sh) -> profiles/<round>/traffic_<tag>.json (round: $NSA_PROFILE_ROUND, default r04), per launch of the kernels whose name contains
the filter (comma separated alternatives; every matching kernel of one step is summed, the per-kernel figures are kept too)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

tag, filt, out, what = sys.argv[1], sys.argv[2].split(","), sys.argv[3], sys.argv[4]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collect(sub, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                if row.get("Counter_Name") == counter and any(s in name for s in filt):
                    acc[name.split("(")[0][:80]].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


fetch, nf = collect("fetch", "FETCH_SIZE")
write, _ = collect("write", "WRITE_SIZE")
kernels = {}
total = 0.0
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    b = (2.0 * f + w) * 1024.0
    kernels[k] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "traffic_bytes": b, "launches_seen": nf.get(k, 0)}
    total += b
rec = {"workload": tag, "command": "tools/prof_hot.py " + what, "kernels": kernels, "fetch_correction": 2.0, "traffic_bytes": total,
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, --kernel-trace only; FETCH_SIZE doubled per MI355X_MICROARCH.md "
               "(gfx950 counts the 128-B requests of 16-B/lane streaming reads at 64 B); mean per launch; traffic_bytes = sum over the "
               "kernels of one step"}
rnd = os.environ.get("NSA_PROFILE_ROUND", "r04")
path = os.path.join(root, "profiles", rnd, f"traffic_{tag}.json")
os.makedirs(os.path.dirname(path), exist_ok=True)
json.dump(rec, open(path, "w"), indent=1)
os.makedirs(os.path.join(root, "gpurun_out", "profiles_" + rnd), exist_ok=True)  # (gpurun merges gpurun_out/ back: copy into profiles/ from there)
json.dump(rec, open(os.path.join(root, "gpurun_out", "profiles_" + rnd, f"traffic_{tag}.json"), "w"), indent=1)
print(json.dumps(rec))
