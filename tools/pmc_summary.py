#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel (name substring filter) the mean of every counter.

    python tools/pmc_summary.py <dir> [name-substring ...]
"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
filt = sys.argv[2:] or ["nsa"]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if not any(s in name for s in filt):
                continue
            short = name.split("(")[0][:60]
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):3d} mean={sum(v) / len(v):.4g}")
