#!/usr/bin/env python3
"""Per-kernel call count / mean duration from a rocprofv3 rocpd database (`rocprofv3 --kernel-trace -d <dir>`).

    python tools/kstats.py <dir-or-db> [name-substring ...]
"""
import glob
import os
import sqlite3
import sys

src = sys.argv[1]
filt = sys.argv[2:]
dbs = [src] if src.endswith(".db") else glob.glob(os.path.join(src, "**", "*.db"), recursive=True)
for path in dbs:
    db = sqlite3.connect(path)
    q = "select name, count(*), avg(end-start), sum(end-start) from kernels group by name order by sum(end-start) desc"
    for name, calls, avg, tot in db.execute(q):
        if filt and not any(f in name for f in filt):
            continue
        print(f"{name[:90]:90s} calls={calls:5d} avg_us={avg / 1e3:10.1f} total_ms={tot / 1e6:9.3f}")
