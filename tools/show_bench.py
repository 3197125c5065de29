#!/usr/bin/env python3
"""digest of a bench.py JSON line: python tools/show_bench.py <file>"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = lambda v: round(v, 4) if isinstance(v, float) else v  # noqa: E731
print({k: r(d[k]) for k in ("metric", "value", "ms_per_step", "n_gpus")})
print("workload:", d["config"]["workload"])
rl = d["roofline"]
print("roofline:", {k: r(v) for k, v in rl.items() if not isinstance(v, (dict, str)) or k in ("bound", "unit", "kernel")})
for sub in ("l2", "mfma"):
    if sub in rl:
        if isinstance(rl.get(sub), dict):
            print("  ", sub, {k: r(v) for k, v in rl[sub].items() if k != "note"})
print("stages_ms:", {k: r(v) for k, v in d.get("stages_ms", {}).items() if k != "note"})
print("roofline_scores:", {k: r(v) for k, v in d.get("roofline_scores", {}).items() if k != "note"})
print("decode_roofline:", {k: r(v) for k, v in d.get("decode_roofline", {}).items() if k not in ("workload", "formula")})
print("cpu_baseline:", {k: r(v) for k, v in d.get("cpu_baseline", {}).items() if k != "sample"})
for k, v in d.get("extra", {}).items():
    if isinstance(v, dict):
        print(k, {a: r(b) for a, b in v.items() if not isinstance(b, (dict, str))})
