#!/usr/bin/env python3
"""Decode step of the hot path, cold (rotating cache sets, bench.decode_bench) and warm, per shape:

    python3 tools/bench_decode_cold.py [B,S ...]        (default: the shapes of bench.py's extras)
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import nsa_vibe_amd as nv  # noqa: E402

dev = torch.device("cuda", 0)
shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [
    (64, 4096), (256, 4096), (64, 16384), (128, 16384), (256, 16384), (64, 65536), (128, 65536), (256, 65536), (512, 65536), (1, 65536)]
print(f"{'B':>4} {'S':>6} {'sets':>4} {'cold us':>8} {'warm us':>8} {'cold GB/s':>9} {'frac':>5} {'warm frac':>9} {'tok/s cold':>11}")
for B, S in shapes:
    d = bench.decode_bench(nv, B, S, 30, dev)
    r = bench.decode_roofline(d, None)
    print(f"{B:4d} {S:6d} {d['cache_sets']:4d} {d['ms_per_step_cold'] * 1e3:8.1f} {d['ms_per_step_warm'] * 1e3:8.1f} {r['achieved']:9.0f} {r['frac']:5.2f} "
          f"{r['warm_same_set_frac_not_an_hbm_figure']:9.2f} {d['tok_per_s']:11.0f}  cold={d['cold']}", flush=True)
    print(json.dumps({k: v for k, v in d.items()}), file=sys.stderr)
