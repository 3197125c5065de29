#!/usr/bin/env python3
"""time of the fused MFMA scorer (bf16, m7c): python tools/bench_scores.py [SxB ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import nsa_vibe_amd as nv  # noqa: E402

dev = torch.device("cuda", 0)
for S, B in [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(4096, 8), (16384, 1), (65536, 1)]:
    meta, Q, Kc, K, V = bench.make_inputs(nv, B, S, dev, 1234)
    fn = lambda: nv.selection_scores(Q, Kc, meta, 0.125, causal_skip=True, leave_skipped=True)  # noqa: E731
    ms = bench.time_events(fn, 8, warm=2)
    fl = 2.0 * B * S * bench.G * bench.H * meta.S_cmp * bench.D
    print(f"S={S} B={B}: scores {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s (one-pass flops)", flush=True)
    del Q, Kc, K, V
    torch.cuda.empty_cache()
