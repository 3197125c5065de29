#!/usr/bin/env python3
"""per-stage times of the prefill hot path (bench.stage_times): python tools/bench_stages.py [SxB ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import nsa_vibe_amd as nv  # noqa: E402

dev = torch.device("cuda", 0)
for S, B in [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(65536, 4), (16384, 1), (4096, 8)]:
    meta, Q, Kc, K, V = bench.make_inputs(nv, B, S, dev, 1234)
    sc, se, at, sa, Ls, Lm, nt, scs = bench.stage_times(nv, meta, Q, Kc, K, V, S, 5)
    print(f"S={S} B={B}: scores {sc:.3f} ms  select {se:.3f} ms  attention {at:.3f} ms  select+attention (one call) {sa:.3f} ms  scores+select (one call) {scs:.3f} ms", flush=True)
    del meta, Q, Kc, K, V
    torch.cuda.empty_cache()
