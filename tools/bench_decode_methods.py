#!/usr/bin/env python3
"""How to time a 20 us decode step: per-call events vs a back-to-back batch vs a replayed HIP graph of the batch."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import nsa_vibe_amd as nv  # noqa: E402

dev = torch.device("cuda", 0)
for B, S in [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(64, 16384), (1, 16384), (64, 65536)]:
    meta, Q, Kc, K, V = bench.make_inputs(nv, B, S, dev, 7)
    q1 = Q[:, -1:].contiguous()
    del Q
    O = torch.empty(B, 1, bench.G, bench.H, bench.D, device=dev, dtype=torch.bfloat16)
    rg = torch.empty(B, bench.G, bench.N_SEL, 2, device=dev, dtype=torch.int32)
    step = lambda: nv.selection_decode_step(q1, Kc, K, V, meta, bench.N_SEL, S - 1, out=O, ranges_out=rg)  # noqa: E731
    per_call = bench.time_events(step, 50, warm=5) * 1e3
    N = 20

    def batch():
        for _ in range(N):
            step()

    b2b = bench.time_events(batch, 10, warm=2) * 1e3 / N
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        step()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            batch()
    torch.cuda.synchronize()
    gr = bench.time_events(g.replay, 10, warm=2) * 1e3 / N
    print(f"B={B} S={S}: per-call events {per_call:6.2f} us | {N} back to back {b2b:6.2f} us | graph of {N} {gr:6.2f} us", flush=True)
    del q1, Kc, K, V
    torch.cuda.empty_cache()
