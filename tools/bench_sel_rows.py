#!/usr/bin/env python3
"""A/B of the two selection-attention forward kernels (one row per wave vs. query tiles of 48/h rows) over batch and length."""
import os
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
import nsa_vibe_amd as nv  # noqa: E402

dev = torch.device("cuda")
shapes = [(s, b) for s in (512, 1024, 2048, 4096, 16384, 65536) for b in (1, 4, 8, 16) if s * b <= 65536]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for S, B in shapes:
    meta, Q, Kc, K, V = bench.make_inputs(nv, B, S, dev, 1234)
    p = nv.selection_scores(Q, Kc, meta, 0.125, causal_skip=True)
    res = {}
    for mode in ("0", "1", "3"):
        nv._lib.set_tuning("SEL_ROWS", int(mode))
        f = lambda: nv.select_and_attend(p, Q, K, V, meta, 16, mode="batched", scale=0.125)  # noqa: E731
        for _ in range(3):
            f()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                f()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        res[mode] = sorted(ts)[len(ts) // 2]
        out = f()[1]
        res["o" + mode] = out
    err = max((res["o0"].float() - res["o" + m].float()).abs().max().item() for m in ("1", "3"))
    print(f"S={S:6d} B={B:3d} rows(B*S*G)={B * S * 2:7d}  one-row {res['0'] * 1e3:8.1f} us   pairs {res['1'] * 1e3:8.1f} us ({res['1'] / res['0']:.2f})   "
          f"48-slot tiles {res['3'] * 1e3:8.1f} us ({res['3'] / res['0']:.2f})  max|dO| {err:.2e}", flush=True)
