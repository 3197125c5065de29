#!/usr/bin/env python3
"""Setup cost of the selection-attention forward kernels: launch with EMPTY ranges (no tile is ever processed) at the bench shape."""
import os
import sys

import torch

sys.path.insert(0, ".")
import nsa_vibe_amd as nv  # noqa: E402

B, S, G, h, D, n = 8, 4096, 2, 6, 64, 16
Q = torch.randn(B, S, G, h, D, device="cuda").bfloat16()
K = torch.randn(B, G, S, D, device="cuda").bfloat16()
V = torch.randn(B, G, S, D, device="cuda").bfloat16()
rg0 = torch.zeros(B, S, G, n, 2, dtype=torch.int32, device="cuda")
rg1 = rg0.clone()
rg1[..., 0, 1] = 32  # one tile per row
for name, rg in (("empty", rg0), ("one tile", rg1)):
    for mode in ("0", "1"):
        nv._lib.set_tuning("SEL_ROWS", int(mode))
        f = lambda: nv.selection_attention_hip(Q, K, V, rg)  # noqa: E731
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name:9s} mode {mode}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
