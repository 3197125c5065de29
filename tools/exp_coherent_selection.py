#!/usr/bin/env python3
"""Upper bound of the row-sharing kernels: group scores that make neighbouring rows select the SAME blocks (what a trained model's
smooth score maps approach), against the random-score bench data.  coherence c: p = c * shared[j] + (1 - c) * random[t, j]."""
import os
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
import nsa_vibe_amd as nv  # noqa: E402

dev = torch.device("cuda")
S, B = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 8)
meta, Q, Kc, K, V = bench.make_inputs(nv, B, S, dev, 1234)
g = torch.Generator(device=dev)
g.manual_seed(3)
shared = torch.rand(1, 1, 2, meta.S_sel, device=dev, generator=g)
rnd = torch.rand(B, S, 2, meta.S_sel, device=dev, generator=g)
for c in (0.0, 0.5, 0.9, 1.0):
    p = (c * shared + (1 - c) * rnd).contiguous()
    line = f"coherence {c:.1f}:"
    for mode, name in (("0", "one-row"), ("1", "pairs"), ("3", "48-slot")):
        nv._lib.set_tuning("SEL_ROWS", int(mode))
        f = lambda: nv.select_and_attend(p, Q, K, V, meta, 16, mode="batched", scale=0.125)  # noqa: E731
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        line += f"  {name} {e0.elapsed_time(e1) / 10 * 1e3:7.1f} us"
    rg = f()[0]
    L = (rg[..., 1] - rg[..., 0]).clamp_min(0).sum(-1).double().mean().item()
    print(line + f"   (mean selected tokens/row {L:.0f})", flush=True)
