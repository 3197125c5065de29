#!/usr/bin/env python3
"""Timing of the selection-attention backward (MFMA vs generic) at the training shape (m7c, bf16)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nsa_vibe_amd as nv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
g = torch.Generator(device="cuda")
g.manual_seed(0)
meta = nv.build_block_meta(S, 32, 16, 64, 16, 512)
mk = lambda *s: torch.randn(*s, device="cuda", generator=g).bfloat16()  # noqa: E731
Q, K, V, dO = mk(B, S, 2, 6, 64), mk(B, 2, S, 64), mk(B, 2, S, 64), mk(B, S, 2, 6, 64)
rg = nv.select_topn_ranges_batched(torch.rand(B, S, 2, meta.S_sel, device="cuda", generator=g), meta, 16, S)


def run(variant, iters):
    q, k, v = (x.clone().requires_grad_(True) for x in (Q, K, V))
    a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    tf = tb = 0.0
    for i in range(iters + 2):
        q.grad = k.grad = v.grad = None
        a.record()
        O = nv.selection_attention_hip(q, k, v, rg, variant=variant)
        b.record()
        O.backward(dO)
        c.record()
        torch.cuda.synchronize()
        if i >= 2:
            tf += a.elapsed_time(b)
            tb += b.elapsed_time(c)
    return tf / iters, tb / iters, (q.grad, k.grad, v.grad)


f2, b2, g2 = run(2, 10)
print(f"S={S} B={B}  MFMA: fwd {f2:.3f} ms  bwd {b2:.3f} ms")
if S * B <= 4096 * 8:
    f1, b1, g1 = run(1, 2)
    print(f"         generic: fwd {f1:.3f} ms  bwd {b1:.3f} ms")
    for x, y, n in zip(g2, g1, ("dQ", "dK", "dV")):
        print(f"   {n}: max|mfma-generic| = {(x.float() - y.float()).abs().max().item():.3e} (max |ref| {y.float().abs().max().item():.2f})")
