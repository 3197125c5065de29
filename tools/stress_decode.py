#!/usr/bin/env python3
"""Stability of the one-launch decode step (team form included): 3000 back-to-back steps per shape, outputs compared bit for bit with the first.
   python tools/stress_decode.py   (on the GPU box)"""
import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd())
import bench, nsa_vibe_amd as nv
dev = torch.device("cuda", 0)
for (B, S) in [(64, 65536), (1, 65536), (64, 16384), (256, 16384), (128, 65536), (200, 40000)]:  # (the last two: one-pass form)
    meta = nv.build_block_meta(S, 32, 16, 64, 16, 512)
    g = torch.Generator(device="cuda"); g.manual_seed(B + S)
    mk = lambda *s: torch.randn(*s, device="cuda", generator=g).bfloat16()
    Q, Kc, K, V = mk(B, 1, 2, 6, 64), mk(B, 2, meta.S_cmp, 64), mk(B, 2, S, 64), mk(B, 2, S, 64)
    ref = nv.selection_decode_step(Q, Kc, K, V, meta, 16, S - 1)
    ref = [r.clone() if torch.is_tensor(r) else r for r in (ref if isinstance(ref, (tuple, list)) else [ref])]
    bad = 0
    for it in range(3000):
        out = nv.selection_decode_step(Q, Kc, K, V, meta, 16, S - 1)
        out = out if isinstance(out, (tuple, list)) else [out]
        if it % 500 == 499:
            torch.cuda.synchronize()
            for a, b in zip(out, ref):
                if torch.is_tensor(a) and not torch.equal(a, b): bad += 1
            print(f"B={B} S={S} it={it+1} mismatches={bad}", flush=True)
    assert bad == 0
print("stress ok")
