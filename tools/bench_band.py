#!/usr/bin/env python3
"""Band attention kernel timing (sliding window w=512 and compressed l=32,d=16) at the m7c shape, bf16."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nsa_vibe_amd as nv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
G, h, D, w, l, d = 2, 6, 64, 512, 32, 16
g = torch.Generator(device="cuda")
g.manual_seed(0)
mk = lambda *s: torch.randn(*s, device="cuda", generator=g).bfloat16()  # noqa: E731
Q, K, V = mk(B, S, G, h, D), mk(B, G, S, D), mk(B, G, S, D)
S_cmp = (S - l) // d + 1
Kc, Vc = mk(B, G, S_cmp, D), mk(B, G, S_cmp, D)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


t = torch.arange(S)
keys_win = (torch.minimum(t + 1, torch.tensor(w))).sum().item()
keys_cmp = torch.where(t + 1 < l, 0, (t + 1 - l) // d + 1).sum().item()
for name, fn, keys in (("win", lambda: nv.sliding_window_attention(Q, K, V, w), keys_win),
                       ("cmp", lambda: nv.batched_causal_attention_compressed(Q, Kc, Vc, l, d), keys_cmp)):
    ms = timeit(fn)
    fl = 4.0 * B * G * h * D * keys
    print(f"{name}: S={S} B={B}  {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s  ({B * S / ms / 1e3:.2f} M tok/s)")
