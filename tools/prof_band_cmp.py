#!/usr/bin/env python3
"""Only the compressed-branch band kernel at one shape (for PMC passes): python tools/prof_band_cmp.py [B] [S] [iters]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nsa_vibe_amd as nv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
S = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
g = torch.Generator(device="cuda")
g.manual_seed(0)
mk = lambda *s: torch.randn(*s, device="cuda", generator=g).bfloat16()  # noqa: E731
S_cmp = (S - 32) // 16 + 1
Q, Kc, Vc = mk(B, S, 2, 6, 64), mk(B, 2, S_cmp, 64), mk(B, 2, S_cmp, 64)
for _ in range(iters):
    nv.batched_causal_attention_compressed(Q, Kc, Vc, 32, 16)
torch.cuda.synchronize()
