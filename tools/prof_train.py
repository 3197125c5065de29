#!/usr/bin/env python3
"""One training-mode forward+backward of the NSAAttention layer (for kernel-trace breakdowns)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nsa_vibe_amd as nv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
torch.manual_seed(0)
m = nv.NSAAttention(768, 12, 2, 64, 64, 32, 16, 64, 16, 512, selector="batched").cuda().bfloat16().train()
x = torch.randn(B, S, 768, device="cuda", dtype=torch.bfloat16, requires_grad=True)
go = torch.randn(B, S, 768, device="cuda", dtype=torch.bfloat16)
for _ in range(4):
    m.zero_grad(set_to_none=True)
    out, _ = m(x, m.new_kv(B, S, "cuda", torch.bfloat16), prefill=True)
    out.backward(go)
torch.cuda.synchronize()
