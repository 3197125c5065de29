#!/usr/bin/env python3
"""Layer training step (fwd + bwd, bf16, m7c shape): native differentiable ops vs the eager-op path."""
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
for mode in ("native", "eager"):
    os.environ["NSA_HIP_EAGER_TRAIN"] = "1" if mode == "eager" else "0"
    a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    tf, tb, n = [], [], 8
    for i in range(n + 2):
        m.zero_grad(set_to_none=True)
        x.grad = None
        kv = m.new_kv(B, S, "cuda", torch.bfloat16)
        a.record()
        out, _ = m(x, kv, prefill=True)
        b.record()
        out.backward(go)
        c.record()
        torch.cuda.synchronize()
        if i >= 2:
            tf.append(a.elapsed_time(b))
            tb.append(b.elapsed_time(c))
    tf, tb = sorted(tf)[len(tf) // 2], sorted(tb)[len(tb) // 2]
    print(f"{mode}: S={S} B={B} fwd {tf:.3f} ms  bwd {tb:.3f} ms  -> {B * S / (tf + tb) / 1e3:.2f} M tok/s per layer")
