#!/usr/bin/env python3
"""m7c_125m TinyLM (12 x LlamaBlockNSA, dim 768, GPT-2 vocabulary; random weights, synthetic tokens) on one GPU:
prefill ms, decode tok/s, training step tok/s.  Attention layers run the native kernels; norms / MLP / embedding / LM head
are PyTorch-ROCm ops.

    python tools/bench_model.py [S] [B] [decode_steps] [--train]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nsa_vibe_amd.llama_block_nsa import TinyLM  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
S = int(args[0]) if len(args) > 0 else 4096
B = int(args[1]) if len(args) > 1 else 1
steps = int(args[2]) if len(args) > 2 else 32
train = "--train" in sys.argv
VOCAB = 50257
torch.manual_seed(0)
dev = torch.device("cuda")
lm = TinyLM(VOCAB, 768, 12, 12, 2, 64, 64, 32, 16, 64, 16, 512, selector="batched").to(dev).to(torch.bfloat16)
nparam = sum(p.numel() for p in lm.parameters())
tok = torch.randint(0, VOCAB, (B, S), device=dev)
if train:
    lm.train()
    opt = torch.optim.SGD(lm.parameters(), lr=1e-4)
    tgt = torch.randint(0, VOCAB, (B, S), device=dev)
    ts = []
    for i in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(lm(tok).float().view(-1, VOCAB), tgt.view(-1))
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    dt = sorted(ts[2:])[len(ts[2:]) // 2]
    print(f"train m7c_125m ({nparam / 1e6:.0f} M params) S={S} B={B}: {dt * 1e3:.1f} ms/step  {B * S / dt / 1e3:.1f} k tok/s  (loss {loss.item():.3f})")
    sys.exit(0)
lm.eval()
with torch.no_grad():
    best = 1e9
    for _ in range(4):
        caches = lm.new_caches(B, S + steps + 16, dev, torch.bfloat16)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        logits = lm.prefill(tok, caches)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"prefill m7c_125m S={S} B={B}: {best * 1e3:.2f} ms  ({B * S / best / 1e6:.2f} M tok/s)")
    nxt = logits.argmax(-1)
    for _ in range(8):
        nxt = lm.decode(nxt, caches, return_next=True)[1]
    dt = 1e9
    for _ in range(max(1, steps // 8)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            nxt = lm.decode(nxt, caches, return_next=True)[1]
        torch.cuda.synchronize()
        dt = min(dt, (time.perf_counter() - t0) / 8)
    print(f"decode m7c_125m ctx={S} B={B}: {dt * 1e3:.3f} ms/token-step  ({B / dt:.0f} tok/s)")
