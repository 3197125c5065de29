#!/usr/bin/env python3
"""Focused workload for rocprofv3 counter passes: the hot path of bench.py at one shape, a few launches.

    rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d <dir> -- python3 tools/prof_attn.py [S] [B] [iters] [stage]
stage: attn | scores | select | all
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nsa_vibe_amd as nv  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
stage = sys.argv[4] if len(sys.argv) > 4 else "attn"

g = torch.Generator(device="cuda")
g.manual_seed(0)
meta = nv.build_block_meta(S, 32, 16, 64, 16, 512)
Q = torch.randn(B, S, 2, 6, 64, device="cuda", generator=g).bfloat16()
Kc = torch.randn(B, 2, meta.S_cmp, 64, device="cuda", generator=g).bfloat16()
K = torch.randn(B, 2, S, 64, device="cuda", generator=g).bfloat16()
V = torch.randn(B, 2, S, 64, device="cuda", generator=g).bfloat16()
p = nv.selection_scores(Q, Kc, meta, causal_skip=True)
rg = nv.select_topn_ranges_batched(p, meta, 16, S)
torch.cuda.synchronize()
for _ in range(iters):
    if stage in ("scores", "all"):
        p = nv.selection_scores(Q, Kc, meta, causal_skip=True)
    if stage in ("select", "all"):
        rg = nv.select_topn_ranges_batched(p, meta, 16, S)
    if stage in ("attn", "all"):
        nv.selection_attention_hip(Q, K, V, rg)
torch.cuda.synchronize()
