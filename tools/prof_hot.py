#!/usr/bin/env python3
"""Focused workloads for rocprofv3 passes (kernel trace or PMC): the hot path of bench.py at one shape, a few launches.

    ... -- python3 tools/prof_hot.py prefill <S> <B> <iters> [attn|scores|select|scsel|all]     (select + attend launch / scorer / both)
    ... -- python3 tools/prof_hot.py bwd <S> <B> <iters>                           (selection attention forward + backward, autograd)
    ... -- python3 tools/prof_hot.py decode  <B> <S_ctx> <iters>                   (nsa_sel_decode_step, the same cache every step: warm)
    ... -- python3 tools/prof_hot.py decode_cold <B> <S_ctx> <iters>               (the steps rotate over bench.py's independent cache sets: cold)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import nsa_vibe_amd as nv  # noqa: E402

mode = sys.argv[1]
dev = torch.device("cuda", 0)
if mode == "prefill":
    S, B, iters = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    stage = sys.argv[5] if len(sys.argv) > 5 else "all"
    meta, Q, Kc, K, V = bench.make_inputs(nv, B, S, dev, 1234)
    p = nv.selection_scores(Q, Kc, meta, causal_skip=True, leave_skipped=True)
    torch.cuda.synchronize()
    for _ in range(iters):
        if stage in ("scores", "all"):
            p = nv.selection_scores(Q, Kc, meta, causal_skip=True, leave_skipped=True)
        if stage in ("attn", "all"):
            nv.select_and_attend(p, Q, K, V, meta, bench.N_SEL, mode="batched")
        if stage == "scsel":  # scores + top-n in one launch (nsa_sel_scores_select)
            nv.selection_scores_select(Q, Kc, meta, bench.N_SEL, mode="batched")
        if stage == "select":
            nv.select_topn_ranges_batched(p, meta, bench.N_SEL, S)
    torch.cuda.synchronize()
elif mode == "bwd":  # selection attention forward + backward on the bench's backward workload (bench.backward_bench)
    S, B, iters = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    meta, Q, Kc, K, V = bench.make_inputs(nv, B, S, dev, 99)
    g = torch.Generator(device=dev)
    g.manual_seed(6)
    rg = nv.select_topn_ranges_batched(torch.rand(B, S, bench.G, meta.S_sel, device=dev, generator=g), meta, bench.N_SEL, S)
    dO = torch.randn(Q.shape, device=dev, generator=g).bfloat16()
    q, k, v = (x.clone().requires_grad_(True) for x in (Q, K, V))
    for _ in range(iters + 1):
        q.grad = k.grad = v.grad = None
        nv.selection_attention_hip(q, k, v, rg).backward(dO)
    torch.cuda.synchronize()
elif mode == "decode_cold":
    B, S, iters = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    n_sets = min(32, -(-bench.COLD_BYTES_BETWEEN_USES // bench.decode_step_bytes(B, S)) + 1)
    meta, sets = bench.decode_cache_sets(nv, B, S, dev, n_sets)
    O = torch.empty(B, 1, bench.G, bench.H, bench.D, device=dev, dtype=torch.bfloat16)
    rg = torch.empty(B, bench.G, bench.N_SEL, 2, device=dev, dtype=torch.int32)
    for i in range(iters + n_sets):
        q1, Kc, K, V = sets[i % n_sets]
        nv.selection_decode_step(q1, Kc, K, V, meta, bench.N_SEL, S - 1, out=O, ranges_out=rg)
    torch.cuda.synchronize()
else:
    B, S, iters = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    meta, Q, Kc, K, V = bench.make_inputs(nv, B, S, dev, 7)
    q1 = Q[:, -1:].contiguous()
    del Q
    O = torch.empty(B, 1, bench.G, bench.H, bench.D, device=dev, dtype=torch.bfloat16)
    rg = torch.empty(B, bench.G, bench.N_SEL, 2, device=dev, dtype=torch.int32)
    for _ in range(iters + 2):
        nv.selection_decode_step(q1, Kc, K, V, meta, bench.N_SEL, S - 1, out=O, ranges_out=rg)
    torch.cuda.synchronize()
