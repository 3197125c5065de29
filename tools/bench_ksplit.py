#!/usr/bin/env python3
"""block-form attention, plain walk against the key-split form (NSA_HIP_SEL_KSPLIT): python tools/bench_ksplit.py [SxB ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import nsa_vibe_amd as nv  # noqa: E402

dev = torch.device("cuda", 0)
for S, B in [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(65536, 1), (65536, 4), (65536, 8), (32768, 4)]:
    meta, Q, Kc, K, V = bench.make_inputs(nv, B, S, dev, 1234)
    p = nv.selection_scores(Q, Kc, meta, 0.125, causal_skip=True, leave_skipped=True)
    rg = nv.select_topn_ranges_batched(p, meta, bench.N_SEL, S)
    del p
    res = {0: [], 1: []}
    outs = {}
    with torch.no_grad():
        for ks in (0, 1, 0, 1):
            nv._lib.set_tuning("SEL_KSPLIT", ks)
            outs[ks] = nv.selection_attention_hip(Q, K, V, rg)
            res[ks].append(bench.time_events(lambda: nv.selection_attention_hip(Q, K, V, rg), 6, warm=2) * 1e3)
    nv._lib.set_tuning("SEL_KSPLIT", -1)
    d = (outs[0].float() - outs[1].float()).abs().max().item()
    print(f"S={S} B={B}: plain {min(res[0]):9.1f} us | key split + merge {min(res[1]):9.1f} us ({min(res[1]) / min(res[0]):.2f}x)  max|dO| {d:.4f}", flush=True)
    del Q, Kc, K, V, outs
    torch.cuda.empty_cache()
