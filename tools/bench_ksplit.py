#!/usr/bin/env python3
"""block-form attention, plain walk against the key-split form (NSA_HIP_SEL_KSPLIT; zone thresholds NSA_HIP_SEL_KSPLIT_T1 / _T2 from the
environment or as T1,T2 pairs after --zones): python tools/bench_ksplit.py [SxB ...] [--zones 16384,32768 0,65536 ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import nsa_vibe_amd as nv  # noqa: E402

dev = torch.device("cuda", 0)
args = sys.argv[1:]
zones = [(-1, -1)]
if "--zones" in args:
    i = args.index("--zones")
    zones = [tuple(int(v) for v in a.split(",")) for a in args[i + 1:]]
    args = args[:i]
for S, B in [tuple(int(v) for v in a.split("x")) for a in args] or [(65536, 1), (65536, 4), (65536, 8), (32768, 4)]:
    meta, Q, Kc, K, V = bench.make_inputs(nv, B, S, dev, 1234)
    p = nv.selection_scores(Q, Kc, meta, 0.125, causal_skip=True, leave_skipped=True)
    rg = nv.select_topn_ranges_batched(p, meta, bench.N_SEL, S)
    del p
    forms = [("plain", 0, -1, -1)] + [(f"zones {a},{b}", 1, a, b) for a, b in zones]
    res = {f[0]: [] for f in forms}
    outs = {}
    with torch.no_grad():
        for _ in range(2):
            for name, ks, t1, t2 in forms:
                nv._lib.set_tuning("SEL_KSPLIT", ks), nv._lib.set_tuning("SEL_KSPLIT_T1", t1), nv._lib.set_tuning("SEL_KSPLIT_T2", t2)
                outs[name] = nv.selection_attention_hip(Q, K, V, rg)
                res[name].append(bench.time_events(lambda: nv.selection_attention_hip(Q, K, V, rg), 6, warm=2) * 1e3)
    nv._lib.set_tuning("SEL_KSPLIT", -1), nv._lib.set_tuning("SEL_KSPLIT_T1", -1), nv._lib.set_tuning("SEL_KSPLIT_T2", -1)
    base = min(res["plain"])
    line = f"S={S} B={B}: plain {base:9.1f} us"
    for name, ks, t1, t2 in forms[1:]:
        d = (outs["plain"].float() - outs[name].float()).abs().max().item()
        line += f" | {name} + merge {min(res[name]):9.1f} ({min(res[name]) / base:.2f}x, max|dO| {d:.4f})"
    print(line, flush=True)
    del Q, Kc, K, V, outs
    torch.cuda.empty_cache()
