#!/usr/bin/env python3
"""share of the fused selector in the select+attend launch: the same launch against the attention alone on the ranges it produced"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import nsa_vibe_amd as nv  # noqa: E402

dev = torch.device("cuda", 0)
for S, B in [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(4096, 8), (16384, 2), (65536, 1)]:
    meta, Q, Kc, K, V = bench.make_inputs(nv, B, S, dev, 1234)
    p = nv.selection_scores(Q, Kc, meta, 0.125, causal_skip=True, leave_skipped=True)
    rg, O = nv.select_and_attend(p, Q, K, V, meta, bench.N_SEL, mode="batched")
    with torch.no_grad():
        O2 = nv.selection_attention_hip(Q, K, V, rg)
    same = bool((O == O2).all())
    t1 = bench.time_events(lambda: nv.select_and_attend(p, Q, K, V, meta, bench.N_SEL, mode="batched"), 10, warm=2) * 1e3
    with torch.no_grad():
        t2 = bench.time_events(lambda: nv.selection_attention_hip(Q, K, V, rg), 10, warm=2) * 1e3
    t3 = bench.time_events(lambda: nv.select_topn_ranges_batched(p, meta, bench.N_SEL, S), 10, warm=2) * 1e3
    print(f"S={S} B={B}: select+attend {t1:8.1f} us | attend only {t2:8.1f} us | selector kernel alone {t3:8.1f} us | identical O {same}", flush=True)
    del Q, Kc, K, V, p
    torch.cuda.empty_cache()
