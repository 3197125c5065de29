#!/usr/bin/env python3
"""A/B of the selection forward kernels in ONE process, interleaved rounds (cdna_hip_programming.md rule 24):
    python tools/bench_blocks.py [SxB ...]     select + attend launch, bf16 m7c; modes: query-tile pairs, block form NT = 1 / 2 / 4"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import nsa_vibe_amd as nv  # noqa: E402

dev = torch.device("cuda", 0)
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(4096, 8), (16384, 2), (65536, 1)]
MODES = [("rows1_vadd", 1, 0, 0), ("rows1", 1, 0, 1), ("blk2", -1, 2, 1), ("blk4", -1, 4, 1)]
for S, B in shapes:
    meta, Q, Kc, K, V = bench.make_inputs(nv, B, S, dev, 1234)
    p = nv.selection_scores(Q, Kc, meta, 0.125, causal_skip=True, leave_skipped=True)
    res = {m[0]: [] for m in MODES}
    outs = {}
    for rnd in range(4):
        for name, rows, blocks, rowsum in MODES:
            nv._lib.set_tuning("SEL_ROWS", rows)
            nv._lib.set_tuning("SEL_BLOCKS", blocks)
            nv._lib.set_tuning("SEL_ROWSUM", rowsum)
            fn = lambda: nv.select_and_attend(p, Q, K, V, meta, bench.N_SEL, mode="batched")  # noqa: E731
            res[name].append(bench.time_events(fn, 5, warm=1))
            if rnd == 0:
                outs[name] = fn()
    ref = outs["rows1_vadd"]
    line = f"S={S} B={B}: " + "  ".join(f"{k} {np.median(v) * 1e3:8.1f} us (min {min(v) * 1e3:.1f})" for k, v in res.items())
    errs = {k: (bool(torch.equal(o[0], ref[0])), float((o[1].float() - ref[1].float()).abs().max())) for k, o in outs.items()}
    print(line)
    print("   ranges equal / max|dO| vs rows1_vadd:", errs, flush=True)
    del Q, Kc, K, V, p, outs
    torch.cuda.empty_cache()
