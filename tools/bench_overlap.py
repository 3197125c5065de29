#!/usr/bin/env python3
"""Does the VALU-bound scorer overlap with the fabric-bound select+attend launch?  Hot path of one batch, (a) as two launches
over the whole batch on one stream, (b) per chunk of sequences on two streams: scores(chunk i+1) runs beside attend(chunk i).
python tools/bench_overlap.py [SxBxCHUNK ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import nsa_vibe_amd as nv  # noqa: E402

dev = torch.device("cuda", 0)


def timed(fn, iters=6, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


for S, B, C in [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(65536, 16, 1), (4096, 8, 2)]:
    meta, Q, Kc, K, V = bench.make_inputs(nv, B, S, dev, 1234)
    s_sc, s_at = torch.cuda.Stream(), torch.cuda.Stream()

    def serial():
        return bench.hot_path(nv, meta, Q, Kc, K, V, S)

    def serial_chunked():
        outs = []
        for b0 in range(0, B, C):
            sl = slice(b0, b0 + C)
            p = nv.selection_scores(Q[sl], Kc[sl], meta, causal_skip=True, leave_skipped=True)
            outs.append(nv.select_and_attend(p, Q[sl], K[sl], V[sl], meta, bench.N_SEL, mode="batched"))
        return outs

    def piped():
        cur = torch.cuda.current_stream()
        s_sc.wait_stream(cur)
        s_at.wait_stream(cur)
        outs, ps = [], []
        for b0 in range(0, B, C):
            sl = slice(b0, b0 + C)
            with torch.cuda.stream(s_sc):
                p = nv.selection_scores(Q[sl], Kc[sl], meta, causal_skip=True, leave_skipped=True)
                ev = torch.cuda.Event()
                ev.record(s_sc)
            with torch.cuda.stream(s_at):
                s_at.wait_event(ev)
                outs.append(nv.select_and_attend(p, Q[sl], K[sl], V[sl], meta, bench.N_SEL, mode="batched"))
                p.record_stream(s_at)
            ps.append(p)
        cur.wait_stream(s_sc)
        cur.wait_stream(s_at)
        return outs

    ref = serial()
    got = piped()
    torch.cuda.synchronize()
    O = torch.cat([o[1] for o in got], 0)
    R = torch.cat([o[0] for o in got], 0)
    same = bool((O == ref[1]).all()) and bool((R == ref[0]).all())
    del ref, got, O, R
    t1, t2, t3 = timed(serial), timed(serial_chunked), timed(piped)
    print(f"S={S} B={B} chunk={C}: one stream {t1:8.3f} ms | chunked, one stream {t2:8.3f} ms | two streams {t3:8.3f} ms "
          f"({t1 / t3:.2f}x)  identical={same}", flush=True)
    del Q, Kc, K, V
    torch.cuda.empty_cache()
