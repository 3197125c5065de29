#!/usr/bin/env python3
"""Is the 64k selection attention bound per CU or by the shared fabric, and does it run beside the scorer on a partition of the CUs?
(a) each stage alone on a stream restricted to a share of the CUs (hipExtStreamCreateWithCUMask): a per-CU bound doubles at half the CUs,
    a fabric bound does not;
(b) the hot path per chunk of sequences on two masked streams: scores(chunk i+1) on one share of the CUs beside select+attend(chunk i) on the rest.
python tools/exp_cu_partition.py [S B CHUNK]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import nsa_vibe_amd as nv  # noqa: E402

S, B, CH = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (65536, 16, 2)))
dev = torch.device("cuda", 0)
hip = C.CDLL("libamdhip64.so")
NCU = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(bits):
    """bits: iterable of CU indices to enable"""
    words = [0] * ((NCU + 31) // 32)
    for i in bits:
        words[i // 32] |= 1 << (i % 32)
    arr = (C.c_uint32 * len(words))(*words)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(len(words)), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


def timed(fn, stream=None, iters=5, warm=2):
    ts = []
    for it in range(warm + iters):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if stream is None:
            a.record(); fn(); b.record()
        else:
            with torch.cuda.stream(stream):
                a.record(stream); fn(); b.record(stream)
        torch.cuda.synchronize()
        if it >= warm:
            ts.append(a.elapsed_time(b))
    return float(np.median(ts))


meta, Q, Kc, K, V = bench.make_inputs(nv, B, S, dev, 1234)
p_all = nv.selection_scores(Q, Kc, meta, causal_skip=True, leave_skipped=True)
torch.cuda.synchronize()
scores = lambda: nv.selection_scores(Q, Kc, meta, causal_skip=True, leave_skipped=True)  # noqa: E731
attend = lambda: nv.select_and_attend(p_all, Q, K, V, meta, bench.N_SEL, mode="batched")  # noqa: E731
print(f"S={S} B={B}, {NCU} CUs; whole device: scores {timed(scores):.3f} ms, select+attend {timed(attend):.3f} ms", flush=True)
masks = {"low half (bits 0..N/2)": range(NCU // 2), "even bits": range(0, NCU, 2), "bits with (i>>2)&1 == 0": [i for i in range(NCU) if not (i >> 2) & 1],
         "3 of 4 (i%4 != 3)": [i for i in range(NCU) if i % 4 != 3], "1 of 4 (i%4 == 0)": range(0, NCU, 4)}
streams = {}
for name, bits in masks.items():
    st = masked_stream(bits)
    streams[name] = st
    print(f"  mask {name:28s}: scores {timed(scores, st):8.3f} ms, select+attend {timed(attend, st):8.3f} ms", flush=True)

# (b) two partitions side by side
def piped(s_sc, s_at):
    cur = torch.cuda.current_stream()
    s_sc.wait_stream(cur)
    s_at.wait_stream(cur)
    outs, ps = [], []
    for b0 in range(0, B, CH):
        sl = slice(b0, b0 + CH)
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


def serial():
    return bench.hot_path(nv, meta, Q, Kc, K, V, S)


t_serial = timed(serial)
print(f"hot path, one stream, whole device: {t_serial:.3f} ms", flush=True)
plain_a, plain_b = torch.cuda.Stream(), torch.cuda.Stream()
print(f"two unmasked streams, chunk {CH}: {timed(lambda: piped(plain_a, plain_b)):.3f} ms", flush=True)
for name, (sc_bits, at_bits) in {
    "scores on even bits | attend on odd bits": (range(0, NCU, 2), range(1, NCU, 2)),
    "scores on low half | attend on high half": (range(NCU // 2), range(NCU // 2, NCU)),
    "scores on i%4==0 | attend on i%4!=0": (range(0, NCU, 4), [i for i in range(NCU) if i % 4]),
    "scores on i%4!=3 | attend on i%4==3": ([i for i in range(NCU) if i % 4 != 3], range(3, NCU, 4)),
}.items():
    a, b = masked_stream(sc_bits), masked_stream(at_bits)
    print(f"{name}, chunk {CH}: {timed(lambda: piped(a, b)):.3f} ms ({t_serial / timed(lambda: piped(a, b)):.2f}x)", flush=True)
