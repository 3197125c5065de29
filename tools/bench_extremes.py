#!/usr/bin/env python3
"""SURVEY 8(d)'s synthetic extremes of the selection attention (bench.attention_extremes) under tuning switches:
python tools/bench_extremes.py "NAME=v,NAME=v" ..."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import nsa_vibe_amd as nv  # noqa: E402
from nsa_vibe_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
for st in sys.argv[1:] or [""]:
    kv = [x.split("=") for x in st.split(",") if x]
    old = {k: _lib.get_tuning(k) for k, _ in kv}
    for k, v in kv:
        _lib.set_tuning(k, int(v))
    r = bench.attention_extremes(nv, dev)
    for name, d in r.items():
        print(f"[{st or 'default':28s}] {name:52s} {d['ms']:8.3f} ms  {d['tflops']:7.1f} TFLOP/s  mfma {d['attn_mfma_frac']:.3f}  qk {d['qk_frac']:.3f}", flush=True)
    for k, v in old.items():
        _lib.set_tuning(k, v)
