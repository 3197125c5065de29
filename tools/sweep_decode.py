#!/usr/bin/env python3
"""decode step under tuning switches: python tools/sweep_decode.py "NAME=v,NAME=v" ... -- one line per setting and shape
(shapes: B=64@16k, 256@16k, 64@64k, 128@16k, 1@64k unless SHAPES=BxS,BxS is set)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import nsa_vibe_amd as nv  # noqa: E402
from nsa_vibe_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
shapes = [tuple(int(v) for v in a.split("x")) for a in os.environ.get("SHAPES", "64x16384,256x16384,64x65536,128x16384,1x65536").split(",")]
settings = sys.argv[1:] or [""]
defaults = {}
for B, S in shapes:
    for st in settings:
        kv = [x.split("=") for x in st.split(",") if x]
        for k, v in kv:
            defaults.setdefault(k, _lib.get_tuning(k))
            _lib.set_tuning(k, int(v))
        d = bench.decode_bench(nv, B, S, 40, dev)
        alg = d["gather_bytes"] + d["kcmp_bytes"]
        print(f"B={B:4d} S={S:6d} [{st or 'default':40s}] {d['ms_per_step'] * 1e3:7.2f} us/step  single {d['ms_per_step_single_call'] * 1e3:7.2f}  "
              f"{alg / (d['ms_per_step'] * 1e-3) / 8e12:6.3f} of HBM peak", flush=True)
        for k, _ in kv:
            _lib.set_tuning(k, defaults[k])
        torch.cuda.empty_cache()
