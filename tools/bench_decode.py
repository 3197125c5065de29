#!/usr/bin/env python3
"""decode step (one native call) of the selected branch: python tools/bench_decode.py [BxS ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import nsa_vibe_amd as nv  # noqa: E402

dev = torch.device("cuda", 0)
for B, S in [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(64, 16384), (64, 32768), (64, 65536), (1, 65536), (8, 16384)]:
    d = bench.decode_bench(nv, B, S, 50, dev)
    print(f"B={B} S={S}: {d['ms_per_step'] * 1e3:7.2f} us/step  {d['tok_per_s']:12.0f} tok/s  "
          f"{(d['gather_bytes'] + d['kcmp_bytes']) / (d['ms_per_step'] * 1e-3) / 1e12:5.2f} TB/s of compulsory reads", flush=True)
    torch.cuda.empty_cache()
