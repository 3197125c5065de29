#!/usr/bin/env python3
"""Lock-step stress of the layer decode step: the three-launch route (default) against the five-launch route (DECODE_BAND = 0) on twin
caches, bit for bit, every step, over a few thousand steps and several batch sizes (the band branches' splits are merged through LDS
behind one barrier by the workgroup that holds them: a race there would show as a rare mismatch).
    python3 tools/stress_layer_decode.py [steps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nsa_vibe_amd as nv
from nsa_vibe_amd import _lib
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
dev = torch.device("cuda")
bad = 0
for B, S, w, n_sel in [(1, 4000, 512, 16), (2, 700, 512, 13), (3, 9000, 512, 16), (8, 1000, 128, 16), (24, 300, 96, 8), (60, 500, 512, 16)]:
    torch.manual_seed(B)
    m = nv.NSAAttention(768, 12, 2, 64, 64, 32, 16, 64, n_sel, w, selector="batched").to(dev).to(torch.bfloat16).eval()
    x = torch.randn(B, S, 768, device=dev, dtype=torch.bfloat16)
    with torch.no_grad():
        kvs = []
        for band in (0, -1):
            _lib.set_tuning("DECODE_BAND", band)
            kv = m.new_kv(B, S + steps + 8, dev, torch.bfloat16)
            m(x, kv, prefill=True)
            kvs.append(kv)
        mism = 0
        for i in range(steps):
            xt = torch.randn(B, 1, 768, device=dev, dtype=torch.bfloat16)
            ys = []
            for kv, band in zip(kvs, (0, -1)):
                _lib.set_tuning("DECODE_BAND", band)
                ys.append(m(xt, kv, prefill=False)[0])
            if not torch.equal(ys[0], ys[1]):
                mism += 1
                if mism <= 3:
                    print(f"  B={B} step {i}: max diff {(ys[0].float() - ys[1].float()).abs().max().item():.3e}", flush=True)
        print(f"B={B} S={S}: {steps} steps, {mism} mismatching steps", flush=True)
        bad += mism
_lib.set_tuning("DECODE_BAND", -1)
sys.exit(1 if bad else 0)
