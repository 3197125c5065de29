#!/usr/bin/env python3
"""few-row projections of the decode step (nsa_linear_small), per shape: python3 tools/bench_linear_small.py [M ...]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nsa_vibe_amd import _lib
from nsa_vibe_amd.selection_scorer import _DT, _stream
dev = torch.device("cuda")
Ms = [int(a) for a in sys.argv[1:]] or [1, 2, 8, 32, 64, 256]
shapes = [("fc1", 3072, 768, 1), ("fc2", 768, 3072, 2), ("out", 768, 768, 2), ("head", 50257, 768, 0)]
L = _lib.lib()
for M in Ms:
    for name, N, K, epi in shapes:
        # rotate over enough weight copies that every call reads its weights from beyond the caches (a 12-block model does)
        ncopy = max(2, int(600e6 // (N * K * 2)))
        Ws = [(torch.randn(N, K, device=dev) / K ** 0.5).bfloat16() for _ in range(ncopy)]
        A = torch.randn(M, K, device=dev).bfloat16()
        res = torch.randn(M, N, device=dev).bfloat16()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        st = _stream(dev)
        def run(i):
            _lib.check(L.nsa_linear_small(A.data_ptr(), Ws[i % ncopy].data_ptr(), out.data_ptr(), M, N, K, _DT[torch.bfloat16], epi,
                                          res.data_ptr() if epi == 2 else None, st), "linear")
        for i in range(5): run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 40
        e0.record()
        for i in range(n): run(i)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        print(f"M={M:4d} {name:5s} N={N:6d} K={K:5d}: {us:7.2f} us/call  ({N * K * 2 / us / 1e3:7.1f} GB/s of weights)", flush=True)
