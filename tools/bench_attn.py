#!/usr/bin/env python3
"""A/B timing of the selection-attention kernel variants in ONE process (interleaved rounds,
cdna_hip_programming.md rule 24).  Variants are chosen through the NSA_HIP_ATTN_* env switches
read by the launcher at every call.

    python tools/bench_attn.py [--rounds 5] [--iters 10]
"""
import argparse
import itertools
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nsa_vibe_amd as nv  # noqa: E402


def make(B, S, seed=0):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    meta = nv.build_block_meta(S, 32, 16, 64, 16, 512)
    Q = torch.randn(B, S, 2, 6, 64, device="cuda", generator=g).bfloat16()
    K = torch.randn(B, 2, S, 64, device="cuda", generator=g).bfloat16()
    V = torch.randn(B, 2, S, 64, device="cuda", generator=g).bfloat16()
    p = torch.rand(B, S, 2, meta.S_sel, device="cuda", generator=g)
    rg = nv.select_topn_ranges_batched(p, meta, 16, S)
    L = float((rg[..., 1] - rg[..., 0]).clamp_min(0).sum().item())
    return Q, K, V, rg, L


def timeit(fn, iters):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--envs", type=str, default="NSA_HIP_ATTN_MAP=0,1,2;NSA_HIP_ATTN_KDIRECT=0,1")
    args = ap.parse_args()
    axes = []
    for spec in args.envs.split(";"):
        k, vs = spec.split("=")
        axes.append([(k, v) for v in vs.split(",")])
    combos = list(itertools.product(*axes))
    for name, (B, S) in {"S4096_B8": (8, 4096), "S16384_B2": (2, 16384), "S65536_B1": (1, 65536)}.items():
        Q, K, V, rg, L = make(B, S)
        ref = None
        res = {c: [] for c in combos}
        for r in range(args.rounds + 1):
            for c in combos:
                for k, v in c:
                    os.environ[k] = v
                if r == 0:  # warm-up + cross-check of every variant against the first
                    O = nv.selection_attention_hip(Q, K, V, rg).float()
                    if ref is None:
                        ref = O
                    assert (O - ref).abs().max().item() < 2e-2, c
                    continue
                res[c].append(timeit(lambda: nv.selection_attention_hip(Q, K, V, rg), args.iters))
        print(f"== {name}: L_sum={L:.3e} tokens, alg bytes={L * 256 / 1e9:.2f} GB")
        for c in combos:
            med, mn = float(np.median(res[c])), float(np.min(res[c]))
            print(f"   {' '.join(f'{k[13:]}={v}' for k, v in c):28s} median {med:.3f} ms  min {mn:.3f} ms  -> {L * 256 / (med * 1e-3) / 1e12:.2f} TB/s alg, "
                  f"{4 * 6 * L * 64 / (med * 1e-3) / 1e12:.0f} TFLOP/s")
