#!/usr/bin/env python3
"""Diagnostics for the selection-attention kernels on the GPU box: per-variant / per-head error
patterns on tiny structured cases (run when the parity tests fail)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nsa_vibe_amd as nv  # noqa: E402
from oracle import nsa_oracle as orc  # noqa: E402


def rb(a, dt):
    return torch.from_numpy(a).to(dt).float().numpy()


def case(name, h, D, S_kv, ranges, dt, variant, seed=0, onehot=False):
    rng = np.random.default_rng(seed)
    Q = rng.standard_normal((1, 1, 1, h, D), dtype=np.float32)
    K = rng.standard_normal((1, 1, S_kv, D), dtype=np.float32)
    V = rng.standard_normal((1, 1, S_kv, D), dtype=np.float32)
    if onehot:  # V[k, d] = k + d/100 : shows which key/dv lands where
        V = (np.arange(S_kv)[:, None] + np.arange(D)[None, :] / 100.0).astype(np.float32)[None, None]
        Q = Q * 0  # uniform softmax -> O = mean of selected V rows
    rg = np.array(ranges, np.int32).reshape(1, 1, 1, -1, 2)
    try:
        O = nv.selection_attention_hip(*(torch.from_numpy(x).cuda().to(dt) for x in (Q, K, V)), torch.from_numpy(rg).cuda(), variant=variant)
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        print(f"{name}: EXC {e}")
        return
    ref = orc.sel_attention_masked(rb(Q, dt), rb(K, dt), rb(V, dt), rg)
    got = O.float().cpu().numpy()
    err = np.abs(got - ref)[0, 0, 0]
    print(f"{name:40s} v{variant} {str(dt)[6:]:9s} max={err.max():.3e} per-head={np.array2string(err.max(-1), precision=2)}")
    if err.max() > 5e-2:
        print("   per-dv(max over heads)[:16] =", np.array2string(err.max(0)[:16], precision=2))
        print("   got[0,:8] =", np.array2string(got[0, 0, 0, 0, :8], precision=3))
        print("   ref[0,:8] =", np.array2string(ref[0, 0, 0, 0, :8], precision=3))


if __name__ == "__main__":
    for dt in (torch.float32, torch.bfloat16):
        case("generic 32 keys", 6, 64, 64, [(0, 32)], dt, 1)
        case("generic 100 keys 2 ranges", 6, 64, 200, [(0, 60), (100, 140)], dt, 1)
    dt = torch.bfloat16
    case("mfma onehot 32 keys", 6, 64, 64, [(0, 32)], dt, 2, onehot=True)
    case("mfma onehot 16 keys", 6, 64, 64, [(0, 16)], dt, 2, onehot=True)
    case("mfma 32 keys", 6, 64, 64, [(0, 32)], dt, 2)
    case("mfma 5 keys", 6, 64, 64, [(3, 8)], dt, 2)
    case("mfma 64 keys", 6, 64, 64, [(0, 64)], dt, 2)
    case("mfma 100 keys 2 ranges", 6, 64, 200, [(0, 60), (100, 140)], dt, 2)
    case("mfma h=16", 16, 64, 128, [(0, 128)], dt, 2)
    case("mfma D=128", 4, 128, 128, [(0, 96)], dt, 2)
    case("mfma D=128 onehot", 4, 128, 64, [(0, 32)], dt, 2, onehot=True)
