#!/usr/bin/env python3
"""Golden vectors for the sliding-window / compressed branch attention (g13), from the IMPORTED reference.

    PYTHONPATH=/root/reference python oracle/make_band_goldens.py

  g13_win_*  reference sliding_window_attention(Q,K,V,w)  (nsa/core/attention_kernels.py:146-178) on seeded inputs.
  g13_cmp_*  compressed branch: torch SDPA under the reference's mask `col < num_cmp(t)` (attention_kernels.py:118-123).
             The reference's own CPU function evaluates it per token through SDPA(is_causal=True) with a single query,
             which attends key 0 only (:139-141) -- its output is stored too (`O_ref_quirk`; bit-exact target of the opt-in parity
             mode batched_causal_attention_compressed_first_key_parity) but is NOT the parity target of the kernel;
             the target is the mask the reference states, evaluated with a true softmax.  Round 2: that output is also produced by a
             REFERENCE function -- grouped_selection_attention_masked (:705-772) on K_cmp / V_cmp with the single range
             [0, num_cmp(t)) per row -- and stored as `O_ref_selection_masked`: the kernel is pinned to it.
Only inputs and outputs are written.  The oracle restatement is checked against every vector before saving.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = os.environ.get("NSA_REFERENCE_ROOT", "/root/reference")
if REF not in sys.path:
    sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from nsa.core.attention_kernels import (batched_causal_attention_compressed, grouped_selection_attention_masked,  # noqa: E402
                                        sliding_window_attention)

from oracle import nsa_oracle as orc  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_grad_enabled(False)


def inputs(seed, B, S, G, h, Dk, Dv, S_kv):
    rng = np.random.default_rng(seed)
    Q = rng.standard_normal((B, S, G, h, Dk), dtype=np.float32)
    K = rng.standard_normal((B, G, S_kv, Dk), dtype=np.float32)
    V = rng.standard_normal((B, G, S_kv, Dv), dtype=np.float32)
    return Q, K, V


win_cases = [  # name, seed, B, S, G, h, Dk, Dv, w
    ("a", 1301, 2, 96, 2, 3, 16, 16, 32),
    ("b", 1302, 1, 40, 1, 6, 64, 64, 512),   # S < w: plain causal
    ("c", 1303, 1, 70, 2, 4, 32, 24, 1),     # w = 1: every row attends itself only
    ("d", 1304, 1, 200, 2, 6, 64, 64, 64),   # m7c head geometry
]
for name, seed, B, S, G, h, Dk, Dv, w in win_cases:
    Q, K, V = inputs(seed, B, S, G, h, Dk, Dv, S)
    O = sliding_window_attention(torch.from_numpy(Q), torch.from_numpy(K), torch.from_numpy(V), w).numpy()
    Oo = orc.sliding_window_attention(Q, K, V, w)
    err = float(np.abs(O - Oo).max())
    print(f"g13_win_{name}: S={S} w={w}  max|ref-oracle| = {err:.2e}")
    assert err < 2e-5
    np.savez_compressed(os.path.join(OUT, f"g13_win_{name}.npz"), Q=Q, K=K, V=V, w=np.int32(w), O=O)

cmp_cases = [  # name, seed, B, S, G, h, Dk, Dv, l, d
    ("a", 1311, 2, 100, 2, 3, 16, 16, 32, 16),
    ("b", 1312, 1, 31, 1, 6, 64, 64, 32, 16),   # no compressed token yet: all rows empty
    ("c", 1313, 1, 130, 2, 6, 64, 64, 16, 8),
]
for name, seed, B, S, G, h, Dk, Dv, l, d in cmp_cases:
    S_cmp = 0 if S < l else (S - l) // d + 1
    Q, K, V = inputs(seed, B, S, G, h, Dk, Dv, S_cmp)
    tQ, tK, tV = torch.from_numpy(Q), torch.from_numpy(K), torch.from_numpy(V)
    O_quirk = batched_causal_attention_compressed(tQ, tK, tV, l, d).numpy()
    tpos = torch.arange(S)
    num_cmp = torch.where(tpos + 1 < l, 0, ((tpos + 1 - l) // d) + 1).clamp(max=S_cmp)  # reference :118-121
    if S_cmp > 0:
        allowed = torch.arange(S_cmp).view(1, S_cmp) < num_cmp.view(S, 1)
        safe = allowed.clone()
        safe[~allowed.any(-1), 0] = True
        q = tQ.permute(0, 2, 3, 1, 4).reshape(B, G * h, S, Dk)
        k = tK.unsqueeze(2).expand(B, G, h, S_cmp, Dk).reshape(B, G * h, S_cmp, Dk)
        v = tV.unsqueeze(2).expand(B, G, h, S_cmp, Dv).reshape(B, G * h, S_cmp, Dv)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=torch.zeros(S, S_cmp).masked_fill(~safe, float("-inf")))
        o = o * allowed.any(-1).view(1, 1, S, 1)
        O = o.reshape(B, G, h, S, Dv).permute(0, 3, 1, 2, 4).contiguous().numpy()
    else:
        O = np.zeros((B, S, G, h, Dv), np.float32)
    Oo = orc.batched_causal_attention_compressed(Q, K, V, l, d)
    err = float(np.abs(O - Oo).max())
    # the same attention through the reference's masked selection executor: one range [0, num_cmp(t)) per row over the compressed tokens
    rg = torch.zeros(B, S, G, 1, 2, dtype=torch.int32)
    rg[..., 0, 1] = num_cmp.view(1, S, 1).to(torch.int32)
    if S_cmp > 0:
        O_masked = grouped_selection_attention_masked(tQ, tK, tV, rg).numpy()
    else:
        O_masked = np.zeros((B, S, G, h, Dv), np.float32)
    err_m = float(np.abs(O_masked - Oo).max())
    print(f"g13_cmp_{name}: S={S} S_cmp={S_cmp}  max|sdpa(mask)-oracle| = {err:.2e}  max|reference masked executor - oracle| = {err_m:.2e}"
          f"   max|quirk - true| = {np.abs(O_quirk - O).max():.2e}")
    assert err < 2e-5 and err_m < 2e-5
    np.savez_compressed(os.path.join(OUT, f"g13_cmp_{name}.npz"), Q=Q, K=K, V=V, l=np.int32(l), d=np.int32(d), O=O,
                        O_ref_selection_masked=O_masked, O_ref_quirk=O_quirk, num_cmp=num_cmp.numpy().astype(np.int32))
print("done")
