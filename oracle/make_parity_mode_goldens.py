#!/usr/bin/env python3
"""Golden vectors for the opt-in parity mode of the selection executor (g15), from the IMPORTED reference.

    PYTHONPATH=/root/reference python oracle/make_parity_mode_goldens.py

  g15_first_key_*  outputs of the reference's default executors grouped_selection_attention_packed (attention_kernels.py:273-388)
                   and grouped_selection_attention (:181-226) on seeded inputs with multi-range, unsorted-slot and empty rows.  Both
                   call SDPA(is_causal=True) with a single query, i.e. they return V at the first gathered key.
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

from nsa.core.attention_kernels import grouped_selection_attention, grouped_selection_attention_packed  # noqa: E402

from oracle import nsa_oracle as orc  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_grad_enabled(False)

cases = [  # name, seed, B, S, G, h, Dk, Dv, S_kv, n
    ("a", 1501, 2, 9, 2, 3, 16, 16, 40, 4),
    ("b", 1502, 1, 12, 1, 6, 64, 64, 200, 16),
    ("c", 1503, 1, 7, 2, 4, 32, 24, 64, 3),
]
for name, seed, B, S, G, h, Dk, Dv, S_kv, n in cases:
    rng = np.random.default_rng(seed)
    Q = rng.standard_normal((B, S, G, h, Dk), dtype=np.float32)
    K = rng.standard_normal((B, G, S_kv, Dk), dtype=np.float32)
    V = rng.standard_normal((B, G, S_kv, Dv), dtype=np.float32)
    starts = rng.integers(0, S_kv - 8, size=(B, S, G, n))
    lens = rng.integers(0, 9, size=(B, S, G, n))  # zero-length ranges are skipped, so the first LIVE slot varies
    ranges = np.stack([starts, starts + lens], axis=-1).astype(np.int32)
    ranges[0, 0, 0] = 0  # a row without any range -> zeros
    ranges[0, 1, 0, 0] = (5, 5)  # leading empty slot
    ranges[0, 2, 0, :, 0], ranges[0, 2, 0, :, 1] = 7, 3  # inverted ranges only -> zeros
    tq, tk, tv, tr = (torch.from_numpy(x) for x in (Q, K, V, ranges))
    O_packed = grouped_selection_attention_packed(tq, tk, tv, tr).numpy()
    O_gather = grouped_selection_attention(tq, tk, tv, tr).numpy()
    assert np.array_equal(O_packed, O_gather), "the reference's two quirk executors disagree"
    O_orc = orc.sel_attention_first_key_parity(Q, V, ranges)
    assert np.array_equal(O_orc, O_packed), f"oracle restatement differs from the reference on case {name}"
    np.savez_compressed(os.path.join(OUT, f"g15_first_key_{name}.npz"), Q=Q, K=K, V=V, ranges=ranges, O=O_packed)
    print(name, "ok", O_packed.shape, "zero rows:", int((np.abs(O_packed).sum(axis=(3, 4)) == 0).sum()))
