#!/usr/bin/env python3
"""Golden vectors for the NSAAttention drop-in AT THE m7c_125m HEAD GEOMETRY (run in the build container; imports the reference).

    python oracle/make_m7c_module_goldens.py

dim 768, 12 heads, G 2 (h = 6), d_k = d_v = 64, l 32, d 16, l' 64, n 16, w 512 (configs/m7c_125m_80g.yaml:1-14): the geometry the
MFMA kernels are built for.  g12 / g18 use dim 64, d_k 16, where only the generic kernels apply (missing 3).
Runs the REFERENCE module (nsa.core.nsa_attention.NSAAttention, CPU fp32) with its production selection route
(NSA_FORCE_SEL_MASK=1) and the gate forced onto the selected branch (fc2.bias = [-1000, 1000, -1000], as the reference's own
test_equiv_full_coverage.py:72 does): prefill of S = 4096 tokens (BASELINE configs[1]; 64 selection blocks, n = 16: a real top-n, not "take everything")
in both selector modes, and 2200 decode steps from an empty cache (decode after prefill of the reference is not a usable oracle, see
make_module_goldens.py).  Weights and inputs come from the PCG64 recipes of tests/golden_inputs.py (bf16-representable values), so
the fixture holds only the outputs of sampled rows: tests/golden/g19_m7c_module.npz.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("NSA_REFERENCE_ROOT", "/root/reference")
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.dont_write_bytecode = True
os.environ["NSA_FORCE_SEL_MASK"] = "1"

import torch  # noqa: E402

import golden_inputs as gi  # noqa: E402
from make_module_goldens import empty_kv  # noqa: E402

torch.set_grad_enabled(False)


def build(batched: bool):
    os.environ["NSA_PREFILL_BATCHED"] = "1" if batched else "0"
    from nsa.core.nsa_attention import NSAAttention

    torch.manual_seed(0)
    attn = NSAAttention(**gi.G19_CFG)
    names_shapes = [(k, tuple(v.shape)) for k, v in attn.state_dict().items()]
    state = {k: torch.from_numpy(v) for k, v in gi.g19_state(names_shapes).items()}
    attn.load_state_dict(state)
    attn.gate.fc2.bias.copy_(torch.tensor([-1000.0, 1000.0, -1000.0]))
    attn.eval()
    return attn, names_shapes


if __name__ == "__main__":
    x_pre, x_dec = (torch.from_numpy(a) for a in gi.g19_inputs())
    rows_pre, rows_dec = gi.g19_rows()
    kw = {}
    for tag, batched in (("seq", False), ("bat", True)):
        attn, names_shapes = build(batched)
        t0 = time.time()
        out, _ = attn(x_pre, empty_kv(attn, gi.G19_B), prefill=True)
        print(f"prefill {tag}: {time.time() - t0:.1f} s, |out| max {float(out.abs().max()):.3f}", flush=True)
        kw[f"out_pre_{tag}"] = out[:, rows_pre].numpy()
    # decode from an empty cache (sequential selector semantics: decode always uses select_topn_ranges)
    attn, names_shapes = build(False)
    kv = empty_kv(attn, gi.G19_B)
    outs = []
    t0 = time.time()
    for i in range(gi.G19_N_DEC):
        o, kv = attn(x_dec[i], kv, prefill=False)
        outs.append(o)
        if i % 200 == 0:
            print(f"decode step {i}: {time.time() - t0:.1f} s", flush=True)
    kw["out_dec"] = torch.stack(outs)[rows_dec].numpy()
    kw["names"] = np.array([n for n, _ in names_shapes])
    kw["shapes"] = np.array([list(s) + [0] * (2 - len(s)) for _, s in names_shapes], np.int64)
    kw["cfg"] = np.array([gi.G19_CFG[k] for k in ("dim", "n_heads", "n_kv_groups", "d_k", "d_v", "l", "d", "l_sel", "n_sel", "w")])
    kw["rows_pre"], kw["rows_dec"] = rows_pre, rows_dec
    path = os.path.join(ROOT, "tests", "golden", "g19_m7c_module.npz")
    np.savez_compressed(path, **kw)
    print("wrote", path, os.path.getsize(path), "bytes; seq vs bat prefill max diff",
          float(np.abs(kw["out_pre_seq"] - kw["out_pre_bat"]).max()))
