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
Round 4: per sampled row / step the fixture also holds what the REFERENCE selected (its ranges) and how decided that selection was (the gap
between the 13th and 14th ranking keys of its p_grp, as g10 stores it) -- the selector functions the module calls are wrapped for the run,
the way the reference's own tests patch them (nsa/tests/test_causality_asserts.py:54-56) -- so the GPU test can require EVERY row with the
reference's ranges to be within tolerance and every row whose gap exceeds the bf16 score noise to have the reference's ranges.
Second case: g20 = the module at BASELINE configs[0]'s exact shape, the CLI defaults of bench/bench_decode.py:63-72 (dim 256, 8 heads, G 2,
d_k = d_v 32, l 32, d 16, l' 64, n 16, w 512): prefill of 512 tokens in both selector modes, 512 + 32 decode steps from an empty cache
(the bench decodes 32 steps behind a 512-token context; the reference's decode after prefill is not a usable oracle), every output kept:
tests/golden/g20_tiny_bench_module.npz.
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


def build(batched: bool, cfg=None, state_fn=None):
    os.environ["NSA_PREFILL_BATCHED"] = "1" if batched else "0"
    from nsa.core.nsa_attention import NSAAttention

    torch.manual_seed(0)
    attn = NSAAttention(**(cfg or gi.G19_CFG))
    names_shapes = [(k, tuple(v.shape)) for k, v in attn.state_dict().items()]
    state = {k: torch.from_numpy(v) for k, v in (state_fn or gi.g19_state)(names_shapes).items()}
    attn.load_state_dict(state)
    attn.gate.fc2.bias.copy_(torch.tensor([-1000.0, 1000.0, -1000.0]))
    attn.eval()
    return attn, names_shapes


class Recorder:
    """wraps the two selector functions the reference module calls (nsa/core/nsa_attention.py:672, 1106-1108, 1576) and keeps, per token
    position, the ranges it got and the 13th / 14th key gap of the scores it selected from (batch element 0)"""

    def __init__(self, n_top):
        import nsa.core.nsa_attention as mod

        self.mod, self.n_top = mod, n_top
        self.orig = (mod.select_topn_ranges, mod.select_topn_ranges_batched)
        self.ranges, self.gaps = {}, {}
        mod.select_topn_ranges = self.seq
        mod.select_topn_ranges_batched = self.bat

    def close(self):
        self.mod.select_topn_ranges, self.mod.select_topn_ranges_batched = self.orig

    def seq(self, p_grp, meta, n_top, t_token, *a, **k):
        r = self.orig[0](p_grp, meta, n_top, t_token, *a, **k)
        t = int(t_token)
        self.ranges[t] = r[0].numpy().astype(np.int32).copy()  # [G,n,2]
        self.gaps[t] = np.array([gi.topn_gap(p_grp[0, g].numpy(), t, self.n_top) for g in range(p_grp.shape[1])])
        return r

    def bat(self, p_grp_all, meta, n_top, S, *a, **k):
        r = self.orig[1](p_grp_all, meta, n_top, S, *a, **k)
        for t in range(int(S)):
            self.ranges[t] = r[0, t].numpy().astype(np.int32).copy()  # [G,W,2]
            self.gaps[t] = np.array([gi.topn_gap(p_grp_all[0, t, g].numpy(), t, self.n_top) for g in range(p_grp_all.shape[2])])
        return r

    def take(self, rows):
        return np.stack([self.ranges[int(t)] for t in rows]), np.stack([self.gaps[int(t)] for t in rows])


def run_case(cfg, state_fn, inputs, rows_pre, rows_dec, n_dec, B):
    x_pre, x_dec = (torch.from_numpy(a) for a in inputs)
    kw = {}
    for tag, batched in (("seq", False), ("bat", True)):
        attn, names_shapes = build(batched, cfg, state_fn)
        rec = Recorder(cfg["n_sel"])
        t0 = time.time()
        out, _ = attn(x_pre, empty_kv(attn, B), prefill=True)
        rec.close()
        print(f"prefill {tag}: {time.time() - t0:.1f} s, |out| max {float(out.abs().max()):.3f}", flush=True)
        kw[f"out_pre_{tag}"] = out[:, rows_pre].numpy()
        kw[f"ranges_pre_{tag}"], kw[f"gap_pre_{tag}"] = rec.take(rows_pre)
    # decode from an empty cache (sequential selector semantics: decode always uses select_topn_ranges)
    attn, names_shapes = build(False, cfg, state_fn)
    rec = Recorder(cfg["n_sel"])
    kv = empty_kv(attn, B)
    outs = []
    t0 = time.time()
    for i in range(n_dec):
        o, kv = attn(x_dec[i], kv, prefill=False)
        outs.append(o)
        if i % 200 == 0:
            print(f"decode step {i}: {time.time() - t0:.1f} s", flush=True)
    rec.close()
    kw["out_dec"] = torch.stack(outs)[rows_dec].numpy()
    kw["ranges_dec"], kw["gap_dec"] = rec.take(rows_dec)
    kw["names"] = np.array([n for n, _ in names_shapes])
    kw["shapes"] = np.array([list(s) + [0] * (2 - len(s)) for _, s in names_shapes], np.int64)
    kw["cfg"] = np.array([cfg[k] for k in ("dim", "n_heads", "n_kv_groups", "d_k", "d_v", "l", "d", "l_sel", "n_sel", "w")])
    kw["rows_pre"], kw["rows_dec"] = rows_pre, rows_dec
    return kw


if __name__ == "__main__":
    which = sys.argv[1:] or ["g20", "g19"]
    if "g20" in which:
        rp, rd = np.arange(gi.G20_S_PRE, dtype=np.int64), np.arange(gi.G20_N_DEC, dtype=np.int64)
        kw = run_case(gi.G20_CFG, gi.g20_state, gi.g20_inputs(), rp, rd, gi.G20_N_DEC, gi.G20_B)
        path = os.path.join(ROOT, "tests", "golden", "g20_tiny_bench_module.npz")
        np.savez_compressed(path, **kw)
        print("wrote", path, os.path.getsize(path), "bytes")
    if "g19" in which:
        rows_pre, rows_dec = gi.g19_rows()
        kw = run_case(gi.G19_CFG, gi.g19_state, gi.g19_inputs(), rows_pre, rows_dec, gi.G19_N_DEC, gi.G19_B)
        path = os.path.join(ROOT, "tests", "golden", "g19_m7c_module.npz")
        old = dict(np.load(path)) if os.path.exists(path) else {}
        np.savez_compressed(path, **kw)
        print("wrote", path, os.path.getsize(path), "bytes; seq vs bat prefill max diff",
              float(np.abs(kw["out_pre_seq"] - kw["out_pre_bat"]).max()))
        for k in ("out_pre_seq", "out_pre_bat", "out_dec"):
            if k in old:
                print(f"  {k}: identical to the previous fixture: {np.array_equal(old[k], kw[k])}")
