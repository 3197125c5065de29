#!/usr/bin/env python3
"""Golden vectors for the NSAAttention drop-in (run in the build container; imports the reference).

    python oracle/make_module_goldens.py

Runs the REFERENCE module (nsa.core.nsa_attention.NSAAttention, CPU fp32) with its production selection route
(NSA_FORCE_SEL_MASK=1) and the gate forced onto the selected branch (fc2.bias = [-1000, 1000, -1000], as the
reference's own test_equiv_full_coverage.py:72 does), in both prefill modes, followed by decode steps.
Stores the state dict, inputs and outputs in tests/golden/g12_module.npz.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("NSA_REFERENCE_ROOT", "/root/reference")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True
os.environ["NSA_FORCE_SEL_MASK"] = "1"

import torch  # noqa: E402

torch.set_grad_enabled(False)

CFG = dict(dim=64, n_heads=4, n_kv_groups=2, d_k=16, d_v=16, l=8, d=4, l_sel=16, n_sel=4, w=16)
S_PRE, N_DEC, B = 80, 100, 2


def empty_kv(attn, B, dtype=torch.float32):
    from nsa.cache.kv_cache import NSA_KV
    from nsa.core.block_index import build_block_meta

    G = attn.n_kv_groups
    zk = torch.zeros((B, G, 0, attn.d_k), dtype=dtype)
    zv = torch.zeros((B, G, 0, attn.d_v), dtype=dtype)
    z64 = lambda: torch.zeros((0,), dtype=torch.int64)  # noqa: E731
    return NSA_KV(K_sel=zk.clone(), V_sel=zv.clone(), K_win=zk.clone(), V_win=zv.clone(), K_cmp_raw_seq=zk.clone(),
                  V_cmp_raw_seq=zv.clone(), K_cmp=zk.clone(), V_cmp=zv.clone(), win_ptr=torch.zeros((B, G), dtype=torch.int64),
                  cmp_emit_next=torch.zeros((B, G), dtype=torch.int64),
                  meta=build_block_meta(0, attn.l, attn.d, attn.l_sel, attn.n_sel, attn.w), reads_pred=z64(), reads_act_total=z64(),
                  reads_act_sel=z64(), reads_act_cmp=z64(), reads_act_win=z64())


def run(batched: bool, state=None):
    os.environ["NSA_PREFILL_BATCHED"] = "1" if batched else "0"
    from nsa.core.nsa_attention import NSAAttention

    torch.manual_seed(0)
    attn = NSAAttention(**CFG)
    if state is not None:
        attn.load_state_dict(state)
    attn.gate.fc2.bias.copy_(torch.tensor([-1000.0, 1000.0, -1000.0]))
    attn.eval()
    rng = np.random.default_rng(12)
    x_pre = torch.from_numpy(rng.standard_normal((B, S_PRE, CFG["dim"]), dtype=np.float32))
    x_dec = torch.from_numpy(rng.standard_normal((N_DEC, B, 1, CFG["dim"]), dtype=np.float32))
    kv = empty_kv(attn, B)
    out_pre, kv = attn(x_pre, kv, prefill=True)
    # Decode is run FROM AN EMPTY CACHE (the reference's own prefill-via-decode flow, nsa_attention.py:1507-1519).
    # Decode after a batched/sequential prefill is not a usable oracle: the reference's prefill never fills
    # K_cmp_raw_seq, so its decode restarts the compressed-token emission schedule (and the RoPE positions of the
    # pooled window) from zero (nsa_attention.py:586-604 count only the decode tokens) -- a cache-bookkeeping bug
    # upstream of the selected branch.
    kv = empty_kv(attn, B)
    outs = []
    for i in range(N_DEC):
        o, kv = attn(x_dec[i], kv, prefill=False)
        outs.append(o)
    return attn, x_pre, x_dec, out_pre, torch.stack(outs)


if __name__ == "__main__":
    attn, x_pre, x_dec, o_seq, d_seq = run(False)
    state = {k: v.clone() for k, v in attn.state_dict().items()}
    _, _, _, o_bat, d_bat = run(True, state)
    kw = {"state_" + k.replace(".", "__"): v.numpy() for k, v in state.items()}
    kw.update(x_pre=x_pre.numpy(), x_dec=x_dec.numpy(), out_pre_seq=o_seq.numpy(), out_dec_seq=d_seq.numpy(), out_pre_bat=o_bat.numpy(),
              out_dec_bat=d_bat.numpy(), cfg=np.array([CFG[k] for k in ("dim", "n_heads", "n_kv_groups", "d_k", "d_v", "l", "d", "l_sel", "n_sel", "w")]))
    path = os.path.join(ROOT, "tests", "golden", "g12_module.npz")
    np.savez_compressed(path, **kw)
    print("wrote", path, os.path.getsize(path), "bytes; |out_pre_seq - out_pre_bat| max =", float((o_seq - o_bat).abs().max()),
          "decode seq vs bat", float((d_seq - d_bat).abs().max()))
