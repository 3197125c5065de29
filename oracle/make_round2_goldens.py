#!/usr/bin/env python3
"""Round-2 golden vectors from the IMPORTED reference (runs in the build container only).

    PYTHONPATH=/root/reference python oracle/make_round2_goldens.py [g16] [g17] [g18]

  g16_bwd_*          dQ / dK / dV by torch autograd through the reference's semantic executor grouped_selection_attention_masked
                     (nsa/core/attention_kernels.py:705-772), the way nsa/tests/test_selection_backward_reference.py:35-37 takes the
                     reference gradient ((O * dO).sum().backward()); seeded cases incl. the reference test's own shape, multi-range /
                     overlapping / empty rows, and an m7c-shaped row block.  Pins the oracle backward (and through it the HIP
                     backward kernels) to the reference's autograd instead of finite differences.
  g17_head_causal_*  outputs of NSAAttention._sdpa_over_ranges (nsa/core/nsa_attention.py:1779-1855), the gather route of the
                     reference's decode / sequential prefill (the only one left under NSA_FORCE_PARITY=1): head i attends the first
                     i+1 gathered tokens.
  g18_parity_module  the reference NSAAttention module run with NSA_FORCE_PARITY=1 (prefill, both selector modes, + decode steps from
                     an empty cache): pins the forced-parity routing arm of the drop-in module.
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

from oracle import nsa_oracle as orc  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
failed = []


def report(tag, ok, extra=""):
    print(f"  [{'ok' if ok else 'MISMATCH'}] {tag} {extra}")
    if not ok:
        failed.append(tag)


def rand_ranges(rng, B, S, G, n, S_kv, max_len):
    starts = rng.integers(0, max(1, S_kv - max_len), size=(B, S, G, n))
    lens = rng.integers(0, max_len + 1, size=(B, S, G, n))
    return np.stack([starts, np.minimum(starts + lens, S_kv)], axis=-1).astype(np.int32)


def g16():
    from nsa.core.attention_kernels import grouped_selection_attention_masked

    cases = [  # name, seed, B, S, G, h, Dk, Dv, S_kv, n, max_len
        ("ref_test_shape", 0, 1, 2, 1, 2, 8, 8, 24, 2, 7),  # the shape of test_selection_backward_reference.py
        ("a", 1601, 2, 5, 2, 3, 16, 16, 48, 4, 9),
        ("b", 1602, 1, 9, 2, 6, 64, 64, 300, 16, 64),  # m7c row geometry: h=6, D=64, n=16, block-sized ranges
        ("c", 1603, 1, 6, 1, 4, 32, 24, 80, 3, 20),  # Dv != Dk
    ]
    for name, seed, B, S, G, h, Dk, Dv, S_kv, n, max_len in cases:
        rng = np.random.default_rng(seed)
        Q = rng.standard_normal((B, S, G, h, Dk), dtype=np.float32)
        K = rng.standard_normal((B, G, S_kv, Dk), dtype=np.float32)
        V = rng.standard_normal((B, G, S_kv, Dv), dtype=np.float32)
        dO = rng.standard_normal((B, S, G, h, Dv), dtype=np.float32)
        rg = rand_ranges(rng, B, S, G, n, S_kv, max_len)
        if name == "ref_test_shape":
            rg = np.array([[[[[0, 6], [12, 18]]], [[[4, 10], [18, 22]]]]], np.int32)
        else:
            rg[0, 0, 0] = 0  # a row without any token: zero output, zero gradient
            rg[0, 1, 0, 1] = rg[0, 1, 0, 0]  # duplicate range (union semantics)
            rg[0, 2, 0, 0] = (0, S_kv)  # a row that covers everything
            if name == "b":  # block-aligned ranges as the selector emits them, clamp at t+1 on the last
                rg[0, 3, 1] = 0
                rg[0, 3, 1, :3] = ((0, 64), (128, 256), (256, 290))
        q, k, v = (torch.from_numpy(x).requires_grad_(True) for x in (Q, K, V))
        O = grouped_selection_attention_masked(q, k, v, torch.from_numpy(rg))
        (O * torch.from_numpy(dO)).sum().backward()
        gQ, gK, gV = q.grad.numpy(), k.grad.numpy(), v.grad.numpy()
        oQ, oK, oV = orc.sel_attention_masked_bwd(Q, K, V, rg, dO)
        O_orc = orc.sel_attention_masked(Q, K, V, rg)
        err = max(float(np.abs(a - b).max()) for a, b in ((oQ, gQ), (oK, gK), (oV, gV)))
        report(f"g16_bwd_{name}: oracle bwd vs reference autograd", err < 2e-5, f"max err {err:.2e}")
        report(f"g16_bwd_{name}: oracle fwd vs reference", float(np.abs(O_orc - O.detach().numpy()).max()) < 1e-5)
        np.savez_compressed(os.path.join(OUT, f"g16_bwd_{name}.npz"), Q=Q, K=K, V=V, ranges=rg, dO=dO, O=O.detach().numpy(), dQ=gQ, dK=gK,
                            dV=gV)


def g17():
    from nsa.core.nsa_attention import NSAAttention

    torch.set_grad_enabled(False)
    cases = [  # name, seed, B, G, h, Dk, Dv, S_kv, n
        ("a", 1701, 3, 2, 4, 16, 16, 40, 4),
        ("b", 1702, 2, 2, 6, 64, 64, 300, 16),
        ("c", 1703, 2, 1, 3, 32, 24, 64, 3),
    ]
    for name, seed, B, G, h, Dk, Dv, S_kv, n in cases:
        rng = np.random.default_rng(seed)
        mod = NSAAttention(dim=G * h * Dk, n_heads=G * h, n_kv_groups=G, d_k=Dk, d_v=Dv)
        Q = rng.standard_normal((B, G, h, Dk), dtype=np.float32)
        K = rng.standard_normal((B, G, S_kv, Dk), dtype=np.float32)
        V = rng.standard_normal((B, G, S_kv, Dv), dtype=np.float32)
        rg = rand_ranges(rng, B, 1, G, n, S_kv, 6)[:, 0]  # [B,G,n,2]
        rg[0, 0] = 0  # no token at all -> zeros
        rg[1, 0, :] = 0
        rg[1, 0, 0] = (7, 9)  # two tokens only: heads >= 2 see both
        if B > 2:
            rg[2, 0, 0], rg[2, 0, 1] = (20, 26), (3, 5)  # slot order differs from token order: the gather is ascending by token
        O = mod._sdpa_over_ranges(torch.from_numpy(Q), torch.from_numpy(K), torch.from_numpy(V), torch.from_numpy(rg.copy())).numpy()
        O_orc = orc.sel_attention_head_causal_parity(Q[:, None], K, V, rg[:, None])[:, 0]
        err = float(np.abs(O_orc - O).max())
        report(f"g17_head_causal_{name}: oracle vs reference _sdpa_over_ranges", err < 1e-5, f"max err {err:.2e}")
        np.savez_compressed(os.path.join(OUT, f"g17_head_causal_{name}.npz"), Q=Q, K=K, V=V, ranges=rg, O=O)


def g18():
    """reference module under NSA_FORCE_PARITY=1, gate forced onto the selected branch (fc2 bias -1000/1000/-1000 as
    test_equiv_full_coverage.py:72 does; the reference's cmp / win routes carry their own single-query quirk): state dict,
    input, outputs of the prefill in both selector modes and of decode steps from an empty cache (the reference's decode after a
    prefill restarts its compressed-token schedule)"""
    os.environ["NSA_FORCE_PARITY"] = "1"
    from nsa.cache.kv_cache import NSA_KV
    from nsa.core.block_index import build_block_meta
    from nsa.core.nsa_attention import NSAAttention

    torch.set_grad_enabled(False)
    cfg = dict(dim=64, n_heads=4, n_kv_groups=2, d_k=16, d_v=16, l=8, d=4, l_sel=16, n_sel=4, w=16)
    S, n_dec, B = 72, 40, 2

    def empty_kv(attn):
        G = attn.n_kv_groups
        zk, zv = torch.zeros((B, G, 0, attn.d_k)), torch.zeros((B, G, 0, attn.d_v))
        z64 = lambda: torch.zeros((0,), dtype=torch.int64)  # noqa: E731
        return NSA_KV(K_sel=zk.clone(), V_sel=zv.clone(), K_win=zk.clone(), V_win=zv.clone(), K_cmp_raw_seq=zk.clone(),
                      V_cmp_raw_seq=zv.clone(), K_cmp=zk.clone(), V_cmp=zv.clone(), win_ptr=torch.zeros((B, G), dtype=torch.int64),
                      cmp_emit_next=torch.zeros((B, G), dtype=torch.int64),
                      meta=build_block_meta(0, attn.l, attn.d, attn.l_sel, attn.n_sel, attn.w), reads_pred=z64(), reads_act_total=z64(),
                      reads_act_sel=z64(), reads_act_cmp=z64(), reads_act_win=z64())

    out, state = {}, None
    rng = np.random.default_rng(1801)
    x = torch.from_numpy(rng.standard_normal((B, S, cfg["dim"]), dtype=np.float32))
    xd = torch.from_numpy(rng.standard_normal((n_dec, B, 1, cfg["dim"]), dtype=np.float32))
    for mode in ("sequential", "batched"):
        os.environ["NSA_PREFILL_BATCHED"] = "1" if mode == "batched" else "0"
        torch.manual_seed(1801)
        m = NSAAttention(**cfg)
        if state is not None:
            m.load_state_dict(state)
        m.gate.fc2.bias.copy_(torch.tensor([-1000.0, 1000.0, -1000.0]))
        m.eval()
        assert m._env_cache["force_parity"]
        y, _ = m(x, empty_kv(m), prefill=True)
        out[f"out_pre_{mode}"] = y.numpy()
        if state is None:
            state = {k: v.clone() for k, v in m.state_dict().items()}
            kv = empty_kv(m)
            ys = []
            for i in range(n_dec):
                yt, kv = m(xd[i], kv, prefill=False)
                ys.append(yt)
            out["out_dec"] = torch.stack(ys).numpy()
    out.update({"state_" + k.replace(".", "__"): v.numpy() for k, v in state.items()})
    out.update(x_pre=x.numpy(), x_dec=xd.numpy(),
               cfg=np.array([cfg[k] for k in ("dim", "n_heads", "n_kv_groups", "d_k", "d_v", "l", "d", "l_sel", "n_sel", "w")]))
    np.savez_compressed(os.path.join(OUT, "g18_parity_module.npz"), **out)
    print("  wrote g18_parity_module.npz", {k: v.shape for k, v in out.items() if k.startswith("out_")},
          "|seq - bat| max", float(np.abs(out["out_pre_sequential"] - out["out_pre_batched"]).max()))


if __name__ == "__main__":
    which = sys.argv[1:] or ["g16", "g17", "g18"]
    for w_ in which:
        print(w_)
        globals()[w_]()
    if failed:
        print("FAILED:", failed)
        sys.exit(1)
    print("all ok")
