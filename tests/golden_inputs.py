"""Seeded input recipes shared by oracle/make_goldens.py (fixture generation, build container)
and the parity tests (CPU + GPU box).  numpy PCG64 streams only -- numpy guarantees
`default_rng(seed)` bit-streams are stable across versions/platforms, so inputs too big to
commit (64k K/V) are regenerated instead of stored; fixtures hold the expected outputs.
"""
import numpy as np


def _rng(*key):
    return np.random.default_rng(list(key))


def randn(rng, *shape):
    return rng.standard_normal(shape, dtype=np.float32)


# g5: test_selection_varlen_semantic.py:46-58 shape (B2 S6 G1 h2 D32 S_kv16, span 3)
def g5_inputs():
    r = _rng(5)
    B, S, G, h, Dk, Dv, S_kv = 2, 6, 1, 2, 32, 32, 16
    Q, K, V = randn(r, B, S, G, h, Dk), randn(r, B, G, S_kv, Dk), randn(r, B, G, S_kv, Dv)
    rg = np.zeros((B, S, G, 2, 2), np.int32)
    for t in range(S):
        rg[:, t, :, 0, 0] = max(0, t - 3)
        rg[:, t, :, 0, 1] = min(S_kv, t + 1)
    return Q, K, V, rg


# g6: test_selection_masked_empty_rows.py:6-20
def g6_inputs():
    r = _rng(6)
    Q, K, V = randn(r, 1, 2, 1, 2, 8), randn(r, 1, 1, 4, 8), randn(r, 1, 1, 4, 8)
    return Q, K, V, np.zeros((1, 2, 1, 1, 2), np.int32)


# g7: test_triton_sel_edge_cases.py:23-39 (clamping of out-of-bounds ranges)
def g7_inputs():
    r = _rng(7)
    B, S, G, h, D, S_kv = 1, 1, 1, 2, 16, 16
    Q, K, V = randn(r, B, S, G, h, D), randn(r, B, G, S_kv, D), randn(r, B, G, S_kv, D)
    rg = np.array([[-5, -1], [10, 100]], np.int32).reshape(1, 1, 1, 2, 2)
    return Q, K, V, rg


# g8: test_triton_sel_parity_gpu.py:21-37 multi-span
def g8_inputs():
    r = _rng(8)
    B, S, G, h, D, S_kv = 4, 1, 1, 2, 64, 192
    Q, K, V = randn(r, B, S, G, h, D), randn(r, B, G, S_kv, D), randn(r, B, G, S_kv, D)
    rg = np.zeros((B, S, G, 3, 2), np.int32)
    rg[..., 0, :] = (16, 40)
    rg[..., 1, :] = (64, 96)
    rg[..., 2, :] = (120, 160)
    return Q, K, V, rg


# g8b: overlapping, unsorted, duplicated, empty and inverted ranges -> union semantics
def g8b_inputs():
    r = _rng(88)
    B, S, G, h, D, S_kv = 2, 3, 2, 3, 32, 100
    Q, K, V = randn(r, B, S, G, h, D), randn(r, B, G, S_kv, D), randn(r, B, G, S_kv, D)
    rg = np.zeros((B, S, G, 6, 2), np.int32)
    rg[:, 0, :] = [[50, 70], [10, 30], [20, 40], [10, 30], [0, 0], [90, 100]]
    rg[:, 1, :] = [[0, 100], [5, 6], [99, 100], [0, 0], [0, 0], [0, 0]]
    rg[:, 2, :] = [[30, 31], [31, 32], [95, 250], [60, 60], [-9, 2], [0, 0]]
    return Q, K, V, rg


def g9_scores(S):
    return _rng(9, S).random((1, S, 2, (S + 63) // 64), dtype=np.float32)


def g9_scores_small(S, S_sel):
    return _rng(90, S).random((2, S, 2, S_sel), dtype=np.float32)


# g10: m7c shape (G2 h6 D64, l32 d16 l'64 n16); sampled rows: first 130, every 997th, last 64
def g10_rows(S):
    ts = sorted(set(list(range(0, min(130, S))) + list(range(0, S, 997)) + list(range(max(0, S - 64), S))))
    return np.array(ts, np.int32)


def g10_q_kcmp(S, ts, G=2, h=6, D=64):
    S_cmp = (S - 32) // 16 + 1
    Kc = randn(_rng(10, S, 1), 1, G, S_cmp, D)
    Qr = randn(_rng(10, S, 2), 1, len(ts), G, h, D)
    return Qr, Kc


def g10_kv(S, G=2, D=64):
    r = _rng(10, S, 3)
    return randn(r, 1, G, S, D), randn(r, 1, G, S, D)


def g11_inputs(ci, S, G, h, D, S_cmp):
    r = _rng(11, ci)
    B = 2 if S <= 512 else 1
    return (randn(r, B, S, G, h, D), randn(r, B, G, S_cmp, D), randn(r, B, G, S, D), randn(r, B, G, S, D))


# g19: the reference NSAAttention module at the m7c_125m head geometry (oracle/make_m7c_module_goldens.py).  Weights and inputs are
# regenerated from PCG64 streams on both sides (a 768-wide state dict is 7 MB: too big to commit) and are bf16-representable, so the
# fp32 reference and a bf16 module see identical parameters and activations at the layer input.
G19_CFG = dict(dim=768, n_heads=12, n_kv_groups=2, d_k=64, d_v=64, l=32, d=16, l_sel=64, n_sel=16, w=512)
G19_S_PRE, G19_N_DEC, G19_B = 4096, 2200, 1


def _bf16_round(a):
    """fp32 array rounded to the nearest bf16 value (ties to even), still stored as fp32"""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(a.shape)


def g19_state(names_shapes):
    """names_shapes: [(state-dict key, shape)] in the reference module's own order -> {key: fp32 array (bf16-representable)}"""
    out = {}
    for i, (name, shape) in enumerate(names_shapes):
        shape = tuple(int(x) for x in shape)
        r = _rng(19, i)
        if len(shape) == 2:
            w = randn(r, *shape) / np.float32(np.sqrt(shape[1]))
        else:
            w = randn(r, *shape) * np.float32(0.02)
        out[name] = _bf16_round(w)
    return out


def g19_inputs():
    r = _rng(19, 1000)
    x_pre = _bf16_round(randn(r, G19_B, G19_S_PRE, G19_CFG["dim"]))
    x_dec = _bf16_round(randn(r, G19_N_DEC, G19_B, 1, G19_CFG["dim"]))
    return x_pre, x_dec


def g19_rows():
    """prefill rows / decode steps whose outputs the fixture keeps"""
    pre = np.unique(np.concatenate([np.arange(0, 200, 7), np.arange(200, G19_S_PRE - 96, 29), np.arange(G19_S_PRE - 96, G19_S_PRE)]))
    dec = np.unique(np.concatenate([np.arange(0, 100, 9), np.arange(100, G19_N_DEC - 40, 37), np.arange(G19_N_DEC - 40, G19_N_DEC)]))
    return pre.astype(np.int64), dec.astype(np.int64)


# g20: the reference NSAAttention module at BASELINE configs[0]'s exact shape = the CLI defaults of bench/bench_decode.py:63-72 (dim 256, 8 heads,
# G 2 -> h 4, d_k = d_v 32, l 32, d 16, l' 64, n 16, w 512), context 512 + 32 decode steps (oracle/make_m7c_module_goldens.py, second case).
# Weights / inputs by the g19 recipe (bf16-representable PCG64 streams, regenerated on both sides).
G20_CFG = dict(dim=256, n_heads=8, n_kv_groups=2, d_k=32, d_v=32, l=32, d=16, l_sel=64, n_sel=16, w=512)
G20_S_PRE, G20_N_DEC, G20_B = 512, 512 + 32, 1


def g20_state(names_shapes):
    out = {}
    for i, (name, shape) in enumerate(names_shapes):
        shape = tuple(int(x) for x in shape)
        r = _rng(20, i)
        w = randn(r, *shape) / np.float32(np.sqrt(shape[1])) if len(shape) == 2 else randn(r, *shape) * np.float32(0.02)
        out[name] = _bf16_round(w)
    return out


def g20_inputs():
    r = _rng(20, 1000)
    x_pre = _bf16_round(randn(r, G20_B, G20_S_PRE, G20_CFG["dim"]))
    x_dec = _bf16_round(randn(r, G20_N_DEC, G20_B, 1, G20_CFG["dim"]))
    return x_pre, x_dec


def topn_gap(p_row, t, n_top=16, l_sel=64):
    """gap between the last picked and the first rejected ranking key of one score row at token t (fp32 keys p - idx * 1e-8 as
    nsa/core/selection_scorer.py:182-184 forms them; candidates = complete blocks minus the forced ones 0, t // l', t // l' - 1);
    inf when nothing is rejected"""
    S_sel = p_row.shape[-1]
    nvalid = min(S_sel, (t + 1) // l_sel)
    key = (np.asarray(p_row, np.float32) - np.arange(S_sel, dtype=np.float32) * np.float32(1e-8)).astype(np.float32)
    ok = np.zeros(S_sel, bool)
    ok[:nvalid] = True
    cb = t // l_sel
    for f in (0, cb, max(cb - 1, 0)):
        if f < S_sel:
            ok[f] = False
    k = np.sort(key[ok])[::-1]
    kk = n_top - 3
    return float(k[kk - 1] - k[kk]) if kk >= 1 and k.size > kk else float("inf")
