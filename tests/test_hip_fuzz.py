"""Seeded randomized sweeps of the three kernel families against the oracle: shapes, dtypes, strides, degenerate ranges, ties.
Every case is small enough for the CPU oracle; the point is coverage of the dispatch (MFMA / generic, split-KV, XCD-aware and
plain mappings, head counts that do not divide 16, D in {32, 64, 128}) rather than size."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# NSA_FUZZ_SEEDS / NSA_FUZZ_OFFSET widen the sweep for a one-off stress run (default: 24 seeds from 0, 12 for the backward)
N_SEEDS = int(os.environ.get("NSA_FUZZ_SEEDS", "24"))
SEED0 = int(os.environ.get("NSA_FUZZ_OFFSET", "0"))
SEEDS = range(SEED0, SEED0 + N_SEEDS)
SEEDS_BWD = range(SEED0, SEED0 + max(1, N_SEEDS // 2))

TOL = {torch.float32: 1e-3, torch.bfloat16: 1e-2, torch.float16: 1e-2}


@pytest.fixture(scope="module")
def nv():
    import nsa_vibe_amd

    assert torch.cuda.is_available()
    return nsa_vibe_amd


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.to(dtype) if dtype is not None else t


def rounded(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).float().numpy()


def _bound(ref, dtype):
    return TOL[dtype] * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("seed", SEEDS)
def test_fuzz_selection_attention(nv, orc, seed):
    rng = np.random.default_rng(7000 + seed)
    dtype = [torch.bfloat16, torch.float16, torch.float32][seed % 3]
    B, G = int(rng.integers(1, 4)), int(rng.integers(1, 5))
    h = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 12, 16]))
    D = int(rng.choice([32, 64, 64, 64, 128]))
    S, S_kv, n = int(rng.integers(1, 90)), int(rng.integers(1, 700)), int(rng.integers(1, 20))
    Q = rng.standard_normal((B, S, G, h, D), dtype=np.float32)
    S_alloc = S_kv + int(rng.integers(0, 40))  # K/V are views of a larger cache
    K = rng.standard_normal((B, G, S_alloc, D), dtype=np.float32)
    V = rng.standard_normal((B, G, S_alloc, D), dtype=np.float32)
    st = rng.integers(-20, S_kv + 20, size=(B, S, G, n))
    ln = rng.integers(-30, 130, size=(B, S, G, n))
    rg = np.stack([st, st + ln], axis=-1).astype(np.int32)
    rg[rng.random((B, S, G)) < 0.1] = 0  # empty rows
    Kd, Vd = dev(K, dtype)[:, :, :S_kv], dev(V, dtype)[:, :, :S_kv]
    O, lse = nv.selection_attention_hip(dev(Q, dtype), Kd, Vd, dev(rg), return_lse=True)
    ref, ref_lse = orc.sel_attention_masked(rounded(Q, dtype), rounded(K[:, :, :S_kv], dtype), rounded(V[:, :, :S_kv], dtype), rg, return_lse=True)
    got = O.float().cpu().numpy()
    assert np.isfinite(got).all() and np.abs(got - ref).max() <= _bound(ref, dtype)
    assert np.array_equal(np.isfinite(lse.cpu().numpy()), np.isfinite(ref_lse))


@pytest.mark.parametrize("seed", SEEDS)
def test_fuzz_band_attention(nv, orc, seed):
    from nsa_vibe_amd.band_attention import band_attention_hip

    rng = np.random.default_rng(8000 + seed)
    dtype = [torch.bfloat16, torch.float16, torch.float32][seed % 3]
    B, G = int(rng.integers(1, 4)), int(rng.integers(1, 5))
    h = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 12, 16]))
    D = int(rng.choice([32, 64, 64, 64, 128]))
    S = int(rng.integers(1, 400))
    if seed % 2 == 0:  # sliding window, possibly a decode-style offset
        band = dict(t0=int(rng.integers(0, 50)) if seed % 4 == 0 else 0, a=0, dd=1, c=0, w=int(rng.integers(0, 200)))
        S_kv = band["t0"] + S + int(rng.integers(0, 10)) - int(rng.integers(0, 5))
    else:  # compressed schedule
        l = int(rng.choice([8, 16, 32]))
        d = int(rng.choice([d_ for d_ in (4, 8, 16, 32) if l % d_ == 0]))
        band = dict(t0=0, a=l, dd=d, c=1, w=2 ** 30)
        S_kv = max(0, (S - l) // d + 1 + int(rng.integers(-2, 3)))
    S_kv = max(S_kv, 0)
    Q = rng.standard_normal((B, S, G, h, D), dtype=np.float32)
    K = rng.standard_normal((B, G, max(S_kv, 1), D), dtype=np.float32)[:, :, :S_kv]
    V = rng.standard_normal((B, G, max(S_kv, 1), D), dtype=np.float32)[:, :, :S_kv]
    O, lse = band_attention_hip(dev(Q, dtype), dev(K, dtype), dev(V, dtype), return_lse=True, **band)
    ref, ref_lse = orc.band_attention(rounded(Q, dtype), rounded(K, dtype), rounded(V, dtype), return_lse=True, **band)
    got = O.float().cpu().numpy()
    assert np.isfinite(got).all() and np.abs(got - ref).max() <= _bound(ref, dtype)
    assert np.array_equal(np.isfinite(lse.cpu().numpy()), np.isfinite(ref_lse))


@pytest.mark.parametrize("seed", SEEDS)
def test_fuzz_selectors(nv, orc, seed):
    rng = np.random.default_rng(9000 + seed)
    l_sel = int(rng.choice([16, 32, 64]))
    d = int(rng.choice([d_ for d_ in (8, 16) if l_sel % d_ == 0]))
    l = int(rng.choice([d, 2 * d]))
    S = int(rng.integers(1, 1500))
    n_top = int(rng.integers(1, 24))
    B, G = int(rng.integers(1, 3)), int(rng.integers(1, 4))
    m = nv.build_block_meta(S, l, d, l_sel, n_top, 64)
    om = orc.build_block_meta(S, l, d, l_sel, n_top, 64)
    p = rng.random((B, S, G, m.S_sel), dtype=np.float32)
    p[rng.random(p.shape) < 0.3] = 0.5  # many exact ties
    if seed % 5 == 0:
        p[:] = 0.0
    r = nv.select_topn_ranges_batched(dev(p), m, n_top, S)
    assert np.array_equal(r.cpu().numpy(), orc.select_topn_ranges_batched(p, om, n_top, S))
    t = int(rng.integers(0, S))
    r1 = nv.select_topn_ranges(dev(p[:, t]), m, n_top, t)
    ref1 = orc.select_topn_ranges(p[:, t], om, n_top, t)
    a, b = orc.normalise_ranges(r1.cpu().numpy()), orc.normalise_ranges(ref1)
    assert a == b


@pytest.mark.parametrize("seed", SEEDS_BWD)
def test_fuzz_backward(nv, orc, seed):
    """selection-attention and band backward (MFMA kernels for bf16/f16 D = 64, generic otherwise) vs the oracle's fp64 backward"""
    from nsa_vibe_amd.band_attention import band_attention_hip

    rng = np.random.default_rng(9500 + seed)
    dtype = [torch.bfloat16, torch.float16, torch.float32][seed % 3]
    B, G = int(rng.integers(1, 3)), int(rng.integers(1, 4))
    h = int(rng.choice([1, 3, 6, 8, 16]))
    D = 64 if seed % 4 else 32
    S, n = int(rng.integers(2, 150)), int(rng.integers(1, 8))
    Q = rng.standard_normal((B, S, G, h, D), dtype=np.float32)
    K = rng.standard_normal((B, G, S, D), dtype=np.float32)
    V = rng.standard_normal((B, G, S, D), dtype=np.float32)
    dO = rng.standard_normal((B, S, G, h, D), dtype=np.float32)
    st = rng.integers(0, S, size=(B, S, G, n))
    rg = np.stack([st, np.minimum(st + rng.integers(0, 70, size=st.shape), S)], axis=-1).astype(np.int32)
    tol = 2e-2 if dtype != torch.float32 else 2e-3

    def check(grads, refs):
        for got, ref, name in zip(grads, refs, ("dQ", "dK", "dV")):
            g = got.float().cpu().numpy()
            assert np.isfinite(g).all(), name
            assert np.abs(g - ref).max() <= tol * max(1.0, np.abs(ref).max()), name

    q, k, v = (dev(x, dtype).requires_grad_(True) for x in (Q, K, V))
    nv.selection_attention_hip(q, k, v, dev(rg)).backward(dev(dO, dtype))
    check((q.grad, k.grad, v.grad), orc.sel_attention_masked_bwd(rounded(Q, dtype), rounded(K, dtype), rounded(V, dtype), rg, rounded(dO, dtype)))
    band = dict(w=int(rng.integers(1, 100))) if seed % 2 else dict(a=16, dd=8, c=1)
    S_kv = S if "w" in band else max((S - 16) // 8 + 1, 0)
    q, k, v = (dev(x, dtype).requires_grad_(True) for x in (Q, K[:, :, :S_kv], V[:, :, :S_kv]))
    band_attention_hip(q, k, v, **band).backward(dev(dO, dtype))
    if S_kv > 0:
        check((q.grad, k.grad, v.grad), orc.band_attention_bwd(rounded(Q, dtype), rounded(K[:, :, :S_kv], dtype), rounded(V[:, :, :S_kv], dtype),
                                                                rounded(dO, dtype), **band))


SEEDS_MOD = range(SEED0, SEED0 + max(1, N_SEEDS // 3))


@pytest.mark.parametrize("seed", SEEDS_MOD)
def test_fuzz_module_native_vs_eager(seed, monkeypatch):
    """random layer geometries: the fused inference route (nsa_layer_prefill + nsa_layer_decode_step) against the eager composition of
    the differentiable ops, fp32 so that the bound is tight (bf16 is covered by the fixed configurations of test_hip_module.py)"""
    from nsa_vibe_amd.nsa_attention import NSAAttention

    rng = np.random.default_rng(9100 + seed)
    G = int(rng.choice([1, 2, 3, 4]))
    h = int(rng.choice([1, 2, 3, 4, 6, 8]))
    dk = int(rng.choice([16, 32, 64, 64, 128]))
    dv = dk if rng.random() < 0.7 else int(rng.choice([16, 32, 64]))
    d = int(rng.choice([2, 4, 8, 16]))
    l = d * int(rng.choice([1, 2, 4]))
    l_sel = d * int(rng.choice([2, 4, 8]))
    n_sel = int(rng.integers(3, 10))
    w = int(rng.integers(1, 150))
    dim = int(rng.choice([64, 96, 128]))
    B, S, n_dec = int(rng.integers(1, 4)), int(rng.integers(1, 260)), int(rng.integers(1, 24))
    selector = ["sequential", "batched"][seed % 2]
    torch.manual_seed(seed)
    m = NSAAttention(dim, G * h, G, dk, dv, l=l, d=d, l_sel=l_sel, n_sel=n_sel, w=w, selector=selector).cuda().float().eval()
    x = torch.randn(B, S + n_dec, dim, device="cuda")
    outs = {}
    for mode in ("native", "eager"):
        if mode == "eager":
            monkeypatch.setenv("NSA_HIP_EAGER_TRAIN", "1")
        else:
            monkeypatch.delenv("NSA_HIP_EAGER_TRAIN", raising=False)
        kv = m.new_kv(B, S + (n_dec if seed % 3 else 0), "cuda", torch.float32)  # every third case grows its cache while decoding
        with torch.set_grad_enabled(mode == "eager"):
            o, kv = m(x[:, :S], kv, prefill=True)
            dec, rgs = [], []
            for t in range(S, S + n_dec):
                y, kv = m(x[:, t: t + 1], kv, prefill=False)
                dec.append(y.detach())
                rgs.append(m._last_ranges.clone())
        outs[mode] = (o.detach(), torch.cat(dec, dim=1), torch.stack(rgs))
    cfg = dict(G=G, h=h, dk=dk, dv=dv, l=l, d=d, l_sel=l_sel, n_sel=n_sel, w=w, dim=dim, B=B, S=S, n_dec=n_dec, selector=selector)
    for a, e in zip(outs["native"][:2], outs["eager"][:2]):
        assert torch.isfinite(a).all(), cfg
        assert (a - e).abs().max().item() <= 1e-3 * max(1.0, e.abs().max().item()), cfg
    # decode ranges: identical sets of selected tokens (the eager selector may order equal-score picks differently only on exact ties)
    assert torch.equal(outs["native"][2].reshape(-1, 2), outs["eager"][2].reshape(-1, 2)), cfg


@pytest.mark.parametrize("seed", SEEDS_MOD)
def test_fuzz_training_gradients_native_vs_eager(seed, monkeypatch):
    """random layer geometries, fp32: gradients of the native training route (native backward kernels behind autograd Functions) against
    torch autograd through the eager composition"""
    from nsa_vibe_amd.nsa_attention import NSAAttention

    rng = np.random.default_rng(9700 + seed)
    G = int(rng.choice([1, 2, 3]))
    h = int(rng.choice([1, 2, 4, 6]))
    dk = int(rng.choice([16, 32, 64, 64]))
    dv = dk if rng.random() < 0.7 else int(rng.choice([16, 32, 64]))
    d = int(rng.choice([2, 4, 8, 16]))
    l = d * int(rng.choice([1, 2, 4]))
    l_sel = d * int(rng.choice([2, 4, 8]))
    n_sel = int(rng.integers(3, 8))
    w = int(rng.integers(1, 100))
    dim = int(rng.choice([64, 96]))
    B, S = int(rng.integers(1, 3)), int(rng.integers(2, 200))
    cfg = dict(G=G, h=h, dk=dk, dv=dv, l=l, d=d, l_sel=l_sel, n_sel=n_sel, w=w, dim=dim, B=B, S=S)
    torch.manual_seed(seed)
    m = NSAAttention(dim, G * h, G, dk, dv, l=l, d=d, l_sel=l_sel, n_sel=n_sel, w=w, selector="batched").cuda().float().train()
    x0 = torch.randn(B, S, dim, device="cuda")
    go = torch.randn(B, S, dim, device="cuda")
    grads = {}
    for mode in ("native", "eager"):
        if mode == "eager":
            monkeypatch.setenv("NSA_HIP_EAGER_TRAIN", "1")
        else:
            monkeypatch.delenv("NSA_HIP_EAGER_TRAIN", raising=False)
        m.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        y, _ = m(x, m.new_kv(B, S, "cuda", torch.float32), prefill=True)
        y.backward(go)
        grads[mode] = [y.detach(), x.grad.detach()] + [torch.zeros_like(p) if p.grad is None else p.grad.detach().clone()
                                                        for p in m.parameters()]  # S < l: the compressed projections see no gradient
    for a, e in zip(grads["native"], grads["eager"]):
        assert torch.isfinite(a).all(), cfg
        assert (a - e).abs().max().item() <= 2e-3 * max(1.0, e.abs().max().item()), cfg


@pytest.mark.parametrize("seed", SEEDS)
def test_fuzz_decode_step(nv, orc, seed, tune):
    """the one-launch decode step over random shapes -- contexts from a few tokens to 131k (t not a multiple of anything, caches longer
    than the context, contexts before the first compressed token / the first complete block / the third block, where the forced blocks
    collapse and nothing is fetched ahead), 1..16 heads per group, 3..24 ranges per row, up to 48 rows, 16 / 8 waves, split or not --
    against the three separate launches (ranges and O bit-identical) and the oracle (selector on the device scores; attention)."""
    rng = np.random.default_rng(9000 + seed)
    dtype = [torch.bfloat16, torch.float16][seed % 2]
    B, G = int(rng.integers(1, 13)), int(rng.choice([1, 2, 4]))
    h = int(rng.choice([1, 2, 3, 4, 6, 6, 6, 8, 12, 16]))
    n = int(rng.choice([3, 4, 8, 16, 16, 16, 24]))
    t = int(rng.choice([int(rng.integers(0, 64)), int(rng.integers(64, 200)), int(rng.integers(200, 3000)), int(rng.integers(3000, 20000)), int(rng.integers(60000, 131000))]))
    S_ctx = t + 1 + int(rng.integers(0, 70))  # the cache may hold more tokens than the step may see
    D = 64
    meta = nv.build_block_meta(t + 1, 32, 16, 64, n, 512)
    mo = orc.build_block_meta(t + 1, 32, 16, 64, n, 512)
    Q = rng.standard_normal((B, 1, G, h, D), dtype=np.float32)
    Kc = rng.standard_normal((B, G, max(meta.S_cmp, 1), D), dtype=np.float32)[:, :, : meta.S_cmp]
    K = rng.standard_normal((B, G, S_ctx, D), dtype=np.float32)
    V = rng.standard_normal((B, G, S_ctx, D), dtype=np.float32)
    Qd, Kcd, Kd, Vd = dev(Q, dtype), dev(Kc, dtype), dev(K, dtype), dev(V, dtype)
    tune("DECODE_UNFUSED", 1)
    tune("DECODE_WAVES", int(rng.choice([-1, 8, 16])))
    O0, r0 = nv.selection_decode_step(Qd, Kcd, Kd, Vd, meta, n, t)
    tune("DECODE_UNFUSED", -1)
    tune("DECODE_SPLIT", int(rng.choice([-1, 1, 2, 4, 16])))
    tune("DECODE_TEAM_SPIN", int(rng.choice([-1, -1, 0])))
    O1, r1 = nv.selection_decode_step(Qd, Kcd, Kd, Vd, meta, n, t)
    torch.cuda.synchronize()
    assert torch.equal(r0, r1) and torch.equal(O0, O1), (B, G, h, n, t)
    if meta.S_cmp > 0:
        p = nv.selection_scores(Qd, Kcd, meta)[:, 0].cpu().numpy()
    else:
        p = np.zeros((B, G, meta.S_sel), np.float32)
    assert norm_ranges(r1.cpu().numpy()) == norm_ranges(orc.select_topn_ranges(p, mo, n, t)), (B, G, h, n, t)
    ref = orc.sel_attention_masked(rounded(Q, dtype), rounded(K, dtype), rounded(V, dtype), r1.cpu().numpy()[:, None])
    assert np.abs(O1.float().cpu().numpy() - ref).max() <= _bound(ref, dtype)


def norm_ranges(r):
    return [[(int(s), int(e)) for s, e in row if e > s] for row in np.asarray(r).reshape(-1, r.shape[-2], 2)]
