"""GPU parity: scoring (p_cmp, Eq.9, Eq.10) and deterministic top-n range selection.

Bars: selected ranges bit-exact given an identical fp32 p_grp; Eq.9 p_slc bit-exact given an
identical fp32 p_cmp; Eq.10 p_grp bit-exact on every BASELINE shape (S_sel >= 64; see
oracle/make_goldens.py for torch's shape-dependent CPU reduction order on small inner dims);
softmax scores within 1e-6 absolute."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nv():
    import nsa_vibe_amd

    assert torch.cuda.is_available()
    return nsa_vibe_amd


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.to(dtype) if dtype is not None else t


def norm(r):
    r = np.asarray(r)
    flat = r.reshape(-1, r.shape[-2], 2)
    return [[(int(s), int(e)) for s, e in row if e > s] for row in flat]


def test_g2_tiebreak(nv):
    g = load_golden("g2_tiebreak")
    m = nv.build_block_meta(64, 4, 2, 4, 8, 8)
    for dt in (torch.float32, torch.float16, torch.bfloat16):  # test_selection_tiebreak.py:17-58
        r = nv.select_topn_ranges(torch.ones(1, 1, m.S_sel, dtype=dt).cuda(), m, 3, 63, force_init=False, force_local=0)
        assert norm(r.cpu().numpy()) == norm(g["r_seq"])
        rb = nv.select_topn_ranges_batched(torch.ones(1, 3, 1, m.S_sel, dtype=dt).cuda(), m, 3, 3, force_init=False, force_local=0)
        assert np.array_equal(rb.cpu().numpy(), g["r_bat"])
    m2 = nv.build_block_meta(1024, 32, 16, 64, 16, 512)
    r = nv.select_topn_ranges(torch.ones(1, 1, 16).cuda(), m2, 3, 1023, force_init=False, force_local=0)
    assert norm(r.cpu().numpy()) == [[(0, 192)]]


def test_g3_v2_converter(nv):
    g = load_golden("g3_v2_converter")
    m = nv.build_block_meta(1024, 32, 16, 64, 16, 512)
    for n in sorted(k[:-4] for k in g.files if k.endswith("_idx")):
        idx = g[n + "_idx"]
        out = nv.convert_indices_to_ranges_batched_v2(dev(idx).long(), m, idx.shape[1])
        assert np.array_equal(out.cpu().numpy(), g[n + "_ranges"]), n


@pytest.mark.parametrize("S", [4096, 65536])
def test_g4_needle(nv, S):
    g = load_golden("g4_needle")
    m = nv.build_block_meta(S, 32, 16, 64, 8, 512)
    p = torch.zeros(1, 1, 2, 1, m.S_cmp, device="cuda")
    p[..., int(g[f"S{S}_cmp_row"])] = 1.0
    p_slc = nv.map_pcmp_to_pslc_batched(p, m)
    p_grp = nv.group_reduce_pslc(p_slc.squeeze(1))
    assert np.array_equal(p_grp.cpu().numpy(), g[f"S{S}_p_grp"])
    r = nv.select_topn_ranges(p_grp, m, 8, S - 1, True, 2)
    assert norm(r.cpu().numpy()) == norm(g[f"S{S}_ranges"])


def test_g9_seq_vs_batched(nv):
    g = load_golden("g9_seq_vs_batched")
    S = 4096
    m = nv.build_block_meta(S, 32, 16, 64, 16, 512)
    p = dev(gi.g9_scores(S))
    rb = nv.select_topn_ranges_batched(p, m, 16, S, True, 2)
    assert np.array_equal(rb.cpu().numpy(), g["r_batched"])
    for i, t in enumerate(g["ts"]):
        r = nv.select_topn_ranges(p[:, int(t)], m, 16, int(t), True, 2)
        assert norm(r.cpu().numpy()) == norm(g["r_seq"][i]), int(t)
    # all rows in one launch == row-by-row
    rows = nv.select_topn_ranges_rows(p, m, 16).cpu().numpy()
    for i, t in enumerate(g["ts"]):
        assert norm(rows[:, int(t)]) == norm(g["r_seq"][i])


def test_g9_small_forced_columns(nv, orc):
    g = load_golden("g9_small_forced_cols")
    for key in g.files:
        S, n = (int(x[1:]) for x in key.split("_"))
        m = nv.build_block_meta(S, 32, 16, 64, 16, 512)
        ps = dev(gi.g9_scores_small(S, m.S_sel))
        out = nv.select_topn_ranges_batched(ps, m, n, S, True, 2).cpu().numpy()
        assert out.shape == g[key].shape and np.array_equal(out, g[key]), key


@pytest.mark.parametrize("S", [4096, 16384, 65536])
def test_g10_m7c_chain(nv, S):
    g = load_golden(f"g10_m7c_S{S}")
    ts = gi.g10_rows(S)
    m = nv.build_block_meta(S, 32, 16, 64, 16, 512)
    S_sel = m.S_sel
    Qr, Kc = gi.g10_q_kcmp(S, ts)
    pin = slice(0, None, max(1, len(ts) // 8))
    # scores within 1e-6 of the reference
    p_cmp = nv.compute_pcmp_all(dev(Qr[:, pin]), dev(Kc), 0.125)
    assert np.abs(p_cmp.cpu().numpy()[0] - g["p_cmp_pin"]).max() < 1e-6
    # Eq.9+Eq.10 bit-exact given the reference's p_cmp
    pg = nv.map_pcmp_to_pgrp(dev(g["p_cmp_pin"]), m)
    assert np.array_equal(pg.cpu().numpy(), g["p_grp_pin"])
    # fused scorer within 1e-6 of the reference's p_grp
    pgf = nv.selection_scores(dev(Qr), dev(Kc), m)
    assert np.abs(pgf.cpu().numpy()[0] - g["p_grp"]).max() < 2e-6
    # ranges bit-exact given the reference's p_grp: sequential mode with per-row t ...
    import nsa_vibe_amd.selection_scorer as ss

    t_rows = dev(np.repeat(ts, 2).astype(np.int32))
    rs = ss._select(dev(g["p_grp"]).reshape(-1, S_sel), len(ts) * 2, 1, 2, 0, t_rows, m, 16, True, 2, 0, 1, 16)
    assert norm(rs.cpu().numpy()) == norm(g["r_seq"])
    # ... and batched mode on the full [1,S,G,S_sel] tensor (other rows zero, as the fixture was made)
    full = torch.zeros(1, S, 2, S_sel, device="cuda")
    full[0, dev(ts).long()] = dev(g["p_grp"])
    rb = nv.select_topn_ranges_batched(full, m, 16, S)
    assert np.array_equal(rb[0, dev(ts).long()].cpu().numpy(), g["r_bat"])


def test_g11_small_chains(nv):
    g = load_golden("g11_small_chains")
    for ci, cfg in enumerate(g["cfgs"]):
        S, l, d, ls, n_top, G, h, D = (int(x) for x in cfg)
        m = nv.build_block_meta(S, l, d, ls, n_top, 512)
        Q, Kc, K, V = gi.g11_inputs(ci, S, G, h, D, m.S_cmp)
        p_cmp = nv.compute_pcmp_all(dev(Q), dev(Kc), 1.0 / np.sqrt(D))
        assert np.abs(p_cmp.cpu().numpy() - g[f"c{ci}_p_cmp"]).max() < 1e-6
        ref_pc = dev(g[f"c{ci}_p_cmp"])
        p_slc = nv.map_pcmp_to_pslc_batched(ref_pc, m)
        assert np.array_equal(p_slc.cpu().numpy(), g[f"c{ci}_p_slc"])  # Eq.9 bit-exact
        p_grp = nv.map_pcmp_to_pgrp(ref_pc, m)
        if bool(g[f"c{ci}_pgrp_bitexact"]):
            assert np.array_equal(p_grp.cpu().numpy(), g[f"c{ci}_p_grp"])
        else:
            assert np.allclose(p_grp.cpu().numpy(), g[f"c{ci}_p_grp"], rtol=3e-7, atol=0)
        cut = int(g[f"c{ci}_cut"])
        pgc = nv.map_pcmp_to_pgrp(ref_pc[..., :cut], m)  # decode: fewer cmp rows than the meta knows
        assert np.allclose(pgc.cpu().numpy(), g[f"c{ci}_p_grp_cut"], rtol=3e-7, atol=0)
        rb = nv.select_topn_ranges_batched(dev(g[f"c{ci}_p_grp"]), m, n_top, S)
        assert np.array_equal(rb.cpu().numpy(), g[f"c{ci}_r_bat"])
        for i, t in enumerate(g[f"c{ci}_ts"]):
            r = nv.select_topn_ranges(dev(g[f"c{ci}_p_grp"])[:, int(t)], m, n_top, int(t), True, 2)
            assert norm(r.cpu().numpy()) == norm(g[f"c{ci}_r_seq"][i])
        # whole chain on the device: scores -> ranges -> attention, against the reference's O
        pg_dev = nv.selection_scores(dev(Q), dev(Kc), m)
        rb_dev = nv.select_topn_ranges_batched(pg_dev, m, n_top, S)
        same = (rb_dev.cpu().numpy() == g[f"c{ci}_r_bat"]).all(axis=(-1, -2))
        # device scores differ from the reference's by ~1e-7 (checked: < 2e-6), so a row may flip only where its last picked and first
        # rejected ranking keys are closer than that: every row whose gap in the REFERENCE p_grp exceeds 4e-6 must match bit for bit
        # (the gate of test_g10_ranges_from_q_k_are_exact_where_the_score_gap_allows; on these fixtures it lets every row through)
        assert np.abs(pg_dev.cpu().numpy() - g[f"c{ci}_p_grp"]).max() < 2e-6
        ts_all = np.repeat(np.arange(S), 1)
        gaps = np.stack([_topn_gap(g[f"c{ci}_p_grp"][b_].reshape(-1, m.S_sel), ts_all, n_top=n_top, l_sel=ls, G=G) for b_ in range(Q.shape[0])])
        gated = gaps.reshape(same.shape) > 4e-6
        assert gated.mean() > 0.9 and same[gated].all(), (ci, float(gated.mean()), float(same.mean()))
        O = nv.selection_attention_hip(dev(Q), dev(K), dev(V), dev(g[f"c{ci}_r_bat"]))
        assert np.abs(O.cpu().numpy() - g[f"c{ci}_O"]).max() <= 1e-3


@pytest.mark.parametrize("S,B,h,D", [(4096, 2, 6, 64), (1000, 1, 6, 64), (1025, 1, 4, 64), (777, 2, 16, 64), (520, 1, 3, 128), (64, 1, 6, 64), (33, 2, 6, 64)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_mfma_scorer_vs_oracle(nv, orc, S, B, h, D, dtype):
    """fused MFMA scorer (closed-form Eq.9 stencil) vs the oracle chain on the rounded inputs, 1e-6 absolute;
    causal_skip returns exactly the same values on every block a selector can read and 0 elsewhere."""
    rng = np.random.default_rng([S, h, D])
    G = 2
    m = nv.build_block_meta(S, 32, 16, 64, 16, 512)
    mo = orc.build_block_meta(S, 32, 16, 64, 16, 512)
    Q = rng.standard_normal((B, S, G, h, D), dtype=np.float32)
    Kc = rng.standard_normal((B, G, m.S_cmp, D), dtype=np.float32) * 1.5
    rd = lambda a: torch.from_numpy(a).to(dtype).float().numpy()  # noqa: E731
    ref = orc.map_pcmp_to_pslc_and_pgrp(orc.compute_pcmp_all(rd(Q), rd(Kc), 1.0 / np.sqrt(D)), mo)[1]
    got = nv.selection_scores(dev(Q, dtype), dev(Kc, dtype), m, variant=2)
    assert got.shape == ref.shape
    assert np.abs(got.cpu().numpy() - ref).max() < 2e-6
    gen = nv.selection_scores(dev(Q, dtype), dev(Kc, dtype), m, variant=1)
    assert np.abs(gen.cpu().numpy() - ref).max() < 2e-6
    skip = nv.selection_scores(dev(Q, dtype), dev(Kc, dtype), m, variant=2, causal_skip=True)
    t = torch.arange(S, device="cuda").view(1, S, 1, 1)
    j = torch.arange(m.S_sel, device="cuda").view(1, 1, 1, -1)
    readable = (j + 1) * 64 <= t + 1
    assert torch.equal(torch.where(readable, skip, 0), torch.where(readable, got, 0))
    # ranges from the skipped scores are identical to ranges from the full scores, both modes
    assert torch.equal(nv.select_topn_ranges_batched(skip, m, 16, S), nv.select_topn_ranges_batched(got, m, 16, S))
    assert torch.equal(nv.select_topn_ranges_rows(skip, m, 16), nv.select_topn_ranges_rows(got, m, 16))


@pytest.mark.parametrize("S_ctx,B,h,D,geom", [(65536, 1, 6, 64, (32, 16, 64)), (4096, 3, 6, 64, (32, 16, 64)), (1000, 2, 4, 32, (16, 8, 32)),
                                              (300, 2, 5, 24, (8, 4, 16)), (40, 1, 6, 64, (32, 16, 64)), (20, 1, 2, 16, (32, 16, 64))])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_decode_scorer_vs_oracle(nv, orc, S_ctx, B, h, D, geom, dtype):
    """decode-shaped scorer (one query row per (b,g), any dtype / block geometry) vs the oracle chain; then the
    whole decode step scores -> sequential ranges -> attention against the oracle."""
    rng = np.random.default_rng([S_ctx, h, D])
    G = 2
    l, d, ls = geom
    m = nv.build_block_meta(S_ctx, l, d, ls, 16, 512)
    mo = orc.build_block_meta(S_ctx, l, d, ls, 16, 512)
    Q = rng.standard_normal((B, 1, G, h, D), dtype=np.float32)
    Kc = rng.standard_normal((B, G, m.S_cmp, D), dtype=np.float32)
    rd = lambda a: torch.from_numpy(a).to(dtype).float().numpy()  # noqa: E731
    got = nv.selection_scores(dev(Q, dtype), dev(Kc, dtype), m, variant=3)
    assert got.shape == (B, 1, G, m.S_sel)
    if m.S_cmp == 0:
        assert not got.any()
        return
    ref = orc.map_pcmp_to_pslc_and_pgrp(orc.compute_pcmp_all(rd(Q), rd(Kc), 1.0 / np.sqrt(D)), mo)[1]
    assert np.abs(got.cpu().numpy() - ref).max() < 2e-6
    auto = nv.selection_scores(dev(Q, dtype), dev(Kc, dtype), m)  # auto route = decode-shaped for few rows
    assert torch.equal(auto, got)
    t = S_ctx - 1
    r = nv.select_topn_ranges(got[:, 0], m, 16, t)
    r_ref = orc.select_topn_ranges(got[:, 0].cpu().numpy(), mo, 16, t)
    assert norm(r.cpu().numpy()) == norm(r_ref)


@pytest.mark.parametrize("S_ctx,B", [(65536, 1), (65536, 32), (16384, 5), (200, 3), (70, 2)])
def test_fused_decode_step(nv, orc, S_ctx, B):
    """nsa_sel_decode_step == the three separate calls, and == the oracle's decode chain (bf16, m7c shape)."""
    rng = np.random.default_rng([S_ctx, B])
    G, h, D, n = 2, 6, 64, 16
    m = nv.build_block_meta(S_ctx, 32, 16, 64, n, 512)
    mo = orc.build_block_meta(S_ctx, 32, 16, 64, n, 512)
    dt = torch.bfloat16
    Q = rng.standard_normal((B, 1, G, h, D), dtype=np.float32)
    Kc = rng.standard_normal((B, G, m.S_cmp, D), dtype=np.float32)
    K = rng.standard_normal((B, G, S_ctx + 37, D), dtype=np.float32)  # preallocated cache longer than the context
    V = rng.standard_normal((B, G, S_ctx + 37, D), dtype=np.float32)
    t = S_ctx - 1
    Kd, Vd = dev(K, dt), dev(V, dt)
    O, rg = nv.selection_decode_step(dev(Q, dt), dev(Kc, dt), Kd[:, :, :S_ctx], Vd[:, :, :S_ctx], m, n, t)
    p = nv.selection_scores(dev(Q, dt), dev(Kc, dt), m)
    r2 = nv.select_topn_ranges(p[:, 0], m, n, t)
    O2 = nv.selection_attention_hip(dev(Q, dt), Kd[:, :, :S_ctx], Vd[:, :, :S_ctx], r2.unsqueeze(1))
    assert torch.equal(rg, r2) and torch.equal(O, O2)
    rd = lambda a: torch.from_numpy(a).to(dt).float().numpy()  # noqa: E731
    r_ref = orc.select_topn_ranges(p[:, 0].cpu().numpy(), mo, n, t)
    assert norm(rg.cpu().numpy()) == norm(r_ref)
    O_ref = orc.sel_attention_masked(rd(Q), rd(K[:, :, :S_ctx]), rd(V[:, :, :S_ctx]), rg.cpu().numpy()[:, None])
    assert np.abs(O.float().cpu().numpy() - O_ref).max() <= 1e-2


def test_selector_vs_oracle_random_configs(nv, orc):
    """random (l', n_top, forced) configurations incl. wide rows (S_sel up to 2048)."""
    rng = np.random.default_rng(123)
    for S, ls, n_top, fi, fl in [(700, 16, 5, True, 2), (4096, 32, 9, False, 3), (2000, 8, 16, True, 0), (130, 64, 2, True, 2),
                                 (65536, 32, 16, True, 2), (300, 64, 16, True, 2), (1000, 64, 1, True, 2), (515, 4, 7, False, 0)]:
        mo = orc.build_block_meta(S, ls, ls, ls, n_top, 512)
        m = nv.build_block_meta(S, ls, ls, ls, n_top, 512)
        rows = min(S, 257)
        p = rng.random((1, S, 2, m.S_sel), dtype=np.float32)
        if S <= 4096:
            ref = orc.select_topn_ranges_batched(p, mo, n_top, S, fi, fl)
            out = nv.select_topn_ranges_batched(dev(p), m, n_top, S, fi, fl).cpu().numpy()
            assert out.shape == ref.shape and np.array_equal(out, ref), (S, ls, n_top)
        ts = np.sort(rng.choice(S, rows, replace=False)).astype(np.int32)
        ref = orc.select_topn_ranges_rows(p[0, ts].reshape(-1, m.S_sel), np.repeat(ts, 2), mo, n_top, fi, fl)
        out = nv.select_topn_ranges_rows(dev(p), m, n_top, 0, fi, fl).cpu().numpy()[0, ts].reshape(-1, n_top, 2)
        assert norm(out) == norm(ref), (S, ls, n_top)


def test_block_meta_matches_golden(nv):
    g = load_golden("g1_block_meta")
    for i, (S, l, d, ls) in enumerate(g["cases"]):
        m = nv.build_block_meta(int(S), int(l), int(d), int(ls), 16, 512)
        assert np.array_equal(m.M_csl_values.numpy(), g[f"c{i}_values"])
        assert np.array_equal(m.M_csl_coo_indices.numpy(), g[f"c{i}_coo"])


def test_full_size_64k_selection_properties(nv):
    """S=65536 batched selection over every row: structural invariants of the ranges."""
    S, n = 65536, 16
    m = nv.build_block_meta(S, 32, 16, 64, n, 512)
    torch.manual_seed(1)
    p = torch.rand(1, S, 2, m.S_sel, device="cuda")
    r = nv.select_topn_ranges_batched(p, m, n, S)
    assert r.shape == (1, S, 2, 16, 2)
    s, e = r[..., 0].long(), r[..., 1].long()
    t = torch.arange(S, device="cuda").view(1, S, 1, 1)
    valid = e > s
    assert (e[valid.expand_as(e)] <= (t + 1).expand_as(e)[valid]).all()  # causality (nsa_attention.py:1124-1133)
    assert ((s % 64 == 0) | ~valid).all() and ((e % 64 == 0) | ~valid).all()  # batched mode: whole blocks only
    nxt_s = s[..., 1:]
    assert ((nxt_s > e[..., :-1]) | ~valid[..., 1:]).all()  # ascending, disjoint, merged (no touching ranges)
    # forced blocks (selection_scorer.py:283-300,344-347): block 0 and block t//64 - 1 are always covered; the
    # current block t//64 only when it is complete (t % 64 == 63) -- batched mode drops the partial block
    tt = torch.cat([torch.arange(128, S, 997, device="cuda"), torch.arange(191, S, 6400, device="cuda")])
    rr = r[0, tt]  # [T,G,n,2]
    cur = tt // 64
    for blk, need in ((torch.zeros_like(cur), torch.ones_like(cur, dtype=torch.bool)), (cur - 1, torch.ones_like(cur, dtype=torch.bool)),
                      (cur, tt % 64 == 63)):
        pos = (blk * 64).view(-1, 1, 1)
        cov = ((rr[..., 0] <= pos) & (pos < rr[..., 1])).any(dim=-1)  # [T,G]
        assert (cov | ~need.view(-1, 1)).all()
    # selected token count once t is large: 13 scored + forced {0, t//64-1} = 15 blocks, 16 when the current
    # block is complete and therefore kept (the partial current block is a forced pick that batched mode drops)
    L = (e - s).clamp_min(0).sum(-1)
    want = torch.where(torch.arange(S, device="cuda") % 64 == 63, 1024, 960).view(S, 1)
    assert (L[0, 2048:] == want[2048:]).all()
    # idempotence / determinism (selection_scorer.py:714-758)
    assert torch.equal(r, nv.select_topn_ranges_batched(p, m, n, S))


@pytest.mark.parametrize("S_ctx,B", [(40, 1), (700, 3), (5000, 2), (16400, 1), (65536, 1), (65536, 32)])
def test_fused_decode_scorer_equals_three_kernel_route(nv, S_ctx, B, tune):
    """decode: logits -> statistics -> Eq.9/10 -> top-n in one launch must give the ranges (bit-exact) and the output of the
    three-kernel route (tuning switch DECODE_UNFUSED = 1), which the other tests pin against the oracle / reference goldens"""
    import torch

    g = torch.Generator(device="cuda")
    g.manual_seed(S_ctx)
    meta = nv.build_block_meta(S_ctx, 32, 16, 64, 16, 512)
    mk = lambda *sh: torch.randn(*sh, device="cuda", generator=g).bfloat16()  # noqa: E731
    Q, Kc = mk(B, 1, 2, 6, 64), mk(B, 2, max(meta.S_cmp, 1), 64)[:, :, : meta.S_cmp]
    K, V = mk(B, 2, S_ctx, 64), mk(B, 2, S_ctx, 64)
    t = S_ctx - 1
    tune("DECODE_UNFUSED", 0)
    O1, r1 = nv.selection_decode_step(Q, Kc, K, V, meta, 16, t)
    tune("DECODE_UNFUSED", 1)
    O2, r2 = nv.selection_decode_step(Q, Kc, K, V, meta, 16, t)
    torch.cuda.synchronize()
    assert torch.equal(r1, r2)
    assert torch.equal(O1, O2)
    # the fused kernel's closed-form Eq.9 taps (l = 2d, l' = 4d) against the same kernel reading the CSC
    tune("DECODE_UNFUSED", 0)
    tune("DECODE_STENCIL", 0)
    O3, r3 = nv.selection_decode_step(Q, Kc, K, V, meta, 16, t)
    torch.cuda.synchronize()
    assert torch.equal(r1, r3)
    assert torch.equal(O1, O3)


@pytest.mark.parametrize("t", [5032, 5055, 16383 + 48])
def test_fused_decode_with_a_meta_older_than_the_cache(nv, tune, t):
    """The reference rebuilds kv.meta only when t enters a new selection block while K_cmp grows every d tokens
    (nsa_attention.py:617-632): compressed rows newer than the meta have no CSC entry and are dropped from p_slc.  They reach only
    the current and the previous block, both forced, so the ranges cannot depend on them: the closed-form taps
    (which see every row) and the CSC of the old meta must select the same ranges."""
    import torch

    g = torch.Generator(device="cuda")
    g.manual_seed(t)
    S_now, S_old = t + 1, (t // 64) * 64 + 1  # the meta dates from the first token of the current selection block
    meta_now, meta_old = nv.build_block_meta(S_now, 32, 16, 64, 16, 512), nv.build_block_meta(S_old, 32, 16, 64, 16, 512)
    assert meta_old.S_sel == meta_now.S_sel and meta_old.S_cmp < meta_now.S_cmp
    mk = lambda *sh: torch.randn(*sh, device="cuda", generator=g).bfloat16()  # noqa: E731
    B = 3
    Q, Kc, K, V = mk(B, 1, 2, 6, 64), mk(B, 2, meta_now.S_cmp, 64), mk(B, 2, S_now, 64), mk(B, 2, S_now, 64)
    tune("DECODE_STENCIL", 1)
    O1, r1 = nv.selection_decode_step(Q, Kc, K, V, meta_old, 16, t)
    tune("DECODE_STENCIL", 0)
    O2, r2 = nv.selection_decode_step(Q, Kc, K, V, meta_old, 16, t)
    O3, r3 = nv.selection_decode_step(Q, Kc, K, V, meta_now, 16, t)
    torch.cuda.synchronize()
    assert torch.equal(r1, r2) and torch.equal(O1, O2)
    assert torch.equal(r1, r3) and torch.equal(O1, O3)


def test_beyond_64k_selection_matches_oracle(nv, orc):
    """S = 131072 (S_sel = 2048: the 32-candidates-per-lane selector, the two-launch fallback of select_and_attend): batched ranges
    bit-exact against the oracle on every row of one group, attention finite and consistent with the separate calls"""
    import torch

    S, n = 131072, 16
    m = nv.build_block_meta(S, 32, 16, 64, n, 512)
    assert m.S_sel == 2048
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    p = torch.rand(1, S, 1, m.S_sel, device="cuda", generator=g)
    r = nv.select_topn_ranges_batched(p, m, n, S)
    om = orc.build_block_meta(S, 32, 16, 64, n, 512)
    ref = orc.select_topn_ranges_batched(p.cpu().numpy(), om, n, S)
    assert np.array_equal(r.cpu().numpy(), ref)
    mk = lambda *sh: torch.randn(*sh, device="cuda", generator=g).bfloat16()  # noqa: E731
    Q, K, V = mk(1, S, 1, 6, 64), mk(1, 1, S, 64), mk(1, 1, S, 64)
    r2, O = nv.select_and_attend(p, Q, K, V, m, n, mode="batched")
    assert torch.equal(r2, r) and torch.isfinite(O).all()
    assert torch.equal(O, nv.selection_attention_hip(Q, K, V, r))


def test_262144_tokens_selector(nv, orc):
    """S = 262144 (S_sel = 4096, 64 candidates per lane): sequential selection of sampled decode positions bit-exact vs the oracle"""
    import torch

    S, n = 262144, 16
    m = nv.build_block_meta(S, 32, 16, 64, n, 512)
    om = orc.build_block_meta(S, 32, 16, 64, n, 512)
    assert m.S_sel == 4096
    g = torch.Generator(device="cuda")
    g.manual_seed(6)
    for t in (262143, 200000, 131072, 131071, 70001):
        p = torch.rand(2, 3, m.S_sel, device="cuda", generator=g)
        p[:, :, ::5] = 0.125  # ties
        r = nv.select_topn_ranges(p, m, n, t).cpu().numpy()
        ref = orc.select_topn_ranges(p.cpu().numpy(), om, n, t)
        assert orc.normalise_ranges(r) == orc.normalise_ranges(ref), t


def test_scores_leave_skipped_entries_unwritten_same_selection(nv):
    """causal_skip with leave_skipped=True (no zero fill of p_grp): every entry a selector may read equals the zero-filled result, so
    ranges and attention are identical; the tensor is pre-poisoned with NaN to prove nothing unwritten is read"""
    torch.manual_seed(4)
    B, S, G, h, D = 2, 1100, 2, 6, 64
    meta = nv.build_block_meta(S, 32, 16, 64, 16, 512)
    Q = torch.randn(B, S, G, h, D, device="cuda").bfloat16()
    Kc = torch.randn(B, G, meta.S_cmp, D, device="cuda").bfloat16()
    K = torch.randn(B, G, S, D, device="cuda").bfloat16()
    V = torch.randn(B, G, S, D, device="cuda").bfloat16()
    ref = nv.selection_scores(Q, Kc, meta, causal_skip=True)
    # poison the allocator's next block of this size, then ask for the unwritten form
    poison = torch.full_like(ref, float("nan"))
    del poison
    got = nv.selection_scores(Q, Kc, meta, causal_skip=True, leave_skipped=True)
    t = torch.arange(S, device="cuda")
    valid = (torch.arange(meta.S_sel, device="cuda")[None, :] + 1) * 64 <= (t[:, None] + 1)  # [S,S_sel]
    m = valid[None, :, None, :].expand_as(ref)
    assert torch.equal(got[m], ref[m])
    for mode in ("batched", "sequential"):
        r0, o0 = nv.select_and_attend(ref, Q, K, V, meta, 16, mode=mode, scale=0.125)
        r1, o1 = nv.select_and_attend(got, Q, K, V, meta, 16, mode=mode, scale=0.125)
        assert torch.equal(r0, r1) and torch.equal(o0, o1)
    assert torch.equal(nv.select_topn_ranges_batched(got, meta, 16, S), nv.select_topn_ranges_batched(ref, meta, 16, S))


def test_per_step_scorer_names_and_verifiers(nv, orc, monkeypatch):
    """compute_pcmp (per-step form, with and without the batch axis), the converter's three names, and the two opt-in verifiers of the
    reference's scorer module (selection_scorer.py:10-39, 365-431, 658-760) on the HIP kernels"""
    torch.manual_seed(2)
    B, G, h, D, S = 2, 2, 4, 64, 700
    meta = nv.build_block_meta(S, 32, 16, 64, 8, 128)
    Q = torch.randn(B, G, h, D, device="cuda")
    Kc = torch.randn(B, G, meta.S_cmp, D, device="cuda")
    p = nv.compute_pcmp(Q, Kc, 0.125)
    ref = torch.softmax(torch.einsum("bghd,bgcd->bghc", Q, Kc) * 0.125, dim=-1)
    assert p.shape == (B, G, h, meta.S_cmp) and (p - ref).abs().max().item() <= 1e-6
    p1 = nv.compute_pcmp(Q[0], Kc[:1], 0.125)
    assert p1.shape == (1, G, h, meta.S_cmp) and torch.equal(p1[0], p[0])
    idx = torch.tensor([[[[0, 1, 2, 5, 6, -1]]]], dtype=torch.int32, device="cuda").expand(1, 3, 2, 6).contiguous()
    a, b_, c = (f(idx, meta, 3) for f in (nv.convert_indices_to_ranges_batched, nv.convert_indices_to_ranges_batched_dispatch,
                                          nv.convert_indices_to_ranges_batched_v2))
    assert torch.equal(a, b_) and torch.equal(a, c)
    monkeypatch.setenv("NSA_VERIFY_EQ9_MAPPING", "1")
    ok, info = nv.verify_mapping_equivalence(p.unsqueeze(1), meta)
    assert ok and info["status"] == "verified" and info["max_abs_diff"] <= 1e-6
    monkeypatch.setenv("NSA_VALIDATE_SELECTION_DETERMINISM", "1")
    p_grp = nv.map_pcmp_to_pgrp(p, meta)
    assert nv.validate_selection_determinism(p_grp, meta, 8, S - 1)


@pytest.mark.parametrize("S,n_top", [(4096, 16), (16384, 16), (65536, 16), (65536, 40), (2048, 5), (700, 16)])
def test_selector_tie_heavy_scores_match_oracle(nv, orc, S, n_top):
    """the threshold select counts on per-lane SORTED keys and picks on the original ones: exact ties are where that can go wrong.
    Scores quantised to a few levels (hundreds of exact ties at the threshold, resolved by ascending block index), all-equal rows,
    rows with fewer valid candidates than picks, zeros (negative ranking keys) and NaN candidates; both selector modes, every
    candidates-per-lane instantiation up to 16; bit-exact against the oracle (order: key descending, index ascending)"""
    rng = np.random.default_rng([S, n_top])
    mo = orc.build_block_meta(S, 32, 16, 64, n_top, 512)
    m = nv.build_block_meta(S, 32, 16, 64, n_top, 512)
    rows = 192
    ts = np.sort(rng.choice(S, rows, replace=False)).astype(np.int32)
    ts[:4] = (0, 63, 64, 129)
    ts[-1] = S - 1
    p = np.floor(rng.random((rows, 2, m.S_sel), dtype=np.float32) * 4.0) / 4.0  # 4 levels
    p[5] = 1.0  # everything equal
    p[6] = 0.0  # keys are -idx*1e-8: negative floats
    p[7, :, ::3] = np.nan  # NaN candidates never compete
    p[8:40] = (np.floor(rng.random((32, 2, m.S_sel), dtype=np.float32) * 64.0) / 64.0).astype(np.float32)
    p[40:60] = rng.random((20, 2, m.S_sel), dtype=np.float32)  # no ties
    p[60:70, :, : m.S_sel // 2] = 0.5  # one long plateau next to random values
    ref = orc.select_topn_ranges_rows(p.reshape(-1, m.S_sel), np.repeat(ts, 2), mo, n_top, True, 2)
    full = np.zeros((1, S, 2, m.S_sel), np.float32)
    full[0, ts] = p
    out = nv.select_topn_ranges_rows(dev(full), m, n_top, 0, True, 2).cpu().numpy()[0, ts].reshape(-1, n_top, 2)
    assert norm(out) == norm(ref)
    if S <= 16384:  # batched mode on whole sequences (the oracle walks every row)
        refb = orc.select_topn_ranges_batched(full, mo, n_top, S)
        outb = nv.select_topn_ranges_batched(dev(full), m, n_top, S).cpu().numpy()
        assert np.array_equal(outb, refb)


def _topn_gap(p_grp_rows, ts, n_top=16, l_sel=64, G=2):
    """gap between the last picked and the first rejected ranking key of every row (reference p_grp, fp32 keys as
    selection_scorer.py:182-184 forms them); inf when the row has no rejected candidate.  Rows whose gap is larger than the score
    error of the device chain must select exactly the reference's blocks."""
    rows, S_sel = p_grp_rows.shape
    gaps = np.full(rows, np.inf, np.float64)
    idx = np.arange(S_sel, dtype=np.float32)
    for r in range(rows):
        t = int(ts[r // G])
        nvalid = min(S_sel, (t + 1) // l_sel)
        key = (p_grp_rows[r].astype(np.float32) - idx * np.float32(1e-8)).astype(np.float32)
        ok = np.zeros(S_sel, bool)
        ok[:nvalid] = True
        cb = t // l_sel
        for f in (0, cb, max(cb - 1, 0)):
            if f < S_sel:
                ok[f] = False
        k = sorted(key[ok].tolist(), reverse=True)
        kk = n_top - 3
        if kk >= 1 and len(k) > kk:
            gaps[r] = k[kk - 1] - k[kk]
    return gaps


@pytest.mark.parametrize("S", [4096, 16384, 65536])
def test_g10_ranges_from_q_k_are_exact_where_the_score_gap_allows(nv, orc, S):
    """SURVEY 7 hard part (d): from Q / K_cmp the device scores differ from the reference's by ~1e-7 (softmax is not bit-pinnable
    across devices), so a row can only flip where its 13th / 14th ranking keys are closer than that.  Gate: every row whose gap in
    the REFERENCE p_grp (g10 goldens) exceeds 4e-6 must reproduce the reference's ranges bit for bit, in both selector modes; the
    mismatch fraction of the remaining rows is printed.  fp32 inputs = the goldens' own inputs; then the bf16 MFMA scorer against the
    oracle chain evaluated on the bf16-rounded inputs, same gate."""
    import nsa_vibe_amd.selection_scorer as ss

    g = load_golden(f"g10_m7c_S{S}")
    ts = gi.g10_rows(S)
    m = nv.build_block_meta(S, 32, 16, 64, 16, 512)
    Qr, Kc = gi.g10_q_kcmp(S, ts)
    t_rows = dev(np.repeat(ts, 2).astype(np.int32))
    for tag, dtype in (("fp32 / goldens", torch.float32), ("bf16 / oracle", torch.bfloat16)):
        if dtype == torch.float32:
            pg_ref, r_seq_ref, r_bat_ref = g["p_grp"], g["r_seq"], g["r_bat"]
        else:
            om = orc.build_block_meta(S, 32, 16, 64, 16, 512)
            rb = lambda a: torch.from_numpy(a).bfloat16().float().numpy()  # noqa: E731
            _, pg_ref = orc.map_pcmp_to_pslc_and_pgrp(orc.compute_pcmp_all(rb(Qr), rb(Kc), 0.125)[0], om)
            r_seq_ref = orc.select_topn_ranges_rows(pg_ref.reshape(-1, m.S_sel), np.repeat(ts, 2), om, 16, True, 2).reshape(len(ts), 2, 16, 2)
            full = np.zeros((1, S, 2, m.S_sel), np.float32)
            full[0, ts] = pg_ref
            r_bat_ref = orc.select_topn_ranges_batched(full, om, 16, S)[0, ts]
        pg = nv.selection_scores(dev(Qr, dtype), dev(Kc, dtype), m)  # [1, rows, G, S_sel] from Q / K_cmp on the device
        err = float(np.abs(pg.cpu().numpy()[0] - pg_ref).max())
        assert err < (2e-6 if dtype == torch.float32 else 4e-6), (tag, err)
        gaps = _topn_gap(pg_ref.reshape(-1, m.S_sel), ts)
        gated = gaps > 4e-6
        rs = ss._select(pg.reshape(-1, m.S_sel), len(ts) * 2, 1, 2, 0, t_rows, m, 16, True, 2, 0, 1, 16).cpu().numpy()
        full_d = torch.zeros(1, S, 2, m.S_sel, device="cuda")
        full_d[0, dev(ts).long()] = pg[0]
        rbd = nv.select_topn_ranges_batched(full_d, m, 16, S)[0, dev(ts).long()].cpu().numpy().reshape(-1, 16, 2)
        same_seq = np.array([a == b for a, b in zip(norm(rs), norm(r_seq_ref))])
        same_bat = (rbd == r_bat_ref.reshape(-1, 16, 2)).all(axis=(-1, -2))
        print(f"S={S} {tag}: |p_grp - ref| max {err:.2e}; rows {gated.size}, gated {int(gated.sum())}; mismatching rows overall "
              f"seq {1 - same_seq.mean():.4f} bat {1 - same_bat.mean():.4f}; among ungated seq "
              f"{(1 - same_seq[~gated].mean()) if (~gated).any() else 0.0:.4f}")
        assert gated.mean() > 0.9  # the gate must not be vacuous
        assert same_seq[gated].all() and same_bat[gated].all(), tag


@pytest.mark.parametrize("S,B", [(777, 2), (4096, 2), (16500, 1)])
def test_flat_scorer_matches_four_tile_form(nv, tune, S, B):
    """h = 6: the scorer with the 8 queries of a wave on three full column tiles (queries 2 and 5 straddle two tiles) against the four-tile
    form -- the same logits, exponentials and stencil per (query, head); only the order of the six head terms of the straddling queries
    differs (fp32 rounding)"""
    import torch

    g = torch.Generator(device="cuda")
    g.manual_seed(S)
    meta = nv.build_block_meta(S, 32, 16, 64, 16, 512)
    Q = torch.randn(B, S, 2, 6, 64, device="cuda", generator=g).bfloat16()
    Kc = torch.randn(B, 2, meta.S_cmp, 64, device="cuda", generator=g).bfloat16()
    out = {}
    for form in (0, 1):
        tune("SCORES_FORM", form)
        out[form] = nv.selection_scores(Q, Kc, meta, 0.125)
    torch.cuda.synchronize()
    assert torch.isfinite(out[1]).all()
    assert (out[0] - out[1]).abs().max().item() <= 2e-6
    # queries that do not straddle (0, 1, 3, 4, 6, 7 of every 8) sum their heads in the same order: bit-identical
    keep = torch.tensor([t % 8 not in (2, 5) for t in range(S)], device="cuda")
    assert torch.equal(out[0][:, keep], out[1][:, keep])


@pytest.mark.parametrize("S,B,dt", [(70, 1, "bf16"), (100, 3, "f16"), (777, 2, "bf16"), (4096, 2, "f16"), (16500, 1, "bf16")])
def test_scorer_on_32x32_tiles_matches_the_16x16_form(nv, tune, S, B, dt):
    """h = 6, D = 64: the 32x32x16 scorer (16 queries per wave; second sweep with a lane per compressed key: head sum in the lane, stencil along
    the lanes) against the four-tile 16x16x32 form -- the same logits and exponentials; the six head terms and the five taps are added in a
    different order (fp32 rounding of sums <= 6).  Lengths that end inside a 64-query workgroup, inside a 32-row half tile and before the
    first full selection block are part of the set."""
    import torch

    g = torch.Generator(device="cuda")
    g.manual_seed(S)
    tdt = torch.bfloat16 if dt == "bf16" else torch.float16
    meta = nv.build_block_meta(S, 32, 16, 64, 16, 512)
    Q = torch.randn(B, S, 2, 6, 64, device="cuda", generator=g).to(tdt)
    Kc = torch.randn(B, 2, meta.S_cmp, 64, device="cuda", generator=g).to(tdt)
    Kc[:, :, ::7] *= 3.0  # a few dominant compressed tokens: the first sweep has to raise its reference maximum along the way
    out = {}
    for form in (0, 2):
        tune("SCORES_FORM", form)
        out[form] = nv.selection_scores(Q, Kc, meta, 0.125, variant=2)           # (variant 2: the MFMA kernels also for the short lengths)
        out[form, "skip"] = nv.selection_scores(Q, Kc, meta, 0.125, causal_skip=True, variant=2)
    torch.cuda.synchronize()
    assert torch.isfinite(out[2]).all()
    assert (out[0] - out[2]).abs().max().item() <= 4e-6
    # with the causal skip every block a selector may read at row t ((j + 1) l' <= t + 1) holds the full result; the others hold 0 or the full
    # result (the skip is per workgroup: 32 queries in the 16x16 forms, 64 here)
    tt = torch.arange(S, device="cuda")[:, None]
    jj = torch.arange(meta.S_sel, device="cuda")[None, :]
    readable = ((jj + 1) * 64 <= tt + 1)[None, :, None, :].expand_as(out[2])
    for form in (0, 2):
        sk, full = out[form, "skip"], out[form]
        assert torch.equal(sk[readable], full[readable])
        assert ((sk[~readable] == 0) | (sk[~readable] == full[~readable])).all()
    # every row is a sum of h softmax rows pushed through Eq.9: it adds up to h (up to the mass of taps that fall off the block grid: none here)
    t = S - 1
    full = out[2][:, t].sum(-1)
    assert (full - 6.0).abs().max().item() <= 1e-4


def test_scorer_on_32x32_tiles_with_logits_far_apart(nv, orc, tune):
    """logits spread over hundreds of units (one compressed token dominates a row, and which one changes along the sequence): the first sweep
    has to raise its reference maximum in the middle of a tile, rows underflow everywhere else; against the oracle on the rounded inputs"""
    import torch

    S, B, G, h, D = 1500, 1, 2, 6, 64
    rng = np.random.default_rng(77)
    m = nv.build_block_meta(S, 32, 16, 64, 16, 512)
    mo = orc.build_block_meta(S, 32, 16, 64, 16, 512)
    Q = rng.standard_normal((B, S, G, h, D), dtype=np.float32) * 4.0
    Kc = rng.standard_normal((B, G, m.S_cmp, D), dtype=np.float32) * 4.0
    Kc[:, :, 37::41] *= 4.0
    rd = lambda a: torch.from_numpy(a).to(torch.bfloat16).float().numpy()  # noqa: E731
    ref = orc.map_pcmp_to_pslc_and_pgrp(orc.compute_pcmp_all(rd(Q), rd(Kc), 1.0 / np.sqrt(D)), mo)[1]
    err = {}
    for form in (2, 0):
        tune("SCORES_FORM", form)
        got = nv.selection_scores(dev(Q, torch.bfloat16), dev(Kc, torch.bfloat16), m, variant=2).cpu().numpy()
        assert np.isfinite(got).all()
        err[form] = float(np.abs(got - ref).max())
    print("max |p_grp - oracle|:", err)
    # exponents of a few hundred carry an fp32 ulp of 3e-5 into the exponential: both forms sit at that level, neither above the other's
    assert err[2] < 6e-5 and err[0] < 6e-5 and err[2] <= 2.0 * err[0] + 1e-6


@pytest.mark.parametrize("geom", [(32, 16, 64), (16, 16, 32)])
def test_fused_decode_with_a_meta_from_before_the_first_compressed_token(nv, tune, geom):
    """a cache whose meta was built while S < l has an EMPTY Eq.9 map (no CSC entries at all) although K_cmp already holds its first token:
    the fused decode kernel must not touch the (null) tap arrays -- both with the closed-form taps and when it reads the CSC (tuning switch,
    and always for a geometry without the closed form) -- and selects what the current meta selects (only forced blocks exist this early)"""
    import torch

    l, d, l_sel = geom
    t = l - 1 + d  # two compressed tokens exist, the meta dates from before the first one
    S_now, S_old = t + 1, l - 4
    meta_now, meta_old = nv.build_block_meta(S_now, l, d, l_sel, 16, 512), nv.build_block_meta(S_old, l, d, l_sel, 16, 512)
    assert meta_old.S_cmp == 0 and meta_old.csc_rows.numel() == 0 and meta_now.S_cmp == 2
    if meta_old.S_sel != meta_now.S_sel:
        pytest.skip("the selection-block count changed in between: the reference rebuilds the meta then")
    g = torch.Generator(device="cuda")
    g.manual_seed(l)
    mk = lambda *sh: torch.randn(*sh, device="cuda", generator=g).bfloat16()  # noqa: E731
    Q, Kc, K, V = mk(2, 1, 2, 4, 64), mk(2, 2, meta_now.S_cmp, 64), mk(2, 2, S_now, 64), mk(2, 2, S_now, 64)
    outs = []
    for stencil in (1, 0):
        tune("DECODE_STENCIL", stencil)
        outs.append(nv.selection_decode_step(Q, Kc, K, V, meta_old, 16, t))
    outs.append(nv.selection_decode_step(Q, Kc, K, V, meta_now, 16, t))
    torch.cuda.synchronize()
    for O, r in outs[:2]:
        assert torch.equal(r, outs[2][1]) and torch.equal(O, outs[2][0])


def test_scorer_64bit_output_offsets(nv, orc):
    """S = 262144 with G = 2: one sequence's p_grp is S G S_sel = 2^31 elements (8 GiB), so the fused scorer must form its output offsets
    in 64 bits (`big_out`, sel_scores_mfma.hip) -- rows from t = 131072 on lie beyond a 32-bit element offset.  Sampled rows on both
    sides of that line against the oracle chain on the bf16-rounded inputs (compute_pcmp_all -> Eq.9 -> Eq.10:
    nsa/core/selection_scorer.py:42-61, 89-116; nsa_attention.py:1091), with causal_skip = 2 as the hot path calls it (entries no
    selector reads stay unwritten: compared on the readable prefix), and the ranges of those rows from the device scores through the
    batched selector against the oracle's on the same scores."""
    S, G, h, D, n = 262144, 2, 6, 64, 16
    m = nv.build_block_meta(S, 32, 16, 64, n, 512)
    om = orc.build_block_meta(S, 32, 16, 64, n, 512)
    assert m.S_sel == 4096 and S * G * m.S_sel >= 2 ** 31
    g = torch.Generator(device="cuda")
    g.manual_seed(262)
    Q = torch.randn(1, S, G, h, D, device="cuda", generator=g).bfloat16()
    Kc = torch.randn(1, G, m.S_cmp, D, device="cuda", generator=g).bfloat16()
    pg = nv.selection_scores(Q, Kc, m, causal_skip=True, leave_skipped=True)
    torch.cuda.synchronize()
    ts = np.array([63, 4097, 131071, 131072, 131073, 200000, 262100, 262143])
    Qr = Q[0, torch.from_numpy(ts).cuda()].float().cpu().numpy()[None]  # [1, rows, G, h, D]
    _, ref = orc.map_pcmp_to_pslc_and_pgrp(orc.compute_pcmp_all(Qr, Kc.float().cpu().numpy(), 0.125)[0], om)  # [rows, G, S_sel]
    got = pg[0, torch.from_numpy(ts).cuda()].cpu().numpy()
    for i, t in enumerate(ts):
        nvalid = (t + 1) // 64  # blocks a selector may read at row t
        assert nvalid >= 1 or t < 63
        assert np.abs(got[i, :, :nvalid] - ref[i, :, :nvalid]).max() < 4e-6, int(t)
    # the rows' ranges: device scores -> device selector against the oracle's selector on the same scores (bit-exact)
    import nsa_vibe_amd.selection_scorer as ss

    t_rows = dev(np.repeat(ts, G).astype(np.int32))
    rows = torch.from_numpy(got.copy()).cuda().reshape(-1, m.S_sel)
    for i, t in enumerate(ts):  # (the unwritten tail holds whatever the allocation held: blank it for the oracle, neither side reads it)
        rows[G * i:G * i + G, (t + 1) // 64:] = 0
    rs = ss._select(rows, len(ts) * G, 1, G, 0, t_rows, m, n, True, 2, 0, 1, n).cpu().numpy()
    want = orc.select_topn_ranges_rows(rows.cpu().numpy(), np.repeat(ts, G), om, n, True, 2)
    assert norm(rs) == norm(want)


@pytest.mark.parametrize("S_ctx,B", [(64, 2), (700, 3), (1041, 2), (4096, 5), (16384, 3), (20000, 3), (32768, 2), (40000, 2), (65536, 2), (65600, 2), (100001, 1),
                                     (131072, 2)])
def test_decode_step_kernel_forms_agree(nv, orc, tune, S_ctx, B):
    """the one-launch decode step (sel_decode_fused.hip) in every form -- 16 / 8 waves per row, the logits phase of a row on one
    workgroup or split over 2 / 4 / 16 (the last arriver finishes the row), or (round 4, DECODE_WIDE = 1) a long row in ONE workgroup with
    four chunks per wave and the later chunks' logits in LDS (20000 / 32768 on 8 waves, 40000 / 65536 on 16; elsewhere the switch changes
    nothing) -- against the three separate launches (DECODE_UNFUSED = 1)
    and the round-2 kernels (DECODE_STEP = 0): ranges bit-identical everywhere, O bit-identical wherever the gather uses the same
    number of waves (the partial records of a row are merged in wave order); and the ranges against the oracle's selector on the
    device scores, O against the oracle's attention.  Reference path: nsa/core/nsa_attention.py:651-672, 704-830."""
    g = torch.Generator(device="cuda")
    g.manual_seed(S_ctx * 7 + B)
    G, h, D, n = 2, 6, 64, 16
    meta = nv.build_block_meta(S_ctx, 32, 16, 64, n, 512)
    mo = orc.build_block_meta(S_ctx, 32, 16, 64, n, 512)
    mk = lambda *sh: torch.randn(*sh, device="cuda", generator=g).bfloat16()  # noqa: E731
    Q, Kc = mk(B, 1, G, h, D), mk(B, G, max(meta.S_cmp, 1), D)[:, :, : meta.S_cmp]
    K, V = mk(B, G, S_ctx + 5, D)[:, :, :S_ctx], mk(B, G, S_ctx + 5, D)[:, :, :S_ctx]  # views of a longer cache
    t = S_ctx - 1
    outs = {}
    for nw in (16, 8):
        tune("DECODE_WAVES", nw)
        tune("DECODE_STEP", 1), tune("DECODE_UNFUSED", 1)
        outs[(nw, "three launches")] = nv.selection_decode_step(Q, Kc, K, V, meta, n, t)
        tune("DECODE_UNFUSED", -1)
        for ns in (-1, 1, 2, 4, 16):
            tune("DECODE_SPLIT", ns)
            outs[(nw, f"step, split {ns}")] = nv.selection_decode_step(Q, Kc, K, V, meta, n, t)
            outs[(nw, f"step, split {ns}, again")] = nv.selection_decode_step(Q, Kc, K, V, meta, n, t)  # tickets left clean, run-to-run bits
        tune("DECODE_SPLIT", -1)
        tune("DECODE_WIDE", 1)
        outs[(nw, "step, wide")] = nv.selection_decode_step(Q, Kc, K, V, meta, n, t)
        outs[(nw, "step, wide, again")] = nv.selection_decode_step(Q, Kc, K, V, meta, n, t)
        tune("DECODE_WIDE", -1)
        tune("DECODE_STEP", 0)
        outs[(nw, "round-2 kernels")] = nv.selection_decode_step(Q, Kc, K, V, meta, n, t)
    torch.cuda.synchronize()
    O_ref, r_ref = outs[(16, "three launches")]
    for (nw, tag), (O, r) in outs.items():
        assert torch.equal(r, r_ref), (nw, tag)
        assert torch.equal(O, outs[(nw, "three launches")][0]), (nw, tag)
        assert (O.float() - O_ref.float()).abs().max().item() <= 1e-2
    p = nv.selection_scores(Q, Kc, meta)
    want = orc.select_topn_ranges(p[:, 0].cpu().numpy(), mo, n, t)
    assert norm(r_ref.cpu().numpy()) == norm(want)
    f = lambda a: a.float().cpu().numpy()  # noqa: E731
    O_or = orc.sel_attention_masked(f(Q), f(K), f(V), r_ref.cpu().numpy()[:, None])
    assert np.abs(f(O_ref) - O_or).max() <= 1e-2


@pytest.mark.parametrize("S_ctx,B,h", [(700, 3, 6), (4096, 5, 6), (16384, 4, 6), (40000, 3, 6), (65536, 4, 6), (65536, 2, 4), (100001, 2, 6), (131072, 2, 3)])
def test_decode_step_one_pass_form(nv, orc, tune, S_ctx, B, h):
    """the one-pass form of the decode step (DECODE_WIDE = 2; what the plan picks for B >= 128 at 64k, where neither the accumulators of one
    workgroup nor co-resident teams hold a row): per chunk the Eq.9 sums of exponentials relative to the chunk's own maximum, scaled to the row's
    log-sum-exp afterwards -- one exponential per logit, one more rounding per score (<= 2 ulp).  The contract for ranges computed from Q / K
    (DESIGN.md 2): every row whose 13th / 14th ranking keys are further apart than the score noise must give the ranges of the exact forms bit
    for bit (here: relative gap > 1e-5 against <= 2.4e-7 of noise; random rows all pass the gate), and with equal ranges O is bit-identical
    (same gather).  Reference path: nsa/core/nsa_attention.py:651-672, selection_scorer.py:42-61, 89-116, 124-249."""
    g = torch.Generator(device="cuda")
    g.manual_seed(S_ctx * 3 + B + h)
    G, D, n = 2, 64, 16
    meta = nv.build_block_meta(S_ctx, 32, 16, 64, n, 512)
    mk = lambda *sh: torch.randn(*sh, device="cuda", generator=g).bfloat16()  # noqa: E731
    Q, Kc, K, V = mk(B, 1, G, h, D), mk(B, G, meta.S_cmp, D), mk(B, G, S_ctx, D), mk(B, G, S_ctx, D)
    t = S_ctx - 1
    p = nv.selection_scores(Q, Kc, meta)[:, 0]  # exact scores [B,G,S_sel]
    cur = t // 64
    cand = p.clone()
    cand[..., 0] = -1
    cand[..., max(cur - 1, 0):] = -1  # forced blocks and the blocks a selector cannot read leave the ranking
    top = cand.topk(min(n - 3 + 1, cand.shape[-1]), dim=-1).values
    k_rest = n - 3
    gated = torch.ones(B, G, dtype=torch.bool, device="cuda")
    if top.shape[-1] > k_rest:
        a, b_ = top[..., k_rest - 1], top[..., k_rest]
        gated = (b_ < 0) | ((a - b_) > 1e-5 * a.abs())
    n_checked = 0
    for nw in (16, 8):
        tune("DECODE_WAVES", nw)
        tune("DECODE_WIDE", -1), tune("DECODE_UNFUSED", 1)
        O0, r0 = nv.selection_decode_step(Q, Kc, K, V, meta, n, t)
        tune("DECODE_UNFUSED", -1), tune("DECODE_WIDE", 2)
        O1, r1 = nv.selection_decode_step(Q, Kc, K, V, meta, n, t)
        O2, r2 = nv.selection_decode_step(Q, Kc, K, V, meta, n, t)
        torch.cuda.synchronize()
        assert torch.equal(r1, r2) and torch.equal(O1, O2)  # run to run
        assert torch.equal(r0[gated], r1[gated]), nw
        assert torch.equal(O0[:, 0][gated], O1[:, 0][gated]), nw
        n_checked += int(gated.sum())
    assert n_checked >= B * G  # the gate must not be vacuous (random rows: gaps of 1e-2 .. 1e-4 relative)


def test_decode_step_at_the_large_batch_64k_shape_the_plan_gives_the_one_pass_form(nv, tune):
    """B = 128 sequences at a 64k context (256 rows x 64 chunks: a team of workgroups per row would not fit the chip, so the automatic plan
    takes the one-pass form; bench.py's decode_B128_S65536 / decode_B256_S65536 lines run it): every row against the exact round-2 kernels
    (DECODE_STEP = 0: logits in LDS, two launches) -- ranges identical wherever the 13th / 14th keys are decided by more than the form's score
    noise (relative 1e-5 against <= 2.4e-7), O identical on those rows up to the different merge grouping (8 vs 16 waves: <= 1 bf16 ulp), and
    run-to-run bit-identical.  K/V 4 GiB."""
    g = torch.Generator(device="cuda")
    g.manual_seed(128)
    B, G, h, D, n, S_ctx = 128, 2, 6, 64, 16, 65536
    meta = nv.build_block_meta(S_ctx, 32, 16, 64, n, 512)
    mk = lambda *sh: torch.randn(*sh, device="cuda", generator=g, dtype=torch.bfloat16)  # noqa: E731
    Q, Kc, K, V = mk(B, 1, G, h, D), mk(B, G, meta.S_cmp, D), mk(B, G, S_ctx, D), mk(B, G, S_ctx, D)
    t = S_ctx - 1
    O1, r1 = nv.selection_decode_step(Q, Kc, K, V, meta, n, t)
    O2, r2 = nv.selection_decode_step(Q, Kc, K, V, meta, n, t)
    tune("DECODE_STEP", 0)
    O0, r0 = nv.selection_decode_step(Q, Kc, K, V, meta, n, t)
    torch.cuda.synchronize()
    assert torch.equal(r1, r2) and torch.equal(O1, O2)
    p = nv.selection_scores(Q, Kc, meta)[:, 0]
    cand = p.clone()
    cur = t // 64
    cand[..., 0] = -1
    cand[..., cur - 1:] = -1
    top = cand.topk(n - 3 + 1, dim=-1).values
    gated = (top[..., n - 4] - top[..., n - 3]) > 1e-5 * top[..., n - 4].abs()
    assert gated.float().mean().item() > 0.9
    assert torch.equal(r0[gated], r1[gated])
    assert (O0[:, 0][gated].float() - O1[:, 0][gated].float()).abs().max().item() <= 1.6e-2
    same = (r0 == r1).all(dim=-1).all(dim=-1)
    print(f"one-pass form at B=128 @64k: rows {same.numel()}, gated {int(gated.sum())}, rows with the exact form's ranges {int(same.sum())}")


@pytest.mark.parametrize("h", [1, 3, 4, 8, 16])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_decode_step_kernel_other_group_sizes(nv, tune, h, dtype):
    """the one-launch decode step with h != 6 heads per group (the head sum of Eq.10 takes its generic form: columns >= h enter as zeros)
    and f16 inputs, unsplit and split, against the three separate launches: ranges and O bit-identical"""
    g = torch.Generator(device="cuda")
    g.manual_seed(50 + h)
    B, G, D, n, S_ctx = 3, 2, 64, 16, 9000
    meta = nv.build_block_meta(S_ctx, 32, 16, 64, n, 512)
    mk = lambda *sh: torch.randn(*sh, device="cuda", generator=g).to(dtype)  # noqa: E731
    Q, Kc, K, V = mk(B, 1, G, h, D), mk(B, G, meta.S_cmp, D), mk(B, G, S_ctx, D), mk(B, G, S_ctx, D)
    t = S_ctx - 1
    tune("DECODE_UNFUSED", 1)
    O0, r0 = nv.selection_decode_step(Q, Kc, K, V, meta, n, t)
    tune("DECODE_UNFUSED", -1)
    for ns in (1, 4):
        tune("DECODE_SPLIT", ns)
        O1, r1 = nv.selection_decode_step(Q, Kc, K, V, meta, n, t)
        torch.cuda.synchronize()
        assert torch.equal(r0, r1) and torch.equal(O0, O1), (h, ns)
    if h <= 6:  # the four-chunks-per-wave form (logits of the later chunks in LDS, 256 h bytes per chunk) with the generic head sum
        S_ctx = 40000
        meta = nv.build_block_meta(S_ctx, 32, 16, 64, n, 512)
        Q, Kc, K, V = mk(B, 1, G, h, D), mk(B, G, meta.S_cmp, D), mk(B, G, S_ctx, D), mk(B, G, S_ctx, D)
        tune("DECODE_SPLIT", -1), tune("DECODE_UNFUSED", 1)
        O0, r0 = nv.selection_decode_step(Q, Kc, K, V, meta, n, S_ctx - 1)
        tune("DECODE_UNFUSED", -1), tune("DECODE_WIDE", 1)
        O1, r1 = nv.selection_decode_step(Q, Kc, K, V, meta, n, S_ctx - 1)
        torch.cuda.synchronize()
        assert torch.equal(r0, r1) and torch.equal(O0, O1), (h, "wide")


@pytest.mark.parametrize("S_ctx", [3000, 16384, 65536])
def test_decode_step_on_tie_heavy_scores(nv, orc, tune, S_ctx):
    """the one-launch decode step (scores in registers, selector fed from LDS, block list -> gather) against the three separate launches
    (DECODE_UNFUSED = 1) and the oracle's selector, on score rows full of exact ties: K_cmp = 0 (every compressed
    column has the same probability: interior blocks tie, the tie order 'lower index first' decides all 13 picks), K_cmp built from 3
    distinct rows (plateaus), a peaked softmax (most blocks exactly 0), and NaN logits in one sequence (no candidates: forced blocks only).
    The reference order (key desc, index asc): nsa/core/selection_scorer.py:182-187; PRD.md:46."""
    g = torch.Generator(device="cuda")
    g.manual_seed(S_ctx)
    B, G, h, D, n = 5, 2, 6, 64, 16
    meta = nv.build_block_meta(S_ctx, 32, 16, 64, n, 512)
    mo = orc.build_block_meta(S_ctx, 32, 16, 64, n, 512)
    mk = lambda *sh: torch.randn(*sh, device="cuda", generator=g).bfloat16()  # noqa: E731
    Q, K, V = mk(B, 1, G, h, D), mk(B, G, S_ctx, D), mk(B, G, S_ctx, D)
    Kc = torch.zeros(B, G, meta.S_cmp, D, device="cuda", dtype=torch.bfloat16)  # b = 0: all ties
    three = mk(3, D)
    Kc[1] = three[torch.randint(0, 3, (G, meta.S_cmp), device="cuda", generator=g)]  # plateaus
    Kc[2] = mk(G, meta.S_cmp, D)
    Kc[2, :, 777 % meta.S_cmp] = Q[2, 0, :, 0] * 40  # one column takes all the mass: the other blocks' scores underflow to exact zeros
    Kc[3] = mk(G, meta.S_cmp, D)  # ordinary
    Kc[4] = mk(G, meta.S_cmp, D)
    Kc[4, 0, 5] = float("nan")  # one NaN logit poisons the row's normaliser: every score NaN, no candidate
    t = S_ctx - 1
    tune("DECODE_UNFUSED", 1)
    O0, r0 = nv.selection_decode_step(Q, Kc, K, V, meta, n, t)
    tune("DECODE_UNFUSED", -1)
    for nw in (16, 8):
        tune("DECODE_WAVES", nw)
        for ns in (-1, 1, 4):
            tune("DECODE_SPLIT", ns)
            O1, r1 = nv.selection_decode_step(Q, Kc, K, V, meta, n, t)
            torch.cuda.synchronize()
            assert torch.equal(r0, r1), (nw, ns)
    tune("DECODE_WAVES", -1), tune("DECODE_SPLIT", -1)
    # the one-pass form (DECODE_WIDE = 2) on the rows its contract covers: the peaked softmax (most scores exact zeros in both forms), the
    # ordinary row and the NaN row (every score NaN: no candidate, forced blocks only) must give the exact forms' ranges; the all-tie and
    # plateau rows are what only the exact forms guarantee (their scores are equal up to the last bit, which the extra rounding can move)
    if meta.S_cmp <= 64 * 8 * 8:
        tune("DECODE_WIDE", 2)
        for nw in (16, 8):
            tune("DECODE_WAVES", nw)
            O2, r2 = nv.selection_decode_step(Q, Kc, K, V, meta, n, t)
            torch.cuda.synchronize()
            assert torch.equal(r0[2:], r2[2:]), nw
        tune("DECODE_WIDE", -1), tune("DECODE_WAVES", -1)
    O1, r1 = nv.selection_decode_step(Q, Kc, K, V, meta, n, t)
    fin = torch.isfinite(O0.float()).all(dim=-1).all(dim=-1)
    assert torch.equal(O0[fin], O1[fin])
    # against the oracle's selector on the device scores (rows whose scores are NaN: compared among the kernels above only)
    p = nv.selection_scores(Q, Kc, meta)[:, 0].cpu().numpy()
    ok = np.isfinite(p).all(axis=-1)
    want = orc.select_topn_ranges(np.where(ok[..., None], p, 0.0).astype(np.float32), mo, n, t)
    got = r1.cpu().numpy()
    for b in range(B):
        for gg in range(G):
            if ok[b, gg]:
                assert norm(got[b, gg][None]) == norm(want[b, gg][None]), (b, gg)
    assert got[0, 0].tolist()[:3] == [[0, 64 * 14], [64 * (t // 64 - 1), t + 1], [0, 0]] or S_ctx < 64 * 16  # all ties: blocks 1..13 win


@pytest.mark.parametrize("S_ctx,B", [(16384, 2), (65536, 3)])
def test_decode_step_team_that_never_assembles(nv, tune, S_ctx, B):
    """split decode step with a poll budget of zero (DECODE_TEAM_SPIN = 0): every workgroup finds its team incomplete and forms the records of
    the whole row itself -- the path that keeps a workgroup from ever depending on another one being scheduled (two streams or processes
    sharing the CUs).  Same instructions on the same data: ranges and O bit-identical to the assembled team and to the three launches."""
    g = torch.Generator(device="cuda")
    g.manual_seed(S_ctx + B)
    G, h, D, n = 2, 6, 64, 16
    meta = nv.build_block_meta(S_ctx, 32, 16, 64, n, 512)
    mk = lambda *sh: torch.randn(*sh, device="cuda", generator=g).bfloat16()  # noqa: E731
    Q, Kc, K, V = mk(B, 1, G, h, D), mk(B, G, meta.S_cmp, D), mk(B, G, S_ctx, D), mk(B, G, S_ctx, D)
    t = S_ctx - 1
    tune("DECODE_UNFUSED", 1)
    O0, r0 = nv.selection_decode_step(Q, Kc, K, V, meta, n, t)
    tune("DECODE_UNFUSED", -1)
    for nw in (16, 8):
        tune("DECODE_WAVES", nw)
        for ns in (2, 4, 16):
            tune("DECODE_SPLIT", ns)
            for spin in (0, -1):
                tune("DECODE_TEAM_SPIN", spin)
                O1, r1 = nv.selection_decode_step(Q, Kc, K, V, meta, n, t)
                O2, r2 = nv.selection_decode_step(Q, Kc, K, V, meta, n, t)  # the tickets were left clean
                torch.cuda.synchronize()
                assert torch.equal(r0, r1) and torch.equal(r0, r2), (nw, ns, spin)
                assert torch.equal(O1, O2), (nw, ns, spin)
                if nw == 16:
                    assert torch.equal(O0, O1), (nw, ns, spin)


@pytest.mark.parametrize("S,B,mode", [(4096, 3, "batched"), (4096, 2, "sequential"), (700, 2, "batched"), (16384, 2, "batched"), (65536, 1, "batched"),
                                      (65536, 1, "sequential"), (4100, 1, "batched")])
def test_scores_and_select_in_one_launch_equals_the_two_launches(nv, orc, tune, S, B, mode):
    """nsa_sel_scores_select (round 4): on the 32x32x16 scorer's route the top-n selection of a query tile runs inside the scorer launch -- the
    select kernel's own row function on the scores the workgroup has just written -- so the ranges must be those of selection_scores ->
    select_topn_ranges_batched / _rows bit for bit, and the scores the same bits wherever a selector reads them; SCORES_SELECT = 0 (two
    launches behind the same entry point) as well.  Reference: nsa/core/nsa_attention.py:1088-1108, 1566-1576."""
    g = torch.Generator(device="cuda")
    g.manual_seed(S + B)
    G, h, D, n = 2, 6, 64, 16
    meta = nv.build_block_meta(S, 32, 16, 64, n, 512)
    Q = torch.randn(B, S, G, h, D, device="cuda", generator=g).bfloat16()
    Kc = torch.randn(B, G, meta.S_cmp, D, device="cuda", generator=g).bfloat16()
    p0 = nv.selection_scores(Q, Kc, meta, causal_skip=True)
    r0 = nv.select_topn_ranges_batched(p0, meta, n, S) if mode == "batched" else nv.select_topn_ranges_rows(p0, meta, n, 0)
    for sw in (1, 0, -1):  # in the launch / its own launch / by context length
        tune("SCORES_SELECT", sw)
        p1, r1 = nv.selection_scores_select(Q, Kc, meta, n, mode=mode)
        p2, r2 = nv.selection_scores_select(Q, Kc, meta, n, mode=mode)
        torch.cuda.synchronize()
        assert torch.equal(r0, r1) and torch.equal(r1, r2), (sw, mode)
        # scores: equal where a selector may read them (row t: blocks j with (j + 1) * 64 <= t + 1)
        t = torch.arange(S, device="cuda")[None, :, None, None]
        j = torch.arange(meta.S_sel, device="cuda")[None, None, None, :]
        ok = (j + 1) * 64 <= t + 1
        assert torch.equal(torch.where(ok, p0, 0), torch.where(ok, p1, 0)), sw


def test_scores_and_select_falls_back_to_two_launches_off_the_mfma32_route(nv, tune):
    """h = 4 (the 16x16 scorer) and fp32 inputs (the generic scorer): nsa_sel_scores_select runs scorer + select kernel back to back, same results"""
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    for h, dtype in ((4, torch.bfloat16), (6, torch.float32)):
        B, S, G, D, n = 2, 1500, 2, 64, 16
        meta = nv.build_block_meta(S, 32, 16, 64, n, 512)
        Q = torch.randn(B, S, G, h, D, device="cuda", generator=g).to(dtype)
        Kc = torch.randn(B, G, meta.S_cmp, D, device="cuda", generator=g).to(dtype)
        p0 = nv.selection_scores(Q, Kc, meta, causal_skip=True)
        r0 = nv.select_topn_ranges_batched(p0, meta, n, S)
        p1, r1 = nv.selection_scores_select(Q, Kc, meta, n)
        torch.cuda.synchronize()
        assert torch.equal(r0, r1), (h, dtype)
