"""CPU: pins the oracle (oracle/nsa_oracle.c) against the golden vectors produced by the
imported reference (oracle/make_goldens.py).  No GPU, no reference needed at run time."""
import numpy as np
import pytest

import golden_inputs as gi
from conftest import load_golden


def norm(orc, r):
    return orc.normalise_ranges(np.asarray(r))


def test_g1_block_meta(orc):
    g = load_golden("g1_block_meta")
    for i, (S, l, d, ls) in enumerate(g["cases"]):
        m = orc.build_block_meta(int(S), int(l), int(d), int(ls), 16, 512)
        assert np.array_equal(m.cmp_starts, g[f"c{i}_cmp_starts"])
        assert np.array_equal(m.sel_starts, g[f"c{i}_sel_starts"])
        assert np.array_equal(m.M_csl_indptr, g[f"c{i}_indptr"])
        assert np.array_equal(m.M_csl_indices, g[f"c{i}_indices"])
        assert np.array_equal(m.M_csl_values, g[f"c{i}_values"])  # bit-exact fp32 weights
        assert np.array_equal(m.M_csl_coo_indices, g[f"c{i}_coo"])


def test_divisibility_guards(orc):
    # reference test_block_math.py:43-47
    with pytest.raises(ValueError):
        orc.build_block_meta(1024, 30, 16, 64, 16, 512)
    with pytest.raises(ValueError):
        orc.build_block_meta(1024, 32, 12, 60, 16, 512)


def test_g2_tiebreak(orc):
    g = load_golden("g2_tiebreak")
    m = orc.build_block_meta(64, 4, 2, 4, 8, 8)
    S_sel = m.sel_starts.size
    assert norm(orc, orc.select_topn_ranges(np.ones((1, 1, S_sel)), m, 3, 63, False, 0)) == norm(orc, g["r_seq"])
    rb = orc.select_topn_ranges_batched(np.ones((1, 3, 1, S_sel)), m, 3, 3, False, 0)
    assert np.array_equal(rb, g["r_bat"])
    m2 = orc.build_block_meta(1024, 32, 16, 64, 16, 512)
    r = orc.select_topn_ranges(np.ones((1, 1, 16)), m2, 3, 1023, False, 0)
    assert norm(orc, r) == norm(orc, g["r_seq2"]) == [[(0, 192)]]


def test_g3_v2_converter(orc):
    g = load_golden("g3_v2_converter")
    m = orc.build_block_meta(1024, 32, 16, 64, 16, 512)
    names = sorted(k[:-4] for k in g.files if k.endswith("_idx"))
    assert len(names) >= 8
    for n in names:
        idx = g[n + "_idx"]
        out = orc.convert_indices_to_ranges_batched_v2(idx, m, idx.shape[1])
        assert np.array_equal(out, g[n + "_ranges"]), n


@pytest.mark.parametrize("S", [4096, 65536])
def test_g4_needle(orc, S):
    g = load_golden("g4_needle")
    m = orc.build_block_meta(S, 32, 16, 64, 8, 512)
    p = np.zeros((1, 2, 1, m.cmp_starts.size), np.float32)
    p[..., int(g[f"S{S}_cmp_row"])] = 1.0
    _, pg = orc.map_pcmp_to_pslc_and_pgrp(p, m)
    assert np.array_equal(pg, g[f"S{S}_p_grp"])
    r = orc.select_topn_ranges(pg, m, 8, S - 1, True, 2)
    assert norm(orc, r) == norm(orc, g[f"S{S}_ranges"])
    # the needle (S//2) is covered in both groups (test_long_context_needle.py:45-50)
    for row in norm(orc, r):
        assert any(s <= S // 2 < e for s, e in row)


@pytest.mark.parametrize("case", ["g5", "g6", "g7", "g8", "g8b"])
def test_g5_8_attention(orc, case):
    g = load_golden("g5_8_attention")
    Q, K, V, rg = getattr(gi, case + "_inputs")()
    for a, b in ((Q, g[case + "_Q"]), (K, g[case + "_K"]), (V, g[case + "_V"]), (rg, g[case + "_ranges"])):
        assert np.array_equal(a, b)  # numpy stream stability: recipe == stored inputs
    O = orc.sel_attention_masked(Q, K, V, rg)
    assert np.abs(O - g[case + "_O"]).max() < 1e-5
    if case == "g6":
        assert not O.any()


def test_g9_seq_vs_batched(orc):
    g = load_golden("g9_seq_vs_batched")
    S = 4096
    m = orc.build_block_meta(S, 32, 16, 64, 16, 512)
    p = gi.g9_scores(S)
    assert np.array_equal(orc.select_topn_ranges_batched(p, m, 16, S, True, 2), g["r_batched"])
    for i, t in enumerate(g["ts"]):
        r = orc.select_topn_ranges(p[:, t], m, 16, int(t), True, 2)
        assert norm(orc, r) == norm(orc, g["r_seq"][i]), t
    # documented divergence: sequential keeps the clamped partial block, batched drops it
    i = list(g["ts"]).index(1500)
    assert norm(orc, g["r_seq"][i])[0][-1][1] == 1501
    assert norm(orc, g["r_batched"][0, 1500])[0][-1][1] == 1472


def test_g9_small_forced_columns(orc):
    g = load_golden("g9_small_forced_cols")
    for key in g.files:
        S, n = (int(x[1:]) for x in key.split("_"))
        m = orc.build_block_meta(S, 32, 16, 64, 16, 512)
        ps = gi.g9_scores_small(S, m.sel_starts.size)
        out = orc.select_topn_ranges_batched(ps, m, n, S, True, 2)
        assert out.shape == g[key].shape and np.array_equal(out, g[key]), key


@pytest.mark.parametrize("S", [4096, 16384, 65536])
def test_g10_m7c_chain(orc, S):
    g = load_golden(f"g10_m7c_S{S}")
    ts = gi.g10_rows(S)
    assert np.array_equal(ts, g["ts"])
    m = orc.build_block_meta(S, 32, 16, 64, 16, 512)
    S_sel = m.sel_starts.size
    Qr, Kc = gi.g10_q_kcmp(S, ts)
    pin = slice(0, None, max(1, len(ts) // 8))
    p_cmp = orc.compute_pcmp_all(Qr[:, pin], Kc, 1.0 / 8.0)
    assert np.abs(p_cmp[0] - g["p_cmp_pin"]).max() < 1e-6
    # bit-exact Eq.9 + Eq.10 chain given the reference's p_cmp
    _, pg = orc.map_pcmp_to_pslc_and_pgrp(g["p_cmp_pin"], m)
    assert np.array_equal(pg, g["p_grp_pin"])
    # bit-exact ranges given the reference's p_grp
    rs = orc.select_topn_ranges_rows(g["p_grp"].reshape(-1, S_sel), np.repeat(ts, 2), m, 16)
    assert norm(orc, rs) == norm(orc, g["r_seq"])
    full = np.zeros((1, S, 2, S_sel), np.float32)
    full[0, ts] = g["p_grp"]
    rb = orc.select_topn_ranges_batched(full, m, 16, S)[0, ts]
    assert np.array_equal(rb, g["r_bat"])
    if S <= 16384:  # attention oracle at 64k rows is exercised on the GPU box (keeps the CPU suite short)
        K, V = gi.g10_kv(S)
        O = orc.sel_attention_masked(Qr, K, V, g["r_bat"][None])
        assert np.abs(O[0] - g["O_bat"]).max() < 1e-5


def test_g11_small_chains(orc):
    g = load_golden("g11_small_chains")
    for ci, (S, l, d, ls, n_top, G, h, D) in enumerate(g["cfgs"]):
        S, l, d, ls, n_top, G, h, D = (int(x) for x in (S, l, d, ls, n_top, G, h, D))
        m = orc.build_block_meta(S, l, d, ls, n_top, 512)
        Q, Kc, K, V = gi.g11_inputs(ci, S, G, h, D, m.cmp_starts.size)
        p_cmp = orc.compute_pcmp_all(Q, Kc, 1.0 / np.sqrt(D))
        assert np.abs(p_cmp - g[f"c{ci}_p_cmp"]).max() < 1e-6
        p_slc, p_grp = orc.map_pcmp_to_pslc_and_pgrp(g[f"c{ci}_p_cmp"], m)
        assert np.array_equal(p_slc, g[f"c{ci}_p_slc"])  # Eq.9 bit-exact
        if bool(g[f"c{ci}_pgrp_bitexact"]):
            assert np.array_equal(p_grp, g[f"c{ci}_p_grp"])  # Eq.10 bit-exact
        else:  # torch's CPU reduction order is shape dependent for small inner dims (see make_goldens.py)
            assert np.allclose(p_grp, g[f"c{ci}_p_grp"], rtol=3e-7, atol=0)
        cut = int(g[f"c{ci}_cut"])
        _, pgc = orc.map_pcmp_to_pslc_and_pgrp(g[f"c{ci}_p_cmp"][..., :cut], m)
        assert np.allclose(pgc, g[f"c{ci}_p_grp_cut"], rtol=3e-7, atol=0)
        rb = orc.select_topn_ranges_batched(g[f"c{ci}_p_grp"], m, n_top, S)
        assert np.array_equal(rb, g[f"c{ci}_r_bat"])
        O = orc.sel_attention_masked(Q, K, V, g[f"c{ci}_r_bat"])
        assert np.abs(O - g[f"c{ci}_O"]).max() < 1e-5
        for i, t in enumerate(g[f"c{ci}_ts"]):
            r = orc.select_topn_ranges(g[f"c{ci}_p_grp"][:, t], m, n_top, int(t), True, 2)
            assert norm(orc, r) == norm(orc, g[f"c{ci}_r_seq"][i])


def test_attention_bwd_oracle_matches_finite_difference(orc):
    rng = np.random.default_rng(3)
    B, S, G, h, D, S_kv = 1, 2, 1, 2, 8, 12
    Q, K, V = (rng.standard_normal(s).astype(np.float32) for s in ((B, S, G, h, D), (B, G, S_kv, D), (B, G, S_kv, D)))
    rg = np.array([[[[[0, 4], [6, 9]]], [[[2, 12], [0, 0]]]]], np.int32)
    dO = rng.standard_normal((B, S, G, h, D)).astype(np.float32)
    dQ, dK, dV = orc.sel_attention_masked_bwd(Q, K, V, rg, dO)

    def loss(q, k, v):
        return float((orc.sel_attention_masked(q, k, v, rg).astype(np.float64) * dO).sum())

    eps = 1e-2
    for arr, grad in ((Q, dQ), (K, dK), (V, dV)):
        for _ in range(6):
            idx = tuple(rng.integers(0, s) for s in arr.shape)
            a1, a2 = arr.copy(), arr.copy()
            a1[idx] += eps
            a2[idx] -= eps
            args1 = [a1 if x is arr else x for x in (Q, K, V)]
            args2 = [a2 if x is arr else x for x in (Q, K, V)]
            fd = (loss(*args1) - loss(*args2)) / (2 * eps)
            assert abs(fd - grad[idx]) < 5e-3 * max(1.0, abs(fd))


@pytest.mark.parametrize("name", ["a", "b", "c", "d"])
def test_g13_sliding_window(orc, name):
    """oracle restatement == reference sliding_window_attention (attention_kernels.py:146-178)"""
    g = load_golden("g13_win_" + name)
    O = orc.sliding_window_attention(g["Q"], g["K"], g["V"], int(g["w"]))
    assert np.abs(O - g["O"]).max() <= 2e-5


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_g13_compressed(orc, name):
    """oracle == SDPA under the reference's num_cmp mask (attention_kernels.py:118-123); band_ranges == num_cmp"""
    g = load_golden("g13_cmp_" + name)
    S, S_cmp = g["Q"].shape[1], g["K"].shape[2]
    rg = orc.band_ranges(S, S_cmp, 0, int(g["l"]), int(g["d"]), 1)
    assert np.array_equal(rg[:, 1], g["num_cmp"]) and not rg[:, 0].any()
    O = orc.batched_causal_attention_compressed(g["Q"], g["K"], g["V"], int(g["l"]), int(g["d"]))
    assert np.abs(O - g["O"]).max() <= 2e-5
    # ... and == the REFERENCE's masked selection executor run on the compressed tokens with the range [0, num_cmp(t)) per row
    # (grouped_selection_attention_masked, attention_kernels.py:705-772): the compressed branch is pinned to a reference function
    assert np.abs(O - g["O_ref_selection_masked"]).max() <= 2e-5
    if name == "b":
        assert not O.any()


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_g15_first_key_parity_mode(orc, name):
    """oracle restatement of the reference's packed / gather executors (attention_kernels.py:181-226, 273-388): V at the first key"""
    g = load_golden("g15_first_key_" + name)
    assert np.array_equal(orc.sel_attention_first_key_parity(g["Q"], g["V"], g["ranges"]), g["O"])


@pytest.mark.parametrize("name", ["ref_test_shape", "a", "b", "c"])
def test_g16_backward_oracle_matches_reference_autograd(orc, name):
    """oracle backward == torch autograd through the reference's grouped_selection_attention_masked (attention_kernels.py:705-772),
    taken as nsa/tests/test_selection_backward_reference.py:35-37 takes it; tolerance = that test's 1e-5"""
    g = load_golden("g16_bwd_" + name)
    dQ, dK, dV = orc.sel_attention_masked_bwd(g["Q"], g["K"], g["V"], g["ranges"], g["dO"])
    for got, want in ((dQ, g["dQ"]), (dK, g["dK"]), (dV, g["dV"])):
        assert np.allclose(got, want, atol=1e-5, rtol=1e-5)
    assert np.abs(orc.sel_attention_masked(g["Q"], g["K"], g["V"], g["ranges"]) - g["O"]).max() <= 1e-5
    if name != "ref_test_shape":
        assert not g["dQ"][0, 0, 0].any()  # the row without a token has no gradient in the reference either


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_g17_head_causal_parity_mode(orc, name):
    """oracle restatement of NSAAttention._sdpa_over_ranges (nsa_attention.py:1779-1855): head i sees the first i+1 gathered tokens"""
    g = load_golden("g17_head_causal_" + name)
    O = orc.sel_attention_head_causal_parity(g["Q"][:, None], g["K"], g["V"], g["ranges"][:, None])[:, 0]
    assert np.abs(O - g["O"]).max() <= 1e-5
    assert not O[0, 0].any()


def test_g19_recipe_and_fixture_are_consistent():
    """g19 (reference module at the m7c geometry): the fixture holds outputs only, weights / inputs are regenerated from the PCG64
    recipes -- check the recipe is bf16-representable and matches the fixture's bookkeeping (names, shapes, sampled rows)"""
    import torch

    g = load_golden("g19_m7c_module")
    names_shapes = [(str(nm), tuple(int(x) for x in sh if x > 0)) for nm, sh in zip(g["names"], g["shapes"])]
    assert {"W_Q.weight", "W_K_sel.weight", "W_V_sel.weight", "out.weight", "gate.fc1.weight", "gate.fc2.bias"} <= {n for n, _ in names_shapes}
    state = gi.g19_state(names_shapes)
    for k, v in state.items():
        assert np.array_equal(torch.from_numpy(v).bfloat16().float().numpy(), v), k
    assert state["W_Q.weight"].shape == (768, 768) and state["W_K_sel.weight"].shape == (128, 768)
    x_pre, x_dec = gi.g19_inputs()
    assert x_pre.shape == (1, 4096, 768) and x_dec.shape == (2200, 1, 1, 768)
    assert np.array_equal(torch.from_numpy(x_pre).bfloat16().float().numpy(), x_pre)
    rows_pre, rows_dec = gi.g19_rows()
    assert np.array_equal(rows_pre, g["rows_pre"]) and np.array_equal(rows_dec, g["rows_dec"])
    assert g["out_pre_seq"].shape == (1, len(rows_pre), 768) and g["out_dec"].shape == (len(rows_dec), 1, 1, 768)
    assert np.isfinite(g["out_pre_seq"]).all() and np.isfinite(g["out_pre_bat"]).all() and np.isfinite(g["out_dec"]).all()
    # round 4: the reference's own selection per sampled row (ranges + the 13th / 14th ranking-key gap of its scores)
    for tag, rows, W in (("pre_seq", rows_pre, 16), ("pre_bat", rows_pre, 16), ("dec", rows_dec, 16)):
        assert g[f"ranges_{tag}"].shape == (len(rows), 2, W, 2) and g[f"gap_{tag}"].shape == (len(rows), 2)
        assert (g[f"gap_{tag}"] >= 0).all()
    # the reference's ranges obey the path's invariants: block 0 forced, nothing beyond t + 1, live ranges ascending and disjoint
    for tag, rows in (("pre_seq", rows_pre), ("dec", rows_dec)):
        for i, t in enumerate(rows):
            for gg in range(2):
                live = [(int(a), int(b)) for a, b in g[f"ranges_{tag}"][i, gg] if b > a]
                assert live and live[0][0] == 0 and live[-1][1] == t + 1, (tag, t)
                assert all(live[k][1] < live[k + 1][0] for k in range(len(live) - 1)), (tag, t)


def test_g20_tiny_bench_shape_selection_matches_the_oracle_selector(orc):
    """BASELINE configs[0] at the exact defaults of bench/bench_decode.py:63-72 (g20 = the reference MODULE at that shape): with n = 16 >= the
    8 selection blocks of a 512-token context the selection takes every complete block, whatever the scores -- the oracle's two selectors on
    arbitrary scores must give the ranges the reference module produced inside its forward, row for row (sequential: :124-249; batched:
    :255-362 + :434-605), and the recipe must be bf16-representable like g19's"""
    import torch

    g = load_golden("g20_tiny_bench_module")
    assert tuple(int(x) for x in g["cfg"]) == (256, 8, 2, 32, 32, 32, 16, 64, 16, 512)
    S = gi.G20_S_PRE
    meta = orc.build_block_meta(S, 32, 16, 64, 16, 512)
    S_sel = meta.sel_starts.size
    p = np.random.default_rng(20).random((1, S, 2, S_sel), dtype=np.float32)
    bat = orc.select_topn_ranges_batched(p, meta, 16, S)[0]
    assert bat.shape == g["ranges_pre_bat"].shape
    live = lambda r: [(int(a), int(b)) for a, b in r if b > a]  # noqa: E731
    for t in range(S):
        seq = orc.select_topn_ranges(p[:, t], meta, 16, t)[0]
        for gg in range(2):
            assert live(seq[gg]) == live(g["ranges_pre_seq"][t, gg]), t
            assert live(bat[t, gg]) == live(g["ranges_pre_bat"][t, gg]), t
    names_shapes = [(str(nm), tuple(int(x) for x in sh if x > 0)) for nm, sh in zip(g["names"], g["shapes"])]
    for k, v in gi.g20_state(names_shapes).items():
        assert np.array_equal(torch.from_numpy(v).bfloat16().float().numpy(), v), k
    assert g["out_pre_seq"].shape == (1, S, 256) and g["out_dec"].shape == (gi.G20_N_DEC, 1, 1, 256)
