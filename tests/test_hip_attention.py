"""GPU parity: selection attention (HIP, through the C ABI) vs the oracle and the golden vectors.

Tolerances (BASELINE.json north_star): |O - ref| <= 1e-3 for fp32 inputs, <= 1e-2 for bf16/fp16
(reference = the masked-SDPA semantics, nsa/core/attention_kernels.py:705-772)."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
from conftest import load_golden

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 1e-3, torch.bfloat16: 1e-2, torch.float16: 1e-2}


@pytest.fixture(scope="module")
def nv():
    import nsa_vibe_amd

    assert torch.cuda.is_available(), "GPU tests need the MI355X box"
    return nsa_vibe_amd


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.to(dtype) if dtype is not None else t


def rounded(a, dtype):
    """numpy fp32 array rounded through `dtype` (what the kernel actually sees)."""
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).float().numpy()


def run_case(nv, orc, Q, K, V, rg, dtype, variant, tol=None):
    qd, kd, vd = dev(Q, dtype), dev(K, dtype), dev(V, dtype)
    O, lse = nv.selection_attention_hip(qd, kd, vd, dev(rg), variant=variant, return_lse=True)
    assert O.dtype == dtype and O.shape == (*Q.shape[:4], V.shape[3])
    ref, ref_lse = orc.sel_attention_masked(rounded(Q, dtype), rounded(K, dtype), rounded(V, dtype), rg, return_lse=True)
    got = O.float().cpu().numpy()
    assert np.isfinite(got).all()
    err = np.abs(got - ref).max()
    assert err <= (tol or TOL[dtype]), f"max|dO|={err:.3e} dtype={dtype} variant={variant}"
    fin = np.isfinite(ref_lse)
    lg = lse.cpu().numpy()
    assert np.array_equal(np.isfinite(lg), fin)
    if fin.any():
        assert np.abs(lg[fin] - ref_lse[fin]).max() <= 2e-2
    return got


@pytest.mark.parametrize("case", ["g5", "g6", "g7", "g8", "g8b"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_golden_cases_generic(nv, orc, case, dtype):
    g = load_golden("g5_8_attention")
    Q, K, V, rg = getattr(gi, case + "_inputs")()
    got = run_case(nv, orc, Q, K, V, rg, dtype, variant=1)
    if dtype == torch.float32:  # directly against the reference's own output
        assert np.abs(got - g[case + "_O"]).max() <= 1e-3
    if case == "g6":
        assert not got.any()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_golden_g8_mfma(nv, orc, dtype):
    g = load_golden("g5_8_attention")
    Q, K, V, rg = gi.g8_inputs()  # D=64, h=2, multi-span
    got = run_case(nv, orc, Q, K, V, rg, dtype, variant=2)
    if dtype == torch.bfloat16:  # the reference's own bf16 result
        assert np.abs(got - g["g8_O_bf16"]).max() <= 1e-2


def test_ranges_int64_not_mutated(nv):
    Q, K, V, rg = gi.g7_inputs()
    r64 = dev(rg).long()
    keep = r64.clone()
    nv.selection_attention_hip(dev(Q), dev(K), dev(V), r64)
    assert torch.equal(r64, keep)  # the reference clamps an int64 ranges tensor in place; we must not


def _rand_ranges(rng, B, S, G, n, S_kv, aligned=False):
    rg = np.zeros((B, S, G, n, 2), np.int32)
    for idx in np.ndindex(B, S, G):
        k = rng.integers(0, n + 1)
        for i in range(k):
            if aligned:
                s = int(rng.integers(0, max(1, S_kv // 64))) * 64
                e = min(S_kv, s + 64 * int(rng.integers(1, 3)))
            else:
                s = int(rng.integers(0, S_kv))
                e = int(min(S_kv, s + rng.integers(0, 150)))
            rg[idx][i] = (s, e)
    return rg


@pytest.mark.parametrize("h,D", [(6, 64), (1, 64), (16, 64), (4, 128), (12, 128)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_mfma_random_shapes(nv, orc, h, D, dtype):
    rng = np.random.default_rng([h, D])
    B, S, G, n, S_kv = 2, 9, 2, 7, 700
    Q = rng.standard_normal((B, S, G, h, D), dtype=np.float32)
    K = rng.standard_normal((B, G, S_kv, D), dtype=np.float32)
    V = rng.standard_normal((B, G, S_kv, D), dtype=np.float32)
    rg = _rand_ranges(rng, B, S, G, n, S_kv)
    rg[0, 0, 0] = 0  # an empty row
    rg[0, 1, 0, 0] = (5, 6)  # a single token
    rg[0, 2, 0, :2] = [(0, 700), (100, 200)]  # whole sequence + contained range
    run_case(nv, orc, Q, K, V, rg, dtype, variant=2)
    run_case(nv, orc, Q, K, V, rg, dtype, variant=1)


@pytest.mark.parametrize("spike_at", [0, 40, 300, 650])
def test_mfma_deferred_max_branch(nv, orc, spike_at):
    """The MFMA kernel raises its running max only when a tile exceeds it by > 8 (log2 units).  Force that
    branch at a chosen tile (one K row aligned with Q, logit ~ +40 above the rest), and cases where the
    max never grows after the first tile / grows in the last tail tile (cdna guide rule 26)."""
    rng = np.random.default_rng(spike_at)
    B, S, G, h, D, S_kv = 1, 4, 1, 6, 64, 700
    Q = rng.standard_normal((B, S, G, h, D), dtype=np.float32)
    K = rng.standard_normal((B, G, S_kv, D), dtype=np.float32) * 0.3
    V = rng.standard_normal((B, G, S_kv, D), dtype=np.float32)
    K[0, 0, spike_at] = Q[0, 0, 0, 0] * 5.0  # head 0 of row 0 gets a huge logit at this key; other heads random
    K[0, 0, 699] = Q[0, 1, 0, 3] * 4.0  # row 1 / head 3 spikes in the masked tail tile
    rg = np.zeros((B, S, G, 3, 2), np.int32)
    rg[:, :, :, 0] = (0, 128)
    rg[:, :, :, 1] = (256, 448)
    rg[:, :, :, 2] = (600, 700)  # 100 tokens: 3 full tiles + a 4-key tail
    run_case(nv, orc, Q, K, V, rg, torch.bfloat16, variant=2)
    run_case(nv, orc, Q, K, V, rg, torch.float16, variant=2)


def test_mfma_matches_generic_many_rows(nv):
    """R large enough that no split-KV is used; aligned 64-token blocks like the real selector."""
    rng = np.random.default_rng(77)
    B, S, G, h, D, n, S_kv = 2, 600, 2, 6, 64, 16, 2048
    Q = torch.from_numpy(rng.standard_normal((B, S, G, h, D), dtype=np.float32)).cuda().bfloat16()
    K = torch.from_numpy(rng.standard_normal((B, G, S_kv, D), dtype=np.float32)).cuda().bfloat16()
    V = torch.from_numpy(rng.standard_normal((B, G, S_kv, D), dtype=np.float32)).cuda().bfloat16()
    rg = dev(_rand_ranges(rng, B, S, G, n, S_kv, aligned=True))
    O2 = nv.selection_attention_hip(Q, K, V, rg, variant=2).float()
    O1 = nv.selection_attention_hip(Q, K, V, rg, variant=1).float()
    assert (O2 - O1).abs().max().item() <= 1e-2


def test_split_kv_few_rows(nv, orc):
    """decode shape: B*G rows only -> tiles of a row are split over several waves + combine."""
    rng = np.random.default_rng(5)
    B, S, G, h, D, n, S_kv = 2, 1, 2, 6, 64, 16, 4096
    Q = rng.standard_normal((B, S, G, h, D), dtype=np.float32)
    K = rng.standard_normal((B, G, S_kv, D), dtype=np.float32)
    V = rng.standard_normal((B, G, S_kv, D), dtype=np.float32)
    rg = _rand_ranges(rng, B, S, G, n, S_kv, aligned=True)
    rg[1, 0, 1] = 0  # empty row through the split path
    rg[0, 0, 1, :] = 0
    rg[0, 0, 1, 0] = (4000, 4001)  # fewer tiles than splits
    run_case(nv, orc, Q, K, V, rg, torch.bfloat16, variant=2)


def test_strided_cache_view(nv, orc):
    """K/V passed as views of a preallocated [B,G,S_max,D] cache (no copy; NSA_KV layout)."""
    rng = np.random.default_rng(9)
    B, S, G, h, D, S_kv, S_max = 1, 5, 2, 6, 64, 300, 512
    Q = rng.standard_normal((B, S, G, h, D), dtype=np.float32)
    Kf = rng.standard_normal((B, G, S_max, D), dtype=np.float32)
    Vf = rng.standard_normal((B, G, S_max, D), dtype=np.float32)
    rg = _rand_ranges(rng, B, S, G, 5, S_kv)
    for dtype, variant in ((torch.bfloat16, 2), (torch.float32, 1)):
        Kc, Vc = dev(Kf, dtype), dev(Vf, dtype)
        O = nv.selection_attention_hip(dev(Q, dtype), Kc[:, :, :S_kv], Vc[:, :, :S_kv], dev(rg), variant=variant)
        ref = orc.sel_attention_masked(rounded(Q, dtype), rounded(Kf[:, :, :S_kv], dtype), rounded(Vf[:, :, :S_kv], dtype), rg)
        assert np.abs(O.float().cpu().numpy() - ref).max() <= TOL[dtype]


def test_degenerate_sizes(nv):
    z = nv.selection_attention_hip(torch.zeros(1, 2, 1, 2, 8).cuda(), torch.zeros(1, 1, 0, 8).cuda(), torch.zeros(1, 1, 0, 8).cuda(),
                                   torch.zeros(1, 2, 1, 3, 2, dtype=torch.int32).cuda())
    assert z.shape == (1, 2, 1, 2, 8) and not z.any()
    z = nv.selection_attention_hip(torch.zeros(0, 2, 1, 2, 8).cuda(), torch.zeros(0, 1, 4, 8).cuda(), torch.zeros(0, 1, 4, 8).cuda(),
                                   torch.zeros(0, 2, 1, 3, 2, dtype=torch.int32).cuda())
    assert z.numel() == 0
    with pytest.raises(RuntimeError):  # more ranges per row than the kernels support -> status code -> RuntimeError
        nv.selection_attention_hip(torch.zeros(1, 1, 1, 1, 8).cuda(), torch.zeros(1, 1, 4, 8).cuda(), torch.zeros(1, 1, 4, 8).cuda(),
                                   torch.zeros(1, 1, 1, 65, 2, dtype=torch.int32).cuda())
    with pytest.raises(RuntimeError):  # CPU tensors: no fallback
        nv.selection_attention_hip(torch.zeros(1, 1, 1, 1, 8), torch.zeros(1, 1, 4, 8), torch.zeros(1, 1, 4, 8),
                                   torch.zeros(1, 1, 1, 1, 2, dtype=torch.int32))


@pytest.mark.parametrize("S", [4096, 16384, 65536])
def test_g10_m7c_rows(nv, orc, S):
    """m7c shape at the BASELINE sequence lengths, sampled rows, against the reference's outputs."""
    g = load_golden(f"g10_m7c_S{S}")
    ts = gi.g10_rows(S)
    Qr, _ = gi.g10_q_kcmp(S, ts)
    K, V = gi.g10_kv(S)
    rg = g["r_bat"][None]
    # fp32, generic kernel, directly against the reference's fp32 output
    O32 = nv.selection_attention_hip(dev(Qr), dev(K), dev(V), dev(rg), variant=1).cpu().numpy()
    assert np.abs(O32[0] - g["O_bat"]).max() <= 1e-3
    # bf16, MFMA kernel: against the reference's fp32 output (input rounding included in the 1e-2 bar)
    Ob = nv.selection_attention_hip(dev(Qr, torch.bfloat16), dev(K, torch.bfloat16), dev(V, torch.bfloat16), dev(rg), variant=2)
    refb = orc.sel_attention_masked(rounded(Qr, torch.bfloat16), rounded(K, torch.bfloat16), rounded(V, torch.bfloat16), rg)
    assert np.abs(Ob.float().cpu().numpy() - refb).max() <= 1e-2
    assert np.abs(Ob.float().cpu().numpy()[0] - g["O_bat"]).max() <= 3e-2
    # sequential-mode ranges (normalised: inverted garbage entries removed)
    rs = g["r_seq"].copy()
    rs[rs[..., 1] <= rs[..., 0]] = 0
    Os = nv.selection_attention_hip(dev(Qr), dev(K), dev(V), dev(rs[None]), variant=1).cpu().numpy()
    assert np.abs(Os[0] - g["O_seq"]).max() <= 1e-3


@pytest.mark.parametrize("dtype,variant", [(torch.float32, 1), (torch.bfloat16, 1)])
def test_backward_vs_oracle(nv, orc, dtype, variant):
    rng = np.random.default_rng(21)
    B, S, G, h, D, S_kv, n = 2, 7, 2, 3, 32, 90, 4
    Q = rng.standard_normal((B, S, G, h, D), dtype=np.float32)
    K = rng.standard_normal((B, G, S_kv, D), dtype=np.float32)
    V = rng.standard_normal((B, G, S_kv, D), dtype=np.float32)
    dO = rng.standard_normal((B, S, G, h, D), dtype=np.float32)
    rg = _rand_ranges(rng, B, S, G, n, S_kv)
    rg[0, 0, 0] = 0
    q, k, v = (dev(x, dtype).requires_grad_(True) for x in (Q, K, V))
    O = nv.selection_attention_hip(q, k, v, dev(rg), variant=variant)
    O.backward(dev(dO, dtype))
    rq, rk, rv = orc.sel_attention_masked_bwd(rounded(Q, dtype), rounded(K, dtype), rounded(V, dtype), rg, rounded(dO, dtype))
    tol = 2e-3 if dtype == torch.float32 else 6e-2
    for got, ref, name in ((q.grad, rq, "dQ"), (k.grad, rk, "dK"), (v.grad, rv, "dV")):
        err = np.abs(got.float().cpu().numpy() - ref).max()
        assert err <= tol * max(1.0, np.abs(ref).max()), f"{name} err {err:.3e}"


def test_backward_m7c_shape_bf16(nv, orc):
    rng = np.random.default_rng(22)
    B, S, G, h, D, S_kv, n = 1, 6, 2, 6, 64, 512, 5
    Q = rng.standard_normal((B, S, G, h, D), dtype=np.float32)
    K = rng.standard_normal((B, G, S_kv, D), dtype=np.float32)
    V = rng.standard_normal((B, G, S_kv, D), dtype=np.float32)
    dO = rng.standard_normal((B, S, G, h, D), dtype=np.float32)
    rg = _rand_ranges(rng, B, S, G, n, S_kv, aligned=True)
    dt = torch.bfloat16
    q, k, v = (dev(x, dt).requires_grad_(True) for x in (Q, K, V))
    nv.selection_attention_hip(q, k, v, dev(rg)).backward(dev(dO, dt))  # forward = MFMA kernel (auto)
    rq, rk, rv = orc.sel_attention_masked_bwd(rounded(Q, dt), rounded(K, dt), rounded(V, dt), rg, rounded(dO, dt))
    for got, ref in ((q.grad, rq), (k.grad, rk), (v.grad, rv)):
        assert np.abs(got.float().cpu().numpy() - ref).max() <= 6e-2 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("B,S,G,h,S_kv,n,aligned", [(1, 6, 2, 6, 512, 5, True), (2, 300, 2, 6, 300, 6, False), (1, 700, 1, 16, 700, 4, True),
                                                   (2, 40, 2, 1, 130, 3, False), (1, 520, 2, 4, 520, 16, True)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_backward_mfma_vs_oracle(nv, orc, B, S, G, h, S_kv, n, aligned, dtype):
    """MFMA backward (dQ query-major, dK/dV key-block-major, no atomics) vs the oracle's fp64 backward; also against the
    generic kernel, and bitwise run-to-run reproducibility."""
    rng = np.random.default_rng([B, S, h, n])
    D = 64
    Q = rng.standard_normal((B, S, G, h, D), dtype=np.float32)
    K = rng.standard_normal((B, G, S_kv, D), dtype=np.float32)
    V = rng.standard_normal((B, G, S_kv, D), dtype=np.float32)
    dO = rng.standard_normal((B, S, G, h, D), dtype=np.float32)
    rg = _rand_ranges(rng, B, S, G, n, S_kv, aligned=aligned)
    rg[0, 0, 0] = 0  # empty row
    rg[0, 1, 0, 0] = (0, S_kv)  # everything, overlapping the other ranges of the row
    grads = []
    for rep in range(2):
        q, k, v = (dev(x, dtype).requires_grad_(True) for x in (Q, K, V))
        nv.selection_attention_hip(q, k, v, dev(rg), variant=2).backward(dev(dO, dtype))
        grads.append((q.grad, k.grad, v.grad))
    for a, b_ in zip(grads[0], grads[1]):
        assert torch.equal(a, b_)  # no atomics: bitwise reproducible
    rq, rk, rv = orc.sel_attention_masked_bwd(rounded(Q, dtype), rounded(K, dtype), rounded(V, dtype), rg, rounded(dO, dtype))
    for got, ref, name in zip(grads[0], (rq, rk, rv), ("dQ", "dK", "dV")):
        err = np.abs(got.float().cpu().numpy() - ref).max()
        assert err <= 3e-2 * max(1.0, np.abs(ref).max()), f"{name} err {err:.3e} (ref max {np.abs(ref).max():.2f})"
    q, k, v = (dev(x, dtype).requires_grad_(True) for x in (Q, K, V))
    nv.selection_attention_hip(q, k, v, dev(rg), variant=1).backward(dev(dO, dtype))
    for got, gen in zip(grads[0], (q.grad, k.grad, v.grad)):
        assert (got.float() - gen.float()).abs().max().item() <= 3e-2 * max(1.0, gen.float().abs().max().item())


def test_properties_full_size_64k(nv):
    """Size-independent properties at the BASELINE full size (S=65536, m7c, bf16, every row)."""
    torch.manual_seed(0)
    B, S, G, h, D, n = 1, 65536, 2, 6, 64, 16
    Q = torch.randn(B, S, G, h, D, device="cuda", dtype=torch.bfloat16)
    K = torch.randn(B, G, S, D, device="cuda", dtype=torch.bfloat16)
    V = torch.randn(B, G, S, D, device="cuda", dtype=torch.bfloat16)
    p = torch.rand(B, S, G, S // 64, device="cuda")
    meta = nv.build_block_meta(S, 32, 16, 64, n, 512)
    rg = nv.select_topn_ranges_batched(p, meta, n, S)
    O = nv.selection_attention_hip(Q, K, V, rg).float()
    assert torch.isfinite(O).all()
    # convex combination of V rows: |O| bounded by max |V| (+ bf16 rounding)
    assert O.abs().max().item() <= V.float().abs().max().item() * 1.01
    # rows with no complete block (t < 63) are empty in batched mode -> zeros
    assert not O[:, :63].any()
    # range order does not matter (union semantics): reverse the range list
    O_rev = nv.selection_attention_hip(Q, K, V, rg.flip(3)).float()
    assert (O - O_rev).abs().max().item() <= 1e-2
    # linearity in V
    V2 = torch.randn_like(V)
    O2 = nv.selection_attention_hip(Q, K, V2, rg).float()
    O12 = nv.selection_attention_hip(Q, K, (V.float() + V2.float()).bfloat16(), rg).float()
    assert (O12 - (O + O2)).abs().max().item() <= 6e-2
    # duplicated ranges change nothing
    O_dup = nv.selection_attention_hip(Q, K, V, torch.cat([rg, rg], dim=3)).float()
    assert (O - O_dup).abs().max().item() <= 1e-2


@pytest.mark.parametrize("mode", ["batched", "sequential"])
@pytest.mark.parametrize("B,S,G,dtype", [(2, 4096, 2, torch.bfloat16), (1, 16384, 2, torch.bfloat16), (1, 700, 3, torch.float16),
                                         (1, 300, 2, torch.float32), (1, 40, 1, torch.bfloat16)])
@pytest.mark.parametrize("fuse", [0, 1])
def test_select_and_attend_equals_separate_calls(nv, mode, B, S, G, dtype, fuse, tune):
    """nsa_sel_select_attn_fwd, as two launches (the default) and with the top-n selection inside the attention launch (SEL_FUSE = 1,
    MFMA route): ranges bit-exact and O identical to select_topn_ranges_{batched,rows} followed by selection_attention_hip"""
    tune("SEL_FUSE", fuse)
    g = torch.Generator(device="cuda")
    g.manual_seed(S + G)
    meta = nv.build_block_meta(S, 32, 16, 64, 16, 512)
    mk = lambda *sh: torch.randn(*sh, device="cuda", generator=g).to(dtype)  # noqa: E731
    Q, K, V = mk(B, S, G, 6, 64), mk(B, G, S, 64), mk(B, G, S, 64)
    p_grp = torch.rand(B, S, G, meta.S_sel, device="cuda", generator=g)
    p_grp[:, :, :, ::7] = 0.25  # ties
    if mode == "batched":
        r_ref = nv.select_topn_ranges_batched(p_grp, meta, 16, S)
    else:
        r_ref = nv.select_topn_ranges_rows(p_grp, meta, 16, 0)
    O_ref, lse_ref = nv.selection_attention_hip(Q, K, V, r_ref, return_lse=True)
    r, O, lse = nv.select_and_attend(p_grp, Q, K, V, meta, 16, mode=mode, return_lse=True)
    torch.cuda.synchronize()
    assert r.shape == r_ref.shape and torch.equal(r, r_ref)
    assert torch.equal(O, O_ref) and torch.equal(lse, lse_ref)


@pytest.mark.parametrize("name", ["a", "b", "c"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_first_key_parity_mode_matches_reference_default_executors(nv, orc, name, dtype):
    """opt-in parity mode == grouped_selection_attention_packed / grouped_selection_attention of the reference (g15, bit-exact: the
    output is a copy of V rows), also through a strided cache view and with int64 ranges"""
    g = load_golden("g15_first_key_" + name)
    Q, K, V = (torch.from_numpy(g[k]).cuda().to(dtype) for k in ("Q", "K", "V"))
    rg = torch.from_numpy(g["ranges"]).cuda()
    want = torch.from_numpy(g["O"]).cuda().to(dtype)
    assert torch.equal(nv.selection_attention_first_key_parity(Q, K, V, rg), want)
    cache = torch.zeros(V.shape[0], V.shape[1], V.shape[2] + 13, V.shape[3], device="cuda", dtype=dtype)
    cache[:, :, : V.shape[2]] = V
    r64 = rg.to(torch.int64)
    keep = r64.clone()
    assert torch.equal(nv.selection_attention_first_key_parity(Q, K, cache[:, :, : V.shape[2]], r64), want)
    assert torch.equal(r64, keep)
    # it is NOT the selected branch: the semantic executor differs wherever a row gathers more than one key
    sem = nv.selection_attention_hip(Q, K, V, rg)
    assert (sem.float() - want.float()).abs().max().item() > 0.1
    # out-of-range ranges are clamped like everywhere else
    S_kv = V.shape[2]
    wild = torch.tensor([[-5, -1], [S_kv - 2, S_kv + 100]], dtype=torch.int32, device="cuda").expand(*rg.shape[:3], 2, 2).contiguous()
    out = nv.selection_attention_first_key_parity(Q, K, V, wild)
    assert torch.equal(out, V[:, :, S_kv - 2][:, None, :, None, :].expand_as(out))


# ---- the query-tile forward kernel (sel_attn_rows_mfma.hip): several rows per wave sharing K/V tiles -------------------------
@pytest.mark.parametrize("mode", ["1", "3"])  # 1 = 16/h rows per wave ("pairs" at h = 6), 3 = 48/h rows
@pytest.mark.parametrize("h,D", [(3, 64), (4, 64), (5, 64), (6, 64), (8, 64), (16, 64), (2, 128), (4, 128), (8, 128)])
def test_query_tile_kernel_against_oracle(nv, orc, mode, h, D, tune):
    """rows of one wave with DIFFERENT selections: unaligned, overlapping and duplicate ranges (partially covered tiles), an empty
    row, a row covering everything, a key range that ends in the last (partial) tile of K/V, odd S (last wave short)"""
    tune("SEL_ROWS", mode)
    rng = np.random.default_rng([h, D, int(mode)])
    B, S, G, n, S_kv = 2, 37, 2, 9, 333  # S_kv not a multiple of 32
    Q = rng.standard_normal((B, S, G, h, D), dtype=np.float32)
    K = rng.standard_normal((B, G, S_kv, D), dtype=np.float32)
    V = rng.standard_normal((B, G, S_kv, D), dtype=np.float32)
    rg = _rand_ranges(rng, B, S, G, n, S_kv)
    rg[0, 0, 0] = 0  # empty row
    rg[0, 1, 0] = 0
    rg[0, 1, 0, 0] = (0, S_kv)  # everything
    rg[0, 2, 0, :3] = [(10, 50), (40, 70), (40, 70)]  # overlap + duplicate
    rg[0, 3, 0, :2] = [(320, 400), (-7, 3)]  # clamped on both ends, ends in the partial last tile
    rg[1, 5, 1] = 0
    rg[1, 5, 1, 0] = (95, 97)  # two keys straddling nothing: one partially covered tile only
    for dtype in (torch.bfloat16, torch.float16):
        run_case(nv, orc, Q, K, V, rg, dtype, variant=2)


def test_query_tile_kernel_long_context_second_bitmap_word_group(nv, tune):
    """S_kv > 65536: the tile schedule spans more than 64 bitmap words (second register of the schedule); against the generic kernel"""
    rng = np.random.default_rng(123)
    B, S, G, h, D, n, S_kv = 1, 24, 2, 6, 64, 16, 100000
    Q = torch.from_numpy(rng.standard_normal((B, S, G, h, D), dtype=np.float32)).cuda().bfloat16()
    K = torch.from_numpy(rng.standard_normal((B, G, S_kv, D), dtype=np.float32)).cuda().bfloat16()
    V = torch.from_numpy(rng.standard_normal((B, G, S_kv, D), dtype=np.float32)).cuda().bfloat16()
    st = rng.integers(0, S_kv // 64, size=(B, S, G, n)) * 64
    rg = np.stack([st, np.minimum(st + 64, S_kv)], axis=-1).astype(np.int32)
    rg[0, :, :, 0] = (0, 64)
    rg[0, :, :, 1] = (S_kv - 100, S_kv)  # the very end, unaligned
    rg[0, 3, 1, 2] = (65500, 65600)  # straddles the word-group boundary (tile 2047 | 2048)
    want = nv.selection_attention_hip(Q, K, V, dev(rg), variant=1).float()
    for mode in ("1", "3"):
        tune("SEL_ROWS", mode)
        got = nv.selection_attention_hip(Q, K, V, dev(rg), variant=2, return_lse=True)
        assert (got[0].float() - want).abs().max().item() <= 1e-2
        assert torch.isfinite(got[1]).all()


@pytest.mark.parametrize("mode", ["sequential", "batched"])
def test_query_tile_kernel_equals_one_row_kernel_on_selector_output(nv, mode, tune):
    """m7c geometry, ranges from the real selector (fused in the launch): all three forward kernels agree, and the fused launch
    writes the same ranges whichever kernel hosts the selector"""
    tune("SEL_FUSE", 1)
    torch.manual_seed(3)
    B, S, G, h, D = 2, 1500, 2, 6, 64
    meta = nv.build_block_meta(S, 32, 16, 64, 16, 512)
    Q = torch.randn(B, S, G, h, D, device="cuda").bfloat16()
    K = torch.randn(B, G, S, D, device="cuda").bfloat16()
    V = torch.randn(B, G, S, D, device="cuda").bfloat16()
    p = torch.rand(B, S, G, meta.S_sel, device="cuda")
    outs = {}
    for m in ("0", "1", "3"):
        tune("SEL_ROWS", m)
        outs[m] = nv.select_and_attend(p, Q, K, V, meta, 16, mode=mode, scale=0.125)
    for m in ("1", "3"):
        assert torch.equal(outs[m][0], outs["0"][0])
        assert (outs[m][1].float() - outs["0"][1].float()).abs().max().item() <= 2e-2


def test_backward_query_tile_dq_long_context_and_kernel_switch(nv, tune):
    """dQ of the query-tile kernel (rows of a wave share tile images) == dQ of the one-row kernel on the same inputs, including a
    context of more than 65536 keys (second schedule register) and rows with different, unaligned, overlapping ranges"""
    rng = np.random.default_rng(2024)
    B, S, G, h, D, n, S_kv = 1, 21, 2, 6, 64, 8, 70000
    Q = dev(rng.standard_normal((B, S, G, h, D), dtype=np.float32), torch.bfloat16)
    K = dev(rng.standard_normal((B, G, S_kv, D), dtype=np.float32), torch.bfloat16)
    V = dev(rng.standard_normal((B, G, S_kv, D), dtype=np.float32), torch.bfloat16)
    dO = dev(rng.standard_normal((B, S, G, h, D), dtype=np.float32), torch.bfloat16)
    st = rng.integers(0, S_kv - 200, size=(B, S, G, n))
    rg = np.stack([st, st + rng.integers(0, 150, size=st.shape)], axis=-1).astype(np.int32)
    rg[0, :, :, 0] = (0, 64)
    rg[0, :, :, 1] = (65500, 65610)  # straddles tile 2047 | 2048
    rg[0, 2, 0] = 0  # empty row
    rg[0, 5, 1, 2] = (69990, 70000)  # the partial last tile
    res = {}
    for mode in ("0", "1"):
        tune("SEL_ROWS", mode)
        q, k, v = (x.clone().requires_grad_(True) for x in (Q, K, V))
        nv.selection_attention_hip(q, k, v, dev(rg), variant=2).backward(dO)
        res[mode] = (q.grad, k.grad, v.grad)
    for a, b_ in zip(res["0"], res["1"]):
        assert torch.isfinite(a).all() and (a.float() - b_.float()).abs().max().item() <= 2e-2 * max(1.0, a.float().abs().max().item())
    assert not res["1"][0][0, 2, 0].any()  # empty row: zero gradient


@pytest.mark.parametrize("variant", [1, 2])
def test_backward_all_ranges_empty_gives_zero_gradients(nv, variant):
    """the reference's edge case (test_selection_backward_edges.py:28-47): forward is zero, every gradient is zero -- and finite"""
    torch.manual_seed(0)
    B, S, G, h, D, S_kv = 2, 9, 2, 6, 64, 130
    q = torch.randn(B, S, G, h, D, device="cuda").bfloat16().requires_grad_(True)
    k = torch.randn(B, G, S_kv, D, device="cuda").bfloat16().requires_grad_(True)
    v = torch.randn(B, G, S_kv, D, device="cuda").bfloat16().requires_grad_(True)
    rg = torch.tensor([[0, 0], [4, 4], [9, 3]], dtype=torch.int64, device="cuda").expand(B, S, G, 3, 2).contiguous()
    o = nv.selection_attention_hip(q, k, v, rg, variant=variant)
    assert not o.any()
    o.backward(torch.randn_like(o))
    for t in (q, k, v):
        assert t.grad is not None and torch.isfinite(t.grad).all() and not t.grad.any()


@pytest.mark.parametrize("name", ["ref_test_shape", "a", "b", "c"])
@pytest.mark.parametrize("dtype,variant", [(torch.float32, 1), (torch.bfloat16, 1), (torch.bfloat16, 2), (torch.float16, 2)])
def test_g16_backward_matches_reference_autograd(nv, name, dtype, variant):
    """HIP backward (generic kernel, and the MFMA kernels where the shape is theirs: bf16/f16, Dk = Dv = 64) against dQ/dK/dV from torch
    autograd through the REFERENCE's grouped_selection_attention_masked (goldens g16, oracle/make_round2_goldens.py; reference
    test nsa/tests/test_selection_backward_reference.py:35-37).  fp32: 2e-4 absolute / relative; half precision: gradients of the
    rounded inputs differ from the fp32 golden by the input rounding, bound 6e-2 of the gradient's scale."""
    g = load_golden("g16_bwd_" + name)
    Q, K, V, rg, dO = g["Q"], g["K"], g["V"], g["ranges"], g["dO"]
    q, k, v = (dev(x, dtype).requires_grad_(True) for x in (Q, K, V))
    if variant == 2 and not (Q.shape[-1] == 64 and V.shape[-1] == 64):
        # the reference's own test shapes have head sizes 8 / 16 / 32 (nsa/tests/test_selection_backward_reference.py): not MFMA shapes at any
        # D -- the generic kernels cover them (variant 1 above).  An explicit request for the MFMA variant must be refused loudly, never
        # served by another kernel behind the caller's back
        with pytest.raises(RuntimeError, match="MFMA variant requested"):
            nv.selection_attention_hip(q, k, v, dev(rg), variant=2)
        return
    O = nv.selection_attention_hip(q, k, v, dev(rg), variant=variant)
    O.backward(dev(dO, dtype))
    assert np.abs(O.detach().float().cpu().numpy() - g["O"]).max() <= TOL[dtype] * (4 if dtype != torch.float32 else 1)
    for got, want, what in ((q.grad, g["dQ"], "dQ"), (k.grad, g["dK"], "dK"), (v.grad, g["dV"], "dV")):
        got = got.float().cpu().numpy()
        if dtype == torch.float32:
            assert np.allclose(got, want, atol=2e-4, rtol=2e-4), what
        else:
            assert np.abs(got - want).max() <= 6e-2 * max(1.0, np.abs(want).max()), what
    if name != "ref_test_shape":
        assert not q.grad[0, 0, 0].any()  # row without a token


@pytest.mark.parametrize("name", ["a", "b", "c"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_g17_head_causal_parity_mode_matches_reference(nv, orc, name, dtype):
    """parity mode of NSAAttention._sdpa_over_ranges (nsa_attention.py:1779-1855; goldens g17 from the imported reference): head i attends
    the first i+1 gathered tokens.  Also on a strided cache view and with int64 ranges (never modified)."""
    g = load_golden("g17_head_causal_" + name)
    Q, K, V, rg = dev(g["Q"], dtype)[:, None], dev(g["K"], dtype), dev(g["V"], dtype), dev(g["ranges"])[:, None]
    O = nv.selection_attention_head_causal_parity(Q, K, V, rg)[:, 0]
    tol = 1e-5 if dtype == torch.float32 else 3e-2
    assert np.abs(O.float().cpu().numpy() - g["O"]).max() <= tol
    if dtype != torch.float32:  # half precision: against the oracle on the rounded inputs
        ref = orc.sel_attention_head_causal_parity(rounded(g["Q"], dtype)[:, None], rounded(g["K"], dtype), rounded(g["V"], dtype),
                                                   g["ranges"][:, None])[:, 0]
        assert np.abs(O.float().cpu().numpy() - ref).max() <= 1e-2
    cache = torch.zeros(V.shape[0], V.shape[1], V.shape[2] + 13, V.shape[3], device="cuda", dtype=dtype)
    cache[:, :, : V.shape[2]] = V
    r64 = rg.to(torch.int64)
    keep = r64.clone()
    O2 = nv.selection_attention_head_causal_parity(Q, K, cache[:, :, : V.shape[2]], r64)[:, 0]
    assert torch.equal(O, O2) and torch.equal(r64, keep)
    assert not O[0, 0].any()


# ---- the block-form forward kernel (sel_attn_blocks_mfma.hip): NT column tiles of 16/h rows per wave, 64-key blocks ------------
@pytest.mark.parametrize("nt", [1, 2, 4])
@pytest.mark.parametrize("h", [1, 2, 3, 4, 5, 6, 8, 16])
def test_block_kernel_against_oracle(nv, orc, nt, h, tune):
    """rows of one wave with DIFFERENT selections: unaligned, overlapping and duplicate ranges (partially covered blocks incl. ones that
    straddle the 32-key word halves of the per-row key mask), an empty row, a row covering everything, ranges ending in the last
    (partial) block of K/V, odd S (last wave / last column tile short), S_kv not a multiple of 64"""
    tune("SEL_ROWS", -1)
    tune("SEL_BLOCKS", nt)
    D = 64
    rng = np.random.default_rng([h, nt, 77])
    B, S, G, n, S_kv = 2, 37, 2, 9, 333
    Q = rng.standard_normal((B, S, G, h, D), dtype=np.float32)
    K = rng.standard_normal((B, G, S_kv, D), dtype=np.float32)
    V = rng.standard_normal((B, G, S_kv, D), dtype=np.float32)
    rg = _rand_ranges(rng, B, S, G, n, S_kv)
    rg[0, 0, 0] = 0  # empty row
    rg[0, 1, 0] = 0
    rg[0, 1, 0, 0] = (0, S_kv)  # everything
    rg[0, 2, 0, :3] = [(10, 50), (40, 70), (40, 70)]  # overlap + duplicate, crosses the block boundary at 64
    rg[0, 3, 0, :2] = [(320, 400), (-7, 3)]  # clamped on both ends, ends in the partial last block
    rg[1, 5, 1] = 0
    rg[1, 5, 1, 0] = (95, 97)  # two keys: one partially covered block, upper word of its key mask
    rg[1, 6, 1] = 0
    rg[1, 6, 1, :2] = [(128, 192), (192, 230)]  # a full block followed by a partial one (the sequential selector's clamp at t+1)
    rg[1, 7, 0] = 0
    rg[1, 7, 0, :2] = [(30, 34), (60, 70)]  # straddles key 32 inside a block and the block boundary
    for dtype in (torch.bfloat16, torch.float16):
        run_case(nv, orc, Q, K, V, rg, dtype, variant=2)


def test_block_kernel_long_context_and_kernel_agreement(nv, tune):
    """S_kv = 131072 (the kernel's limit: 2048 blocks = 64 schedule words), aligned selector-like blocks + an unaligned tail range: the
    block form, the query-tile form and the generic kernel agree; the lse is finite; beyond the limit the query-tile form takes over"""
    rng = np.random.default_rng(321)
    B, S, G, h, D, n, S_kv = 1, 24, 2, 6, 64, 16, 131072
    Q = torch.from_numpy(rng.standard_normal((B, S, G, h, D), dtype=np.float32)).cuda().bfloat16()
    K = torch.from_numpy(rng.standard_normal((B, G, S_kv, D), dtype=np.float32)).cuda().bfloat16()
    V = torch.from_numpy(rng.standard_normal((B, G, S_kv, D), dtype=np.float32)).cuda().bfloat16()
    st = rng.integers(0, S_kv // 64, size=(B, S, G, n)) * 64
    rg = np.stack([st, np.minimum(st + 64, S_kv)], axis=-1).astype(np.int32)
    rg[0, :, :, 0] = (0, 64)
    rg[0, :, :, 1] = (S_kv - 100, S_kv)  # the very end, unaligned start
    rg[0, 3, 1, 2] = (65500, 65600)
    want = nv.selection_attention_hip(Q, K, V, dev(rg), variant=1).float()
    tune("SEL_ROWS", -1)
    for nt in (1, 4):
        tune("SEL_BLOCKS", nt)
        got = nv.selection_attention_hip(Q, K, V, dev(rg), variant=2, return_lse=True)
        assert (got[0].float() - want).abs().max().item() <= 1e-2
        assert torch.isfinite(got[1]).all()


@pytest.mark.parametrize("mode", ["sequential", "batched"])
def test_block_kernel_equals_other_kernels_on_selector_output(nv, mode, tune):
    """m7c geometry, ranges from the real selector (fused in the launch): the block form (every NT) against the one-row kernel -- same
    ranges bit for bit whichever kernel hosts the selector, outputs within rounding; S not a multiple of the rows per wave"""
    tune("SEL_FUSE", 1)
    torch.manual_seed(4)
    B, S, G, h, D = 2, 1501, 2, 6, 64
    meta = nv.build_block_meta(S, 32, 16, 64, 16, 512)
    Q = torch.randn(B, S, G, h, D, device="cuda").bfloat16()
    K = torch.randn(B, G, S, D, device="cuda").bfloat16()
    V = torch.randn(B, G, S, D, device="cuda").bfloat16()
    p = torch.rand(B, S, G, meta.S_sel, device="cuda")
    tune("SEL_ROWS", 0)
    ref = nv.select_and_attend(p, Q, K, V, meta, 16, mode=mode, scale=0.125, return_lse=True)
    tune("SEL_ROWS", -1)
    for nt in (1, 2, 4):
        tune("SEL_BLOCKS", nt)
        got = nv.select_and_attend(p, Q, K, V, meta, 16, mode=mode, scale=0.125, return_lse=True)
        assert torch.equal(got[0], ref[0])
        assert (got[1].float() - ref[1].float()).abs().max().item() <= 2e-2
        fin = torch.isfinite(ref[2])
        assert torch.equal(torch.isfinite(got[2]), fin) and (got[2][fin] - ref[2][fin]).abs().max().item() <= 2e-2
        sep = nv.selection_attention_hip(Q, K, V, got[0], scale=0.125)  # ranges tensor instead of the fused selector: same kernel, same result
        assert torch.equal(sep, got[1])


def test_block_kernel_forced_max_raise_at_chosen_blocks(nv, orc, tune):
    """the deferred-max branch is data dependent: spike one key per chosen block against one query head so that the running max has
    to be raised (and O / l rescaled) at the 1st, a middle and the last block of a row, in a column tile whose other row does NOT
    own that block (off slots must stay untouched by the rescale of their neighbours)"""
    tune("SEL_ROWS", -1)
    tune("SEL_BLOCKS", 4)
    rng = np.random.default_rng(99)
    B, S, G, h, D, S_kv = 1, 16, 1, 6, 64, 1024
    Q = rng.standard_normal((B, S, G, h, D), dtype=np.float32)
    K = rng.standard_normal((B, G, S_kv, D), dtype=np.float32) * 0.3
    V = rng.standard_normal((B, G, S_kv, D), dtype=np.float32)
    rg = np.zeros((B, S, G, 4, 2), np.int32)
    for t in range(S):
        blocks = [0, 3 + (t % 5), 9, 15] if t % 2 == 0 else [1, 4, 9, 12]
        rg[0, t, 0] = [(64 * j, 64 * j + 64) for j in blocks]
    for t, blk_, head, gain in ((0, 0, 0, 6.0), (0, 9, 2, 14.0), (0, 15, 2, 30.0), (3, 12, 5, 25.0), (6, 4, 1, 18.0)):
        key = 64 * blk_ + 17
        K[0, 0, key] = Q[0, t, 0, head] * gain / np.sqrt(D) * 8 / np.linalg.norm(Q[0, t, 0, head])
    for dtype in (torch.bfloat16, torch.float16):
        run_case(nv, orc, Q, K, V, rg, dtype, variant=2, tol=2e-2)


# ---- the decode form (sel_attn_decode.hpp): one 1024-thread workgroup per row, partials merged through LDS --------------------
@pytest.mark.parametrize("h", [1, 3, 6, 8, 16])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_decode_workgroup_kernel_against_oracle(nv, orc, h, dtype, tune):
    """S = 1 rows with arbitrary ranges: unaligned, overlapping, duplicate and inverted ranges, an empty row, one key, a range that
    covers everything (more 64-key chunks than the workgroup has waves: waves loop and merge their own chunks first), the tail
    of K/V, a strided cache view; and the split-KV route (DECODE_WG = 0) gives the same result within rounding"""
    rng = np.random.default_rng([h, 5])
    B, G, D, n, S_kv = 7, 2, 64, 16, 3000
    Q = rng.standard_normal((B, 1, G, h, D), dtype=np.float32)
    K = rng.standard_normal((B, G, S_kv, D), dtype=np.float32)
    V = rng.standard_normal((B, G, S_kv, D), dtype=np.float32)
    rg = _rand_ranges(rng, B, 1, G, n, S_kv)
    rg[0, 0, 0] = 0  # empty row
    rg[1, 0, 0] = 0
    rg[1, 0, 0, 0] = (0, S_kv)  # everything: 47 chunks
    rg[1, 0, 1, :3] = [(10, 50), (40, 70), (40, 70)]  # overlap + duplicate
    rg[2, 0, 0] = 0
    rg[2, 0, 0, 0] = (2999, 3000)  # one key, the last one
    rg[2, 0, 1, :3] = [(2900, 3100), (-5, 3), (700, 600)]  # clamped on both ends, inverted
    rg[3, 0, 0] = [(64 * j, 64 * j + 64) for j in (0, 3, 4, 9, 12, 13, 14, 20, 21, 25, 30, 33, 38, 40, 44, 45)]  # selector-like
    rg[3, 0, 0, -1] = (64 * 45, 64 * 45 + 17)  # clamp at t+1
    qd, kd, vd, rd = dev(Q, dtype), dev(K, dtype), dev(V, dtype), dev(rg)
    tune("DECODE_WG", -1)
    O = nv.selection_attention_hip(qd, kd, vd, rd)
    ref = orc.sel_attention_masked(rounded(Q, dtype), rounded(K, dtype), rounded(V, dtype), rg)
    assert np.abs(O.float().cpu().numpy() - ref).max() <= TOL[dtype]
    assert not O[0, 0, 0].any()
    cache = torch.zeros(B, G, S_kv + 77, D, device="cuda", dtype=dtype)
    cache[:, :, :S_kv] = vd
    kcache = torch.zeros(B, G, S_kv + 77, D, device="cuda", dtype=dtype)
    kcache[:, :, :S_kv] = kd
    assert torch.equal(nv.selection_attention_hip(qd, kcache[:, :, :S_kv], cache[:, :, :S_kv], rd), O)  # strided views, bitwise
    assert torch.equal(nv.selection_attention_hip(qd, kd, vd, rd), O)  # fixed merge order: run-to-run reproducible
    tune("DECODE_WG", 0)
    O_split = nv.selection_attention_hip(qd, kd, vd, rd)
    assert (O_split.float() - O.float()).abs().max().item() <= TOL[dtype]


def test_backward_properties_at_config5_size(nv):
    """BASELINE config 5 shape of the selected branch (m7c, S = 4096, B = 8, bf16, ranges from the batched selector): the backward at
    full size is finite, bitwise reproducible (no atomics), exactly linear in dO (scaling dO by 2 is exact in every product), leaves
    zero dK / dV on keys no row selected (the cache is longer than the selected prefix) and zero dQ on rows without a token"""
    torch.manual_seed(11)
    B, S, G, h, D, n = 8, 4096, 2, 6, 64, 16
    S_kv = S + 512
    meta = nv.build_block_meta(S, 32, 16, 64, n, 512)
    Q = torch.randn(B, S, G, h, D, device="cuda").bfloat16()
    K = torch.randn(B, G, S_kv, D, device="cuda").bfloat16()
    V = torch.randn(B, G, S_kv, D, device="cuda").bfloat16()
    dO = torch.randn(B, S, G, h, D, device="cuda").bfloat16()
    rg = nv.select_topn_ranges_batched(torch.rand(B, S, G, meta.S_sel, device="cuda"), meta, n, S)
    assert not rg[:, :63].any()  # batched mode: rows before the first complete block select nothing

    def grads(d):
        q, k, v = (x.clone().requires_grad_(True) for x in (Q, K, V))
        nv.selection_attention_hip(q, k, v, rg).backward(d)
        return q.grad, k.grad, v.grad

    g1, g1b, g2 = grads(dO), grads(dO), grads(dO * 2)
    for a, b_, c in zip(g1, g1b, g2):
        assert torch.isfinite(a).all()
        assert torch.equal(a, b_)  # reproducible
        assert torch.equal(a.float() * 2, c.float())  # linear in dO (a power-of-two factor is exact)
    assert not g1[1][:, :, S:].any() and not g1[2][:, :, S:].any()  # keys beyond every range: never hit
    assert not g1[0][:, :63].any()  # rows without a token
    assert g1[1][:, :, :S].abs().amax(dim=(2, 3)).min().item() > 0 and g1[0][:, 64:].abs().amax(dim=(2, 3, 4)).min().item() > 0


@pytest.mark.parametrize("S", [300, 1501, 4096])
def test_flat_block_kernel_matches_four_tile_form(nv, orc, tune, S):
    """h = 6: the block-form attention with the 8 rows of a wave on three full column tiles against the four-tile form and the oracle: a column
    is one (row, head) pair either way, so the two forms differ by rounding only (different MFMA groupings of the same columns)"""
    torch.manual_seed(S)
    B, G, h, D = 2, 2, 6, 64
    meta = nv.build_block_meta(S, 32, 16, 64, 16, 512)
    Q = torch.randn(B, S, G, h, D, device="cuda").bfloat16()
    K = torch.randn(B, G, S, D, device="cuda").bfloat16()
    V = torch.randn(B, G, S, D, device="cuda").bfloat16()
    rg = nv.select_topn_ranges_batched(torch.rand(B, S, G, meta.S_sel, device="cuda"), meta, 16, S)
    out = {}
    for flat in (0, 1):
        tune("SEL_FLAT", flat)
        out[flat] = nv.selection_attention_hip(Q, K, V, rg, return_lse=True)
    torch.cuda.synchronize()
    assert (out[0][0].float() - out[1][0].float()).abs().max().item() <= 2e-2
    fin = torch.isfinite(out[0][1])
    assert torch.equal(fin, torch.isfinite(out[1][1])) and (out[0][1][fin] - out[1][1][fin]).abs().max().item() <= 1e-3
    want = orc.sel_attention_masked(Q.float().cpu().numpy(), K.float().cpu().numpy(), V.float().cpu().numpy(), rg.cpu().numpy())
    assert np.abs(out[1][0].float().cpu().numpy() - want).max() <= 1e-2


# zone thresholds of the key-split form (SEL_KSPLIT_T1 / _T2 as fractions of S): every row in four classes; every row in two; the mixed form
# (rows whole / in two / in four by position, what a 64k context gets); whole + four only; and the defaults (16k / 32k: a short context is
# walked whole through the split launcher)
KS_ZONES = [(0.0, 0.0), (0.0, 1e9), (0.25, 0.5), (0.4, 0.4), (None, None)]


def _ks_tune(tune, S, z):
    for name, f in (("SEL_KSPLIT_T1", z[0]), ("SEL_KSPLIT_T2", z[1])):
        tune(name, -1 if f is None else int(min(f * S, 2 ** 30)))


@pytest.mark.parametrize("B,S,G", [(2, 4096, 2), (1, 1501, 2), (3, 700, 1), (1, 300, 4), (9, 1024, 2)])
def test_key_split_block_kernel_matches_plain_walk(nv, orc, tune, B, S, G):
    """the block form with the keys of a row split over 2 / 4 workgroup sets by the row's position (8-block stripes dealt to the classes in
    turn, partial (m, l, O / l) records in f16, merged by a second launch in class order; rows below the first threshold are walked whole and
    written directly) against the plain walk and the oracle, output and log-sum-exp, for every zone layout; forced on by the tuning switch
    (by itself it only applies to long contexts).  Semantics: nsa/core/attention_kernels.py:705-772."""
    torch.manual_seed(S + B)
    h, D = 6, 64
    meta = nv.build_block_meta(S, 32, 16, 64, 16, 512)
    Q = torch.randn(B, S, G, h, D, device="cuda").bfloat16()
    K = torch.randn(B, G, S, D, device="cuda").bfloat16()
    V = torch.randn(B, G, S, D, device="cuda").bfloat16()
    rg = nv.select_topn_ranges_batched(torch.rand(B, S, G, meta.S_sel, device="cuda"), meta, 16, S)
    rg[:, 5] = 0  # a row without any key
    rg[:, S - 7] = 0  # (one in the last zone too)
    tune("SEL_FLAT", 0)
    tune("SEL_KSPLIT", 0)
    plain = nv.selection_attention_hip(Q, K, V, rg, return_lse=True)
    want = orc.sel_attention_masked(Q.float().cpu().numpy(), K.float().cpu().numpy(), V.float().cpu().numpy(), rg.cpu().numpy())
    tune("SEL_KSPLIT", 1)
    for z in KS_ZONES:
        _ks_tune(tune, S, z)
        out = nv.selection_attention_hip(Q, K, V, rg, return_lse=True)
        torch.cuda.synchronize()
        assert (plain[0].float() - out[0].float()).abs().max().item() <= 1.6e-2, z  # one bf16 ulp at |O| < 4
        fin = torch.isfinite(plain[1])
        assert torch.equal(fin, torch.isfinite(out[1])) and (plain[1][fin] - out[1][fin]).abs().max().item() <= 1e-3, z
        assert not out[0][:, 5].any() and not out[0][:, S - 7].any(), z
        assert np.abs(out[0].float().cpu().numpy() - want).max() <= 1e-2, z
        if z == (None, None):  # every row below the first threshold: the plain walk's arithmetic, only the workgroup order differs
            assert torch.equal(out[0], plain[0])
        # run to run: bit-identical (the classes are merged in a fixed order)
        again = nv.selection_attention_hip(Q, K, V, rg)
        assert torch.equal(again, out[0]), z


def test_key_split_rows_at_the_end_of_a_longer_context(nv, orc, tune):
    """the zones follow the rows' POSITIONS (row + S_kv - S): a chunk of 512 query rows at the end of a 3000-key cache, thresholds at 2600 /
    2800 positions, so the chunk holds rows of all three zones although its row indices start at 0"""
    torch.manual_seed(11)
    B, S, S_kv, G, h, D = 2, 512, 3000, 2, 6, 64
    Q = torch.randn(B, S, G, h, D, device="cuda").bfloat16()
    K = torch.randn(B, G, S_kv, D, device="cuda").bfloat16()
    V = torch.randn(B, G, S_kv, D, device="cuda").bfloat16()
    blk = torch.rand(B, S, G, (S_kv - S) // 64, device="cuda").topk(12, dim=-1).indices.sort(dim=-1).values.int()
    rg = torch.zeros(B, S, G, 16, 2, dtype=torch.int32, device="cuda")
    rg[..., :12, 0] = blk * 64
    rg[..., :12, 1] = blk * 64 + 64
    pos = (S_kv - S + torch.arange(S, device="cuda", dtype=torch.int32))[None, :, None]
    rg[..., 12, 0] = (pos // 64) * 64  # the row's own partial block
    rg[..., 12, 1] = pos + 1
    tune("SEL_ROWS", -1), tune("SEL_BLOCKS", -1)  # (the block form at its default width: the only one with a key-split form -- a suite run
    # under NSA_HIP_SEL_ROWS / NSA_HIP_SEL_BLOCKS would otherwise compare the plain walk with itself in the last assertion)
    tune("SEL_FLAT", 0), tune("SEL_KSPLIT", 0)
    plain = nv.selection_attention_hip(Q, K, V, rg)
    tune("SEL_KSPLIT", 1), tune("SEL_KSPLIT_T1", 2600), tune("SEL_KSPLIT_T2", 2800)
    out = nv.selection_attention_hip(Q, K, V, rg)
    torch.cuda.synchronize()
    assert (plain.float() - out.float()).abs().max().item() <= 1.6e-2
    want = orc.sel_attention_masked(Q.float().cpu().numpy(), K.float().cpu().numpy(), V.float().cpu().numpy(), rg.cpu().numpy())
    assert np.abs(out.float().cpu().numpy() - want).max() <= 1e-2
    # the zones really follow the positions: rows below position 2600 (row < 112; workgroups hold 32 rows: boundary row 128) are walked whole --
    # the plain walk's bits -- while the split rows went through f16 records and differ from it in some last bits
    assert torch.equal(out[:, :96], plain[:, :96])
    assert not torch.equal(out[:, 128:], plain[:, 128:])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("h", [6, 4, 8, 3])
def test_key_split_other_head_counts_and_f16(nv, orc, tune, dtype, h):
    """key-split form beyond h = 6 / bf16: with f16 inputs the f16 partial records round at the precision of the output itself (the merge
    rounds twice), and h != 6 changes the rows per column tile (4, 2 and 5 rows per tile) -- each against the oracle at the north-star
    tolerance and against the plain walk"""
    torch.manual_seed(100 + h)
    B, S, G, D = 2, 2500, 2, 64
    meta = nv.build_block_meta(S, 32, 16, 64, 16, 512)
    Q = torch.randn(B, S, G, h, D, device="cuda").to(dtype)
    K = torch.randn(B, G, S, D, device="cuda").to(dtype)
    V = torch.randn(B, G, S, D, device="cuda").to(dtype)
    rg = nv.select_topn_ranges_batched(torch.rand(B, S, G, meta.S_sel, device="cuda"), meta, 16, S)
    tune("SEL_FLAT", 0)
    want, want_lse = orc.sel_attention_masked(Q.float().cpu().numpy(), K.float().cpu().numpy(), V.float().cpu().numpy(), rg.cpu().numpy(),
                                              return_lse=True)
    fin = np.isfinite(want_lse)
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    tune("SEL_KSPLIT", 0)
    plain = nv.selection_attention_hip(Q, K, V, rg, return_lse=True)
    assert np.abs(plain[0].float().cpu().numpy() - want).max() <= TOL[dtype], (h, dtype)
    tune("SEL_KSPLIT", 1)
    for z in KS_ZONES[:4]:  # (rows per workgroup are 64 / 32 / 80 for h = 4 / 8 / 3: the zone boundaries round to them)
        _ks_tune(tune, S, z)
        out = nv.selection_attention_hip(Q, K, V, rg, return_lse=True)
        torch.cuda.synchronize()
        assert np.abs(out[0].float().cpu().numpy() - want).max() <= TOL[dtype], (z, h, dtype)
        assert np.array_equal(np.isfinite(out[1].cpu().numpy()), fin) and np.abs(out[1].cpu().numpy()[fin] - want_lse[fin]).max() <= 2e-2, (z, h)
        assert (plain[0].float() - out[0].float()).abs().max().item() <= 4 * ulp * 1.01, (z, h)  # a rounding step of the output at |O| < 4


def test_key_split_records_survive_values_beyond_the_f16_range(nv, orc, tune):
    """bf16 V with |V| up to 3e5 (beyond f16's 65504): the partial records carry a power-of-two scale, so the key-split form stays finite and
    agrees with the plain walk wherever that one is (a zero-weight inf record turned into NaN in the merge)"""
    torch.manual_seed(77)
    B, S, G, h, D = 1, 1200, 2, 6, 64
    meta = nv.build_block_meta(S, 32, 16, 64, 16, 512)
    Q = torch.randn(B, S, G, h, D, device="cuda").bfloat16()
    K = torch.randn(B, G, S, D, device="cuda").bfloat16()
    V = (torch.randn(B, G, S, D, device="cuda") * 1e5).bfloat16()
    rg = nv.select_topn_ranges_batched(torch.rand(B, S, G, meta.S_sel, device="cuda"), meta, 16, S)
    tune("SEL_FLAT", 0)
    out = {}
    for ks in (0, 1):
        tune("SEL_KSPLIT", ks), tune("SEL_KSPLIT_T1", 300), tune("SEL_KSPLIT_T2", 700)  # rows whole / in two / in four classes
        out[ks] = nv.selection_attention_hip(Q, K, V, rg).float()
    torch.cuda.synchronize()
    assert torch.isfinite(out[0]).all() and torch.isfinite(out[1]).all()
    assert V.float().abs().max().item() > 65504 * 2
    scale = out[0].abs().max().item()
    assert (out[0] - out[1]).abs().max().item() <= scale * 2.0 ** -7
    want = orc.sel_attention_masked(Q.float().cpu().numpy(), K.float().cpu().numpy(), V.float().cpu().numpy(), rg.cpu().numpy())
    assert np.abs(out[1].cpu().numpy() - want).max() <= 1e-2 * scale
