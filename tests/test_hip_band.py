"""GPU parity: sliding-window / compressed branch attention (HIP band kernel, through the C ABI) vs the oracle and the
golden vectors of the reference's sliding_window_attention.  Tolerances as for the selected branch: 1e-3 fp32, 1e-2 bf16/f16."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 1e-3, torch.bfloat16: 1e-2, torch.float16: 1e-2}
W_INF = 2 ** 30


@pytest.fixture(scope="module")
def nv():
    import nsa_vibe_amd

    assert torch.cuda.is_available(), "GPU tests need the MI355X box"
    return nsa_vibe_amd


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.to(dtype) if dtype is not None else t


def rounded(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).float().numpy()


def rand_qkv(seed, B, S, G, h, Dk, Dv, S_kv):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((B, S, G, h, Dk), dtype=np.float32), rng.standard_normal((B, G, S_kv, Dk), dtype=np.float32),
            rng.standard_normal((B, G, S_kv, Dv), dtype=np.float32))


def check(nv, orc, Q, K, V, dtype, variant, band, tol=None, Kd=None, Vd=None):
    from nsa_vibe_amd.band_attention import band_attention_hip

    kd = Kd if Kd is not None else dev(K, dtype)
    vd = Vd if Vd is not None else dev(V, dtype)
    O, lse = band_attention_hip(dev(Q, dtype), kd, vd, variant=variant, return_lse=True, **band)
    ref, ref_lse = orc.band_attention(rounded(Q, dtype), rounded(K, dtype), rounded(V, dtype), return_lse=True, **band)
    got = O.float().cpu().numpy()
    assert O.dtype == dtype and got.shape == ref.shape and np.isfinite(got).all()
    err = np.abs(got - ref).max()
    # rows with one to three keys return (almost) a V row itself, |O| up to ~4: the bound is relative to the output magnitude
    # there (bf16 rounding of P and of O is 2^-9 each); for |O| <= 1 it is the absolute north-star tolerance
    bound = (tol or TOL[dtype]) * max(1.0, float(np.abs(ref).max()))
    assert err <= bound, f"max|dO|={err:.3e} (bound {bound:.3e}) dtype={dtype} variant={variant} band={band}"
    assert np.abs(got - ref).mean() <= 0.1 * (tol or TOL[dtype])
    lg, fin = lse.cpu().numpy(), np.isfinite(ref_lse)
    assert np.array_equal(np.isfinite(lg), fin)
    if fin.any():
        assert np.abs(lg[fin] - ref_lse[fin]).max() <= 2e-2
    return got


@pytest.mark.parametrize("name", ["a", "b", "c", "d"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_golden_sliding_window(nv, orc, name, dtype):
    g = load_golden("g13_win_" + name)
    got = check(nv, orc, g["Q"], g["K"], g["V"], dtype, 0, dict(w=int(g["w"])))
    assert np.abs(got - g["O"]).max() <= (1e-3 if dtype == torch.float32 else 6e-2)  # bf16: input rounding vs the fp32 reference
    if name in ("b", "d") and dtype == torch.bfloat16:  # D = 64: these run on the MFMA kernel
        check(nv, orc, g["Q"], g["K"], g["V"], dtype, 2, dict(w=int(g["w"])))


@pytest.mark.parametrize("name", ["a", "b", "c"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_golden_compressed(nv, orc, name, dtype):
    g = load_golden("g13_cmp_" + name)
    band = dict(a=int(g["l"]), dd=int(g["d"]), c=1)
    got = check(nv, orc, g["Q"], g["K"], g["V"], dtype, 0, band)
    assert np.abs(got - g["O"]).max() <= (1e-3 if dtype == torch.float32 else 6e-2)
    # the same output from a reference function: its masked selection executor over the compressed tokens, range [0, num_cmp(t)) per row
    assert np.abs(got - g["O_ref_selection_masked"]).max() <= (1e-3 if dtype == torch.float32 else 6e-2)
    O2 = nv.batched_causal_attention_compressed(dev(g["Q"], dtype), dev(g["K"], dtype), dev(g["V"], dtype), int(g["l"]), int(g["d"]))
    assert np.array_equal(O2.float().cpu().numpy(), got)
    # opt-in parity mode: the reference's own CPU result (key 0 only), bit for bit
    if g["K"].shape[2] > 0:
        Oq = nv.batched_causal_attention_compressed_first_key_parity(dev(g["Q"], dtype), dev(g["K"], dtype), dev(g["V"], dtype),
                                                                     int(g["l"]), int(g["d"]))
        want = torch.from_numpy(g["O_ref_quirk"]).to(dtype) if dtype != torch.float32 else torch.from_numpy(g["O_ref_quirk"])
        if dtype == torch.float32:
            assert np.array_equal(Oq.cpu().numpy(), g["O_ref_quirk"])
        else:  # V rounded to the activation dtype first, then copied: equals the rounded reference output
            assert torch.equal(Oq.cpu(), want)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,S,G,h,w", [(2, 4096, 2, 6, 512), (1, 4099, 8, 6, 512), (3, 700, 1, 6, 512), (1, 2100, 2, 4, 100),
                                       (1, 1500, 2, 5, 33), (2, 900, 1, 16, 64), (1, 777, 3, 1, 512), (2, 5000, 2, 8, 1)])
def test_sliding_window_mfma_vs_oracle(nv, orc, dtype, B, S, G, h, w):
    """m7c geometry and other head counts: NT=3 kernel (>= 2048 token groups) and NT=1 kernel, XCD-aware and plain mapping"""
    Q, K, V = rand_qkv(1400 + S + h, B, S, G, h, 64, 64, S)
    check(nv, orc, Q, K, V, dtype, 2, dict(w=w))


@pytest.mark.parametrize("dtype", [torch.bfloat16])
@pytest.mark.parametrize("B,S,G,h,l,d", [(2, 4096, 2, 6, 32, 16), (1, 16384, 2, 6, 32, 16), (1, 1000, 2, 6, 16, 8), (1, 300, 1, 4, 64, 64)])
def test_compressed_mfma_vs_oracle(nv, orc, dtype, B, S, G, h, l, d):
    S_cmp = (S - l) // d + 1
    Q, K, V = rand_qkv(1500 + S, B, S, G, h, 64, 64, S_cmp)
    check(nv, orc, Q, K, V, dtype, 2, dict(a=l, dd=d, c=1))


@pytest.mark.parametrize("B", [1, 4, 64])
@pytest.mark.parametrize("t0", [0, 37, 511, 5000])
def test_decode_rows(nv, orc, B, t0):
    """decode: one query at position t0 against a preallocated cache view (strided K/V); split-KV path for few rows"""
    G, h, S_max = 2, 6, 5100
    Q, K, V = rand_qkv(1600 + t0 + B, B, 1, G, h, 64, 64, S_max)
    Kc, Vc = dev(K, torch.bfloat16), dev(V, torch.bfloat16)
    S_kv = t0 + 1
    check(nv, orc, Q, K[:, :, :S_kv], V[:, :, :S_kv], torch.bfloat16, 0, dict(t0=t0, w=512), Kd=Kc[:, :, :S_kv], Vd=Vc[:, :, :S_kv])
    n_cmp = 0 if S_kv < 32 else (S_kv - 32) // 16 + 1
    check(nv, orc, Q, K[:, :, :n_cmp], V[:, :, :n_cmp], torch.bfloat16, 0, dict(t0=t0, a=32, dd=16, c=1), Kd=Kc[:, :, :n_cmp], Vd=Vc[:, :, :n_cmp])


def test_generic_matches_mfma_and_other_head_dims(nv, orc):
    Q, K, V = rand_qkv(1700, 1, 300, 2, 3, 128, 96, 300)
    check(nv, orc, Q, K, V, torch.float32, 0, dict(w=77))
    check(nv, orc, Q, K, V, torch.bfloat16, 0, dict(w=77))
    Q, K, V = rand_qkv(1701, 1, 300, 2, 6, 64, 64, 300)
    a = check(nv, orc, Q, K, V, torch.bfloat16, 1, dict(w=128))
    b = check(nv, orc, Q, K, V, torch.bfloat16, 2, dict(w=128))
    assert np.abs(a - b).max() <= 1e-2


def test_empty_inputs(nv):
    Q = torch.randn(1, 5, 2, 6, 64, device="cuda", dtype=torch.bfloat16)
    K = torch.empty(1, 2, 0, 64, device="cuda", dtype=torch.bfloat16)
    O = nv.sliding_window_attention(Q, K, K, 512)
    assert O.shape == (1, 5, 2, 6, 64) and not O.any()
    K = torch.randn(1, 2, 5, 64, device="cuda", dtype=torch.bfloat16)
    assert not nv.sliding_window_attention(Q, K, K, 0).any()
    with pytest.raises(RuntimeError):
        nv.sliding_window_attention(Q.float(), K, K, 4)


@pytest.mark.parametrize("band", [dict(w=40), dict(a=16, dd=8, c=1)])
def test_backward_vs_oracle(nv, orc, band):
    """autograd through the band forward + the selection backward kernels fed with one range per row"""
    from nsa_vibe_amd.band_attention import band_attention_hip

    B, S, G, h = 2, 150, 2, 6
    S_kv = S if "w" in band else (S - 16) // 8 + 1
    Q, K, V = rand_qkv(1800, B, S, G, h, 64, 64, S_kv)
    dO = np.random.default_rng(1801).standard_normal((B, S, G, h, 64), dtype=np.float32)
    dt = torch.bfloat16
    q, k, v = (dev(x, dt).requires_grad_(True) for x in (Q, K, V))
    band_attention_hip(q, k, v, **band).backward(dev(dO, dt))
    rq, rk, rv = orc.band_attention_bwd(rounded(Q, dt), rounded(K, dt), rounded(V, dt), rounded(dO, dt), **band)
    for got, ref, name in ((q.grad, rq, "dQ"), (k.grad, rk, "dK"), (v.grad, rv, "dV")):
        err = np.abs(got.float().cpu().numpy() - ref).max()
        assert err <= 2e-2 * max(1.0, np.abs(ref).max()), f"{name}: {err:.3e}"


@pytest.mark.parametrize("band,B,S,G", [(dict(w=512), 2, 2100, 4), (dict(a=32, dd=16, c=1), 2, 2100, 4), (dict(w=100), 1, 777, 3)])
def test_backward_dense_dq_kernel_vs_oracle(nv, orc, band, B, S, G):
    """the 48-slot dQ kernel (>= 2048 token groups) and its 16-slot variant, m7c head geometry"""
    from nsa_vibe_amd.band_attention import band_attention_hip

    h = 6
    S_kv = S if "w" in band else (S - 32) // 16 + 1
    Q, K, V = rand_qkv(1900 + S, B, S, G, h, 64, 64, S_kv)
    dO = np.random.default_rng(1901).standard_normal((B, S, G, h, 64), dtype=np.float32)
    dt = torch.bfloat16
    q, k, v = (dev(x, dt).requires_grad_(True) for x in (Q, K, V))
    band_attention_hip(q, k, v, variant=2, **band).backward(dev(dO, dt))
    rq, rk, rv = orc.band_attention_bwd(rounded(Q, dt), rounded(K, dt), rounded(V, dt), rounded(dO, dt), **band)
    for got, ref, name in ((q.grad, rq, "dQ"), (k.grad, rk, "dK"), (v.grad, rv, "dV")):
        g = got.float().cpu().numpy()
        assert np.isfinite(g).all(), name
        err = np.abs(g - ref).max()
        assert err <= 2e-2 * max(1.0, np.abs(ref).max()), f"{name}: {err:.3e}"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,S,G,h,band", [(2, 3000, 2, 6, dict(w=512)), (1, 5000, 8, 2, dict(a=32, dd=16, c=1)), (1, 900, 1, 16, dict(w=64)),
                                          (3, 300, 2, 5, dict(w=77))])
def test_band_mfma_head_dim_128(nv, orc, dtype, B, S, G, h, band):
    """D = 128: 2 column tiles per wave (32 slots) once there are enough token groups, the 16-slot kernel below that"""
    S_kv = S if "w" in band else (S - 32) // 16 + 1
    Q, K, V = rand_qkv(2100 + S + h, B, S, G, h, 128, 128, S_kv)
    check(nv, orc, Q, K, V, dtype, 2, band)


@pytest.mark.parametrize("B", [1, 5])
def test_decode_rows_head_dim_128(nv, orc, B):
    G, h, t0 = 2, 4, 3000
    Q, K, V = rand_qkv(2200 + B, B, 1, G, h, 128, 128, t0 + 1)
    check(nv, orc, Q, K, V, torch.bfloat16, 2, dict(t0=t0, w=512))
    n_cmp = (t0 + 1 - 32) // 16 + 1
    check(nv, orc, Q, K[:, :, :n_cmp], V[:, :, :n_cmp], torch.bfloat16, 2, dict(t0=t0, a=32, dd=16, c=1))
