"""Regression: buffer descriptors built from K/V pointers whose LOW dword has bit 31 set.

`__builtin_amdgcn_readfirstlane` returns a signed int.  A kernel that ORs the low half of a pointer into a 64-bit descriptor base
without going through an unsigned 32-bit temporary sign-extends it into the high half whenever bit 31 of the address is set: the
wave then reads `0xffffffff'<low dword>` (a memory fault, seen in round 2 in the decode workgroup kernel).  Whether a
test allocation has that bit set is luck -- unless the allocation is at least 4 GiB long: then it contains addresses of both kinds and
the views can be placed on purpose.  Every kernel that builds a descriptor (`make_buffer_rsrc`: decode workgroup form, fused decode
step, block / query-tile / one-row prefill forms, their backward, the band kernels) runs here on such views and must reproduce,
bit for bit, what it computes on ordinary tensors (the oracle pins those elsewhere).
Reference semantics of the executors: nsa/core/attention_kernels.py:705-772."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nv():
    import nsa_vibe_amd

    assert torch.cuda.is_available(), "GPU tests need the MI355X box"
    return nsa_vibe_amd


@pytest.fixture(scope="module")
def arena():
    """one allocation of 4 GiB + 512 MiB: it holds a 2 GiB stretch of addresses with bit 31 set wherever it starts"""
    buf = torch.empty((4 << 30) + (512 << 20), dtype=torch.uint8, device="cuda")
    yield buf
    del buf
    torch.cuda.empty_cache()


class _Carver:
    """hands out 256-byte aligned views of the arena whose first AND last byte have bit 31 of the address set"""

    def __init__(self, arena):
        self.arena = arena
        base = arena.data_ptr()
        self.off = (0x80000000 - (base & 0xFFFFFFFF)) % (1 << 32)  # first offset with bit 31 set (and bits 0..30 clear)
        self.end = self.off + 0x80000000

    def take(self, like: torch.Tensor) -> torch.Tensor:
        nbytes = like.numel() * like.element_size()
        assert self.off + nbytes <= self.end and self.off + nbytes <= self.arena.numel(), "arena stretch exhausted"
        v = self.arena[self.off:self.off + nbytes].view(like.dtype).view(like.shape)
        self.off += (nbytes + 255) & ~255
        v.copy_(like)
        assert v.data_ptr() & 0x80000000, "view must have bit 31 of its address set"
        assert (v.data_ptr() + nbytes - 1) & 0x80000000
        return v


def _mk(g, *sh, dtype=torch.bfloat16):
    return torch.randn(*sh, device="cuda", generator=g).to(dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_decode_forms_on_bit31_addresses(nv, arena, dtype, tune):
    """the case that faulted in round 2 (decode workgroup kernel, V descriptor) + the fused decode step + the split-KV route"""
    g = torch.Generator(device="cuda")
    g.manual_seed(31)
    B, G, h, D, n, S_ctx = 5, 2, 6, 64, 16, 8192
    meta = nv.build_block_meta(S_ctx, 32, 16, 64, n, 512)
    Q, Kc = _mk(g, B, 1, G, h, D, dtype=dtype), _mk(g, B, G, meta.S_cmp, D, dtype=dtype)
    K, V = _mk(g, B, G, S_ctx + 100, D, dtype=dtype), _mk(g, B, G, S_ctx + 100, D, dtype=dtype)  # a cache longer than the context
    c = _Carver(arena)
    Kb, Vb, Kcb, Qb = c.take(K), c.take(V), c.take(Kc), c.take(Q)
    t = S_ctx - 1
    for unfused in (0, 1):
        tune("DECODE_UNFUSED", unfused)
        O0, r0 = nv.selection_decode_step(Q, Kc, K[:, :, :S_ctx], V[:, :, :S_ctx], meta, n, t)
        O1, r1 = nv.selection_decode_step(Qb, Kcb, Kb[:, :, :S_ctx], Vb[:, :, :S_ctx], meta, n, t)
        torch.cuda.synchronize()
        assert torch.equal(r0, r1) and torch.equal(O0, O1) and torch.isfinite(O1.float()).all()
    # the standalone decode executor (S = 1 rows, arbitrary ranges) in its workgroup form and its split-KV form
    rg = torch.randint(0, S_ctx, (B, 1, G, n, 2), device="cuda", generator=g, dtype=torch.int64).sort(dim=-1).values.int()
    for wg in (-1, 0):
        tune("DECODE_WG", wg)
        O0 = nv.selection_attention_hip(Q, K[:, :, :S_ctx], V[:, :, :S_ctx], rg)
        O1 = nv.selection_attention_hip(Qb, Kb[:, :, :S_ctx], Vb[:, :, :S_ctx], rg)
        torch.cuda.synchronize()
        assert torch.equal(O0, O1) and torch.isfinite(O1.float()).all()


def test_prefill_forms_on_bit31_addresses(nv, arena, tune):
    """block form (four-tile, flat, key-split), query-tile form, one-row form, MFMA backward, band kernels (window + compressed)"""
    g = torch.Generator(device="cuda")
    g.manual_seed(32)
    B, S, G, h, D, n = 2, 1536, 2, 6, 64, 16
    meta = nv.build_block_meta(S, 32, 16, 64, n, 512)
    Q, K, V = _mk(g, B, S, G, h, D), _mk(g, B, G, S, D), _mk(g, B, G, S, D)
    rg = nv.select_topn_ranges_batched(torch.rand(B, S, G, meta.S_sel, device="cuda", generator=g), meta, n, S)
    c = _Carver(arena)
    Qb, Kb, Vb = c.take(Q), c.take(K), c.take(V)
    forms = [("SEL_BLOCKS", 4, "SEL_FLAT", 0), ("SEL_BLOCKS", 4, "SEL_FLAT", 1), ("SEL_BLOCKS", 4, "SEL_KSPLIT", 1),
             ("SEL_BLOCKS", 0, "SEL_ROWS", 1), ("SEL_BLOCKS", 0, "SEL_ROWS", 0), ("SEL_BLOCKS", 0, "SEL_ROWS", 3)]
    tune("SEL_KSPLIT_T1", 400), tune("SEL_KSPLIT_T2", 900)  # the key-split form with rows in all three zones
    for a, av, b_, bv in forms:
        tune("SEL_FLAT", 0), tune("SEL_KSPLIT", 0), tune("SEL_ROWS", -1)
        tune(a, av), tune(b_, bv)
        O0 = nv.selection_attention_hip(Q, K, V, rg)
        O1 = nv.selection_attention_hip(Qb, Kb, Vb, rg)
        torch.cuda.synchronize()
        assert torch.equal(O0, O1) and torch.isfinite(O1.float()).all(), (a, av, b_, bv)
    tune("SEL_BLOCKS", -1), tune("SEL_FLAT", -1), tune("SEL_KSPLIT", -1), tune("SEL_ROWS", -1)
    # backward (its kernels stage Q / dO / K / V rows through descriptors of their own)
    dO = _mk(g, B, S, G, h, D)
    dOb = c.take(dO)

    def grads(q, k, v, d):
        q, k, v = (x.detach().requires_grad_(True) for x in (q, k, v))
        nv.selection_attention_hip(q, k, v, rg).backward(d)
        return q.grad, k.grad, v.grad

    for x, y in zip(grads(Q, K, V, dO), grads(Qb, Kb, Vb, dOb)):
        assert torch.equal(x, y) and torch.isfinite(y.float()).all()
    # band kernels: sliding window and the compressed branch
    Kc, Vc = _mk(g, B, G, meta.S_cmp, D), _mk(g, B, G, meta.S_cmp, D)
    Kcb, Vcb = c.take(Kc), c.take(Vc)
    assert torch.equal(nv.sliding_window_attention(Q, K, V, 512), nv.sliding_window_attention(Qb, Kb, Vb, 512))
    assert torch.equal(nv.batched_causal_attention_compressed(Q, Kc, Vc, 32, 16), nv.batched_causal_attention_compressed(Qb, Kcb, Vcb, 32, 16))
    # decode shapes of the band kernel (S = 1: the interval split over the waves)
    q1, q1b = Q[:, S - 1:S].contiguous(), c.take(Q[:, S - 1:S].contiguous())
    assert torch.equal(nv.sliding_window_attention(q1, K, V, 512, t0=S - 1), nv.sliding_window_attention(q1b, Kb, Vb, 512, t0=S - 1))
    torch.cuda.synchronize()
