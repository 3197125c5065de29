"""Sliding-window and compressed branch attention on MI355X (band attention kernel).

Operator names and argument order follow the reference (nsa/core/attention_kernels.py):
    sliding_window_attention(Q, K, V, w)                      :146-178  (banded causal softmax, keys [t-w+1 .. t])
    batched_causal_attention_compressed(Q, K_cmp, V_cmp, l, d) :106-143  (keys [0, num_cmp(t)), num_cmp from the
                                                                          emission schedule :118-121)
Layouts: Q [B,S,G,h,Dk], K/V [B,G,S_kv,D*] -> O [B,S,G,h,Dv] in V's dtype.  `t0` is the absolute position of query
row 0 (0 for prefill; a decode step passes S = 1 and t0 = position of the new token), so the same operator serves
the decode calls of nsa_attention.py:674-703.  The compressed branch uses the true softmax over the emitted tokens
-- the reference's per-token SDPA call (is_causal=True with one query, :139-141) attends key 0 only, a quirk that
is deliberately not reproduced.
Both are one C-ABI call (nsa_band_attn_fwd); the backward (nsa_band_attn_bwd) takes dQ with a dense query-major kernel and
dK/dV with the key-block-major selection backward kernels (the band written as one [lo, hi) range per row).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from .selection_scorer import _DT, _need_gpu, _stream, workspace

_W_INF = 2 ** 30


def band_ranges(S: int, S_kv: int, t0: int, a: int, dd: int, c: int, w: int, device) -> torch.Tensor:
    """[S,2] int32 key interval [lo, hi) of every query row (the rule of nsa_band_attn_fwd, include/nsa_sel_hip.h)."""
    e = torch.arange(S, device=device, dtype=torch.int64) + (t0 + 1 - a)
    hi = torch.where(e < 0, torch.zeros_like(e), torch.div(e.clamp_min(0), dd, rounding_mode="floor") + c).clamp(max=S_kv)
    lo = (hi - w).clamp_min(0)
    return torch.stack((lo, hi), dim=-1).to(torch.int32)


def _fwd(Q, K, V, band, scale, variant, want_lse):
    dev = _need_gpu(Q, K, V)
    if not (Q.dtype == K.dtype == V.dtype) or Q.dtype not in _DT:
        raise RuntimeError(f"band attention: Q/K/V must share a dtype in fp32/bf16/fp16 (got {Q.dtype},{K.dtype},{V.dtype})")
    if Q.dim() != 5 or K.dim() != 4 or V.dim() != 4:
        raise RuntimeError("band attention: expected Q[B,S,G,h,Dk] K[B,G,S_kv,Dk] V[B,G,S_kv,Dv]")
    B, S, G, h, Dk = Q.shape
    S_kv, Dv = K.shape[2], V.shape[3]
    if K.shape[:2] != (B, G) or V.shape[:3] != (B, G, S_kv) or K.shape[3] != Dk:
        raise RuntimeError("band attention: inconsistent shapes")
    t0, a, dd, c, w = band
    Qc = Q.contiguous()
    Kc = K if K.stride(-1) == 1 else K.contiguous()
    Vc = V if V.stride(-1) == 1 else V.contiguous()
    O = torch.empty((B, S, G, h, Dv), dtype=V.dtype, device=dev)
    lse = torch.empty((B, S, G, h), dtype=torch.float32, device=dev) if want_lse else None
    if O.numel() == 0:
        return O, lse, (Qc, Kc, Vc)
    L = _lib.lib()
    dt = _DT[Q.dtype]
    ws = workspace(dev, L.nsa_band_attn_fwd_workspace(B, S, G, h, Dk, Dv, dt), "band")
    rc = L.nsa_band_attn_fwd(Qc.data_ptr(), Kc.data_ptr() if S_kv else None, Vc.data_ptr() if S_kv else None, O.data_ptr(),
                             lse.data_ptr() if lse is not None else None, B, S, G, h, Dk, Dv, S_kv,
                             Kc.stride(0), Kc.stride(1), Kc.stride(2), Vc.stride(0), Vc.stride(1), Vc.stride(2),
                             int(t0), int(a), int(dd), int(c), int(min(w, _W_INF)), dt, float(scale) if scale else 0.0, int(variant),
                             ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, _stream(dev))
    _lib.check(rc, "nsa_band_attn_fwd")
    return O, lse, (Qc, Kc, Vc)


class _BandAttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Q, K, V, band, scale, variant):
        O, lse, (Qc, Kc, Vc) = _fwd(Q, K, V, band, scale, variant, True)
        ctx.save_for_backward(Qc, Kc, Vc, O, lse)
        ctx.band, ctx.scale = band, scale
        ctx.bwd_variant = 0 if variant != 1 else 1
        return O

    @staticmethod
    def backward(ctx, dO):
        Qc, Kc, Vc, O, lse = ctx.saved_tensors
        dev = Qc.device
        B, S, G, h, Dk = Qc.shape
        S_kv, Dv = Kc.shape[2], Vc.shape[3]
        t0, a, dd, c, w = ctx.band
        dO = dO.contiguous()
        dQ = torch.empty_like(Qc)
        dK = torch.empty((B, G, S_kv, Dk), dtype=torch.float32, device=dev)
        dV = torch.empty((B, G, S_kv, Dv), dtype=torch.float32, device=dev)
        L = _lib.lib()
        dt = _DT[Qc.dtype]
        ws = workspace(dev, L.nsa_band_attn_bwd_workspace(B, S, G, h, Dk, Dv, S_kv, dt, ctx.bwd_variant) + 256, "band_bwd")
        wptr = (ws.data_ptr() + 255) & ~255
        rc = L.nsa_band_attn_bwd(Qc.data_ptr(), Kc.data_ptr(), Vc.data_ptr(), O.data_ptr(), lse.data_ptr(), dO.data_ptr(), dQ.data_ptr(),
                                 dK.data_ptr(), dV.data_ptr(), B, S, G, h, Dk, Dv, S_kv,
                                 Kc.stride(0), Kc.stride(1), Kc.stride(2), Vc.stride(0), Vc.stride(1), Vc.stride(2),
                                 int(t0), int(a), int(dd), int(c), int(min(w, _W_INF)), dt, float(ctx.scale) if ctx.scale else 0.0,
                                 int(ctx.bwd_variant), wptr, ws.numel() - (wptr - ws.data_ptr()), _stream(dev))
        _lib.check(rc, "nsa_band_attn_bwd")
        return dQ, dK.to(Kc.dtype), dV.to(Vc.dtype), None, None, None


def band_attention_hip(Q, K, V, *, t0: int = 0, a: int = 0, dd: int = 1, c: int = 0, w: int = _W_INF,
                       scale: Optional[float] = None, variant: int = 0, return_lse: bool = False):
    """Row t attends keys [max(0, hi - w), hi), hi = (t0+t+1 >= a) ? min(S_kv, (t0+t+1-a)//dd + c) : 0."""
    band = (int(t0), int(a), int(dd), int(c), int(w))
    if torch.is_grad_enabled() and (Q.requires_grad or K.requires_grad or V.requires_grad):
        if return_lse:
            raise RuntimeError("return_lse is not available on the autograd path")
        if K.shape[2] == 0:
            return Q.new_zeros(Q.shape[:-1] + (V.shape[-1],)) + 0.0 * Q.sum()
        return _BandAttnFn.apply(Q, K, V, band, scale, variant)
    O, lse, _ = _fwd(Q, K, V, band, scale, variant, return_lse)
    return (O, lse) if return_lse else O


def sliding_window_attention(Q, K, V, w: int, *, t0: int = 0, scale: Optional[float] = None, variant: int = 0):
    """Keys [t-w+1 .. t] of row t (reference attention_kernels.py:146-178; w <= 0 or no keys -> zeros, :153-154)."""
    return band_attention_hip(Q, K, V, t0=t0, a=0, dd=1, c=0, w=max(int(w), 0), scale=scale, variant=variant)


def batched_causal_attention_compressed(Q, K_cmp, V_cmp, l: int, d: int, *, t0: int = 0, scale: Optional[float] = None,
                                        variant: int = 0):
    """Keys [0, num_cmp(t)), num_cmp(t) = 0 if t+1 < l else (t+1-l)//d + 1, clamped to S_cmp (attention_kernels.py:118-121)."""
    return band_attention_hip(Q, K_cmp, V_cmp, t0=t0, a=int(l), dd=int(d), c=1, w=_W_INF, scale=scale, variant=variant)


def batched_causal_attention_compressed_first_key_parity(Q, K_cmp, V_cmp, l: int, d: int, *, t0: int = 0):
    """PARITY MODE, opt-in: what the reference's own batched_causal_attention_compressed returns on CPU (attention_kernels.py:106-143).
    It evaluates every token through SDPA(is_causal=True) with ONE query (:139-141), so only compressed token 0 is visible:
    O[b,t,g,h,:] = V_cmp[b,g,0] when num_cmp(t) > 0, zeros otherwise (max |quirk - true softmax| ~ 5 on random inputs: g13 goldens keep it as
    O_ref_quirk).  The same first-key rule as the selection branch's packed executor, so it runs on nsa_sel_attn_first_key_parity with the
    single range [0, num_cmp(t)) per row.  Q and K_cmp do not influence the result.  Inference only."""
    from .selection_attention import selection_attention_first_key_parity

    B, S, G = Q.shape[:3]
    rg = band_ranges(S, K_cmp.shape[2], int(t0), int(l), int(d), 1, _W_INF, Q.device)  # [S, 2]: [0, num_cmp(t))
    ranges = rg.view(1, S, 1, 1, 2).expand(B, S, G, 1, 2).contiguous()
    return selection_attention_first_key_parity(Q, K_cmp, V_cmp, ranges)
