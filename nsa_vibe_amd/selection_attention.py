"""Selection attention executor on MI355X: `selection_attention_hip(Q, K, V, ranges)`.

Same call signature and result layout as every selection executor of the reference
(nsa/core/attention_kernels.py:181-186,273-278,391-396,705-710 and the native slot
nsa/kernels/cuda_sel_kernel/__init__.py:47-52):

    Q [B,S,G,h,Dk], K [B,G,S_kv,Dk], V [B,G,S_kv,Dv], ranges [B,S,G,n,2] int32/int64
    -> O [B,S,G,h,Dv] in V's dtype, on V's device.

Semantics = the reference's masked path (attention_kernels.py:705-772): softmax over the UNION
of the clamped ranges, non-causal inside the selected set, empty rows -> zeros.  Inputs are
borrowed (the caller's ranges tensor is never modified -- the reference clamps an int64 ranges
tensor in place, attention_kernels.py:723-724).  Differentiable: backward runs the HIP backward
kernel (dQ, dK, dV) with P recomputed from the forward's log-sum-exp.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from .selection_scorer import _DT, _need_gpu, _stream, workspace


def _prep_kv(X: torch.Tensor) -> torch.Tensor:
    return X if X.stride(-1) == 1 else X.contiguous()


def _prep_ranges(ranges: torch.Tensor) -> torch.Tensor:
    """[B,S,G,n,2] int32 contiguous.  Like the reference's range normaliser (kernels/triton_sel_kernel/__init__.py:17-39) extra singleton
    dimensions are squeezed away and a 4-D tensor is taken as batch-less; the clamp to [0,S_kv] happens in the kernels."""
    while ranges.dim() > 5:
        d = next((i for i in range(ranges.dim()) if ranges.size(i) == 1), None)
        if d is None:
            break
        ranges = ranges.squeeze(d)
    if ranges.dim() == 4:
        ranges = ranges.unsqueeze(0)
    if ranges.dim() != 5 or ranges.shape[-1] != 2:
        raise ValueError(f"ranges must be [B,S,G,n,2] (start,end), got {tuple(ranges.shape)}")
    if ranges.dtype != torch.int32:
        ranges = ranges.to(torch.int32)  # a copy: the caller's tensor stays untouched
    return ranges.contiguous()


def _fwd(Q, K, V, ranges, scale, variant, want_lse):
    dev = _need_gpu(Q, K, V, ranges)
    if not (Q.dtype == K.dtype == V.dtype) or Q.dtype not in _DT:
        raise RuntimeError(f"selection_attention_hip: Q/K/V must share a dtype in fp32/bf16/fp16 (got {Q.dtype},{K.dtype},{V.dtype})")
    if Q.dim() != 5 or K.dim() != 4 or V.dim() != 4 or ranges.dim() != 5 or ranges.shape[-1] != 2:
        raise RuntimeError("selection_attention_hip: expected Q[B,S,G,h,Dk] K[B,G,S_kv,Dk] V[B,G,S_kv,Dv] ranges[B,S,G,n,2]")
    B, S, G, h, Dk = Q.shape
    S_kv, Dv = K.shape[2], V.shape[3]
    n = ranges.shape[3]
    if K.shape[:2] != (B, G) or V.shape[:3] != (B, G, S_kv) or K.shape[3] != Dk or ranges.shape[:3] != (B, S, G):
        raise RuntimeError("selection_attention_hip: inconsistent shapes")
    Qc, Kc, Vc, rg = Q.contiguous(), _prep_kv(K), _prep_kv(V), _prep_ranges(ranges)
    O = torch.empty((B, S, G, h, Dv), dtype=V.dtype, device=dev)
    lse = torch.empty((B, S, G, h), dtype=torch.float32, device=dev) if want_lse else None
    if O.numel() == 0:
        return O, lse, (Qc, Kc, Vc, rg)
    L = _lib.lib()
    dt = _DT[Q.dtype]
    ws = workspace(dev, L.nsa_sel_attn_fwd_workspace_kv(B, S, G, h, Dk, Dv, S_kv, n, dt), "attn")
    rc = L.nsa_sel_attn_fwd(Qc.data_ptr(), Kc.data_ptr(), Vc.data_ptr(), rg.data_ptr(), O.data_ptr(),
                            lse.data_ptr() if lse is not None else None, B, S, G, h, Dk, Dv, S_kv, n,
                            Kc.stride(0), Kc.stride(1), Kc.stride(2), Vc.stride(0), Vc.stride(1), Vc.stride(2),
                            dt, float(scale) if scale else 0.0, int(variant),
                            ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, _stream(dev))
    _lib.check(rc, "nsa_sel_attn_fwd")
    return O, lse, (Qc, Kc, Vc, rg)


class _SelAttnFn(torch.autograd.Function):
    """autograd wrapper (pattern of the reference's Triton wrapper,
    nsa/kernels/triton_sel_kernel/__init__.py:125-142: save Q,K,V,ranges; backward -> dQ,dK,dV,None)."""

    @staticmethod
    def forward(ctx, Q, K, V, ranges, scale, variant):
        O, lse, (Qc, Kc, Vc, rg) = _fwd(Q, K, V, ranges, scale, variant, True)
        ctx.save_for_backward(Qc, Kc, Vc, rg, O, lse)
        ctx.scale = scale
        ctx.bwd_variant = 0 if variant != 1 else 1  # variant 1 keeps the generic kernels on both passes
        return O

    @staticmethod
    def backward(ctx, dO):
        Qc, Kc, Vc, rg, O, lse = ctx.saved_tensors
        dev = Qc.device
        B, S, G, h, Dk = Qc.shape
        S_kv, Dv = Kc.shape[2], Vc.shape[3]
        dO = dO.contiguous()
        dQ = torch.empty_like(Qc)
        dK = torch.empty((B, G, S_kv, Dk), dtype=torch.float32, device=dev)
        dV = torch.empty((B, G, S_kv, Dv), dtype=torch.float32, device=dev)
        L = _lib.lib()
        ws = workspace(dev, L.nsa_sel_attn_bwd_workspace(B, S, G, h, Dk, Dv, S_kv, _DT[Qc.dtype], ctx.bwd_variant), "attn_bwd")
        rc = L.nsa_sel_attn_bwd(Qc.data_ptr(), Kc.data_ptr(), Vc.data_ptr(), rg.data_ptr(), O.data_ptr(),
                                lse.data_ptr(), dO.data_ptr(), dQ.data_ptr(), dK.data_ptr(), dV.data_ptr(),
                                B, S, G, h, Dk, Dv, S_kv, rg.shape[3],
                                Kc.stride(0), Kc.stride(1), Kc.stride(2), Vc.stride(0), Vc.stride(1), Vc.stride(2),
                                _DT[Qc.dtype], float(ctx.scale) if ctx.scale else 0.0, int(ctx.bwd_variant),
                                ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, _stream(dev))
        _lib.check(rc, "nsa_sel_attn_bwd")
        return dQ, dK.to(Kc.dtype), dV.to(Vc.dtype), None, None, None


def selection_attention_hip(Q: torch.Tensor, K: torch.Tensor, V: torch.Tensor, ranges: torch.Tensor, *,
                            scale: Optional[float] = None, variant: int = 0, return_lse: bool = False):
    """The MI355X selection executor (drop-in for selection_attention_cuda / grouped_selection_attention_masked).

    variant: 0 auto (MFMA kernel when dtype/shape allow, else the generic kernel), 1 generic, 2 MFMA."""
    if torch.is_grad_enabled() and (Q.requires_grad or K.requires_grad or V.requires_grad):
        if return_lse:
            raise RuntimeError("return_lse is not available on the autograd path")
        return _SelAttnFn.apply(Q, K, V, ranges, scale, variant)
    O, lse, _ = _fwd(Q, K, V, ranges, scale, variant, return_lse)
    return (O, lse) if return_lse else O


def selection_attention_first_key_parity(Q: torch.Tensor, K: torch.Tensor, V: torch.Tensor, ranges: torch.Tensor) -> torch.Tensor:
    """PARITY MODE, opt-in: what the reference's default packed / gather executors return (grouped_selection_attention_packed,
    attention_kernels.py:273-388; grouped_selection_attention, :181-226).  Their SDPA call is `is_causal=True` with one query, so only
    the first gathered key is visible and O[b,t,g,h,:] = V[b,g,start of the first non-empty range] for every head (zeros when the row
    has none).  Same executor signature as selection_attention_hip; Q and K do not influence the result.  Inference only."""
    dev = _need_gpu(Q, K, V)
    if torch.is_grad_enabled() and V.requires_grad:
        raise RuntimeError("selection_attention_first_key_parity is an inference-only parity mode")
    B, S, G, h, _ = Q.shape
    S_kv, Dv = V.shape[2], V.shape[3]
    Vv, rg = _prep_kv(V), _prep_ranges(ranges)
    O = torch.empty((B, S, G, h, Dv), dtype=V.dtype, device=dev)
    rc = _lib.lib().nsa_sel_attn_first_key_parity(Vv.data_ptr(), rg.data_ptr(), O.data_ptr(), B, S, G, h, Dv, S_kv, rg.shape[3],
                                                  Vv.stride(0), Vv.stride(1), Vv.stride(2), _DT[V.dtype], _stream(dev))
    _lib.check(rc, "nsa_sel_attn_first_key_parity")
    return O


def selection_attention_head_causal_parity(Q: torch.Tensor, K: torch.Tensor, V: torch.Tensor, ranges: torch.Tensor, *,
                                           scale: Optional[float] = None) -> torch.Tensor:
    """PARITY MODE, opt-in: what NSAAttention._sdpa_over_ranges returns (nsa/core/nsa_attention.py:1779-1855), the gather route the
    reference's decode and sequential prefill fall to (and the only one left under NSA_FORCE_PARITY=1).  It hands SDPA the h heads of a
    group as the query LENGTH with is_causal=True, so head i attends the first i+1 tokens of the gathered union (ascending token
    order); rows without a token give zeros.  Executor signature ([B,S,G,...] tensors); inference only."""
    dev = _need_gpu(Q, K, V)
    if torch.is_grad_enabled() and (Q.requires_grad or K.requires_grad or V.requires_grad):
        raise RuntimeError("selection_attention_head_causal_parity is an inference-only parity mode")
    if not (Q.dtype == K.dtype == V.dtype) or Q.dtype not in _DT:
        raise RuntimeError("selection_attention_head_causal_parity: Q/K/V must share a dtype in fp32/bf16/fp16")
    B, S, G, h, Dk = Q.shape
    S_kv, Dv = V.shape[2], V.shape[3]
    Qc, Kk, Vv, rg = Q.contiguous(), _prep_kv(K), _prep_kv(V), _prep_ranges(ranges)
    if rg.shape[:3] != (B, S, G) or Kk.shape[:3] != (B, G, S_kv):
        raise RuntimeError("selection_attention_head_causal_parity: inconsistent shapes")
    O = torch.empty((B, S, G, h, Dv), dtype=V.dtype, device=dev)
    rc = _lib.lib().nsa_sel_attn_head_causal_parity(Qc.data_ptr(), Kk.data_ptr(), Vv.data_ptr(), rg.data_ptr(), O.data_ptr(), B, S, G, h, Dk,
                                                    Dv, S_kv, rg.shape[3], Kk.stride(0), Kk.stride(1), Kk.stride(2), Vv.stride(0),
                                                    Vv.stride(1), Vv.stride(2), _DT[Q.dtype], float(scale) if scale else 0.0, _stream(dev))
    _lib.check(rc, "nsa_sel_attn_head_causal_parity")
    return O


def selection_decode_step(Q: torch.Tensor, K_cmp: torch.Tensor, K: torch.Tensor, V: torch.Tensor, meta, n_top: int, t_token: int,
                          *, scale: Optional[float] = None, out: Optional[torch.Tensor] = None,
                          ranges_out: Optional[torch.Tensor] = None):
    """One decode step of the selected branch in a single native call.

    Q [B,1,G,h,Dk], K_cmp [B,G,S_cmp,Dk], K/V [B,G,S_kv,D] (views of a preallocated cache are fine), meta = BlockMeta
    covering t_token.  Returns (O [B,1,G,h,Dv], ranges [B,G,n_top,2] int32) = what the decode branch of the
    reference computes with compute_pcmp_all -> map_pcmp_to_pslc_batched -> sum(dim=3) -> select_topn_ranges ->
    selection executor (nsa/core/nsa_attention.py:651-672, 704-830), sequential-selector semantics."""
    dev = _need_gpu(Q, K_cmp, K, V)
    B, S, G, h, Dk = Q.shape
    if S != 1:
        raise RuntimeError("selection_decode_step: decode requires S == 1")
    S_kv, Dv = K.shape[2], V.shape[3]
    S_cmp, S_sel = K_cmp.shape[2], meta.S_sel
    Qc, Kc, Kk, Vv = Q.contiguous(), _prep_kv(K_cmp), _prep_kv(K), _prep_kv(V)
    O = out if out is not None else torch.empty((B, 1, G, h, Dv), dtype=V.dtype, device=dev)
    rg = ranges_out if ranges_out is not None else torch.empty((B, G, n_top, 2), dtype=torch.int32, device=dev)
    L = _lib.lib()
    dt = _DT[Q.dtype]
    ws = workspace(dev, L.nsa_sel_decode_step_workspace(B, G, h, Dk, Dv, S_cmp, S_sel, n_top, dt) + 16, "decode")
    wptr = (ws.data_ptr() + 15) & ~15
    cptr, crows, cvals = meta.device_csc(dev)
    rc = L.nsa_sel_decode_step(Qc.data_ptr(), Kc.data_ptr(), Kk.data_ptr(), Vv.data_ptr(), cptr.data_ptr(), crows.data_ptr(),
                               cvals.data_ptr(), rg.data_ptr(), O.data_ptr(), B, G, h, Dk, Dv, S_cmp, S_sel, S_kv,
                               int(meta.l), int(meta.d), int(meta.l_sel), int(n_top), int(t_token),
                               Kc.stride(0), Kc.stride(1), Kc.stride(2), Kk.stride(0), Kk.stride(1), Kk.stride(2),
                               Vv.stride(0), Vv.stride(1), Vv.stride(2), dt, float(scale) if scale else 0.0,
                               wptr, ws.numel() - (wptr - ws.data_ptr()), _stream(dev))
    _lib.check(rc, "nsa_sel_decode_step")
    return O, rg


def select_and_attend(p_grp: torch.Tensor, Q: torch.Tensor, K: torch.Tensor, V: torch.Tensor, meta, n_top: int, *,
                      mode: str = "batched", t0: int = 0, force_init: bool = True, force_local: int = 2,
                      scale: Optional[float] = None, return_lse: bool = False):
    """Prefill (inference): top-n selection from the group scores and the selection attention in ONE native call
    (nsa_sel_select_attn_fwd: the select kernel and the attention kernel back to back, or -- tuning switch SEL_FUSE = 1 -- the selector
    inside the attention kernel on the MFMA route).  p_grp [B,S,G,S_sel] fp32.
    mode "batched" = select_topn_ranges_batched semantics (selection_scorer.py:255-362), "sequential" = select_topn_ranges per
    row at token t0 + s (:124-249).  Returns (ranges [B,S,G,W,2] int32, O [B,S,G,h,Dv]) -- bit-identical to
    select_topn_ranges_batched / select_topn_ranges_rows followed by selection_attention_hip."""
    from .selection_scorer import batched_ranges_width

    dev = _need_gpu(p_grp, Q, K, V)
    if torch.is_grad_enabled() and (Q.requires_grad or K.requires_grad or V.requires_grad):
        raise RuntimeError("select_and_attend is the inference form; with autograd use select_topn_ranges_* + selection_attention_hip")
    B, S, G, h, Dk = Q.shape
    S_kv, Dv, S_sel = K.shape[2], V.shape[3], p_grp.shape[-1]
    if p_grp.shape[:3] != (B, S, G) or p_grp.dtype != torch.float32:
        raise RuntimeError("select_and_attend: p_grp must be fp32 [B,S,G,S_sel]")
    if mode == "batched":
        md, W = _lib.NSA_SEL_BATCHED, batched_ranges_width(S_sel, meta.l_sel, n_top, S, force_init, force_local)
    elif mode == "sequential":
        md, W = _lib.NSA_SEL_SEQUENTIAL, n_top
    else:
        raise ValueError("mode must be 'batched' or 'sequential'")
    Qc, Kc, Vc, pg = Q.contiguous(), _prep_kv(K), _prep_kv(V), p_grp.contiguous()
    ranges = torch.empty((B, S, G, W, 2), dtype=torch.int32, device=dev)
    O = torch.empty((B, S, G, h, Dv), dtype=V.dtype, device=dev)
    lse = torch.empty((B, S, G, h), dtype=torch.float32, device=dev) if return_lse else None
    if O.numel() == 0 or W == 0:
        O.zero_()
        return (ranges, O, lse) if return_lse else (ranges, O)
    L = _lib.lib()
    dt = _DT[Q.dtype]
    ws = workspace(dev, L.nsa_sel_attn_fwd_workspace_kv(B, S, G, h, Dk, Dv, S_kv, W, dt), "attn")
    rc = L.nsa_sel_select_attn_fwd(pg.data_ptr(), int(t0), None, S_sel, int(meta.l_sel), int(n_top), int(bool(force_init)), int(force_local),
                                   md, S, ranges.data_ptr(), W, Qc.data_ptr(), Kc.data_ptr(), Vc.data_ptr(), O.data_ptr(),
                                   lse.data_ptr() if lse is not None else None, B, S, G, h, Dk, Dv, S_kv,
                                   Kc.stride(0), Kc.stride(1), Kc.stride(2), Vc.stride(0), Vc.stride(1), Vc.stride(2),
                                   dt, float(scale) if scale else 0.0, ws.data_ptr() if ws is not None else None,
                                   ws.numel() if ws is not None else 0, _stream(dev))
    _lib.check(rc, "nsa_sel_select_attn_fwd")
    return (ranges, O, lse) if return_lse else (ranges, O)


# the reference's executor names, bound to the HIP implementation
grouped_selection_attention_masked = selection_attention_hip
selection_attention_cuda = selection_attention_hip
# the "merged-range varlen gather" executors (attention_kernels.py:391-542, 545-702) compute the same masked softmax through an FA-2
# varlen pack; here the gather is what the kernels do anyway
selection_attention_varlen_all = selection_attention_hip
selection_attention_varlen_all_v2 = selection_attention_hip
# the default packed / gather executors are first-gathered-key functions (see selection_attention_first_key_parity): their names are
# bound to that parity mode, NOT to the semantic executor, so that code calling them by name keeps the reference's numbers
grouped_selection_attention_packed = selection_attention_first_key_parity
grouped_selection_attention = selection_attention_first_key_parity


def hip_sel_available() -> bool:
    """Counterpart of cuda_sel_available() (nsa/kernels/cuda_sel_kernel/__init__.py:43-44)."""
    try:
        _lib.lib()
    except (ImportError, OSError, AttributeError):
        return False
    return torch.cuda.is_available()
