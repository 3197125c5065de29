"""LlamaBlockNSA and the TinyLM stack around the MI355X NSAAttention.

Same module tree / parameter names as the reference (nsa/model/llama_block_nsa.py:10-106: norm1, attn, norm2, mlp.fc1,
mlp.fc2; scripts/train_showcase.py:30-110 TinyLM: embed, blocks, norm_f, lm_head), so their checkpoints load unchanged.
The MLP, the embedding and the LM head are plain PyTorch-ROCm ops (hipBLASLt GEMMs): they are outside the hot path; the
attention layer and RMSNorm (one native kernel each way) are native.  Beyond the reference's prefill-only `forward(x)`, the block and the model
carry a per-layer cache so that `decode(x_t, caches)` runs one token through all layers.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

import ctypes

from . import _lib
from .kv_cache import NSA_KV
from .nsa_attention import NSAAttention
from .selection_scorer import _stream, workspace


def _rmsnorm_native_ok(x: torch.Tensor, w: torch.Tensor) -> bool:
    """GPU tensors of a kernel dtype, rows the vectorised kernels take (dim % 8 == 0, <= 4096); NSA_HIP_EAGER_TRAIN=1 keeps the eager chain"""
    from .selection_scorer import _DT

    return (x.is_cuda and x.dtype in _DT and w.dtype == x.dtype and x.shape[-1] % 8 == 0 and x.shape[-1] <= 4096 and x.numel() > 0
            and os.getenv("NSA_HIP_EAGER_TRAIN", "0") != "1")


class _RMSNormFn(torch.autograd.Function):
    """RMSNorm rows through nsa_rmsnorm_rows / nsa_rmsnorm_rows_bwd (one kernel each way instead of the 6 + ~10 elementwise / reduction
    kernels of the eager chain; the forward rounds where the chain rounds, so both routes give the same activations)"""

    @staticmethod
    def forward(ctx, x, w, eps):
        from .selection_scorer import _DT

        xc = x.contiguous()
        wc = w.contiguous()
        y = torch.empty_like(xc)
        M, dim = xc.numel() // xc.shape[-1], xc.shape[-1]
        rc = _lib.lib().nsa_rmsnorm_rows(xc.data_ptr(), wc.data_ptr(), y.data_ptr(), M, dim, eps, _DT[xc.dtype], _stream(xc.device))
        _lib.check(rc, "nsa_rmsnorm_rows")
        ctx.save_for_backward(xc, wc)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        from .selection_scorer import _DT

        xc, wc = ctx.saved_tensors
        dyc = dy.contiguous()
        M, dim = xc.numel() // xc.shape[-1], xc.shape[-1]
        dx, dw = torch.empty_like(xc), torch.empty_like(wc)
        L = _lib.lib()
        ws = workspace(xc.device, L.nsa_rmsnorm_rows_bwd_workspace(M, dim) + 256, "rmsnorm_bwd")
        wptr = (ws.data_ptr() + 255) & ~255
        rc = L.nsa_rmsnorm_rows_bwd(xc.data_ptr(), wc.data_ptr(), dyc.data_ptr(), dx.data_ptr(), dw.data_ptr(), M, dim, ctx.eps, _DT[xc.dtype],
                                    wptr, ws.numel() - (wptr - ws.data_ptr()), _stream(xc.device))
        _lib.check(rc, "nsa_rmsnorm_rows_bwd")
        return dx, dw, None


class RMSNorm(nn.Module):
    """x * rsqrt(mean(x^2) + eps) * weight (llama_block_nsa.py:10-19)"""

    def __init__(self, dim: int, eps: float = 1e-6) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.eps = eps

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if _rmsnorm_native_ok(x, self.weight):
            return _RMSNormFn.apply(x, self.weight, float(self.eps))
        rms = x.pow(2).mean(dim=-1, keepdim=True).add(self.eps).rsqrt()
        return (x * rms) * self.weight


class MLP(nn.Module):
    """fc2(silu(fc1(x))), hidden = 4 dim, no biases (llama_block_nsa.py:22-30)"""

    def __init__(self, dim: int, hidden_mult: int = 4) -> None:
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden_mult * dim, bias=False)
        self.fc2 = nn.Linear(hidden_mult * dim, dim, bias=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.fc2(F.silu(self.fc1(x)))


class LlamaBlockNSA(nn.Module):
    def __init__(self, dim: int, n_heads: int, n_kv_groups: int, d_k: int, d_v: int, l: int = 32, d: int = 16, l_sel: int = 64,
                 n_sel: int = 16, w: int = 512, **attn_kwargs) -> None:
        super().__init__()
        self.norm1 = RMSNorm(dim)
        self.attn = NSAAttention(dim=dim, n_heads=n_heads, n_kv_groups=n_kv_groups, d_k=d_k, d_v=d_v, l=l, d=d, l_sel=l_sel,
                                 n_sel=n_sel, w=w, **attn_kwargs)
        self.norm2 = RMSNorm(dim)
        self.mlp = MLP(dim)

    def forward(self, x: torch.Tensor, kv: Optional[NSA_KV] = None, *, prefill: bool = True, return_kv: bool = False):
        """x [B,S,dim].  Reference form: `block(x)` = prefill over a fresh cache (llama_block_nsa.py:64-106).  With `kv` given the
        block appends to that cache (prefill=True for the first S tokens, prefill=False for one decode token)."""
        if kv is None:
            kv = self.attn.new_kv(x.shape[0], x.shape[1], x.device, x.dtype)
        self.attn._check_kv(x, kv)  # the native calls below walk kv.B sequences of the cache's dtype: it must match x
        if not prefill and x.shape[1] == 1 and self.attn._native_ok(x) and self.norm1.weight.dtype == x.dtype:
            y = self._decode_native(x, kv)
            return (y, kv) if return_kv else y
        if prefill and kv.t == 0 and self.attn._native_ok(x) and self.norm1.weight.dtype == x.dtype:
            y = self._prefill_native(x, kv)
            return (y, kv) if return_kv else y
        out, kv = self.attn(self.norm1(x), kv, prefill=prefill)
        x = x + out
        x = x + self.mlp(self.norm2(x))
        return (x, kv) if return_kv else x


    # ---- inference prefill: native RMSNorm + native layer core, residuals folded into the GEMMs (addmm) -----------------
    def _rmsnorm_native(self, x2d: torch.Tensor, norm: RMSNorm) -> torch.Tensor:
        from .selection_scorer import _DT

        y = torch.empty_like(x2d)
        rc = _lib.lib().nsa_rmsnorm_rows(x2d.data_ptr(), norm.weight.data_ptr(), y.data_ptr(), x2d.shape[0], x2d.shape[1], float(norm.eps),
                                         _DT[x2d.dtype], _stream(x2d.device))
        _lib.check(rc, "nsa_rmsnorm_rows")
        return y

    def _prefill_native(self, x: torch.Tensor, kv: NSA_KV) -> torch.Tensor:
        B, S, dim = x.shape
        x2 = x.reshape(B * S, dim)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        xn = self._rmsnorm_native(x2, self.norm1).view(B, S, dim)
        O, _ = self.attn._prefill_native(xn, kv, mix_only=True)                  # [B,S,H*Dv], before the output projection
        h = torch.addmm(x2, O.view(B * S, -1), self.attn.out.weight.t())          # x + out(O)
        u = F.silu(F.linear(self._rmsnorm_native(h, self.norm2), self.mlp.fc1.weight))
        return torch.addmm(h, u, self.mlp.fc2.weight.t()).view(B, S, dim)         # h + fc2(u)

    # ---- one native call per block and decode token (nsa_block_decode_step) ------------------------------------------
    def _block_desc(self):
        a = self.attn
        ldesc, _ = a._layer_desc()
        ps = [self.norm1.weight, self.norm2.weight, self.mlp.fc1.weight, self.mlp.fc2.weight]
        key = (id(ldesc),) + tuple((p.data_ptr(), p._version) for p in ps)
        if getattr(self, "_bdesc_key", None) != key:
            keep = [p.detach().contiguous() for p in ps]
            d = _lib.NsaBlockDesc()
            ctypes.memmove(ctypes.byref(d.attn), ctypes.byref(ldesc), ctypes.sizeof(ldesc))
            d.norm1_w, d.norm2_w, d.mlp_w1, d.mlp_w2 = (k.data_ptr() for k in keep)
            d.mlp_hidden, d.norm_eps = self.mlp.fc1.out_features, float(self.norm1.eps)
            self._bdesc_key, self._bdesc, self._bdesc_keep = key, d, keep
        return self._bdesc

    def _decode_native(self, x: torch.Tensor, kv: NSA_KV) -> torch.Tensor:
        a = self.attn
        t, B, dev = kv.t, x.shape[0], x.device
        kv.ensure_capacity(t + 1)
        if kv.meta.S_sel == 0:
            kv.ensure_meta(max(t + 1, a.l_sel))
        elif t + 1 > kv.meta.S_sel * a.l_sel:
            kv.ensure_meta(t + 1)
        L = _lib.lib()
        desc = self._block_desc()
        ctx = getattr(kv, "_blk_ctx", None)
        if ctx is None or ctx[0] is not desc:
            kd = a._kv_desc(kv)
            ws = workspace(dev, L.nsa_block_decode_step_workspace(ctypes.byref(desc), B, kd.S_max) + 256, "block_decode")
            wptr = (ws.data_ptr() + 255) & ~255
            ranges = torch.empty((B, a.n_kv_groups, a.n_sel, 2), dtype=torch.int32, device=dev)
            gates = torch.empty((B, 1, a.n_kv_groups, 3), dtype=torch.float32, device=dev)
            ctx = kv._blk_ctx = (desc, ctypes.byref(desc), ctypes.byref(kd), kd, ws, wptr, ws.numel() - (wptr - ws.data_ptr()), ranges, gates)
        _, desc_ref, kd_ref, _, _, wptr, wsize, ranges, gates = ctx
        cptr, crows, cvals = kv.meta.device_csc(dev)
        xc = x.reshape(B, a.dim)
        if not xc.is_contiguous():
            xc = xc.contiguous()
        y = torch.empty((B, 1, a.dim), dtype=x.dtype, device=dev)
        rc = L.nsa_block_decode_step(desc_ref, kd_ref, xc.data_ptr(), y.data_ptr(), t, cptr.data_ptr(), crows.data_ptr(), cvals.data_ptr(),
                                     int(kv.meta.S_sel), ranges.data_ptr(), gates.data_ptr(), wptr, wsize, _stream(dev))
        _lib.check(rc, "nsa_block_decode_step")
        S_raw = t + 1
        num_cmp = 0 if S_raw < a.l else (S_raw - a.l) // a.d + 1
        kv.t, kv.n_cmp = S_raw, num_cmp
        kv.append_reads(num_cmp, S_raw)
        a._last_ranges, a._last_gates = ranges, gates
        return y


class TinyLM(nn.Module):
    """embed -> n_layers x LlamaBlockNSA -> norm_f -> lm_head (scripts/train_showcase.py:30-110, without its checkpointing knobs)"""

    def __init__(self, vocab: int, dim: int, n_layers: int, n_heads: int, n_kv_groups: int, d_k: int, d_v: int, l: int, d: int,
                 l_sel: int, n_sel: int, w: int, **attn_kwargs) -> None:
        super().__init__()
        self.embed = nn.Embedding(vocab, dim)
        self.blocks = nn.ModuleList([LlamaBlockNSA(dim, n_heads, n_kv_groups, d_k, d_v, l, d, l_sel, n_sel, w, **attn_kwargs)
                                     for _ in range(n_layers)])
        self.norm_f = RMSNorm(dim)
        self.lm_head = nn.Linear(dim, vocab, bias=False)

    def forward(self, tokens: torch.Tensor) -> torch.Tensor:
        """tokens [B,S] -> logits [B,S,vocab] (training / scoring form of the reference)"""
        x = self.embed(tokens)
        for blk in self.blocks:
            x = blk(x)
        return self.lm_head(self.norm_f(x))

    # ---- serving form: per-layer caches ---------------------------------------------------------------------------
    def new_caches(self, B: int, S_max: int, device, dtype) -> List[NSA_KV]:
        return [blk.attn.new_kv(B, S_max, device, dtype) for blk in self.blocks]

    def prefill(self, tokens: torch.Tensor, caches: List[NSA_KV], last_only: bool = True) -> torch.Tensor:
        x = self.embed(tokens)
        for blk, kv in zip(self.blocks, caches):
            x = blk(x, kv, prefill=True)
        x = x[:, -1:] if last_only else x
        return self.lm_head(self.norm_f(x))

    def decode(self, tokens: torch.Tensor, caches: List[NSA_KV], return_next: bool = False):
        """tokens [B,1] -> logits [B,1,vocab] (and, with return_next, their argmax [B,1]); every layer appends the token to its cache.
        Inference on the GPU = ONE native call for the whole stack (nsa_model_decode_step)."""
        if self._native_decode_ok(tokens, caches):
            return self._decode_native(tokens, caches, return_next)
        x = self.embed(tokens)
        for blk, kv in zip(self.blocks, caches):
            x = blk(x, kv, prefill=False)
        logits = self.lm_head(self.norm_f(x))
        return (logits, logits.argmax(-1)) if return_next else logits

    def _native_decode_ok(self, tokens: torch.Tensor, caches: List[NSA_KV]) -> bool:
        w = self.embed.weight
        if not (tokens.is_cuda and tokens.shape[1] == 1 and w.is_cuda and self.norm_f.weight.dtype == w.dtype == self.lm_head.weight.dtype):
            return False
        probe = torch.empty((0, 1, w.shape[1]), dtype=w.dtype, device=w.device)
        return all(blk.attn._native_ok(probe) and blk.norm1.weight.dtype == w.dtype for blk in self.blocks) and \
            len({kv.t for kv in caches}) == 1 and len({kv._K_sel.shape[2] for kv in caches}) == 1

    def _decode_native(self, tokens: torch.Tensor, caches: List[NSA_KV], return_next: bool):
        from .selection_scorer import _DT  # noqa: F401

        a0, kv0 = self.blocks[0].attn, caches[0]
        t, B, dev = kv0.t, tokens.shape[0], tokens.device
        probe = torch.empty((B, 1, 0), dtype=self.embed.weight.dtype, device=dev)
        for blk, kv in zip(self.blocks, caches):
            blk.attn._check_kv(probe, kv)
            kv.ensure_capacity(t + 1)
        if kv0.meta.S_sel == 0:
            meta = kv0.ensure_meta(max(t + 1, a0.l_sel))
        elif t + 1 > kv0.meta.S_sel * a0.l_sel:
            meta = kv0.ensure_meta(t + 1)
        else:
            meta = kv0.meta
        L = _lib.lib()
        n = len(self.blocks)
        descs = [blk._block_desc() for blk in self.blocks]
        key = (tuple(id(d) for d in descs), tuple((id(kv), kv._K_sel.data_ptr()) for kv in caches), self.embed.weight.data_ptr(), self.lm_head.weight.data_ptr(),
               self.norm_f.weight.data_ptr())
        ctx = getattr(self, "_dec_ctx", None)
        if ctx is None or ctx[0] != key:
            barr = (_lib.NsaBlockDesc * n)(*descs)
            karr = (_lib.NsaKvDesc * n)(*[blk.attn._kv_desc(kv) for blk, kv in zip(self.blocks, caches)])
            ws = workspace(dev, L.nsa_model_decode_step_workspace(barr, n, B, karr[0].S_max) + 256, "model_decode")
            wptr = (ws.data_ptr() + 255) & ~255
            nxt = torch.empty((B, 1), dtype=torch.int32, device=dev)
            ctx = self._dec_ctx = (key, barr, karr, ws, wptr, ws.numel() - (wptr - ws.data_ptr()), nxt)
        _, barr, karr, _, wptr, wsize, nxt = ctx
        cptr, crows, cvals = meta.device_csc(dev)
        tok32 = tokens.reshape(B).to(torch.int32)
        logits = torch.empty((B, 1, self.lm_head.out_features), dtype=self.embed.weight.dtype, device=dev)
        rc = L.nsa_model_decode_step(barr, karr, n, tok32.data_ptr(), self.embed.weight.data_ptr(), self.norm_f.weight.data_ptr(),
                                     self.lm_head.weight.data_ptr(), self.lm_head.out_features, logits.data_ptr(),
                                     nxt.data_ptr() if return_next else None, t, cptr.data_ptr(), crows.data_ptr(), cvals.data_ptr(),
                                     int(meta.S_sel), wptr, wsize, _stream(dev))
        _lib.check(rc, "nsa_model_decode_step")
        S_raw = t + 1
        for blk, kv in zip(self.blocks, caches):
            a = blk.attn
            num_cmp = 0 if S_raw < a.l else (S_raw - a.l) // a.d + 1
            kv.t, kv.n_cmp, kv.meta = S_raw, num_cmp, meta
            kv.meta_seq_len = kv0.meta_seq_len
            kv.append_reads(num_cmp, S_raw)
        return (logits, nxt.to(torch.int64)) if return_next else logits
