"""LlamaBlockNSA and the TinyLM stack around the MI355X NSAAttention.

Same module tree / parameter names as the reference (nsa/model/llama_block_nsa.py:10-106: norm1, attn, norm2, mlp.fc1,
mlp.fc2; scripts/train_showcase.py:30-110 TinyLM: embed, blocks, norm_f, lm_head), so their checkpoints load unchanged.
RMSNorm, the MLP, the embedding and the LM head are plain PyTorch-ROCm ops (hipBLASLt GEMMs): they are outside the hot
path; the attention layer is the native one.  Beyond the reference's prefill-only `forward(x)`, the block and the model
carry a per-layer cache so that `decode(x_t, caches)` runs one token through all layers.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .kv_cache import NSA_KV
from .nsa_attention import NSAAttention


class RMSNorm(nn.Module):
    """x * rsqrt(mean(x^2) + eps) * weight (llama_block_nsa.py:10-19)"""

    def __init__(self, dim: int, eps: float = 1e-6) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.eps = eps

    def forward(self, x: torch.Tensor) -> torch.Tensor:
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
        out, kv = self.attn(self.norm1(x), kv, prefill=prefill)
        x = x + out
        x = x + self.mlp(self.norm2(x))
        return (x, kv) if return_kv else x


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

    def decode(self, tokens: torch.Tensor, caches: List[NSA_KV]) -> torch.Tensor:
        """tokens [B,1] -> logits [B,1,vocab]; every layer appends the token to its cache"""
        x = self.embed(tokens)
        for blk, kv in zip(self.blocks, caches):
            x = blk(x, kv, prefill=False)
        return self.lm_head(self.norm_f(x))
