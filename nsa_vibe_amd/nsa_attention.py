"""NSAAttention with the selected branch on MI355X.

Same operator surface as the reference module (nsa/core/nsa_attention.py:188-206 constructor, :509 forward(x, kv, *,
prefill) -> (out, kv), :407-445 monitor getters) and the same state-dict keys (W_Q, W_K_sel, W_V_sel, W_K_win,
W_V_win, W_K_cmp, W_V_cmp, out, gate.fc1, gate.fc2), so a checkpoint of the reference loads unchanged.

What runs where:
  * selected branch (the hot path): scores -> top-n ranges -> selection attention on the HIP kernels
    (selection_scorer.py / selection_attention.py); no Python loop over t, no host sync, no O(S^2) mask.
    Semantics = the reference's masked route (NSA_FORCE_SEL_MASK=1, the production setting).
  * compressed and sliding branches: the HIP band-attention kernel (band_attention.py).  They use true causal
    softmax attention; the reference's default SDPA routes for them degenerate to "first key only" for
    single-query calls (is_causal=True with L_q=1) -- a quirk that is deliberately not reproduced.
  * projections, RoPE, avg-pool phi, gate MLP: plain PyTorch-ROCm ops.
Selector semantics: `selector="sequential"` (reference default prefill, :1521-1723, and decode) or `"batched"`
(NSA_PREFILL_BATCHED=1, :978-1448).
"""
from __future__ import annotations

import ctypes
import math
import os
import warnings
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .band_attention import batched_causal_attention_compressed, sliding_window_attention
from .kv_cache import NSA_KV
from .selection_attention import (
    select_and_attend,
    selection_attention_first_key_parity,
    selection_attention_head_causal_parity,
    selection_attention_hip,
    selection_decode_step,
)
from .selection_scorer import (_DT, _stream, select_topn_ranges_batched, select_topn_ranges_rows, selection_scores, selection_scores_select,
                               workspace)

_ORIG_SELECTORS = (select_topn_ranges_batched, select_topn_ranges_rows, selection_scores)


def _scores_and_ranges(Q, K_cmp, meta, n_sel, selector, S, scale):
    """group scores -> top-n ranges of every row.  One native call (nsa_sel_scores_select: on the MFMA scorer's route one launch) -- unless
    somebody has replaced the scorer / selector functions of THIS module (the reference's tests patch these names on nsa.core.nsa_attention,
    nsa/tests/test_causality_asserts.py:54-56): a patched function must be the one that runs."""
    g = globals()
    if (g["select_topn_ranges_batched"], g["select_topn_ranges_rows"], g["selection_scores"]) == _ORIG_SELECTORS:
        return selection_scores_select(Q, K_cmp, meta, n_sel, mode=selector, scale=scale)[1]
    p_grp = g["selection_scores"](Q, K_cmp, meta, scale, causal_skip=True, leave_skipped=True)
    if selector == "batched":
        return g["select_topn_ranges_batched"](p_grp, meta, n_sel, S, True, 2)
    return g["select_topn_ranges_rows"](p_grp, meta, n_sel, 0, True, 2)


def apply_rope(x: torch.Tensor, pos: torch.Tensor, base: float = 10000.0, scale: float = 1.0) -> torch.Tensor:
    """Rotary embedding over adjacent pairs of the last dim, angles in fp32 (formula of nsa/core/rope.py:16-51).
    x [..., S, D], pos [S]."""
    D = x.shape[-1]
    inv_freq = base ** (-2.0 * torch.arange(D // 2, device=x.device, dtype=torch.float32) / D)
    ang = (pos.to(torch.float32) / (scale if scale > 0 else 1.0)).unsqueeze(-1) * inv_freq  # [S, D/2]
    sin, cos = torch.sin(ang).to(x.dtype), torch.cos(ang).to(x.dtype)
    x2 = x.reshape(*x.shape[:-1], D // 2, 2)
    x0, x1 = x2[..., 0], x2[..., 1]
    return torch.stack((x0 * cos - x1 * sin, x0 * sin + x1 * cos), dim=-1).reshape(x.shape)


def avg_pool_phi(K_rope: torch.Tensor, V: torch.Tensor, l: int, d: int):
    """phi = mean over windows of l tokens, stride d, on RoPE'd K and raw V (nsa/core/compress_pool.py:9-38)."""
    B, G, S, Dk = K_rope.shape
    if S < l:
        return K_rope[:, :, :0], V[:, :, :0]
    pool = lambda X: F.avg_pool1d(X.reshape(B * G, S, -1).transpose(1, 2), kernel_size=l, stride=d).transpose(1, 2)  # noqa: E731
    Kc, Vc = pool(K_rope), pool(V)
    return Kc.reshape(B, G, Kc.shape[1], Dk), Vc.reshape(B, G, Vc.shape[1], V.shape[-1])


def _top2_gap(g: torch.Tensor) -> torch.Tensor:
    """largest minus second largest of the 3 gate logits (torch.topk(k=2) of a [..., 3] tensor costs a 150 us kernel)"""
    a, b, c = g[..., 0], g[..., 1], g[..., 2]
    hi = torch.maximum(torch.maximum(a, b), c)
    mid = torch.maximum(torch.minimum(a, b), torch.minimum(torch.maximum(a, b), c))
    return hi - mid


class GateMLP(nn.Module):
    """Same parameters / init as the reference gate (nsa_attention.py:32-82): softmax over (cmp, sel, win)."""

    def __init__(self, d_k: int, hidden: Optional[int] = None):
        super().__init__()
        hidden = hidden or max(1, d_k // 2)
        self.fc1 = nn.Linear(d_k, hidden)
        self.fc2 = nn.Linear(hidden, 3)
        nn.init.xavier_uniform_(self.fc2.weight, gain=0.1)
        nn.init.zeros_(self.fc2.bias)
        # debugging overrides of the reference, read once at construction like there (nsa_attention.py:42-48)
        self._force_uniform_gate = os.getenv("NSA_FORCE_UNIFORM_GATE", "0").lower() in ("1", "true", "yes")
        fb = (os.getenv("NSA_FORCE_BRANCH") or "").strip().lower()
        self._force_branch = fb if fb in ("cmp", "sel", "win") else None
        self._forced_fc2 = None

    def forced(self) -> bool:
        return self._force_uniform_gate or self._force_branch is not None

    def fc2_params(self):
        """(weight, bias) of fc2 as the kernels should see them: the real parameters, or -- under NSA_FORCE_UNIFORM_GATE / NSA_FORCE_BRANCH --
        constants that make the same kernels emit exactly 1/3,1/3,1/3 resp. the one-hot of the forced branch (zero weight; bias 0 resp.
        +-1000, which is beyond the one-hot threshold of 50)"""
        if not self.forced():
            return self.fc2.weight, self.fc2.bias
        w = self.fc2.weight
        key = (w.device, w.dtype)
        if self._forced_fc2 is None or self._forced_fc2[0] != key:
            b = torch.zeros(3, device=w.device, dtype=w.dtype)
            if not self._force_uniform_gate:
                b.fill_(-1000.0)
                b[("cmp", "sel", "win").index(self._force_branch)] = 1000.0
            self._forced_fc2 = (key, torch.zeros_like(w).detach(), b)
        return self._forced_fc2[1], self._forced_fc2[2]

    def forward(self, q_pooled: torch.Tensor, tau: float = 1.0) -> torch.Tensor:
        if self._force_uniform_gate:  # reference :51-57
            return torch.full((*q_pooled.shape[:-1], 3), 1.0 / 3.0, device=q_pooled.device, dtype=q_pooled.dtype)
        if self._force_branch is not None:  # reference :58-70
            one = torch.zeros((*q_pooled.shape[:-1], 3), device=q_pooled.device, dtype=q_pooled.dtype)
            one[..., ("cmp", "sel", "win").index(self._force_branch)] = 1.0
            return one
        g = self.fc2(F.silu(self.fc1(q_pooled))) / max(tau, 1e-6)
        p = F.softmax(g, dim=-1)
        peaked = _top2_gap(g.detach()) > 50.0  # hard one-hot when extremely peaked (reference :70-81), sync free
        one_hot = F.one_hot(torch.argmax(g, dim=-1), 3).to(p.dtype)
        return torch.where(peaked.unsqueeze(-1), one_hot, p)


def _set_plain(module, name, value):
    """attribute of a module that is neither parameter, buffer nor submodule, set without nn.Module.__setattr__ (whose isinstance checks
    against Parameter cost tens of microseconds per assignment -- more host time than a decode step's three launches)"""
    object.__setattr__(module, name, value)


def _gate_probs_fn(q_pooled, w1, b1, w2, b2, tau):
    """functional form of GateMLP.forward (used to take the gate MLP's gradient in _GateCombineFn.backward)"""
    g = F.linear(F.silu(F.linear(q_pooled, w1, b1)), w2, b2) / max(tau, 1e-6)
    p = F.softmax(g, dim=-1)
    peaked = _top2_gap(g.detach()) > 50.0
    one_hot = F.one_hot(torch.argmax(g, dim=-1), 3).to(p.dtype)
    return torch.where(peaked.unsqueeze(-1), one_hot, p)


class _RopeAppendFn(torch.autograd.Function):
    """fused-projection epilogue (RoPE + append into the preallocated caches) as a differentiable op: returns Q and the S
    appended rows of the six caches (views of the cache buffers); backward = nsa_rope_cache_append_bwd."""

    @staticmethod
    def forward(ctx, proj, module, kv, S):
        L, dev = _lib.lib(), proj.device
        desc, _ = module._layer_desc()
        kd = module._kv_desc(kv)
        B = proj.shape[0]
        proj = proj.contiguous()
        Q = torch.empty((B, S, module.n_kv_groups, module.h_per_group, module.d_k), dtype=proj.dtype, device=dev)
        _lib.check(L.nsa_rope_cache_append(ctypes.byref(desc), ctypes.byref(kd), proj.data_ptr(), Q.data_ptr(), S, 0, _stream(dev)),
                   "nsa_rope_cache_append")
        ctx.module, ctx.B, ctx.S, ctx.NT = module, B, S, proj.shape[-1]
        return (Q, kv._K_sel[:, :, :S], kv._V_sel[:, :, :S], kv._K_win[:, :, :S], kv._V_win[:, :, :S], kv._K_raw[:, :, :S],
                kv._V_raw[:, :, :S])

    @staticmethod
    def backward(ctx, dQ, *dcache):
        m, B, S = ctx.module, ctx.B, ctx.S
        L, dev = _lib.lib(), dQ.device if dQ is not None else dcache[0].device
        desc, _ = m._layer_desc()
        ref = dQ if dQ is not None else next(d for d in dcache if d is not None)
        if dQ is None:
            dQ = torch.zeros((B, S, m.n_kv_groups, m.h_per_group, m.d_k), dtype=ref.dtype, device=dev)
        keep = [dQ.contiguous()] + [None if d is None else d.contiguous() for d in dcache]
        dproj = torch.empty((B, S, ctx.NT), dtype=ref.dtype, device=dev)
        ptr = [None if t is None else t.data_ptr() for t in keep]
        _lib.check(L.nsa_rope_cache_append_bwd(ctypes.byref(desc), B, S, 0, *ptr, dproj.data_ptr(), _stream(dev)), "nsa_rope_cache_append_bwd")
        return dproj, None, None, None


class _CmpPoolFn(torch.autograd.Function):
    """compressed-token pooling phi over the raw-token cache as a differentiable op (K_raw / V_raw are taken as inputs only
    to connect the graph; the kernel reads the cache they are views of)"""

    @staticmethod
    def forward(ctx, K_raw, V_raw, module, kv, S):
        L, dev = _lib.lib(), K_raw.device
        desc, _ = module._layer_desc()
        kd = module._kv_desc(kv)
        n_cmp = 0 if S < module.l else (S - module.l) // module.d + 1
        _lib.check(L.nsa_cmp_pool_append(ctypes.byref(desc), ctypes.byref(kd), 0, n_cmp, _stream(dev)), "nsa_cmp_pool_append")
        kv.n_cmp = n_cmp
        ctx.module, ctx.S, ctx.n_cmp, ctx.B = module, S, n_cmp, K_raw.shape[0]
        return kv._K_cmp[:, :, :n_cmp], kv._V_cmp[:, :, :n_cmp]

    @staticmethod
    def backward(ctx, dKc, dVc):
        m, S, n_cmp, B = ctx.module, ctx.S, ctx.n_cmp, ctx.B
        ref = dKc if dKc is not None else dVc
        L, dev = _lib.lib(), ref.device
        desc, _ = m._layer_desc()
        G = m.n_kv_groups
        dKc = torch.zeros((B, G, n_cmp, m.d_k), dtype=ref.dtype, device=dev) if dKc is None else dKc.contiguous()
        dVc = torch.zeros((B, G, n_cmp, m.d_v), dtype=ref.dtype, device=dev) if dVc is None else dVc.contiguous()
        dKr = torch.empty((B, G, S, m.d_k), dtype=ref.dtype, device=dev)
        dVr = torch.empty((B, G, S, m.d_v), dtype=ref.dtype, device=dev)
        _lib.check(L.nsa_cmp_pool_bwd(ctypes.byref(desc), B, S, n_cmp, dKc.data_ptr() if n_cmp else None, dVc.data_ptr() if n_cmp else None,
                                      dKr.data_ptr(), dVr.data_ptr(), _stream(dev)), "nsa_cmp_pool_bwd")
        return dKr, dVr, None, None, None


class _GateCombineFn(torch.autograd.Function):
    """gate MLP + 3-branch combine: native forward; backward = one native pass (dO_i = gate_i dO, dgate = sum O_i dO) plus the
    gradient of the tiny gate MLP taken with torch on the recomputed [R,Dk] -> [R,3] network"""

    @staticmethod
    def forward(ctx, Q, O_cmp, O_sel, O_win, w1, b1, w2, b2, module):
        L, dev = _lib.lib(), Q.device
        desc, _ = module._layer_desc()
        Qc, Oc, Os, Ow = Q.contiguous(), O_cmp.contiguous(), O_sel.contiguous(), O_win.contiguous()
        B, S, G = Qc.shape[:3]
        O = torch.empty_like(Os)
        gates = torch.empty((B, S, G, 3), dtype=torch.float32, device=dev)
        _lib.check(L.nsa_gate_combine(ctypes.byref(desc), Qc.data_ptr(), Oc.data_ptr(), Os.data_ptr(), Ow.data_ptr(), O.data_ptr(),
                                      gates.data_ptr(), B * S * G, _stream(dev)), "nsa_gate_combine")
        ctx.save_for_backward(Qc, Oc, Os, Ow, gates, w1, b1, w2, b2)
        ctx.module = module
        _set_plain(module, "_last_gates", gates)
        return O

    @staticmethod
    def backward(ctx, dO):
        Qc, Oc, Os, Ow, gates, w1, b1, w2, b2 = ctx.saved_tensors
        m = ctx.module
        L, dev = _lib.lib(), dO.device
        desc, _ = m._layer_desc()
        dO = dO.contiguous()
        dOc, dOs, dOw = torch.empty_like(Oc), torch.empty_like(Os), torch.empty_like(Ow)
        dg = torch.empty_like(gates)
        R = gates.numel() // 3
        _lib.check(L.nsa_gate_combine_bwd(ctypes.byref(desc), dO.data_ptr(), Oc.data_ptr(), Os.data_ptr(), Ow.data_ptr(), gates.data_ptr(),
                                          dOc.data_ptr(), dOs.data_ptr(), dOw.data_ptr(), dg.data_ptr(), R, _stream(dev)), "nsa_gate_combine_bwd")
        with torch.enable_grad():
            ins = [t.detach().requires_grad_(True) for t in (Qc, w1, b1, w2, b2)]
            g = _gate_probs_fn(ins[0].mean(dim=3), ins[1], ins[2], ins[3], ins[4], m.gate_temp)
            grads = torch.autograd.grad(g, ins, dg.to(g.dtype), allow_unused=True)
        return (grads[0], dOc, dOs, dOw, grads[1], grads[2], grads[3], grads[4], None)


class NSAAttention(nn.Module):
    def __init__(self, dim: int, n_heads: int, n_kv_groups: int, d_k: int, d_v: int, l: int = 32, d: int = 16, l_sel: int = 64,
                 n_sel: int = 16, w: int = 512, phi: str = "avg", gate_hidden: Optional[int] = None, gate_temp: float = 1.0,
                 rope_impl: str = "llama", use_flash: bool = True, use_triton_sel: bool = False, *,
                 selector: Optional[str] = None, query_chunk: int = 2048) -> None:
        super().__init__()
        assert n_heads % n_kv_groups == 0, "heads must be divisible by kv groups"
        if l % d != 0 or l_sel % d != 0:
            raise ValueError("M0 requires d|l and d|l_sel; set valid block sizes/stride.")
        if (phi or "avg").lower() != "avg":
            raise ValueError("only phi='avg' is provided (the reference's learnable phi is disabled by a bug, SURVEY 4)")
        self.dim, self.n_heads, self.n_kv_groups, self.h_per_group = dim, n_heads, n_kv_groups, n_heads // n_kv_groups
        self.d_k, self.d_v, self.l, self.d, self.l_sel, self.n_sel, self.w = d_k, d_v, l, d, l_sel, n_sel, w
        self.gate_temp = gate_temp
        self.rope_scale = float(os.getenv("NSA_ROPE_SCALE", "1.0") or 1.0)
        if selector is None:
            selector = "batched" if os.getenv("NSA_PREFILL_BATCHED", "0").lower() in ("1", "true", "yes") else "sequential"
        assert selector in ("sequential", "batched")
        self.selector = selector
        self.query_chunk = query_chunk  # reserved: the fused scorer never materialises p_cmp, so prefill is not query-chunked (p_grp is 512 MiB at 64k)
        self.W_Q = nn.Linear(dim, n_heads * d_k, bias=False)
        self.W_K_sel = nn.Linear(dim, n_kv_groups * d_k, bias=False)
        self.W_V_sel = nn.Linear(dim, n_kv_groups * d_v, bias=False)
        self.W_K_win = nn.Linear(dim, n_kv_groups * d_k, bias=False)
        self.W_V_win = nn.Linear(dim, n_kv_groups * d_v, bias=False)
        self.W_K_cmp = nn.Linear(dim, n_kv_groups * d_k, bias=False)
        self.W_V_cmp = nn.Linear(dim, n_kv_groups * d_v, bias=False)
        self.out = nn.Linear(n_heads * d_v, dim, bias=False)
        self.gate = GateMLP(d_k, gate_hidden)
        self._last_gates: Optional[torch.Tensor] = None
        self._last_ranges: Optional[torch.Tensor] = None
        # Routing contract of the reference (nsa_attention.py:300-332 flag cache read at construction, :704-708 / :1205-1211):
        # NSA_FORCE_PARITY=1 disables every fast selection route, which leaves the reference on its gather executors -- batched
        # prefill on grouped_selection_attention (first gathered key, attention_kernels.py:181-226), decode and sequential prefill
        # on _sdpa_over_ranges (head i sees the first i+1 gathered keys, :1779-1855).  Here that flag routes the selected branch
        # to the two parity-mode executors that reproduce those outputs; everything else stays on the native kernels.
        self._force_parity = os.getenv("NSA_FORCE_PARITY", "0").lower() in ("1", "true", "yes")
        # A failed native executor is counted and the reference falls back to its next executor and returns normally
        # (cuda_sel_kernel/__init__.py:60-68, nsa_attention.py:764-782; pinned by nsa/tests/test_cuda_loader_fallback.py:6-38).  Here the
        # "next executor" of the one-call native layer is the same layer composed from the separate native entry points (still the HIP
        # kernels; there is no CPU or torch-SDPA executor to fall to).  NSA_HIP_STRICT=1 (read at construction) raises instead.
        self._strict = os.getenv("NSA_HIP_STRICT", "0").lower() in ("1", "true", "yes")
        self._fallback_counters = {k: 0 for k in ("selection_triton_fails", "selection_cuda_fails", "selection_hip_fails",
                                                  "selection_pack_fails", "selection_mask_fails", "compressed_fa2_fails",
                                                  "sliding_fa2_fails", "total_fallbacks")}

    # ---- cache -------------------------------------------------------------------------------
    def new_kv(self, B: int, S_max: int, device, dtype) -> NSA_KV:
        return NSA_KV(B, self.n_kv_groups, self.d_k, self.d_v, S_max, self.l, self.d, self.l_sel, self.n_sel, self.w, device, dtype)

    # ---- monitors (reference :407-445); computed on demand so the forward pass never syncs -----
    def get_gate_stats(self) -> Optional[dict]:
        g = self._last_gates
        if g is None:
            return None
        g = g.detach().float().reshape(-1, 3)
        ent = -(g * (g + 1e-8).log()).sum(-1)
        return {"entropy_mean": float(ent.mean()), "entropy_min": float(ent.min()), "max_gate_mean": float(g.max(-1).values.mean()),
                "max_gate_max": float(g.max()), "branch_shares": [float(x) for x in g.mean(0)]}

    def get_fallback_counters(self) -> dict:
        return self._fallback_counters.copy()

    def reset_fallback_counters(self) -> dict:
        prev = self._fallback_counters.copy()
        for k in self._fallback_counters:
            self._fallback_counters[k] = 0
        return prev

    def get_selection_stats(self) -> Optional[dict]:
        r = self._last_ranges
        if r is None:
            return None
        L = (r[..., 1] - r[..., 0]).clamp_min(0).sum(-1).to(torch.int64)
        k_max = int(L.max()) if L.numel() else 0
        return {"k_mean": float(L.float().mean()) if L.numel() else 0.0, "k_max": k_max, "rows": int(L.numel()),
                "pct_at_max": float((L == k_max).float().mean()) if k_max > 0 else 0.0, "l_sel": self.l_sel, "n_sel": self.n_sel}

    # ---- native layer path (inference): descriptors for the C ABI ------------------------------
    _QKV = ("W_Q", "W_K_sel", "W_V_sel", "W_K_win", "W_V_win", "W_K_cmp", "W_V_cmp")

    def _native_ok(self, x: torch.Tensor) -> bool:
        """the fused kernels cover inference (no autograd graph) on the GPU; training keeps the differentiable eager ops"""
        if self._force_parity:
            return False  # the one-call layer fuses the masked-semantics executor; parity mode composes the layer from its stages
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return False
        return x.is_cuda and x.dtype in _DT and self.W_Q.weight.dtype == x.dtype and self.gate.fc1.out_features <= 64

    def _train_native_ok(self, x: torch.Tensor) -> bool:
        """training (autograd) on the GPU: the layer kernels run as differentiable ops (native backward kernels)"""
        return (torch.is_grad_enabled() and x.is_cuda and x.dtype in _DT and self.W_Q.weight.dtype == x.dtype and not self._force_parity
                and self.gate.fc1.out_features <= 64 and os.getenv("NSA_HIP_EAGER_TRAIN", "0") != "1")

    def _layer_desc(self):
        """(nsa_layer_desc, fused W_qkv) -- rebuilt when a parameter was modified or moved (tensor version counters)"""
        # (the parameters through the modules' own dictionaries: nn.Module.__getattr__ on thirteen dotted paths was the largest item of a
        # decode step's host time)
        mods = self._modules
        gate = mods["gate"]
        fc1 = gate._modules["fc1"]._parameters
        ps = [mods[n]._parameters["weight"] for n in self._QKV] + [mods["out"]._parameters["weight"], fc1["weight"], fc1["bias"],
                                                                    *gate.fc2_params()]
        key = tuple([(p.data_ptr(), p._version) for p in ps])
        if getattr(self, "_desc_key", None) != key or self._desc_keep[0].dtype != ps[0].dtype:
            W_qkv = torch.cat([p.detach() for p in ps[:7]], dim=0).contiguous()
            keep = [W_qkv] + [p.detach().contiguous() for p in ps[7:]]
            d = _lib.NsaLayerDesc()
            d.dim, d.G, d.h, d.Dk, d.Dv = self.dim, self.n_kv_groups, self.h_per_group, self.d_k, self.d_v
            d.l, d.d, d.l_sel, d.n_sel, d.w = self.l, self.d, self.l_sel, self.n_sel, self.w
            d.gate_hidden, d.dtype = self.gate.fc1.out_features, _DT[W_qkv.dtype]
            d.rope_base, d.rope_scale, d.gate_tau = 10000.0, self.rope_scale, float(self.gate_temp)
            d.W_qkv, d.W_out = keep[0].data_ptr(), keep[1].data_ptr()
            d.gate_w1, d.gate_b1, d.gate_w2, d.gate_b2 = (k.data_ptr() for k in keep[2:])
            _set_plain(self, "_desc_key", key)
            _set_plain(self, "_desc", d)
            _set_plain(self, "_desc_keep", keep)
        return self._desc, self._desc_keep[0]

    @staticmethod
    def _kv_desc(kv: NSA_KV):
        d = getattr(kv, "_desc", None)
        if d is None:
            d = _lib.NsaKvDesc()
            d.K_sel, d.V_sel, d.K_win, d.V_win = kv._K_sel.data_ptr(), kv._V_sel.data_ptr(), kv._K_win.data_ptr(), kv._V_win.data_ptr()
            d.K_raw, d.V_raw, d.K_cmp, d.V_cmp = kv._K_raw.data_ptr(), kv._V_raw.data_ptr(), kv._K_cmp.data_ptr(), kv._V_cmp.data_ptr()
            d.B, d.S_max, d.n_cmp_max = kv.B, kv._K_sel.shape[2], kv._K_cmp.shape[2]
            kv._desc = d
        return d

    # ---- helpers ---------------------------------------------------------------------------
    def _project(self, x: torch.Tensor, pos: torch.Tensor):
        B, S, _ = x.shape
        G, h = self.n_kv_groups, self.h_per_group
        # the reference rotates Q over the flattened [n_heads*d_k] axis (nsa_attention.py:552-560,1002-1009) and K per group
        Q = apply_rope(self.W_Q(x), pos, scale=self.rope_scale).view(B, S, G, h, self.d_k)
        kv = lambda W: W(x).view(B, S, G, -1).permute(0, 2, 1, 3)  # noqa: E731  -> [B,G,S,D]
        K_sel = apply_rope(kv(self.W_K_sel), pos, scale=self.rope_scale)
        K_win = apply_rope(kv(self.W_K_win), pos, scale=self.rope_scale)
        return Q, K_sel, kv(self.W_V_sel), K_win, kv(self.W_V_win), kv(self.W_K_cmp), kv(self.W_V_cmp)

    def _combine(self, Q, O_cmp, O_sel, O_win):
        B, S = Q.shape[:2]
        gates = self.gate(Q.mean(dim=3), tau=self.gate_temp)  # [B,S,G,3]
        _set_plain(self, "_last_gates", gates)
        O = gates[..., 0:1].unsqueeze(3) * O_cmp + gates[..., 1:2].unsqueeze(3) * O_sel + gates[..., 2:3].unsqueeze(3) * O_win
        return self.out(O.reshape(B, S, self.n_heads * self.d_v))

    # ---- forward ---------------------------------------------------------------------------
    def _check_kv(self, x: torch.Tensor, kv: NSA_KV) -> None:
        """The C ABI sees raw pointers: the kernels walk kv.B sequences of the cache's dtype, so a cache built for another batch
        size, dtype or device would be read and written out of bounds.  Refuse it here."""
        buf = kv._K_sel
        if x.shape[0] != kv.B or buf.dtype != x.dtype or buf.device != x.device:
            raise RuntimeError(f"NSA_KV does not match the input: cache B={kv.B} {buf.dtype} on {buf.device}, "
                               f"x B={x.shape[0]} {x.dtype} on {x.device}")
        if (kv.G, kv.d_k, kv.d_v) != (self.n_kv_groups, self.d_k, self.d_v):
            raise RuntimeError(f"NSA_KV geometry (G={kv.G}, d_k={kv.d_k}, d_v={kv.d_v}) does not match the module "
                               f"(G={self.n_kv_groups}, d_k={self.d_k}, d_v={self.d_v})")

    def forward(self, x: torch.Tensor, kv: NSA_KV, *, prefill: bool):
        assert x.dim() == 3, "x must be [B,S,dim]"
        if prefill:
            assert x.shape[1] > 0, f"Prefill mode requires S > 0, got S={x.shape[1]}"
        else:
            assert x.shape[1] == 1, f"Decode mode requires S=1 (single token), got S={x.shape[1]}."
        self._check_kv(x, kv)
        # a strided [B,S,dim] view (e.g. x[:, :S] of a longer buffer) is copied first: the projection would otherwise run as a strided-batched
        # GEMM, and PyTorch 2.10 + rocm7.0 returns wrong values from that route at some shapes (3 x 2100 x 768 -> 1536, bf16: errors of
        # several units, a memory fault when the result is consumed; tools/dbg_gemm.py) -- the contiguous 2-D GEMM is correct
        if not x.is_contiguous():
            x = x.contiguous()
        one_call = self._native_ok(x)
        try:
            return self._prefill(x, kv) if prefill else self._decode(x, kv)
        except RuntimeError as e:
            # the reference's router counts a failed native executor (nsa_attention.py:764-782) ...
            self._fallback_counters["selection_hip_fails"] += 1
            self._fallback_counters["total_fallbacks"] += 1
            self._last_error = str(e)
            if self._strict or not one_call or "HIP error" in str(e):
                # ... the per-stage composition is the last executor there is: no CPU / eager-SDPA route exists by design; and a HIP
                # runtime / device error (a launch failure, a fault) is not a status return to route around: raised as it is
                raise
            # ... and falls back to its next executor and returns normally: the layer composed from the separate native calls
            warnings.warn(f"nsa_vibe_amd: the one-call native layer failed ({e}); falling back to the per-stage native route", RuntimeWarning)
            try:
                return self._prefill(x, kv, one_call=False) if prefill else self._decode(x, kv, one_call=False)
            except Exception as e2:
                raise e2 from e  # (both tracebacks: the per-stage route's failure, caused by the one-call route's)

    def _prefill(self, x: torch.Tensor, kv: NSA_KV, one_call: bool = True):
        B, S, _ = x.shape
        assert kv.t == 0, "prefill expects an empty cache"
        native = one_call and self._native_ok(x)
        if native:
            return self._prefill_native(x, kv)
        elif one_call and self._train_native_ok(x):
            return self._prefill_train_native(x, kv)
        else:
            pos = torch.arange(S, device=x.device)
            Q, K_sel, V_sel, K_win, V_win, K_raw, V_raw = self._project(x, pos)
            kv.write_tokens(K_sel, V_sel, K_win, V_win, K_raw, V_raw)
            K_cmp, V_cmp = avg_pool_phi(apply_rope(K_raw, pos), V_raw, self.l, self.d)
            kv.write_compressed(K_cmp, V_cmp, at=0)
        meta = kv.ensure_meta(S)
        scale = 1.0 / math.sqrt(self.d_k)
        Qc = Q.contiguous()
        # ---- selected branch (HIP): scores -> ranges -> attention
        ranges = _scores_and_ranges(Qc, kv.K_cmp, meta, self.n_sel, self.selector, S, scale)
        if self._force_parity:  # reference gather routes: batched -> first gathered key, sequential -> _sdpa_over_ranges
            parity = selection_attention_first_key_parity if self.selector == "batched" else selection_attention_head_causal_parity
            O_sel = parity(Qc, kv.K_sel, kv.V_sel, ranges)
        else:
            O_sel = selection_attention_hip(Qc, kv.K_sel, kv.V_sel, ranges, scale=scale)
        _set_plain(self, "_last_ranges", ranges)
        # ---- compressed + sliding branches (HIP band kernel)
        O_cmp = batched_causal_attention_compressed(Qc, kv.K_cmp, kv.V_cmp, self.l, self.d, scale=scale)
        O_win = sliding_window_attention(Qc, kv._K_win[:, :, :S], kv._V_win[:, :, :S], self.w, scale=scale)
        return self._combine(Q, O_cmp, O_sel, O_win), kv

    def _prefill_native(self, x: torch.Tensor, kv: NSA_KV, mix_only: bool = False):
        """inference prefill: fused projection GEMM -> ONE native call for everything up to the output projection
        (nsa_layer_prefill: RoPE + cache append, pooling, scores, top-n + selection attention, sliding / compressed branches,
        gates + combine) -> output GEMM"""
        from .selection_scorer import batched_ranges_width

        B, S, _ = x.shape
        kv.ensure_capacity(S)
        L, dev = _lib.lib(), x.device
        desc, W_qkv = self._layer_desc()
        kd = self._kv_desc(kv)
        meta = kv.ensure_meta(S)
        proj = F.linear(x, W_qkv)
        G = self.n_kv_groups
        if self.selector == "batched":
            mode, W = _lib.NSA_SEL_BATCHED, batched_ranges_width(meta.S_sel, self.l_sel, self.n_sel, S, True, 2)
        else:
            mode, W = _lib.NSA_SEL_SEQUENTIAL, self.n_sel
        ranges = torch.empty((B, S, G, W, 2), dtype=torch.int32, device=dev)
        gates = torch.empty((B, S, G, 3), dtype=torch.float32, device=dev)
        O = torch.empty((B, S, self.n_heads * self.d_v), dtype=x.dtype, device=dev)
        ws = workspace(dev, L.nsa_layer_prefill_workspace(ctypes.byref(desc), B, S, int(meta.S_sel)) + 256, "layer_prefill")
        wptr = (ws.data_ptr() + 255) & ~255
        cptr, crows, cvals = meta.device_csc(dev)
        rc = L.nsa_layer_prefill(ctypes.byref(desc), ctypes.byref(kd), proj.data_ptr(), S, mode, cptr.data_ptr(), crows.data_ptr(),
                                 cvals.data_ptr(), int(meta.S_sel), ranges.data_ptr(), W, O.data_ptr(), gates.data_ptr(), wptr,
                                 ws.numel() - (wptr - ws.data_ptr()), _stream(dev))
        _lib.check(rc, "nsa_layer_prefill")
        kv.t = S
        kv.n_cmp = 0 if S < self.l else (S - self.l) // self.d + 1
        _set_plain(self, "_last_ranges", ranges)
        _set_plain(self, "_last_gates", gates)
        return O if mix_only else self.out(O), kv

    def _prefill_train_native(self, x: torch.Tensor, kv: NSA_KV):
        """training forward: one fused projection GEMM, then every stage is a differentiable native op (the attention branches
        with their backward kernels, RoPE/append, pooling and gate/combine with theirs)"""
        B, S, _ = x.shape
        kv.ensure_capacity(S)
        W_qkv = torch.cat([getattr(self, n).weight for n in self._QKV], dim=0)
        proj = F.linear(x, W_qkv)
        Q, K_sel, V_sel, K_win, V_win, K_raw, V_raw = _RopeAppendFn.apply(proj, self, kv, S)
        kv.t = S
        K_cmp, V_cmp = _CmpPoolFn.apply(K_raw, V_raw, self, kv, S)
        meta = kv.ensure_meta(S)
        scale = 1.0 / math.sqrt(self.d_k)
        with torch.no_grad():  # the selection itself is not differentiable (top-n indices)
            ranges = _scores_and_ranges(Q.detach(), kv.K_cmp, meta, self.n_sel, self.selector, S, scale)
        _set_plain(self, "_last_ranges", ranges)
        O_sel = selection_attention_hip(Q, K_sel, V_sel, ranges, scale=scale)
        O_cmp = batched_causal_attention_compressed(Q, K_cmp, V_cmp, self.l, self.d, scale=scale)
        O_win = sliding_window_attention(Q, K_win, V_win, self.w, scale=scale)
        O = _GateCombineFn.apply(Q, O_cmp, O_sel, O_win, self.gate.fc1.weight, self.gate.fc1.bias, *self.gate.fc2_params(),
                                 self)
        return self.out(O.reshape(B, S, self.n_heads * self.d_v)), kv

    def _decode_native(self, x: torch.Tensor, kv: NSA_KV):
        """the whole decode step in one native call (nsa_layer_decode_step): ~15 kernel launches, no host sync"""
        t, B, dev = kv.t, x.shape[0], x.device
        kv.ensure_capacity(t + 1)
        if kv.meta.S_sel == 0:  # block metadata refresh policy of the reference (:606-632)
            kv.ensure_meta(max(t + 1, self.l_sel))
        elif t + 1 > kv.meta.S_sel * self.l_sel:
            kv.ensure_meta(t + 1)
        S_raw = t + 1
        num_cmp = 0 if S_raw < self.l else (S_raw - self.l) // self.d + 1
        L = _lib.lib()
        desc, _ = self._layer_desc()
        ctx = getattr(kv, "_dec_ctx", None)  # per-cache constants of the native call (descriptors, workspace, monitors)
        if ctx is None or ctx[0] is not desc:
            kd = self._kv_desc(kv)
            ws = workspace(dev, L.nsa_layer_decode_step_workspace(ctypes.byref(desc), B, kd.S_max) + 256, "layer_decode")
            wptr = (ws.data_ptr() + 255) & ~255
            ranges = torch.empty((B, self.n_kv_groups, self.n_sel, 2), dtype=torch.int32, device=dev)
            gates = torch.empty((B, 1, self.n_kv_groups, 3), dtype=torch.float32, device=dev)
            ctx = kv._dec_ctx = (desc, ctypes.byref(desc), ctypes.byref(kd), kd, ws, wptr, ws.numel() - (wptr - ws.data_ptr()), ranges, gates)
        _, desc_ref, kd_ref, _, _, wptr, wsize, ranges, gates = ctx
        cptr, crows, cvals = kv.meta.device_csc(dev)
        xc = x.reshape(B, self.dim)
        if not xc.is_contiguous():
            xc = xc.contiguous()
        y = torch.empty((B, 1, self.dim), dtype=x.dtype, device=dev)
        rc = L.nsa_layer_decode_step(desc_ref, kd_ref, xc.data_ptr(), y.data_ptr(), t, cptr.data_ptr(), crows.data_ptr(), cvals.data_ptr(),
                                     int(kv.meta.S_sel), ranges.data_ptr(), gates.data_ptr(), wptr, wsize, _stream(dev))
        _lib.check(rc, "nsa_layer_decode_step")
        kv.t, kv.n_cmp = S_raw, num_cmp
        kv.append_reads(num_cmp, S_raw)
        _set_plain(self, "_last_ranges", ranges)
        _set_plain(self, "_last_gates", gates)
        return y, kv

    def _decode(self, x: torch.Tensor, kv: NSA_KV, one_call: bool = True):
        if one_call and self._native_ok(x):
            return self._decode_native(x, kv)
        t = kv.t  # position of the new token
        pos = torch.tensor([t], device=x.device)
        Q, K_sel, V_sel, K_win, V_win, K_raw, V_raw = self._project(x, pos)
        kv.write_tokens(K_sel, V_sel, K_win, V_win, K_raw, V_raw)
        S_raw = kv.t
        if S_raw >= self.l and (S_raw - self.l) % self.d == 0:  # emit one compressed token (reference :588-604)
            p_last = torch.arange(S_raw - self.l, S_raw, device=x.device)
            K_new = apply_rope(kv._K_raw[:, :, S_raw - self.l: S_raw], p_last).mean(dim=2, keepdim=True)
            V_new = kv._V_raw[:, :, S_raw - self.l: S_raw].mean(dim=2, keepdim=True)
            kv.write_compressed(K_new.to(kv._K_cmp.dtype), V_new.to(kv._V_cmp.dtype))
        # block metadata refresh policy of the reference (:606-632): rebuild only when t leaves the covered blocks
        if kv.meta.S_sel == 0:
            kv.ensure_meta(max(t + 1, self.l_sel))
        elif t + 1 > kv.meta.S_sel * self.l_sel:
            kv.ensure_meta(t + 1)
        num_cmp = 0 if S_raw < self.l else (S_raw - self.l) // self.d + 1
        kv.append_reads(num_cmp, S_raw)
        scale = 1.0 / math.sqrt(self.d_k)
        Qc = Q.contiguous()
        O_sel, ranges = selection_decode_step(Qc, kv.K_cmp, kv.K_sel, kv.V_sel, kv.meta, self.n_sel, t, scale=scale)
        if self._force_parity:  # the reference's decode gather route (_sdpa_over_ranges, nsa_attention.py:830)
            O_sel = selection_attention_head_causal_parity(Qc, kv.K_sel, kv.V_sel, ranges.unsqueeze(1))
        _set_plain(self, "_last_ranges", ranges)
        # the query sits at position t: window = the last w cached tokens, compressed = every token emitted so far
        O_win = sliding_window_attention(Qc, kv._K_win[:, :, :S_raw], kv._V_win[:, :, :S_raw], self.w, t0=t, scale=scale)
        O_cmp = batched_causal_attention_compressed(Qc, kv.K_cmp, kv.V_cmp, self.l, self.d, t0=t, scale=scale)
        return self._combine(Q, O_cmp, O_sel, O_win), kv
