"""ctypes/numpy front-end of the CPU oracle (oracle/nsa_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of nsa_oracle.c.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (nsa_vibe_amd/) never does.

Function names follow the reference (nsa/core/block_index.py, selection_scorer.py,
attention_kernels.py); arrays are numpy, fp32 / int32, C-contiguous.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libnsa_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (Makefile in this directory)."""
    src = os.path.join(_HERE, "nsa_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libnsa_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


@dataclass
class OracleBlockMeta:
    """Same fields as the reference BlockMeta (nsa/core/block_index.py:7-22), numpy arrays."""

    l: int
    d: int
    l_sel: int
    n_sel: int
    w: int
    cmp_starts: np.ndarray
    sel_starts: np.ndarray
    M_csl_indptr: np.ndarray
    M_csl_indices: np.ndarray
    M_csl_values: np.ndarray
    M_csl_coo_indices: np.ndarray
    M_csl_coo_values: np.ndarray


def build_block_meta(seq_len: int, l: int, d: int, l_sel: int, n_sel: int, w: int) -> OracleBlockMeta:
    """nsa/core/block_index.py:74-99."""
    if l % d != 0 or l_sel % d != 0:
        raise ValueError("Require d|l and d|l_sel in M0")
    if d <= 0 or l <= 0 or l_sel <= 0:
        raise ValueError("Block parameters must be positive")
    L = lib()
    s_cmp, s_sel = C.c_int(), C.c_int()
    L.nsa_oracle_block_counts(seq_len, l, d, l_sel, C.byref(s_cmp), C.byref(s_sel))
    S_cmp, S_sel = s_cmp.value, s_sel.value
    indptr = np.zeros(S_cmp + 1, np.int32)
    cap = max(1, S_cmp * (l // l_sel + 2))
    indices = np.zeros(cap, np.int32)
    values = np.zeros(cap, np.float32)
    nnz = L.nsa_oracle_build_csr(seq_len, l, d, l_sel, _p(indptr), _p(indices), _p(values))
    indices, values = indices[:nnz].copy(), values[:nnz].copy()
    rows = np.repeat(np.arange(S_cmp, dtype=np.int32), np.diff(indptr))
    return OracleBlockMeta(
        l, d, l_sel, n_sel, w,
        (np.arange(S_cmp, dtype=np.int32) * d).astype(np.int32),
        (np.arange(S_sel, dtype=np.int32) * l_sel).astype(np.int32),
        indptr, indices, values,
        np.stack([rows, indices]).astype(np.int32), values.copy(),
    )


def compute_pcmp_all(Q, K_cmp, scale: float) -> np.ndarray:
    """nsa/core/selection_scorer.py:42-61.  Q [B,S,G,h,Dk], K_cmp [B,G,S_cmp,Dk] -> [B,S,G,h,S_cmp]."""
    Q, K_cmp = _f32(Q), _f32(K_cmp)
    B, S, G, h, Dk = Q.shape
    S_cmp = K_cmp.shape[2]
    P = np.zeros((B, S, G, h, S_cmp), np.float32)
    lib().nsa_oracle_pcmp_all(_p(Q), _p(K_cmp), _p(P), B, S, G, h, Dk, S_cmp, C.c_float(scale))
    return P


def map_pcmp_to_pslc_and_pgrp(p_cmp_all, meta: OracleBlockMeta):
    """selection_scorer.py:89-116 then .sum(dim=3) (nsa_attention.py:1091).

    p_cmp_all [..., h, S_cmp_cur] -> (p_slc [..., h, S_sel], p_grp [..., S_sel])."""
    p = _f32(p_cmp_all)
    lead = p.shape[:-2]
    h, S_cmp_cur = p.shape[-2:]
    S_sel = int(meta.sel_starts.size)
    R = int(np.prod(lead)) if lead else 1
    p_slc = np.zeros((R, h, S_sel), np.float32)
    p_grp = np.zeros((R, S_sel), np.float32)
    if S_cmp_cur > 0 and S_sel > 0:
        lib().nsa_oracle_map_pcmp_to_pgrp(
            _p(p), C.c_long(R), h, S_cmp_cur, _p(meta.M_csl_indptr), _p(meta.M_csl_indices),
            _p(meta.M_csl_values), int(meta.cmp_starts.size), S_sel, _p(p_slc), _p(p_grp))
    return p_slc.reshape(*lead, h, S_sel), p_grp.reshape(*lead, S_sel)


def select_topn_ranges_rows(p_grp_rows, t_tokens, meta: OracleBlockMeta, n_top: int,
                            force_init: bool = True, force_local: int = 2) -> np.ndarray:
    """Sequential-mode selector on a list of rows with per-row token position."""
    p = _f32(p_grp_rows)
    R, S_sel = p.shape
    t = _i32(t_tokens)
    out = np.zeros((R, n_top, 2), np.int32)
    lib().nsa_oracle_select_topn_seq(_p(p), C.c_long(R), S_sel, int(meta.l_sel), n_top, _p(t),
                                     int(bool(force_init)), int(force_local), _p(out))
    return out


def select_topn_ranges(p_grp, meta: OracleBlockMeta, n_top: int, t_token: int,
                       force_init: bool = True, force_local: int = 2) -> np.ndarray:
    """nsa/core/selection_scorer.py:124-249.  p_grp [B,G,S_sel] -> int32 [B,G,n_top,2]."""
    p = _f32(p_grp)
    B, G, S_sel = p.shape
    t = np.full(B * G, t_token, np.int32)
    return select_topn_ranges_rows(p.reshape(B * G, S_sel), t, meta, n_top, force_init,
                                   force_local).reshape(B, G, n_top, 2)


def select_topn_ranges_batched(p_grp_all, meta: OracleBlockMeta, n_top: int, S: int,
                               force_init: bool = True, force_local: int = 2) -> np.ndarray:
    """nsa/core/selection_scorer.py:255-362 (+ v2 converter :434-605).  [B,S,G,S_sel] -> [B,S,G,K,2]."""
    p = _f32(p_grp_all)
    B, S_q, G, S_sel = p.shape
    assert S_q == S
    K = C.c_int()
    L = lib()
    L.nsa_oracle_select_topn_batched(None, B, S, G, S_sel, int(meta.l_sel), n_top,
                                     int(bool(force_init)), int(force_local), None, C.byref(K))
    out = np.zeros((B, S, G, K.value, 2), np.int32)
    L.nsa_oracle_select_topn_batched(_p(p), B, S, G, S_sel, int(meta.l_sel), n_top,
                                     int(bool(force_init)), int(force_local), _p(out), C.byref(K))
    return out


def convert_indices_to_ranges_batched_v2(indices, meta: OracleBlockMeta, S: int) -> np.ndarray:
    """nsa/core/selection_scorer.py:434-605.  indices [B,S,G,K] -> int32 [B,S,G,K,2]."""
    x = _i32(indices)
    B, S_q, G, K = x.shape
    out = np.zeros((B, S_q, G, K, 2), np.int32)
    if K == 0:
        return out
    t_rows = _i32(np.broadcast_to(np.arange(S_q, dtype=np.int32)[None, :, None], (B, S_q, G)))
    lib().nsa_oracle_indices_to_ranges_v2(_p(x), C.c_long(B * S_q * G), K, int(meta.sel_starts.size),
                                          int(meta.l_sel), _p(t_rows), _p(out))
    return out


def sel_attention_masked(Q, K, V, ranges, scale: float | None = None, return_lse: bool = False):
    """nsa/core/attention_kernels.py:705-772.  fp32 math; returns O [B,S,G,h,Dv] (fp32)."""
    Q, K, V = _f32(Q), _f32(K), _f32(V)
    rg = _i32(ranges)
    B, S, G, h, Dk = Q.shape
    S_kv, Dv = K.shape[2], V.shape[3]
    n = rg.shape[3]
    if scale is None:
        scale = 1.0 / float(np.sqrt(Dk))
    O = np.zeros((B, S, G, h, Dv), np.float32)
    lse = np.zeros((B, S, G, h), np.float32)
    if S_kv > 0 and n > 0:
        lib().nsa_oracle_sel_attention_masked(_p(Q), _p(K), _p(V), _p(rg), _p(O), _p(lse), B, S, G,
                                              h, Dk, Dv, S_kv, n, C.c_float(scale))
    else:
        lse[:] = -np.inf
    return (O, lse) if return_lse else O


def sel_attention_masked_bwd(Q, K, V, ranges, dO, scale: float | None = None):
    """Gradient of sel_attention_masked w.r.t. Q, K, V (fp64 inner math)."""
    Q, K, V, dO = _f32(Q), _f32(K), _f32(V), _f32(dO)
    rg = _i32(ranges)
    B, S, G, h, Dk = Q.shape
    S_kv, Dv = K.shape[2], V.shape[3]
    n = rg.shape[3]
    if scale is None:
        scale = 1.0 / float(np.sqrt(Dk))
    dQ, dK, dV = np.zeros_like(Q), np.zeros_like(K), np.zeros_like(V)
    lib().nsa_oracle_sel_attention_masked_bwd(_p(Q), _p(K), _p(V), _p(rg), _p(dO), _p(dQ), _p(dK),
                                              _p(dV), B, S, G, h, Dk, Dv, S_kv, n, C.c_float(scale))
    return dQ, dK, dV


def band_ranges(S: int, S_kv: int, t0: int = 0, a: int = 0, dd: int = 1, c: int = 0, w: int = 2 ** 30) -> np.ndarray:
    """Key interval [lo, hi) of query row t (position t0 + t) for the sliding / compressed branches, [S,2] int32:
    hi = 0 if t0+t+1 < a else min(S_kv, (t0+t+1-a)//dd + c); lo = max(0, hi - w).
    Sliding window (nsa/core/attention_kernels.py:159-161: allowed = col <= row & col >= row-(w-1)): a=0, dd=1, c=0.
    Compressed (attention_kernels.py:118-121: num_cmp = 0 if t+1 < l else (t+1-l)//d + 1, clamped to S_cmp): a=l, dd=d, c=1."""
    e = np.arange(S, dtype=np.int64) + (t0 + 1 - a)
    hi = np.where(e < 0, 0, np.maximum(e, 0) // dd + c)
    hi = np.minimum(hi, S_kv)
    lo = np.maximum(hi - w, 0)
    return np.stack((lo, hi), axis=-1).astype(np.int32)


def band_attention(Q, K, V, *, t0: int = 0, a: int = 0, dd: int = 1, c: int = 0, w: int = 2 ** 30, scale=None,
                   return_lse: bool = False):
    """Softmax attention of row t over its key interval (band_ranges); rows with an empty interval give zeros.
    Restated through sel_attention_masked with one range per row (the masked SDPA of attention_kernels.py:163-177
    is the same math as :705-772 with this mask)."""
    Q = _f32(Q)
    B, S, G = Q.shape[:3]
    rg = np.broadcast_to(band_ranges(S, np.asarray(K).shape[2], t0, a, dd, c, w).reshape(1, S, 1, 1, 2), (B, S, G, 1, 2))
    return sel_attention_masked(Q, K, V, np.ascontiguousarray(rg), scale, return_lse)


def band_attention_bwd(Q, K, V, dO, *, t0: int = 0, a: int = 0, dd: int = 1, c: int = 0, w: int = 2 ** 30, scale=None):
    Q = _f32(Q)
    B, S, G = Q.shape[:3]
    rg = np.broadcast_to(band_ranges(S, np.asarray(K).shape[2], t0, a, dd, c, w).reshape(1, S, 1, 1, 2), (B, S, G, 1, 2))
    return sel_attention_masked_bwd(Q, K, V, np.ascontiguousarray(rg), dO, scale)


def sliding_window_attention(Q, K, V, w: int, *, t0: int = 0, scale=None):
    """nsa/core/attention_kernels.py:146-178 (w <= 0 or no keys -> zeros, :153-154)."""
    return band_attention(Q, K, V, t0=t0, a=0, dd=1, c=0, w=max(int(w), 0), scale=scale)


def batched_causal_attention_compressed(Q, K_cmp, V_cmp, l: int, d: int, *, t0: int = 0, scale=None):
    """Compressed branch with the mask of attention_kernels.py:118-123 and a true softmax over the allowed tokens
    (the reference's per-token call at :139-141 degenerates to key 0; see nsa_vibe_amd/band_attention.py)."""
    return band_attention(Q, K_cmp, V_cmp, t0=t0, a=int(l), dd=int(d), c=1, scale=scale)


def sel_attention_first_key_parity(Q, V, ranges) -> np.ndarray:
    """PARITY MODE of the reference's packed / gather executors (attention_kernels.py:181-226, 273-388): SDPA(is_causal=True) with one
    query sees only the first gathered key, so every head returns V at the start of the first non-empty range (slot order); a row
    without a range stays zero.  Pure numpy (small cases)."""
    V, r = np.asarray(V), np.asarray(ranges)
    B, S, G, h = np.asarray(Q).shape[:4]
    S_kv = V.shape[2]
    O = np.zeros((B, S, G, h, V.shape[3]), dtype=V.dtype)
    for b in range(B):
        for t in range(S):
            for g in range(G):
                for s0, e0 in r[b, t, g]:
                    s0 = min(max(int(s0), 0), S_kv)
                    e0 = min(max(int(e0), s0), S_kv)
                    if e0 > s0:
                        O[b, t, g] = V[b, g, s0]
                        break
    return O


def sel_attention_head_causal_parity(Q, K, V, ranges, scale=None) -> np.ndarray:
    """PARITY MODE of NSAAttention._sdpa_over_ranges (nsa_attention.py:1779-1855): tokens of the union of the clamped ranges in ascending
    order; SDPA(is_causal=True) sees the h heads as query positions, so head i attends the first i+1 gathered tokens; a row without a
    token gives zeros (the reference feeds a single zero key / value).  Pure numpy (small cases)."""
    Q, K, V, r = (np.asarray(x) for x in (Q, K, V, ranges))
    B, S, G, h, Dk = Q.shape
    S_kv = K.shape[2]
    sc = float(scale) if scale else 1.0 / np.sqrt(Dk)
    O = np.zeros((B, S, G, h, V.shape[3]), dtype=np.float32)
    for b in range(B):
        for t in range(S):
            for g in range(G):
                m = np.zeros(S_kv, dtype=bool)
                for s0, e0 in r[b, t, g]:
                    s0, e0 = min(max(int(s0), 0), S_kv), min(max(int(e0), 0), S_kv)
                    if e0 > s0:
                        m[s0:e0] = True
                idx = np.nonzero(m)[0][:h]
                if idx.size == 0:
                    continue
                k, v = K[b, g, idx].astype(np.float32), V[b, g, idx].astype(np.float32)
                for i in range(h):
                    nv = min(i + 1, idx.size)
                    s = (Q[b, t, g, i].astype(np.float32) @ k[:nv].T) * sc
                    p = np.exp(s - s.max())
                    O[b, t, g, i] = (p / p.sum()) @ v[:nv]
    return O.astype(V.dtype) if V.dtype == np.float32 else O


def normalise_ranges(r: np.ndarray) -> list:
    """Drop e<=s entries (SURVEY 7 hard part (c)); returns nested lists of (s,e) per row."""
    r = np.asarray(r)
    flat = r.reshape(-1, r.shape[-2], 2)
    return [[(int(s), int(e)) for s, e in row if e > s] for row in flat]


def num_threads() -> int:
    return int(lib().nsa_oracle_num_threads())


def set_num_threads(n: int) -> None:
    lib().nsa_oracle_set_num_threads(int(n))
