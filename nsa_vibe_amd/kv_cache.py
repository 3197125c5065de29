"""NSA_KV with preallocated storage.

Same field names and layouts as the reference cache (nsa/cache/kv_cache.py:8-26: K_sel/V_sel/K_win/V_win/
K_cmp_raw_seq/V_cmp_raw_seq [B,G,S,D], K_cmp/V_cmp [B,G,S_cmp,D], meta, read counters) -- but the tensors are
views into buffers allocated once for `S_max` tokens, so appending a decode token is a row write instead of the
reference's O(S) `torch.cat` copy per step (kv_cache.py:28-30).  The selection kernels take the element strides
of these views through the C ABI, so nothing is ever re-packed.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from .block_index import BlockMeta, build_block_meta


_META_CACHE: dict = {}


class NSA_KV:
    def __init__(self, B: int, G: int, d_k: int, d_v: int, S_max: int, l: int, d: int, l_sel: int, n_sel: int, w: int,
                 device, dtype, auto_grow: bool = True):
        self.B, self.G, self.d_k, self.d_v, self.S_max = B, G, d_k, d_v, S_max
        self.auto_grow = auto_grow
        self.l, self.d, self.l_sel, self.n_sel, self.w = l, d, l_sel, n_sel, w
        n_cmp_max = 0 if S_max < l else (S_max - l) // d + 1
        mk = lambda n, dim: torch.empty((B, G, max(n, 1), dim), device=device, dtype=dtype)  # noqa: E731
        self._K_sel, self._V_sel = mk(S_max, d_k), mk(S_max, d_v)
        self._K_win, self._V_win = mk(S_max, d_k), mk(S_max, d_v)
        self._K_raw, self._V_raw = mk(S_max, d_k), mk(S_max, d_v)
        self._K_cmp, self._V_cmp = mk(n_cmp_max, d_k), mk(n_cmp_max, d_v)
        self.t = 0  # tokens stored
        self.n_cmp = 0  # compressed tokens emitted
        self.meta: BlockMeta = build_block_meta(0, l, d, l_sel, n_sel, w)
        self.meta_seq_len = 0
        # decode read counters (reference: kv_cache.py:22-26,51-65) kept on the host: no device sync per step
        self.reads_pred: List[int] = []
        self.reads_act_total: List[int] = []
        self.reads_act_sel: List[int] = []
        self.reads_act_cmp: List[int] = []
        self.reads_act_win: List[int] = []

    # ---- reference field names as views -------------------------------------------------
    @property
    def K_sel(self):
        return self._K_sel[:, :, : self.t]

    @property
    def V_sel(self):
        return self._V_sel[:, :, : self.t]

    @property
    def K_win(self):  # the reference keeps only the last w tokens; the window is applied by the reader here
        return self._K_win[:, :, max(0, self.t - self.w): self.t]

    @property
    def V_win(self):
        return self._V_win[:, :, max(0, self.t - self.w): self.t]

    @property
    def K_cmp_raw_seq(self):
        return self._K_raw[:, :, : self.t]

    @property
    def V_cmp_raw_seq(self):
        return self._V_raw[:, :, : self.t]

    @property
    def K_cmp(self):
        return self._K_cmp[:, :, : self.n_cmp]

    @property
    def V_cmp(self):
        return self._V_cmp[:, :, : self.n_cmp]

    # ---- capacity -------------------------------------------------------------------------
    def reserve(self, S_max: int) -> None:
        """grow the buffers to hold S_max tokens, keeping what is stored (the one O(S) copy the reference pays on every step,
        kv_cache.py:28-30, paid here once per doubling).  Cached native descriptors of the old buffers are dropped."""
        if S_max <= self._K_sel.shape[2]:
            return
        n_cmp_max = 0 if S_max < self.l else (S_max - self.l) // self.d + 1

        def grow(buf, n, used):
            new = torch.empty((self.B, self.G, max(n, 1), buf.shape[3]), device=buf.device, dtype=buf.dtype)
            if used:
                new[:, :, :used] = buf[:, :, :used]
            return new

        for name in ("_K_sel", "_V_sel", "_K_win", "_V_win", "_K_raw", "_V_raw"):
            setattr(self, name, grow(getattr(self, name), S_max, self.t))
        self._K_cmp, self._V_cmp = grow(self._K_cmp, n_cmp_max, self.n_cmp), grow(self._V_cmp, n_cmp_max, self.n_cmp)
        self.S_max = S_max
        for attr in ("_desc", "_dec_ctx", "_blk_ctx"):
            self.__dict__.pop(attr, None)

    def ensure_capacity(self, n_tokens: int) -> None:
        """room for n_tokens in total: grows geometrically (auto_grow) or raises"""
        cap = self._K_sel.shape[2]
        if n_tokens <= cap:
            return
        if not self.auto_grow:
            raise RuntimeError(f"NSA_KV capacity exceeded: {n_tokens} > S_max={self.S_max}")
        self.reserve(max(n_tokens, 2 * cap))

    # ---- updates --------------------------------------------------------------------------
    def write_tokens(self, K_sel, V_sel, K_win, V_win, K_raw, V_raw) -> None:
        """append S new tokens ([B,G,S,D] each) at position t (update_selection_raw/update_window/append_cmp_raw)."""
        S = K_sel.shape[2]
        self.ensure_capacity(self.t + S)
        sl = slice(self.t, self.t + S)
        self._K_sel[:, :, sl], self._V_sel[:, :, sl] = K_sel, V_sel
        self._K_win[:, :, sl], self._V_win[:, :, sl] = K_win, V_win
        self._K_raw[:, :, sl], self._V_raw[:, :, sl] = K_raw, V_raw
        self.t += S

    def write_compressed(self, K_cmp, V_cmp, at: Optional[int] = None) -> None:
        n = K_cmp.shape[2]
        at = self.n_cmp if at is None else at
        self._K_cmp[:, :, at: at + n], self._V_cmp[:, :, at: at + n] = K_cmp, V_cmp
        self.n_cmp = at + n

    def ensure_meta(self, seq_len: int) -> BlockMeta:
        """block metadata for seq_len tokens; metadata objects are immutable and shared through a small process-wide cache
        (the host build + the upload of the Eq.9 map cost ~0.15 ms, as much as a 4k prefill's attention kernel)"""
        key = (seq_len, self.l, self.d, self.l_sel, self.n_sel, self.w)
        meta = _META_CACHE.get(key)
        if meta is None:
            if len(_META_CACHE) >= 256:
                _META_CACHE.pop(next(iter(_META_CACHE)))
            meta = _META_CACHE[key] = build_block_meta(seq_len, self.l, self.d, self.l_sel, self.n_sel, self.w)
        self.meta = meta
        self.meta_seq_len = seq_len
        return self.meta

    def append_reads(self, num_cmp: int, S_raw: int) -> None:
        sel, win = self.n_sel * self.l_sel, min(self.w, S_raw)
        total = num_cmp + sel + win  # the reads formula of nsa_attention.py:634-635
        self.reads_pred.append(total)
        self.reads_act_total.append(total)
        self.reads_act_sel.append(sel)
        self.reads_act_cmp.append(num_cmp)
        self.reads_act_win.append(win)
