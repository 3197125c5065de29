"""Block metadata for the Eq.9 compressed->selection map.

Mirror of the reference's nsa/core/block_index.py (same names, fields, dtypes, error
behaviour) with two differences that matter on the device:
  * the CSR is built in closed form, O(nnz), by the C ABI (`nsa_build_block_meta_host`)
    instead of the reference's O(S_cmp*S_sel) Python double loop (block_index.py:51-60);
  * the gather form (CSC: per selection block the (cmp row, weight) list in ascending row
    order) is kept resident on each device, so nothing is `.to(device)`-ed per call
    (the reference re-uploads the COO on every call, selection_scorer.py:100-101).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, Tuple

import numpy as np
import torch

from . import _lib


@dataclass
class BlockMeta:
    """Same public fields as the reference BlockMeta (nsa/core/block_index.py:7-22)."""

    l: int
    d: int
    l_sel: int
    n_sel: int
    w: int
    cmp_starts: torch.Tensor  # [S_cmp] int32
    sel_starts: torch.Tensor  # [S_sel] int32
    M_csl_indptr: torch.Tensor
    M_csl_indices: torch.Tensor
    M_csl_values: torch.Tensor
    M_csl_coo_indices: torch.Tensor  # [2,nnz] rows, cols
    M_csl_coo_values: torch.Tensor  # [nnz]
    # gather form (host) + per-device copies
    csc_ptr: torch.Tensor = None  # [S_sel+1] int32
    csc_rows: torch.Tensor = None  # [nnz] int32
    csc_vals: torch.Tensor = None  # [nnz] fp32
    _dev: Dict[str, Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = field(default_factory=dict, repr=False)

    @property
    def S_cmp(self) -> int:
        return int(self.cmp_starts.numel())

    @property
    def S_sel(self) -> int:
        return int(self.sel_starts.numel())

    def device_csc(self, device) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        key = str(device)
        if key not in self._dev:
            self._dev[key] = tuple(t.to(device) for t in (self.csc_ptr, self.csc_rows, self.csc_vals))
        return self._dev[key]


def build_block_starts(seq_len: int, l: int, d: int, l_sel: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """nsa/core/block_index.py:25-36."""
    if d <= 0 or l <= 0 or l_sel <= 0:
        raise ValueError("Block parameters must be positive")
    max_cmp = 0 if seq_len < l else (seq_len - l) // d + 1
    max_sel = 0 if seq_len <= 0 else (seq_len + l_sel - 1) // l_sel
    return (torch.arange(max_cmp, dtype=torch.int32) * d, torch.arange(max_sel, dtype=torch.int32) * l_sel)


def _build_arrays(seq_len: int, l: int, d: int, l_sel: int):
    L = _lib.lib()
    s_cmp, s_sel, nnz = C.c_int(), C.c_int(), C.c_int()
    rc = L.nsa_block_counts(seq_len, l, d, l_sel, C.byref(s_cmp), C.byref(s_sel), C.byref(nnz))
    if rc != 0:
        raise ValueError(_lib.last_error())
    S_cmp, S_sel, n = s_cmp.value, s_sel.value, nnz.value
    indptr = torch.zeros(S_cmp + 1, dtype=torch.int32)
    indices = torch.zeros(n, dtype=torch.int32)
    values = torch.zeros(n, dtype=torch.float32)
    cptr = torch.zeros(S_sel + 1, dtype=torch.int32)
    crows = torch.zeros(n, dtype=torch.int32)
    cvals = torch.zeros(n, dtype=torch.float32)
    rc = L.nsa_build_block_meta_host(seq_len, l, d, l_sel, indptr.data_ptr(), indices.data_ptr(), values.data_ptr(),
                                     cptr.data_ptr(), crows.data_ptr(), cvals.data_ptr())
    if rc != 0:
        raise ValueError(_lib.last_error())
    return indptr, indices, values, cptr, crows, cvals


def build_M_csl_csr(seq_len: int, l: int, d: int, l_sel: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """nsa/core/block_index.py:43-71 (fractional-overlap CSR)."""
    if d <= 0 or l <= 0 or l_sel <= 0:
        raise ValueError("Block parameters must be positive")
    indptr, indices, values, *_ = _build_arrays(seq_len, l, d, l_sel)
    return indptr, indices, values


def build_block_meta(seq_len: int, l: int, d: int, l_sel: int, n_sel: int, w: int) -> BlockMeta:
    """nsa/core/block_index.py:74-99."""
    if l % d != 0 or l_sel % d != 0:
        raise ValueError("Require d|l and d|l_sel in M0")
    cmp_starts, sel_starts = build_block_starts(seq_len, l, d, l_sel)
    indptr, indices, values, cptr, crows, cvals = _build_arrays(seq_len, l, d, l_sel)
    # (numpy: torch.repeat_interleave on these few hundred host integers was measured at 18 ms per call on the GPU box's 128-thread host --
    # a decode loop rebuilds the metadata every l_sel tokens)
    ip = indptr.numpy()
    rows = torch.from_numpy(np.repeat(np.arange(cmp_starts.numel(), dtype=np.int32), (ip[1:] - ip[:-1]).astype(np.int64)))
    coo = torch.stack([rows.to(torch.int32), indices.clone()], dim=0)
    return BlockMeta(l=l, d=d, l_sel=l_sel, n_sel=n_sel, w=w, cmp_starts=cmp_starts, sel_starts=sel_starts,
                     M_csl_indptr=indptr, M_csl_indices=indices, M_csl_values=values, M_csl_coo_indices=coo,
                     M_csl_coo_values=values.clone(), csc_ptr=cptr, csc_rows=crows, csc_vals=cvals)
