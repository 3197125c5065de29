"""Selection scoring and deterministic top-n range selection on MI355X.

Host-side mirror of the reference's nsa/core/selection_scorer.py: same function names,
argument meaning and output layouts; every function runs hand-written HIP kernels through
the C ABI (include/nsa_sel_hip.h).  Tensors must live on a HIP device -- there is no CPU path.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import _lib
from .block_index import BlockMeta

_DT = {torch.float32: _lib.NSA_DT_F32, torch.bfloat16: _lib.NSA_DT_BF16, torch.float16: _lib.NSA_DT_F16}
_WS: Dict[tuple, torch.Tensor] = {}


def _need_gpu(*ts: torch.Tensor) -> torch.device:
    dev = ts[0].device
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError("nsa_vibe_amd kernels need HIP device tensors (no CPU fallback exists)")
        if t.device != dev:
            raise RuntimeError("all tensors must be on the same device")
    return dev


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(dev: torch.device) -> int:
    """raw handle of torch's current stream on dev (a decode step makes this call once per layer: the Stream object of
    torch.cuda.current_stream costs tens of microseconds of host time, the raw getter about one)"""
    if _raw_stream is not None:
        idx = dev.index if isinstance(dev, torch.device) else None
        return _raw_stream(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(dev).cuda_stream


def workspace(dev: torch.device, nbytes: int, tag: str = "ws") -> Optional[torch.Tensor]:
    """Grow-on-demand scratch per (device, stream) (the reference keeps process-global workspaces too,
    nsa/core/attention_kernels.py:25-26,64-103; keyed by stream here so concurrent streams cannot race on it)."""
    if nbytes <= 0:
        return None
    key = (tag, dev.index, _stream(dev))  # per stream: two streams never share scratch
    buf = _WS.get(key)
    # grown on demand; given back when a much smaller shape follows a big one (the key-split records of a 64k prefill are gigabytes)
    if buf is None or buf.numel() < nbytes or buf.numel() > max(4 * nbytes, 256 << 20):
        _WS.pop(key, None)
        buf = None
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
        _WS[key] = buf
    return buf


def _kc_strides(K_cmp: torch.Tensor):
    if K_cmp.stride(-1) != 1:
        K_cmp = K_cmp.contiguous()
    return K_cmp, K_cmp.stride(0), K_cmp.stride(1), K_cmp.stride(2)


def compute_pcmp_all(Q_all: torch.Tensor, K_cmp: torch.Tensor, scale: float, out_dtype=None) -> torch.Tensor:
    """Q_all [B,S,G,h,Dk], K_cmp [B,G,S_cmp,Dk] -> p_cmp_all [B,S,G,h,S_cmp]
    (reference: selection_scorer.py:42-61; softmax over ALL S_cmp columns, fp32 math).
    Result dtype = Q's dtype as in the reference unless out_dtype is given."""
    dev = _need_gpu(Q_all, K_cmp)
    B, S, G, h, Dk = Q_all.shape
    S_cmp = K_cmp.shape[2]
    Q_all = Q_all.contiguous()
    K_cmp, sb, sg, ss = _kc_strides(K_cmp)
    p = torch.empty((B, S, G, h, S_cmp), dtype=torch.float32, device=dev)
    if p.numel():
        rc = _lib.lib().nsa_pcmp_all(Q_all.data_ptr(), K_cmp.data_ptr(), p.data_ptr(), B, S, G, h, Dk, S_cmp, sb, sg, ss,
                                     _DT[Q_all.dtype], float(scale), _stream(dev))
        _lib.check(rc, "nsa_pcmp_all")
    return p.to(out_dtype or Q_all.dtype)


def _map(p_cmp_all: torch.Tensor, meta: BlockMeta, want_pslc: bool):
    dev = _need_gpu(p_cmp_all)
    lead = p_cmp_all.shape[:-2]
    h, S_cmp = p_cmp_all.shape[-2:]
    S_sel = meta.S_sel
    R = 1
    for x in lead:
        R *= int(x)
    p32 = p_cmp_all.to(torch.float32).contiguous()
    p_grp = torch.zeros((*lead, S_sel), dtype=torch.float32, device=dev)
    p_slc = torch.zeros((*lead, h, S_sel), dtype=torch.float32, device=dev) if want_pslc else None
    if S_cmp > 0 and R > 0 and S_sel > 0:  # selection_scorer.py:97-98: zeros when S_cmp == 0
        cptr, crows, cvals = meta.device_csc(dev)
        rc = _lib.lib().nsa_map_pcmp_to_pgrp(p32.data_ptr(), R, h, S_cmp, cptr.data_ptr(), crows.data_ptr(), cvals.data_ptr(),
                                             S_sel, p_slc.data_ptr() if want_pslc else None, p_grp.data_ptr(), _stream(dev))
        _lib.check(rc, "nsa_map_pcmp_to_pgrp")
    return p_slc, p_grp


def map_pcmp_to_pslc_batched(p_cmp_all: torch.Tensor, meta: BlockMeta) -> torch.Tensor:
    """p_cmp_all [B,S,G,h,S_cmp] -> p_slc_all [B,S,G,h,S_sel] (Eq.9; selection_scorer.py:89-116).
    Bit-identical to the reference's CPU result for an identical fp32 input."""
    p_slc, _ = _map(p_cmp_all, meta, True)
    return p_slc.to(p_cmp_all.dtype)


def map_pcmp_to_pslc(p_cmp: torch.Tensor, meta: BlockMeta) -> torch.Tensor:
    """Per-step form, p_cmp [B,G,h,S_cmp] -> [B,G,h,S_sel] (selection_scorer.py:64-86)."""
    return map_pcmp_to_pslc_batched(p_cmp, meta)


def map_pcmp_to_pgrp(p_cmp_all: torch.Tensor, meta: BlockMeta) -> torch.Tensor:
    """Eq.9 followed by Eq.10 in one kernel: [...,h,S_cmp] -> fp32 [...,S_sel] (heads summed in ascending h)."""
    return _map(p_cmp_all, meta, False)[1]


def group_reduce_pslc(p_slc: torch.Tensor) -> torch.Tensor:
    """Eq.10 (selection_scorer.py:119-121): sum over the heads of a group, dim=2 of [B,G,h,S_sel];
    accumulated in ascending h so the result does not depend on a library reduction order."""
    acc = p_slc.select(2, 0).clone()
    for i in range(1, p_slc.shape[2]):
        acc = acc + p_slc.select(2, i)
    return acc


def selection_scores(Q_all: torch.Tensor, K_cmp: torch.Tensor, meta: BlockMeta, scale: Optional[float] = None,
                     causal_skip: bool = False, variant: int = 0, leave_skipped: bool = False) -> torch.Tensor:
    """Fused A2+A3+A4: Q [B,S,G,h,Dk], K_cmp [B,G,S_cmp,Dk] -> p_grp [B,S,G,S_sel] fp32 without
    materialising p_cmp (nsa_attention.py:1073-1091 in one call).

    causal_skip=True does not compute the blocks neither selector can pick at row t ((j+1) l' > t+1; both mask
    them to -inf, selection_scorer.py:156,276-280): such an entry holds 0 or, where its workgroup computed the block for a later row, the
    full value; with leave_skipped=True those entries are left
    uninitialised (no zero fill of the tensor) -- for results that go straight to the selectors.  variant: 0 auto, 1 generic
    (any dtype/geometry, query-chunked), 2 the MFMA kernel (bf16/f16, default block geometry)."""
    dev = _need_gpu(Q_all, K_cmp)
    B, S, G, h, Dk = Q_all.shape
    S_cmp, S_sel = K_cmp.shape[2], meta.S_sel
    Q_all = Q_all.contiguous()
    K_cmp, sb, sg, ss = _kc_strides(K_cmp)
    p_grp = torch.empty((B, S, G, S_sel), dtype=torch.float32, device=dev)
    if p_grp.numel() == 0:
        return p_grp
    L = _lib.lib()
    if variant == 0 and B * S * G > 1024 and not (sb % 8 == 0 and sg % 8 == 0 and ss % 8 == 0 and K_cmp.data_ptr() % 16 == 0):
        variant = 1  # rows not 16-byte aligned: the MFMA route would refuse them
    nbytes = L.nsa_sel_scores_workspace(B, S, G, h, Dk, S_cmp, S_sel, int(meta.l), int(meta.d), int(meta.l_sel), _DT[Q_all.dtype],
                                        int(variant))
    ws = workspace(dev, nbytes, "scores")
    cptr, crows, cvals = meta.device_csc(dev)
    rc = L.nsa_sel_scores(Q_all.data_ptr(), K_cmp.data_ptr(), p_grp.data_ptr(), B, S, G, h, Dk, S_cmp, sb, sg, ss,
                          cptr.data_ptr(), crows.data_ptr(), cvals.data_ptr(), S_sel, int(meta.l), int(meta.d), int(meta.l_sel),
                          (2 if leave_skipped else 1) if causal_skip else 0, int(variant), _DT[Q_all.dtype], float(scale) if scale else 0.0,
                          ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, _stream(dev))
    _lib.check(rc, "nsa_sel_scores")
    return p_grp


def selection_scores_select(Q_all: torch.Tensor, K_cmp: torch.Tensor, meta: BlockMeta, n_top: int, *, mode: str = "batched", t0: int = 0,
                            scale: Optional[float] = None, force_init: bool = True, force_local: int = 2, leave_skipped: bool = True):
    """Prefill (inference): group scores AND the top-n ranges of every row in ONE native call (nsa_sel_scores_select) -- on the MFMA scorer's
    default route (h = 6, Dk = 64, default block geometry, bf16 / f16, up to 1024 selection blocks) in one LAUNCH: a workgroup selects the
    ranges of its 64 query rows right behind its second sweep, reading its scores back out of L2.  Q [B,S,G,h,Dk], K_cmp [B,G,S_cmp,Dk] ->
    (p_grp [B,S,G,S_sel] fp32 with blocks no selector can read skipped, ranges [B,S,G,W,2] int32).  mode "batched" =
    select_topn_ranges_batched (nsa/core/selection_scorer.py:255-362), "sequential" = select_topn_ranges at token t0 + s (:124-249).
    ranges are bit-identical to selection_scores(..., causal_skip=True) followed by select_topn_ranges_batched / _rows."""
    dev = _need_gpu(Q_all, K_cmp)
    B, S, G, h, Dk = Q_all.shape
    S_cmp, S_sel = K_cmp.shape[2], meta.S_sel
    if mode == "batched":
        md, W = _lib.NSA_SEL_BATCHED, batched_ranges_width(S_sel, meta.l_sel, n_top, S, force_init, force_local)
    elif mode == "sequential":
        md, W = _lib.NSA_SEL_SEQUENTIAL, n_top
    else:
        raise ValueError("mode must be 'batched' or 'sequential'")
    Q_all = Q_all.contiguous()
    K_cmp, sb, sg, ss = _kc_strides(K_cmp)
    p_grp = torch.empty((B, S, G, S_sel), dtype=torch.float32, device=dev)
    ranges = torch.empty((B, S, G, W, 2), dtype=torch.int32, device=dev)
    if p_grp.numel() == 0 or W == 0:
        return p_grp, ranges
    L = _lib.lib()
    dt = _DT[Q_all.dtype]
    geo = (int(meta.l), int(meta.d), int(meta.l_sel))
    nbytes = max(L.nsa_sel_scores_workspace(B, S, G, h, Dk, S_cmp, S_sel, *geo, dt, 0), L.nsa_sel_scores_workspace(B, S, G, h, Dk, S_cmp, S_sel, *geo, dt, 1))
    ws = workspace(dev, nbytes, "scores")
    cptr, crows, cvals = meta.device_csc(dev)
    rc = L.nsa_sel_scores_select(Q_all.data_ptr(), K_cmp.data_ptr(), p_grp.data_ptr(), B, S, G, h, Dk, S_cmp, sb, sg, ss,
                                 cptr.data_ptr(), crows.data_ptr(), cvals.data_ptr(), S_sel, *geo, 2 if leave_skipped else 1, dt,
                                 float(scale) if scale else 0.0, int(t0), int(n_top), int(bool(force_init)), int(force_local), md, S,
                                 ranges.data_ptr(), W, ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0,
                                 _stream(dev))
    _lib.check(rc, "nsa_sel_scores_select")
    return p_grp, ranges


def _select(p_rows: torch.Tensor, R: int, S: int, G: int, t0: int, t_rows, meta: BlockMeta, n_top: int, force_init,
            force_local, mode: int, S_total: int, width: int) -> torch.Tensor:
    dev = p_rows.device
    out = torch.empty((R, width, 2), dtype=torch.int32, device=dev)
    if R == 0 or width == 0:
        return out
    p32 = p_rows.to(torch.float32).contiguous()
    rc = _lib.lib().nsa_select_topn_ranges(p32.data_ptr(), R, S, G, int(t0), t_rows.data_ptr() if t_rows is not None else None,
                                           meta.S_sel if p_rows.shape[-1] == meta.S_sel else p_rows.shape[-1], int(meta.l_sel),
                                           int(n_top), int(bool(force_init)), int(force_local), mode, int(S_total),
                                           out.data_ptr(), width, _stream(dev))
    _lib.check(rc, "nsa_select_topn_ranges")
    return out


def select_topn_ranges(p_grp: torch.Tensor, meta: BlockMeta, n_top: int, t_token: int, force_init: bool = True,
                       force_local: int = 2, _skip_validation: bool = False) -> torch.Tensor:
    """p_grp [B,G,S_sel] -> int32 [B,G,n_top,2] (decode / sequential prefill; selection_scorer.py:124-249).
    Where the reference would emit inverted garbage ranges (fewer valid candidates than n_top-3)
    this returns [0,0] padding instead; compare on the entries with end > start."""
    _need_gpu(p_grp)
    B, G, S_sel = p_grp.shape
    out = _select(p_grp.reshape(B * G, S_sel), B * G, 1, G, int(t_token), None, meta, n_top, force_init, force_local,
                  _lib.NSA_SEL_SEQUENTIAL, 1, n_top)
    return out.view(B, G, n_top, 2)


def select_topn_ranges_rows(p_grp_all: torch.Tensor, meta: BlockMeta, n_top: int, t0: int = 0, force_init: bool = True,
                            force_local: int = 2) -> torch.Tensor:
    """Sequential-mode selector for every row of p_grp_all [B,S,G,S_sel] at token t0+s in ONE launch
    (what the reference's default prefill computes with a Python loop over t, nsa_attention.py:1574-1576)."""
    _need_gpu(p_grp_all)
    B, S, G, S_sel = p_grp_all.shape
    out = _select(p_grp_all.reshape(B * S * G, S_sel), B * S * G, S, G, t0, None, meta, n_top, force_init, force_local,
                  _lib.NSA_SEL_SEQUENTIAL, S, n_top)
    return out.view(B, S, G, n_top, 2)


def batched_ranges_width(meta_or_S_sel, l_sel: int, n_top: int, S: int, force_init: bool = True, force_local: int = 2) -> int:
    S_sel = meta_or_S_sel.S_sel if isinstance(meta_or_S_sel, BlockMeta) else int(meta_or_S_sel)
    return int(_lib.lib().nsa_batched_ranges_width(int(S), S_sel, int(l_sel), int(n_top), int(bool(force_init)), int(force_local)))


def select_topn_ranges_batched(p_grp_all: torch.Tensor, meta: BlockMeta, n_top: int, S: int, force_init: bool = True,
                               force_local: int = 2) -> torch.Tensor:
    """p_grp_all [B,S,G,S_sel] -> int32 [B,S,G,K,2] (batched prefill / training; selection_scorer.py:255-362
    followed by the v2 range converter :434-605).  Bit-exact including the zero padding."""
    _need_gpu(p_grp_all)
    B, S_q, G, S_sel = p_grp_all.shape
    K = batched_ranges_width(S_sel, meta.l_sel, n_top, S, force_init, force_local)
    out = _select(p_grp_all.reshape(B * S_q * G, S_sel), B * S_q * G, S_q, G, 0, None, meta, n_top, force_init, force_local,
                  _lib.NSA_SEL_BATCHED, S, K)
    return out.view(B, S_q, G, K, 2)


def convert_indices_to_ranges_batched_v2(indices: torch.Tensor, meta: BlockMeta, S: int) -> torch.Tensor:
    """indices [B,S,G,K] ascending, -1 padded -> int32 [B,S,G,K,2] (selection_scorer.py:434-605)."""
    dev = _need_gpu(indices)
    B, S_q, G, K = indices.shape
    out = torch.zeros((B, S_q, G, K, 2), dtype=torch.int32, device=dev)
    if out.numel():
        idx = indices.to(torch.int32).contiguous()
        rc = _lib.lib().nsa_indices_to_ranges_v2(idx.data_ptr(), B * S_q * G, S_q, G, 0, K, meta.S_sel, int(meta.l_sel),
                                                 out.data_ptr(), _stream(dev))
        _lib.check(rc, "nsa_indices_to_ranges_v2")
    return out


convert_indices_to_ranges_batched_dispatch = convert_indices_to_ranges_batched_v2


# ---- the rest of the reference's scorer surface: per-step form, converter names, verifiers --------------------------------------
def compute_pcmp(Q: torch.Tensor, K_cmp: torch.Tensor, scale: float) -> torch.Tensor:
    """Single-step form (selection_scorer.py:10-39): Q [B,G,h,Dk] (or [G,h,Dk] with an implicit batch of one), K_cmp [B,G,S_cmp,Dk]
    -> p_cmp [B,G,h,S_cmp], the softmax over all S_cmp columns."""
    if Q.dim() == 3:
        Q = Q.unsqueeze(0)
    return compute_pcmp_all(Q.unsqueeze(1), K_cmp, scale)[:, 0]


def convert_indices_to_ranges_batched(indices: torch.Tensor, meta: BlockMeta, S: int) -> torch.Tensor:
    """The reference's loop converter (selection_scorer.py:380-431) and its vectorised v2 (:434-605) define the same function (pinned
    by the g3 goldens); both names run the one HIP converter."""
    return convert_indices_to_ranges_batched_v2(indices, meta, S)


convert_indices_to_ranges_batched_dispatch = convert_indices_to_ranges_batched  # selection_scorer.py:365-377 (NSA_SEL_RANGES_V2 switch)


def map_pcmp_to_pslc_slow_path(p_cmp_all: torch.Tensor, meta: BlockMeta) -> torch.Tensor:
    """Eq.9 as a plain dense product p_cmp . M (M [S_cmp,S_sel] from the CSR arrays): the reference's "slow mathematical path"
    (selection_scorer.py:608-655) used to verify the fast mapping.  A checker, not a product path."""
    S_cmp = p_cmp_all.shape[-1]
    rows = meta.M_csl_coo_indices[0].long()
    keep = rows < S_cmp
    M = torch.zeros((S_cmp, meta.S_sel), dtype=torch.float32)
    M.index_put_((rows[keep], meta.M_csl_coo_indices[1].long()[keep]), meta.M_csl_coo_values.float()[keep], accumulate=True)
    return (p_cmp_all.to(torch.float32) @ M.to(p_cmp_all.device)).to(p_cmp_all.dtype)


def verify_mapping_equivalence(p_cmp_all: torch.Tensor, meta: BlockMeta, rtol: float = 1e-5, atol: float = 1e-8):
    """(ok, details) like the reference's verifier (selection_scorer.py:658-711): the HIP mapping against the dense product; runs only
    when NSA_VERIFY_EQ9_MAPPING is set, otherwise reports "skipped"."""
    import os

    if os.getenv("NSA_VERIFY_EQ9_MAPPING", "0").lower() not in ("1", "true", "yes"):
        return True, {"status": "skipped", "reason": "NSA_VERIFY_EQ9_MAPPING not set"}
    with torch.no_grad():
        fast = map_pcmp_to_pslc_batched(p_cmp_all, meta).float()
        slow = map_pcmp_to_pslc_slow_path(p_cmp_all, meta).float()
        diff = (fast - slow).abs()
        ok = bool(torch.allclose(fast, slow, rtol=rtol, atol=atol))
        return ok, {"status": "verified" if ok else "mismatch", "max_abs_diff": float(diff.max()) if diff.numel() else 0.0,
                    "mean_abs_diff": float(diff.mean()) if diff.numel() else 0.0,
                    "max_rel_diff": float((diff / (slow.abs() + atol)).max()) if diff.numel() else 0.0,
                    "shape": list(p_cmp_all.shape), "rtol": rtol, "atol": atol}


def validate_selection_determinism(p_grp: torch.Tensor, meta: BlockMeta, n_top: int, t_token: int, num_trials: int = 5) -> bool:
    """Repeat the sequential selector and compare (selection_scorer.py:714-760); runs only when NSA_VALIDATE_SELECTION_DETERMINISM is
    set.  The HIP selector has no atomics and a fixed tie rule, so this is a regression guard rather than a live risk."""
    import os

    if os.getenv("NSA_VALIDATE_SELECTION_DETERMINISM", "0").lower() not in ("1", "true", "yes") or p_grp.requires_grad:
        return True
    with torch.no_grad():
        first = select_topn_ranges(p_grp.clone(), meta, n_top, t_token, True, 2)
        return all(torch.equal(first, select_topn_ranges(p_grp.clone(), meta, n_top, t_token, True, 2)) for _ in range(num_trials - 1))
