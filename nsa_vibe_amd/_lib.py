"""ctypes binding of libnsa_sel_hip.so (C ABI: include/nsa_sel_hip.h).

The library is the product: there is NO CPU or PyTorch fallback behind it.  If it is missing
or a call fails this module raises (ImportError / RuntimeError) -- loudly, by design.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnsa_sel_hip.so")
CSRC = os.path.join(_HERE, "csrc")

NSA_DT_F32, NSA_DT_BF16, NSA_DT_F16 = 0, 1, 2
NSA_SEL_SEQUENTIAL, NSA_SEL_BATCHED = 0, 1

_vp, _i, _i64, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t



class NsaLayerDesc(C.Structure):
    """struct nsa_layer_desc (include/nsa_sel_hip.h)"""
    _fields_ = [(n, _i) for n in ("dim", "G", "h", "Dk", "Dv", "l", "d", "l_sel", "n_sel", "w", "gate_hidden", "dtype")] + \
               [(n, _f) for n in ("rope_base", "rope_scale", "gate_tau")] + \
               [(n, _vp) for n in ("W_qkv", "W_out", "gate_w1", "gate_b1", "gate_w2", "gate_b2")]


class NsaKvDesc(C.Structure):
    """struct nsa_kv_desc (include/nsa_sel_hip.h)"""
    _fields_ = [(n, _vp) for n in ("K_sel", "V_sel", "K_win", "V_win", "K_raw", "V_raw", "K_cmp", "V_cmp")] + \
               [(n, _i) for n in ("B", "S_max", "n_cmp_max")]


class NsaBlockDesc(C.Structure):
    """struct nsa_block_desc (include/nsa_sel_hip.h)"""
    _fields_ = [("attn", NsaLayerDesc), ("norm1_w", _vp), ("norm2_w", _vp), ("mlp_w1", _vp), ("mlp_w2", _vp), ("mlp_hidden", _i),
                ("norm_eps", _f)]


_pl, _pk, _pb = C.POINTER(NsaLayerDesc), C.POINTER(NsaKvDesc), C.POINTER(NsaBlockDesc)

# name -> (restype, argtypes); mirrors include/nsa_sel_hip.h one to one
SIGNATURES = {
    "nsa_hip_abi_version": (_i, []),
    "nsa_hip_last_error": (C.c_char_p, []),
    "nsa_hip_device_check": (_i, [_i, C.POINTER(_i), C.POINTER(_sz)]),
    "nsa_hip_set_tuning": (_i, [C.c_char_p, _i]),
    "nsa_hip_get_tuning": (_i, [C.c_char_p, C.POINTER(_i)]),
    "nsa_sel_attn_fwd_workspace": (_sz, [_i] * 8),
    "nsa_sel_attn_fwd_workspace_kv": (_sz, [_i] * 9),
    "nsa_sel_attn_fwd": (_i, [_vp] * 6 + [_i] * 8 + [_i64] * 6 + [_i, _f, _i, _vp, _sz, _vp]),
    "nsa_sel_attn_first_key_parity": (_i, [_vp] * 3 + [_i] * 7 + [_i64] * 3 + [_i, _vp]),
    "nsa_sel_attn_head_causal_parity": (_i, [_vp] * 5 + [_i] * 8 + [_i64] * 6 + [_i, _f, _vp]),
    "nsa_sel_attn_bwd_workspace": (_sz, [_i] * 9),
    "nsa_sel_attn_bwd": (_i, [_vp] * 10 + [_i] * 8 + [_i64] * 6 + [_i, _f, _i, _vp, _sz, _vp]),
    "nsa_band_attn_fwd_workspace": (_sz, [_i] * 7),
    "nsa_band_attn_fwd": (_i, [_vp] * 5 + [_i] * 7 + [_i64] * 6 + [_i] * 5 + [_i, _f, _i, _vp, _sz, _vp]),
    "nsa_linear_small": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "nsa_rmsnorm_rows": (_i, [_vp, _vp, _vp, _i, _i, _f, _i, _vp]),
    "nsa_rmsnorm_rows_bwd_workspace": (_sz, [_i, _i]),
    "nsa_rmsnorm_rows_bwd": (_i, [_vp] * 5 + [_i, _i, _f, _i, _vp, _sz, _vp]),
    "nsa_model_decode_step_workspace": (_sz, [_pb, _i, _i, _i]),
    "nsa_model_decode_step": (_i, [_pb, _pk, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "nsa_block_decode_step_workspace": (_sz, [_pb, _i, _i]),
    "nsa_block_decode_step": (_i, [_pb, _pk, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "nsa_rope_cache_append": (_i, [_pl, _pk, _vp, _vp, _i, _i, _vp]),
    "nsa_cmp_pool_append": (_i, [_pl, _pk, _i, _i, _vp]),
    "nsa_gate_combine": (_i, [_pl, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "nsa_rope_cache_append_bwd": (_i, [_pl, _i, _i, _i] + [_vp] * 8 + [_vp]),
    "nsa_cmp_pool_bwd": (_i, [_pl, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "nsa_gate_combine_bwd": (_i, [_pl] + [_vp] * 9 + [_i64, _vp]),
    "nsa_layer_prefill_workspace": (_sz, [_pl, _i, _i, _i]),
    "nsa_layer_prefill": (_i, [_pl, _pk, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "nsa_layer_decode_step_workspace": (_sz, [_pl, _i, _i]),
    "nsa_layer_decode_step": (_i, [_pl, _pk, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "nsa_band_attn_bwd_workspace": (_sz, [_i] * 9),
    "nsa_band_attn_bwd": (_i, [_vp] * 9 + [_i] * 7 + [_i64] * 6 + [_i] * 5 + [_i, _f, _i, _vp, _sz, _vp]),
    "nsa_block_counts": (_i, [_i] * 4 + [C.POINTER(_i)] * 3),
    "nsa_build_block_meta_host": (_i, [_i] * 4 + [_vp] * 6),
    "nsa_map_pcmp_to_pgrp": (_i, [_vp, _i64, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "nsa_pcmp_all": (_i, [_vp, _vp, _vp] + [_i] * 6 + [_i64] * 3 + [_i, _f, _vp]),
    "nsa_sel_scores_workspace": (_sz, [_i] * 12),
    "nsa_sel_scores": (_i, [_vp, _vp, _vp] + [_i] * 6 + [_i64] * 3 + [_vp, _vp, _vp] + [_i] * 7 + [_f, _vp, _sz, _vp]),
    "nsa_sel_scores_select": (_i, [_vp, _vp, _vp] + [_i] * 6 + [_i64] * 3 + [_vp, _vp, _vp] + [_i] * 6 + [_f] + [_i] * 6 + [_vp, _i, _vp, _sz, _vp]),
    "nsa_sel_decode_step_workspace": (_sz, [_i] * 9),
    "nsa_sel_decode_step": (_i, [_vp] * 9 + [_i] * 13 + [_i64] * 9 + [_i, _f, _vp, _sz, _vp]),
    "nsa_batched_ranges_width": (_i, [_i] * 6),
    "nsa_select_topn_ranges": (_i, [_vp, _i64, _i, _i, _i, _vp] + [_i] * 7 + [_vp, _i, _vp]),
    "nsa_sel_select_attn_fwd": (_i, [_vp, _i, _vp] + [_i] * 7 + [_vp, _i] + [_vp] * 5 + [_i] * 7 + [_i64] * 6 + [_i, _f, _vp, _sz, _vp]),
    "nsa_indices_to_ranges_v2": (_i, [_vp, _i64] + [_i] * 6 + [_vp, _vp]),
}

_lib = None
_loaded_path = None


def build(verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1))]
    r = subprocess.run(cmd, capture_output=not verbose, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libnsa_sel_hip.so failed:\n" + (r.stdout or "") + (r.stderr or ""))
    return LIB_PATH


def lib():
    """Load the C-ABI library (once).  Raises ImportError if it has not been built."""
    global _lib, _loaded_path
    if _lib is None:
        # NSA_HIP_LIB: measurement aid only -- another BUILD of this library (an A/B pair, a TIMELINE or ablation build under ab/, see
        # csrc/Makefile) is loaded in place of the product without ever being copied over it; bench.py and smoke() print which file ran
        path = os.environ.get("NSA_HIP_LIB") or LIB_PATH
        if not os.path.exists(path):
            raise ImportError(
                f"{path} not found: the HIP extension is required (no fallback path exists). "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C nsa_vibe_amd/csrc`.")
        L = C.CDLL(path)
        _loaded_path = os.path.abspath(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        if L.nsa_hip_abi_version() != 1:
            raise ImportError("libnsa_sel_hip.so ABI version mismatch")
        _lib = L
    return _lib


def loaded_library() -> dict:
    """path and sha256 (first 16 hex digits) of the library this process runs on: recorded by bench.py and smoke()"""
    import hashlib

    lib()
    with open(_loaded_path, "rb") as f:
        sha = hashlib.sha256(f.read()).hexdigest()[:16]
    return {"path": os.path.relpath(_loaded_path, os.path.dirname(_HERE)), "sha16": sha, "product": _loaded_path == os.path.abspath(LIB_PATH)}


def set_tuning(name: str, value: int) -> None:
    """process-wide A/B switch of the native library (include/nsa_sel_hip.h: nsa_hip_set_tuning); -1 = automatic"""
    check(lib().nsa_hip_set_tuning(name.encode(), int(value)), "nsa_hip_set_tuning")


def get_tuning(name: str) -> int:
    v = _i(0)
    check(lib().nsa_hip_get_tuning(name.encode(), C.byref(v)), "nsa_hip_get_tuning")
    return v.value


def last_error() -> str:
    return (lib().nsa_hip_last_error() or b"").decode()


def check(rc: int, what: str) -> None:
    """Non-zero status -> RuntimeError (the exception type the reference's router catches,
    nsa/kernels/cuda_sel_kernel/__init__.py:60-68)."""
    if rc != 0:
        raise RuntimeError(f"{what} failed (status {rc}): {last_error()}")
