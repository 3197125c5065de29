"""nsa_vibe_amd -- MI355X (gfx950) implementation of nsa-vibe's selected-branch attention hot path.

Host side = PyTorch-ROCm plumbing that mirrors the reference's operator surface
(nsa/core/{block_index,selection_scorer,attention_kernels}.py, nsa/kernels/cuda_sel_kernel);
the arithmetic lives in hand-written HIP kernels behind the C ABI include/nsa_sel_hip.h
(libnsa_sel_hip.so).  There is no CPU or eager fallback: a missing library raises.
"""
from . import _lib  # noqa: F401
from .band_attention import (  # noqa: F401
    band_attention_hip,
    batched_causal_attention_compressed,
    batched_causal_attention_compressed_first_key_parity,
    sliding_window_attention,
)
from .block_index import BlockMeta, build_block_meta, build_block_starts, build_M_csl_csr  # noqa: F401
from .selection_attention import (  # noqa: F401
    grouped_selection_attention,
    grouped_selection_attention_masked,
    grouped_selection_attention_packed,
    selection_attention_cuda,
    selection_attention_varlen_all,
    selection_attention_varlen_all_v2,
    hip_sel_available,
    select_and_attend,
    selection_attention_first_key_parity,
    selection_attention_head_causal_parity,
    selection_attention_hip,
    selection_decode_step,
)
from .selection_scorer import (  # noqa: F401
    batched_ranges_width,
    compute_pcmp,
    compute_pcmp_all,
    convert_indices_to_ranges_batched,
    convert_indices_to_ranges_batched_dispatch,
    convert_indices_to_ranges_batched_v2,
    group_reduce_pslc,
    map_pcmp_to_pgrp,
    map_pcmp_to_pslc,
    map_pcmp_to_pslc_batched,
    map_pcmp_to_pslc_slow_path,
    select_topn_ranges,
    select_topn_ranges_batched,
    select_topn_ranges_rows,
    selection_scores,
    selection_scores_select,
    validate_selection_determinism,
    verify_mapping_equivalence,
)

__version__ = "0.1.0"
from .kv_cache import NSA_KV  # noqa: E402,F401
from .nsa_attention import GateMLP, NSAAttention  # noqa: E402,F401
