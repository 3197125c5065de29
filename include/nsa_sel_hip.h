/*
 * nsa_sel_hip.h -- C ABI of libnsa_sel_hip.so: the MI355X (gfx950) implementation of
 * nsa-vibe's selected-branch attention hot path.
 *
 * This is the drop-in boundary.  Plain pointers and sizes only (no torch types).  Every
 * pointer is a DEVICE pointer unless its comment says "host".  All kernels are enqueued
 * on `stream` (a hipStream_t passed as void*; NULL = the default stream) and never
 * synchronise.  Inputs are borrowed and never written; outputs are caller allocated.
 *
 * Error convention (replaces the Python exceptions of the reference's plugin slot,
 * nsa/kernels/cuda_sel_kernel/__init__.py:60-68 and nsa/core/nsa_attention.py:764-782):
 * every entry point returns 0 on success, a negative NSA_ERR_* code otherwise, and
 * nsa_hip_last_error() returns a thread-local message.  The Python shim raises
 * RuntimeError on non-zero, which is what the reference's router catches.
 *
 * Reference interfaces replaced (paths relative to the reference repo):
 *   nsa_sel_attn_fwd           sel_forward(Q,K,V,ranges)            nsa/kernels/cuda_sel_kernel/sel_cuda.cpp:28-73
 *                              selection_attention_cuda             nsa/kernels/cuda_sel_kernel/__init__.py:47-68
 *                              grouped_selection_attention_masked   nsa/core/attention_kernels.py:705-772 (semantics)
 *   nsa_sel_attn_first_key_parity  grouped_selection_attention_packed / grouped_selection_attention (parity mode)
 *                                                                   nsa/core/attention_kernels.py:273-388, 181-226
 *   nsa_sel_attn_head_causal_parity  NSAAttention._sdpa_over_ranges (parity mode)   nsa/core/nsa_attention.py:1779-1855
 *   nsa_sel_attn_bwd           analytic backward / autograd of the masked SDPA
 *                                                                   nsa/kernels/triton_sel_kernel/__init__.py:125-231
 *   nsa_sel_scores             compute_pcmp_all + map_pcmp_to_pslc_batched + .sum(dim=3)
 *                                                                   nsa/core/selection_scorer.py:42-61, 89-116; nsa_attention.py:1073-1091
 *   nsa_pcmp_all               compute_pcmp_all                     nsa/core/selection_scorer.py:42-61
 *   nsa_map_pcmp_to_pgrp       map_pcmp_to_pslc_batched + group sum nsa/core/selection_scorer.py:89-121
 *   nsa_select_topn_ranges     select_topn_ranges / _batched (+v2)  nsa/core/selection_scorer.py:124-249, 255-362, 434-605
 *   nsa_indices_to_ranges_v2   convert_indices_to_ranges_batched_v2 nsa/core/selection_scorer.py:434-605
 *   nsa_batched_ranges_width   forced-column rule of the batched selector  selection_scorer.py:283-308,339-354
 *   nsa_build_block_meta_host  build_block_meta                     nsa/core/block_index.py:74-99
 */
#ifndef NSA_SEL_HIP_H
#define NSA_SEL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NSA_HIP_ABI_VERSION 1

#if defined(__GNUC__)
#define NSA_API __attribute__((visibility("default")))
#else
#define NSA_API
#endif

/* element types of Q/K/V/O */
#define NSA_DT_F32 0
#define NSA_DT_BF16 1
#define NSA_DT_F16 2

/* selector semantics */
#define NSA_SEL_SEQUENTIAL 0 /* select_topn_ranges        (decode, sequential prefill) */
#define NSA_SEL_BATCHED 1    /* select_topn_ranges_batched (batched prefill, training)  */

/* error codes */
#define NSA_OK 0
#define NSA_ERR_INVALID (-1)     /* bad argument / unsupported shape */
#define NSA_ERR_HIP (-2)         /* a HIP runtime call failed        */
#define NSA_ERR_WORKSPACE (-3)   /* workspace too small              */
#define NSA_ERR_NO_DEVICE (-4)   /* no gfx950 device                 */

NSA_API int nsa_hip_abi_version(void);
NSA_API const char *nsa_hip_last_error(void);
/* Measurement / A-B switches (kernel form, staging, mapping).  Each switch is seeded once per process from the environment
 * variable NSA_HIP_<NAME> and can be changed afterwards only through this call (nothing reads the environment on the launch
 * path).  name: "SEL_ROWS", "ATTN_MAP", "ATTN_STAGE", "BAND_STAGE", "DECODE_UNFUSED", "SEL_BLOCKS", "DECODE_WG", "SEL_ROWSUM", "DECODE_STENCIL", "SEL_FUSE", "SCORES_FORM", "SEL_FLAT", "SEL_KSPLIT", "DECODE_STOP" (TIMELINE builds
 * only), "DECODE_WAVES", "DECODE_SPLIT", "DECODE_STEP", "DECODE_TEAM_SPIN", "DECODE_WIDE", "SEL_KSPLIT_T1", "SEL_KSPLIT_T2", "SCORES_SELECT", "DECODE_BAND" (with or without the NSA_HIP_ prefix); value -1 = automatic where the switch has an automatic setting.  Results never depend on a switch beyond the
 * tolerances stated for the entry point. */
NSA_API int nsa_hip_set_tuning(const char *name, int value);
NSA_API int nsa_hip_get_tuning(const char *name, int *value);
/* 0 if device `dev` is a gfx950 part; fills cu_count / total memory (host pointers, nullable). */
NSA_API int nsa_hip_device_check(int dev, int *cu_count, size_t *hbm_bytes);

/* ---------------------------------------------------------------------------------------
 * Selection attention forward.
 *   Q [B,S,G,h,Dk]  O [B,S,G,h,Dv]   contiguous
 *   K [B,G,S_kv,Dk] V [B,G,S_kv,Dv]  innermost dim contiguous; the b/g/token strides are given in
 *                                    ELEMENTS so a preallocated cache [B,G,S_max,D] can be passed
 *                                    without a copy (NSA_KV layout, nsa/cache/kv_cache.py:8-30)
 *   ranges [B,S,G,n,2] int32 [start,end) token ranges.  Semantics = union of the ranges after
 *          clamping to [0,S_kv] (attention_kernels.py:721-732); end<=start entries are ignored;
 *          a row with no allowed token yields zeros (attention_kernels.py:734-749,769-771).
 *   lse    [B,S,G,h] fp32, nullable: log-sum-exp of the scaled logits (needed by the backward).
 *   scale  softmax scale; pass <=0 for the default Dk^-1/2.
 *   variant 0 = auto, 1 = generic VALU kernel, 2 = MFMA kernels (bf16/f16, Dk=Dv in {64,128}, h<=16): with many rows the block form
 *           (8 consecutive rows of one (b,g) per wave walk the union of their 64-key blocks; from 64k keys on as two key halves on
 *           different XCDs plus a merge launch) or the query-tile form, with few rows (decode) one workgroup per row or one row per
 *           wave with its tiles split over several waves.
 *   workspace: nsa_sel_attn_fwd_workspace() bytes, 16-byte aligned (may be 0).  Holds the partial records when few rows are split
 *           over KV and when a long context is split into key halves (288 B per (row, head): sized for S_kv = S); with a smaller or
 *           no workspace the call falls back to the forms that need none.
 * ------------------------------------------------------------------------------------- */
NSA_API size_t nsa_sel_attn_fwd_workspace(int B, int S, int G, int h, int Dk, int Dv, int n_ranges, int dtype);
/* the same for query rows that are the LAST S positions of a longer context (S_kv > S: a chunk of a chunked prefill): the key-split form splits
 * rows by their position S_kv - S + row, so it may need records where the S_kv = S query above sees none.  A launch whose workspace is too
 * small for the split form runs the unsplit walk (same results within one ulp of the output dtype, slower at long contexts). */
NSA_API size_t nsa_sel_attn_fwd_workspace_kv(int B, int S, int G, int h, int Dk, int Dv, int S_kv, int n_ranges, int dtype);
NSA_API int nsa_sel_attn_fwd(const void *Q, const void *K, const void *V, const int32_t *ranges, void *O,
                     float *lse, int B, int S, int G, int h, int Dk, int Dv, int S_kv, int n_ranges,
                     int64_t k_stride_b, int64_t k_stride_g, int64_t k_stride_s, int64_t v_stride_b,
                     int64_t v_stride_g, int64_t v_stride_s, int dtype, float scale, int variant,
                     void *workspace, size_t workspace_bytes, void *stream);

/* Opt-in PARITY MODE of the reference's default packed / gather executors (grouped_selection_attention_packed,
 * nsa/core/attention_kernels.py:273-388; grouped_selection_attention, :181-226): they call SDPA with is_causal=True and a single
 * query, so the query sees only the first gathered key and every head's output is V[b,g,start of the first non-empty range]
 * (slot order; ranges clamped to [0,S_kv] as everywhere here), zeros for a row without a range.  Not the semantics of the
 * selected branch -- kept so that outputs of the reference's default routing can be reproduced bit for bit. */
NSA_API int nsa_sel_attn_first_key_parity(const void *V, const int32_t *ranges, void *O, int B, int S, int G, int h, int Dv,
                                  int S_kv, int n_ranges, int64_t v_stride_b, int64_t v_stride_g, int64_t v_stride_s,
                                  int dtype, void *stream);

/* Opt-in PARITY MODE of NSAAttention._sdpa_over_ranges (nsa/core/nsa_attention.py:1779-1855), the reference's gather route of the
 * decode and sequential-prefill paths (what NSA_FORCE_PARITY=1 leaves them on, :704-708, :830, :1687): the union of the clamped ranges
 * is gathered in ascending token order and SDPA(is_causal=True) is called with the h heads in the query-length position, so head i
 * attends the first i+1 gathered tokens (top-left aligned causal mask).  Rows without a token give zeros.  h <= 16.  Not the
 * semantics of the selected branch -- kept so that outputs of the reference's forced-parity routing can be reproduced. */
NSA_API int nsa_sel_attn_head_causal_parity(const void *Q, const void *K, const void *V, const int32_t *ranges, void *O, int B, int S, int G,
                                    int h, int Dk, int Dv, int S_kv, int n_ranges, int64_t k_stride_b, int64_t k_stride_g,
                                    int64_t k_stride_s, int64_t v_stride_b, int64_t v_stride_g, int64_t v_stride_s, int dtype,
                                    float scale, void *stream);

/* Backward.  dO like O; dQ like Q (dtype); dK/dV are fp32 [B,G,S_kv,D] contiguous, fully written by the callee.
 * O and lse come from the forward.
 *   variant 0 auto, 1 generic (any dtype/shape: one wave per query row, dK/dV by fp32 atomics),
 *   2 MFMA (bf16/f16, Dk = Dv = 64, h <= 16): dQ query-major + dK/dV key-block-major, no atomics, reproducible.
 *   workspace: nsa_sel_attn_bwd_workspace() bytes (MFMA route: B*S*G*h floats for rowsum(dO*O) plus, when the query
 *   rows are split over workgroups, the per-split dK/dV partial sums that are added in fixed order; else 0). */
NSA_API size_t nsa_sel_attn_bwd_workspace(int B, int S, int G, int h, int Dk, int Dv, int S_kv, int dtype, int variant);
NSA_API int nsa_sel_attn_bwd(const void *Q, const void *K, const void *V, const int32_t *ranges, const void *O,
                     const float *lse, const void *dO, void *dQ, float *dK, float *dV, int B, int S,
                     int G, int h, int Dk, int Dv, int S_kv, int n_ranges, int64_t k_stride_b,
                     int64_t k_stride_g, int64_t k_stride_s, int64_t v_stride_b, int64_t v_stride_g,
                     int64_t v_stride_s, int dtype, float scale, int variant, void *workspace,
                     size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * Band attention forward: the sliding-window and the compressed branch (next row of the scope table after the selected
 * branch).  Query row t (absolute position t0 + t) attends the contiguous key interval [max(0, hi - w), hi) with
 *     hi(t) = (t0 + t + 1 >= a) ? min(S_kv, (t0 + t + 1 - a) / dd + c) : 0 ;
 *   sliding window (sliding_window_attention, nsa/core/attention_kernels.py:146-178): a = 0, dd = 1, c = 0, w = window;
 *   compressed (emission schedule num_cmp(t), attention_kernels.py:118-121): a = l, dd = d, c = 1, w = INT_MAX (K/V = K_cmp/V_cmp).
 * Layouts, strides, scale, lse and the empty-row rule (zeros, lse = -inf) as in nsa_sel_attn_fwd.  Prefill passes t0 = 0 and
 * all S rows; a decode step passes S = 1 and t0 = position of the new token.
 *   variant 0 = auto, 1 = generic VALU kernel (any dtype, D <= 256), 2 = MFMA kernel (bf16/f16, Dk = Dv = 64, h <= 16).
 * ------------------------------------------------------------------------------------- */
NSA_API size_t nsa_band_attn_fwd_workspace(int B, int S, int G, int h, int Dk, int Dv, int dtype);
NSA_API int nsa_band_attn_fwd(const void *Q, const void *K, const void *V, void *O, float *lse, int B, int S, int G, int h,
                      int Dk, int Dv, int S_kv, int64_t k_stride_b, int64_t k_stride_g, int64_t k_stride_s,
                      int64_t v_stride_b, int64_t v_stride_g, int64_t v_stride_s, int t0, int a, int dd, int c, int w,
                      int dtype, float scale, int variant, void *workspace, size_t workspace_bytes, void *stream);

/* Band attention backward (same interval rule and layouts as nsa_band_attn_fwd; O and lse from the forward; dQ in the activation
 * dtype, dK/dV fp32 [B,G,S_kv,D] fully written).  MFMA route (bf16/f16, Dk = Dv = 64): dQ by a dense 48-slot query-major kernel, dK/dV by
 * the key-block-major kernels of nsa_sel_attn_bwd fed with one [lo,hi) range per row; otherwise the generic selection backward. */
NSA_API size_t nsa_band_attn_bwd_workspace(int B, int S, int G, int h, int Dk, int Dv, int S_kv, int dtype, int variant);
NSA_API int nsa_band_attn_bwd(const void *Q, const void *K, const void *V, const void *O, const float *lse, const void *dO, void *dQ,
                      float *dK, float *dV, int B, int S, int G, int h, int Dk, int Dv, int S_kv, int64_t k_stride_b,
                      int64_t k_stride_g, int64_t k_stride_s, int64_t v_stride_b, int64_t v_stride_g, int64_t v_stride_s, int t0,
                      int a, int dd, int c, int w, int dtype, float scale, int variant, void *workspace, size_t workspace_bytes,
                      void *stream);

/* ---------------------------------------------------------------------------------------
 * Layer-level entry points: NSAAttention around the three branches (nsa/core/nsa_attention.py).  Plain-C descriptors:
 *   nsa_layer_desc  geometry + weights of one NSAAttention module (state-dict tensors, row-major [out,in] like nn.Linear;
 *                   W_qkv is the row-wise concatenation W_Q | W_K_sel | W_V_sel | W_K_win | W_V_win | W_K_cmp | W_V_cmp)
 *   nsa_kv_desc     the preallocated NSA_KV buffers (nsa/cache/kv_cache.py:8-26), each [B,G,S_max,D] contiguous,
 *                   K_cmp/V_cmp [B,G,n_cmp_max,D]
 * All tensors of one call share `dtype`.
 * ------------------------------------------------------------------------------------- */
typedef struct nsa_layer_desc {
    int dim, G, h, Dk, Dv, l, d, l_sel, n_sel, w, gate_hidden, dtype;
    float rope_base, rope_scale, gate_tau;
    const void *W_qkv;   /* [G*h*Dk + 3*G*Dk + 3*G*Dv, dim] */
    const void *W_out;   /* [dim, G*h*Dv] */
    const void *gate_w1; /* [gate_hidden, Dk] */
    const void *gate_b1; /* [gate_hidden] */
    const void *gate_w2; /* [3, gate_hidden] */
    const void *gate_b2; /* [3] */
} nsa_layer_desc;

typedef struct nsa_kv_desc {
    void *K_sel, *V_sel, *K_win, *V_win, *K_raw, *V_raw;
    void *K_cmp, *V_cmp;
    int B, S_max, n_cmp_max;
} nsa_kv_desc;

/* out[M,N] = epilogue(A[M,K] . W[N,K]^T) for few rows (decode projections).  epilogue 0: none, 1: silu, 2: + residual[M,N]. */
NSA_API int nsa_linear_small(const void *A, const void *W, void *out, int M, int N, int K, int dtype, int epilogue, const void *residual,
                     void *stream);
/* y[M,dim] = RMSNorm(x) * w, one wave per row, rounded where the reference's eager chain rounds (llama_block_nsa.py:10-19). */
NSA_API int nsa_rmsnorm_rows(const void *x, const void *w, void *y, int M, int dim, float eps, int dtype, void *stream);
/* Backward of nsa_rmsnorm_rows (training): dx[M,dim] and dw[dim] (activation dtype) from x, w, dy.  dim % 8 == 0, dim <= 4096, 16-byte
 * aligned tensors; dw is a fixed-order sum over workgroup partials (no atomics).  workspace: nsa_rmsnorm_rows_bwd_workspace bytes. */
NSA_API size_t nsa_rmsnorm_rows_bwd_workspace(int M, int dim);
NSA_API int nsa_rmsnorm_rows_bwd(const void *x, const void *w, const void *dy, void *dx, void *dw, int M, int dim, float eps, int dtype,
                         void *workspace, size_t workspace_bytes, void *stream);
/* RoPE (Q over the flattened head axis, K_sel/K_win per group; nsa_attention.py:552-572, 1002-1024) on a fused projection
 * proj [B,S,NQ+3GDk+3GDv] and append of the S tokens at cache position t0: Q_out [B,S,G,h,Dk]. */
NSA_API int nsa_rope_cache_append(const nsa_layer_desc *L, const nsa_kv_desc *kv, const void *proj, void *Q_out, int S, int t0,
                          void *stream);
/* Emit compressed tokens j0 <= j < j1: K_cmp[j] = mean_{i<l} RoPE(K_raw[j d + i]), V_cmp[j] = mean V_raw (compress_pool.py:9-38). */
NSA_API int nsa_cmp_pool_append(const nsa_layer_desc *L, const nsa_kv_desc *kv, int j0, int j1, void *stream);
/* Gate MLP + combine (nsa_attention.py:32-82, 85-124): O_out = sum_i gate_i O_i; gates_out [R,3] fp32 nullable; R = B*S*G rows. */
NSA_API int nsa_gate_combine(const nsa_layer_desc *L, const void *Q, const void *O_cmp, const void *O_sel, const void *O_win,
                     void *O_out, float *gates_out, int64_t R, void *stream);
/* Backward of the three layer kernels (training path; the attention branches have their own backward entry points).
 *   rope_cache_append_bwd: dQ [B,S,G,h,Dk] and the gradients of the S appended rows of each cache, [B,G,S,D] contiguous
 *                          (NULL = zero), -> dproj [B,S,NQ+3GDk+3GDv] (the rotation is undone on the gradient).
 *   cmp_pool_bwd:          dK_cmp/dV_cmp [B,G,n_cmp,D] -> dK_raw/dV_raw [B,G,S,D]  (raw rows 0..S-1, windows j d .. j d + l).
 *   gate_combine_bwd:      dO [R,h,Dv] -> dO_cmp/dO_sel/dO_win = gate_i dO and dgates [R,3] fp32 = sum O_i dO;
 *                          the gradient through the gate MLP (a [R,Dk] -> [R,3] network) is left to the caller. */
NSA_API int nsa_rope_cache_append_bwd(const nsa_layer_desc *L, int B, int S, int t0, const void *dQ, const void *dK_sel,
                              const void *dV_sel, const void *dK_win, const void *dV_win, const void *dK_raw,
                              const void *dV_raw, void *dproj, void *stream);
NSA_API int nsa_cmp_pool_bwd(const nsa_layer_desc *L, int B, int S, int n_cmp, const void *dK_cmp, const void *dV_cmp, void *dK_raw,
                     void *dV_raw, void *stream);
NSA_API int nsa_gate_combine_bwd(const nsa_layer_desc *L, const void *dO, const void *O_cmp, const void *O_sel, const void *O_win,
                         const float *gates, void *dO_cmp, void *dO_sel, void *dO_win, float *dgates, int64_t R, void *stream);
/* One decode step of a whole LlamaBlockNSA (nsa/model/llama_block_nsa.py:33-106: x + attn(norm1(x)), then + mlp(norm2(.))) in one
 * call: RMSNorm -> nsa_layer_decode_step with the residual added in the output projection's epilogue -> RMSNorm -> fc1 + silu ->
 * fc2 + residual.  x, y [B,dim]; the MLP weights are row-major [out,in] like nn.Linear. */
typedef struct nsa_block_desc {
    nsa_layer_desc attn;
    const void *norm1_w, *norm2_w; /* [dim] */
    const void *mlp_w1;            /* [mlp_hidden, dim] */
    const void *mlp_w2;            /* [dim, mlp_hidden] */
    int mlp_hidden;
    float norm_eps;
} nsa_block_desc;
NSA_API size_t nsa_block_decode_step_workspace(const nsa_block_desc *Bk, int B, int S_max);
NSA_API int nsa_block_decode_step(const nsa_block_desc *Bk, const nsa_kv_desc *kv, const void *x, void *y, int t, const int32_t *csc_ptr,
                          const int32_t *csc_rows, const float *csc_vals, int S_sel, int32_t *ranges_out, float *gates_out,
                          void *workspace, size_t workspace_bytes, void *stream);

/* One decode step of a whole TinyLM-style stack (scripts/train_showcase.py:30-110) in one call: embedding rows of tokens[B] ->
 * n_blocks x nsa_block_decode_step (all at position t, each with its own cache) -> final RMSNorm -> LM head -> logits [B,vocab]
 * (activation dtype) and, when next_tokens != NULL, their argmax.  blocks / kvs are host arrays of n_blocks descriptors. */
NSA_API size_t nsa_model_decode_step_workspace(const nsa_block_desc *blocks, int n_blocks, int B, int S_max);
NSA_API int nsa_model_decode_step(const nsa_block_desc *blocks, const nsa_kv_desc *kvs, int n_blocks, const int32_t *tokens,
                          const void *embed, const void *norm_f_w, const void *lm_head, int vocab, void *logits,
                          int32_t *next_tokens, int t, const int32_t *csc_ptr, const int32_t *csc_rows, const float *csc_vals,
                          int S_sel, void *workspace, size_t workspace_bytes, void *stream);

/* Prefill of the whole layer between the two big GEMMs, in one call (nsa_attention.py:978-1448 / 1521-1723): proj [B,S,NQ+3GDk+3GDv]
 * (= x @ W_qkv^T) -> RoPE + append of the S tokens into the EMPTY caches -> compressed-token pooling -> selection scores ->
 * top-n + selection attention -> sliding and compressed branches -> gates + combine -> O_mix [B,S,G*h*Dv] (input of the output
 * projection).  selector = NSA_SEL_BATCHED or NSA_SEL_SEQUENTIAL; ranges_out [B,S,G,W,2] with W = nsa_batched_ranges_width(S, ...)
 * resp. n_sel; gates_out [B,S,G,3] fp32 nullable.  csc_* / S_sel: Eq.9 map of the metadata for S tokens. */
NSA_API size_t nsa_layer_prefill_workspace(const nsa_layer_desc *L, int B, int S, int S_sel);
NSA_API int nsa_layer_prefill(const nsa_layer_desc *L, const nsa_kv_desc *kv, const void *proj, int S, int selector,
                      const int32_t *csc_ptr, const int32_t *csc_rows, const float *csc_vals, int S_sel, int32_t *ranges_out,
                      int out_width, void *O_mix, float *gates_out, void *workspace, size_t workspace_bytes, void *stream);
/* One decode step of the whole layer in one call (nsa_attention.py:509-830, decode branch): x [B,dim] is the new token at
 * position t (= tokens already cached); appends it to the caches, emits a compressed token when due, runs the three branches,
 * the gate and the output projection -> y [B,dim].  csc_* / S_sel: the Eq.9 map of the block metadata covering t
 * (nsa_build_block_meta_host).  ranges_out [B,G,n_sel,2] int32 and gates_out [B,G,3] fp32 are nullable monitors. */
NSA_API size_t nsa_layer_decode_step_workspace(const nsa_layer_desc *L, int B, int S_max);
NSA_API int nsa_layer_decode_step(const nsa_layer_desc *L, const nsa_kv_desc *kv, const void *x, void *y, int t,
                          const int32_t *csc_ptr, const int32_t *csc_rows, const float *csc_vals, int S_sel,
                          int32_t *ranges_out, float *gates_out, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * Eq.9 map in gather (CSC) form.  For selection block j the entries csc_ptr[j]..csc_ptr[j+1]
 * list (cmp row, weight) in ASCENDING cmp row -- the order the reference's CPU scatter_add
 * accumulates in -- so p_slc is bit-identical to the reference given an identical fp32 p_cmp.
 * nsa_build_block_meta_host fills both the reference's CSR/COO arrays and this CSC (host
 * pointers; call with NULL arrays to get the sizes).
 * ------------------------------------------------------------------------------------- */
NSA_API int nsa_block_counts(int seq_len, int l, int d, int l_sel, int *S_cmp, int *S_sel, int *nnz);
NSA_API int nsa_build_block_meta_host(int seq_len, int l, int d, int l_sel, int32_t *csr_indptr /*S_cmp+1*/,
                              int32_t *csr_indices /*nnz*/, float *csr_values /*nnz*/,
                              int32_t *csc_ptr /*S_sel+1*/, int32_t *csc_rows /*nnz*/,
                              float *csc_vals /*nnz*/);

/* p_cmp [R,h,S_cmp_cur] fp32 -> p_slc [R,h,S_sel] (nullable) and p_grp [R,S_sel] (Eq.10, heads
 * summed in ascending h).  CSC rows >= S_cmp_cur are dropped (selection_scorer.py:103-108). */
NSA_API int nsa_map_pcmp_to_pgrp(const float *p_cmp, int64_t R, int h, int S_cmp_cur, const int32_t *csc_ptr,
                         const int32_t *csc_rows, const float *csc_vals, int S_sel, float *p_slc,
                         float *p_grp, void *stream);

/* p_cmp = softmax over ALL S_cmp columns of Q K_cmp^T * scale, fp32 out [B,S,G,h,S_cmp].
 * Q [B,S,G,h,Dk] contiguous; K_cmp [B,G,S_cmp,Dk] with element strides. */
NSA_API int nsa_pcmp_all(const void *Q, const void *K_cmp, float *p_cmp, int B, int S, int G, int h, int Dk,
                 int S_cmp, int64_t kc_stride_b, int64_t kc_stride_g, int64_t kc_stride_s, int dtype,
                 float scale, void *stream);

/* Fused scorer: Q,K_cmp -> p_grp [B,S,G,S_sel] fp32 without materialising p_cmp / p_slc.
 *   l, d, l_sel: the block geometry the CSC was built for.  For l = 2d, l' = 4d (the reference default
 *   32/16/64) and bf16/f16 inputs with Dk in {64,128}, h <= 16, a single MFMA kernel evaluates Eq.9 in
 *   closed form; every other case runs the query-chunked generic path and needs the workspace.
 *   causal_skip != 0: entries p_grp[b,t,g,j] of blocks the selector can never pick at t
 *   ((j+1) l' > t+1, masked to -inf by both selectors) need not be computed: such an entry is returned as 0 -- or, on the MFMA route, as
 *   its full computed value where the workgroup of the query (64 / 32 consecutive queries share a sweep that stops at the LAST block any
 *   of them can read) computed the block for a later query of the group.  causal_skip == 2 additionally leaves the entries no query of the
 *   workgroup can read UNWRITTEN (saves the zero fill of p_grp -- 512 MiB at S = 64k).  Either way only entries with (j+1) l' <= t+1 are
 *   defined values of the reference: a caller that sums or ranks p_grp itself must apply that mask (both selectors here do).
 *   Few query rows (B*S*G <= 1024: decode) run a decode-shaped pair of kernels that spreads the K_cmp sweep of a
 *   row over many workgroups (any dtype / geometry); it needs B*S*G*h*S_cmp floats of workspace.
 *   variant: 0 auto, 1 generic, 2 MFMA (prefill), 3 decode-shaped.
 *   workspace: nsa_sel_scores_workspace(same shape/geometry/dtype/variant) bytes (0 for the MFMA route). */
NSA_API size_t nsa_sel_scores_workspace(int B, int S, int G, int h, int Dk, int S_cmp, int S_sel, int l, int d, int l_sel,
                                int dtype, int variant);
NSA_API int nsa_sel_scores(const void *Q, const void *K_cmp, float *p_grp, int B, int S, int G, int h, int Dk,
                   int S_cmp, int64_t kc_stride_b, int64_t kc_stride_g, int64_t kc_stride_s,
                   const int32_t *csc_ptr, const int32_t *csc_rows, const float *csc_vals, int S_sel,
                   int l, int d, int l_sel, int causal_skip, int variant /* 0 auto, 1 generic, 2 MFMA, 3 decode */,
                   int dtype, float scale, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * Deterministic top-n + forced blocks + range merge.
 *   p_grp [R,S_sel] fp32 with R = B*S*G rows ordered (b,s,g); row r sits at token
 *   t = t0 + (r / G) % S  (decode: S = 1, t0 = current position), or t_rows[r] if non-NULL.
 *   mode NSA_SEL_SEQUENTIAL: out [R,n_top,2]; mode NSA_SEL_BATCHED: out [R,K,2] with
 *   K = nsa_batched_ranges_width(...); out_width must equal that width.
 *   Ranking key = fp32(p) - fp32(idx)*1e-8f (unfused), descending, index ascending on ties.
 *   Sequential-mode note: where the reference's topk would have to pick -inf entries (fewer valid
 *   candidates than n_top - 3) it emits inverted garbage ranges; this ABI emits [0,0] instead.
 * ------------------------------------------------------------------------------------- */
NSA_API int nsa_batched_ranges_width(int S, int S_sel, int l_sel, int n_top, int force_init, int force_local);
NSA_API int nsa_select_topn_ranges(const float *p_grp, int64_t R, int S, int G, int t0, const int32_t *t_rows,
                           int S_sel, int l_sel, int n_top, int force_init, int force_local, int mode,
                           int S_total /* batched: the S the forced-column rule is evaluated for */,
                           int32_t *ranges_out, int out_width, void *stream);

/* ---------------------------------------------------------------------------------------
 * Scores + top-n selection in one call (prefill; round 4): the arguments of nsa_sel_scores (variant 0) followed by those of
 * nsa_select_topn_ranges (rows in the [B,S,G] order of Q, token of row (b,s,g) = t0 + s).  Replaces compute_pcmp_all ->
 * map_pcmp_to_pslc_batched -> sum(dim=3) -> select_topn_ranges[_batched] (nsa/core/nsa_attention.py:1088-1108, 1566-1576;
 * nsa/core/selection_scorer.py:42-61, 89-116, 124-249, 255-362).  Where the 32x32x16 MFMA scorer applies (h = 6, Dk = 64, default block
 * geometry, bf16 / f16, S_sel <= 1024) and the context is long enough for it to pay (S_cmp >= 3072; tuning switch "SCORES_SELECT") a workgroup
 * selects the ranges of its 64 query rows right behind its second sweep, with the select kernel's own row function: ONE launch, the scores are read back out of L2 (no HBM read of p_grp), the selector's scalar work runs beside
 * the scorer's matrix / vector work.  Everywhere else the scorer and the select kernel are launched back to back.  p_grp is still written
 * (same contract as nsa_sel_scores with the given causal_skip); ranges_out is bit-identical to the two separate calls either way.
 * workspace: the larger of nsa_sel_scores_workspace(..., variant 0) and (..., variant 1) bytes (rows that are not 16-byte aligned take the
 * generic scorer).
 * ------------------------------------------------------------------------------------- */
NSA_API int nsa_sel_scores_select(const void *Q, const void *K_cmp, float *p_grp, int B, int S, int G, int h, int Dk, int S_cmp,
                          int64_t kc_stride_b, int64_t kc_stride_g, int64_t kc_stride_s, const int32_t *csc_ptr, const int32_t *csc_rows,
                          const float *csc_vals, int S_sel, int l, int d, int l_sel, int causal_skip, int dtype, float scale, int t0,
                          int n_top, int force_init, int force_local, int mode, int S_total, int32_t *ranges_out, int out_width,
                          void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * Top-n selection + selection attention in one call (prefill): the arguments of nsa_select_topn_ranges followed by those of
 * nsa_sel_attn_fwd; ranges_out [B,S,G,out_width,2] is produced AND consumed.  The select kernel and the attention kernel are
 * launched back to back on the stream (the default since round 2: the selector is faster as its own launch); with the tuning
 * switch SEL_FUSE = 1 the selection runs inside the attention kernel on the MFMA route (one launch).  Results are those of the
 * two separate calls, bit for bit, either way.
 * ------------------------------------------------------------------------------------- */
NSA_API int nsa_sel_select_attn_fwd(const float *p_grp, int t0, const int32_t *t_rows, int S_sel, int l_sel, int n_top, int force_init,
                            int force_local, int mode, int S_total, int32_t *ranges_out, int out_width, const void *Q, const void *K,
                            const void *V, void *O, float *lse, int B, int S, int G, int h, int Dk, int Dv, int S_kv,
                            int64_t k_stride_b, int64_t k_stride_g, int64_t k_stride_s, int64_t v_stride_b, int64_t v_stride_g,
                            int64_t v_stride_s, int dtype, float scale, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * One decode step of the selected branch in ONE call (host launch overhead matters at batch 1):
 *   Q [B,1,G,h,Dk] -> decode-shaped scores -> sequential top-n ranges at token t_token -> selection attention over
 *   K/V[:, :, :S_kv].  Replaces the calls of the decode branch of NSAAttention.forward
 *   (nsa/core/nsa_attention.py:651 compute_pcmp_all, :658 map_pcmp_to_pslc_batched, :670 head sum,
 *   :672 select_topn_ranges, :704-830 selection executor).
 *   ranges_out [B,G,n_top,2] int32, O [B,1,G,h,Dv]; workspace: nsa_sel_decode_step_workspace() bytes, 16-B aligned.
 *   Kernel form (one launch wherever the default block geometry, bf16 / f16 and Dk = Dv = 64 allow): chosen from (B*G, S_cmp) -- logits
 *   in registers for rows of up to 32 chunks of 64 compressed rows, a team of workgroups per row for longer rows while B*G teams fit the
 *   chip, and for more rows than that (B >= 128 at a 64k context) the one-pass form, whose group scores carry one more rounding (<= 2 ulp)
 *   than the other forms': its ranges are theirs wherever the (n_top - 3)-th and the next ranking key differ by more than that (tuning
 *   switch "DECODE_WIDE": 0 = never, 1 = the exact one-workgroup form instead).  Every form is bitwise reproducible run to run.
 * ------------------------------------------------------------------------------------- */
NSA_API size_t nsa_sel_decode_step_workspace(int B, int G, int h, int Dk, int Dv, int S_cmp, int S_sel, int n_top, int dtype);
NSA_API int nsa_sel_decode_step(const void *Q, const void *K_cmp, const void *K, const void *V, const int32_t *csc_ptr,
                        const int32_t *csc_rows, const float *csc_vals, int32_t *ranges_out, void *O, int B, int G, int h,
                        int Dk, int Dv, int S_cmp, int S_sel, int S_kv, int l, int d, int l_sel, int n_top, int t_token,
                        int64_t kc_stride_b, int64_t kc_stride_g, int64_t kc_stride_s, int64_t k_stride_b,
                        int64_t k_stride_g, int64_t k_stride_s, int64_t v_stride_b, int64_t v_stride_g,
                        int64_t v_stride_s, int dtype, float scale, void *workspace, size_t workspace_bytes, void *stream);

/* indices [R,K] int32 ascending with -1 padding -> ranges [R,K,2]; clamp end to t+1. */
NSA_API int nsa_indices_to_ranges_v2(const int32_t *indices, int64_t R, int S, int G, int t0, int K, int S_sel,
                             int l_sel, int32_t *ranges_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NSA_SEL_HIP_H */
