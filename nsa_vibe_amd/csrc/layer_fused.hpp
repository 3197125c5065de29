// Argument blocks + launchers of the layer-level kernels (layer_fused.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nsa {

struct RopeAppendParams {
    const void *proj;  // [B*S, NQ + 3 G Dk + 3 G Dv]: Q | K_sel | V_sel | K_win | V_win | K_raw | V_raw
    void *Q_out;       // [B*S, NQ]
    void *cache[6];    // K_sel, V_sel, K_win, V_win, K_raw, V_raw: [B,G,S_max,D] contiguous
    int B, S, G, h, Dk, Dv, S_max, t0;
    float rope_base, inv_scale;
};
struct CmpPoolParams {
    const void *K_raw, *V_raw;  // [nbg, S_max, D]
    void *K_cmp, *V_cmp;        // [nbg, n_cmp_max, D]
    int nbg, S_max, n_cmp_max, Dk, Dv, l, d, j0, j1;
    float rope_base, inv_scale;
};
struct GateCombineParams {
    const void *Q, *O_cmp, *O_sel, *O_win;
    void *O_out;
    float *gates_out;  // [R,3] or null
    const void *w1, *b1, *w2, *b2;
    int64_t R;
    int h, Dk, Dv, Hd;
    float tau;
};
struct DecodeFinishParams {
    const void *Q;
    const float *part[3];  // cmp, sel, win: split-KV partial records (ns > 1) ...
    const void *O[3];      // ... or the final branch output (ns == 1)
    int ns[3];
    void *O_out;
    float *gates_out;
    const void *w1, *b1, *w2, *b2;
    int64_t R;
    int h, Dk, Dv, Hd;
    float tau;
};
bool qkv_can_fold_norm(const RopeAppendParams &P, const void *X, const void *W, int K, int dtype);
int launch_qkv_rope_append(const RopeAppendParams &P, const void *X, const void *W, int K, int dtype, hipStream_t st, const void *norm_w = nullptr,
                           float norm_eps = 0.f);
bool linear_small_can_fold_norm(int dtype, int M, int N, int K, const void *A, const void *W);
int launch_linear_small_norm(const void *A, const void *W, void *out, int M, int N, int K, int dtype, int epi, const void *res, const void *norm_w,
                             float eps, hipStream_t st);
int launch_decode_finish(const DecodeFinishParams &P, int dtype, hipStream_t st);
// output projection of a decode step fed by the three branch outputs and the row gates (few rows: M <= 8): out = mix(O_cmp, O_sel, O_win) . W^T
bool linear_small_mix_supported(int dtype, int M, int N, int K, int G, const void *Oc, const void *Os, const void *Ow, const void *W);
int launch_linear_small_mix(const void *Oc, const void *Os, const void *Ow, const float *gates, const void *W, void *out, int M, int N, int K, int G,
                            int dtype, int epi, const void *res, hipStream_t st);
int launch_linear_small(const void *A, const void *W, void *out, int M, int N, int K, int dtype, hipStream_t st);
// epi: 0 none, 1 silu, 2 + res[M,N]
int launch_linear_small_epi(const void *A, const void *W, void *out, int M, int N, int K, int dtype, int epi, const void *res, hipStream_t st);
int launch_embed_rows(const int32_t *tokens, const void *embed, void *x, int B, int dim, int vocab, int dtype, hipStream_t st);
size_t argmax_rows_workspace(int B, int vocab);
int launch_argmax_rows(const void *logits, int32_t *next, int B, int vocab, int dtype, void *ws, hipStream_t st);
int launch_rmsnorm_rows(const void *x, const void *w, void *y, int M, int dim, float eps, int dtype, hipStream_t st);
size_t rmsnorm_rows_bwd_workspace(int M, int dim);
int launch_rmsnorm_rows_bwd(const void *x, const void *w, const void *dy, void *dx, void *dw, int M, int dim, float eps, int dtype,
                            void *workspace, size_t workspace_bytes, hipStream_t st);
int launch_rope_cache_append(const RopeAppendParams &P, int dtype, hipStream_t st);
int launch_cmp_pool(const CmpPoolParams &P, int dtype, hipStream_t st);
int launch_rope_cache_append_bwd(const RopeAppendParams &P, int dtype, hipStream_t st);
int launch_cmp_pool_bwd(const CmpPoolParams &P, const void *dKc, const void *dVc, void *dKr, void *dVr, int S, int n_cmp, int dtype,
                        hipStream_t st);
int launch_gate_combine(const GateCombineParams &P, int dtype, hipStream_t st);
int launch_gate_combine_bwd(const GateCombineParams &P, const void *dO, const float *gates, void *dOc, void *dOs, void *dOw, float *dgates,
                            int dtype, hipStream_t st);

}  // namespace nsa
