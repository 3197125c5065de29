// Internal forms of a few C-ABI entry points: same arguments plus the split-KV hand-over used by the fused decode step
// (defer != 0: the partial records stay in the workspace, *ns_used tells how many splits were written; 1 = O is final).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <hip/hip_runtime.h>

namespace nsa {

int sel_attn_fwd_impl(const void *Q, const void *K, const void *V, const int32_t *ranges, void *O, float *lse, int B, int S, int G, int h,
                      int Dk, int Dv, int S_kv, int n_ranges, int64_t ksb, int64_t ksg, int64_t kss, int64_t vsb, int64_t vsg, int64_t vss,
                      int dtype, float scale, int variant, void *workspace, size_t workspace_bytes, void *stream, int defer, int *ns_used);
int band_attn_fwd_impl(const void *Q, const void *K, const void *V, void *O, float *lse, int B, int S, int G, int h, int Dk, int Dv, int S_kv,
                       int64_t ksb, int64_t ksg, int64_t kss, int64_t vsb, int64_t vsg, int64_t vss, int t0, int a, int dd, int c, int w,
                       int dtype, float scale, int variant, void *workspace, size_t workspace_bytes, void *stream, int defer, int *ns_used);
struct DecBandPair;  // sel_attn_params.hpp
// band / band_taken: the layer step's sliding + compressed branches (split form, deferred combine); *band_taken = 1 when the selected branch ran
// as the one-launch decode step and carried them on its launch (otherwise the caller launches them itself)
int sel_decode_step_impl(const void *Q, const void *K_cmp, const void *K, const void *V, const int32_t *csc_ptr, const int32_t *csc_rows,
                         const float *csc_vals, int32_t *ranges_out, void *O, int B, int G, int h, int Dk, int Dv, int S_cmp, int S_sel,
                         int S_kv, int l, int d, int l_sel, int n_top, int t_token, int64_t kcb, int64_t kcg, int64_t kcs, int64_t ksb,
                         int64_t ksg, int64_t kss, int64_t vsb, int64_t vsg, int64_t vss, int dtype, float scale, void *workspace,
                         size_t workspace_bytes, void *stream, int defer, int *ns_used, float **part_used, const DecBandPair *band = nullptr,
                         int *band_taken = nullptr);

bool decode_score_select_supported(int dtype, int h, int Dk, int S_cmp, int S_sel, int64_t csb, int64_t csg, int64_t css, const void *Q,
                                   const void *Kc, int64_t rows);
struct DecAttnArgs;  // sel_attn_decode.hpp: non-null = the row's selection attention runs in the same launch
int launch_decode_score_select(const void *Q, const void *Kc, int B, int G, int h, int Dk, int S_cmp, int64_t csb, int64_t csg, int64_t css,
                               const int32_t *csc_ptr, const int32_t *csc_rows, const float *csc_vals, int S_sel, int l_sel, int n_top,
                               int t_token, int dtype, float scale, int32_t *ranges_out, hipStream_t st, const DecAttnArgs *attend,
                               int stencil);
// the one-launch decode step (sel_decode_fused.hip): default block geometry, bf16 / f16, Dk = Dv = 64
bool decode_step_supported(int64_t R, int dtype, int h, int Dk, int Dv, int S_cmp, int S_sel, int S_kv, int l, int d, int l_sel, int n_top, int t_token,
                           int64_t kcb, int64_t kcg, int64_t kcs, int64_t ksb, int64_t ksg, int64_t kss, int64_t vsb, int64_t vsg, int64_t vss,
                           const void *Q, const void *Kc, const void *K, const void *V);
size_t decode_step_workspace(int64_t R, int h, int S_cmp);
int launch_decode_step(const void *Q, const void *Kc, const void *K, const void *V, void *O, int32_t *ranges_out, int B, int G, int h, int S_cmp,
                       int S_sel, int S_kv, int n_top, int t_token, int64_t kcb, int64_t kcg, int64_t kcs, int64_t ksb, int64_t ksg, int64_t kss,
                       int64_t vsb, int64_t vsg, int64_t vss, int dtype, float scale, void *ws, size_t ws_bytes, hipStream_t st,
                       const DecBandPair *band = nullptr);

}  // namespace nsa
