// C ABI of the layer-level entry points (include/nsa_sel_hip.h, "Layer-level entry points").
#include <cmath>

#include "nsa_common.hpp"
#include "layer_fused.hpp"
#include "nsa_internal.hpp"
#include "sel_attn_params.hpp"

using namespace nsa;

static bool dt_ok(int dt) { return dt == NSA_DT_F32 || dt == NSA_DT_BF16 || dt == NSA_DT_F16; }
static size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }
static size_t esize(int dt) { return dt == NSA_DT_F32 ? 4 : 2; }

static int check_layer(const nsa_layer_desc *L, const char *who) {
    NSA_CHECK_ARG(L, "%s: null layer descriptor", who);
    NSA_CHECK_ARG(dt_ok(L->dtype), "%s: unknown dtype %d", who, L->dtype);
    NSA_CHECK_ARG(L->dim >= 1 && L->G >= 1 && L->h >= 1 && L->Dk >= 2 && L->Dv >= 1 && L->Dk % 2 == 0 && L->Dv % 2 == 0,
                  "%s: bad geometry (Dk, Dv must be even)", who);
    NSA_CHECK_ARG(L->l >= 1 && L->d >= 1 && L->l_sel >= 1 && L->n_sel >= 1 && L->w >= 0, "%s: bad block parameters", who);
    return NSA_OK;
}
static int check_kv(const nsa_kv_desc *kv, const char *who) {
    NSA_CHECK_ARG(kv && kv->K_sel && kv->V_sel && kv->K_win && kv->V_win && kv->K_raw && kv->V_raw && kv->K_cmp && kv->V_cmp,
                  "%s: null cache pointer", who);
    NSA_CHECK_ARG(kv->B >= 1 && kv->S_max >= 1 && kv->n_cmp_max >= 1, "%s: bad cache sizes", who);
    return NSA_OK;
}

extern "C" {

int nsa_linear_small(const void *A, const void *W, void *out, int M, int N, int K, int dtype, int epilogue, const void *residual, void *stream) {
    NSA_CHECK_ARG(dt_ok(dtype), "linear_small: unknown dtype %d", dtype);
    NSA_CHECK_ARG(M >= 0 && N >= 0 && K >= 1, "linear_small: bad sizes");
    NSA_CHECK_ARG(epilogue >= 0 && epilogue <= 2 && (epilogue != 2 || residual), "linear_small: bad epilogue");
    if (M == 0 || N == 0) return NSA_OK;
    NSA_CHECK_ARG(A && W && out, "linear_small: null pointer");
    return launch_linear_small_epi(A, W, out, M, N, K, dtype, epilogue, residual, (hipStream_t)stream);
}

int nsa_rmsnorm_rows(const void *x, const void *w, void *y, int M, int dim, float eps, int dtype, void *stream) {
    NSA_CHECK_ARG(dt_ok(dtype), "rmsnorm_rows: unknown dtype %d", dtype);
    NSA_CHECK_ARG(M >= 0 && dim >= 1, "rmsnorm_rows: bad sizes");
    if (M == 0) return NSA_OK;
    NSA_CHECK_ARG(x && w && y, "rmsnorm_rows: null pointer");
    return launch_rmsnorm_rows(x, w, y, M, dim, eps, dtype, (hipStream_t)stream);
}

size_t nsa_rmsnorm_rows_bwd_workspace(int M, int dim) { return M > 0 && dim > 0 ? rmsnorm_rows_bwd_workspace(M, dim) : 0; }

int nsa_rmsnorm_rows_bwd(const void *x, const void *w, const void *dy, void *dx, void *dw, int M, int dim, float eps, int dtype,
                         void *workspace, size_t workspace_bytes, void *stream) {
    NSA_CHECK_ARG(dt_ok(dtype), "rmsnorm_rows_bwd: unknown dtype %d", dtype);
    NSA_CHECK_ARG(M >= 1 && dim >= 1, "rmsnorm_rows_bwd: bad sizes");
    NSA_CHECK_ARG(x && w && dy && dx && dw, "rmsnorm_rows_bwd: null pointer");
    return launch_rmsnorm_rows_bwd(x, w, dy, dx, dw, M, dim, eps, dtype, workspace, workspace_bytes, (hipStream_t)stream);
}

int nsa_rope_cache_append(const nsa_layer_desc *L, const nsa_kv_desc *kv, const void *proj, void *Q_out, int S, int t0,
                          void *stream) {
    if (int rc = check_layer(L, "rope_cache_append")) return rc;
    if (int rc = check_kv(kv, "rope_cache_append")) return rc;
    NSA_CHECK_ARG(S >= 0 && t0 >= 0 && t0 + S <= kv->S_max, "rope_cache_append: tokens [%d,%d) exceed the cache capacity %d", t0, t0 + S,
                  kv->S_max);
    if (S == 0) return NSA_OK;
    NSA_CHECK_ARG(proj && Q_out, "rope_cache_append: null pointer");
    RopeAppendParams P{};
    P.proj = proj;
    P.Q_out = Q_out;
    P.cache[0] = kv->K_sel; P.cache[1] = kv->V_sel; P.cache[2] = kv->K_win; P.cache[3] = kv->V_win; P.cache[4] = kv->K_raw; P.cache[5] = kv->V_raw;
    P.B = kv->B; P.S = S; P.G = L->G; P.h = L->h; P.Dk = L->Dk; P.Dv = L->Dv; P.S_max = kv->S_max; P.t0 = t0;
    P.rope_base = L->rope_base > 0.f ? L->rope_base : 10000.0f;
    P.inv_scale = 1.0f / (L->rope_scale > 0.f ? L->rope_scale : 1.0f);
    return launch_rope_cache_append(P, L->dtype, (hipStream_t)stream);
}

int nsa_cmp_pool_append(const nsa_layer_desc *L, const nsa_kv_desc *kv, int j0, int j1, void *stream) {
    if (int rc = check_layer(L, "cmp_pool_append")) return rc;
    if (int rc = check_kv(kv, "cmp_pool_append")) return rc;
    NSA_CHECK_ARG(j0 >= 0 && j1 >= j0 && j1 <= kv->n_cmp_max, "cmp_pool_append: tokens [%d,%d) exceed n_cmp_max %d", j0, j1, kv->n_cmp_max);
    NSA_CHECK_ARG(j1 == j0 || (int64_t)(j1 - 1) * L->d + L->l <= kv->S_max, "cmp_pool_append: window past the cache capacity");
    CmpPoolParams P{};
    P.K_raw = kv->K_raw; P.V_raw = kv->V_raw; P.K_cmp = kv->K_cmp; P.V_cmp = kv->V_cmp;
    P.nbg = kv->B * L->G; P.S_max = kv->S_max; P.n_cmp_max = kv->n_cmp_max; P.Dk = L->Dk; P.Dv = L->Dv; P.l = L->l; P.d = L->d;
    P.j0 = j0; P.j1 = j1;
    P.rope_base = L->rope_base > 0.f ? L->rope_base : 10000.0f;
    P.inv_scale = 1.0f;  // the reference pools apply_rope(K_raw, pos) WITHOUT the NSA_ROPE_SCALE position scaling (compress_pool.py:20)
    return launch_cmp_pool(P, L->dtype, (hipStream_t)stream);
}

int nsa_gate_combine(const nsa_layer_desc *L, const void *Q, const void *O_cmp, const void *O_sel, const void *O_win, void *O_out,
                     float *gates_out, int64_t R, void *stream) {
    if (int rc = check_layer(L, "gate_combine")) return rc;
    NSA_CHECK_ARG(R >= 0, "gate_combine: negative size");
    if (R == 0) return NSA_OK;
    NSA_CHECK_ARG(Q && O_cmp && O_sel && O_win && O_out && L->gate_w1 && L->gate_b1 && L->gate_w2 && L->gate_b2, "gate_combine: null pointer");
    GateCombineParams P{};
    P.Q = Q; P.O_cmp = O_cmp; P.O_sel = O_sel; P.O_win = O_win; P.O_out = O_out; P.gates_out = gates_out;
    P.w1 = L->gate_w1; P.b1 = L->gate_b1; P.w2 = L->gate_w2; P.b2 = L->gate_b2;
    P.R = R; P.h = L->h; P.Dk = L->Dk; P.Dv = L->Dv; P.Hd = L->gate_hidden; P.tau = L->gate_tau;
    return launch_gate_combine(P, L->dtype, (hipStream_t)stream);
}

int nsa_rope_cache_append_bwd(const nsa_layer_desc *L, int B, int S, int t0, const void *dQ, const void *dK_sel, const void *dV_sel,
                              const void *dK_win, const void *dV_win, const void *dK_raw, const void *dV_raw, void *dproj, void *stream) {
    if (int rc = check_layer(L, "rope_cache_append_bwd")) return rc;
    NSA_CHECK_ARG(B >= 0 && S >= 0 && t0 >= 0, "rope_cache_append_bwd: negative size");
    if (B == 0 || S == 0) return NSA_OK;
    NSA_CHECK_ARG(dQ && dproj, "rope_cache_append_bwd: null pointer");
    RopeAppendParams P{};
    P.proj = dproj;              // written
    P.Q_out = (void *)dQ;        // read
    P.cache[0] = (void *)dK_sel; P.cache[1] = (void *)dV_sel; P.cache[2] = (void *)dK_win; P.cache[3] = (void *)dV_win;
    P.cache[4] = (void *)dK_raw; P.cache[5] = (void *)dV_raw;
    P.B = B; P.S = S; P.G = L->G; P.h = L->h; P.Dk = L->Dk; P.Dv = L->Dv; P.S_max = S; P.t0 = t0;
    P.rope_base = L->rope_base > 0.f ? L->rope_base : 10000.0f;
    P.inv_scale = 1.0f / (L->rope_scale > 0.f ? L->rope_scale : 1.0f);
    return launch_rope_cache_append_bwd(P, L->dtype, (hipStream_t)stream);
}

int nsa_cmp_pool_bwd(const nsa_layer_desc *L, int B, int S, int n_cmp, const void *dK_cmp, const void *dV_cmp, void *dK_raw, void *dV_raw,
                     void *stream) {
    if (int rc = check_layer(L, "cmp_pool_bwd")) return rc;
    NSA_CHECK_ARG(B >= 0 && S >= 0 && n_cmp >= 0, "cmp_pool_bwd: negative size");
    if (B == 0 || S == 0) return NSA_OK;
    NSA_CHECK_ARG(dK_raw && dV_raw && ((dK_cmp && dV_cmp) || n_cmp == 0), "cmp_pool_bwd: null pointer");
    NSA_CHECK_ARG(n_cmp == 0 || (int64_t)(n_cmp - 1) * L->d + L->l <= S, "cmp_pool_bwd: windows reach past S");
    CmpPoolParams P{};
    P.nbg = B * L->G; P.Dk = L->Dk; P.Dv = L->Dv; P.l = L->l; P.d = L->d;
    P.rope_base = L->rope_base > 0.f ? L->rope_base : 10000.0f;
    P.inv_scale = 1.0f;  // as the forward: no position scaling inside the pooled keys (compress_pool.py:20)
    return launch_cmp_pool_bwd(P, dK_cmp, dV_cmp, dK_raw, dV_raw, S, n_cmp, L->dtype, (hipStream_t)stream);
}

int nsa_gate_combine_bwd(const nsa_layer_desc *L, const void *dO, const void *O_cmp, const void *O_sel, const void *O_win, const float *gates,
                         void *dO_cmp, void *dO_sel, void *dO_win, float *dgates, int64_t R, void *stream) {
    if (int rc = check_layer(L, "gate_combine_bwd")) return rc;
    NSA_CHECK_ARG(R >= 0, "gate_combine_bwd: negative size");
    if (R == 0) return NSA_OK;
    NSA_CHECK_ARG(dO && O_cmp && O_sel && O_win && gates && dO_cmp && dO_sel && dO_win && dgates, "gate_combine_bwd: null pointer");
    GateCombineParams P{};
    P.O_cmp = O_cmp; P.O_sel = O_sel; P.O_win = O_win;
    P.R = R; P.h = L->h; P.Dk = L->Dk; P.Dv = L->Dv;
    return launch_gate_combine_bwd(P, dO, gates, dO_cmp, dO_sel, dO_win, dgates, L->dtype, (hipStream_t)stream);
}

// prefill workspace: Q | p_grp | O_cmp | O_sel | O_win | scorer scratch | attention scratch | band scratch
struct PrefillWs {
    size_t q, pgrp, ocmp, osel, owin, sc, att, band, total, sc_bytes, att_bytes, band_bytes;
};
static PrefillWs prefill_ws(const nsa_layer_desc *L, int B, int S, int S_sel) {
    PrefillWs w;
    const size_t e = esize(L->dtype);
    const size_t NQ = (size_t)L->G * L->h * L->Dk, NO = (size_t)L->G * L->h * L->Dv;
    const int n_cmp = S < L->l ? 0 : (S - L->l) / L->d + 1;
    size_t o = 0;
    w.q = o; o += up256((size_t)B * S * NQ * e);
    w.pgrp = o; o += up256(sizeof(float) * (size_t)B * S * L->G * (size_t)(S_sel > 0 ? S_sel : 1));
    w.ocmp = o; o += up256((size_t)B * S * NO * e);
    w.osel = o; o += up256((size_t)B * S * NO * e);
    w.owin = o; o += up256((size_t)B * S * NO * e);
    w.sc_bytes = nsa_sel_scores_workspace(B, S, L->G, L->h, L->Dk, n_cmp, S_sel, L->l, L->d, L->l_sel, L->dtype, 0);
    const size_t sc1 = nsa_sel_scores_workspace(B, S, L->G, L->h, L->Dk, n_cmp, S_sel, L->l, L->d, L->l_sel, L->dtype, 1);
    if (sc1 > w.sc_bytes) w.sc_bytes = sc1;  // the generic route may be taken for unaligned inputs
    w.sc = o; o += up256(w.sc_bytes);
    w.att_bytes = nsa_sel_attn_fwd_workspace(B, S, L->G, L->h, L->Dk, L->Dv, L->n_sel, L->dtype);
    w.att = o; o += up256(w.att_bytes);
    w.band_bytes = nsa_band_attn_fwd_workspace(B, S, L->G, L->h, L->Dk, L->Dv, L->dtype);
    w.band = o; o += up256(w.band_bytes);
    w.total = o;
    return w;
}

size_t nsa_layer_prefill_workspace(const nsa_layer_desc *L, int B, int S, int S_sel) {
    if (!L || !dt_ok(L->dtype) || B < 1 || S < 1) return 0;
    return prefill_ws(L, B, S, S_sel).total;
}

int nsa_layer_prefill(const nsa_layer_desc *L, const nsa_kv_desc *kv, const void *proj, int S, int selector, const int32_t *csc_ptr,
                      const int32_t *csc_rows, const float *csc_vals, int S_sel, int32_t *ranges_out, int out_width, void *O_mix,
                      float *gates_out, void *workspace, size_t workspace_bytes, void *stream) {
    if (int rc = check_layer(L, "layer_prefill")) return rc;
    if (int rc = check_kv(kv, "layer_prefill")) return rc;
    NSA_CHECK_ARG(proj && O_mix && ranges_out, "layer_prefill: null pointer");
    NSA_CHECK_ARG(S >= 1 && S <= kv->S_max, "layer_prefill: %d tokens exceed the cache capacity %d", S, kv->S_max);
    NSA_CHECK_ARG(S_sel >= 1 && (int64_t)S_sel * L->l_sel >= S, "layer_prefill: block metadata (S_sel=%d) does not cover %d tokens", S_sel, S);
    NSA_CHECK_ARG(selector == NSA_SEL_BATCHED || selector == NSA_SEL_SEQUENTIAL, "layer_prefill: unknown selector %d", selector);
    const int B = kv->B, G = L->G, h = L->h, Dk = L->Dk, Dv = L->Dv, dt = L->dtype;
    const PrefillWs W = prefill_ws(L, B, S, S_sel);
    NSA_CHECK_ARG(workspace && ((uintptr_t)workspace % 256 == 0) && workspace_bytes >= W.total,
                  "layer_prefill: workspace missing, misaligned or too small");
    unsigned char *ws = (unsigned char *)workspace;
    void *Q = ws + W.q, *Ocmp = ws + W.ocmp, *Osel = ws + W.osel, *Owin = ws + W.owin;
    float *p_grp = (float *)(ws + W.pgrp);
    const int n_cmp = S < L->l ? 0 : (S - L->l) / L->d + 1;
    NSA_CHECK_ARG(n_cmp <= kv->n_cmp_max, "layer_prefill: compressed cache too small");
    if (int rc = nsa_rope_cache_append(L, kv, proj, Q, S, 0, stream)) return rc;
    if (int rc = nsa_cmp_pool_append(L, kv, 0, n_cmp, stream)) return rc;
    const int64_t ksb = (int64_t)G * kv->S_max * Dk, ksg = (int64_t)kv->S_max * Dk;
    const int64_t vsb = (int64_t)G * kv->S_max * Dv, vsg = (int64_t)kv->S_max * Dv;
    const int64_t kcb = (int64_t)G * kv->n_cmp_max * Dk, kcg = (int64_t)kv->n_cmp_max * Dk;
    const int64_t vcb = (int64_t)G * kv->n_cmp_max * Dv, vcg = (int64_t)kv->n_cmp_max * Dv;
    const float scale = 1.0f / sqrtf((float)Dk);
    // selected branch: scores (blocks no selector can read at row t are skipped) -> top-n + attention
    const bool aligned = ((uintptr_t)kv->K_cmp % 16 == 0) && kcb % 8 == 0 && kcg % 8 == 0 && Dk % 8 == 0;
    if (aligned && n_cmp >= 1 && tuning(TUNE_SEL_FUSE) <= 0) {
        // scores + top-n in one call (round 4: on the 32x32x16 scorer's route one LAUNCH, the selection in the scorer's epilogue), then the attention
        if (int rc = nsa_sel_scores_select(Q, kv->K_cmp, p_grp, B, S, G, h, Dk, n_cmp, kcb, kcg, Dk, csc_ptr, csc_rows, csc_vals, S_sel, L->l,
                                           L->d, L->l_sel, 2 /* skipped blocks stay unwritten: only the selector reads p_grp */, dt, scale, 0,
                                           L->n_sel, 1, 2, selector, S, ranges_out, out_width, ws + W.sc, W.sc_bytes, stream))
            return rc;
        if (int rc = nsa_sel_attn_fwd(Q, kv->K_sel, kv->V_sel, ranges_out, Osel, nullptr, B, S, G, h, Dk, Dv, S, out_width, ksb, ksg, Dk, vsb, vsg,
                                      Dv, dt, scale, 0, ws + W.att, W.att_bytes, stream))
            return rc;
    } else {
        if (int rc = nsa_sel_scores(Q, kv->K_cmp, p_grp, B, S, G, h, Dk, n_cmp, kcb, kcg, Dk, csc_ptr, csc_rows, csc_vals, S_sel, L->l, L->d,
                                    L->l_sel, 2 /* skipped blocks stay unwritten: only the selector below reads p_grp */, aligned ? 0 : 1, dt, scale,
                                    ws + W.sc, W.sc_bytes, stream))
            return rc;
        if (int rc = nsa_sel_select_attn_fwd(p_grp, 0, nullptr, S_sel, L->l_sel, L->n_sel, 1, 2, selector, S, ranges_out, out_width, Q, kv->K_sel,
                                             kv->V_sel, Osel, nullptr, B, S, G, h, Dk, Dv, S, ksb, ksg, Dk, vsb, vsg, Dv, dt, scale, ws + W.att,
                                             W.att_bytes, stream))
            return rc;
    }
    // sliding and compressed branches
    if (int rc = nsa_band_attn_fwd(Q, kv->K_win, kv->V_win, Owin, nullptr, B, S, G, h, Dk, Dv, S, ksb, ksg, Dk, vsb, vsg, Dv, 0, 0, 1, 0, L->w,
                                   dt, scale, 0, ws + W.band, W.band_bytes, stream))
        return rc;
    if (int rc = nsa_band_attn_fwd(Q, kv->K_cmp, kv->V_cmp, Ocmp, nullptr, B, S, G, h, Dk, Dv, n_cmp, kcb, kcg, Dk, vcb, vcg, Dv, 0, L->l, L->d,
                                   1, 1 << 30, dt, scale, 0, ws + W.band, W.band_bytes, stream))
        return rc;
    return nsa_gate_combine(L, Q, Ocmp, Osel, Owin, O_mix, gates_out, (int64_t)B * S * G, stream);
}

// workspace: proj | Q | O_cmp | O_sel | O_win | O_mix | ranges | selection-decode scratch | band scratch
struct DecodeWs {
    size_t proj, q, ocmp, osel, owin, omix, ranges, gates, sel, band, band2, total, sel_bytes, band_bytes;
};
static DecodeWs decode_ws(const nsa_layer_desc *L, int B, int S_max) {
    DecodeWs w;
    const size_t e = esize(L->dtype);
    const size_t NQ = (size_t)L->G * L->h * L->Dk, NO = (size_t)L->G * L->h * L->Dv;
    const size_t NT = NQ + 3 * (size_t)L->G * L->Dk + 3 * (size_t)L->G * L->Dv;
    const int n_cmp_max = S_max < L->l ? 0 : (S_max - L->l) / L->d + 1;
    const int S_sel_max = (S_max + L->l_sel - 1) / L->l_sel + 1;
    size_t o = 0;
    w.proj = o; o += up256(B * NT * e);
    w.q = o; o += up256(B * NQ * e);
    w.ocmp = o; o += up256(B * NO * e);
    w.osel = o; o += up256(B * NO * e);
    w.owin = o; o += up256(B * NO * e);
    w.omix = o; o += up256(B * NO * e);
    w.ranges = o; o += up256(sizeof(int32_t) * (size_t)B * L->G * L->n_sel * 2);
    w.gates = o; o += up256(sizeof(float) * (size_t)B * L->G * 3);
    w.sel_bytes = nsa_sel_decode_step_workspace(B, L->G, L->h, L->Dk, L->Dv, n_cmp_max, S_sel_max, L->n_sel, L->dtype);
    w.sel = o; o += up256(w.sel_bytes);
    w.band_bytes = nsa_band_attn_fwd_workspace(B, 1, L->G, L->h, L->Dk, L->Dv, L->dtype);
    w.band = o; o += up256(w.band_bytes);
    w.band2 = o; o += up256(w.band_bytes);
    w.total = o;
    return w;
}

size_t nsa_layer_decode_step_workspace(const nsa_layer_desc *L, int B, int S_max) {
    if (!L || !dt_ok(L->dtype) || B < 1 || S_max < 1) return 0;
    return decode_ws(L, B, S_max).total;
}

static int layer_decode_step_impl(const nsa_layer_desc *L, const nsa_kv_desc *kv, const void *x, void *y, int t, const int32_t *csc_ptr,
                                  const int32_t *csc_rows, const float *csc_vals, int S_sel, int32_t *ranges_out, float *gates_out,
                                  void *workspace, size_t workspace_bytes, void *stream, const void *residual, const void *norm_w = nullptr,
                                  float norm_eps = 0.f) {
    if (int rc = check_layer(L, "layer_decode_step")) return rc;
    if (int rc = check_kv(kv, "layer_decode_step")) return rc;
    NSA_CHECK_ARG(x && y && L->W_qkv && L->W_out, "layer_decode_step: null pointer");
    NSA_CHECK_ARG(t >= 0 && t < kv->S_max, "layer_decode_step: position %d outside the cache capacity %d", t, kv->S_max);
    NSA_CHECK_ARG(S_sel >= 1 && (int64_t)S_sel * L->l_sel >= t + 1, "layer_decode_step: block metadata (S_sel=%d) does not cover token %d", S_sel, t);
    const int B = kv->B;
    const DecodeWs W = decode_ws(L, B, kv->S_max);
    NSA_CHECK_ARG(workspace && ((uintptr_t)workspace % 256 == 0) && workspace_bytes >= W.total,
                  "layer_decode_step: workspace missing, misaligned or too small");
    unsigned char *ws = (unsigned char *)workspace;
    hipStream_t st = (hipStream_t)stream;
    const int dt = L->dtype;
    const int G = L->G, h = L->h, Dk = L->Dk, Dv = L->Dv;
    const int NO = G * h * Dv;
    void *proj = ws + W.proj, *Q = ws + W.q, *Ocmp = ws + W.ocmp, *Osel = ws + W.osel, *Owin = ws + W.owin, *Omix = ws + W.omix;
    int32_t *ranges = ranges_out ? ranges_out : (int32_t *)(ws + W.ranges);

    RopeAppendParams RP{};
    RP.proj = proj;
    RP.Q_out = Q;
    RP.cache[0] = kv->K_sel; RP.cache[1] = kv->V_sel; RP.cache[2] = kv->K_win; RP.cache[3] = kv->V_win; RP.cache[4] = kv->K_raw; RP.cache[5] = kv->V_raw;
    RP.B = B; RP.S = 1; RP.G = G; RP.h = h; RP.Dk = Dk; RP.Dv = Dv; RP.S_max = kv->S_max; RP.t0 = t;
    RP.rope_base = L->rope_base > 0.f ? L->rope_base : 10000.0f;
    RP.inv_scale = 1.0f / (L->rope_scale > 0.f ? L->rope_scale : 1.0f);
    // 1+2. fused QKV projection with RoPE + cache append at position t in its epilogue
    if (int rc = launch_qkv_rope_append(RP, x, L->W_qkv, L->dim, dt, st, norm_w, norm_eps)) return rc;
    // 3. emit a compressed token when a window completes (nsa_attention.py:588-604)
    const int S_raw = t + 1;
    const int n_cmp = S_raw < L->l ? 0 : (S_raw - L->l) / L->d + 1;
    NSA_CHECK_ARG(n_cmp <= kv->n_cmp_max, "layer_decode_step: compressed cache too small");
    if (S_raw >= L->l && (S_raw - L->l) % L->d == 0)
        if (int rc = nsa_cmp_pool_append(L, kv, n_cmp - 1, n_cmp, stream)) return rc;
    const int64_t ksb = (int64_t)G * kv->S_max * Dk, ksg = (int64_t)kv->S_max * Dk;
    const int64_t vsb = (int64_t)G * kv->S_max * Dv, vsg = (int64_t)kv->S_max * Dv;
    const int64_t kcb = (int64_t)G * kv->n_cmp_max * Dk, kcg = (int64_t)kv->n_cmp_max * Dk;
    const int64_t vcb = (int64_t)G * kv->n_cmp_max * Dv, vcg = (int64_t)kv->n_cmp_max * Dv;
    const float scale = 1.0f / sqrtf((float)Dk);
    // When the final pass can take split-KV partial records (Dv = 64) the three branches skip their own combine kernels:
    // one kernel then merges the splits of all branches, evaluates the gate and mixes.
    const int defer = Dv == 64 ? 1 : 0;
    DecodeFinishParams F{};
    F.Q = Q; F.O_out = Omix; F.gates_out = gates_out;
    F.w1 = L->gate_w1; F.b1 = L->gate_b1; F.w2 = L->gate_w2; F.b2 = L->gate_b2;
    F.R = (int64_t)B * G; F.h = h; F.Dk = Dk; F.Dv = Dv; F.Hd = L->gate_hidden; F.tau = L->gate_tau;
    F.O[0] = Ocmp; F.O[1] = Osel; F.O[2] = Owin;
    // 4 + 5. the three branches.  The sliding and the compressed branch run in split-KV form with the combine left to the finish kernel; when
    // the selected branch runs as the one-launch decode step they ride on ITS launch (workgroups behind the step's own: sel_decode_fused.hip),
    // otherwise they are one launch of their own.
    int ns_band = 1;
    band_attn_workspace(B, 1, G, h, Dk, Dv, dt, &ns_band);
    const bool dual = defer && ns_band > 1 && n_cmp > 0 && L->w > 0 && band_attn_mfma_supported(dt, h, Dk, Dv) &&
                      ((uintptr_t)kv->K_win % 16 == 0) && ((uintptr_t)kv->V_win % 16 == 0) && ((uintptr_t)kv->K_cmp % 16 == 0) &&
                      ((uintptr_t)kv->V_cmp % 16 == 0) && ksb * 2 < ((int64_t)1 << 31) && vsb * 2 < ((int64_t)1 << 31);
    DecBandPair BP{};
    BandAttnParams &PW = BP.w, &PC = BP.c;
    if (dual) {
        PW.Q = Q; PW.K = kv->K_win; PW.V = kv->V_win; PW.O = Owin;
        PW.B = B; PW.S = 1; PW.G = G; PW.h = h; PW.Dk = Dk; PW.Dv = Dv; PW.S_kv = S_raw;
        PW.ksb = ksb; PW.ksg = ksg; PW.kss = Dk; PW.vsb = vsb; PW.vsg = vsg; PW.vss = Dv;
        PW.scale = scale; PW.t0 = t; PW.a = 0; PW.dd = 1; PW.c = 0; PW.w = L->w;
        PW.part = (float *)(ws + W.band); PW.nsplit = ns_band; PW.defer_combine = 1;
        PC = PW;
        PC.K = kv->K_cmp; PC.V = kv->V_cmp; PC.O = Ocmp; PC.S_kv = n_cmp;
        PC.ksb = kcb; PC.ksg = kcg; PC.vsb = vcb; PC.vsg = vcg;
        PC.a = L->l; PC.dd = L->d; PC.c = 1; PC.w = 1 << 30;
        PC.part = (float *)(ws + W.band2);
    }
    const int band_mode = tuning(TUNE_DECODE_BAND);  // 0 own launch, 1 ride, 2 ride + merge in the workgroup, -1 / 3: 2 + the mix in the output projection
    const bool ride = dual && Dk == 64 && band_mode != 0;
    float *gates = gates_out ? gates_out : (float *)(ws + W.gates);
    if (ride && band_mode != 1 && h <= 16) {
        BP.mg.on = 1;
        BP.mg.Hd = L->gate_hidden;
        BP.mg.tau = L->gate_tau;
        BP.mg.gw1 = L->gate_w1; BP.mg.gb1 = L->gate_b1; BP.mg.gw2 = L->gate_w2; BP.mg.gb2 = L->gate_b2;
        BP.mg.gates = gates;
    }
    float *sel_part = nullptr;
    int band_taken = 0;
    if (int rc = sel_decode_step_impl(Q, kv->K_cmp, kv->K_sel, kv->V_sel, csc_ptr, csc_rows, csc_vals, ranges, Osel, B, G, h, Dk, Dv, n_cmp,
                                      S_sel, S_raw, L->l, L->d, L->l_sel, L->n_sel, t, kcb, kcg, Dk, ksb, ksg, Dk, vsb, vsg, Dv, dt, scale,
                                      ws + W.sel, W.sel_bytes, stream, defer, &F.ns[1], &sel_part, ride ? &BP : nullptr, &band_taken))
        return rc;
    F.part[1] = sel_part;
    if (dual && band_taken && BP.mg.on) {
        // O_win, O_cmp (merged by the workgroups that held their splits), O_sel and the row gates are final: with few rows the mix is the
        // A operand of the output projection -- three launches per step
        F.ns[2] = F.ns[0] = 1;
        // (measured: the mix in the projection wins up to 32 rows -- 43.3 -> 40.5 us at B = 32 -- and loses from 64 on, where every one of the
        // projection's 48 workgroups would redo the mix of all rows: 47.1 -> 50.6 us; DECODE_BAND = 3 takes it at any batch)
        if (((band_mode < 0 && B <= 32) || band_mode == 3) && F.ns[1] == 1 &&
            linear_small_mix_supported(dt, B, L->dim, NO, G, Ocmp, Osel, Owin, L->W_out) && Dv == 64)
            return launch_linear_small_mix(Ocmp, Osel, Owin, gates, L->W_out, y, B, L->dim, NO, G, dt, residual ? 2 : 0, residual, st);
    } else if (dual) {
        if (!band_taken)
            if (int rc = launch_band_attn_fwd_dual(PW, PC, dt, st)) return rc;
        F.ns[2] = F.ns[0] = ns_band;
        F.part[2] = PW.part;
        F.part[0] = PC.part;
    } else {
        if (int rc = band_attn_fwd_impl(Q, kv->K_win, kv->V_win, Owin, nullptr, B, 1, G, h, Dk, Dv, S_raw, ksb, ksg, Dk, vsb, vsg, Dv, t, 0, 1, 0,
                                        L->w, dt, scale, 0, ws + W.band, W.band_bytes, stream, defer, &F.ns[2]))
            return rc;
        F.part[2] = (const float *)(ws + W.band);
        if (int rc = band_attn_fwd_impl(Q, kv->K_cmp, kv->V_cmp, Ocmp, nullptr, B, 1, G, h, Dk, Dv, n_cmp, kcb, kcg, Dk, vcb, vcg, Dv, t, L->l,
                                        L->d, 1, 1 << 30, dt, scale, 0, ws + W.band2, W.band_bytes, stream, defer, &F.ns[0]))
            return rc;
        F.part[0] = (const float *)(ws + W.band2);
    }
    // 6. split combine + gates + mix, 7. output projection
    if (defer) {
        if (int rc = launch_decode_finish(F, dt, st)) return rc;
    } else {
        if (int rc = nsa_gate_combine(L, Q, Ocmp, Osel, Owin, Omix, gates_out, (int64_t)B * G, stream)) return rc;
    }
    return launch_linear_small_epi(Omix, L->W_out, y, B, L->dim, NO, dt, residual ? 2 : 0, residual, st);
}

int nsa_layer_decode_step(const nsa_layer_desc *L, const nsa_kv_desc *kv, const void *x, void *y, int t, const int32_t *csc_ptr,
                          const int32_t *csc_rows, const float *csc_vals, int S_sel, int32_t *ranges_out, float *gates_out,
                          void *workspace, size_t workspace_bytes, void *stream) {
    return layer_decode_step_impl(L, kv, x, y, t, csc_ptr, csc_rows, csc_vals, S_sel, ranges_out, gates_out, workspace, workspace_bytes, stream,
                                  nullptr);
}

// block workspace: xn | h | hn | u [B, mlp_hidden] | layer decode workspace
struct BlockWs {
    size_t xn, h, hn, u, layer, total, layer_bytes;
};
static BlockWs block_ws(const nsa_block_desc *Bk, int B, int S_max) {
    BlockWs w;
    const size_t e = esize(Bk->attn.dtype);
    size_t o = 0;
    w.xn = o; o += up256((size_t)B * Bk->attn.dim * e);
    w.h = o; o += up256((size_t)B * Bk->attn.dim * e);
    w.hn = o; o += up256((size_t)B * Bk->attn.dim * e);
    w.u = o; o += up256((size_t)B * Bk->mlp_hidden * e);
    w.layer_bytes = decode_ws(&Bk->attn, B, S_max).total;
    w.layer = o; o += up256(w.layer_bytes);
    w.total = o;
    return w;
}

size_t nsa_block_decode_step_workspace(const nsa_block_desc *Bk, int B, int S_max) {
    if (!Bk || !dt_ok(Bk->attn.dtype) || B < 1 || S_max < 1 || Bk->mlp_hidden < 1) return 0;
    return block_ws(Bk, B, S_max).total;
}

int nsa_block_decode_step(const nsa_block_desc *Bk, const nsa_kv_desc *kv, const void *x, void *y, int t, const int32_t *csc_ptr,
                          const int32_t *csc_rows, const float *csc_vals, int S_sel, int32_t *ranges_out, float *gates_out,
                          void *workspace, size_t workspace_bytes, void *stream) {
    NSA_CHECK_ARG(Bk, "block_decode_step: null descriptor");
    if (int rc = check_layer(&Bk->attn, "block_decode_step")) return rc;
    if (int rc = check_kv(kv, "block_decode_step")) return rc;
    NSA_CHECK_ARG(x && y && Bk->norm1_w && Bk->norm2_w && Bk->mlp_w1 && Bk->mlp_w2 && Bk->mlp_hidden >= 1, "block_decode_step: null pointer");
    const int B = kv->B, dim = Bk->attn.dim, dt = Bk->attn.dtype;
    const BlockWs W = block_ws(Bk, B, kv->S_max);
    NSA_CHECK_ARG(workspace && ((uintptr_t)workspace % 256 == 0) && workspace_bytes >= W.total,
                  "block_decode_step: workspace missing, misaligned or too small");
    unsigned char *ws = (unsigned char *)workspace;
    hipStream_t st = (hipStream_t)stream;
    void *xn = ws + W.xn, *h = ws + W.h, *hn = ws + W.hn, *u = ws + W.u;
    const float eps = Bk->norm_eps > 0.f ? Bk->norm_eps : 1e-6f;
    // small batches: both RMSNorms are folded into the projections that consume them (two launches fewer per block)
    RopeAppendParams probe{};
    probe.B = B; probe.G = Bk->attn.G; probe.h = Bk->attn.h; probe.Dk = Bk->attn.Dk; probe.Dv = Bk->attn.Dv;
    const bool fold1 = qkv_can_fold_norm(probe, x, Bk->attn.W_qkv, dim, dt);
    if (!fold1)
        if (int rc = launch_rmsnorm_rows(x, Bk->norm1_w, xn, B, dim, eps, dt, st)) return rc;
    // h = x + attn(norm1(x)): the residual rides in the output projection's epilogue
    if (int rc = layer_decode_step_impl(&Bk->attn, kv, fold1 ? x : xn, h, t, csc_ptr, csc_rows, csc_vals, S_sel, ranges_out, gates_out,
                                        ws + W.layer, W.layer_bytes, stream, x, fold1 ? Bk->norm1_w : nullptr, eps))
        return rc;
    if (linear_small_can_fold_norm(dt, B, Bk->mlp_hidden, dim, h, Bk->mlp_w1)) {
        if (int rc = launch_linear_small_norm(h, Bk->mlp_w1, u, B, Bk->mlp_hidden, dim, dt, 1, nullptr, Bk->norm2_w, eps, st)) return rc;
    } else {
        if (int rc = launch_rmsnorm_rows(h, Bk->norm2_w, hn, B, dim, eps, dt, st)) return rc;
        if (int rc = launch_linear_small_epi(hn, Bk->mlp_w1, u, B, Bk->mlp_hidden, dim, dt, 1, nullptr, st)) return rc;  // silu(fc1)
    }
    return launch_linear_small_epi(u, Bk->mlp_w2, y, B, dim, Bk->mlp_hidden, dt, 2, h, st);  // fc2 + h
}

// model workspace: x ping | x pong | block workspace
size_t nsa_model_decode_step_workspace(const nsa_block_desc *blocks, int n_blocks, int B, int S_max) {
    if (!blocks || n_blocks < 1 || B < 1 || S_max < 1) return 0;
    const size_t xb = up256((size_t)B * blocks[0].attn.dim * esize(blocks[0].attn.dtype));
    size_t bw = 0;
    for (int i = 0; i < n_blocks; ++i) {
        const size_t w = nsa_block_decode_step_workspace(&blocks[i], B, S_max);
        if (w == 0) return 0;
        if (w > bw) bw = w;
    }
    return 2 * xb + up256(bw) + up256((size_t)B * 64 * 8);  // + argmax partials (up to 64 chunks of 4096 logits)
}

int nsa_model_decode_step(const nsa_block_desc *blocks, const nsa_kv_desc *kvs, int n_blocks, const int32_t *tokens, const void *embed,
                          const void *norm_f_w, const void *lm_head, int vocab, void *logits, int32_t *next_tokens, int t,
                          const int32_t *csc_ptr, const int32_t *csc_rows, const float *csc_vals, int S_sel, void *workspace,
                          size_t workspace_bytes, void *stream) {
    NSA_CHECK_ARG(blocks && kvs && n_blocks >= 1 && tokens && embed && norm_f_w && lm_head && logits && vocab >= 1, "model_decode_step: null pointer");
    const int B = kvs[0].B, dim = blocks[0].attn.dim, dt = blocks[0].attn.dtype;
    for (int i = 0; i < n_blocks; ++i)
        NSA_CHECK_ARG(kvs[i].B == B && blocks[i].attn.dim == dim && blocks[i].attn.dtype == dt && kvs[i].S_max == kvs[0].S_max,
                      "model_decode_step: blocks / caches disagree on batch, width, dtype or capacity");
    const size_t need = nsa_model_decode_step_workspace(blocks, n_blocks, B, kvs[0].S_max);
    NSA_CHECK_ARG(need > 0 && workspace && ((uintptr_t)workspace % 256 == 0) && workspace_bytes >= need,
                  "model_decode_step: workspace missing, misaligned or too small");
    hipStream_t st = (hipStream_t)stream;
    const size_t xb = up256((size_t)B * dim * esize(dt));
    unsigned char *ws = (unsigned char *)workspace;
    void *xa = ws, *xc = ws + xb;
    unsigned char *bws = ws + 2 * xb;
    if (int rc = launch_embed_rows(tokens, embed, xa, B, dim, vocab, dt, st)) return rc;
    for (int i = 0; i < n_blocks; ++i) {
        if (int rc = nsa_block_decode_step(&blocks[i], &kvs[i], xa, xc, t, csc_ptr, csc_rows, csc_vals, S_sel, nullptr, nullptr, bws,
                                           workspace_bytes - 2 * xb, stream))
            return rc;
        void *tmp = xa;
        xa = xc;
        xc = tmp;
    }
    const float eps = blocks[0].norm_eps > 0.f ? blocks[0].norm_eps : 1e-6f;
    if (linear_small_can_fold_norm(dt, B, vocab, dim, xa, lm_head)) {
        if (int rc = launch_linear_small_norm(xa, lm_head, logits, B, vocab, dim, dt, 0, nullptr, norm_f_w, eps, st)) return rc;
    } else {
        if (int rc = launch_rmsnorm_rows(xa, norm_f_w, xc, B, dim, eps, dt, st)) return rc;
        if (int rc = launch_linear_small_epi(xc, lm_head, logits, B, vocab, dim, dt, 0, nullptr, st)) return rc;
    }
    if (next_tokens) {
        NSA_CHECK_ARG(argmax_rows_workspace(B, vocab) <= (size_t)B * 64 * 8, "model_decode_step: vocabulary too large for the in-call argmax");
        return launch_argmax_rows(logits, next_tokens, B, vocab, dt, ws + need - up256((size_t)B * 64 * 8), st);
    }
    return NSA_OK;
}

}  // extern "C"
