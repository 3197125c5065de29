// C ABI of libnsa_sel_hip.so (see include/nsa_sel_hip.h).  Host-side argument checking,
// error reporting and kernel dispatch; no torch types, no allocation, no synchronisation.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <stdlib.h>
#include <strings.h>

#include <atomic>
#include <cmath>
#include <mutex>
#include <vector>

#include "nsa_common.hpp"
#include "sel_attn_params.hpp"
#include "nsa_internal.hpp"
#include "sel_select_row.hpp"

namespace nsa {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what) {
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return NSA_ERR_HIP;
}

// ---- tuning switches -------------------------------------------------------------------------
static const char *const g_tune_names[TUNE_COUNT] = {"SEL_ROWS", "ATTN_MAP", "ATTN_STAGE", "BAND_STAGE", "DECODE_UNFUSED", "SEL_BLOCKS", "DECODE_WG", "SEL_ROWSUM", "DECODE_STENCIL", "SEL_FUSE", "SCORES_FORM", "SEL_FLAT", "SEL_KSPLIT", "DECODE_STOP", "DECODE_WAVES", "DECODE_SPLIT", "DECODE_STEP", "DECODE_TEAM_SPIN", "DECODE_WIDE", "SEL_KSPLIT_T1", "SEL_KSPLIT_T2", "SCORES_SELECT", "DECODE_BAND"};
static const int g_tune_defaults[TUNE_COUNT] = {-1, -1, 1, 1, -1, -1, -1, 1, 1, 0, -1, -1, -1, 0, -1, -1, 1, -1, -1, -1, -1, -1, -1};
static std::atomic<int> g_tune[TUNE_COUNT];
static std::once_flag g_tune_once;

static void tune_init() {
    for (int i = 0; i < TUNE_COUNT; ++i) {
        char name[64];
        snprintf(name, sizeof(name), "NSA_HIP_%s", g_tune_names[i]);
        const char *e = getenv(name);
        int v = e ? atoi(e) : g_tune_defaults[i];
#ifndef NSA_DEC_TS
        if (i == TUNE_DECODE_STOP) v = 0;  // TIMELINE build only (see nsa_hip_set_tuning)
#endif
        g_tune[i].store(v, std::memory_order_relaxed);
    }
}

int tuning(Tune t) {
    std::call_once(g_tune_once, tune_init);
    return g_tune[t].load(std::memory_order_relaxed);
}

static int tune_index(const char *name) {
    if (!name) return -1;
    if (strncmp(name, "NSA_HIP_", 8) == 0) name += 8;
    for (int i = 0; i < TUNE_COUNT; ++i)
        if (strcasecmp(name, g_tune_names[i]) == 0) return i;
    return -1;
}

// implemented in the kernel translation units
int launch_select_topn(const float *, int64_t, int, int, int, const int32_t *, int, int, int, int, int, int, int,
                       int32_t *, int, hipStream_t);
int launch_indices_to_ranges(const int32_t *, int64_t, int, int, int, int, int, int, int32_t *, hipStream_t);
int batched_width(int, int, int, int, int, int);
int launch_map_pcmp(const float *, int64_t, int, int, const int32_t *, const int32_t *, const float *, int, float *,
                    float *, hipStream_t);
int launch_pcmp(const void *, const void *, float *, int64_t, int64_t, int, int, int, int, int, int64_t, int64_t,
                int64_t, int, float, hipStream_t);
size_t scores_workspace(int64_t, int, int);
int launch_sel_scores(const void *, const void *, float *, int, int, int, int, int, int, int64_t, int64_t, int64_t,
                      const int32_t *, const int32_t *, const float *, int, int, float, void *, size_t, hipStream_t);

constexpr int64_t DECODE_MAX_ROWS = 1024;  // rows (B*S*G) up to which the decode-shaped scorer is used
size_t decode_scores_workspace(int64_t, int, int);
int launch_decode_scores(const void *, const void *, float *, int, int, int, int, int, int, int64_t, int64_t, int64_t,
                         const int32_t *, const int32_t *, const float *, int, int, float, void *, size_t, hipStream_t);
bool scores_mfma_supported(int, int, int, int, int, int);
int launch_sel_scores_mfma(const void *, const void *, float *, int, int, int, int, int, int, int64_t, int64_t, int64_t, int,
                           int, int, float, int, hipStream_t, const SelectParams *, int *);

int launch_sel_first_key(const void *, const int32_t *, void *, int64_t, int, int, int, int, int, int, int64_t, int64_t, int64_t, int,
                         hipStream_t);

static bool dtype_ok(int dt) { return dt == NSA_DT_F32 || dt == NSA_DT_BF16 || dt == NSA_DT_F16; }

}  // namespace nsa

using namespace nsa;

extern "C" {

int nsa_hip_abi_version(void) { return NSA_HIP_ABI_VERSION; }

const char *nsa_hip_last_error(void) { return g_err; }

int nsa_hip_set_tuning(const char *name, int value) {
    const int i = tune_index(name);
    NSA_CHECK_ARG(i >= 0, "unknown tuning switch '%s'", name ? name : "(null)");
#ifndef NSA_DEC_TS
    // the phase stops of the fused decode kernel leave O and the ranges unwritten: they exist in the TIMELINE build only
    NSA_CHECK_ARG(i != TUNE_DECODE_STOP || value == 0, "DECODE_STOP is a measurement aid of the TIMELINE build (make TIMELINE=1)");
#endif
    std::call_once(g_tune_once, tune_init);
    g_tune[i].store(value, std::memory_order_relaxed);
    return NSA_OK;
}

int nsa_hip_get_tuning(const char *name, int *value) {
    const int i = tune_index(name);
    NSA_CHECK_ARG(i >= 0 && value, "unknown tuning switch '%s'", name ? name : "(null)");
    *value = tuning((Tune)i);
    return NSA_OK;
}

int nsa_hip_device_check(int dev, int *cu_count, size_t *hbm_bytes) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= dev || dev < 0) {
        set_error("no HIP device %d (count %d)", dev, n);
        return NSA_ERR_NO_DEVICE;
    }
    hipDeviceProp_t prop;
    NSA_HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s, this library is built for gfx950 only", dev, prop.gcnArchName);
        return NSA_ERR_NO_DEVICE;
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    return NSA_OK;
}

// ------------------------------------------------------------------------------ attention
size_t nsa_sel_attn_fwd_workspace_kv(int B, int S, int G, int h, int Dk, int Dv, int S_kv, int n_ranges, int dtype) {
    if (!sel_attn_mfma_supported(dtype, h, Dk, Dv)) return 0;
    const size_t split_kv = sel_attn_mfma_workspace((int64_t)B * S * G, h, Dv, nullptr);
    // key-split form of the block kernel (long contexts): its zones follow the rows' positions S_kv - S + row
    const size_t key_split = sel_attn_ksplit_workspace(dtype, h, Dk, Dv, S, S_kv > S ? S_kv : S, n_ranges, (int64_t)B * S * G);
    return split_kv > key_split ? split_kv : key_split;
}

size_t nsa_sel_attn_fwd_workspace(int B, int S, int G, int h, int Dk, int Dv, int n_ranges, int dtype) {
    return nsa_sel_attn_fwd_workspace_kv(B, S, G, h, Dk, Dv, S, n_ranges, dtype);  // the prefill case S_kv = S
}

}  // extern "C"

int nsa::sel_attn_fwd_impl(const void *Q, const void *K, const void *V, const int32_t *ranges, void *O, float *lse, int B, int S, int G,
                           int h, int Dk, int Dv, int S_kv, int n_ranges, int64_t ksb, int64_t ksg, int64_t kss, int64_t vsb, int64_t vsg,
                           int64_t vss, int dtype, float scale, int variant, void *workspace, size_t workspace_bytes, void *stream,
                           int defer, int *ns_used) {
    if (ns_used) *ns_used = 1;
    NSA_CHECK_ARG(dtype_ok(dtype), "sel_attn_fwd: unknown dtype %d", dtype);
    NSA_CHECK_ARG(B >= 0 && S >= 0 && G >= 1 && h >= 1 && Dk >= 1 && Dv >= 1 && S_kv >= 0 && n_ranges >= 0,
                  "sel_attn_fwd: negative size");
    NSA_CHECK_ARG(n_ranges <= 64, "sel_attn_fwd: at most 64 ranges per row are supported (got %d)", n_ranges);
    NSA_CHECK_ARG(Dk <= 256 && Dv <= 256, "sel_attn_fwd: Dk/Dv up to 256 supported (got %d/%d)", Dk, Dv);
    const int64_t R = (int64_t)B * S * G;
    if (R == 0) return NSA_OK;
    NSA_CHECK_ARG(Q && O && ranges || n_ranges == 0, "sel_attn_fwd: null pointer");
    NSA_CHECK_ARG((K && V) || S_kv == 0, "sel_attn_fwd: null K/V");
    hipStream_t st = (hipStream_t)stream;
    const size_t esz = dtype == NSA_DT_F32 ? 4 : 2;
    if (S_kv == 0 || n_ranges == 0) {  // attention_kernels.py:718-719
        NSA_HIP_TRY(hipMemsetAsync(O, 0, (size_t)R * h * Dv * esz, st));
        if (lse) {
            std::vector<float> neg;  // lse = -inf: fill with the bit pattern 0xff800000
            NSA_HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)lse, (int)0xff800000, (size_t)R * h, st));
        }
        return NSA_OK;
    }
    SelAttnParams P{};
    P.Q = Q; P.K = K; P.V = V; P.ranges = ranges; P.O = O; P.lse = lse; P.R = R;
    P.S = S; P.G = G; P.h = h; P.Dk = Dk; P.Dv = Dv; P.S_kv = S_kv; P.n = n_ranges;
    P.ksb = ksb; P.ksg = ksg; P.kss = kss; P.vsb = vsb; P.vsg = vsg; P.vss = vss;
    P.scale = scale > 0.f ? scale : 1.0f / sqrtf((float)Dk);
    P.part = nullptr; P.nsplit = 1;
    const bool fast_ok = sel_attn_mfma_supported(dtype, h, Dk, Dv) && kss % 8 == 0 && vss % 8 == 0 && ksb % 8 == 0 &&
                         vsb % 8 == 0 && ksg % 8 == 0 && vsg % 8 == 0 && ((uintptr_t)Q % 16 == 0) &&
                         ((uintptr_t)K % 16 == 0) && ((uintptr_t)V % 16 == 0);
    if (variant == 2) NSA_CHECK_ARG(fast_ok, "sel_attn_fwd: MFMA variant requested but shape/dtype/alignment unsupported");
    NSA_CHECK_ARG(variant >= 0 && variant <= 2, "sel_attn_fwd: unknown variant %d", variant);
    if (variant == 0 && S == 1 && !lse && sel_attn_decode_wg_supported(dtype, h, Dk, Dv, n_ranges, ksb, ksg, kss, vsb, vsg, vss, Q, K, V) &&
        (int64_t)S_kv * 128 < ((int64_t)1 << 31))  // decode: one workgroup per row, partials merged through LDS, no combine launch
        return launch_sel_attn_decode_wg(Q, K, V, ranges, O, R, G, h, S_kv, n_ranges, ksb, ksg, kss, vsb, vsg, vss, dtype, P.scale, st);
    if (variant == 2 || (variant == 0 && fast_ok)) {
        int ns = 1;
        const size_t need = sel_attn_mfma_workspace(R, h, Dv, &ns);
        if (ns > 1 && workspace && workspace_bytes >= need && ((uintptr_t)workspace % 16 == 0)) {
            P.part = (float *)workspace;
            P.nsplit = ns;
            P.defer_combine = defer;
            if (ns_used && defer) *ns_used = ns;
        } else if (ns == 1 && workspace && ((uintptr_t)workspace % 16 == 0)) {
            P.ks_ws = workspace;  // the block kernel may split the keys of a pair over two XCD groups (partial records go here)
            P.ks_bytes = workspace_bytes;
        }
        return launch_sel_attn_fwd_mfma(P, dtype, st);
    }
    return launch_sel_attn_fwd_generic(P, dtype, st);
}


extern "C" {

int nsa_sel_attn_fwd(const void *Q, const void *K, const void *V, const int32_t *ranges, void *O, float *lse, int B,
                     int S, int G, int h, int Dk, int Dv, int S_kv, int n_ranges, int64_t ksb, int64_t ksg,
                     int64_t kss, int64_t vsb, int64_t vsg, int64_t vss, int dtype, float scale, int variant,
                     void *workspace, size_t workspace_bytes, void *stream) {
    return sel_attn_fwd_impl(Q, K, V, ranges, O, lse, B, S, G, h, Dk, Dv, S_kv, n_ranges, ksb, ksg, kss, vsb, vsg, vss, dtype, scale, variant,
                             workspace, workspace_bytes, stream, 0, nullptr);
}

int nsa_sel_attn_first_key_parity(const void *V, const int32_t *ranges, void *O, int B, int S, int G, int h, int Dv, int S_kv, int n_ranges,
                                  int64_t vsb, int64_t vsg, int64_t vss, int dtype, void *stream) {
    NSA_CHECK_ARG(dtype_ok(dtype), "sel_attn_first_key_parity: unknown dtype %d", dtype);
    NSA_CHECK_ARG(B >= 0 && S >= 0 && G >= 1 && h >= 1 && Dv >= 1 && S_kv >= 0 && n_ranges >= 0, "sel_attn_first_key_parity: negative size");
    NSA_CHECK_ARG(n_ranges <= 64, "sel_attn_first_key_parity: at most 64 ranges per row are supported (got %d)", n_ranges);
    const int64_t R = (int64_t)B * S * G;
    if (R == 0) return NSA_OK;
    NSA_CHECK_ARG(O && (ranges || n_ranges == 0) && (V || S_kv == 0), "sel_attn_first_key_parity: null pointer");
    const int esz = dtype == NSA_DT_F32 ? 4 : 2;
    if (S_kv == 0 || n_ranges == 0) {
        NSA_HIP_TRY(hipMemsetAsync(O, 0, (size_t)R * h * Dv * esz, (hipStream_t)stream));
        return NSA_OK;
    }
    return launch_sel_first_key(V, ranges, O, R, S, G, h, Dv, n_ranges, S_kv, vsb, vsg, vss, esz, (hipStream_t)stream);
}

int nsa_sel_attn_head_causal_parity(const void *Q, const void *K, const void *V, const int32_t *ranges, void *O, int B, int S, int G, int h,
                                    int Dk, int Dv, int S_kv, int n_ranges, int64_t ksb, int64_t ksg, int64_t kss, int64_t vsb, int64_t vsg,
                                    int64_t vss, int dtype, float scale, void *stream) {
    NSA_CHECK_ARG(dtype_ok(dtype), "sel_attn_head_causal_parity: unknown dtype %d", dtype);
    NSA_CHECK_ARG(B >= 0 && S >= 0 && G >= 1 && h >= 1 && Dk >= 1 && Dv >= 1 && S_kv >= 0 && n_ranges >= 0,
                  "sel_attn_head_causal_parity: negative size");
    NSA_CHECK_ARG(n_ranges <= 64, "sel_attn_head_causal_parity: at most 64 ranges per row are supported (got %d)", n_ranges);
    const int64_t R = (int64_t)B * S * G;
    if (R == 0) return NSA_OK;
    NSA_CHECK_ARG(Q && O && (ranges || n_ranges == 0) && ((K && V) || S_kv == 0), "sel_attn_head_causal_parity: null pointer");
    const int esz = dtype == NSA_DT_F32 ? 4 : 2;
    if (S_kv == 0 || n_ranges == 0) {
        NSA_HIP_TRY(hipMemsetAsync(O, 0, (size_t)R * h * Dv * esz, (hipStream_t)stream));
        return NSA_OK;
    }
    SelAttnParams P{};
    P.Q = Q, P.K = K, P.V = V, P.ranges = ranges, P.O = O, P.lse = nullptr;
    P.R = R, P.S = S, P.G = G, P.h = h, P.Dk = Dk, P.Dv = Dv, P.S_kv = S_kv, P.n = n_ranges;
    P.ksb = ksb, P.ksg = ksg, P.kss = kss, P.vsb = vsb, P.vsg = vsg, P.vss = vss;
    P.scale = scale > 0.f ? scale : 1.0f / sqrtf((float)Dk);
    return launch_sel_head_causal(P, dtype, (hipStream_t)stream);
}

size_t nsa_sel_attn_bwd_workspace(int B, int S, int G, int h, int Dk, int Dv, int S_kv, int dtype, int variant) {
    if (variant == 1 || !sel_attn_bwd_mfma_supported(dtype, h, Dk, Dv)) return 0;
    return sel_attn_bwd_mfma_workspace((int64_t)B * S * G, h, S, (int64_t)B * G, S_kv);
}

int nsa_sel_attn_bwd(const void *Q, const void *K, const void *V, const int32_t *ranges, const void *O,
                     const float *lse, const void *dO, void *dQ, float *dK, float *dV, int B, int S, int G, int h,
                     int Dk, int Dv, int S_kv, int n_ranges, int64_t ksb, int64_t ksg, int64_t kss, int64_t vsb,
                     int64_t vsg, int64_t vss, int dtype, float scale, int variant, void *workspace, size_t workspace_bytes,
                     void *stream) {
    NSA_CHECK_ARG(dtype_ok(dtype), "sel_attn_bwd: unknown dtype %d", dtype);
    NSA_CHECK_ARG(B >= 0 && S >= 0 && G >= 1 && h >= 1 && Dk >= 1 && Dv >= 1 && S_kv >= 0 && n_ranges >= 0,
                  "sel_attn_bwd: negative size");
    NSA_CHECK_ARG(n_ranges <= 64, "sel_attn_bwd: at most 64 ranges per row");
    NSA_CHECK_ARG(Dk <= 256 && Dv <= 256, "sel_attn_bwd: Dk/Dv up to 256 supported");
    NSA_CHECK_ARG(variant >= 0 && variant <= 2, "sel_attn_bwd: unknown variant %d", variant);
    hipStream_t st = (hipStream_t)stream;
    const int64_t R = (int64_t)B * S * G;
    const size_t esz = dtype == NSA_DT_F32 ? 4 : 2;
    const bool empty = R == 0 || S_kv == 0 || n_ranges == 0;
    const bool fast_ok = !empty && sel_attn_bwd_mfma_supported(dtype, h, Dk, Dv) && kss % 8 == 0 && vss % 8 == 0 && ksb % 8 == 0 &&
                         vsb % 8 == 0 && ksg % 8 == 0 && vsg % 8 == 0 && ((uintptr_t)Q % 16 == 0) && ((uintptr_t)K % 16 == 0) &&
                         ((uintptr_t)V % 16 == 0) && ((uintptr_t)dO % 16 == 0) && ((uintptr_t)O % 16 == 0) && (int64_t)B * G <= 65535 &&
                         workspace && workspace_bytes >= sel_attn_bwd_mfma_workspace(R, h, S, (int64_t)B * G, S_kv) && ((uintptr_t)workspace % 16 == 0);
    if (variant == 2) NSA_CHECK_ARG(fast_ok, "sel_attn_bwd: MFMA variant requested but shape/dtype/alignment/workspace unsupported");
    const bool fast = fast_ok && variant != 1;
    if (!fast && (int64_t)B * G * S_kv > 0) {  // the generic kernel accumulates with atomics; the MFMA route writes every element
        NSA_HIP_TRY(hipMemsetAsync(dK, 0, sizeof(float) * (size_t)B * G * S_kv * Dk, st));
        NSA_HIP_TRY(hipMemsetAsync(dV, 0, sizeof(float) * (size_t)B * G * S_kv * Dv, st));
    }
    if (R == 0) return NSA_OK;
    if (S_kv == 0 || n_ranges == 0) {
        NSA_HIP_TRY(hipMemsetAsync(dQ, 0, (size_t)R * h * Dk * esz, st));
        return NSA_OK;
    }
    NSA_CHECK_ARG(Q && K && V && ranges && O && lse && dO && dQ && dK && dV, "sel_attn_bwd: null pointer");
    SelAttnBwdParams P{};
    P.Q = Q; P.K = K; P.V = V; P.ranges = ranges; P.O = O; P.lse = lse; P.dO = dO; P.dQ = dQ; P.dK = dK; P.dV = dV;
    P.R = R; P.S = S; P.G = G; P.h = h; P.Dk = Dk; P.Dv = Dv; P.S_kv = S_kv; P.n = n_ranges;
    P.ksb = ksb; P.ksg = ksg; P.kss = kss; P.vsb = vsb; P.vsg = vsg; P.vss = vss;
    P.scale = scale > 0.f ? scale : 1.0f / sqrtf((float)Dk);
    if (fast) return launch_sel_attn_bwd_mfma(P, dtype, (float *)workspace, st);
    return launch_sel_attn_bwd_generic(P, dtype, st);
}

// ------------------------------------------------------------------------------ band attention
size_t nsa_band_attn_fwd_workspace(int B, int S, int G, int h, int Dk, int Dv, int dtype) {
    return band_attn_workspace(B, S, G, h, Dk, Dv, dtype, nullptr);
}

}  // extern "C"

int nsa::band_attn_fwd_impl(const void *Q, const void *K, const void *V, void *O, float *lse, int B, int S, int G, int h, int Dk, int Dv,
                            int S_kv, int64_t ksb, int64_t ksg, int64_t kss, int64_t vsb, int64_t vsg, int64_t vss, int t0, int a, int dd,
                            int c, int w, int dtype, float scale, int variant, void *workspace, size_t workspace_bytes, void *stream,
                            int defer, int *ns_used) {
    if (ns_used) *ns_used = 1;
    NSA_CHECK_ARG(dtype_ok(dtype), "band_attn_fwd: unknown dtype %d", dtype);
    NSA_CHECK_ARG(B >= 0 && S >= 0 && G >= 1 && h >= 1 && Dk >= 1 && Dv >= 1 && S_kv >= 0, "band_attn_fwd: negative size");
    NSA_CHECK_ARG(t0 >= 0 && dd >= 1 && w >= 0, "band_attn_fwd: need t0 >= 0, dd >= 1, w >= 0");
    NSA_CHECK_ARG(variant >= 0 && variant <= 2, "band_attn_fwd: unknown variant %d", variant);
    hipStream_t st = (hipStream_t)stream;
    const int64_t R = (int64_t)B * S * G;
    if (R == 0) return NSA_OK;
    // S_kv == 0 or w == 0: every interval is empty; the generic kernel writes the zeros / -inf rows
    NSA_CHECK_ARG(Q && O && ((K && V) || S_kv == 0), "band_attn_fwd: null pointer");
    BandAttnParams P{};
    P.Q = Q; P.K = K; P.V = V; P.O = O; P.lse = lse;
    P.B = B; P.S = S; P.G = G; P.h = h; P.Dk = Dk; P.Dv = Dv; P.S_kv = S_kv;
    P.ksb = ksb; P.ksg = ksg; P.kss = kss; P.vsb = vsb; P.vsg = vsg; P.vss = vss;
    P.scale = scale > 0.f ? scale : 1.0f / sqrtf((float)Dk);
    P.t0 = t0; P.a = a; P.dd = dd; P.c = c; P.w = w;
    const bool fast_ok = S_kv > 0 && w > 0 && band_attn_mfma_supported(dtype, h, Dk, Dv) && kss % 8 == 0 && vss % 8 == 0 && ksb % 8 == 0 &&
                         vsb % 8 == 0 && ksg % 8 == 0 && vsg % 8 == 0 && ((uintptr_t)Q % 16 == 0) && ((uintptr_t)K % 16 == 0) &&
                         ((uintptr_t)V % 16 == 0) && ((uintptr_t)O % 8 == 0) && (int64_t)S_kv * kss * 2 < ((int64_t)1 << 31) &&
                         (int64_t)S_kv * vss * 2 < ((int64_t)1 << 31);
    if (variant == 2) NSA_CHECK_ARG(fast_ok, "band_attn_fwd: MFMA variant requested but shape/dtype/alignment unsupported");
    if (fast_ok && variant != 1) {
        int ns = 1;
        const size_t need = band_attn_workspace(B, S, G, h, Dk, Dv, dtype, &ns);
        if (ns > 1 && workspace && workspace_bytes >= need && ((uintptr_t)workspace % 16 == 0)) {
            P.part = (float *)workspace;
            P.nsplit = ns;
            P.defer_combine = defer;
            if (ns_used && defer) *ns_used = ns;
        }
        return launch_band_attn_fwd_mfma(P, dtype, st);
    }
    return launch_band_attn_fwd_generic(P, dtype, st);
}


extern "C" {

int nsa_band_attn_fwd(const void *Q, const void *K, const void *V, void *O, float *lse, int B, int S, int G, int h, int Dk,
                      int Dv, int S_kv, int64_t ksb, int64_t ksg, int64_t kss, int64_t vsb, int64_t vsg, int64_t vss, int t0,
                      int a, int dd, int c, int w, int dtype, float scale, int variant, void *workspace, size_t workspace_bytes,
                      void *stream) {
    return band_attn_fwd_impl(Q, K, V, O, lse, B, S, G, h, Dk, Dv, S_kv, ksb, ksg, kss, vsb, vsg, vss, t0, a, dd, c, w, dtype, scale, variant,
                              workspace, workspace_bytes, stream, 0, nullptr);
}

// Band attention backward.  dQ: dense 48-slot kernel (MFMA route); dK/dV: the key-block-major selection backward kernels fed
// with the band as one range per row.  workspace = selection-backward workspace (its delta area is shared) + the ranges.
static size_t band_bwd_ranges_off(int B, int S, int G, int h, int Dk, int Dv, int S_kv, int dtype, int variant) {
    return (nsa_sel_attn_bwd_workspace(B, S, G, h, Dk, Dv, S_kv, dtype, variant) + 255) & ~(size_t)255;
}

size_t nsa_band_attn_bwd_workspace(int B, int S, int G, int h, int Dk, int Dv, int S_kv, int dtype, int variant) {
    return band_bwd_ranges_off(B, S, G, h, Dk, Dv, S_kv, dtype, variant) + sizeof(int32_t) * 2 * (size_t)B * S * G;
}

int nsa_band_attn_bwd(const void *Q, const void *K, const void *V, const void *O, const float *lse, const void *dO, void *dQ, float *dK,
                      float *dV, int B, int S, int G, int h, int Dk, int Dv, int S_kv, int64_t ksb, int64_t ksg, int64_t kss, int64_t vsb,
                      int64_t vsg, int64_t vss, int t0, int a, int dd, int c, int w, int dtype, float scale, int variant, void *workspace,
                      size_t workspace_bytes, void *stream) {
    NSA_CHECK_ARG(dtype_ok(dtype), "band_attn_bwd: unknown dtype %d", dtype);
    NSA_CHECK_ARG(B >= 0 && S >= 0 && G >= 1 && h >= 1 && Dk >= 1 && Dv >= 1 && S_kv >= 0, "band_attn_bwd: negative size");
    NSA_CHECK_ARG(t0 >= 0 && dd >= 1 && w >= 0, "band_attn_bwd: need t0 >= 0, dd >= 1, w >= 0");
    NSA_CHECK_ARG(variant >= 0 && variant <= 2, "band_attn_bwd: unknown variant %d", variant);
    const int64_t R = (int64_t)B * S * G;
    if (R == 0) return NSA_OK;
    hipStream_t st = (hipStream_t)stream;
    const size_t roff = band_bwd_ranges_off(B, S, G, h, Dk, Dv, S_kv, dtype, variant);
    NSA_CHECK_ARG(workspace && ((uintptr_t)workspace % 256 == 0) && workspace_bytes >= roff + sizeof(int32_t) * 2 * (size_t)R,
                  "band_attn_bwd: workspace missing, misaligned or too small");
    int32_t *ranges = (int32_t *)((unsigned char *)workspace + roff);
    if (int rc = launch_band_ranges(ranges, B, S, G, S_kv, t0, a, dd, c, w, st)) return rc;
    const bool fast_ok = S_kv > 0 && w > 0 && roff > 0 && band_attn_mfma_supported(dtype, h, Dk, Dv) && sel_attn_bwd_mfma_supported(dtype, h, Dk, Dv) &&
                         kss % 8 == 0 && vss % 8 == 0 && ksb % 8 == 0 && vsb % 8 == 0 && ksg % 8 == 0 && vsg % 8 == 0 &&
                         ((uintptr_t)Q % 16 == 0) && ((uintptr_t)K % 16 == 0) && ((uintptr_t)V % 16 == 0) && ((uintptr_t)dO % 16 == 0) &&
                         ((uintptr_t)O % 16 == 0) && (int64_t)B * G <= 65535 && (int64_t)S_kv * kss * 2 < ((int64_t)1 << 31) &&
                         (int64_t)S_kv * vss * 2 < ((int64_t)1 << 31);
    if (variant == 2) NSA_CHECK_ARG(fast_ok, "band_attn_bwd: MFMA variant requested but shape/dtype/alignment unsupported");
    if (!fast_ok || variant == 1)
        return nsa_sel_attn_bwd(Q, K, V, ranges, O, lse, dO, dQ, dK, dV, B, S, G, h, Dk, Dv, S_kv, 1, ksb, ksg, kss, vsb, vsg, vss, dtype, scale,
                                variant == 2 ? 0 : variant, workspace, roff, stream);
    NSA_CHECK_ARG(Q && K && V && O && lse && dO && dQ && dK && dV, "band_attn_bwd: null pointer");
    float *delta = (float *)workspace;
    if (int rc = launch_bwd_delta(O, dO, delta, R * h, Dv, dtype, st)) return rc;
    BandAttnParams BP{};
    BP.Q = Q; BP.K = K; BP.V = V;
    BP.B = B; BP.S = S; BP.G = G; BP.h = h; BP.Dk = Dk; BP.Dv = Dv; BP.S_kv = S_kv;
    BP.ksb = ksb; BP.ksg = ksg; BP.kss = kss; BP.vsb = vsb; BP.vsg = vsg; BP.vss = vss;
    BP.scale = scale > 0.f ? scale : 1.0f / sqrtf((float)Dk);
    BP.t0 = t0; BP.a = a; BP.dd = dd; BP.c = c; BP.w = w;
    if (int rc = launch_band_attn_bwd_dq(BP, dO, lse, delta, dQ, dtype, st)) return rc;
    SelAttnBwdParams P{};
    P.Q = Q; P.K = K; P.V = V; P.ranges = ranges; P.O = O; P.lse = lse; P.dO = dO; P.dQ = dQ; P.dK = dK; P.dV = dV;
    P.R = R; P.S = S; P.G = G; P.h = h; P.Dk = Dk; P.Dv = Dv; P.S_kv = S_kv; P.n = 1;
    P.ksb = ksb; P.ksg = ksg; P.kss = kss; P.vsb = vsb; P.vsg = vsg; P.vss = vss;
    P.scale = BP.scale;
    P.skip_delta_dq = 1;
    return launch_sel_attn_bwd_mfma(P, dtype, delta, st);
}

// ------------------------------------------------------------------------------ block meta (host)
int nsa_block_counts(int seq_len, int l, int d, int l_sel, int *S_cmp, int *S_sel, int *nnz) {
    NSA_CHECK_ARG(l > 0 && d > 0 && l_sel > 0, "Block parameters must be positive");
    NSA_CHECK_ARG(l % d == 0 && l_sel % d == 0, "Require d|l and d|l_sel in M0");  // block_index.py:75-77
    const int sc = seq_len < l ? 0 : (seq_len - l) / d + 1;
    const int ss = seq_len <= 0 ? 0 : (seq_len + l_sel - 1) / l_sel;
    if (S_cmp) *S_cmp = sc;
    if (S_sel) *S_sel = ss;
    if (nnz) {
        // cmp block i = [i d, i d + l) overlaps selection blocks floor(i d / l') .. floor((i d + l - 1) / l'), capped
        int64_t cnt = 0;
        for (int i = 0; i < sc; ++i) {
            const int j0 = (i * d) / l_sel;
            int j1 = (i * d + l - 1) / l_sel;
            if (j1 > ss - 1) j1 = ss - 1;
            if (j1 >= j0) cnt += j1 - j0 + 1;
        }
        *nnz = (int)cnt;
    }
    return NSA_OK;
}

// Closed form of build_M_csl_csr (block_index.py:43-71): O(nnz) instead of the reference's
// O(S_cmp * S_sel) double loop.  Weight = overlap / total overlap, evaluated in double and
// rounded once to fp32 exactly as torch.tensor(python floats, dtype=float32) does.
int nsa_build_block_meta_host(int seq_len, int l, int d, int l_sel, int32_t *csr_indptr, int32_t *csr_indices,
                              float *csr_values, int32_t *csc_ptr, int32_t *csc_rows, float *csc_vals) {
    int sc, ss, nnz;
    int rc = nsa_block_counts(seq_len, l, d, l_sel, &sc, &ss, &nnz);
    if (rc) return rc;
    std::vector<int32_t> ind((size_t)nnz), rows((size_t)nnz);
    std::vector<float> val((size_t)nnz);
    std::vector<int32_t> iptr((size_t)sc + 1, 0);
    int k = 0;
    for (int i = 0; i < sc; ++i) {
        const int a0 = i * d, a1 = i * d + l;
        const int j0 = a0 / l_sel;
        int j1 = (a1 - 1) / l_sel;
        if (j1 > ss - 1) j1 = ss - 1;
        int total = 0;
        for (int j = j0; j <= j1; ++j) {
            const int lo = a0 > j * l_sel ? a0 : j * l_sel;
            const int hi = a1 < (j + 1) * l_sel ? a1 : (j + 1) * l_sel;
            total += hi - lo;
        }
        for (int j = j0; j <= j1; ++j) {
            const int lo = a0 > j * l_sel ? a0 : j * l_sel;
            const int hi = a1 < (j + 1) * l_sel ? a1 : (j + 1) * l_sel;
            ind[k] = j;
            rows[k] = i;
            val[k] = (float)((double)(hi - lo) / (double)total);
            ++k;
        }
        iptr[(size_t)i + 1] = k;
    }
    if (csr_indptr) memcpy(csr_indptr, iptr.data(), sizeof(int32_t) * ((size_t)sc + 1));
    if (csr_indices && nnz) memcpy(csr_indices, ind.data(), sizeof(int32_t) * (size_t)nnz);
    if (csr_values && nnz) memcpy(csr_values, val.data(), sizeof(float) * (size_t)nnz);
    if (csc_ptr) {
        // counting sort by column; stable in the row index => ascending cmp row inside each column
        std::vector<int32_t> cnt((size_t)ss + 1, 0);
        for (int e = 0; e < nnz; ++e) cnt[(size_t)ind[e] + 1]++;
        for (int j = 0; j < ss; ++j) cnt[(size_t)j + 1] += cnt[j];
        memcpy(csc_ptr, cnt.data(), sizeof(int32_t) * ((size_t)ss + 1));
        if (csc_rows && csc_vals) {
            std::vector<int32_t> pos(cnt.begin(), cnt.end() - 1);
            for (int e = 0; e < nnz; ++e) {
                const int p = pos[ind[e]]++;
                csc_rows[p] = rows[e];
                csc_vals[p] = val[e];
            }
        }
    }
    return NSA_OK;
}

// ------------------------------------------------------------------------------ scores
int nsa_map_pcmp_to_pgrp(const float *p_cmp, int64_t R, int h, int S_cmp_cur, const int32_t *csc_ptr,
                         const int32_t *csc_rows, const float *csc_vals, int S_sel, float *p_slc, float *p_grp,
                         void *stream) {
    NSA_CHECK_ARG(p_grp && csc_ptr, "map_pcmp_to_pgrp: null pointer");
    return launch_map_pcmp(p_cmp, R, h, S_cmp_cur, csc_ptr, csc_rows, csc_vals, S_sel, p_slc, p_grp, (hipStream_t)stream);
}

int nsa_pcmp_all(const void *Q, const void *K_cmp, float *p_cmp, int B, int S, int G, int h, int Dk, int S_cmp,
                 int64_t csb, int64_t csg, int64_t css, int dtype, float scale, void *stream) {
    NSA_CHECK_ARG(dtype_ok(dtype), "pcmp_all: unknown dtype %d", dtype);
    NSA_CHECK_ARG(B >= 0 && S >= 0 && G >= 1, "pcmp_all: bad sizes");
    if (scale <= 0.f) scale = 1.0f / sqrtf((float)Dk);
    return launch_pcmp(Q, K_cmp, p_cmp, 0, (int64_t)B * S * G, S, G, h, Dk, S_cmp, csb, csg, css, dtype, scale,
                       (hipStream_t)stream);
}

// route of nsa_sel_scores for (shape, dtype, geometry, variant): 1 generic, 2 MFMA (prefill), 3 decode-shaped
static int scores_route(int B, int S, int G, int h, int Dk, int S_cmp, int S_sel, int l, int d, int l_sel, int dtype, int variant) {
    const int64_t R = (int64_t)B * S * G;
    if (variant != 0) return variant;
    if (R > 0 && S_cmp >= 1 && S_sel > 0 && R <= DECODE_MAX_ROWS && h <= 64 && (size_t)h * Dk * 4 <= 64 * 1024) return 3;
    if (scores_mfma_supported(dtype, h, Dk, l, d, l_sel) && S_cmp >= 1 && R > 0 && S_sel > 0 && (int64_t)B * G <= 65535) return 2;
    return 1;
}

size_t nsa_sel_scores_workspace(int B, int S, int G, int h, int Dk, int S_cmp, int S_sel, int l, int d, int l_sel, int dtype,
                                int variant) {
    const int64_t R = (int64_t)B * S * G;
    switch (scores_route(B, S, G, h, Dk, S_cmp, S_sel, l, d, l_sel, dtype, variant)) {
        case 2: return 0;
        case 3: return decode_scores_workspace(R, h, S_cmp);
        default: return S_cmp > 0 ? scores_workspace(R, h, S_cmp) : 0;
    }
}

int nsa_sel_scores(const void *Q, const void *K_cmp, float *p_grp, int B, int S, int G, int h, int Dk, int S_cmp,
                   int64_t csb, int64_t csg, int64_t css, const int32_t *csc_ptr, const int32_t *csc_rows,
                   const float *csc_vals, int S_sel, int l, int d, int l_sel, int causal_skip, int variant, int dtype,
                   float scale, void *workspace, size_t workspace_bytes, void *stream) {
    NSA_CHECK_ARG(dtype_ok(dtype), "sel_scores: unknown dtype %d", dtype);
    NSA_CHECK_ARG(B >= 0 && S >= 0 && G >= 1 && h >= 1, "sel_scores: bad sizes");
    NSA_CHECK_ARG(variant >= 0 && variant <= 3, "sel_scores: unknown variant %d", variant);
    if (scale <= 0.f) scale = 1.0f / sqrtf((float)Dk);
    const int route = scores_route(B, S, G, h, Dk, S_cmp, S_sel, l, d, l_sel, dtype, variant);
    if (route == 3 && S_cmp >= 1 && (int64_t)B * S * G > 0 && S_sel > 0)
        return launch_decode_scores(Q, K_cmp, p_grp, B, S, G, h, Dk, S_cmp, csb, csg, css, csc_ptr, csc_rows, csc_vals, S_sel, dtype,
                                    scale, workspace, workspace_bytes, (hipStream_t)stream);
    if (route == 2) {
        const bool ok = scores_mfma_supported(dtype, h, Dk, l, d, l_sel) && S_cmp >= 1 && (int64_t)B * S * G > 0 && S_sel > 0 &&
                        csb % 8 == 0 && csg % 8 == 0 && css % 8 == 0 && ((uintptr_t)Q % 16 == 0) && ((uintptr_t)K_cmp % 16 == 0) &&
                        (int64_t)B * G <= 65535;
        NSA_CHECK_ARG(ok, "sel_scores: MFMA route needs bf16/f16, Dk in {64,128}, h <= 16, l = 2d, l' = 4d and 16-byte aligned rows "
                          "(pass variant 1 for the generic route)");
        return launch_sel_scores_mfma(Q, K_cmp, p_grp, B, S, G, h, Dk, S_cmp, csb, csg, css, S_sel, d, dtype, scale, causal_skip,
                                      (hipStream_t)stream, nullptr, nullptr);
    }
    return launch_sel_scores(Q, K_cmp, p_grp, B, S, G, h, Dk, S_cmp, csb, csg, css, csc_ptr, csc_rows, csc_vals, S_sel,
                             dtype, scale, workspace, workspace_bytes, (hipStream_t)stream);
}

// scores + top-n ranges of every row (prefill): on the 32x32x16 scorer route the selection of a query tile runs inside the scorer launch,
// everywhere else the scorer and the select kernel are launched back to back -- the same ranges bit for bit either way
int nsa_sel_scores_select(const void *Q, const void *K_cmp, float *p_grp, int B, int S, int G, int h, int Dk, int S_cmp, int64_t csb, int64_t csg,
                          int64_t css, const int32_t *csc_ptr, const int32_t *csc_rows, const float *csc_vals, int S_sel, int l, int d,
                          int l_sel, int causal_skip, int dtype, float scale, int t0, int n_top, int force_init, int force_local, int mode,
                          int S_total, int32_t *ranges_out, int out_width, void *workspace, size_t workspace_bytes, void *stream) {
    NSA_CHECK_ARG(dtype_ok(dtype), "sel_scores_select: unknown dtype %d", dtype);
    NSA_CHECK_ARG(B >= 0 && S >= 0 && G >= 1 && h >= 1 && out_width >= 0, "sel_scores_select: bad sizes");
    const int64_t R = (int64_t)B * S * G;
    if (R == 0 || out_width == 0) return NSA_OK;
    NSA_CHECK_ARG(p_grp && ranges_out, "sel_scores_select: null pointer");
    if (scale <= 0.f) scale = 1.0f / sqrtf((float)Dk);
    const int route = scores_route(B, S, G, h, Dk, S_cmp, S_sel, l, d, l_sel, dtype, 0);
    const bool mfma_ok = route == 2 && S_cmp >= 1 && S_sel > 0 && csb % 8 == 0 && csg % 8 == 0 && css % 8 == 0 && ((uintptr_t)Q % 16 == 0) &&
                         ((uintptr_t)K_cmp % 16 == 0) && (int64_t)B * G <= 65535;
    if (mfma_ok) {
        SelectParams SP{};
        SP.p_grp = p_grp; SP.t_rows = nullptr; SP.out = ranges_out; SP.R = R; SP.S = S; SP.G = G; SP.t0 = t0;
        if (int rc = select_params_fill(&SP, S_sel, l_sel, n_top, force_init, force_local, mode, S_total, out_width)) return rc;
        int done = 0;
        if (int rc = launch_sel_scores_mfma(Q, K_cmp, p_grp, B, S, G, h, Dk, S_cmp, csb, csg, css, S_sel, d, dtype, scale, causal_skip,
                                            (hipStream_t)stream, &SP, &done))
            return rc;
        if (done) return NSA_OK;
    } else if (int rc = nsa_sel_scores(Q, K_cmp, p_grp, B, S, G, h, Dk, S_cmp, csb, csg, css, csc_ptr, csc_rows, csc_vals, S_sel, l, d, l_sel,
                                       causal_skip, route == 2 ? 1 : 0 /* rows the MFMA route cannot take: the generic one */, dtype, scale,
                                       workspace, workspace_bytes, stream)) {
        return rc;
    }
    return nsa_select_topn_ranges(p_grp, R, S, G, t0, nullptr, S_sel, l_sel, n_top, force_init, force_local, mode, S_total, ranges_out,
                                  out_width, stream);
}

// ------------------------------------------------------------------------------ selection
int nsa_batched_ranges_width(int S, int S_sel, int l_sel, int n_top, int force_init, int force_local) {
    return batched_width(S, S_sel, l_sel, n_top, force_init, force_local);
}

int nsa_select_topn_ranges(const float *p_grp, int64_t R, int S, int G, int t0, const int32_t *t_rows, int S_sel,
                           int l_sel, int n_top, int force_init, int force_local, int mode, int S_total,
                           int32_t *ranges_out, int out_width, void *stream) {
    NSA_CHECK_ARG(p_grp && ranges_out || R == 0, "select_topn_ranges: null pointer");
    return launch_select_topn(p_grp, R, S, G, t0, t_rows, S_sel, l_sel, n_top, force_init, force_local, mode, S_total,
                              ranges_out, out_width, (hipStream_t)stream);
}

int nsa_sel_select_attn_fwd(const float *p_grp, int t0, const int32_t *t_rows, int S_sel, int l_sel, int n_top, int force_init,
                            int force_local, int mode, int S_total, int32_t *ranges_out, int out_width, const void *Q, const void *K,
                            const void *V, void *O, float *lse, int B, int S, int G, int h, int Dk, int Dv, int S_kv, int64_t ksb,
                            int64_t ksg, int64_t kss, int64_t vsb, int64_t vsg, int64_t vss, int dtype, float scale, void *workspace,
                            size_t workspace_bytes, void *stream) {
    NSA_CHECK_ARG(dtype_ok(dtype), "sel_select_attn_fwd: unknown dtype %d", dtype);
    NSA_CHECK_ARG(B >= 0 && S >= 1 && G >= 1 && h >= 1 && Dk >= 1 && Dv >= 1 && S_kv >= 0 && out_width >= 0, "sel_select_attn_fwd: bad sizes");
    const int64_t R = (int64_t)B * S * G;
    if (R == 0) return NSA_OK;
    NSA_CHECK_ARG(p_grp && ranges_out, "sel_select_attn_fwd: null pointer");
    int ns = 1;
    const bool fast_ok = sel_attn_mfma_supported(dtype, h, Dk, Dv) && kss % 8 == 0 && vss % 8 == 0 && ksb % 8 == 0 && vsb % 8 == 0 &&
                         ksg % 8 == 0 && vsg % 8 == 0 && ((uintptr_t)Q % 16 == 0) && ((uintptr_t)K % 16 == 0) && ((uintptr_t)V % 16 == 0) &&
                         S_kv > 0 && out_width >= 1 && out_width <= 64 && S_sel >= 1 && S_sel <= 1024 &&
                         (sel_attn_mfma_workspace(R, h, Dv, &ns), ns == 1);
    // One launch (the selector inside the attention kernel) or two.  The fused selector runs 8 rows one after the other in a wave of a
    // kernel held to 2 waves per SIMD by its 253 VGPRs: its scalar chains run at latency and it adds 36 / 120 / 424 us at 4k x 8 / 16k x 2 /
    // 64k x 1, where the select kernel on its own (a row per wave, 6+ waves per SIMD) takes 28 / 57 / 243 us (profiles/r02
    // i_selector_share.txt): two launches are the default.
    // (never fused where the attention would split the keys over two XCD groups: that form takes its ranges from memory, and both ways of
    // calling must give the same bits)
    const bool fuse = tuning(TUNE_SEL_FUSE) > 0 && sel_attn_ksplit_workspace(dtype, h, Dk, Dv, S, S_kv, out_width, R) == 0;
    if (!fast_ok || !fuse) {  // two launches
        if (int rc = nsa_select_topn_ranges(p_grp, R, S, G, t0, t_rows, S_sel, l_sel, n_top, force_init, force_local, mode, S_total, ranges_out,
                                            out_width, stream))
            return rc;
        return nsa_sel_attn_fwd(Q, K, V, ranges_out, O, lse, B, S, G, h, Dk, Dv, S_kv, out_width, ksb, ksg, kss, vsb, vsg, vss, dtype, scale, 0,
                                workspace, workspace_bytes, stream);
    }
    NSA_CHECK_ARG(Q && K && V && O, "sel_select_attn_fwd: null pointer");
    SelectParams SP{};
    SP.p_grp = p_grp; SP.t_rows = t_rows; SP.out = ranges_out; SP.R = R; SP.S = S; SP.G = G; SP.t0 = t0;
    if (int rc = select_params_fill(&SP, S_sel, l_sel, n_top, force_init, force_local, mode, S_total, out_width)) return rc;
    SelAttnParams P{};
    P.Q = Q; P.K = K; P.V = V; P.ranges = ranges_out; P.O = O; P.lse = lse; P.R = R;
    P.S = S; P.G = G; P.h = h; P.Dk = Dk; P.Dv = Dv; P.S_kv = S_kv; P.n = out_width;
    P.ksb = ksb; P.ksg = ksg; P.kss = kss; P.vsb = vsb; P.vsg = vsg; P.vss = vss;
    P.scale = scale > 0.f ? scale : 1.0f / sqrtf((float)Dk);
    P.part = nullptr; P.nsplit = 1;
    P.fuse_select = 1;
    P.select = &SP;
    return launch_sel_attn_fwd_mfma(P, dtype, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------ fused decode step
static size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

size_t nsa_sel_decode_step_workspace(int B, int G, int h, int Dk, int Dv, int S_cmp, int S_sel, int n_top, int dtype) {
    const size_t a = align16(nsa_sel_scores_workspace(B, 1, G, h, Dk, S_cmp, S_sel, 0, 0, 0, dtype, 3));
    const size_t p = align16(sizeof(float) * (size_t)B * G * (size_t)(S_sel > 0 ? S_sel : 1));
    const size_t c = align16(nsa_sel_attn_fwd_workspace(B, 1, G, h, Dk, Dv, n_top, dtype));
    return a + p + c;
}

}  // extern "C"

int nsa::sel_decode_step_impl(const void *Q, const void *K_cmp, const void *K, const void *V, const int32_t *csc_ptr, const int32_t *csc_rows,
                              const float *csc_vals, int32_t *ranges_out, void *O, int B, int G, int h, int Dk, int Dv, int S_cmp, int S_sel,
                              int S_kv, int l, int d, int l_sel, int n_top, int t_token, int64_t kcb, int64_t kcg, int64_t kcs, int64_t ksb,
                              int64_t ksg, int64_t kss, int64_t vsb, int64_t vsg, int64_t vss, int dtype, float scale, void *workspace,
                              size_t workspace_bytes, void *stream, int defer, int *ns_used, float **part_used, const DecBandPair *band,
                              int *band_taken) {
    if (band_taken) *band_taken = 0;
    NSA_CHECK_ARG(workspace && ((uintptr_t)workspace % 16 == 0) &&
                      workspace_bytes >= nsa_sel_decode_step_workspace(B, G, h, Dk, Dv, S_cmp, S_sel, n_top, dtype),
                  "decode_step: workspace missing, misaligned or too small");
    NSA_CHECK_ARG(n_top >= 1 && n_top <= 64, "decode_step: n_top must be in [1,64]");
    unsigned char *w = (unsigned char *)workspace;
    const size_t a = align16(nsa_sel_scores_workspace(B, 1, G, h, Dk, S_cmp, S_sel, 0, 0, 0, dtype, 3));
    const size_t p = align16(sizeof(float) * (size_t)B * G * (size_t)(S_sel > 0 ? S_sel : 1));
    float *p_grp = (float *)(w + a);
    int rc;
    const float sc = scale > 0.f ? scale : 1.0f / sqrtf((float)Dk);
    const bool wg_attn = sel_attn_decode_wg_supported(dtype, h, Dk, Dv, n_top, ksb, ksg, kss, vsb, vsg, vss, Q, K, V) && S_kv >= 1 &&
                         (int64_t)S_kv * 128 < ((int64_t)1 << 31);
    const int stencil = (l == 2 * d && l_sel == 4 * d) ? 1 : 0;  // Eq.9 in closed form (the fused kernel then reads no CSC arrays)
    if (decode_step_supported((int64_t)B * G, dtype, h, Dk, Dv, S_cmp, S_sel, S_kv, l, d, l_sel, n_top, t_token, kcb, kcg, kcs, ksb, ksg, kss, vsb, vsg, vss, Q, K_cmp,
                              K, V)) {  // scores -> statistics -> Eq.9/10 -> sequential top-n -> selection attention: ONE launch, O is final
        if (ns_used) *ns_used = 1;
        if (band && band_taken) *band_taken = 1;
        return launch_decode_step(Q, K_cmp, K, V, O, ranges_out, B, G, h, S_cmp, S_sel, S_kv, n_top, t_token, kcb, kcg, kcs, ksb, ksg, kss, vsb, vsg,
                                  vss, dtype, sc, w, a, (hipStream_t)stream, band_taken ? band : nullptr);
    }
    (void)wg_attn;
    if (decode_score_select_supported(dtype, h, Dk, S_cmp, S_sel, kcb, kcg, kcs, Q, K_cmp, (int64_t)B * G)) {
        // scores -> statistics -> Eq.9/10 -> sequential top-n in one launch (bit-identical to the route below)
        rc = launch_decode_score_select(Q, K_cmp, B, G, h, Dk, S_cmp, kcb, kcg, kcs, csc_ptr, csc_rows, csc_vals, S_sel, l_sel, n_top, t_token,
                                        dtype, sc, ranges_out, (hipStream_t)stream, nullptr, stencil);
        if (rc) return rc;
    } else {
        rc = nsa_sel_scores(Q, K_cmp, p_grp, B, 1, G, h, Dk, S_cmp, kcb, kcg, kcs, csc_ptr, csc_rows, csc_vals, S_sel, l, d, l_sel, 1,
                            S_cmp >= 1 ? 3 : 1, dtype, scale, w, a, stream);
        if (rc) return rc;
        rc = nsa_select_topn_ranges(p_grp, (int64_t)B * G, 1, G, t_token, nullptr, S_sel, l_sel, n_top, 1, 2, NSA_SEL_SEQUENTIAL, 1,
                                    ranges_out, n_top, stream);
        if (rc) return rc;
    }
    if (part_used) *part_used = (float *)(w + a + p);
    return sel_attn_fwd_impl(Q, K, V, ranges_out, O, nullptr, B, 1, G, h, Dk, Dv, S_kv, n_top, ksb, ksg, kss, vsb, vsg, vss, dtype, scale, 0,
                             w + a + p, workspace_bytes - a - p, stream, defer, ns_used);
}


extern "C" {

int nsa_sel_decode_step(const void *Q, const void *K_cmp, const void *K, const void *V, const int32_t *csc_ptr,
                        const int32_t *csc_rows, const float *csc_vals, int32_t *ranges_out, void *O, int B, int G, int h, int Dk,
                        int Dv, int S_cmp, int S_sel, int S_kv, int l, int d, int l_sel, int n_top, int t_token, int64_t kcb,
                        int64_t kcg, int64_t kcs, int64_t ksb, int64_t ksg, int64_t kss, int64_t vsb, int64_t vsg, int64_t vss,
                        int dtype, float scale, void *workspace, size_t workspace_bytes, void *stream) {
    return sel_decode_step_impl(Q, K_cmp, K, V, csc_ptr, csc_rows, csc_vals, ranges_out, O, B, G, h, Dk, Dv, S_cmp, S_sel, S_kv, l, d, l_sel,
                                n_top, t_token, kcb, kcg, kcs, ksb, ksg, kss, vsb, vsg, vss, dtype, scale, workspace, workspace_bytes, stream, 0,
                                nullptr, nullptr);
}

int nsa_indices_to_ranges_v2(const int32_t *indices, int64_t R, int S, int G, int t0, int K, int S_sel, int l_sel,
                             int32_t *ranges_out, void *stream) {
    NSA_CHECK_ARG((indices && ranges_out) || R == 0 || K == 0, "indices_to_ranges_v2: null pointer");
    NSA_CHECK_ARG(S >= 1 && G >= 1, "indices_to_ranges_v2: bad sizes");
    return launch_indices_to_ranges(indices, R, S, G, t0, K, S_sel, l_sel, ranges_out, (hipStream_t)stream);
}

}  // extern "C"
