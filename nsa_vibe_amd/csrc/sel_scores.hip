// Selection scores: p_cmp softmax (A2), Eq.9 map + Eq.10 head sum (A3/A4) and the fused scorer.
//
// Reference: compute_pcmp_all (nsa/core/selection_scorer.py:42-61), map_pcmp_to_pslc_batched
// (:89-116) and p_slc.sum(dim=3) (nsa/core/nsa_attention.py:670,1091,1570).
//
// Bit-exactness contract: given an identical fp32 p_cmp the map kernel reproduces the reference's
// CPU result bit for bit -- per (row, head, selection block) it accumulates p_cmp[r]*w over the CSC
// list in ascending compressed row r, product and sum rounded separately (the order in which the
// CPU scatter_add visits the COO entries), then sums the heads in ascending h.
#include "nsa_common.hpp"

namespace nsa {

// ---------------------------------------------------------------------------------------
// Eq.9 + Eq.10.  thread = (row, selection block j); p_cmp [R,h,S_cmp_cur].
// ---------------------------------------------------------------------------------------
struct MapParams {
    const float *p_cmp;
    const int32_t *csc_ptr, *csc_rows;
    const float *csc_vals;
    float *p_slc, *p_grp;
    int64_t R;
    int h, S_cmp_cur, S_sel;
};

__global__ __launch_bounds__(256) void map_pcmp_kernel(MapParams P) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P.R * P.S_sel) return;
    const int64_t row = idx / P.S_sel;
    const int j = (int)(idx % P.S_sel);
    const int k0 = P.csc_ptr[j], k1 = P.csc_ptr[j + 1];
    float grp = 0.f;
    for (int hh = 0; hh < P.h; ++hh) {
        const float *pc = P.p_cmp + (row * P.h + hh) * (int64_t)P.S_cmp_cur;
        float acc = 0.f;
        for (int k = k0; k < k1; ++k) {
            const int r = P.csc_rows[k];
            if (r < P.S_cmp_cur) acc = __fadd_rn(acc, __fmul_rn(pc[r], P.csc_vals[k]));
        }
        if (P.p_slc) P.p_slc[(row * P.h + hh) * (int64_t)P.S_sel + j] = acc;
        grp = __fadd_rn(grp, acc);
    }
    P.p_grp[idx] = grp;
}

int launch_map_pcmp(const float *p_cmp, int64_t R, int h, int S_cmp_cur, const int32_t *csc_ptr,
                    const int32_t *csc_rows, const float *csc_vals, int S_sel, float *p_slc, float *p_grp,
                    hipStream_t st) {
    NSA_CHECK_ARG(R >= 0 && h >= 1 && S_cmp_cur >= 0 && S_sel >= 0, "map: bad sizes");
    if (R == 0 || S_sel == 0) return NSA_OK;
    MapParams P{p_cmp, csc_ptr, csc_rows, csc_vals, p_slc, p_grp, R, h, S_cmp_cur, S_sel};
    const int64_t n = R * S_sel;
    hipLaunchKernelGGL(map_pcmp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, P);
    NSA_LAUNCH_CHECK("map_pcmp");
    return NSA_OK;
}

// ---------------------------------------------------------------------------------------
// p_cmp: one wave per (row, head): logits over S_cmp columns -> softmax, fp32 out.
// VALU implementation (each lane owns columns lane, lane+64, ...); used by the parity API
// nsa_pcmp_all and, query-chunked, by the first fused scorer.
// ---------------------------------------------------------------------------------------
struct PcmpParams {
    const void *Q;   // [R,h,Dk]
    const void *Kc;  // [B,G,S_cmp,Dk] strided
    float *p_cmp;    // [Rchunk,h,S_cmp]
    int64_t row0, nrows;  // rows [row0,row0+nrows) of the full R = B*S*G
    int S, G, h, Dk, S_cmp;
    int64_t csb, csg, css;
    float scale;
};

template <typename T>
__global__ __launch_bounds__(256) void pcmp_kernel(PcmpParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id();
    const int wave = uniform((int)(threadIdx.x >> 6));
    const int64_t wid = (int64_t)blockIdx.x * 4 + wave;  // (local row, head)
    if (wid >= P.nrows * P.h) return;
    const int64_t lrow = wid / P.h;
    const int hh = (int)(wid % P.h);
    const int64_t row = P.row0 + lrow;
    const int g = (int)(row % P.G);
    const int b = (int)(row / ((int64_t)P.G * P.S));
    const int Dk = P.Dk;
    float *qs = (float *)smem + (size_t)wave * Dk;
    const T *q = (const T *)P.Q + (row * P.h + hh) * (int64_t)Dk;
    for (int e = lane; e < Dk; e += 64) qs[e] = Elt<T>::to_f(q[e]);
    wave_lds_fence();
    const T *kb = (const T *)P.Kc + (int64_t)b * P.csb + (int64_t)g * P.csg;
    float *out = P.p_cmp + (lrow * P.h + hh) * (int64_t)P.S_cmp;
    float mx = -INFINITY;
    for (int c = lane; c < P.S_cmp; c += 64) {
        const T *kr = kb + (int64_t)c * P.css;
        float acc = 0.f;
        for (int e = 0; e < Dk; ++e) acc = fmaf(qs[e], Elt<T>::to_f(kr[e]), acc);
        acc *= P.scale;
        out[c] = acc;
        mx = fmaxf(mx, acc);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int c = lane; c < P.S_cmp; c += 64) {
        const float e = expf(out[c] - mx);
        out[c] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    for (int c = lane; c < P.S_cmp; c += 64) out[c] = out[c] / sum;
}

int launch_pcmp(const void *Q, const void *Kc, float *p_cmp, int64_t row0, int64_t nrows, int S, int G, int h,
                int Dk, int S_cmp, int64_t csb, int64_t csg, int64_t css, int dtype, float scale, hipStream_t st) {
    NSA_CHECK_ARG(h >= 1 && Dk >= 1 && Dk <= 4096, "pcmp: bad h/Dk");
    if (nrows == 0 || S_cmp == 0) return NSA_OK;
    PcmpParams P{Q, Kc, p_cmp, row0, nrows, S, G, h, Dk, S_cmp, csb, csg, css, scale};
    const int64_t waves = nrows * h;
    const unsigned grid = (unsigned)((waves + 3) / 4);
    const size_t lds = 4 * sizeof(float) * (size_t)Dk;
    switch (dtype) {
        case NSA_DT_F32: hipLaunchKernelGGL(pcmp_kernel<float>, dim3(grid), dim3(256), lds, st, P); break;
        case NSA_DT_BF16: hipLaunchKernelGGL(pcmp_kernel<__bf16>, dim3(grid), dim3(256), lds, st, P); break;
        case NSA_DT_F16: hipLaunchKernelGGL(pcmp_kernel<_Float16>, dim3(grid), dim3(256), lds, st, P); break;
        default: NSA_CHECK_ARG(false, "pcmp: unknown dtype %d", dtype);
    }
    NSA_LAUNCH_CHECK("pcmp");
    return NSA_OK;
}

// ---------------------------------------------------------------------------------------
// Fused scorer v1: query-chunked composition (p_cmp chunk in the workspace, never the whole
// [B,S,G,h,S_cmp] tensor -- the reference's O(S*S_cmp) memory wall,
// docs/NSA_CHUNKED_SELECTION_SPEC.md:15-19).
// ---------------------------------------------------------------------------------------
constexpr size_t SCORES_WS_TARGET = (size_t)256 << 20;

size_t scores_workspace(int64_t R, int h, int S_cmp) {
    const size_t per_row = sizeof(float) * (size_t)h * (size_t)(S_cmp > 0 ? S_cmp : 1);
    size_t rows = SCORES_WS_TARGET / per_row;
    if (rows < 1) rows = 1;
    if ((int64_t)rows > R) rows = (size_t)(R > 0 ? R : 1);
    return rows * per_row;
}

int launch_sel_scores(const void *Q, const void *Kc, float *p_grp, int B, int S, int G, int h, int Dk, int S_cmp,
                      int64_t csb, int64_t csg, int64_t css, const int32_t *csc_ptr, const int32_t *csc_rows,
                      const float *csc_vals, int S_sel, int dtype, float scale, void *ws, size_t ws_bytes,
                      hipStream_t st) {
    const int64_t R = (int64_t)B * S * G;
    if (R == 0 || S_sel == 0) return NSA_OK;
    if (S_cmp == 0) {  // selection_scorer.py:97-98 -> zeros
        NSA_HIP_TRY(hipMemsetAsync(p_grp, 0, sizeof(float) * (size_t)R * S_sel, st));
        return NSA_OK;
    }
    const size_t per_row = sizeof(float) * (size_t)h * (size_t)S_cmp;
    NSA_CHECK_ARG(ws != nullptr && ws_bytes >= per_row, "scores: workspace too small (%zu < %zu)", ws_bytes, per_row);
    const int64_t chunk = (int64_t)(ws_bytes / per_row);
    for (int64_t r0 = 0; r0 < R; r0 += chunk) {
        const int64_t nr = (R - r0 < chunk) ? R - r0 : chunk;
        int rc = launch_pcmp(Q, Kc, (float *)ws, r0, nr, S, G, h, Dk, S_cmp, csb, csg, css, dtype, scale, st);
        if (rc) return rc;
        rc = launch_map_pcmp((const float *)ws, nr, h, S_cmp, csc_ptr, csc_rows, csc_vals, S_sel, nullptr,
                             p_grp + r0 * S_sel, st);
        if (rc) return rc;
    }
    return NSA_OK;
}

}  // namespace nsa
