// Selection scores: p_cmp softmax (A2), Eq.9 map + Eq.10 head sum (A3/A4) and the fused scorer.
//
// Reference: compute_pcmp_all (nsa/core/selection_scorer.py:42-61), map_pcmp_to_pslc_batched
// (:89-116) and p_slc.sum(dim=3) (nsa/core/nsa_attention.py:670,1091,1570).
//
// Bit-exactness contract: given an identical fp32 p_cmp the map kernel reproduces the reference's
// CPU result bit for bit -- per (row, head, selection block) it accumulates p_cmp[r]*w over the CSC
// list in ascending compressed row r, product and sum rounded separately (the order in which the
// CPU scatter_add visits the COO entries), then sums the heads in ascending h.
#include <type_traits>

#include <stdlib.h>

#include "nsa_common.hpp"
#include "sel_attn_decode.hpp"
#include "sel_select_row.hpp"
#include "nsa_internal.hpp"

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
// Decode-shaped scorer (few query rows, long context): the work per row is one 512-KiB sweep over
// K_cmp at 64k, so it is spread over S_cmp/256 workgroups instead of one workgroup per (b,g):
//   kernel 1: thread = compressed row; logits of all h heads (log2 domain) -> workspace [R,h,S_cmp]
//   kernel 2: workgroup = query row; per-head max / sum, then per selection block the CSC taps are
//             normalised, weighted and summed over heads (any block geometry) -> p_grp [R,S_sel]
// ---------------------------------------------------------------------------------------
struct DecodeParams {
    const void *Q;   // [R,h,Dk]
    const void *Kc;  // strided
    float *x;        // [R,h,S_cmp] logits * scale * log2(e)
    float *part;     // [R,h,nchunk,2] per 64-row chunk: max, sum exp2(x - max)
    float *p_grp;    // [R,S_sel]
    const int32_t *csc_ptr, *csc_rows;
    const float *csc_vals;
    int64_t R;
    int S, G, h, Dk, S_cmp, S_sel;
    int64_t csb, csg, css;
    float c2;
    int stencil;  // fused decode kernel: l = 2d and l' = 4d, Eq.9 is the closed-form 5-tap stencil (no CSC loads)
};

constexpr int DEC_HMAX = 16;

__host__ __device__ inline int dec_nchunk(int S_cmp) { return (S_cmp + 63) / 64; }

// kernel 1: one wave per 64 compressed rows of one query row.  lane = compressed row (its K_cmp row is read with
// 16-byte loads), all heads of the group at once; the wave also leaves the chunk's (max, sum exp2) per head so
// kernel 2 never has to sweep the logits for the softmax statistics.
template <typename T>
__global__ __launch_bounds__(64) void decode_logits_kernel(DecodeParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *qs = (float *)smem;  // [h][Dk]
    constexpr int EPV = 16 / sizeof(T);  // elements per 16-byte load
    const int lane = threadIdx.x;
    const int64_t row = blockIdx.y;
    const int chunk = blockIdx.x, nchunk = dec_nchunk(P.S_cmp);
    const int g = (int)(row % P.G);
    const int b = (int)(row / ((int64_t)P.G * P.S));
    const T *q = (const T *)P.Q + row * (int64_t)P.h * P.Dk;
    for (int i = lane; i < P.h * P.Dk; i += 64) qs[i] = Elt<T>::to_f(q[i]);
    wave_lds_fence();
    const int c = chunk * 64 + lane;
    const bool valid = c < P.S_cmp;
    const T *kr = (const T *)P.Kc + (int64_t)b * P.csb + (int64_t)g * P.csg + (int64_t)(valid ? c : P.S_cmp - 1) * P.css;
    const bool vec = (P.Dk % EPV == 0) && (((uintptr_t)kr & 15) == 0) && ((P.css * sizeof(T)) % 16 == 0);
    for (int h0 = 0; h0 < P.h; h0 += DEC_HMAX) {
        const int hc = min(DEC_HMAX, P.h - h0);
        float acc[DEC_HMAX];
#pragma unroll
        for (int i = 0; i < DEC_HMAX; ++i) acc[i] = 0.f;
        if (vec) {
            for (int e0 = 0; e0 < P.Dk; e0 += EPV) {
                const u32x4 raw = *(const u32x4 *)(kr + e0);
                const T *kv = (const T *)&raw;
#pragma unroll
                for (int j = 0; j < EPV; ++j) {
                    const float kf = Elt<T>::to_f(kv[j]);
#pragma unroll
                    for (int i = 0; i < DEC_HMAX; ++i)
                        if (i < hc) acc[i] = fmaf(qs[(h0 + i) * P.Dk + e0 + j], kf, acc[i]);
                }
            }
        } else {
            for (int e = 0; e < P.Dk; ++e) {
                const float kf = Elt<T>::to_f(kr[e]);
#pragma unroll
                for (int i = 0; i < DEC_HMAX; ++i)
                    if (i < hc) acc[i] = fmaf(qs[(h0 + i) * P.Dk + e], kf, acc[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < DEC_HMAX; ++i)
            if (i < hc) {
                const float x = acc[i] * P.c2;
                if (valid) P.x[(row * P.h + h0 + i) * (int64_t)P.S_cmp + c] = x;
                const float m = wave_max(valid ? x : -INFINITY);
                const float l = wave_sum(valid ? __builtin_amdgcn_exp2f(x - m) : 0.f);
                if (lane == 0) {
                    float *pr = P.part + ((row * P.h + h0 + i) * (int64_t)nchunk + chunk) * 2;
                    pr[0] = m;
                    pr[1] = l;
                }
            }
    }
}

// kernel 1, MFMA form (bf16/f16, Dk % 32 == 0, h <= 16): the 64 compressed rows of the chunk are the MFMA rows,
// the h heads the 16 columns; K_cmp fragments are loaded straight into the A-operand layout (one pass, no reuse).
template <typename T, int KSTEPS>
__global__ __launch_bounds__(64, 2) void decode_logits_mfma_kernel(DecodeParams P) {
    typedef typename std::conditional<std::is_same<T, __bf16>::value, bf16x8, f16x8>::type x8;
    const int lane = threadIdx.x, rho = lane & 15, q = lane >> 4;
    const int64_t row = blockIdx.y;
    const int chunk = blockIdx.x, nchunk = dec_nchunk(P.S_cmp);
    const int g = (int)(row % P.G);
    const int b = (int)(row / ((int64_t)P.G * P.S));
    const int Dk = 32 * KSTEPS;
    x8 qf[KSTEPS];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
        u32x4 raw = {0u, 0u, 0u, 0u};
        if (rho < P.h) raw = *(const u32x4 *)((const T *)P.Q + (row * P.h + rho) * (int64_t)Dk + 32 * s + 8 * q);
        qf[s] = __builtin_bit_cast(x8, raw);
    }
    const T *kb = (const T *)P.Kc + (int64_t)b * P.csb + (int64_t)g * P.csg;
    f32x4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = min(chunk * 64 + 16 * u + rho, P.S_cmp - 1);
        acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            const x8 a = *(const x8 *)(kb + (int64_t)c * P.css + 32 * s + 8 * q);
            if constexpr (std::is_same<T, __bf16>::value) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[s], acc[u], 0, 0, 0);
            else acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, qf[s], acc[u], 0, 0, 0);
        }
    }
    // lane (head rho, group q) holds rows 16u + 4q + j of the chunk
    float m = -INFINITY;
    float x[16];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = chunk * 64 + 16 * u + 4 * q + j;
            const float v = acc[u][j] * P.c2;
            x[4 * u + j] = v;
            if (c < P.S_cmp) {
                if (rho < P.h) P.x[(row * P.h + rho) * (int64_t)P.S_cmp + c] = v;
                m = fmaxf(m, v);
            }
        }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (chunk * 64 + 16 * u + 4 * q + j < P.S_cmp) l += __builtin_amdgcn_exp2f(x[4 * u + j] - m);
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (q == 0 && rho < P.h) {
        float *pr = P.part + ((row * P.h + rho) * (int64_t)nchunk + chunk) * 2;
        pr[0] = m;
        pr[1] = l;
    }
}

// kernel 2: one wave per 64 selection blocks of one query row: merge the chunk statistics per head (cheap, redone by
// every wave of the row), then lane = selection block: gather the CSC taps (normalised on the fly), weight, sum over
// heads in ascending h.
__global__ __launch_bounds__(64) void decode_pgrp_kernel(DecodeParams P) {
    __shared__ float mlog_s[64];
    const int64_t row = blockIdx.y;
    const int lane = threadIdx.x;
    const int nchunk = dec_nchunk(P.S_cmp);
    // per-head softmax statistics from the chunk records.  The kernel is latency bound (a handful of waves per decode
    // step): the loads of 8 heads are issued together, the arithmetic per head is unchanged.
    for (int h0 = 0; h0 < P.h; h0 += 8) {
        if (nchunk <= 64) {
            float mv[8], lv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool ok = h0 + i < P.h && lane < nchunk;
                const float *pr = P.part + (row * P.h + min(h0 + i, P.h - 1)) * (int64_t)nchunk * 2;
                mv[i] = ok ? pr[2 * lane] : -INFINITY;
                lv[i] = ok ? pr[2 * lane + 1] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (h0 + i >= P.h) break;
                const float m = wave_max(mv[i]);
                const float l = wave_sum(lane < nchunk ? lv[i] * __builtin_amdgcn_exp2f(mv[i] - m) : 0.f);
                if (lane == 0) mlog_s[h0 + i] = m + __builtin_amdgcn_logf(l);
            }
        } else {
            for (int hh = h0; hh < min(P.h, h0 + 8); ++hh) {
                const float *pr = P.part + (row * P.h + hh) * (int64_t)nchunk * 2;
                float m = -INFINITY;
                for (int c = lane; c < nchunk; c += 64) m = fmaxf(m, pr[2 * c]);
                m = wave_max(m);
                float l = 0.f;
                for (int c = lane; c < nchunk; c += 64) l += pr[2 * c + 1] * __builtin_amdgcn_exp2f(pr[2 * c] - m);
                l = wave_sum(l);
                if (lane == 0) mlog_s[hh] = m + __builtin_amdgcn_logf(l);
            }
        }
    }
    wave_lds_fence();
    const int j = blockIdx.x * 64 + lane;
    if (j >= P.S_sel) return;
    const int k0 = P.csc_ptr[j], k1 = P.csc_ptr[j + 1];
    float grp = 0.f;
    if (k1 - k0 <= 8) {
        // taps of this selection block (ascending cmp row), then the logits of 4 heads x 8 taps per round of loads
        int rr[8];
        float vv[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const bool ok = k0 + t < k1;
            const int r = ok ? P.csc_rows[k0 + t] : P.S_cmp;
            rr[t] = r < P.S_cmp ? r : -1;
            vv[t] = ok ? P.csc_vals[k0 + t] : 0.f;
        }
        for (int h0 = 0; h0 < P.h; h0 += 4) {
            float xs[4][8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float *x = P.x + (row * P.h + min(h0 + i, P.h - 1)) * (int64_t)P.S_cmp;
#pragma unroll
                for (int t = 0; t < 8; ++t) xs[i][t] = rr[t] >= 0 ? x[rr[t]] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (h0 + i >= P.h) break;
                const float ml = mlog_s[h0 + i];
                float acc = 0.f;
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    if (rr[t] >= 0) acc = __fadd_rn(acc, __fmul_rn(__builtin_amdgcn_exp2f(xs[i][t] - ml), vv[t]));
                grp = __fadd_rn(grp, acc);
            }
        }
    } else {
        for (int hh = 0; hh < P.h; ++hh) {
            const float *x = P.x + (row * P.h + hh) * (int64_t)P.S_cmp;
            const float ml = mlog_s[hh];
            float acc = 0.f;
            for (int k = k0; k < k1; ++k) {
                const int r = P.csc_rows[k];
                if (r < P.S_cmp) acc = __fadd_rn(acc, __fmul_rn(__builtin_amdgcn_exp2f(x[r] - ml), P.csc_vals[k]));
            }
            grp = __fadd_rn(grp, acc);
        }
    }
    P.p_grp[row * (int64_t)P.S_sel + j] = grp;
}

// ---------------------------------------------------------------------------------------
// Decode, fused: logits -> statistics -> Eq.9/10 -> top-n ranges in ONE launch.  One 1024-thread workgroup per query row
// (b,g): the logits of the row ([h, S_cmp] fp32, 98 KB at 64k context) and its group scores stay in LDS, phases are separated
// by workgroup barriers.  The arithmetic of every phase is the arithmetic of decode_logits_mfma_kernel, decode_pgrp_kernel
// and select_topn_kernel (same operations, same order), so p_grp and the ranges are bit-identical to the 3-kernel route;
// it exists because a decode step is a chain of tiny launches and each one costs ~5 us of latency.
// ---------------------------------------------------------------------------------------
template <typename T, int KSTEPS, bool ATTEND>
__global__ __launch_bounds__(1024) void decode_score_select_kernel(DecodeParams P, SelectParams SP, int cand, int t_token, DecAttnArgs AT, int stop) {
    typedef typename std::conditional<std::is_same<T, __bf16>::value, bf16x8, f16x8>::type x8;
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    const int lane = threadIdx.x & 63, wave = uniform((int)(threadIdx.x >> 6)), rho = lane & 15, q = lane >> 4;
    constexpr int NW = 16;
    const int64_t row = blockIdx.x;
    const int nchunk = dec_nchunk(P.S_cmp);
    float *xs = dsm;                                   // [h][S_cmp]
    float *part = xs + (size_t)P.h * P.S_cmp;          // [h][nchunk][2]
    float *mlog_s = part + (size_t)P.h * nchunk * 2;   // [64]
    float *pg = mlog_s + 64;                           // [S_sel]
    int *scr = (int *)(pg + P.S_sel);                  // [128] run extraction scratch of the selector
    const int g = (int)(row % P.G);
    const int b = (int)(row / ((int64_t)P.G * P.S));
    constexpr int Dk = 32 * KSTEPS;
    DEC_TS(0);
    // ---- phase 0 (generic block geometry only): the Eq.9 taps of this thread's selection block go out first (csc_ptr -> rows / weights
    // are two dependent global round trips, 3 us when they start in phase 2b; here they fly behind the K_cmp loads of phase 1).  The default
    // geometry (l = 2d, l' = 4d) needs no table: block j takes rows 4j-1 .. 4j+3 with weights 1/2, 1, 1, 1, 1/2 -- the
    // CSC of a decode cache whose meta is older than its K_cmp lacks the newest rows, but those only reach the current and the previous
    // block, both forced: the ranges are the same (p_grp stays inside the kernel).
    const int j_pre = wave * 64 + lane;  // the block of this thread in the first round of phase 2b
    int k0_pre = 0, k1_pre = 0, rr_pre[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float vv_pre[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (!P.stencil) {
        const int jj = min(j_pre, P.S_sel - 1);  // unconditional loads (clamped): see taps_pre
        k0_pre = P.csc_ptr[jj];
        k1_pre = P.csc_ptr[jj + 1];
    }
    auto taps_pre = [&]() {  // second hop: issued behind the first K_cmp loads, so waiting for csc_ptr does not hold those back.  Inside a
                             // non-empty column the loads are unconditional (index clamped into the column) and their values are not
                             // looked at before phase 2b.  An EMPTY column loads nothing: the CSC of a cache whose meta dates from before
                             // the first compressed token has no entries at all (null arrays) while S_cmp is already 1.
        if (k1_pre > k0_pre) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int k = min(k0_pre + t, k1_pre - 1);
                rr_pre[t] = P.csc_rows[k];
                vv_pre[t] = P.csc_vals[k];
            }
        }
    };
    // ---- phase 1: logits of 64 compressed rows per wave and step (MFMA rows), heads = columns
    x8 qf[KSTEPS];  // columns >= h repeat the last head: their results are never stored (kept for the attention phase)
    {
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) qf[s] = *(const x8 *)((const T *)P.Q + (row * P.h + min(rho, P.h - 1)) * (int64_t)Dk + 32 * s + 8 * q);
        const T *kb = (const T *)P.Kc + (int64_t)b * P.csb + (int64_t)g * P.csg;
        auto load_chunk = [&](int chunk, x8 (&a)[4][KSTEPS]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = min(chunk * 64 + 16 * u + rho, P.S_cmp - 1);
#pragma unroll
                for (int s = 0; s < KSTEPS; ++s) a[u][s] = *(const x8 *)(kb + (int64_t)c * P.css + 32 * s + 8 * q);
            }
        };
        auto do_chunk = [&](int chunk, const x8 (&a)[4][KSTEPS]) {
            f32x4 acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KSTEPS; ++s) {
                    if constexpr (std::is_same<T, __bf16>::value) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u][s], qf[s], acc[u], 0, 0, 0);
                    else acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u][s], qf[s], acc[u], 0, 0, 0);
                }
            }
            float m = -INFINITY;
            float x[16];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = chunk * 64 + 16 * u + 4 * q + j;
                    const float v = acc[u][j] * P.c2;
                    x[4 * u + j] = v;
                    if (c < P.S_cmp) {
                        if (rho < P.h) xs[(size_t)rho * P.S_cmp + c] = v;
                        m = fmaxf(m, v);
                    }
                }
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            float l = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (chunk * 64 + 16 * u + 4 * q + j < P.S_cmp) l += __builtin_amdgcn_exp2f(x[4 * u + j] - m);
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
            if (q == 0 && rho < P.h && chunk < nchunk) {
                float *pr = part + ((size_t)rho * nchunk + chunk) * 2;
                pr[0] = m;
                pr[1] = l;
            }
        };
        // first chunk: loads unconditional and in straight-line code (a chunk past the end re-reads the last row and stores nothing),
        // the taps' second hop right behind them -- exact wait counts: the MFMAs wait for K_cmp only, not for the taps
        x8 a[4][KSTEPS];
        load_chunk(wave, a);
        __builtin_amdgcn_sched_barrier(0);  // the K_cmp loads go out before anything waits for csc_ptr
        if (!P.stencil) taps_pre();
        do_chunk(wave, a);
        for (int chunk = wave + NW; chunk < nchunk; chunk += NW) {
            load_chunk(chunk, a);
            do_chunk(chunk, a);
        }
    }
#ifdef NSA_DEC_TS
    if (stop == 1) return;  // measurement aid of the TIMELINE build (TUNE_DECODE_STOP): every thread leaves together
#endif
    DEC_TS(1);
    __syncthreads();
    DEC_TS(2);
    // ---- phase 2a: softmax statistics per head (one wave per head)
    for (int hh = wave; hh < P.h; hh += NW) {
        const float *pr = part + (size_t)hh * nchunk * 2;
        float m, l;
        if (nchunk <= 64) {
            const float mv = lane < nchunk ? pr[2 * lane] : -INFINITY;
            const float lv = lane < nchunk ? pr[2 * lane + 1] : 0.f;
            m = wave_max(mv);
            l = wave_sum(lane < nchunk ? lv * __builtin_amdgcn_exp2f(mv - m) : 0.f);
        } else {
            m = -INFINITY;
            for (int c = lane; c < nchunk; c += 64) m = fmaxf(m, pr[2 * c]);
            m = wave_max(m);
            l = 0.f;
            for (int c = lane; c < nchunk; c += 64) l += pr[2 * c + 1] * __builtin_amdgcn_exp2f(pr[2 * c] - m);
            l = wave_sum(l);
        }
        if (lane == 0) mlog_s[hh] = m + __builtin_amdgcn_logf(l);
    }
    __syncthreads();
    DEC_TS(3);
    // ---- phase 2b: Eq.9 taps + Eq.10 head sum, lane = selection block
    const int nslab = (P.S_sel + 63) >> 6;
    if (P.stencil && 2 * nslab <= NW) {
        // fewer 64-block slabs than waves (16k context: 4): a (slab, head) pair per wave instead of a slab with all its heads -- a head is
        // one dependent chain (LDS round trip, 5 exp, 5 adds) and a SIMD with one active wave runs it at latency; the head sums are then
        // added in ascending head order as before (bit-identical)
        float *acch = (float *)(scr + 128);  // [h][S_sel]
        for (int item = wave; item < nslab * P.h; item += NW) {
            const int hh = item / nslab, j = (item - hh * nslab) * 64 + lane;
            if (j >= P.S_sel) continue;
            const float *x = xs + (size_t)hh * P.S_cmp;
            const float ml = mlog_s[hh];
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                const int r = 4 * j - 1 + t;
                const bool ok = r >= 0 && r < P.S_cmp;
                const float nxt = __fadd_rn(acc, __fmul_rn(__builtin_amdgcn_exp2f(x[ok ? r : 0] - ml), (t == 0 || t == 4) ? 0.5f : 1.0f));
                acc = ok ? nxt : acc;
            }
            acch[(size_t)hh * P.S_sel + j] = acc;
        }
        __syncthreads();
        for (int jb = wave; jb < nslab; jb += NW) {
            const int j = jb * 64 + lane;
            if (j >= P.S_sel) continue;
            float grp = 0.f;
            for (int hh = 0; hh < P.h; ++hh) grp = __fadd_rn(grp, acch[(size_t)hh * P.S_sel + j]);
            pg[j] = grp;
        }
    } else
    for (int jb = wave; jb * 64 < P.S_sel; jb += NW) {
        const int j = jb * 64 + lane;
        if (j >= P.S_sel) continue;
        const bool pre = jb == wave;  // the taps fetched in phase 0
        const int k0 = P.stencil ? 0 : pre ? k0_pre : P.csc_ptr[j], k1 = P.stencil ? 5 : pre ? k1_pre : P.csc_ptr[j + 1];
        float grp = 0.f;
        if (k1 - k0 <= 8) {  // taps once, logits from LDS
            int rr[8];
            float vv[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const bool ok = k0 + t < k1;
                int r = P.S_cmp;
                float v = 0.f;
                if (P.stencil) {
                    r = ok ? 4 * j - 1 + t : P.S_cmp;
                    v = (t == 0 || t == 4) ? 0.5f : 1.0f;
                } else if (pre) {
                    r = ok ? rr_pre[t] : P.S_cmp;
                    v = ok ? vv_pre[t] : 0.f;
                } else if (ok) {
                    r = P.csc_rows[k0 + t];
                    v = P.csc_vals[k0 + t];
                }
                rr[t] = (r >= 0 && r < P.S_cmp) ? r : -1;
                vv[t] = v;
            }
            for (int hh = 0; hh < P.h; ++hh) {
                const float *x = xs + (size_t)hh * P.S_cmp;
                const float ml = mlog_s[hh];
                float acc = 0.f;
#pragma unroll
                for (int t = 0; t < 8; ++t) {  // branch-free: the 8 LDS reads of a head are in flight together (a branch per tap made each
                                               // read its own round trip: 48 of them, 2.5 us); a missing tap leaves acc untouched
                    const float nxt = __fadd_rn(acc, __fmul_rn(__builtin_amdgcn_exp2f(x[max(rr[t], 0)] - ml), vv[t]));
                    acc = rr[t] >= 0 ? nxt : acc;
                }
                grp = __fadd_rn(grp, acc);
            }
        } else {
            for (int hh = 0; hh < P.h; ++hh) {
                const float *x = xs + (size_t)hh * P.S_cmp;
                const float ml = mlog_s[hh];
                float acc = 0.f;
                for (int k = k0; k < k1; ++k) {
                    const int r = P.csc_rows[k];
                    if (r < P.S_cmp) acc = __fadd_rn(acc, __fmul_rn(__builtin_amdgcn_exp2f(x[r] - ml), P.csc_vals[k]));
                }
                grp = __fadd_rn(grp, acc);
            }
        }
        pg[j] = grp;
    }
#ifdef NSA_DEC_TS
    if (stop == 2) return;
#endif
    DEC_TS(4);
    __syncthreads();
    DEC_TS(5);
    // ---- phase 3: top-n + forced blocks + merge (one wave)
    int rs = 0, re = 0;
    if (wave == 0) {
        int32_t *out = SP.out + row * (int64_t)SP.W * 2;
        switch (cand) {
            case 1: select_topn_row_regs<1>(SP, pg, t_token, rs, re, scr); break;
            case 2: select_topn_row_regs<2>(SP, pg, t_token, rs, re, scr); break;
            case 4: select_topn_row_regs<4>(SP, pg, t_token, rs, re, scr); break;
            case 8: select_topn_row_regs<8>(SP, pg, t_token, rs, re, scr); break;
            case 16: select_topn_row_regs<16>(SP, pg, t_token, rs, re, scr); break;
            default: select_topn_row_regs<32>(SP, pg, t_token, rs, re, scr); break;
        }
        DEC_TS(6);
        if (lane < SP.W) {
            out[2 * lane] = rs;
            out[2 * lane + 1] = re;
        }
    }
    // ---- phase 4: selection attention of the row over the ranges just chosen (same workgroup: the 16 waves take the 64-key chunks,
    // partials merged through LDS -- sel_attn_decode.hpp).  The logits / scores in LDS are dead by now: their space holds the V tiles.
    (void)rs, (void)re, (void)AT;  // (the attention of the row is its own launch behind this kernel: the one-launch step is sel_decode_fused.hip)
}

static size_t decode_fused_lds(int h, int S_cmp, int S_sel) {
    const size_t split = 2 * ((S_sel + 63) / 64) <= 16 ? (size_t)h * S_sel : 0;  // per-head block sums of the head-split phase 2b
    return sizeof(float) * ((size_t)h * S_cmp + (size_t)h * dec_nchunk(S_cmp) * 2 + 64 + (size_t)S_sel + 128 + split);
}

// fused route available?  (bf16/f16 MFMA shapes, the row's logits fit in LDS)
bool decode_score_select_supported(int dtype, int h, int Dk, int S_cmp, int S_sel, int64_t csb, int64_t csg, int64_t css, const void *Q,
                                   const void *Kc, int64_t rows) {
    const int mode = tuning(TUNE_DECODE_UNFUSED);  // A/B switch for measurements and for the equivalence test: -1 auto, 0 fused, 1 unfused
    if (mode > 0) return false;
    // one workgroup per row sweeps the row's whole K_cmp: with few rows and a long context the 3-kernel route (one wave per 64 compressed
    // rows, spread over the chip) is faster (64k, B = 1: 73.6 vs 79.8 us per layer step); with >= 64 rows every CU has a row either way
    const bool long_ok = mode == 0 || S_cmp <= 2048 || rows >= 64;
    return (dtype == NSA_DT_BF16 || dtype == NSA_DT_F16) && (Dk == 64 || Dk == 128) && h <= 16 && S_cmp >= 1 && S_sel >= 1 &&
           S_sel <= 64 * 32 && css % 8 == 0 && csb % 8 == 0 && csg % 8 == 0 && ((uintptr_t)Q % 16 == 0) && ((uintptr_t)Kc % 16 == 0) && long_ok &&
           decode_fused_lds(h, S_cmp, S_sel) <= 150 * 1024;
}

int launch_decode_score_select(const void *Q, const void *Kc, int B, int G, int h, int Dk, int S_cmp, int64_t csb, int64_t csg, int64_t css,
                               const int32_t *csc_ptr, const int32_t *csc_rows, const float *csc_vals, int S_sel, int l_sel, int n_top,
                               int t_token, int dtype, float scale, int32_t *ranges_out, hipStream_t st, const DecAttnArgs *attend,
                               int stencil) {
    const int64_t R = (int64_t)B * G;
    NSA_CHECK_ARG(R <= 65535 * 32, "decode scorer: too many rows");
    DecodeParams P{Q, Kc, nullptr, nullptr, nullptr, csc_ptr, csc_rows, csc_vals, R, 1, G, h, Dk, S_cmp, S_sel, csb, csg, css, scale * LOG2E,
                   stencil && tuning(TUNE_DECODE_STENCIL) ? 1 : 0};
    SelectParams SP{};
    if (int rc = select_params_sequential(&SP, S_sel, l_sel, n_top, 1, 2, n_top)) return rc;
    SP.out = ranges_out;
    SP.R = R;
    SP.S = 1;
    SP.G = G;
    SP.t0 = t_token;
    const int c = (S_sel + 63) / 64;
    const int cand = c <= 1 ? 1 : c <= 2 ? 2 : c <= 4 ? 4 : c <= 8 ? 8 : c <= 16 ? 16 : 32;
    size_t lds = decode_fused_lds(h, S_cmp, S_sel);
    void (*k)(DecodeParams, SelectParams, int, int, DecAttnArgs, int);
    DecAttnArgs AT{};
    NSA_CHECK_ARG(attend == nullptr, "decode scorer: the attention is not part of this launch any more (sel_decode_fused.hip)");
    if (dtype == NSA_DT_BF16) k = Dk == 64 ? decode_score_select_kernel<__bf16, 2, false> : decode_score_select_kernel<__bf16, 4, false>;
    else k = Dk == 64 ? decode_score_select_kernel<_Float16, 2, false> : decode_score_select_kernel<_Float16, 4, false>;
    if (lds > 64 * 1024) {  // raise the dynamic-LDS limit once per kernel (the runtime call costs about a millisecond)
        static void *raised[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        bool done = false;
        for (void *r : raised) done |= (r == (void *)k);
        if (!done) {
            NSA_HIP_TRY(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            for (void *&r : raised)
                if (!r) {
                    r = (void *)k;
                    break;
                }
        }
    }
    hipLaunchKernelGGL(k, dim3((unsigned)R), dim3(1024), lds, st, P, SP, cand, t_token, AT, tuning(TUNE_DECODE_STOP));
    NSA_LAUNCH_CHECK("decode_score_select");
    return NSA_OK;
}

size_t decode_scores_workspace(int64_t R, int h, int S_cmp) {
    const size_t sc = (size_t)(S_cmp > 0 ? S_cmp : 1);
    const size_t three_kernel = sizeof(float) * (size_t)R * h * (sc + 2 * (size_t)dec_nchunk((int)sc));
    const size_t split_step = decode_step_workspace(R, h, (int)sc);  // the one-launch step with the logits phase split over workgroups
    return three_kernel > split_step ? three_kernel : split_step;
}

int launch_decode_scores(const void *Q, const void *Kc, float *p_grp, int B, int S, int G, int h, int Dk, int S_cmp, int64_t csb,
                         int64_t csg, int64_t css, const int32_t *csc_ptr, const int32_t *csc_rows, const float *csc_vals,
                         int S_sel, int dtype, float scale, void *ws, size_t ws_bytes, hipStream_t st) {
    const int64_t R = (int64_t)B * S * G;
    NSA_CHECK_ARG(h <= 64 && Dk >= 1 && (size_t)h * Dk * 4 <= 64 * 1024, "decode scorer: h/Dk too large");
    NSA_CHECK_ARG(ws && ws_bytes >= decode_scores_workspace(R, h, S_cmp), "decode scorer: workspace too small");
    NSA_CHECK_ARG(R <= 65535, "decode scorer: too many rows");
    DecodeParams P{Q, Kc, (float *)ws, (float *)ws + (size_t)R * h * S_cmp, p_grp, csc_ptr, csc_rows, csc_vals, R, S, G, h, Dk, S_cmp,
                   S_sel, csb, csg, css, scale * LOG2E};
    const dim3 grid1((unsigned)dec_nchunk(S_cmp), (unsigned)R);
    const size_t lds = sizeof(float) * (size_t)h * Dk;
    const bool mfma = (dtype == NSA_DT_BF16 || dtype == NSA_DT_F16) && (Dk == 64 || Dk == 128) && h <= 16 && css % 8 == 0 &&
                      csb % 8 == 0 && csg % 8 == 0 && ((uintptr_t)Q % 16 == 0) && ((uintptr_t)Kc % 16 == 0);
    if (mfma) {
        if (dtype == NSA_DT_BF16) {
            if (Dk == 64) hipLaunchKernelGGL((decode_logits_mfma_kernel<__bf16, 2>), grid1, dim3(64), 0, st, P);
            else hipLaunchKernelGGL((decode_logits_mfma_kernel<__bf16, 4>), grid1, dim3(64), 0, st, P);
        } else {
            if (Dk == 64) hipLaunchKernelGGL((decode_logits_mfma_kernel<_Float16, 2>), grid1, dim3(64), 0, st, P);
            else hipLaunchKernelGGL((decode_logits_mfma_kernel<_Float16, 4>), grid1, dim3(64), 0, st, P);
        }
    } else {
        switch (dtype) {
            case NSA_DT_F32: hipLaunchKernelGGL(decode_logits_kernel<float>, grid1, dim3(64), lds, st, P); break;
            case NSA_DT_BF16: hipLaunchKernelGGL(decode_logits_kernel<__bf16>, grid1, dim3(64), lds, st, P); break;
            case NSA_DT_F16: hipLaunchKernelGGL(decode_logits_kernel<_Float16>, grid1, dim3(64), lds, st, P); break;
            default: NSA_CHECK_ARG(false, "decode scorer: unknown dtype %d", dtype);
        }
    }
    NSA_LAUNCH_CHECK("decode_logits");
    hipLaunchKernelGGL(decode_pgrp_kernel, dim3((unsigned)((S_sel + 63) / 64), (unsigned)R), dim3(64), 0, st, P);
    NSA_LAUNCH_CHECK("decode_pgrp");
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

#ifdef NSA_DEC_TS
extern "C" __attribute__((visibility("default"))) int nsa_debug_read_ts(long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(nsa::g_dec_ts), sizeof(long long) * 64);
}
#endif
