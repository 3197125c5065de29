// Generic selection-attention kernels (any of f32/bf16/f16, any Dk/Dv <= 256, any h).
// Correctness-first VALU implementation: one wave per query row (b,t,g); fp32 math throughout,
// exact (max-subtracted) online softmax.  It is the fp32 parity path and the fall-back shape
// coverage for the MFMA kernel (sel_attn_mfma.hip); the semantics are those of the reference's
// grouped_selection_attention_masked (nsa/core/attention_kernels.py:705-772).
#include "nsa_common.hpp"
#include "sel_attn_params.hpp"

namespace nsa {

constexpr int GEN_HC = 8;  // heads processed per pass (accumulators live in registers)

// LDS per wave: seg ints | tok[64] | q[GEN_HC*Dk] f32 | p[GEN_HC*64] f32
__host__ __device__ inline size_t gen_wave_lds_bytes(int Dk) {
    return sizeof(int) * (SEG_INTS + 64) + sizeof(float) * (size_t)(GEN_HC * Dk + GEN_HC * 64);
}

// token at position p of the concatenated segments (binary search over the prefix offsets)
__device__ __forceinline__ int token_at(const int *seg, int nseg, int p) {
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (seg[2 * mid + 1] <= p) lo = mid; else hi = mid - 1;
    }
    return seg[2 * lo] + (p - seg[2 * lo + 1]);
}

template <typename T, int DVC /* ceil(Dv/64) */>
__global__ __launch_bounds__(256) void sel_attn_fwd_generic_kernel(SelAttnParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id();
    const int wave = uniform((int)(threadIdx.x >> 6));
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= P.R) return;
    const int Dk = P.Dk, Dv = P.Dv, h = P.h;
    unsigned char *wbase = smem + (size_t)wave * gen_wave_lds_bytes(Dk);
    int *seg = (int *)wbase;
    int *tok = seg + SEG_INTS;
    float *qs = (float *)(tok + 64);
    float *ps = qs + GEN_HC * Dk;

    const int g = (int)(row % P.G);
    const int b = (int)(row / ((int64_t)P.G * P.S));
    const T *Kb = (const T *)P.K + (int64_t)b * P.ksb + (int64_t)g * P.ksg;
    const T *Vb = (const T *)P.V + (int64_t)b * P.vsb + (int64_t)g * P.vsg;
    const T *Qr = (const T *)P.Q + row * (int64_t)h * Dk;
    T *Or = (T *)P.O + row * (int64_t)h * Dv;

    int nseg;
    const int L = normalise_ranges(P.ranges + row * (int64_t)P.n * 2, P.n, P.S_kv, seg, &nseg);

    for (int h0 = 0; h0 < h; h0 += GEN_HC) {
        const int hc = min(GEN_HC, h - h0);
        for (int i = lane; i < hc * Dk; i += 64) qs[i] = Elt<T>::to_f(Qr[(int64_t)h0 * Dk + i]);
        wave_lds_fence();
        float m[GEN_HC], l[GEN_HC], acc[GEN_HC][DVC];
#pragma unroll
        for (int i = 0; i < GEN_HC; ++i) {
            m[i] = -INFINITY;
            l[i] = 0.f;
#pragma unroll
            for (int c = 0; c < DVC; ++c) acc[i][c] = 0.f;
        }
        for (int p0 = 0; p0 < L; p0 += 64) {
            const int p = p0 + lane;
            const bool valid = p < L;
            const int t = valid ? token_at(seg, nseg, p) : 0;
            tok[lane] = t;
            // scores for this lane's key, all heads of the chunk
            float sc[GEN_HC];
#pragma unroll
            for (int i = 0; i < GEN_HC; ++i) sc[i] = 0.f;
            const T *kr = Kb + (int64_t)t * P.kss;
            for (int e = 0; e < Dk; ++e) {
                float kv = Elt<T>::to_f(kr[e]);
#pragma unroll
                for (int i = 0; i < GEN_HC; ++i)
                    if (i < hc) sc[i] = fmaf(qs[i * Dk + e], kv, sc[i]);
            }
#pragma unroll
            for (int i = 0; i < GEN_HC; ++i) {
                if (i < hc) {
                    float s = valid ? sc[i] * P.scale : -INFINITY;
                    float tm = wave_max(s);  // tile has >=1 valid key -> finite
                    float mn = fmaxf(m[i], tm);
                    float alpha = __expf(m[i] - mn);  // m=-inf first time -> 0
                    float pe = valid ? __expf(s - mn) : 0.f;
                    l[i] = l[i] * alpha + wave_sum(pe);
                    m[i] = mn;
#pragma unroll
                    for (int c = 0; c < DVC; ++c) acc[i][c] *= alpha;
                    ps[i * 64 + lane] = pe;
                }
            }
            wave_lds_fence();
            const int nk = min(64, L - p0);
            for (int k = 0; k < nk; ++k) {
                const T *vr = Vb + (int64_t)tok[k] * P.vss;
#pragma unroll
                for (int c = 0; c < DVC; ++c) {
                    const int dv = c * 64 + lane;
                    float v = (dv < Dv) ? Elt<T>::to_f(vr[dv]) : 0.f;
#pragma unroll
                    for (int i = 0; i < GEN_HC; ++i)
                        if (i < hc) acc[i][c] = fmaf(ps[i * 64 + k], v, acc[i][c]);
                }
            }
            wave_lds_fence();
        }
#pragma unroll
        for (int i = 0; i < GEN_HC; ++i) {
            if (i < hc) {
                const float inv = (L > 0) ? 1.f / l[i] : 0.f;
#pragma unroll
                for (int c = 0; c < DVC; ++c) {
                    const int dv = c * 64 + lane;
                    if (dv < Dv) Or[(int64_t)(h0 + i) * Dv + dv] = Elt<T>::from_f(acc[i][c] * inv);
                }
                if (P.lse && lane == 0) P.lse[row * h + h0 + i] = (L > 0) ? m[i] + __logf(l[i]) : -INFINITY;
            }
        }
        wave_lds_fence();
    }
}

// ---------------------------------------------------------------------------------------
// Backward (generic): one wave per query row, recompute P from the forward's LSE.
//   dQ[h,:]  = sum_k dS[h,k] K[k,:]         (written, dtype T)
//   dK[k,:] += sum_h dS[h,k] Q[h,:]         (fp32 atomics, 256-B contiguous per wave instruction)
//   dV[k,:] += sum_h P[h,k] dO[h,:]
//   dS = P * (dP - delta) * scale,  dP[h,k] = dO[h,:].V[k,:],  delta[h] = dO[h,:].O[h,:]
// Follows the structure of the reference's analytic backward
// (nsa/kernels/triton_sel_kernel/__init__.py:163-231) without its first-key quirk (:217-219).
// ---------------------------------------------------------------------------------------
// LDS per wave: seg | tok[64] | q[HC*Dk] | do[HC*Dv] | p[HC*64] | ds[HC*64]
__host__ __device__ inline size_t gen_bwd_wave_lds_bytes(int Dk, int Dv) {
    return sizeof(int) * (SEG_INTS + 64) + sizeof(float) * (size_t)(GEN_HC * Dk + GEN_HC * Dv + 2 * GEN_HC * 64);
}

template <typename T, int DC /* ceil(max(Dk,Dv)/64) */>
__global__ __launch_bounds__(256) void sel_attn_bwd_generic_kernel(SelAttnBwdParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id();
    const int wave = uniform((int)(threadIdx.x >> 6));
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= P.R) return;
    const int Dk = P.Dk, Dv = P.Dv, h = P.h;
    unsigned char *wbase = smem + (size_t)wave * gen_bwd_wave_lds_bytes(Dk, Dv);
    int *seg = (int *)wbase;
    int *tok = seg + SEG_INTS;
    float *qs = (float *)(tok + 64);
    float *dos = qs + GEN_HC * Dk;
    float *ps = dos + GEN_HC * Dv;
    float *dss = ps + GEN_HC * 64;

    const int g = (int)(row % P.G);
    const int b = (int)(row / ((int64_t)P.G * P.S));
    const T *Kb = (const T *)P.K + (int64_t)b * P.ksb + (int64_t)g * P.ksg;
    const T *Vb = (const T *)P.V + (int64_t)b * P.vsb + (int64_t)g * P.vsg;
    float *dKb = P.dK + ((int64_t)b * P.G + g) * (int64_t)P.S_kv * Dk;
    float *dVb = P.dV + ((int64_t)b * P.G + g) * (int64_t)P.S_kv * Dv;
    const T *Qr = (const T *)P.Q + row * (int64_t)h * Dk;
    const T *Or = (const T *)P.O + row * (int64_t)h * Dv;
    const T *dOr = (const T *)P.dO + row * (int64_t)h * Dv;
    T *dQr = (T *)P.dQ + row * (int64_t)h * Dk;

    int nseg;
    const int L = normalise_ranges(P.ranges + row * (int64_t)P.n * 2, P.n, P.S_kv, seg, &nseg);

    for (int h0 = 0; h0 < h; h0 += GEN_HC) {
        const int hc = min(GEN_HC, h - h0);
        for (int i = lane; i < hc * Dk; i += 64) qs[i] = Elt<T>::to_f(Qr[(int64_t)h0 * Dk + i]);
        for (int i = lane; i < hc * Dv; i += 64) dos[i] = Elt<T>::to_f(dOr[(int64_t)h0 * Dv + i]);
        float delta[GEN_HC], lse[GEN_HC], dq[GEN_HC][DC];
#pragma unroll
        for (int i = 0; i < GEN_HC; ++i) {
            float part = 0.f;
            if (i < hc)
                for (int e = lane; e < Dv; e += 64)
                    part += Elt<T>::to_f(dOr[(int64_t)(h0 + i) * Dv + e]) * Elt<T>::to_f(Or[(int64_t)(h0 + i) * Dv + e]);
            delta[i] = wave_sum(part);
            lse[i] = (i < hc) ? P.lse[row * h + h0 + i] : 0.f;
#pragma unroll
            for (int c = 0; c < DC; ++c) dq[i][c] = 0.f;
        }
        wave_lds_fence();
        for (int p0 = 0; p0 < L; p0 += 64) {
            const int p = p0 + lane;
            const bool valid = p < L;
            const int t = valid ? token_at(seg, nseg, p) : 0;
            tok[lane] = t;
            float sc[GEN_HC], dp[GEN_HC];
#pragma unroll
            for (int i = 0; i < GEN_HC; ++i) sc[i] = dp[i] = 0.f;
            const T *kr = Kb + (int64_t)t * P.kss;
            const T *vr = Vb + (int64_t)t * P.vss;
            for (int e = 0; e < Dk; ++e) {
                float kv = Elt<T>::to_f(kr[e]);
#pragma unroll
                for (int i = 0; i < GEN_HC; ++i)
                    if (i < hc) sc[i] = fmaf(qs[i * Dk + e], kv, sc[i]);
            }
            for (int e = 0; e < Dv; ++e) {
                float vv = Elt<T>::to_f(vr[e]);
#pragma unroll
                for (int i = 0; i < GEN_HC; ++i)
                    if (i < hc) dp[i] = fmaf(dos[i * Dv + e], vv, dp[i]);
            }
#pragma unroll
            for (int i = 0; i < GEN_HC; ++i) {
                if (i < hc) {
                    float pe = valid ? __expf(sc[i] * P.scale - lse[i]) : 0.f;
                    ps[i * 64 + lane] = pe;
                    dss[i * 64 + lane] = pe * (dp[i] - delta[i]) * P.scale;
                }
            }
            wave_lds_fence();
            const int nk = min(64, L - p0);
            for (int k = 0; k < nk; ++k) {
                const int tk = tok[k];
                const T *kr2 = Kb + (int64_t)tk * P.kss;
#pragma unroll
                for (int c = 0; c < DC; ++c) {
                    const int d = c * 64 + lane;
                    if (d < Dk) {
                        float kv = Elt<T>::to_f(kr2[d]);
                        float dk = 0.f;
#pragma unroll
                        for (int i = 0; i < GEN_HC; ++i)
                            if (i < hc) {
                                float ds = dss[i * 64 + k];
                                dq[i][c] = fmaf(ds, kv, dq[i][c]);
                                dk = fmaf(ds, qs[i * Dk + d], dk);
                            }
                        atomicAdd(dKb + (int64_t)tk * Dk + d, dk);
                    }
                    if (d < Dv) {
                        float dv = 0.f;
#pragma unroll
                        for (int i = 0; i < GEN_HC; ++i)
                            if (i < hc) dv = fmaf(ps[i * 64 + k], dos[i * Dv + d], dv);
                        atomicAdd(dVb + (int64_t)tk * Dv + d, dv);
                    }
                }
            }
            wave_lds_fence();
        }
#pragma unroll
        for (int i = 0; i < GEN_HC; ++i)
            if (i < hc) {
#pragma unroll
                for (int c = 0; c < DC; ++c) {
                    const int d = c * 64 + lane;
                    if (d < Dk) dQr[(int64_t)(h0 + i) * Dk + d] = Elt<T>::from_f(dq[i][c]);
                }
            }
        wave_lds_fence();
    }
}

// ---- host launchers ---------------------------------------------------------------------
template <typename T>
static int launch_fwd_generic_t(const SelAttnParams &P, hipStream_t st) {
    const int dvc = (P.Dv + 63) / 64;
    const size_t lds = 4 * gen_wave_lds_bytes(P.Dk);
    NSA_CHECK_ARG(lds <= 160 * 1024, "generic kernel: Dk=%d needs %zu B LDS", P.Dk, lds);
    const unsigned grid = (unsigned)((P.R + 3) / 4);
    void (*k)(SelAttnParams) = nullptr;
    switch (dvc) {
        case 1: k = sel_attn_fwd_generic_kernel<T, 1>; break;
        case 2: k = sel_attn_fwd_generic_kernel<T, 2>; break;
        case 3: k = sel_attn_fwd_generic_kernel<T, 3>; break;
        case 4: k = sel_attn_fwd_generic_kernel<T, 4>; break;
        default: NSA_CHECK_ARG(false, "generic kernel supports Dv <= 256 (got %d)", P.Dv);
    }
    if (lds > 64 * 1024) NSA_HIP_TRY(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, P);
    NSA_LAUNCH_CHECK("sel_attn_fwd_generic");
    return NSA_OK;
}

int launch_sel_attn_fwd_generic(const SelAttnParams &P, int dtype, hipStream_t st) {
    switch (dtype) {
        case NSA_DT_F32: return launch_fwd_generic_t<float>(P, st);
        case NSA_DT_BF16: return launch_fwd_generic_t<__bf16>(P, st);
        case NSA_DT_F16: return launch_fwd_generic_t<_Float16>(P, st);
    }
    set_error("unknown dtype %d", dtype);
    return NSA_ERR_INVALID;
}

template <typename T>
static int launch_bwd_generic_t(const SelAttnBwdParams &P, hipStream_t st) {
    const int dc = (max(P.Dk, P.Dv) + 63) / 64;
    const size_t lds = 4 * gen_bwd_wave_lds_bytes(P.Dk, P.Dv);
    NSA_CHECK_ARG(lds <= 160 * 1024, "generic bwd kernel: Dk=%d Dv=%d needs %zu B LDS", P.Dk, P.Dv, lds);
    const unsigned grid = (unsigned)((P.R + 3) / 4);
    void (*k)(SelAttnBwdParams) = nullptr;
    switch (dc) {
        case 1: k = sel_attn_bwd_generic_kernel<T, 1>; break;
        case 2: k = sel_attn_bwd_generic_kernel<T, 2>; break;
        case 3: k = sel_attn_bwd_generic_kernel<T, 3>; break;
        case 4: k = sel_attn_bwd_generic_kernel<T, 4>; break;
        default: NSA_CHECK_ARG(false, "generic bwd kernel supports D <= 256");
    }
    if (lds > 64 * 1024) NSA_HIP_TRY(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, P);
    NSA_LAUNCH_CHECK("sel_attn_bwd_generic");
    return NSA_OK;
}

int launch_sel_attn_bwd_generic(const SelAttnBwdParams &P, int dtype, hipStream_t st) {
    switch (dtype) {
        case NSA_DT_F32: return launch_bwd_generic_t<float>(P, st);
        case NSA_DT_BF16: return launch_bwd_generic_t<__bf16>(P, st);
        case NSA_DT_F16: return launch_bwd_generic_t<_Float16>(P, st);
    }
    set_error("unknown dtype %d", dtype);
    return NSA_ERR_INVALID;
}

// ---- parity mode of the reference's packed / gather executors (attention_kernels.py:181-226, 273-388) ----------------------
// Those call SDPA with is_causal=True and ONE query, which lets the query see only the first gathered key: every head's output is
// V[b,g,first token of the first non-empty range] (slot order), zeros when the row has no range.  Kept as an opt-in parity mode.
template <typename E>
__global__ __launch_bounds__(256) void sel_first_key_kernel(const E *__restrict__ V, const int32_t *__restrict__ ranges, E *__restrict__ O,
                                                            int64_t R, int S, int G, int h, int Dv, int n, int S_kv, int64_t vsb, int64_t vsg,
                                                            int64_t vss) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const int g = (int)(row % G);
    const int64_t b = row / ((int64_t)S * G);
    int s = 0, e = 0;
    if (lane < n) {
        s = ranges[(row * n + lane) * 2 + 0];
        e = ranges[(row * n + lane) * 2 + 1];
        s = min(max(s, 0), S_kv);
        e = min(max(e, s), S_kv);
    }
    const unsigned long long live = __ballot(e > s);
    E *o = O + row * (int64_t)h * Dv;
    if (live == 0ull) {
        for (int i = lane; i < h * Dv; i += 64) o[i] = E(0);
        return;
    }
    const int tok = __shfl(s, __ffsll((long long)live) - 1, 64);
    const E *v = V + b * vsb + g * vsg + (int64_t)tok * vss;
    for (int i = lane; i < h * Dv; i += 64) o[i] = v[i % Dv];
}

int launch_sel_first_key(const void *V, const int32_t *ranges, void *O, int64_t R, int S, int G, int h, int Dv, int n, int S_kv, int64_t vsb,
                         int64_t vsg, int64_t vss, int esz, hipStream_t st) {
    const unsigned grid = (unsigned)((R + 3) / 4);
    if (esz == 4)
        hipLaunchKernelGGL(sel_first_key_kernel<uint32_t>, dim3(grid), dim3(256), 0, st, (const uint32_t *)V, ranges, (uint32_t *)O, R, S, G, h,
                           Dv, n, S_kv, vsb, vsg, vss);
    else
        hipLaunchKernelGGL(sel_first_key_kernel<uint16_t>, dim3(grid), dim3(256), 0, st, (const uint16_t *)V, ranges, (uint16_t *)O, R, S, G, h,
                           Dv, n, S_kv, vsb, vsg, vss);
    NSA_LAUNCH_CHECK("sel_first_key");
    return NSA_OK;
}

// ---- parity mode of NSAAttention._sdpa_over_ranges (nsa_attention.py:1779-1855): the reference's decode / sequential-prefill gather route ----
// It gathers the union of the clamped ranges in ascending token order and calls SDPA(is_causal=True) with the h heads of the group in
// the QUERY-LENGTH position (q is [1,h,Dk]), so the top-left aligned causal mask lets head i see the first i+1 gathered tokens only.
// One wave per row; fp32 math; a row without tokens gives zeros (the reference feeds one zero key/value).
constexpr int HC_MAX = 16;  // heads (= visible tokens) supported

template <typename T>
__global__ __launch_bounds__(256) void sel_head_causal_kernel(SelAttnParams P) {
    __shared__ int s_seg[4][SEG_INTS];
    __shared__ int s_tok[4][HC_MAX];
    __shared__ float s_p[4][HC_MAX * HC_MAX];
    const int lane = lane_id();
    const int wave = uniform((int)(threadIdx.x >> 6));
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= P.R) return;
    const int Dk = P.Dk, Dv = P.Dv, h = P.h;
    int *seg = s_seg[wave];
    int *tok = s_tok[wave];
    float *pp = s_p[wave];
    const int g = (int)(row % P.G);
    const int b = (int)(row / ((int64_t)P.G * P.S));
    const T *Kb = (const T *)P.K + (int64_t)b * P.ksb + (int64_t)g * P.ksg;
    const T *Vb = (const T *)P.V + (int64_t)b * P.vsb + (int64_t)g * P.vsg;
    const T *Qr = (const T *)P.Q + row * (int64_t)h * Dk;
    T *Or = (T *)P.O + row * (int64_t)h * Dv;
    int nseg;
    const int L = normalise_ranges(P.ranges + row * (int64_t)P.n * 2, P.n, P.S_kv, seg, &nseg);
    const int Lh = min(L, h);  // tokens any head can see
    if (Lh == 0) {
        for (int i = lane; i < h * Dv; i += 64) Or[i] = Elt<T>::from_f(0.f);
        return;
    }
    if (lane < Lh) tok[lane] = token_at(seg, nseg, lane);
    wave_lds_fence();
    // scaled logits of the visible (head i, token j <= i) pairs
    for (int p = lane; p < h * Lh; p += 64) {
        const int i = p / Lh, j = p - i * Lh;
        float s = -INFINITY;
        if (j <= i) {
            const T *q = Qr + (int64_t)i * Dk;
            const T *k = Kb + (int64_t)tok[j] * P.kss;
            s = 0.f;
            for (int d = 0; d < Dk; ++d) s = fmaf(Elt<T>::to_f(q[d]), Elt<T>::to_f(k[d]), s);
            s *= P.scale;
        }
        pp[i * HC_MAX + j] = s;
    }
    wave_lds_fence();
    if (lane < h) {  // softmax of head `lane` over its visible tokens
        const int nv = min(lane + 1, Lh);
        float m = -INFINITY, l = 0.f;
        for (int j = 0; j < nv; ++j) m = fmaxf(m, pp[lane * HC_MAX + j]);
        for (int j = 0; j < nv; ++j) l += __expf(pp[lane * HC_MAX + j] - m);
        for (int j = 0; j < Lh; ++j) pp[lane * HC_MAX + j] = j < nv ? __expf(pp[lane * HC_MAX + j] - m) / l : 0.f;
    }
    wave_lds_fence();
    for (int o = lane; o < h * Dv; o += 64) {
        const int i = o / Dv, d = o - i * Dv;
        float acc = 0.f;
        for (int j = 0; j < Lh; ++j) acc = fmaf(pp[i * HC_MAX + j], Elt<T>::to_f(Vb[(int64_t)tok[j] * P.vss + d]), acc);
        Or[o] = Elt<T>::from_f(acc);
    }
}

int launch_sel_head_causal(const SelAttnParams &P, int dtype, hipStream_t st) {
    NSA_CHECK_ARG(P.h <= HC_MAX, "sel_attn_head_causal_parity: at most %d heads per group (got %d)", HC_MAX, P.h);
    const unsigned grid = (unsigned)((P.R + 3) / 4);
    switch (dtype) {
        case NSA_DT_F32: hipLaunchKernelGGL(sel_head_causal_kernel<float>, dim3(grid), dim3(256), 0, st, P); break;
        case NSA_DT_BF16: hipLaunchKernelGGL(sel_head_causal_kernel<__bf16>, dim3(grid), dim3(256), 0, st, P); break;
        default: hipLaunchKernelGGL(sel_head_causal_kernel<_Float16>, dim3(grid), dim3(256), 0, st, P); break;
    }
    NSA_LAUNCH_CHECK("sel_head_causal");
    return NSA_OK;
}

}  // namespace nsa
