// Layer-level pieces of NSAAttention around the three attention branches (the "next" rows of the scope table):
//   * small-M linear (decode projections): out[M,N] = A[M,K] . W[N,K]^T
//   * RoPE + KV-cache append of a fused QKV projection (reference: nsa_attention.py:552-572 decode, :1002-1024 prefill;
//     rope.py:16-51)
//   * compressed-token emission phi = mean over l raw tokens of RoPE'd K and raw V (compress_pool.py:9-38,
//     nsa_attention.py:588-604)
//   * gate MLP + 3-branch combine (nsa_attention.py:32-82, 85-124)
// Arithmetic follows the PyTorch operator chain the reference runs, including where it rounds to the activation dtype
// (rnd() below), so a bf16 module gives the same numbers whether these kernels or the eager ops are used.
#include <type_traits>

#include "attn_mfma_tiles.hpp"
#include "layer_fused.hpp"
#include "layer_gate.hpp"

namespace nsa {

// rotation of pair index i (of D/2) at position pos: (sin, cos) rounded to the activation dtype as the eager chain has them
template <typename T>
__device__ __forceinline__ void rope_sincos(int i, int D, float pos, float base, float inv_scale, float &sn, float &cs) {
    const float e = (-2.0f * (float)i) / (float)D;
    const float inv_freq = powf(base, e);
    const float ang = (pos * inv_scale) * inv_freq;
    sincosf(ang, &sn, &cs);
    sn = rnd<T>(sn);
    cs = rnd<T>(cs);
}
template <typename T>
__device__ __forceinline__ void rope_rotate(float x0, float x1, float sn, float cs, float &r0, float &r1) {
    r0 = rnd<T>(rnd<T>(x0 * cs) - rnd<T>(x1 * sn));
    r1 = rnd<T>(rnd<T>(x0 * sn) + rnd<T>(x1 * cs));
}
// rotate the pair (x0, x1) of pair index i (of D/2) at position pos
template <typename T>
__device__ __forceinline__ void rope_pair(float x0, float x1, int i, int D, float pos, float base, float inv_scale, float &r0, float &r1) {
    float sn, cs;
    rope_sincos<T>(i, D, pos, base, inv_scale, sn, cs);
    rope_rotate<T>(x0, x1, sn, cs, r0, r1);
}

// 8 elements of the activation dtype held raw (one 16-byte load) -> floats
template <typename T>
__device__ __forceinline__ void raw8(const u32x4 &raw, float (&out)[8]) {
    static_assert(sizeof(T) == 2, "raw8: 16-bit element types");
    const T *e = (const T *)&raw;
#pragma unroll
    for (int j = 0; j < 8; ++j) out[j] = Elt<T>::to_f(e[j]);
}
// sum of squares of a row held raw in up to two chunks per lane, with row_rms's rounding (chunk c valid when has[c])
template <typename T>
__device__ __forceinline__ float row_rms_raw(const u32x4 (&x)[2], bool has1, int K, float eps) {
    float acc = 0.f, v[8];
    raw8<T>(x[0], v);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += rnd<T>(v[j] * v[j]);
    if (has1) {
        raw8<T>(x[1], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += rnd<T>(v[j] * v[j]);
    }
    float r = rnd<T>(wave_sum(acc) / (float)K);
    r = rnd<T>(r + eps);
    return rnd<T>(1.0f / sqrtf(r));
}

// ------------------------------------------------------------------------------------------ small-M linear
// one wave per output column n (W row n stays in registers), rows of A in chunks of 8
// RMSNorm folded into a small-M projection: rsqrt(mean(x^2) + eps) of one row, computed by the whole wave with the rounding points
// of the eager chain (llama_block_nsa.py:16-19); the caller then feeds rnd(rnd(x * r) * g) into its dot products
template <typename T>
__device__ __forceinline__ float row_rms(const T *xr, int K, bool vec, float eps) {
    float acc = 0.f;
    for (int k = lane_id() * 8; k < K; k += 512) {
        float v[8];
        load8<T>(xr + k, K - k, vec, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += rnd<T>(v[j] * v[j]);
    }
    float r = rnd<T>(wave_sum(acc) / (float)K);
    r = rnd<T>(r + eps);
    return rnd<T>(1.0f / sqrtf(r));
}

template <typename T>
__device__ __forceinline__ float linear_epilogue(float acc, int epi, const T *res, int64_t idx) {
    float v = rnd<T>(acc);  // the GEMM result in the activation dtype, then the fused elementwise op rounds again like the eager chain
    if (epi == 1) v = rnd<T>(v / (1.f + expf(-v)));       // silu
    else if (epi == 2) v = v + Elt<T>::to_f(res[idx]);  // + residual
    return v;
}

template <typename T>
__global__ __launch_bounds__(256) void linear_small_kernel(const T *__restrict__ A, const T *__restrict__ W, T *__restrict__ out, int M,
                                                           int N, int K, int epi, const T *__restrict__ res, const T *__restrict__ norm_w,
                                                           float norm_eps) {
    const int lane = lane_id();
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const T *w = W + (int64_t)n * K;
    const bool vec = (K % 8 == 0) && (((uintptr_t)A | (uintptr_t)W) % 16 == 0);
    for (int m0 = 0; m0 < M; m0 += 8) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int mm = min(8, M - m0);
        float rms[8];
        if (norm_w)
            for (int r = 0; r < mm; ++r) rms[r] = row_rms<T>(A + (int64_t)(m0 + r) * K, K, vec, norm_eps);
        for (int k = lane * 8; k < K; k += 512) {
            float wv[8], av[8], gv[8];
            load8<T>(w + k, K - k, vec, wv);
            if (norm_w) load8<T>(norm_w + k, K - k, vec && ((uintptr_t)norm_w % 16 == 0), gv);
            for (int r = 0; r < mm; ++r) {
                load8<T>(A + (int64_t)(m0 + r) * K + k, K - k, vec, av);
                if (norm_w) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) av[j] = rnd<T>(rnd<T>(av[j] * rms[r]) * gv[j]);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[r] = fmaf(wv[j], av[j], acc[r]);
            }
        }
        for (int r = 0; r < mm; ++r) {
            const float s = wave_sum(acc[r]);
            if (lane == 0) out[(int64_t)(m0 + r) * N + n] = Elt<T>::from_f(linear_epilogue<T>(s, epi, res, (int64_t)(m0 + r) * N + n));
        }
    }
}

// GEMV form for 1-2 rows and many columns (LM head): one wave = 4 output columns, so 4 weight rows are in flight per wave and the
// weight matrix streams at memory speed; x (normalised on the fly when norm_w is given) is read once per wave
template <typename T>
__global__ __launch_bounds__(256) void linear_gemv4_kernel(const T *__restrict__ A, const T *__restrict__ W, T *__restrict__ out, int M, int N,
                                                           int K, int epi, const T *__restrict__ res, const T *__restrict__ norm_w,
                                                           float norm_eps) {
    const int lane = lane_id();
    const int n0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    if (n0 >= N) return;
    const bool vec = (K % 8 == 0) && (((uintptr_t)A | (uintptr_t)W) % 16 == 0);
    float rms[2] = {1.f, 1.f};
    if (norm_w)
        for (int r = 0; r < M; ++r) rms[r] = row_rms<T>(A + (int64_t)r * K, K, vec, norm_eps);
    float acc[4][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    for (int k = lane * 8; k < K; k += 512) {
        float wv[4][8], av[2][8], gv[8];
#pragma unroll
        for (int c = 0; c < 4; ++c) load8<T>(W + (int64_t)min(n0 + c, N - 1) * K + k, K - k, vec, wv[c]);
        if (norm_w) load8<T>(norm_w + k, K - k, vec && ((uintptr_t)norm_w % 16 == 0), gv);
        for (int r = 0; r < M; ++r) {
            load8<T>(A + (int64_t)r * K + k, K - k, vec, av[r]);
            if (norm_w) {
#pragma unroll
                for (int j = 0; j < 8; ++j) av[r][j] = rnd<T>(rnd<T>(av[r][j] * rms[r]) * gv[j]);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[c][r] = fmaf(wv[c][j], av[r][j], acc[c][r]);
        }
    }
    for (int r = 0; r < M; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float sum = wave_sum(acc[c][r]);
            if (lane == 0 && n0 + c < N) out[(int64_t)r * N + n0 + c] = Elt<T>::from_f(linear_epilogue<T>(sum, epi, res, (int64_t)r * N + n0 + c));
        }
}

// Latency forms of the two kernels above for 1-2 rows of 16-bit activations with K <= 512 NC (16-byte aligned rows): a decode step is a
// chain of such launches and each is as long as its chain of dependent memory round trips -- the generic k loops wait for the loads of
// one 512-element chunk before they issue the next (fc2 at K = 3072: six round trips in a row).  Here every load of the wave (weights,
// norm weights, the rows of A) goes out before the first use; the arithmetic is the generic kernels' in the same order (same bits).
template <typename T, int NC>
__global__ __launch_bounds__(256) void linear_small_fast_kernel(const T *__restrict__ A, const T *__restrict__ W, T *__restrict__ out, int M, int N,
                                                                int K, int epi, const T *__restrict__ res, const T *__restrict__ norm_w,
                                                                float norm_eps) {
    const int lane = lane_id();
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const T *w = W + (int64_t)n * K;
    u32x4 wr[NC], gr[NC], ar[2][NC];
    bool has[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int k = lane * 8 + 512 * c;
        has[c] = k < K;
        const int kc = min(k, K - 8);
        wr[c] = *(const u32x4 *)(w + kc);
        if (norm_w) gr[c] = *(const u32x4 *)(norm_w + kc);
#pragma unroll
        for (int r = 0; r < 2; ++r)
            if (r < M) ar[r][c] = *(const u32x4 *)(A + (int64_t)r * K + kc);
    }
    float rms[2] = {1.f, 1.f};
    if (norm_w) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
            if (r < M) {
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    if (has[c]) {
                        float v[8];
                        raw8<T>(ar[r][c], v);
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc += rnd<T>(v[j] * v[j]);
                    }
                float q = rnd<T>(wave_sum(acc) / (float)K);
                q = rnd<T>(q + norm_eps);
                rms[r] = rnd<T>(1.0f / sqrtf(q));
            }
    }
    float acc[2] = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NC; ++c)
        if (has[c]) {
            float wv[8], gv[8], av[8];
            raw8<T>(wr[c], wv);
            if (norm_w) raw8<T>(gr[c], gv);
#pragma unroll
            for (int r = 0; r < 2; ++r)
                if (r < M) {
                    raw8<T>(ar[r][c], av);
                    if (norm_w) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) av[j] = rnd<T>(rnd<T>(av[j] * rms[r]) * gv[j]);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[r] = fmaf(wv[j], av[j], acc[r]);
                }
        }
#pragma unroll
    for (int r = 0; r < 2; ++r)
        if (r < M) {
            const float sum = wave_sum(acc[r]);
            if (lane == 0) out[(int64_t)r * N + n] = Elt<T>::from_f(linear_epilogue<T>(sum, epi, res, (int64_t)r * N + n));
        }
}

template <typename T, int NC>
__global__ __launch_bounds__(256) void linear_gemv4_fast_kernel(const T *__restrict__ A, const T *__restrict__ W, T *__restrict__ out, int M, int N,
                                                                int K, int epi, const T *__restrict__ res, const T *__restrict__ norm_w,
                                                                float norm_eps) {
    const int lane = lane_id();
    const int n0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    if (n0 >= N) return;
    u32x4 wr[4][NC], gr[NC], ar[2][NC];
    bool has[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int k = lane * 8 + 512 * c;
        has[c] = k < K;
        const int kc = min(k, K - 8);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) wr[cc][c] = *(const u32x4 *)(W + (int64_t)min(n0 + cc, N - 1) * K + kc);
        if (norm_w) gr[c] = *(const u32x4 *)(norm_w + kc);
#pragma unroll
        for (int r = 0; r < 2; ++r)
            if (r < M) ar[r][c] = *(const u32x4 *)(A + (int64_t)r * K + kc);
    }
    float rms[2] = {1.f, 1.f};
    if (norm_w) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
            if (r < M) {
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    if (has[c]) {
                        float v[8];
                        raw8<T>(ar[r][c], v);
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc += rnd<T>(v[j] * v[j]);
                    }
                float q = rnd<T>(wave_sum(acc) / (float)K);
                q = rnd<T>(q + norm_eps);
                rms[r] = rnd<T>(1.0f / sqrtf(q));
            }
    }
    float acc[4][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
    for (int c = 0; c < NC; ++c)
        if (has[c]) {
            float gv[8], av[2][8];
            if (norm_w) raw8<T>(gr[c], gv);
#pragma unroll
            for (int r = 0; r < 2; ++r)
                if (r < M) {
                    raw8<T>(ar[r][c], av[r]);
                    if (norm_w) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) av[r][j] = rnd<T>(rnd<T>(av[r][j] * rms[r]) * gv[j]);
                    }
                }
#pragma unroll
            for (int r = 0; r < 2; ++r)
                if (r < M) {
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        float wv[8];
                        raw8<T>(wr[cc][c], wv);
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[cc][r] = fmaf(wv[j], av[r][j], acc[cc][r]);
                    }
                }
        }
#pragma unroll
    for (int r = 0; r < 2; ++r)
        if (r < M) {
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const float sum = wave_sum(acc[cc][r]);
                if (lane == 0 && n0 + cc < N) out[(int64_t)r * N + n0 + cc] = Elt<T>::from_f(linear_epilogue<T>(sum, epi, res, (int64_t)r * N + n0 + cc));
            }
        }
}

// decode, few rows: the three-branch mix as the A operand of the output projection -- A[r, k] = g_cmp O_cmp + g_sel O_sel + g_win O_win of
// group k / (K / G) with the gate probabilities gates[r G + g][3] (evaluated by the launch that produced the branches), rounded to the
// activation dtype exactly where the mix kernel rounds (mix3), so this is decode_finish + linear_small in one launch, same bits
template <typename T, int NC>
__global__ __launch_bounds__(256) void linear_small_mix_kernel(const T *__restrict__ Oc, const T *__restrict__ Os, const T *__restrict__ Ow,
                                                               const float *__restrict__ gates, const T *__restrict__ W, T *__restrict__ out, int M,
                                                               int N, int K, int G, int epi, const T *__restrict__ res) {
    const int lane = lane_id();
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const T *w = W + (int64_t)n * K;
    const int kpg = K / G;
    // (M <= 2, K <= 512 NC: every load of the wave out before the first use, like linear_small_fast_kernel)
    u32x4 wr[NC], ocr[2][NC], osr[2][NC], owr[2][NC];
    float pr[2][NC][3];
    bool has[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int k = lane * 8 + 512 * c;
        has[c] = k < K;
        const int kc = min(k, K - 8), g = kc / kpg;
        wr[c] = *(const u32x4 *)(w + kc);
#pragma unroll
        for (int r = 0; r < 2; ++r)
            if (r < M) {
                const int64_t o = (int64_t)r * K + kc;
                ocr[r][c] = *(const u32x4 *)(Oc + o);
                osr[r][c] = *(const u32x4 *)(Os + o);
                owr[r][c] = *(const u32x4 *)(Ow + o);
                const float *gp = gates + ((int64_t)r * G + g) * 3;
#pragma unroll
                for (int i = 0; i < 3; ++i) pr[r][c][i] = gp[i];
            }
    }
    float acc[2] = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NC; ++c)
        if (has[c]) {
            float wv[8];
            raw8<T>(wr[c], wv);
#pragma unroll
            for (int r = 0; r < 2; ++r)
                if (r < M) {
                    float oc[8], os[8], ow[8];
                    raw8<T>(ocr[r][c], oc);
                    raw8<T>(osr[r][c], os);
                    raw8<T>(owr[r][c], ow);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[r] = fmaf(wv[j], rnd<T>(mix3<T>(pr[r][c], oc[j], os[j], ow[j])), acc[r]);
                }
        }
#pragma unroll
    for (int r = 0; r < 2; ++r)
        if (r < M) {
            const float sum = wave_sum(acc[r]);
            if (lane == 0) out[(int64_t)r * N + n] = Elt<T>::from_f(linear_epilogue<T>(sum, epi, res, (int64_t)r * N + n));
        }
}

template <typename T>
static int linear_small_t(const void *A, const void *W, void *out, int M, int N, int K, int epi, const void *res, const void *norm_w, float eps,
                          hipStream_t st) {
    if constexpr (sizeof(T) == 2) {
        const bool fast = M <= 2 && K >= 8 && K % 8 == 0 && K <= 4096 && (((uintptr_t)A | (uintptr_t)W) % 16 == 0) &&
                          (!norm_w || (uintptr_t)norm_w % 16 == 0);
        if (fast) {
            const int nc = (K + 511) / 512;
            const bool g4 = N >= 4096 && nc <= 2;
            const dim3 grid(g4 ? (unsigned)((N + 15) / 16) : (unsigned)((N + 3) / 4));
#define NSA_LSF(KERN, NC_) hipLaunchKernelGGL((KERN<T, NC_>), grid, dim3(256), 0, st, (const T *)A, (const T *)W, (T *)out, M, N, K, epi, (const T *)res, (const T *)norm_w, eps)
            if (g4) NSA_LSF(linear_gemv4_fast_kernel, 2);
            else if (nc <= 2) NSA_LSF(linear_small_fast_kernel, 2);
            else if (nc <= 4) NSA_LSF(linear_small_fast_kernel, 4);
            else if (nc <= 6) NSA_LSF(linear_small_fast_kernel, 6);
            else NSA_LSF(linear_small_fast_kernel, 8);
#undef NSA_LSF
            NSA_LAUNCH_CHECK("linear_small(fast)");
            return NSA_OK;
        }
    }
    if (M <= 2 && N >= 4096)
        hipLaunchKernelGGL(linear_gemv4_kernel<T>, dim3((unsigned)((N + 15) / 16)), dim3(256), 0, st, (const T *)A, (const T *)W, (T *)out, M, N, K,
                           epi, (const T *)res, (const T *)norm_w, eps);
    else
        hipLaunchKernelGGL(linear_small_kernel<T>, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, (const T *)A, (const T *)W, (T *)out, M, N, K, epi,
                           (const T *)res, (const T *)norm_w, eps);
    NSA_LAUNCH_CHECK("linear_small");
    return NSA_OK;
}
static int launch_linear_small_valu(const void *A, const void *W, void *out, int M, int N, int K, int dtype, int epi, const void *res,
                                    const void *norm_w, float eps, hipStream_t st) {
    if (dtype == NSA_DT_F32) return linear_small_t<float>(A, W, out, M, N, K, epi, res, norm_w, eps, st);
    if (dtype == NSA_DT_BF16) return linear_small_t<__bf16>(A, W, out, M, N, K, epi, res, norm_w, eps, st);
    return linear_small_t<_Float16>(A, W, out, M, N, K, epi, res, norm_w, eps, st);
}

// ------------------------------------------------------------------------------------------ RoPE + cache append
// prefill form.  A thread owns ONE column pair (its rotation frequency, destination tensor and column are computed once) and
// walks the tokens: consecutive threads = consecutive columns, so loads and stores stay coalesced.
template <typename T>
__global__ __launch_bounds__(256) void rope_cache_append_kernel(RopeAppendParams P) {
    const int NQ = P.G * P.h * P.Dk, GK = P.G * P.Dk, GV = P.G * P.Dv;
    const int NT = NQ + 3 * GK + 3 * GV;
    const int cp = blockIdx.x * 256 + threadIdx.x;
    if (cp >= NT / 2) return;
    const int col = 2 * cp;
    // destination: base pointer + per-batch and per-token strides (elements), rotation: pair index i of D_rope (or none)
    T *dst;
    int64_t sb, ss;
    int ri = -1, rD = 1;
    if (col < NQ) {
        dst = (T *)P.Q_out + col;
        sb = (int64_t)P.S * NQ;
        ss = NQ;
        ri = col >> 1;
        rD = NQ;
    } else {
        int c = col - NQ;
        const int pairw = GK + GV;
        const int sp = c / pairw;
        c -= sp * pairw;
        const bool isv = c >= GK;
        if (isv) c -= GK;
        const int D = isv ? P.Dv : P.Dk;
        const int g = c / D, dc = c - g * D;
        dst = (T *)P.cache[2 * sp + (isv ? 1 : 0)] + ((int64_t)g * P.S_max + P.t0) * D + dc;
        sb = (int64_t)P.G * P.S_max * D;
        ss = D;
        if (!isv && sp < 2) {
            ri = dc >> 1;
            rD = P.Dk;
        }
    }
    const float inv_freq = ri >= 0 ? powf(P.rope_base, (-2.0f * (float)ri) / (float)rD) : 0.f;
    // positions outside, sequences inside: the rotation (sincosf: most of this kernel's arithmetic) depends on the position only
    for (int s = blockIdx.y; s < P.S; s += gridDim.y) {
        float sn = 0.f, cs = 1.f;
        if (ri >= 0) {
            const float ang = ((float)(P.t0 + s) * P.inv_scale) * inv_freq;
            sincosf(ang, &sn, &cs);
            sn = rnd<T>(sn);
            cs = rnd<T>(cs);
        }
        for (int b = blockIdx.z; b < P.B; b += gridDim.z) {
            const T *src = (const T *)P.proj + ((int64_t)b * P.S + s) * NT + col;
            float x0 = Elt<T>::to_f(src[0]), x1 = Elt<T>::to_f(src[1]);
            if (ri >= 0) {
                const float r0 = rnd<T>(rnd<T>(x0 * cs) - rnd<T>(x1 * sn));
                x1 = rnd<T>(rnd<T>(x0 * sn) + rnd<T>(x1 * cs));
                x0 = r0;
            }
            T *d = dst + b * sb + s * ss;
            d[0] = Elt<T>::from_f(x0);
            d[1] = Elt<T>::from_f(x1);
        }
    }
}

int launch_rope_cache_append(const RopeAppendParams &P, int dtype, hipStream_t st) {
    const int NT = P.G * P.h * P.Dk + 3 * P.G * P.Dk + 3 * P.G * P.Dv;
    const int64_t ntok = (int64_t)P.B * P.S;
    if (ntok == 0) return NSA_OK;
    // (rows of positions x slices of the batch: about 2048 of them; a thread shares one rotation between the sequences of its slice)
    const unsigned gy = (unsigned)std::min<int64_t>(P.S, 2048);
    const unsigned gz = (unsigned)std::max<int64_t>(1, std::min<int64_t>(P.B, 2048 / gy));
    const dim3 grid((unsigned)((NT / 2 + 255) / 256), gy, gz);
    if (dtype == NSA_DT_F32) hipLaunchKernelGGL(rope_cache_append_kernel<float>, grid, dim3(256), 0, st, P);
    else if (dtype == NSA_DT_BF16) hipLaunchKernelGGL(rope_cache_append_kernel<__bf16>, grid, dim3(256), 0, st, P);
    else hipLaunchKernelGGL(rope_cache_append_kernel<_Float16>, grid, dim3(256), 0, st, P);
    NSA_LAUNCH_CHECK("rope_cache_append");
    return NSA_OK;
}

// backward of rope_cache_append: d(proj) from dQ [B,S,NQ] and the six cache-slice gradients [B,G,S,D] (null = zero).  The
// rotation is orthogonal, so the gradient of a rotated pair is the pair rotated back by the same angle.
template <typename T>
__global__ __launch_bounds__(256) void rope_cache_append_bwd_kernel(RopeAppendParams P) {
    const int NQ = P.G * P.h * P.Dk, GK = P.G * P.Dk, GV = P.G * P.Dv;
    const int NT = NQ + 3 * GK + 3 * GV;
    const int cp = blockIdx.x * 256 + threadIdx.x;
    if (cp >= NT / 2) return;
    const int col = 2 * cp;
    const T *src;  // gradient of the forward's destination
    int64_t sb, ss;
    int ri = -1, rD = 1;
    if (col < NQ) {
        src = (const T *)P.Q_out + col;
        sb = (int64_t)P.S * NQ;
        ss = NQ;
        ri = col >> 1;
        rD = NQ;
    } else {
        int c = col - NQ;
        const int pairw = GK + GV;
        const int sp = c / pairw;
        c -= sp * pairw;
        const bool isv = c >= GK;
        if (isv) c -= GK;
        const int D = isv ? P.Dv : P.Dk;
        const int g = c / D, dc = c - g * D;
        const T *base = (const T *)P.cache[2 * sp + (isv ? 1 : 0)];  // [B,G,S,D] gradient tensor (S rows, not S_max)
        src = base ? base + (int64_t)g * P.S * D + dc : nullptr;
        sb = (int64_t)P.G * P.S * D;
        ss = D;
        if (!isv && sp < 2) {
            ri = dc >> 1;
            rD = P.Dk;
        }
    }
    const float inv_freq = ri >= 0 ? powf(P.rope_base, (-2.0f * (float)ri) / (float)rD) : 0.f;
    const int64_t ntok = (int64_t)P.B * P.S;
    for (int64_t row = blockIdx.y; row < ntok; row += gridDim.y) {
        const int b = (int)(row / P.S), s = (int)(row - (int64_t)b * P.S);
        float g0 = 0.f, g1 = 0.f;
        if (src) {
            const T *p = src + b * sb + s * ss;
            g0 = Elt<T>::to_f(p[0]);
            g1 = Elt<T>::to_f(p[1]);
        }
        if (ri >= 0) {
            const float ang = ((float)(P.t0 + s) * P.inv_scale) * inv_freq;
            float sn, cs;
            sincosf(ang, &sn, &cs);
            sn = rnd<T>(sn);
            cs = rnd<T>(cs);
            const float r0 = g0 * cs + g1 * sn;  // y0 = x0 c - x1 s, y1 = x0 s + x1 c  =>  dx0 = g0 c + g1 s, dx1 = -g0 s + g1 c
            g1 = g1 * cs - g0 * sn;
            g0 = r0;
        }
        T *d = (T *)P.proj + row * NT + col;
        d[0] = Elt<T>::from_f(g0);
        d[1] = Elt<T>::from_f(g1);
    }
}

int launch_rope_cache_append_bwd(const RopeAppendParams &P, int dtype, hipStream_t st) {
    const int NT = P.G * P.h * P.Dk + 3 * P.G * P.Dk + 3 * P.G * P.Dv;
    const int64_t ntok = (int64_t)P.B * P.S;
    if (ntok == 0) return NSA_OK;
    const dim3 grid((unsigned)((NT / 2 + 255) / 256), (unsigned)std::min<int64_t>(ntok, 2048));
    if (dtype == NSA_DT_F32) hipLaunchKernelGGL(rope_cache_append_bwd_kernel<float>, grid, dim3(256), 0, st, P);
    else if (dtype == NSA_DT_BF16) hipLaunchKernelGGL(rope_cache_append_bwd_kernel<__bf16>, grid, dim3(256), 0, st, P);
    else hipLaunchKernelGGL(rope_cache_append_bwd_kernel<_Float16>, grid, dim3(256), 0, st, P);
    NSA_LAUNCH_CHECK("rope_cache_append_bwd");
    return NSA_OK;
}

// decode form: the fused QKV projection and the RoPE + cache append in one kernel.  One wave per PAIR of adjacent output
// columns (a rotation pair), rows of x in chunks of 8; lane r of the wave finishes row r of the chunk.
template <typename T>
__device__ __forceinline__ void rope_store_pair(const RopeAppendParams &P, int b, int s, int col, float x0, float x1) {
    const int NQ = P.G * P.h * P.Dk, GK = P.G * P.Dk, GV = P.G * P.Dv;
    const float pos = (float)(P.t0 + s);
    if (col < NQ) {
        rope_pair<T>(x0, x1, col >> 1, NQ, pos, P.rope_base, P.inv_scale, x0, x1);
        T *dst = (T *)P.Q_out + ((int64_t)b * P.S + s) * NQ + col;
        dst[0] = Elt<T>::from_f(x0);
        dst[1] = Elt<T>::from_f(x1);
        return;
    }
    int c = col - NQ;
    const int pairw = GK + GV;
    const int sp = c / pairw;
    c -= sp * pairw;
    const bool isv = c >= GK;
    if (isv) c -= GK;
    const int D = isv ? P.Dv : P.Dk;
    const int g = c / D, dc = c - g * D;
    if (!isv && sp < 2) rope_pair<T>(x0, x1, dc >> 1, P.Dk, pos, P.rope_base, P.inv_scale, x0, x1);
    T *dst = (T *)P.cache[2 * sp + (isv ? 1 : 0)] + (((int64_t)b * P.G + g) * P.S_max + (P.t0 + s)) * D + dc;
    dst[0] = Elt<T>::from_f(x0);
    dst[1] = Elt<T>::from_f(x1);
}

template <typename T>
__global__ __launch_bounds__(256, 2) void qkv_rope_append_kernel(RopeAppendParams P, const T *__restrict__ X, const T *__restrict__ W, int K,
                                                              const T *__restrict__ norm_w, float norm_eps) {
    const int lane = lane_id();
    const int NT = P.G * P.h * P.Dk + 3 * P.G * P.Dk + 3 * P.G * P.Dv;
    const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (2 * pair >= NT) return;
    const T *w0 = W + (int64_t)(2 * pair) * K, *w1 = w0 + K;
    const bool vec = (K % 8 == 0) && (((uintptr_t)X | (uintptr_t)W) % 16 == 0);
    const int M = P.B;  // S == 1
    for (int m0 = 0; m0 < M; m0 += 8) {
        float a0[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, a1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int mm = min(8, M - m0);
        float rms[8];
        if (norm_w)
            for (int r = 0; r < mm; ++r) rms[r] = row_rms<T>(X + (int64_t)(m0 + r) * K, K, vec, norm_eps);
        for (int k = lane * 8; k < K; k += 512) {
            float u0[8], u1[8], xv[8], gv[8];
            load8<T>(w0 + k, K - k, vec, u0);
            load8<T>(w1 + k, K - k, vec, u1);
            if (norm_w) load8<T>(norm_w + k, K - k, vec && ((uintptr_t)norm_w % 16 == 0), gv);
            for (int r = 0; r < mm; ++r) {
                load8<T>(X + (int64_t)(m0 + r) * K + k, K - k, vec, xv);
                if (norm_w) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) xv[j] = rnd<T>(rnd<T>(xv[j] * rms[r]) * gv[j]);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    a0[r] = fmaf(u0[j], xv[j], a0[r]);
                    a1[r] = fmaf(u1[j], xv[j], a1[r]);
                }
            }
        }
        float x0 = 0.f, x1 = 0.f;
        for (int r = 0; r < mm; ++r) {
            const float s0 = wave_sum(a0[r]), s1 = wave_sum(a1[r]);
            if (lane == r) {
                x0 = rnd<T>(s0);  // the projection output in the activation dtype, as the separate GEMM leaves it
                x1 = rnd<T>(s1);
            }
        }
        if (lane < mm) rope_store_pair<T>(P, m0 + lane, 0, 2 * pair, x0, x1);
    }
}

// destination and rotation of the column pair starting at `col` of a decode-step projection row (S = 1): rope_store_pair's cases with the
// position-only part (powf + sincosf) separated, so that a kernel can evaluate it while its loads are in flight
template <typename T>
struct RopeDest {
    float sn, cs;
    bool rot;
    T *dst;           // row b = 0
    int64_t dst_row;  // elements between consecutive rows b
    // with_rotation = false: destination only (sn / cs come from elsewhere, e.g. from a thread of the workgroup that evaluated them once)
    __device__ __forceinline__ void init(const RopeAppendParams &P, int col, bool with_rotation = true) {
        const int NQ = P.G * P.h * P.Dk, GK = P.G * P.Dk, GV = P.G * P.Dv;
        const float pos = (float)P.t0;
        sn = 0.f;
        cs = 1.f;
        if (col < NQ) {
            rot = true;
            if (with_rotation) rope_sincos<T>(col >> 1, NQ, pos, P.rope_base, P.inv_scale, sn, cs);
            dst = (T *)P.Q_out + col;
            dst_row = NQ;
            return;
        }
        int c = col - NQ;
        const int pairw = GK + GV;
        const int sp = min(c / pairw, 2);
        c -= sp * pairw;
        const bool isv = c >= GK;
        if (isv) c -= GK;
        const int D = isv ? P.Dv : P.Dk;
        const int g = c / D, dc = c - g * D;
        rot = !isv && sp < 2;
        if (rot && with_rotation) rope_sincos<T>(dc >> 1, P.Dk, pos, P.rope_base, P.inv_scale, sn, cs);
        dst = (T *)P.cache[2 * sp + (isv ? 1 : 0)] + ((int64_t)g * P.S_max + P.t0) * D + dc;
        dst_row = (int64_t)P.G * P.S_max * D;
    }
    __device__ __forceinline__ void store(int b, float x0, float x1) const {
        if (rot) rope_rotate<T>(x0, x1, sn, cs, x0, x1);
        dst[b * dst_row] = Elt<T>::from_f(x0);
        dst[b * dst_row + 1] = Elt<T>::from_f(x1);
    }
};

// latency form of qkv_rope_append_kernel for 1-2 rows, K <= 512 NC (see linear_small_fast_kernel): every load out first, the pair's
// rotation (powf + sincosf, ~1 us of arithmetic that depends on the position only) evaluated while they are in flight
template <typename T, int NC>
__global__ __launch_bounds__(256, 2) void qkv_rope_append_fast_kernel(RopeAppendParams P, const T *__restrict__ X, const T *__restrict__ W, int K,
                                                                   const T *__restrict__ norm_w, float norm_eps) {
    const int lane = lane_id();
    const int NQ = P.G * P.h * P.Dk, GK = P.G * P.Dk, GV = P.G * P.Dv;
    const int NT = NQ + 3 * GK + 3 * GV;
    const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (2 * pair >= NT) return;
    const T *w0 = W + (int64_t)(2 * pair) * K, *w1 = w0 + K;
    const int M = P.B;  // S == 1, M <= 2
    u32x4 u0r[NC], u1r[NC], gr[NC], xr[2][NC];
    bool has[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int k = lane * 8 + 512 * c;
        has[c] = k < K;
        const int kc = min(k, K - 8);
        u0r[c] = *(const u32x4 *)(w0 + kc);
        u1r[c] = *(const u32x4 *)(w1 + kc);
        if (norm_w) gr[c] = *(const u32x4 *)(norm_w + kc);
#pragma unroll
        for (int r = 0; r < 2; ++r)
            if (r < M) xr[r][c] = *(const u32x4 *)(X + (int64_t)r * K + kc);
    }
    RopeDest<T> rd;  // destination and rotation of this wave's column pair, evaluated under the loads
    rd.init(P, 2 * pair);
    float rms[2] = {1.f, 1.f};
    if (norm_w) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
            if (r < M) {
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    if (has[c]) {
                        float v[8];
                        raw8<T>(xr[r][c], v);
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc += rnd<T>(v[j] * v[j]);
                    }
                float q = rnd<T>(wave_sum(acc) / (float)K);
                q = rnd<T>(q + norm_eps);
                rms[r] = rnd<T>(1.0f / sqrtf(q));
            }
    }
    float a0[2] = {0.f, 0.f}, a1[2] = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NC; ++c)
        if (has[c]) {
            float u0[8], u1[8], gv[8], xv[8];
            raw8<T>(u0r[c], u0);
            raw8<T>(u1r[c], u1);
            if (norm_w) raw8<T>(gr[c], gv);
#pragma unroll
            for (int r = 0; r < 2; ++r)
                if (r < M) {
                    raw8<T>(xr[r][c], xv);
                    if (norm_w) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) xv[j] = rnd<T>(rnd<T>(xv[j] * rms[r]) * gv[j]);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        a0[r] = fmaf(u0[j], xv[j], a0[r]);
                        a1[r] = fmaf(u1[j], xv[j], a1[r]);
                    }
                }
        }
#pragma unroll
    for (int r = 0; r < 2; ++r)
        if (r < M) {
            const float x0 = rnd<T>(wave_sum(a0[r])), x1 = rnd<T>(wave_sum(a1[r]));  // the projection output in the activation dtype
            if (lane == 0) rd.store(r, x0, x1);
        }
}

// ---- MFMA form for 9 <= M rows (batched decode): workgroup = 16 output columns x 64 rows, the 4 waves split the K axis.
// C^T[n, m] = W[n,:] . X[m,:]: W rows are the MFMA A operand (each W element is fetched once, straight from global in
// fragment shape), the rows of X the B operand (L1/L2 resident), all loads of a wave issued before its first MFMA; the four
// K-partials meet in LDS.  Thread t of the epilogue owns row m = t % 64 and columns 4 (t / 64) .. +3 (two rotation pairs).
// MIX (decode output projection): the rows of X are the three-branch mix, formed on the fly from X = O_cmp, mx.Os, mx.Ow and the row
// gates mx.gates[m G + g][3] with the rounding of the mix kernel (the same operand values as decode_finish + this kernel: same bits)
struct LinearMixArgs {
    const void *Os, *Ow;
    const float *gates;
    int G;
};
// NWV = waves of a workgroup that split the K axis: 4, or 8 for long rows (K >= 2048: fc2 of the MLP) -- a wave's k-steps go out in
// rounds of 6, each round a memory round trip, so K = 3072 on 4 waves was four of them in a row (15.5 us at 32 rows, 8.5 for fc1)
template <typename T, bool ROPE, bool MIX = false, int NWV = 4>
__global__ __launch_bounds__(NWV * 64, 2) void linear_mfma_kernel(RopeAppendParams P, const T *__restrict__ X, const T *__restrict__ W,
                                                               T *__restrict__ out, int M, int N, int K, int epi, const T *__restrict__ res,
                                                               LinearMixArgs mx) {
    static_assert(NWV == 4 || (NWV == 8 && !ROPE && !MIX), "eight waves: the plain projection only");
    using MT_ = MfmaT<T>;
    using x8 = typename MT_::x8;
    __shared__ float part[NWV][16][65];
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6), rho = lane & 15, q = lane >> 4;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 64;
    const int ksteps = K / 32, per = (ksteps + NWV - 1) / NWV;
    const int s0 = wave * per, s1 = min(ksteps, s0 + per);
    // the residual of this thread's outputs, fetched now: in the epilogue it would be one more dependent memory round trip
    constexpr int CPT = 16 / NWV;  // columns per thread of the epilogue
    [[maybe_unused]] float resv[CPT];
    if constexpr (!ROPE) {
        if (epi == 2) {
            const int m = min(m0 + (int)(threadIdx.x & 63), M - 1), nq = (int)(threadIdx.x >> 6);
#pragma unroll
            for (int r = 0; r < CPT; ++r) resv[r] = Elt<T>::to_f(res[(int64_t)m * N + min(n0 + CPT * nq + r, N - 1)]);
        }
    }
    f32x4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const T *wrow = W + (int64_t)min(n0 + rho, N - 1) * K + 8 * q;  // the last column tile may be partial: clamp, store guarded
    // ROPE (P.S == 1): the rotations of the workgroup's 8 column pairs depend on the position only -- threads 0..7 evaluate one each (powf +
    // sincosf) while the loads are in flight and pass it through LDS; every thread then needs only the destinations of its two pairs
    [[maybe_unused]] RopeDest<T> rd[2];
    __shared__ float rsc[8][2];
    if constexpr (ROPE) {
        if (threadIdx.x < 8) {
            RopeDest<T> one;
            one.init(P, min(n0 + 2 * (int)threadIdx.x, N - 2));
            rsc[threadIdx.x][0] = one.sn;
            rsc[threadIdx.x][1] = one.cs;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) rd[p].init(P, min(n0 + 4 * (int)(threadIdx.x >> 6) + 2 * p, N - 2), false);
    }
    if constexpr (MIX) {
        const int kpg = K / mx.G;
        const int nmt = min(4, (M - m0 + 15) >> 4);  // 16-row tiles that hold a row (the plain form computes clamped duplicates instead)
        auto run = [&](auto RC) {
            constexpr int R = decltype(RC)::value;  // k-steps in flight: 3 R (nmt + 1) + ... loads per round
            for (int sb = s0; sb < s1; sb += R) {
                x8 wf[R], oc[R][4], os[R][4], ow[R][4];
                float pr[R][4][3];
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const int s = min(sb + i, s1 - 1);
                    wf[i] = *(const x8 *)(wrow + 32 * s);
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        if (mt < nmt) {
                            const int m = min(m0 + 16 * mt + rho, M - 1);
                            const int64_t o = (int64_t)m * K + 32 * s + 8 * q;
                            oc[i][mt] = *(const x8 *)(X + o);
                            os[i][mt] = *(const x8 *)((const T *)mx.Os + o);
                            ow[i][mt] = *(const x8 *)((const T *)mx.Ow + o);
                            const float *gp = mx.gates + ((int64_t)m * mx.G + (32 * s + 8 * q) / kpg) * 3;
#pragma unroll
                            for (int k = 0; k < 3; ++k) pr[i][mt][k] = gp[k];
                        }
                }
#pragma unroll
                for (int i = 0; i < R; ++i)
                    if (sb + i < s1) {
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
                            if (mt < nmt) {
                                x8 xm;
#pragma unroll
                                for (int j = 0; j < 8; ++j)
                                    xm[j] = Elt<T>::from_f(mix3<T>(pr[i][mt], Elt<T>::to_f(oc[i][mt][j]), Elt<T>::to_f(os[i][mt][j]), Elt<T>::to_f(ow[i][mt][j])));
                                acc[mt] = MT_::mma(wf[i], xm, acc[mt]);
                            }
                    }
            }
        };
        if (nmt == 1) run(std::integral_constant<int, 6>{});
        else run(std::integral_constant<int, 2>{});
    } else {
        const int nmt = min(4, (M - m0 + 15) >> 4);  // 16-row tiles that hold a row: the others are neither fetched nor multiplied
        for (int sb = s0; sb < s1; sb += 6) {  // 6 k-steps (up to 30 loads) in flight per round
            x8 wf[6], xf[6][4];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int s = min(sb + i, s1 - 1);
                wf[i] = *(const x8 *)(wrow + 32 * s);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    if (mt < nmt) xf[i][mt] = *(const x8 *)(X + (int64_t)min(m0 + 16 * mt + rho, M - 1) * K + 32 * s + 8 * q);
            }
#pragma unroll
            for (int i = 0; i < 6; ++i)
                if (sb + i < s1) {
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        if (mt < nmt) acc[mt] = MT_::mma(wf[i], xf[i][mt], acc[mt]);
                }
        }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wave][4 * q + r][16 * mt + rho] = acc[mt][r];
    __syncthreads();
    const int m = m0 + (threadIdx.x & 63), nq = threadIdx.x >> 6;
    float v[CPT];
#pragma unroll
    for (int r = 0; r < CPT; ++r) {
        const int n = CPT * nq + r, mm = threadIdx.x & 63;
        v[r] = (part[0][n][mm] + part[1][n][mm]) + (part[2][n][mm] + part[3][n][mm]);
        if constexpr (NWV == 8) v[r] += (part[4][n][mm] + part[5][n][mm]) + (part[6][n][mm] + part[7][n][mm]);
    }
    if (m >= M) return;
    if (ROPE) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
            if (n0 + 4 * nq + 2 * p < N) {
                rd[p].sn = rsc[2 * nq + p][0];  // (written before the barrier above)
                rd[p].cs = rsc[2 * nq + p][1];
                rd[p].store(m, rnd<T>(v[2 * p]), rnd<T>(v[2 * p + 1]));
            }
    } else {
#pragma unroll
        for (int r = 0; r < CPT; ++r) {
            if (n0 + CPT * nq + r >= N) break;
            const int64_t idx = (int64_t)m * N + n0 + CPT * nq + r;
            float y = rnd<T>(v[r]);  // linear_epilogue's steps, the residual from the registers
            if (epi == 1) y = rnd<T>(y / (1.f + expf(-y)));
            else if (epi == 2) y = y + resv[r];
            out[idx] = Elt<T>::from_f(y);
        }
    }
}

static bool linear_mfma_ok(int dtype, int M, int N, int K, const void *X, const void *W) {
    return (dtype == NSA_DT_BF16 || dtype == NSA_DT_F16) && M >= 3 && K % 32 == 0 &&
           (((uintptr_t)X | (uintptr_t)W) % 16 == 0);
}

int launch_linear_small_epi(const void *A, const void *W, void *out, int M, int N, int K, int dtype, int epi, const void *res, hipStream_t st) {
    if (linear_mfma_ok(dtype, M, N, K, A, W)) {
        RopeAppendParams P{};
        const dim3 g2((unsigned)((N + 15) / 16), (unsigned)((M + 63) / 64));
        if (K >= 2048) {
            if (dtype == NSA_DT_BF16) hipLaunchKernelGGL((linear_mfma_kernel<__bf16, false, false, 8>), g2, dim3(512), 0, st, P, (const __bf16 *)A, (const __bf16 *)W, (__bf16 *)out, M, N, K, epi, (const __bf16 *)res, LinearMixArgs{});
            else hipLaunchKernelGGL((linear_mfma_kernel<_Float16, false, false, 8>), g2, dim3(512), 0, st, P, (const _Float16 *)A, (const _Float16 *)W, (_Float16 *)out, M, N, K, epi, (const _Float16 *)res, LinearMixArgs{});
        } else if (dtype == NSA_DT_BF16) hipLaunchKernelGGL((linear_mfma_kernel<__bf16, false>), g2, dim3(256), 0, st, P, (const __bf16 *)A, (const __bf16 *)W, (__bf16 *)out, M, N, K, epi, (const __bf16 *)res, LinearMixArgs{});
        else hipLaunchKernelGGL((linear_mfma_kernel<_Float16, false>), g2, dim3(256), 0, st, P, (const _Float16 *)A, (const _Float16 *)W, (_Float16 *)out, M, N, K, epi, (const _Float16 *)res, LinearMixArgs{});
        NSA_LAUNCH_CHECK("linear_small(mfma)");
        return NSA_OK;
    }
    return launch_linear_small_valu(A, W, out, M, N, K, dtype, epi, res, nullptr, 0.f, st);
}
// RMSNorm(x) folded into the projection (rows <= 8: the VALU kernel); returns NSA_ERR_INVALID without side effects when not applicable
bool linear_small_can_fold_norm(int dtype, int M, int N, int K, const void *A, const void *W) { return !linear_mfma_ok(dtype, M, N, K, A, W); }
bool linear_small_mix_supported(int dtype, int M, int N, int K, int G, const void *Oc, const void *Os, const void *Ow, const void *W) {
    return (dtype == NSA_DT_BF16 || dtype == NSA_DT_F16) && M >= 1 && G >= 1 && K % (32 * G) == 0 && N >= 1 && (M >= 3 || K <= 2048) &&
           (((uintptr_t)Oc | (uintptr_t)Os | (uintptr_t)Ow | (uintptr_t)W) % 16 == 0);
}

// rows 1-2: the VALU dot products of linear_small_kernel; from 3 on: the MFMA kernel (what launch_linear_small_epi picks for a ready-made A)
int launch_linear_small_mix(const void *Oc, const void *Os, const void *Ow, const float *gates, const void *W, void *out, int M, int N, int K, int G,
                            int dtype, int epi, const void *res, hipStream_t st) {
    NSA_CHECK_ARG(linear_small_mix_supported(dtype, M, N, K, G, Oc, Os, Ow, W) && gates && out, "linear_small_mix: unsupported shape");
    const bool bf = dtype == NSA_DT_BF16;
    if (linear_mfma_ok(dtype, M, N, K, Oc, W)) {
        const dim3 g2((unsigned)((N + 15) / 16), (unsigned)((M + 63) / 64));
        const RopeAppendParams P{};
        const LinearMixArgs mx{Os, Ow, gates, G};
        if (bf) hipLaunchKernelGGL((linear_mfma_kernel<__bf16, false, true>), g2, dim3(256), 0, st, P, (const __bf16 *)Oc, (const __bf16 *)W, (__bf16 *)out, M, N, K, epi, (const __bf16 *)res, mx);
        else hipLaunchKernelGGL((linear_mfma_kernel<_Float16, false, true>), g2, dim3(256), 0, st, P, (const _Float16 *)Oc, (const _Float16 *)W, (_Float16 *)out, M, N, K, epi, (const _Float16 *)res, mx);
    } else {
        NSA_CHECK_ARG(M <= 2 && K <= 2048, "linear_small_mix: the VALU form takes 1-2 rows of at most 2048 elements");
        const dim3 grid((unsigned)((N + 3) / 4));
#define NSA_LSM(T_, NC_) hipLaunchKernelGGL((linear_small_mix_kernel<T_, NC_>), grid, dim3(256), 0, st, (const T_ *)Oc, (const T_ *)Os, (const T_ *)Ow, gates, (const T_ *)W, (T_ *)out, M, N, K, G, epi, (const T_ *)res)
        if (K <= 1024) {
            if (bf) NSA_LSM(__bf16, 2);
            else NSA_LSM(_Float16, 2);
        } else {
            if (bf) NSA_LSM(__bf16, 4);
            else NSA_LSM(_Float16, 4);
        }
#undef NSA_LSM
    }
    NSA_LAUNCH_CHECK("linear_small_mix");
    return NSA_OK;
}

int launch_linear_small_norm(const void *A, const void *W, void *out, int M, int N, int K, int dtype, int epi, const void *res, const void *norm_w,
                             float eps, hipStream_t st) {
    return launch_linear_small_valu(A, W, out, M, N, K, dtype, epi, res, norm_w, eps, st);
}
int launch_linear_small(const void *A, const void *W, void *out, int M, int N, int K, int dtype, hipStream_t st) {
    return launch_linear_small_epi(A, W, out, M, N, K, dtype, 0, nullptr, st);
}

// norm_w != nullptr: RMSNorm(X) is folded into the projection (only taken by the VALU form, i.e. when qkv_can_fold_norm())
bool qkv_can_fold_norm(const RopeAppendParams &P, const void *X, const void *W, int K, int dtype) {
    const int NT = P.G * P.h * P.Dk + 3 * P.G * P.Dk + 3 * P.G * P.Dv;
    return !linear_mfma_ok(dtype, P.B, NT, K, X, W);
}
int launch_qkv_rope_append(const RopeAppendParams &P, const void *X, const void *W, int K, int dtype, hipStream_t st, const void *norm_w,
                           float norm_eps) {
    const int NT = P.G * P.h * P.Dk + 3 * P.G * P.Dk + 3 * P.G * P.Dv;
    if (linear_mfma_ok(dtype, P.B, NT, K, X, W)) {
        NSA_CHECK_ARG(norm_w == nullptr, "qkv_rope_append: the MFMA form takes normalised input");
        const dim3 g2((unsigned)((NT + 15) / 16), (unsigned)((P.B + 63) / 64));
        if (dtype == NSA_DT_BF16) hipLaunchKernelGGL((linear_mfma_kernel<__bf16, true>), g2, dim3(256), 0, st, P, (const __bf16 *)X, (const __bf16 *)W, (__bf16 *)nullptr, P.B, NT, K, 0, (const __bf16 *)nullptr, LinearMixArgs{});
        else hipLaunchKernelGGL((linear_mfma_kernel<_Float16, true>), g2, dim3(256), 0, st, P, (const _Float16 *)X, (const _Float16 *)W, (_Float16 *)nullptr, P.B, NT, K, 0, (const _Float16 *)nullptr, LinearMixArgs{});
        NSA_LAUNCH_CHECK("qkv_rope_append(mfma)");
        return NSA_OK;
    }
    const dim3 grid((unsigned)((NT / 2 + 3) / 4)), block(256);
    if (dtype != NSA_DT_F32 && P.B <= 2 && P.S == 1 && K >= 8 && K % 8 == 0 && K <= 2048 && (((uintptr_t)X | (uintptr_t)W) % 16 == 0) &&
        (!norm_w || (uintptr_t)norm_w % 16 == 0)) {
        const bool bf = dtype == NSA_DT_BF16;
        if (K <= 1024) {
            if (bf) hipLaunchKernelGGL((qkv_rope_append_fast_kernel<__bf16, 2>), grid, block, 0, st, P, (const __bf16 *)X, (const __bf16 *)W, K, (const __bf16 *)norm_w, norm_eps);
            else hipLaunchKernelGGL((qkv_rope_append_fast_kernel<_Float16, 2>), grid, block, 0, st, P, (const _Float16 *)X, (const _Float16 *)W, K, (const _Float16 *)norm_w, norm_eps);
        } else {
            if (bf) hipLaunchKernelGGL((qkv_rope_append_fast_kernel<__bf16, 4>), grid, block, 0, st, P, (const __bf16 *)X, (const __bf16 *)W, K, (const __bf16 *)norm_w, norm_eps);
            else hipLaunchKernelGGL((qkv_rope_append_fast_kernel<_Float16, 4>), grid, block, 0, st, P, (const _Float16 *)X, (const _Float16 *)W, K, (const _Float16 *)norm_w, norm_eps);
        }
        NSA_LAUNCH_CHECK("qkv_rope_append(fast)");
        return NSA_OK;
    }
    if (dtype == NSA_DT_F32) hipLaunchKernelGGL(qkv_rope_append_kernel<float>, grid, block, 0, st, P, (const float *)X, (const float *)W, K, (const float *)norm_w, norm_eps);
    else if (dtype == NSA_DT_BF16) hipLaunchKernelGGL(qkv_rope_append_kernel<__bf16>, grid, block, 0, st, P, (const __bf16 *)X, (const __bf16 *)W, K, (const __bf16 *)norm_w, norm_eps);
    else hipLaunchKernelGGL(qkv_rope_append_kernel<_Float16>, grid, block, 0, st, P, (const _Float16 *)X, (const _Float16 *)W, K, (const _Float16 *)norm_w, norm_eps);
    NSA_LAUNCH_CHECK("qkv_rope_append");
    return NSA_OK;
}

// ------------------------------------------------------------------------------------------ compressed-token emission
// one 64-thread block per (b, g, j): K_cmp[j] = mean_i rope(K_raw[j d + i], pos = j d + i), V_cmp[j] = mean_i V_raw[j d + i]
template <typename T>
__global__ __launch_bounds__(64) void cmp_pool_kernel(CmpPoolParams P) {
    const int nj = P.j1 - P.j0;
    const int j = P.j0 + (int)(blockIdx.x % nj);
    const int bg = (int)(blockIdx.x / nj);
    const T *Kr = (const T *)P.K_raw + (int64_t)bg * P.S_max * P.Dk;
    const T *Vr = (const T *)P.V_raw + (int64_t)bg * P.S_max * P.Dv;
    T *Kc = (T *)P.K_cmp + ((int64_t)bg * P.n_cmp_max + j) * P.Dk;
    T *Vc = (T *)P.V_cmp + ((int64_t)bg * P.n_cmp_max + j) * P.Dv;
    const int r0 = j * P.d;
    for (int p = threadIdx.x; p < P.Dk / 2; p += 64) {
        float a0 = 0.f, a1 = 0.f;
        const float inv_freq = powf(P.rope_base, (-2.0f * (float)p) / (float)P.Dk);  // (rope_sincos's value, once per pair instead of per token)
        if (P.l <= 32) {
            // the l loads of the pair first (the loop below waits for each token's load behind the previous token's sincosf: a block was
            // as long as l dependent memory round trips)
            float xa[32], xb[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const T *src = Kr + (int64_t)(r0 + min(i, P.l - 1)) * P.Dk + 2 * p;
                xa[i] = Elt<T>::to_f(src[0]);
                xb[i] = Elt<T>::to_f(src[1]);
            }
#pragma unroll
            for (int i = 0; i < 32; ++i)
                if (i < P.l) {
                    float x0, x1, sn, cs;
                    sincosf(((float)(r0 + i) * P.inv_scale) * inv_freq, &sn, &cs);
                    rope_rotate<T>(xa[i], xb[i], rnd<T>(sn), rnd<T>(cs), x0, x1);
                    a0 += x0;
                    a1 += x1;
                }
        } else
        for (int i = 0; i < P.l; ++i) {
            const T *src = Kr + (int64_t)(r0 + i) * P.Dk + 2 * p;
            float x0, x1, sn, cs;
            sincosf(((float)(r0 + i) * P.inv_scale) * inv_freq, &sn, &cs);
            rope_rotate<T>(Elt<T>::to_f(src[0]), Elt<T>::to_f(src[1]), rnd<T>(sn), rnd<T>(cs), x0, x1);
            a0 += x0;
            a1 += x1;
        }
        Kc[2 * p] = Elt<T>::from_f(a0 / (float)P.l);
        Kc[2 * p + 1] = Elt<T>::from_f(a1 / (float)P.l);
    }
    for (int c = threadIdx.x; c < P.Dv; c += 64) {
        float a = 0.f;
        for (int i = 0; i < P.l; ++i) a += Elt<T>::to_f(Vr[(int64_t)(r0 + i) * P.Dv + c]);
        Vc[c] = Elt<T>::from_f(a / (float)P.l);
    }
}

// the same with the l rotations of a pair spread over the threads of a 256-thread block (the 64-thread form walks them one after the other:
// l x (powf + sincosf) in a row, 24 us for the ONE token a decode step emits): rotated keys and raw values go to LDS, then one thread per
// column sums them in the original order i = 0 .. l-1 -- the same fp32 additions, the same bits.  (Sharing a rotation between the (b, g)
// pairs of a block -- it depends on position and pair only -- was tried: fewer, serial blocks, 30 -> 45 us at 4k x 8.)
template <typename T>
__global__ __launch_bounds__(256) void cmp_pool_wide_kernel(CmpPoolParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *rk = (float *)smem;    // [l][Dk] rotated raw keys
    float *rv = rk + P.l * P.Dk;  // [l][Dv] raw values
    const int nj = P.j1 - P.j0;
    const int j = P.j0 + (int)(blockIdx.x % nj);
    const int bg = (int)(blockIdx.x / nj);
    const T *Kr = (const T *)P.K_raw + (int64_t)bg * P.S_max * P.Dk;
    const T *Vr = (const T *)P.V_raw + (int64_t)bg * P.S_max * P.Dv;
    T *Kc = (T *)P.K_cmp + ((int64_t)bg * P.n_cmp_max + j) * P.Dk;
    T *Vc = (T *)P.V_cmp + ((int64_t)bg * P.n_cmp_max + j) * P.Dv;
    const int r0 = j * P.d, hp = P.Dk / 2;
    for (int it = threadIdx.x; it < P.l * hp; it += 256) {
        const int i = it / hp, pp = it - i * hp;
        const T *src = Kr + (int64_t)(r0 + i) * P.Dk + 2 * pp;
        float x0, x1;
        rope_pair<T>(Elt<T>::to_f(src[0]), Elt<T>::to_f(src[1]), pp, P.Dk, (float)(r0 + i), P.rope_base, P.inv_scale, x0, x1);
        rk[i * P.Dk + 2 * pp] = x0;
        rk[i * P.Dk + 2 * pp + 1] = x1;
    }
    for (int it = threadIdx.x; it < P.l * P.Dv; it += 256) {
        const int i = it / P.Dv, c = it - i * P.Dv;
        rv[it] = Elt<T>::to_f(Vr[(int64_t)(r0 + i) * P.Dv + c]);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < P.Dk + P.Dv; c += 256) {
        const bool isv = c >= P.Dk;
        const float *col = isv ? rv + (c - P.Dk) : rk + c;
        const int stride = isv ? P.Dv : P.Dk;
        float a = 0.f;
        for (int i = 0; i < P.l; ++i) a += col[i * stride];
        if (isv) Vc[c - P.Dk] = Elt<T>::from_f(a / (float)P.l);
        else Kc[c] = Elt<T>::from_f(a / (float)P.l);
    }
}

int launch_cmp_pool(const CmpPoolParams &P, int dtype, hipStream_t st) {
    const int64_t nblk = (int64_t)P.nbg * (P.j1 - P.j0);
    if (nblk <= 0) return NSA_OK;
    NSA_CHECK_ARG(nblk < ((int64_t)1 << 31), "cmp_pool: too many blocks");
    const size_t lds = sizeof(float) * (size_t)P.l * (size_t)(P.Dk + P.Dv);
    // (a prefill pools thousands of tokens: there the 64-thread form's many small blocks fill the chip better; the wide form is for the few
    // tokens of a decode step, where the length of one block's chain is the kernel's time)
    if (lds <= 48 * 1024 && P.Dk % 2 == 0 && nblk <= 1024) {
        if (dtype == NSA_DT_F32) hipLaunchKernelGGL(cmp_pool_wide_kernel<float>, dim3((unsigned)nblk), dim3(256), lds, st, P);
        else if (dtype == NSA_DT_BF16) hipLaunchKernelGGL(cmp_pool_wide_kernel<__bf16>, dim3((unsigned)nblk), dim3(256), lds, st, P);
        else hipLaunchKernelGGL(cmp_pool_wide_kernel<_Float16>, dim3((unsigned)nblk), dim3(256), lds, st, P);
        NSA_LAUNCH_CHECK("cmp_pool(wide)");
        return NSA_OK;
    }
    if (dtype == NSA_DT_F32) hipLaunchKernelGGL(cmp_pool_kernel<float>, dim3((unsigned)nblk), dim3(64), 0, st, P);
    else if (dtype == NSA_DT_BF16) hipLaunchKernelGGL(cmp_pool_kernel<__bf16>, dim3((unsigned)nblk), dim3(64), 0, st, P);
    else hipLaunchKernelGGL(cmp_pool_kernel<_Float16>, dim3((unsigned)nblk), dim3(64), 0, st, P);
    NSA_LAUNCH_CHECK("cmp_pool");
    return NSA_OK;
}

// backward of the pooling: raw row r receives 1/l of the gradient of every compressed token whose window contains it
// (K: rotated back by the angle of position r).  dK_cmp/dV_cmp [nbg, n_cmp, D] -> dK_raw/dV_raw [nbg, S, D] (S rows).
template <typename T>
__global__ __launch_bounds__(64) void cmp_pool_bwd_kernel(CmpPoolParams P, const T *__restrict__ dKc, const T *__restrict__ dVc,
                                                          T *__restrict__ dKr, T *__restrict__ dVr, int S, int n_cmp) {
    const int r = (int)(blockIdx.x % S);
    const int bg = (int)(blockIdx.x / S);
    // windows j with j d <= r < j d + l
    const int jhi = min(n_cmp - 1, r / P.d);
    const int jlo = max(0, (r - P.l + P.d) / P.d);  // ceil((r - l + 1) / d) for r - l + 1 > 0, else 0
    const float inv_l = 1.0f / (float)P.l;
    for (int p = threadIdx.x; p < P.Dk / 2; p += 64) {
        float g0 = 0.f, g1 = 0.f;
        for (int j = jlo; j <= jhi; ++j) {
            if (j * P.d + P.l <= r) continue;
            const T *src = dKc + ((int64_t)bg * n_cmp + j) * P.Dk + 2 * p;
            g0 += Elt<T>::to_f(src[0]);
            g1 += Elt<T>::to_f(src[1]);
        }
        g0 *= inv_l;
        g1 *= inv_l;
        const float inv_freq = powf(P.rope_base, (-2.0f * (float)p) / (float)P.Dk);
        float sn, cs;
        sincosf(((float)r * P.inv_scale) * inv_freq, &sn, &cs);
        sn = rnd<T>(sn);
        cs = rnd<T>(cs);
        T *dst = dKr + ((int64_t)bg * S + r) * P.Dk + 2 * p;
        dst[0] = Elt<T>::from_f(g0 * cs + g1 * sn);
        dst[1] = Elt<T>::from_f(g1 * cs - g0 * sn);
    }
    for (int c = threadIdx.x; c < P.Dv; c += 64) {
        float g = 0.f;
        for (int j = jlo; j <= jhi; ++j) {
            if (j * P.d + P.l <= r) continue;
            g += Elt<T>::to_f(dVc[((int64_t)bg * n_cmp + j) * P.Dv + c]);
        }
        dVr[((int64_t)bg * S + r) * P.Dv + c] = Elt<T>::from_f(g * inv_l);
    }
}

int launch_cmp_pool_bwd(const CmpPoolParams &P, const void *dKc, const void *dVc, void *dKr, void *dVr, int S, int n_cmp, int dtype,
                        hipStream_t st) {
    const int64_t nblk = (int64_t)P.nbg * S;
    if (nblk <= 0) return NSA_OK;
    NSA_CHECK_ARG(nblk < ((int64_t)1 << 31), "cmp_pool_bwd: too many blocks");
    if (dtype == NSA_DT_F32) hipLaunchKernelGGL(cmp_pool_bwd_kernel<float>, dim3((unsigned)nblk), dim3(64), 0, st, P, (const float *)dKc, (const float *)dVc, (float *)dKr, (float *)dVr, S, n_cmp);
    else if (dtype == NSA_DT_BF16) hipLaunchKernelGGL(cmp_pool_bwd_kernel<__bf16>, dim3((unsigned)nblk), dim3(64), 0, st, P, (const __bf16 *)dKc, (const __bf16 *)dVc, (__bf16 *)dKr, (__bf16 *)dVr, S, n_cmp);
    else hipLaunchKernelGGL(cmp_pool_bwd_kernel<_Float16>, dim3((unsigned)nblk), dim3(64), 0, st, P, (const _Float16 *)dKc, (const _Float16 *)dVc, (_Float16 *)dKr, (_Float16 *)dVr, S, n_cmp);
    NSA_LAUNCH_CHECK("cmp_pool_bwd");
    return NSA_OK;
}

// ------------------------------------------------------------------------------------------ RMSNorm of a few rows
// y = (x * rsqrt(mean(x^2) + eps)) * w, one wave per row, rounded where the eager chain rounds (llama_block_nsa.py:16-19)
template <typename T>
__device__ __forceinline__ void store8(T *p, const float (&v)[8]) {  // 8 consecutive elements, 16-byte aligned for 2-byte types
    if constexpr (sizeof(T) == 2) {
        u32x4 raw;
        T *e = (T *)&raw;
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = Elt<T>::from_f(v[j]);
        *(u32x4 *)p = raw;
    } else {
        *(f32x4 *)p = (f32x4){v[0], v[1], v[2], v[3]};
        *(f32x4 *)(p + 4) = (f32x4){v[4], v[5], v[6], v[7]};
    }
}

template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_rows_kernel(const T *__restrict__ x, const T *__restrict__ w, T *__restrict__ y, int M, int dim,
                                                           float eps, int vec) {
    const int lane = lane_id();
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const T *xr = x + (int64_t)m * dim;
    T *yr = y + (int64_t)m * dim;
    if (vec) {  // dim % 8 == 0 and 16-byte aligned rows: whole rows per load instruction
        const float r = row_rms<T>(xr, dim, true, eps);
        for (int k = lane * 8; k < dim; k += 512) {
            float xv[8], wv[8], o[8];
            load8<T>(xr + k, 8, true, xv);
            load8<T>(w + k, 8, true, wv);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = rnd<T>(xv[j] * r) * wv[j];
            store8<T>(yr + k, o);
        }
        return;
    }
    float acc = 0.f;
    for (int i = lane; i < dim; i += 64) {
        const float v = Elt<T>::to_f(xr[i]);
        acc += rnd<T>(v * v);
    }
    float r = rnd<T>(wave_sum(acc) / (float)dim);
    r = rnd<T>(r + eps);
    r = rnd<T>(1.0f / sqrtf(r));
    for (int i = lane; i < dim; i += 64) yr[i] = Elt<T>::from_f(rnd<T>(Elt<T>::to_f(xr[i]) * r) * Elt<T>::to_f(w[i]));
}

int launch_rmsnorm_rows(const void *x, const void *w, void *y, int M, int dim, float eps, int dtype, hipStream_t st) {
    if (M == 0) return NSA_OK;
    const dim3 grid((unsigned)((M + 3) / 4)), block(256);
    const int vec = dim % 8 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)w % 16 == 0 && (uintptr_t)y % 16 == 0;
    if (dtype == NSA_DT_F32) hipLaunchKernelGGL(rmsnorm_rows_kernel<float>, grid, block, 0, st, (const float *)x, (const float *)w, (float *)y, M, dim, eps, vec);
    else if (dtype == NSA_DT_BF16) hipLaunchKernelGGL(rmsnorm_rows_kernel<__bf16>, grid, block, 0, st, (const __bf16 *)x, (const __bf16 *)w, (__bf16 *)y, M, dim, eps, vec);
    else hipLaunchKernelGGL(rmsnorm_rows_kernel<_Float16>, grid, block, 0, st, (const _Float16 *)x, (const _Float16 *)w, (_Float16 *)y, M, dim, eps, vec);
    NSA_LAUNCH_CHECK("rmsnorm_rows");
    return NSA_OK;
}

// backward of y = (x r) w, r = rsqrt(mean(x^2) + eps):  dx = r (g - xh mean(g xh)),  g = dy w,  xh = x r;  dw = sum_rows dy xh.
// A workgroup owns RMS_BWD_ROWS consecutive rows (one wave per row at a time); the per-column dw sums stay in registers across the
// rows of a wave, are added over the 4 waves through LDS and leave as one fp32 partial row per workgroup; a second kernel adds the
// partial rows in fixed order (no atomics: reproducible).  dim % 8 == 0, dim <= 4096.
constexpr int RMS_BWD_ROWS = 64;
constexpr int RMS_BWD_CH = 8;  // 8-element chunks per lane: dim <= 512 * 8
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_rows_bwd_kernel(const T *__restrict__ x, const T *__restrict__ w, const T *__restrict__ dy,
                                                               T *__restrict__ dx, float *__restrict__ part, int M, int dim, float eps) {
    __shared__ float red[4][64 * 8];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * RMS_BWD_ROWS;
    float dwacc[RMS_BWD_CH][8];
#pragma unroll
    for (int c = 0; c < RMS_BWD_CH; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) dwacc[c][j] = 0.f;
    for (int m = r0 + wave; m < min(M, r0 + RMS_BWD_ROWS); m += 4) {
        const T *xr = x + (int64_t)m * dim, *gr = dy + (int64_t)m * dim;
        const float r = row_rms<T>(xr, dim, true, eps);
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < RMS_BWD_CH; ++c) {
            const int k = lane * 8 + 512 * c;
            if (k < dim) {
                float xv[8], gv[8], wv[8];
                load8<T>(xr + k, 8, true, xv);
                load8<T>(gr + k, 8, true, gv);
                load8<T>(w + k, 8, true, wv);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xh = xv[j] * r;
                    dot = fmaf(gv[j] * wv[j], xh, dot);
                    dwacc[c][j] = fmaf(gv[j], xh, dwacc[c][j]);
                }
            }
        }
        const float mean = wave_sum(dot) / (float)dim;
        T *dr = dx + (int64_t)m * dim;
#pragma unroll
        for (int c = 0; c < RMS_BWD_CH; ++c) {
            const int k = lane * 8 + 512 * c;
            if (k < dim) {
                float xv[8], gv[8], wv[8], o[8];
                load8<T>(xr + k, 8, true, xv);
                load8<T>(gr + k, 8, true, gv);
                load8<T>(w + k, 8, true, wv);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = r * (gv[j] * wv[j] - xv[j] * r * mean);
                store8<T>(dr + k, o);
            }
        }
    }
    // dw partial of the workgroup: waves added in fixed order
#pragma unroll
    for (int c = 0; c < RMS_BWD_CH; ++c) {
        if (512 * c >= dim) break;  // uniform
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) red[wave][lane * 8 + j] = dwacc[c][j];
        __syncthreads();
        for (int i = threadIdx.x; i < 512; i += 256) {
            const int k = 512 * c + i;
            if (k < dim) part[(int64_t)blockIdx.x * dim + k] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
        }
    }
}

// dw[k] = sum over the workgroup partials, fixed order: a workgroup owns 64 columns, its 4 waves take every 4th partial row with 8
// independent accumulators (the loads of 8 rows are in flight together; one dependent chain per column made this kernel 235 us)
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_dw_reduce_kernel(const float *__restrict__ part, T *__restrict__ dw, int nparts, int dim) {
    __shared__ float red[4][64];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + lane;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (k < dim) {
        int p = wave;
        for (; p + 28 < nparts; p += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += part[(int64_t)(p + 4 * u) * dim + k];
        }
        for (; p < nparts; p += 4) acc[0] += part[(int64_t)p * dim + k];
    }
    red[wave][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (wave == 0 && k < dim) dw[k] = Elt<T>::from_f(((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane]);
}

size_t rmsnorm_rows_bwd_workspace(int M, int dim) { return (size_t)((M + RMS_BWD_ROWS - 1) / RMS_BWD_ROWS) * dim * sizeof(float); }

template <typename T>
static int launch_rmsnorm_bwd_t(const void *x, const void *w, const void *dy, void *dx, void *dw, int M, int dim, float eps, float *part,
                                hipStream_t st) {
    const int nparts = (M + RMS_BWD_ROWS - 1) / RMS_BWD_ROWS;
    hipLaunchKernelGGL(rmsnorm_rows_bwd_kernel<T>, dim3((unsigned)nparts), dim3(256), 0, st, (const T *)x, (const T *)w, (const T *)dy, (T *)dx,
                       part, M, dim, eps);
    NSA_LAUNCH_CHECK("rmsnorm_rows_bwd");
    hipLaunchKernelGGL(rmsnorm_dw_reduce_kernel<T>, dim3((unsigned)((dim + 63) / 64)), dim3(256), 0, st, (const float *)part, (T *)dw, nparts,
                       dim);
    NSA_LAUNCH_CHECK("rmsnorm_dw_reduce");
    return NSA_OK;
}

int launch_rmsnorm_rows_bwd(const void *x, const void *w, const void *dy, void *dx, void *dw, int M, int dim, float eps, int dtype,
                            void *workspace, size_t workspace_bytes, hipStream_t st) {
    NSA_CHECK_ARG(dim % 8 == 0 && dim <= 512 * RMS_BWD_CH, "rmsnorm_rows_bwd: dim must be a multiple of 8 and <= %d (got %d)", 512 * RMS_BWD_CH, dim);
    NSA_CHECK_ARG((uintptr_t)x % 16 == 0 && (uintptr_t)w % 16 == 0 && (uintptr_t)dy % 16 == 0 && (uintptr_t)dx % 16 == 0,
                  "rmsnorm_rows_bwd: 16-byte aligned tensors required");
    NSA_CHECK_ARG(workspace && workspace_bytes >= rmsnorm_rows_bwd_workspace(M, dim) && (uintptr_t)workspace % 16 == 0,
                  "rmsnorm_rows_bwd: workspace too small");
    if (dtype == NSA_DT_F32) return launch_rmsnorm_bwd_t<float>(x, w, dy, dx, dw, M, dim, eps, (float *)workspace, st);
    if (dtype == NSA_DT_BF16) return launch_rmsnorm_bwd_t<__bf16>(x, w, dy, dx, dw, M, dim, eps, (float *)workspace, st);
    return launch_rmsnorm_bwd_t<_Float16>(x, w, dy, dx, dw, M, dim, eps, (float *)workspace, st);
}

// ------------------------------------------------------------------------------------------ gate MLP + combine
// one wave per (b, s, g) row: q_pooled = mean_h Q -> fc1 -> silu -> fc2 -> / tau -> softmax (one-hot when the top two
// logits are more than 50 apart, nsa_attention.py:70-81) -> O = g_cmp O_cmp + g_sel O_sel + g_win O_win
template <typename T>
__global__ __launch_bounds__(256) void gate_combine_kernel(GateCombineParams P) {
    __shared__ float sqp[4][256];
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= P.R) return;
    float pr[3];
    if constexpr (sizeof(T) == 2) {
        // the m7c geometry with 16-byte aligned branch outputs: every load of the row -- Q, the gate weights AND the three branch outputs (one
        // 16-byte piece per lane instead of six 2-byte loads) -- is out before the gate arithmetic starts
        const int ne = P.h * P.Dv;
        if (P.Dk == 64 && P.Hd <= 32 && P.h <= 8 && ne % 8 == 0 && ne <= 512 &&
            (((uintptr_t)P.O_cmp | (uintptr_t)P.O_sel | (uintptr_t)P.O_win | (uintptr_t)P.O_out) % 16 == 0)) {
            GateFast<T> gf;
            gf.load((const T *)P.Q + row * P.h * 64, P.h, P.Hd, P.w1, P.b1, P.w2, P.b2);
            const int64_t base = row * ne;
            const bool mine = lane * 8 < ne;
            const int off = mine ? lane * 8 : 0;
            const u32x4 rc = *(const u32x4 *)((const T *)P.O_cmp + base + off), rs = *(const u32x4 *)((const T *)P.O_sel + base + off),
                        rw = *(const u32x4 *)((const T *)P.O_win + base + off);
            gf.compute(P.h, P.Hd, P.tau, sqp[wave], pr);
            if (P.gates_out && lane < 3) P.gates_out[row * 3 + lane] = lane == 0 ? pr[0] : (lane == 1 ? pr[1] : pr[2]);
            if (mine) {
                float oc[8], os[8], ow[8];
                raw8<T>(rc, oc);
                raw8<T>(rs, os);
                raw8<T>(rw, ow);
                u32x4 outv;
                T *ov = (T *)&outv;
#pragma unroll
                for (int j = 0; j < 8; ++j) ov[j] = Elt<T>::from_f(mix3<T>(pr, oc[j], os[j], ow[j]));
                *(u32x4 *)((T *)P.O_out + base + off) = outv;
            }
            return;
        }
    }
    if (P.Dk == 64 && P.Hd <= 32 && P.h <= 8) {
        GateFast<T> gf;
        gf.load((const T *)P.Q + row * P.h * 64, P.h, P.Hd, P.w1, P.b1, P.w2, P.b2);
        gf.compute(P.h, P.Hd, P.tau, sqp[wave], pr);
    } else {
        gate_probs<T>((const T *)P.Q + row * P.h * P.Dk, P.h, P.Dk, P.Hd, P.w1, P.b1, P.w2, P.b2, P.tau, sqp[wave], pr);
    }
    if (P.gates_out && lane < 3) P.gates_out[row * 3 + lane] = lane == 0 ? pr[0] : (lane == 1 ? pr[1] : pr[2]);
    const int64_t base = row * P.h * P.Dv;
    const T *Oc = (const T *)P.O_cmp + base, *Os = (const T *)P.O_sel + base, *Ow = (const T *)P.O_win + base;
    T *Oo = (T *)P.O_out + base;
    for (int e = lane; e < P.h * P.Dv; e += 64)
        Oo[e] = Elt<T>::from_f(mix3<T>(pr, Elt<T>::to_f(Oc[e]), Elt<T>::to_f(Os[e]), Elt<T>::to_f(Ow[e])));
}

// backward of the combine: dO_i = gate_i dO (activation dtype) and dgate_i = sum_{h,d} O_i dO (fp32) per row; the gradient of
// the tiny gate MLP itself is taken by the host side from dgate.
template <typename T>
__global__ __launch_bounds__(256) void gate_combine_bwd_kernel(GateCombineParams P, const T *__restrict__ dO, const float *__restrict__ gates,
                                                               T *__restrict__ dOc, T *__restrict__ dOs, T *__restrict__ dOw,
                                                               float *__restrict__ dgates) {
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= P.R) return;
    const float g0 = gates[row * 3], g1 = gates[row * 3 + 1], g2 = gates[row * 3 + 2];
    const int64_t base = row * P.h * P.Dv;
    const T *Oc = (const T *)P.O_cmp + base, *Os = (const T *)P.O_sel + base, *Ow = (const T *)P.O_win + base;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int e = lane; e < P.h * P.Dv; e += 64) {
        const float d = Elt<T>::to_f(dO[base + e]);
        a0 = fmaf(Elt<T>::to_f(Oc[e]), d, a0);
        a1 = fmaf(Elt<T>::to_f(Os[e]), d, a1);
        a2 = fmaf(Elt<T>::to_f(Ow[e]), d, a2);
        dOc[base + e] = Elt<T>::from_f(g0 * d);
        dOs[base + e] = Elt<T>::from_f(g1 * d);
        dOw[base + e] = Elt<T>::from_f(g2 * d);
    }
    a0 = wave_sum(a0);
    a1 = wave_sum(a1);
    a2 = wave_sum(a2);
    if (lane == 0) {
        dgates[row * 3] = a0;
        dgates[row * 3 + 1] = a1;
        dgates[row * 3 + 2] = a2;
    }
}

int launch_gate_combine_bwd(const GateCombineParams &P, const void *dO, const float *gates, void *dOc, void *dOs, void *dOw, float *dgates,
                            int dtype, hipStream_t st) {
    if (P.R == 0) return NSA_OK;
    const dim3 grid((unsigned)((P.R + 3) / 4)), block(256);
    if (dtype == NSA_DT_F32) hipLaunchKernelGGL(gate_combine_bwd_kernel<float>, grid, block, 0, st, P, (const float *)dO, gates, (float *)dOc, (float *)dOs, (float *)dOw, dgates);
    else if (dtype == NSA_DT_BF16) hipLaunchKernelGGL(gate_combine_bwd_kernel<__bf16>, grid, block, 0, st, P, (const __bf16 *)dO, gates, (__bf16 *)dOc, (__bf16 *)dOs, (__bf16 *)dOw, dgates);
    else hipLaunchKernelGGL(gate_combine_bwd_kernel<_Float16>, grid, block, 0, st, P, (const _Float16 *)dO, gates, (_Float16 *)dOc, (_Float16 *)dOs, (_Float16 *)dOw, dgates);
    NSA_LAUNCH_CHECK("gate_combine_bwd");
    return NSA_OK;
}

// decode: split-KV combine of the three branches + gate + mix in one pass.  One wave per (row, head), Dv = 64 (lane = column).
// Each branch arrives either as split-KV partial records (ns > 1, layout of sel_attn_combine_kernel) or final (ns == 1).
template <typename T>
__global__ __launch_bounds__(256) void decode_finish_kernel(DecodeFinishParams P) {
    __shared__ float sqp[4][256];
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
    if (wid >= P.R * P.h) return;
    const int64_t row = wid / P.h;
    const int hh = (int)(wid - row * P.h);
    constexpr int D = 64;
    const T *Qr = (const T *)P.Q + row * P.h * P.Dk;
    const bool fast = P.Dk == 64 && P.Hd <= 32 && P.h <= 8;
    // ---- every load first (split indices are clamped instead of predicated: a clamped duplicate gets weight 0 below)
    GateFast<T> gf;
    if (fast) gf.load(Qr, P.h, P.Hd, P.w1, P.b1, P.w2, P.b2);
    float mv[3], lv[3], pv[3][16], fin[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int ns = P.ns[i];
        const bool split = ns > 1;
        const float *base = split ? P.part[i] + ((row * ns) * (int64_t)P.h + hh) * (D + PART_PAD) : (const float *)nullptr;
        const int64_t sstride = (int64_t)P.h * (D + PART_PAD);
        fin[i] = split ? 0.f : Elt<T>::to_f(((const T *)P.O[i])[wid * D + lane]);
        mv[i] = (split && lane < ns) ? base[lane * sstride] : -INFINITY;
        lv[i] = (split && lane < ns) ? base[lane * sstride + 1] : 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) pv[i][s] = split ? base[min(s, ns - 1) * sstride + PART_PAD + lane] : 0.f;
    }
    // ---- arithmetic
    float pr[3];
    if (fast) gf.compute(P.h, P.Hd, P.tau, sqp[wave], pr);
    else gate_probs<T>(Qr, P.h, P.Dk, P.Hd, P.w1, P.b1, P.w2, P.b2, P.tau, sqp[wave], pr);
    float o[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float mmax = wave_max(mv[i]);
        const float w = (mv[i] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mv[i] - mmax);  // lanes >= ns: 0
        const float ltot = wave_sum(lv[i] * w);
        float acc = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = fmaf(pv[i][s], __shfl(w, s, 64), acc);
        // the branch output in the activation dtype, as its own combine pass leaves it
        o[i] = P.ns[i] > 1 ? rnd<T>(acc * (ltot > 0.f ? 1.f / ltot : 0.f)) : fin[i];
    }
    if (P.gates_out && hh == 0 && lane < 3) P.gates_out[row * 3 + lane] = lane == 0 ? pr[0] : (lane == 1 ? pr[1] : pr[2]);
    ((T *)P.O_out)[wid * D + lane] = Elt<T>::from_f(mix3<T>(pr, o[0], o[1], o[2]));
}

int launch_decode_finish(const DecodeFinishParams &P, int dtype, hipStream_t st) {
    if (P.R == 0) return NSA_OK;
    NSA_CHECK_ARG(P.Dv == 64 && P.Dk <= 256 && P.Hd >= 1 && P.Hd <= 64, "decode_finish: Dv = 64, Dk <= 256, hidden <= 64 supported");
    for (int i = 0; i < 3; ++i) NSA_CHECK_ARG(P.ns[i] >= 1 && P.ns[i] <= 16 && (P.ns[i] == 1 ? P.O[i] != nullptr : P.part[i] != nullptr), "decode_finish: bad branch input");
    const dim3 grid((unsigned)((P.R * P.h + 3) / 4)), block(256);
    if (dtype == NSA_DT_BF16) hipLaunchKernelGGL(decode_finish_kernel<__bf16>, grid, block, 0, st, P);
    else if (dtype == NSA_DT_F16) hipLaunchKernelGGL(decode_finish_kernel<_Float16>, grid, block, 0, st, P);
    else hipLaunchKernelGGL(decode_finish_kernel<float>, grid, block, 0, st, P);
    NSA_LAUNCH_CHECK("decode_finish");
    return NSA_OK;
}

int launch_gate_combine(const GateCombineParams &P, int dtype, hipStream_t st) {
    if (P.R == 0) return NSA_OK;
    NSA_CHECK_ARG(P.Dk <= 256 && P.Hd >= 1 && P.Hd <= 64, "gate_combine: Dk <= 256 and 1 <= hidden <= 64 supported");
    const dim3 grid((unsigned)((P.R + 3) / 4)), block(256);
    if (dtype == NSA_DT_F32) hipLaunchKernelGGL(gate_combine_kernel<float>, grid, block, 0, st, P);
    else if (dtype == NSA_DT_BF16) hipLaunchKernelGGL(gate_combine_kernel<__bf16>, grid, block, 0, st, P);
    else hipLaunchKernelGGL(gate_combine_kernel<_Float16>, grid, block, 0, st, P);
    NSA_LAUNCH_CHECK("gate_combine");
    return NSA_OK;
}

// ------------------------------------------------------------------------------------------ model-level helpers (decode)
// x[b, :] = embed[tokens[b], :]
template <typename T>
__global__ __launch_bounds__(256) void embed_rows_kernel(const int32_t *__restrict__ tokens, const T *__restrict__ embed, T *__restrict__ x, int B,
                                                         int dim, int vocab) {
    const int b = blockIdx.x;
    const int tok = min(max(tokens[b], 0), vocab - 1);
    for (int i = threadIdx.x; i < dim; i += 256) x[(int64_t)b * dim + i] = embed[(int64_t)tok * dim + i];
}
int launch_embed_rows(const int32_t *tokens, const void *embed, void *x, int B, int dim, int vocab, int dtype, hipStream_t st) {
    if (B == 0) return NSA_OK;
    if (dtype == NSA_DT_F32) hipLaunchKernelGGL(embed_rows_kernel<float>, dim3(B), dim3(256), 0, st, tokens, (const float *)embed, (float *)x, B, dim, vocab);
    else if (dtype == NSA_DT_BF16) hipLaunchKernelGGL(embed_rows_kernel<__bf16>, dim3(B), dim3(256), 0, st, tokens, (const __bf16 *)embed, (__bf16 *)x, B, dim, vocab);
    else hipLaunchKernelGGL(embed_rows_kernel<_Float16>, dim3(B), dim3(256), 0, st, tokens, (const _Float16 *)embed, (_Float16 *)x, B, dim, vocab);
    NSA_LAUNCH_CHECK("embed_rows");
    return NSA_OK;
}

// next[b] = argmax_v logits[b, v] (first maximum wins).  Stage 1: grid (chunks, B), one workgroup per 4096-value chunk -> (max, index)
// partials; stage 2: one wave per row over the partials.
constexpr int ARGMAX_CHUNK = 4096;
template <typename T>
__global__ __launch_bounds__(256) void argmax_part_kernel(const T *__restrict__ logits, float *__restrict__ pv, int32_t *__restrict__ pi, int vocab) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int b = blockIdx.y, lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    const T *row = logits + (int64_t)b * vocab;
    const int v0 = blockIdx.x * ARGMAX_CHUNK, v1 = min(vocab, v0 + ARGMAX_CHUNK);
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int v = v0 + threadIdx.x; v < v1; v += 256) {
        const float x = Elt<T>::to_f(row[v]);
        if (x > best) {
            best = x;
            bi = v;
        }
    }
    const float wm = wave_max(best);
    int cand = best == wm ? bi : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
    if (lane == 0) {
        sv[wave] = wm;
        si[wave] = cand;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = sv[0];
        int i = si[0];
        for (int w = 1; w < 4; ++w)
            if (sv[w] > m || (sv[w] == m && si[w] < i)) {
                m = sv[w];
                i = si[w];
            }
        pv[(int64_t)b * gridDim.x + blockIdx.x] = m;
        pi[(int64_t)b * gridDim.x + blockIdx.x] = i;
    }
}
__global__ __launch_bounds__(64) void argmax_final_kernel(const float *__restrict__ pv, const int32_t *__restrict__ pi, int32_t *__restrict__ next,
                                                         int nchunk) {
    const int b = blockIdx.x, lane = lane_id();
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < nchunk; c += 64) {
        const float x = pv[(int64_t)b * nchunk + c];
        const int i = pi[(int64_t)b * nchunk + c];
        if (x > best || (x == best && i < bi)) {
            best = x;
            bi = i;
        }
    }
    const float wm = wave_max(best);
    int cand = best == wm ? bi : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
    if (lane == 0) next[b] = cand == 0x7fffffff ? 0 : cand;
}
size_t argmax_rows_workspace(int B, int vocab) { return (size_t)B * ((vocab + ARGMAX_CHUNK - 1) / ARGMAX_CHUNK) * 8; }
int launch_argmax_rows(const void *logits, int32_t *next, int B, int vocab, int dtype, void *ws, hipStream_t st) {
    if (B == 0) return NSA_OK;
    const int nchunk = (vocab + ARGMAX_CHUNK - 1) / ARGMAX_CHUNK;
    float *pv = (float *)ws;
    int32_t *pi = (int32_t *)(pv + (size_t)B * nchunk);
    const dim3 grid((unsigned)nchunk, (unsigned)B);
    if (dtype == NSA_DT_F32) hipLaunchKernelGGL(argmax_part_kernel<float>, grid, dim3(256), 0, st, (const float *)logits, pv, pi, vocab);
    else if (dtype == NSA_DT_BF16) hipLaunchKernelGGL(argmax_part_kernel<__bf16>, grid, dim3(256), 0, st, (const __bf16 *)logits, pv, pi, vocab);
    else hipLaunchKernelGGL(argmax_part_kernel<_Float16>, grid, dim3(256), 0, st, (const _Float16 *)logits, pv, pi, vocab);
    NSA_LAUNCH_CHECK("argmax_part");
    hipLaunchKernelGGL(argmax_final_kernel, dim3(B), dim3(64), 0, st, (const float *)pv, (const int32_t *)pi, next, nchunk);
    NSA_LAUNCH_CHECK("argmax_final");
    return NSA_OK;
}

}  // namespace nsa
