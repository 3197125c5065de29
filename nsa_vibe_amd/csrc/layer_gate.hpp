// Gate MLP + three-branch mix of NSAAttention (nsa_attention.py:32-82, 85-124) as wave-level device functions, shared by the layer kernels
// (layer_fused.hip) and the decode step launch that carries the band branches (sel_decode_fused.hip evaluates the gates of a row there).
// Arithmetic follows the PyTorch operator chain the reference runs, including where it rounds to the activation dtype (rnd()).
#pragma once
#include "attn_mfma_tiles.hpp"

namespace nsa {

template <typename T>
__device__ __forceinline__ float rnd(float x) {
    return Elt<T>::to_f(Elt<T>::from_f(x));
}

// 8 consecutive elements as floats (16-byte loads when `vec`, i.e. K % 8 == 0 and 16-byte aligned rows)
template <typename T>
__device__ __forceinline__ void load8(const T *p, int nvalid, bool vec, float (&out)[8]) {
    if (vec) {
        if constexpr (sizeof(T) == 2) {
            const u32x4 raw = *(const u32x4 *)p;
            const T *e = (const T *)&raw;
#pragma unroll
            for (int j = 0; j < 8; ++j) out[j] = Elt<T>::to_f(e[j]);
        } else {
            const f32x4 a = *(const f32x4 *)p, b = *(const f32x4 *)(p + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                out[j] = a[j];
                out[4 + j] = b[j];
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] = j < nvalid ? Elt<T>::to_f(p[j]) : 0.f;
    }
}

// gate probabilities of one row (wave-cooperative; sqp = 256 wave-private floats)
template <typename T>
__device__ __forceinline__ void gate_probs(const T *Qr, int h, int Dk, int Hd, const void *w1_, const void *b1_, const void *w2_,
                                           const void *b2_, float tau, float *sqp, float (&pr)[3]) {
    const int lane = lane_id();
    for (int dk = lane; dk < Dk; dk += 64) {
        float a = 0.f;
        for (int hh = 0; hh < h; ++hh) a += Elt<T>::to_f(Qr[hh * Dk + dk]);
        sqp[dk] = rnd<T>(a / (float)h);
    }
    wave_lds_fence();
    // fc1 + silu + fc2: hidden unit j = j0 + lane % 32, the two half-waves split the Dk axis of its dot product (16-byte
    // weight loads); the fc2 contributions of the units are summed over the lanes at the end
    float g3[3] = {0.f, 0.f, 0.f};
    {
        const int half = lane >> 5, dspan = (Dk + 1) >> 1, d0 = half * dspan, d1 = min(Dk, d0 + dspan);
        const bool vec = (dspan % 8 == 0) && ((uintptr_t)w1_ % 16 == 0);
        for (int j0 = 0; j0 < Hd; j0 += 32) {
            const int j = j0 + (lane & 31);
            float a = 0.f;
            if (j < Hd) {
                const T *w1 = (const T *)w1_ + (int64_t)j * Dk;
                for (int dk = d0; dk < d1; dk += 8) {
                    float wv[8];
                    load8<T>(w1 + dk, d1 - dk, vec, wv);
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (dk + e < d1) a = fmaf(wv[e], sqp[dk + e], a);
                }
            }
            const float other = __shfl_xor(a, 32, 64);
            a = half == 0 ? a + other : other + a;  // low half + high half on both lanes
            if (j < Hd && half == 0) {
                a = rnd<T>(a + Elt<T>::to_f(((const T *)b1_)[j]));
                const float act = rnd<T>(a / (1.f + expf(-a)));  // silu
#pragma unroll
                for (int k = 0; k < 3; ++k) g3[k] = fmaf(Elt<T>::to_f(((const T *)w2_)[k * Hd + j]), act, g3[k]);
            }
        }
    }
    float gl[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        gl[k] = rnd<T>(wave_sum(g3[k]) + Elt<T>::to_f(((const T *)b2_)[k]));
        gl[k] = rnd<T>(gl[k] / fmaxf(tau, 1e-6f));
    }
    const float mx = fmaxf(gl[0], fmaxf(gl[1], gl[2]));
    float den = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        pr[k] = expf(gl[k] - mx);
        den += pr[k];
    }
    int arg = 0;
    if (gl[1] > gl[arg]) arg = 1;
    if (gl[2] > gl[arg]) arg = 2;
    float second = -INFINITY;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (k != arg) second = fmaxf(second, gl[k]);
    const bool peaked = (gl[arg] - second) > 50.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) pr[k] = peaked ? (k == arg ? 1.f : 0.f) : rnd<T>(pr[k] / den);
}

// The m7c geometry (Dk = 64, hidden <= 32, h <= 8) with every global load issued up front and no data-dependent branch
// before the arithmetic: in decode this kernel is a handful of waves and its time is the length of its load -> use chains.
// Same arithmetic, same rounding points as gate_probs.
template <typename T>
struct GateFast {
    float qv[8], w1v[4][8], b1v, w2v[3], b2v[3];
    __device__ __forceinline__ void load(const T *Qr, int h, int Hd, const void *w1_, const void *b1_, const void *w2_, const void *b2_) {
        const int lane = lane_id(), j = min(lane & 31, Hd - 1), half = lane >> 5;
#pragma unroll
        for (int hh = 0; hh < 8; ++hh) qv[hh] = Elt<T>::to_f(Qr[min(hh, h - 1) * 64 + lane]);
        const T *w1 = (const T *)w1_ + (int64_t)j * 64 + 32 * half;
        const bool vec = ((uintptr_t)w1_ % 16) == 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) load8<T>(w1 + 8 * c, 8, vec, w1v[c]);
        b1v = Elt<T>::to_f(((const T *)b1_)[j]);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            w2v[k] = Elt<T>::to_f(((const T *)w2_)[k * Hd + j]);
            b2v[k] = Elt<T>::to_f(((const T *)b2_)[k]);
        }
    }
    __device__ __forceinline__ void compute(int h, int Hd, float tau, float *sqp, float (&pr)[3]) const {
        const int lane = lane_id(), half = lane >> 5;
        float a = 0.f;
#pragma unroll
        for (int hh = 0; hh < 8; ++hh) a += hh < h ? qv[hh] : 0.f;
        sqp[lane] = rnd<T>(a / (float)h);
        wave_lds_fence();
        a = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) a = fmaf(w1v[c][e], sqp[32 * half + 8 * c + e], a);
        const float other = __shfl_xor(a, 32, 64);
        a = half == 0 ? a + other : other + a;
        a = rnd<T>(a + b1v);
        const float act = rnd<T>(a / (1.f + expf(-a)));
        const bool mine = half == 0 && (lane & 31) < Hd;
        float gl[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            gl[k] = rnd<T>(wave_sum(mine ? w2v[k] * act : 0.f) + b2v[k]);
            gl[k] = rnd<T>(gl[k] / fmaxf(tau, 1e-6f));
        }
        const float mx = fmaxf(gl[0], fmaxf(gl[1], gl[2]));
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            pr[k] = expf(gl[k] - mx);
            den += pr[k];
        }
        int arg = 0;
        if (gl[1] > gl[arg]) arg = 1;
        if (gl[2] > gl[arg]) arg = 2;
        float second = -INFINITY;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (k != arg) second = fmaxf(second, gl[k]);
        const bool peaked = (gl[arg] - second) > 50.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) pr[k] = peaked ? (k == arg ? 1.f : 0.f) : rnd<T>(pr[k] / den);
    }
};

template <typename T>
__device__ __forceinline__ float mix3(const float (&pr)[3], float oc, float os, float ow) {
    const float t1 = rnd<T>(pr[0] * oc), t2 = rnd<T>(pr[1] * os);
    const float t3 = rnd<T>(t1 + t2), t4 = rnd<T>(pr[2] * ow);
    return t3 + t4;
}

}  // namespace nsa
