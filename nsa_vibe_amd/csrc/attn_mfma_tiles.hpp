// Pieces shared by the MFMA attention kernels (selection attention, band attention): MFMA / transposing-read wrappers
// per element type, the LDS tile geometry with its XOR swizzles, and the split-KV partial record + combine kernel.
#pragma once
#include "nsa_common.hpp"
#include "sel_attn_params.hpp"

namespace nsa {

template <typename T>
struct MfmaT;
template <>
struct MfmaT<__bf16> {
    using x8 = bf16x8;
    using x4 = bf16x4;
    __device__ static f32x4 mma(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    __device__ static x4 tr(const unsigned char *p) {
        return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) x4 *)p);
    }
};
template <>
struct MfmaT<_Float16> {
    using x8 = f16x8;
    using x4 = f16x4;
    __device__ static f32x4 mma(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    __device__ static x4 tr(const unsigned char *p) {
        typedef __attribute__((ext_vector_type(4))) short s16x4;
        const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p);
        return __builtin_bit_cast(x4, r);
    }
};

template <int D>
struct Geo {
    static constexpr int ROWB = D * 2;          // bytes per K/V row
    static constexpr int PIECES = D / 8;        // 16-B pieces per row
    static constexpr int RPI = 64 / PIECES;     // rows covered by one wave-wide 16-B load
    static constexpr int NLD = 32 / RPI;        // loads per operand per 32-key tile
    static constexpr int KSTEPS = D / 32;       // MFMA k-steps of the QK product
    static constexpr int MT = D / 16;           // 16-row dv tiles of the PV product
    static constexpr int TILE_BYTES = 32 * ROWB;
    static constexpr int SEG_BYTES = ((SEG_INTS * 4 + 15) / 16) * 16;
    static constexpr int WAVE_LDS = 2 * TILE_BYTES + SEG_BYTES;
    __device__ static int swz_k(int row) { return row & (PIECES - 1); }
    __device__ static int swz_v(int row) { return D == 64 ? ((row >> 1) & 3) : (row & 7); }
};

constexpr int PART_PAD = 4;  // split-KV partial record per head: m, l, 2 pad floats, then D accumulators (16-B aligned)
constexpr float RESCALE_THR = 8.0f;  // log2 units: raise the running max only when a tile exceeds it by more than this

// combine split-KV partials: one wave per (row, head); the nsplit (m,l) records are read by nsplit lanes at once
// and the accumulators by all lanes with every split's load in flight together (no dependent-load chain)
template <typename T, int D>
__global__ __launch_bounds__(256) void sel_attn_combine_kernel(SelAttnParams P) {
    const int lane = lane_id();
    const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int h = P.h, ns = P.nsplit;  // ns <= 16
    if (wid >= P.R * h) return;
    const int64_t row = wid / h;
    const int hh = (int)(wid % h);
    const float *base = P.part + ((row * ns) * (int64_t)h + hh) * (D + PART_PAD);
    const int64_t sstride = (int64_t)h * (D + PART_PAD);  // floats between consecutive splits of this (row, head)
    float m = -INFINITY, l = 0.f;
    if (lane < ns) {
        m = base[lane * sstride];
        l = base[lane * sstride + 1];
    }
    const float mmax = wave_max(m);
    const float w = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - mmax);
    const float ltot = wave_sum(l * w);
    float acc[D / 64];
#pragma unroll
    for (int c = 0; c < D / 64; ++c) acc[c] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        if (s < ns) {
            const float ws = __shfl(w, s, 64);
#pragma unroll
            for (int c = 0; c < D / 64; ++c) acc[c] = fmaf(base[s * sstride + PART_PAD + c * 64 + lane], ws, acc[c]);
        }
    }
    const float inv = (ltot > 0.f) ? 1.f / ltot : 0.f;
    T *Or = (T *)P.O + (row * (int64_t)h + hh) * D;
#pragma unroll
    for (int c = 0; c < D / 64; ++c) Or[c * 64 + lane] = Elt<T>::from_f(acc[c] * inv);
    if (P.lse && lane == 0) P.lse[row * h + hh] = (ltot > 0.f) ? (mmax + __builtin_amdgcn_logf(ltot)) * LN2 : -INFINITY;
}

}  // namespace nsa
