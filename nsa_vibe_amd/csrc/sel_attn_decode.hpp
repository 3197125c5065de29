// Decode form of the selection attention: ONE 1024-thread workgroup per query row (b,g), S = 1.
//
// A decode step reads every selected K/V row exactly once (n*l' rows per (b,g)): it is bound by HBM latency, not by any pipe.
// The split-KV route (sel_attn_mfma.hip, SPLIT) spreads a row over 16 independent waves that each walk their 32-key tiles one
// DMA round trip after the other and leave partial records in global memory for a second launch to combine: 13 + 7 us per step
// at B = 64, 16k context (profiles/r02) for 33 MB of reads.  Here the 16 waves of one workgroup each take 64-key chunks of the
// row's selected union and issue ALL loads of a chunk at once -- K straight into registers in MFMA A-fragment shape (8
// global_load_dwordx4: the GEMV / decode form of the CDNA guide), V by 8 LDS-DMA pieces into the wave's 8 KiB of LDS for the
// transposing reads -- so a chunk costs one memory round trip; the per-wave (m, l, O^T) partials are merged through LDS behind
// one barrier and the row's O is written once.  No workspace, no second launch, and the same device function is the last phase
// of the fused decode scorer (sel_scores.hip: logits -> statistics -> Eq.9/10 -> top-n -> attention in one launch).
// Semantics: union of the clamped ranges (normalise_ranges_lanes), end <= start ignored, empty row -> zeros
// (attention_kernels.py:705-772).  bf16 / f16, Dk = Dv = 64, h <= 16, V rows contiguous (128 B apart).
#pragma once
#include "attn_mfma_tiles.hpp"

namespace nsa {
#ifdef NSA_DEC_TS
static __device__ long long g_dec_ts[64];
#define DEC_TS(i) do { if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) g_dec_ts[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DEC_TS(i)
#endif

struct DecAttnArgs {
    const void *Q;  // [R,h,64]
    const void *K;  // [B,G,S_kv,64] strided
    const void *V;
    void *O;        // [R,h,64]
    int G, h, S_kv, n;
    int64_t ksb, ksg, kss, vsb, vsg, vss;
    float c2;  // scale * log2(e)
};

constexpr int DEC_ATT_WAVES = 16;
constexpr int DEC_ATT_TILE = 64 * 128;                         // V chunk per wave
constexpr int DEC_ATT_TAIL = ((SEG_INTS * 4 + 15) / 16) * 16;  // sorted segments of the row
constexpr int DEC_ATT_LDS = DEC_ATT_WAVES * DEC_ATT_TILE + DEC_ATT_TAIL;

// lds: DEC_ATT_LDS bytes, 16-byte aligned; lanes i < n of wave 0 pass range i in (rs, re) (unclamped); every thread of the
// 1024-thread workgroup must call (two workgroup barriers inside).  SORTED: the ranges are ascending and disjoint with the live ones
// first (what select_topn_row_regs emits): the sort / union pass is skipped, only the clamp to [0, S_kv] remains.
// qf_in: the row's Q^T fragments if the caller holds them already (lane (rho, q): Q[head min(rho, h-1)][32 s + 8 q ..], s = 0, 1).
template <typename T, bool SORTED = false>
__device__ __forceinline__ void decode_attend_row(const DecAttnArgs &A, int64_t row, int rs, int re, unsigned char *lds,
                                                  const typename MfmaT<T>::x8 *qf_in = nullptr) {
    using M = MfmaT<T>;
    using x8 = typename M::x8;
    using x4 = typename M::x4;
    constexpr int ROWB = 128;
    const int lane = lane_id();
    const int wave = uniform((int)(threadIdx.x >> 6));
    const int rho = lane & 15, q = lane >> 4;
    const int h = A.h;
    const int g = (int)(row % A.G);
    const int64_t b = row / A.G;
    int *seg = (int *)(lds + DEC_ATT_WAVES * DEC_ATT_TILE);
    unsigned char *vl = lds + wave * DEC_ATT_TILE;

    // Q^T fragments (B operand): column = head (columns >= h repeat the last head: their results are never stored)
    x8 qf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
        qf[s] = qf_in ? qf_in[s] : __builtin_bit_cast(x8, *(const u32x4 *)((const T *)A.Q + (row * h + min(rho, h - 1)) * 64 + 32 * s + 8 * q));
    DEC_TS(10);
    int nseg, sstart = 0, slen = 0;  // chunk table in registers: lane i holds segment i (start, length)
    if constexpr (SORTED) {
        if (wave == 0 && lane < A.n) {
            const int s = min(max(rs, 0), A.S_kv), e = min(max(re, s), A.S_kv);
            seg[2 * lane] = s;
            seg[2 * lane + 1] = e - s;
        }
        __syncthreads();
        nseg = A.n;
        if (lane < nseg) {
            sstart = seg[2 * lane];
            slen = seg[2 * lane + 1];
        }
    } else {
        if (wave == 0) {
            int ns;
            const int total = normalise_ranges_lanes(rs, re, A.n, A.S_kv, seg, &ns);
            if (lane == 0) {
                seg[SEG_INTS - 2] = ns;
                seg[SEG_INTS - 1] = total;
            }
        }
        __syncthreads();
        nseg = uniform(seg[SEG_INTS - 2]);  // wave uniform by construction: tell the compiler (scalar loop control below)
        if (lane < nseg) {
            sstart = seg[2 * lane];
            slen = seg[2 * lane + 3] - seg[2 * lane + 1];
        }
    }
    const int nck = (slen + 63) >> 6;
    int inc = nck;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
    }
    const int cb = inc - nck;
    const int NC = uniform(__shfl(inc, 63, 64));
    DEC_TS(11);

    const unsigned char *Kb = (const unsigned char *)((const T *)A.K + b * A.ksb + (int64_t)g * A.ksg);
    const unsigned char *Vb = (const unsigned char *)((const T *)A.V + b * A.vsb + (int64_t)g * A.vsg);
    const int64_t krowb = A.kss * 2;
    const int ld_row = lane >> 3, ld_piece = lane & 7;
    [[maybe_unused]] const uint32_t vsw = (uint32_t)(((((ld_piece >> 1) ^ ((ld_row >> 1) & 3)) << 1) | (ld_piece & 1)) << 4);
    uint32_t vrd0[4];
    {
        const int qq = rho >> 2, pp = rho & 3, r = 4 * q + qq;
#pragma unroll
        for (int m = 0; m < 4; ++m) vrd0[m] = r * ROWB + ((m ^ ((r >> 1) & 3)) << 5) + 8 * pp;
    }
    // (readfirstlane returns a SIGNED int: the halves go through uint32_t, or a low half >= 2^31 sign-extends into the high one)
    const uint64_t va = (uint64_t)Vb;
    const uint32_t va_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)va), va_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(va >> 32));
    [[maybe_unused]] const auto vrs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(((uint64_t)va_hi << 32) | (uint64_t)va_lo), (short)0, __builtin_amdgcn_readfirstlane((int)((int64_t)(A.S_kv - 1) * ROWB + ROWB)), 0x00020000);

    f32x4 o[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) o[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrun = -INFINITY, lrun = 0.f;

    for (int c = wave; c < NC; c += DEC_ATT_WAVES) {
        // segment of chunk c: the lane whose [cb, cb + nck) holds c
        const unsigned long long hit = __ballot(lane < nseg && c >= cb && c < cb + nck);
        const int sl = __builtin_ctzll(hit);
        const int s0 = __builtin_amdgcn_readlane(sstart, sl), sn = __builtin_amdgcn_readlane(slen, sl), c0 = __builtin_amdgcn_readlane(cb, sl);
        const int tok0 = s0 + 64 * (c - c0);
        const int len = min(64, s0 + sn - tok0);  // keys of this chunk, >= 1
        // ---- every load of the chunk goes out at once: V by LDS-DMA, K straight to registers (rows past the chunk re-read its last row)
#if defined(__HIP_DEVICE_COMPILE__)
        {
            typedef __attribute__((address_space(3))) void lds_void;
            const int vs = uniform(tok0 * ROWB);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int rc = min(8 * i + ld_row, len - 1);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (lds_void *)(vl + i * 1024), 16, rc * ROWB + vsw, vs, 0, 0);
            }
        }
#endif
        x8 kfr[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kr = tok0 + min(16 * u + rho, len - 1);
#pragma unroll
            for (int s = 0; s < 2; ++s) kfr[u][s] = __builtin_bit_cast(x8, *(const u32x4 *)(Kb + (int64_t)kr * krowb + 64 * s + 16 * q));
        }
        f32x4 sacc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            sacc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) sacc[u] = M::mma(kfr[u][s], qf[s], sacc[u]);
        }
        DEC_TS(12);
        float x[16];
        float vmax = -INFINITY;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = (16 * u + 4 * q + j < len) ? sacc[u][j] * A.c2 : -INFINITY;
                x[4 * u + j] = v;
                vmax = fmaxf(vmax, v);
            }
        vmax = fmaxf(vmax, __shfl_xor(vmax, 16, 64));
        vmax = fmaxf(vmax, __shfl_xor(vmax, 32, 64));
        const float mnew = fmaxf(mrun, vmax);  // finite: the chunk has at least one key
        const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);  // 0 for the first chunk
        mrun = mnew;
        float psum = 0.f;
        x8 pf[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float pe = __builtin_amdgcn_exp2f(x[i] - mnew);
            psum += pe;
            pf[i >> 3][i & 7] = Elt<T>::from_f(pe);
        }
        lrun = lrun * alpha + psum;
        DEC_TS(13);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the V pieces have landed (LDS-DMA completion is a vmcnt event)
        __builtin_amdgcn_sched_barrier(0);
        DEC_TS(14);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            o[m] *= alpha;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const x4 lo = M::tr(vl + vrd0[m] + hf * 32 * ROWB), hi = M::tr(vl + vrd0[m] + (hf * 32 + 16) * ROWB);
                x8 vf;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    vf[j] = lo[j];
                    vf[4 + j] = hi[j];
                }
                o[m] = M::mma(vf, pf[hf], o[m]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // transposing reads done before the next chunk's DMA overwrites the tile
        __builtin_amdgcn_sched_barrier(0);
    }
    DEC_TS(15);
    // ---- partial record of this wave over its own V tile: m[16] | l[16] | o[16 slots][64]
    float ltot = lrun + __shfl_xor(lrun, 16, 64);
    ltot += __shfl_xor(ltot, 32, 64);
    float *pm = (float *)vl, *pl = pm + 16, *po = pm + 32;
    if (q == 0) {
        pm[rho] = mrun;
        pl[rho] = ltot;
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) *(f32x4 *)(po + rho * 64 + 16 * m + 4 * q) = o[m];
    __syncthreads();
    DEC_TS(16);
    // ---- merge: thread (head, d) walks the 16 partial records in wave order (fixed order: bitwise reproducible)
    const int tid = threadIdx.x;
    if (tid < h * 64) {
        const int hh = tid >> 6, d = tid & 63;
        float mm = -INFINITY;
#pragma unroll
        for (int w = 0; w < DEC_ATT_WAVES; ++w) mm = fmaxf(mm, ((const float *)(lds + w * DEC_ATT_TILE))[hh]);
        float acc = 0.f, l = 0.f;
        if (mm > -INFINITY) {
#pragma unroll
            for (int w = 0; w < DEC_ATT_WAVES; ++w) {
                const float *pw = (const float *)(lds + w * DEC_ATT_TILE);
                const float wgt = __builtin_amdgcn_exp2f(pw[hh] - mm);  // exp2(-inf) = 0 for a wave without a chunk
                l = fmaf(pw[16 + hh], wgt, l);
                acc = fmaf(pw[32 + hh * 64 + d], wgt, acc);
            }
        }
        ((T *)A.O)[(row * h + hh) * 64 + d] = Elt<T>::from_f(l > 0.f ? acc / l : 0.f);
    }
    DEC_TS(17);
}

}  // namespace nsa
