// MFMA selection-attention forward for gfx950 (bf16 / f16, Dk = Dv = D in {64,128}, h <= 16).
//
// Mapping (one wave64 = one query row (b,t,g); a 256-thread workgroup = 4 independent rows):
//   * The h query heads of the GQA group are the 16 columns of a 16x16x32 MFMA (columns >= h are
//     zero padding), so every K/V byte fetched is shared by all heads of the group -- K/V are read
//     once per group, never per head (the reference's Triton kernels re-read them per head,
//     nsa/kernels/triton_sel_kernel/sel_fwd.py:143-238).
//   * The selected ranges are normalised to sorted disjoint segments (union semantics of
//     attention_kernels.py:721-732) and walked in 32-token tiles; tail tiles are masked.
//   * S^T[key, head] = K_tile[32 x D] . Q^T[D x 16]  : A = K rows from LDS (ds_read_b128, XOR
//     swizzled 16-B pieces), B = Q^T fragments held in registers for the whole row.
//   * online softmax in fp32 (exp2 domain); keys live in the accumulator registers and the 4 lane
//     groups, so the row max needs two cross-lane steps per tile and the row sum none until the end.
//   * O^T[dv, head] += V^T[dv x 32 keys] . P^T[32 keys x 16] : B = P^T taken straight from the S^T
//     accumulators (cvt to bf16, no lane movement: the k index of the PV MFMA is mapped onto the
//     accumulator's key order), A = V^T read with ds_read_b64_tr_b16 (hardware transpose) from a
//     row-major, XOR-swizzled V tile.
//   * HBM/L2 -> LDS staging is register staged with one tile of prefetch: the global loads of tile
//     i+1 are issued before the MFMAs of tile i.  Every global load instruction covers whole 128-B
//     (D=64) / 256-B (D=128) rows: lanes are row-linear, 16 B per lane.
// All LDS is wave private: no workgroup barrier anywhere in the kernel.
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

template <typename T, int D, bool SPLIT>
__global__ __launch_bounds__(256) void sel_attn_fwd_mfma_kernel(SelAttnParams P) {
    using M = MfmaT<T>;
    using G_ = Geo<D>;
    using x8 = typename M::x8;
    using x4 = typename M::x4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id();
    const int wave = uniform((int)(threadIdx.x >> 6));
    const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
    const int nsplit = SPLIT ? P.nsplit : 1;
    const int64_t row = SPLIT ? wid / nsplit : wid;
    const int sp = SPLIT ? (int)(wid % nsplit) : 0;
    if (row >= P.R) return;

    unsigned char *kl = smem + (size_t)wave * G_::WAVE_LDS;
    unsigned char *vl = kl + G_::TILE_BYTES;
    int *seg = (int *)(vl + G_::TILE_BYTES);

    const int h = P.h;
    const int g = (int)(row % P.G);
    const int b = (int)(row / ((int64_t)P.G * P.S));
    const T *Kb = (const T *)P.K + (int64_t)b * P.ksb + (int64_t)g * P.ksg;
    const T *Vb = (const T *)P.V + (int64_t)b * P.vsb + (int64_t)g * P.vsg;
    const T *Qr = (const T *)P.Q + row * (int64_t)h * D;

    int nseg;
    const int L = normalise_ranges(P.ranges + row * (int64_t)P.n * 2, P.n, P.S_kv, seg, &nseg);

    const int rho = lane & 15;  // MFMA row/col index of this lane
    const int q = lane >> 4;    // k-chunk group of this lane

    // ---- Q^T fragments (B operand): lane holds Q[head rho][32 s + 8 q .. +7]
    x8 qf[G_::KSTEPS];
#pragma unroll
    for (int s = 0; s < G_::KSTEPS; ++s) {
        u32x4 raw = {0u, 0u, 0u, 0u};
        if (rho < h) raw = *(const u32x4 *)(Qr + (int64_t)rho * D + 32 * s + 8 * q);
        qf[s] = __builtin_bit_cast(x8, raw);
    }

    // ---- tile iterator over the segments (all wave-uniform)
    int it_seg = 0, it_pos = 0, it_cnt = 0;
    auto next_tile = [&](int &tok0, int &nvalid) -> bool {
        while (true) {
            if (it_seg >= nseg) return false;
            const int off0 = uniform(seg[2 * it_seg + 1]);
            const int off1 = uniform(seg[2 * it_seg + 3]);
            const int len = off1 - off0;
            if (it_pos >= len) {
                ++it_seg;
                it_pos = 0;
                continue;
            }
            tok0 = uniform(seg[2 * it_seg]) + it_pos;
            nvalid = min(32, len - it_pos);
            it_pos += 32;
            if (SPLIT) {
                const bool mine = (it_cnt % nsplit) == sp;
                ++it_cnt;
                if (!mine) continue;
            }
            return true;
        }
    };

    const int ld_row = lane / G_::PIECES;    // row within one wave-wide load
    const int ld_piece = lane % G_::PIECES;  // 16-B piece within the row
    u32x4 kreg[G_::NLD], vreg[G_::NLD];
    auto issue_loads = [&](int tok0, int nvalid) {
#pragma unroll
        for (int i = 0; i < G_::NLD; ++i) {
            const int r = i * G_::RPI + ld_row;
            const int64_t t = tok0 + min(r, nvalid - 1);
            kreg[i] = *(const u32x4 *)(Kb + t * P.kss + ld_piece * 8);
            vreg[i] = *(const u32x4 *)(Vb + t * P.vss + ld_piece * 8);
        }
    };

    f32x4 o[G_::MT];
#pragma unroll
    for (int m = 0; m < G_::MT; ++m) o[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrun = -INFINITY, lrun = 0.f;
    const float c2 = P.scale * LOG2E;

    int tok0 = 0, nvalid = 0;
    bool have = next_tile(tok0, nvalid);
    if (have) issue_loads(tok0, nvalid);

    while (have) {
        const int cur_nvalid = nvalid;
        // ---- stage registers -> LDS (swizzled)
#pragma unroll
        for (int i = 0; i < G_::NLD; ++i) {
            const int r = i * G_::RPI + ld_row;
            *(u32x4 *)(kl + r * G_::ROWB + ((ld_piece ^ G_::swz_k(r)) << 4)) = kreg[i];
            *(u32x4 *)(vl + r * G_::ROWB + ((((ld_piece >> 1) ^ G_::swz_v(r)) << 5) | ((ld_piece & 1) << 4))) = vreg[i];
        }
        // ---- prefetch the next tile while this one is consumed
        have = next_tile(tok0, nvalid);
        if (have) issue_loads(tok0, nvalid);
        wave_lds_fence();

        // ---- S^T = K . Q^T
        f32x4 sacc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            sacc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int r = 16 * u + rho;
#pragma unroll
            for (int s = 0; s < G_::KSTEPS; ++s) {
                const x8 a = *(const x8 *)(kl + r * G_::ROWB + (((4 * s + q) ^ G_::swz_k(r)) << 4));
                sacc[u] = M::mma(a, qf[s], sacc[u]);
            }
        }
        // ---- online softmax (exp2 domain); key of sacc[u][j] is 16u + 4q + j
        float x[8];
        float tmax = -INFINITY;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int key = 16 * u + 4 * q + j;
                const float v = (key < cur_nvalid) ? sacc[u][j] * c2 : -INFINITY;
                x[4 * u + j] = v;
                tmax = fmaxf(tmax, v);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mnew = fmaxf(mrun, tmax);
        const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
        mrun = mnew;
        float psum = 0.f;
        x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float pe = __builtin_amdgcn_exp2f(x[j] - mnew);
            psum += pe;
            pf[j] = Elt<T>::from_f(pe);
        }
        lrun = lrun * alpha + psum;
#pragma unroll
        for (int m = 0; m < G_::MT; ++m) o[m] *= alpha;

        // ---- O^T += V^T . P^T   (k index j<4 -> key 4q+j, j>=4 -> key 16+4q+(j-4))
        const int qq = rho >> 2, pp = rho & 3;
        const int r0 = 4 * q + qq, r1 = 16 + 4 * q + qq;
#pragma unroll
        for (int m = 0; m < G_::MT; ++m) {
            const x4 lo = M::tr(vl + r0 * G_::ROWB + ((m ^ G_::swz_v(r0)) << 5) + 8 * pp);
            const x4 hi = M::tr(vl + r1 * G_::ROWB + ((m ^ G_::swz_v(r1)) << 5) + 8 * pp);
            x8 a;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = lo[j];
                a[4 + j] = hi[j];
            }
            o[m] = M::mma(a, pf, o[m]);
        }
        wave_lds_fence();  // LDS tile is rewritten at the top of the next iteration
    }

    // ---- epilogue
    float ltot = lrun + __shfl_xor(lrun, 16, 64);
    ltot += __shfl_xor(ltot, 32, 64);
    if (SPLIT) {
        // partial record [row][sp][head][2 + D]: m (log2 domain), l, unnormalised O
        float *pr = P.part + ((row * nsplit + sp) * (int64_t)h + rho) * (D + PART_PAD);
        if (rho < h) {
            if (q == 0) {
                pr[0] = mrun;
                pr[1] = ltot;
            }
#pragma unroll
            for (int m = 0; m < G_::MT; ++m) *(f32x4 *)(pr + PART_PAD + 16 * m + 4 * q) = o[m];
        }
        return;
    }
    if (rho < h) {
        const float inv = (L > 0) ? 1.f / ltot : 0.f;
        T *Or = (T *)P.O + (row * (int64_t)h + rho) * D;
#pragma unroll
        for (int m = 0; m < G_::MT; ++m) {
            x4 ov;
#pragma unroll
            for (int j = 0; j < 4; ++j) ov[j] = Elt<T>::from_f(o[m][j] * inv);
            *(x4 *)(Or + 16 * m + 4 * q) = ov;
        }
        if (P.lse && q == 0) P.lse[row * h + rho] = (L > 0) ? (mrun + __builtin_amdgcn_logf(ltot)) * LN2 : -INFINITY;
    }
}

// combine split-KV partials: one wave per row, lane = dv
template <typename T, int D>
__global__ __launch_bounds__(256) void sel_attn_combine_kernel(SelAttnParams P) {
    const int lane = lane_id();
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= P.R) return;
    const int h = P.h, ns = P.nsplit;
    for (int hh = 0; hh < h; ++hh) {
        float mmax = -INFINITY;
        for (int s = 0; s < ns; ++s) mmax = fmaxf(mmax, P.part[((row * ns + s) * (int64_t)h + hh) * (D + PART_PAD)]);
        float ltot = 0.f;
        float acc[D / 64];
#pragma unroll
        for (int c = 0; c < D / 64; ++c) acc[c] = 0.f;
        for (int s = 0; s < ns; ++s) {
            const float *pr = P.part + ((row * ns + s) * (int64_t)h + hh) * (D + PART_PAD);
            const float w = (pr[0] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(pr[0] - mmax);
            ltot += pr[1] * w;
#pragma unroll
            for (int c = 0; c < D / 64; ++c) acc[c] += pr[PART_PAD + c * 64 + lane] * w;
        }
        const float inv = (ltot > 0.f) ? 1.f / ltot : 0.f;
        T *Or = (T *)P.O + (row * (int64_t)h + hh) * D;
#pragma unroll
        for (int c = 0; c < D / 64; ++c) Or[c * 64 + lane] = Elt<T>::from_f(acc[c] * inv);
        if (P.lse && lane == 0) P.lse[row * h + hh] = (ltot > 0.f) ? (mmax + __builtin_amdgcn_logf(ltot)) * LN2 : -INFINITY;
    }
}

// ---- host side ----------------------------------------------------------------------------
bool sel_attn_mfma_supported(int dtype, int h, int Dk, int Dv) {
    return (dtype == NSA_DT_BF16 || dtype == NSA_DT_F16) && Dk == Dv && (Dk == 64 || Dk == 128) && h >= 1 && h <= 16;
}

// Few rows (decode): split each row's tiles over several waves so the launch fills the chip.
static int pick_nsplit(int64_t R) {
    const int64_t target = 256 * 8;  // waves wanted in flight
    if (R >= target / 2) return 1;
    int ns = (int)((target + R - 1) / R);
    if (ns > 16) ns = 16;  // a row has at most n*l'/32 = 32 tiles at the m7c shape
    return ns < 1 ? 1 : ns;
}

size_t sel_attn_mfma_workspace(int64_t R, int h, int Dv, int *nsplit_out) {
    const int ns = pick_nsplit(R);
    if (nsplit_out) *nsplit_out = ns;
    return ns > 1 ? (size_t)R * ns * h * (Dv + PART_PAD) * sizeof(float) : 0;
}

template <typename T, int D>
static int launch_mfma_t(const SelAttnParams &P0, hipStream_t st) {
    SelAttnParams P = P0;
    const size_t lds = 4 * (size_t)Geo<D>::WAVE_LDS;
    const bool split = P.part != nullptr && P.nsplit > 1;
    if (!split) P.nsplit = 1;
    const int64_t waves = P.R * P.nsplit;
    const unsigned grid = (unsigned)((waves + 3) / 4);
    auto k = split ? sel_attn_fwd_mfma_kernel<T, D, true> : sel_attn_fwd_mfma_kernel<T, D, false>;
    if (lds > 64 * 1024) NSA_HIP_TRY(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, P);
    NSA_LAUNCH_CHECK("sel_attn_fwd_mfma");
    if (split) {
        hipLaunchKernelGGL((sel_attn_combine_kernel<T, D>), dim3((unsigned)((P.R + 3) / 4)), dim3(256), 0, st, P);
        NSA_LAUNCH_CHECK("sel_attn_combine");
    }
    return NSA_OK;
}

int launch_sel_attn_fwd_mfma(const SelAttnParams &P, int dtype, hipStream_t st) {
    NSA_CHECK_ARG(sel_attn_mfma_supported(dtype, P.h, P.Dk, P.Dv), "MFMA kernel: unsupported dtype/h/Dk/Dv = %d/%d/%d/%d", dtype, P.h, P.Dk, P.Dv);
    NSA_CHECK_ARG(P.kss % 8 == 0 && P.vss % 8 == 0 && P.ksb % 8 == 0 && P.vsb % 8 == 0 && P.ksg % 8 == 0 && P.vsg % 8 == 0,
                  "MFMA kernel: K/V strides must be multiples of 8 elements (16 B)");
    NSA_CHECK_ARG(((uintptr_t)P.Q % 16 == 0) && ((uintptr_t)P.K % 16 == 0) && ((uintptr_t)P.V % 16 == 0) && ((uintptr_t)P.O % 8 == 0),
                  "MFMA kernel: Q/K/V must be 16-byte aligned");
    if (dtype == NSA_DT_BF16) return P.Dk == 64 ? launch_mfma_t<__bf16, 64>(P, st) : launch_mfma_t<__bf16, 128>(P, st);
    return P.Dk == 64 ? launch_mfma_t<_Float16, 64>(P, st) : launch_mfma_t<_Float16, 128>(P, st);
}

}  // namespace nsa
