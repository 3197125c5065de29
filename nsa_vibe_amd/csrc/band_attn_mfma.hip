// MFMA band attention forward for gfx950 (bf16 / f16, Dk = Dv = 64, h <= 16): the sliding-window and the compressed
// branch of NSA (reference: sliding_window_attention, nsa/core/attention_kernels.py:146-178; compressed branch with the
// num_cmp(t) emission schedule, :106-143).  Query row t attends the contiguous key interval [lo(t), hi(t)) defined in
// sel_attn_params.hpp (BandAttnParams); rows with an empty interval give zeros.
//
// Mapping: one wave64 = TPW consecutive tokens of one (b,g) with all their h heads: NT*16 (token, head) "slots" are the
// columns of NT 16x16x32 MFMA tiles (h = 6, NT = 3: 8 tokens x 6 heads = 48 slots, every column used).  A K/V tile of 32
// keys is brought into wave-private LDS once (LDS-DMA) and its fragments are reused by all NT column tiles, so the band
// around TPW tokens costs (w + TPW) key rows per wave instead of w per token.  The product is computed transposed like
// the selection kernel (S^T = K.Q^T, O^T += V^T.P^T): softmax statistics are per lane (= per slot) and the S^T
// accumulators feed the PV MFMA without any lane movement.  Tiles that lie inside every slot's interval take the
// mask-free path with deferred max; only the tiles on the window edge / causal diagonal apply per-slot masks.
// A 256-thread workgroup = 4 consecutive token groups; all LDS is wave private (no workgroup barrier).
#include <stdlib.h>

#include "attn_mfma_tiles.hpp"
#include "band_attn_fwd_body.hpp"

namespace nsa {

template <typename T, int D, int NT, bool SPLIT, int STAGE>
__global__ __launch_bounds__(256, 2) void band_attn_fwd_kernel(BandAttnParams P) {
    band_attn_body<T, D, NT, SPLIT, STAGE>(P, blockIdx.x);
}

// decode: the sliding and the compressed branch of one step in ONE launch (two argument blocks, split-KV form)
template <typename T, int D>
__global__ __launch_bounds__(256, 2) void band_attn_fwd_dual_kernel(BandAttnParams P0, BandAttnParams P1, unsigned grid0) {
    if (blockIdx.x < grid0) band_attn_body<T, D, 1, true, 1>(P0, blockIdx.x);
    else band_attn_body<T, D, 1, true, 1>(P1, blockIdx.x - grid0);
}

// ------------------------------------------------------------------------------------------------------------------------
// Backward, dQ: the same 48-slot query-major walk as the forward.  P is recomputed from the forward's log-sum-exp, so there is
// no running max: per key tile  S^T = K.Q^T,  dP^T = V.dO^T,  dS^T = P^T o (dP^T - delta) * scale  (delta per slot = per lane
// column),  dQ^T[d, slot] += K^T[d x keys] . dS^T  -- dS^T goes from the accumulators straight into the B operand, K^T comes
// from a second LDS image of the K tile (transposing reads), V is staged in the row layout.  dK / dV of the band are taken
// by the key-block-major selection backward kernel (the band written as one range per row).
// ------------------------------------------------------------------------------------------------------------------------
struct BandBwdExtra {
    const void *dO;      // [B,S,G,h,D]
    const float *lse;    // [B,S,G,h]  natural-log log-sum-exp of the forward (-inf: empty row)
    const float *delta;  // [B,S,G,h]  rowsum(dO * O)
    void *dQ;            // [B,S,G,h,D]
};

template <typename T, int NT>
__global__ __launch_bounds__(256, 2) void band_attn_bwd_dq_kernel(BandAttnParams P, BandBwdExtra E) {
    using M = MfmaT<T>;
    using G_ = Geo<64>;
    using x8 = typename M::x8;
    using x4 = typename M::x4;
    constexpr int D = 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id();
    const int wave = uniform((int)(threadIdx.x >> 6));
    const int tpw = P.tpw, h = P.h;
    const int ngrp = (P.S + tpw - 1) / tpw;
    const int nbg = P.B * P.G;
    const int W = (ngrp + 3) >> 2;
    int bg, tc;
    if (P.map_mode == 2) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        bg = (idx / W) * 8 + xcd;
        tc = idx % W;
    } else {
        bg = blockIdx.x / W;
        tc = blockIdx.x % W;
    }
    const int grp = 4 * tc + wave;
    if (grp >= ngrp || bg >= nbg) return;
    const int b = bg / P.G, g = bg - b * P.G;
    const int tw0 = grp * tpw, ntok = min(tpw, P.S - tw0);
    unsigned char *kl = smem + (size_t)wave * (3 * G_::TILE_BYTES);  // K rows | K transposable | V rows
    unsigned char *ktl = kl + G_::TILE_BYTES, *vl = ktl + G_::TILE_BYTES;

    const int hi_min = band_hi(P.t0, P.a, P.dd, P.c, P.S_kv, tw0);
    const int hi_max = band_hi(P.t0, P.a, P.dd, P.c, P.S_kv, tw0 + ntok - 1);
    const int klo = max(0, hi_min - P.w), lo_max = max(0, hi_max - P.w);
    const int rho = lane & 15, q = lane >> 4;
    int hi_s[NT], lo_s[NT];
    int64_t orow[NT];
    float lse2[NT], dlt[NT];
    x8 qf[NT][2], dof[NT][2];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int slot = 16 * n + rho, tok = slot / h, head = slot - tok * h;
        const bool used = tok < ntok;
        const int t = tw0 + tok;
        hi_s[n] = used ? band_hi(P.t0, P.a, P.dd, P.c, P.S_kv, t) : 0;
        lo_s[n] = max(0, hi_s[n] - P.w);
        orow[n] = used ? ((((int64_t)b * P.S + t) * P.G + g) * h + head) : -1;
        const float l = used ? E.lse[orow[n]] : -INFINITY;
        lse2[n] = l > -INFINITY ? l * LOG2E : INFINITY;  // empty / unused slot: exp2(s - inf) = 0
        dlt[n] = used ? E.delta[orow[n]] : 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 rq = {0u, 0u, 0u, 0u}, rd = {0u, 0u, 0u, 0u};
            if (used) {
                rq = *(const u32x4 *)((const T *)P.Q + orow[n] * D + 32 * s + 8 * q);
                rd = *(const u32x4 *)((const T *)E.dO + orow[n] * D + 32 * s + 8 * q);
            }
            qf[n][s] = __builtin_bit_cast(x8, rq);
            dof[n][s] = __builtin_bit_cast(x8, rd);
        }
    }

    const unsigned char *Kb = (const unsigned char *)((const T *)P.K + (int64_t)b * P.ksb + (int64_t)g * P.ksg);
    const unsigned char *Vb = (const unsigned char *)((const T *)P.V + (int64_t)b * P.vsb + (int64_t)g * P.vsg);
    const int64_t krowb = P.kss * 2, vrowb = P.vss * 2;
    auto make_rsrc = [&](const unsigned char *base, int64_t bytes) {
        const uint64_t a = (uint64_t)base;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), (short)0,
                                                 __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
    };
    [[maybe_unused]] const auto krs = make_rsrc(Kb, (int64_t)(P.S_kv - 1) * krowb + G_::ROWB);
    [[maybe_unused]] const auto vrs = make_rsrc(Vb, (int64_t)(P.S_kv - 1) * vrowb + G_::ROWB);
    [[maybe_unused]] const int krowb32 = uniform((int)krowb), vrowb32 = uniform((int)vrowb);
    [[maybe_unused]] const int ld_row = lane / G_::PIECES, ld_piece = lane % G_::PIECES;
    uint32_t krd0[2], trd0[4];
#pragma unroll
    for (int s = 0; s < 2; ++s) krd0[s] = rho * G_::ROWB + (((4 * s + q) ^ G_::swz_k(rho)) << 4);
    {
        const int qq = rho >> 2, pp = rho & 3, r = 4 * q + qq;
#pragma unroll
        for (int m = 0; m < 4; ++m) trd0[m] = r * G_::ROWB + ((m ^ G_::swz_v(r)) << 5) + 8 * pp;
    }
    // one 32-key tile -> three wave-private images (rows past the end of K/V re-read the last row; they are masked)
    auto issue_dma = [&](int tok0) {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef __attribute__((address_space(3))) void lds_void;
        const int ks = uniform(tok0 * krowb32), vs = uniform(tok0 * vrowb32);
        const int last = P.S_kv - 1 - tok0;
#pragma unroll
        for (int i = 0; i < G_::NLD; ++i) {
            const int r = i * G_::RPI + ld_row;
            const int rc = min(r, last);
            const uint32_t pk = (uint32_t)((ld_piece ^ G_::swz_k(r)) << 4);
            const uint32_t pt = (uint32_t)(((((ld_piece >> 1) ^ G_::swz_v(r)) << 1) | (ld_piece & 1)) << 4);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (lds_void *)(kl + i * 1024), 16, rc * krowb32 + pk, ks, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (lds_void *)(ktl + i * 1024), 16, rc * krowb32 + pt, ks, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (lds_void *)(vl + i * 1024), 16, rc * vrowb32 + pk, vs, 0, 0);
        }
#else
        (void)tok0;
#endif
    };

    f32x4 dq[NT][4];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < 4; ++m) dq[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float c2 = P.scale * LOG2E;
    const int ntile = hi_max > klo ? (hi_max - klo + 31) >> 5 : 0;
    if (ntile > 0) issue_dma(klo);
    for (int tile = 0; tile < ntile; ++tile) {
        const int tok0 = klo + 32 * tile;
        x8 kfr[2][2], vfr[2][2], ka[4];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                kfr[u][s] = *(const x8 *)(kl + krd0[s] + u * 16 * G_::ROWB);
                vfr[u][s] = *(const x8 *)(vl + krd0[s] + u * 16 * G_::ROWB);
            }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const x4 lo = M::tr(ktl + trd0[m]), hi = M::tr(ktl + trd0[m] + 16 * G_::ROWB);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ka[m][j] = lo[j];
                ka[m][4 + j] = hi[j];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (tile + 1 < ntile) issue_dma(tok0 + 32);
        const bool interior = tok0 >= lo_max && tok0 + 32 <= hi_min;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            f32x4 sacc[2], dpacc[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                sacc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                dpacc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    sacc[u] = M::mma(kfr[u][s], qf[n][s], sacc[u]);
                    dpacc[u] = M::mma(vfr[u][s], dof[n][s], dpacc[u]);
                }
            }
            x8 dsf;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float p = __builtin_amdgcn_exp2f(fmaf(sacc[u][j], c2, -lse2[n]));
                    if (!interior) {  // wave uniform: only the window-edge / diagonal tiles pay for the masks
                        const int key = tok0 + 16 * u + 4 * q + j;
                        p = (key >= lo_s[n] && key < hi_s[n]) ? p : 0.f;
                    }
                    dsf[4 * u + j] = Elt<T>::from_f(p * (dpacc[u][j] - dlt[n]) * P.scale);
                }
#pragma unroll
            for (int m = 0; m < 4; ++m) dq[n][m] = M::mma(ka[m], dsf, dq[n][m]);
        }
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        if (orow[n] < 0) continue;
        T *dst = (T *)E.dQ + orow[n] * D;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            x4 ov;
#pragma unroll
            for (int j = 0; j < 4; ++j) ov[j] = Elt<T>::from_f(dq[n][m][j]);
            *(x4 *)(dst + 16 * m + 4 * q) = ov;
        }
    }
}

// ---- host side ----------------------------------------------------------------------------
bool band_attn_mfma_supported(int dtype, int h, int Dk, int Dv) {
    return (dtype == NSA_DT_BF16 || dtype == NSA_DT_F16) && Dk == Dv && (Dk == 64 || Dk == 128) && h >= 1 && h <= 16;
}

// NT = 3 column tiles per wave (48 / h tokens) once there are enough tokens to fill the chip, NT = 1 (16 / h tokens)
// below that; with very few token groups (decode) the key interval is also split over several waves.
// (D = 128: 2 column tiles, the accumulators are twice as wide)
static void band_plan(int B, int S, int G, int h, int D, int *nt, int *tpw, int *nsplit) {
    const int ntw = D == 64 ? 3 : 2;
    const int tpw3 = 16 * ntw / h, tpw1 = 16 / h;
    const int64_t grp3 = (int64_t)B * G * ((S + tpw3 - 1) / tpw3);
    if (grp3 >= 2048) {
        *nt = ntw, *tpw = tpw3, *nsplit = 1;
        return;
    }
    *nt = 1, *tpw = tpw1;
    const int64_t grp1 = (int64_t)B * G * ((S + tpw1 - 1) / tpw1);
    int ns = 1;
    if (grp1 < 1024) {
        ns = (int)((2048 + grp1 - 1) / grp1);
        if (ns > 16) ns = 16;
    }
    *nsplit = ns;
}

size_t band_attn_workspace(int B, int S, int G, int h, int Dk, int Dv, int dtype, int *nsplit_out) {
    int nt = 1, tpw = 1, ns = 1;
    if (band_attn_mfma_supported(dtype, h, Dk, Dv)) band_plan(B, S, G, h, Dk, &nt, &tpw, &ns);
    if (nsplit_out) *nsplit_out = ns;
    return ns > 1 ? (size_t)B * S * G * ns * h * (Dv + PART_PAD) * sizeof(float) : 0;
}

template <typename T, int D>
static int launch_band_t(const BandAttnParams &P0, hipStream_t st) {
    BandAttnParams P = P0;
    int nt = 1, ns = 1;
    band_plan(P.B, P.S, P.G, P.h, D, &nt, &P.tpw, &ns);
    const bool split = ns > 1 && P.part != nullptr && P.nsplit == ns;
    if (!split) P.nsplit = 1;
    const int64_t nbg = (int64_t)P.B * P.G;
    const int ngrp = (P.S + P.tpw - 1) / P.tpw;
    const size_t lds = 4 * (size_t)(2 * Geo<D>::TILE_BYTES);
    if (split) {
        const int64_t waves = nbg * ngrp * P.nsplit;
        hipLaunchKernelGGL((band_attn_fwd_kernel<T, D, 1, true, 1>), dim3((unsigned)((waves + 3) / 4)), dim3(256), lds, st, P);
        NSA_LAUNCH_CHECK("band_attn_fwd(split)");
        if (P.defer_combine) return NSA_OK;
        SelAttnParams C{};
        C.O = P.O;
        C.lse = P.lse;
        C.R = nbg * P.S;
        C.h = P.h;
        C.part = P.part;
        C.nsplit = P.nsplit;
        hipLaunchKernelGGL((sel_attn_combine_kernel<T, D>), dim3((unsigned)((C.R * C.h + 3) / 4)), dim3(256), 0, st, C);
        NSA_LAUNCH_CHECK("band_attn_combine");
        return NSA_OK;
    }
    const int64_t W = (ngrp + 3) / 4;
    P.map_mode = (nbg % 8 == 0) ? 2 : 1;
    NSA_CHECK_ARG(nbg * W < ((int64_t)1 << 31), "band_attn: too many workgroups for one launch");
    const unsigned grid = (unsigned)(nbg * W);
    const int stage = tuning(TUNE_BAND_STAGE);  // A/B switch: 0 = register staging, 1 = LDS-DMA  // measured: LDS-DMA 750 vs register staging 650 TFLOP/s (compressed branch, 64k)
    constexpr int NTW = D == 64 ? 3 : 2;
    bool done = false;
    if constexpr (D == 64) {
        if (nt == NTW && stage == 0) {
            hipLaunchKernelGGL((band_attn_fwd_kernel<T, D, NTW, false, 0>), dim3(grid), dim3(256), lds, st, P);
            done = true;
        }
    }
    if (done) {
    } else if (nt == NTW) hipLaunchKernelGGL((band_attn_fwd_kernel<T, D, NTW, false, 1>), dim3(grid), dim3(256), lds, st, P);
    else hipLaunchKernelGGL((band_attn_fwd_kernel<T, D, 1, false, 1>), dim3(grid), dim3(256), lds, st, P);
    NSA_LAUNCH_CHECK("band_attn_fwd");
    return NSA_OK;
}

int launch_band_attn_fwd_mfma(const BandAttnParams &P, int dtype, hipStream_t st) {
    NSA_CHECK_ARG(band_attn_mfma_supported(dtype, P.h, P.Dk, P.Dv), "band MFMA kernel: unsupported dtype/h/Dk/Dv = %d/%d/%d/%d", dtype, P.h,
                  P.Dk, P.Dv);
    NSA_CHECK_ARG(P.kss % 8 == 0 && P.vss % 8 == 0 && P.ksb % 8 == 0 && P.vsb % 8 == 0 && P.ksg % 8 == 0 && P.vsg % 8 == 0,
                  "band MFMA kernel: K/V strides must be multiples of 8 elements (16 B)");
    NSA_CHECK_ARG(((uintptr_t)P.Q % 16 == 0) && ((uintptr_t)P.K % 16 == 0) && ((uintptr_t)P.V % 16 == 0) && ((uintptr_t)P.O % 8 == 0),
                  "band MFMA kernel: Q/K/V must be 16-byte aligned");
    NSA_CHECK_ARG((int64_t)P.S_kv * P.kss * 2 < ((int64_t)1 << 31) && (int64_t)P.S_kv * P.vss * 2 < ((int64_t)1 << 31),
                  "band MFMA kernel: one (b,g) K/V slab must be smaller than 2 GiB (buffer addressing)");
    if (dtype == NSA_DT_BF16) return P.Dk == 64 ? launch_band_t<__bf16, 64>(P, st) : launch_band_t<__bf16, 128>(P, st);
    return P.Dk == 64 ? launch_band_t<_Float16, 64>(P, st) : launch_band_t<_Float16, 128>(P, st);
}

// Both branches of a decode step in one launch.  Requires both to be in split form with deferred combine (the caller combines).
bool band_dual_plan(BandAttnParams *P0, BandAttnParams *P1, int dtype, int64_t waves[2]) {
    BandAttnParams *P[2] = {P0, P1};
    for (int i = 0; i < 2; ++i) {
        int nt = 1, ns = 1;
        band_plan(P[i]->B, P[i]->S, P[i]->G, P[i]->h, P[i]->Dk, &nt, &P[i]->tpw, &ns);
        if (!(nt == 1 && ns > 1 && P[i]->part && P[i]->nsplit == ns && P[i]->defer_combine)) return false;
        waves[i] = (int64_t)P[i]->B * P[i]->G * ((P[i]->S + P[i]->tpw - 1) / P[i]->tpw) * ns;
    }
    return P0->Dk == P1->Dk && band_attn_mfma_supported(dtype, P0->h, P0->Dk, P0->Dv);
}

int launch_band_attn_fwd_dual(const BandAttnParams &A0, const BandAttnParams &A1, int dtype, hipStream_t st) {
    BandAttnParams P[2] = {A0, A1};
    int64_t waves[2];
    NSA_CHECK_ARG(band_dual_plan(&P[0], &P[1], dtype, waves), "band dual launch: both branches in split form with deferred combine, one supported shape");
    const unsigned grid[2] = {(unsigned)((waves[0] + 3) / 4), (unsigned)((waves[1] + 3) / 4)};
    const bool d64 = P[0].Dk == 64;
    const size_t lds = 4 * (size_t)(2 * (d64 ? Geo<64>::TILE_BYTES : Geo<128>::TILE_BYTES));
    const dim3 g2(grid[0] + grid[1]);
    if (dtype == NSA_DT_BF16) {
        if (d64) hipLaunchKernelGGL((band_attn_fwd_dual_kernel<__bf16, 64>), g2, dim3(256), lds, st, P[0], P[1], grid[0]);
        else hipLaunchKernelGGL((band_attn_fwd_dual_kernel<__bf16, 128>), g2, dim3(256), lds, st, P[0], P[1], grid[0]);
    } else {
        if (d64) hipLaunchKernelGGL((band_attn_fwd_dual_kernel<_Float16, 64>), g2, dim3(256), lds, st, P[0], P[1], grid[0]);
        else hipLaunchKernelGGL((band_attn_fwd_dual_kernel<_Float16, 128>), g2, dim3(256), lds, st, P[0], P[1], grid[0]);
    }
    NSA_LAUNCH_CHECK("band_attn_fwd_dual");
    return NSA_OK;
}

// dQ of the band (bf16/f16, Dk = Dv = 64).  delta = rowsum(dO * O) [B,S,G,h] fp32 comes from the caller.
template <typename T>
static int launch_band_dq_t(BandAttnParams P, const BandBwdExtra &E, hipStream_t st) {
    // 48 slots per wave when there are enough token groups, else 16
    const int tpw3 = 48 / P.h, tpw1 = 16 / P.h;
    const int64_t nbg = (int64_t)P.B * P.G;
    const bool big = nbg * ((P.S + tpw3 - 1) / tpw3) >= 2048;
    P.tpw = big ? tpw3 : tpw1;
    const int ngrp = (P.S + P.tpw - 1) / P.tpw;
    const int64_t W = (ngrp + 3) / 4;
    P.map_mode = (nbg % 8 == 0) ? 2 : 1;
    NSA_CHECK_ARG(nbg * W < ((int64_t)1 << 31), "band_attn_bwd: too many workgroups for one launch");
    const size_t lds = 4 * (size_t)(3 * Geo<64>::TILE_BYTES);
    if (big) hipLaunchKernelGGL((band_attn_bwd_dq_kernel<T, 3>), dim3((unsigned)(nbg * W)), dim3(256), lds, st, P, E);
    else hipLaunchKernelGGL((band_attn_bwd_dq_kernel<T, 1>), dim3((unsigned)(nbg * W)), dim3(256), lds, st, P, E);
    NSA_LAUNCH_CHECK("band_attn_bwd_dq");
    return NSA_OK;
}

int launch_band_attn_bwd_dq(const BandAttnParams &P, const void *dO, const float *lse, const float *delta, void *dQ, int dtype,
                            hipStream_t st) {
    NSA_CHECK_ARG(band_attn_mfma_supported(dtype, P.h, P.Dk, P.Dv) && P.Dk == 64, "band dQ kernel: unsupported dtype/h/Dk/Dv");
    BandBwdExtra E{dO, lse, delta, dQ};
    if (dtype == NSA_DT_BF16) return launch_band_dq_t<__bf16>(P, E, st);
    return launch_band_dq_t<_Float16>(P, E, st);
}

}  // namespace nsa
