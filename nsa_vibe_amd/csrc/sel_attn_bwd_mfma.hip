// MFMA backward of the selection attention for gfx950 (bf16 / f16, Dk = Dv = 64, h <= 16).
//
// Reference: autograd of the masked SDPA (nsa/core/attention_kernels.py:705-772); the structure of the reference's
// analytic backward (nsa/kernels/triton_sel_kernel/__init__.py:163-231: recompute P, dS = P*(dP - delta), dQ/dK/dV)
// without its first-key quirk (:217-219).  P is recomputed from the forward's log-sum-exp.
//
// Three kernels, no atomics, bitwise reproducible:
//   1. delta[row,h] = sum_dv dO*O.
//   2. dQ, query-major (one wave per query row, the forward's mapping): per 32-key tile
//        S^T = K.Q^T, dP^T = V.dO^T, dS^T = exp2(S^T c - lse) * (dP^T - delta) * scale, dQ^T += K^T.dS^T
//      (K is staged twice in LDS: a row image for the S^T A-operand and a transposable image for the dQ^T A-operand.)
//   3. dK, dV, KEY-BLOCK-major: one workgroup owns 64 keys of one (b,g) and keeps their dK/dV (64x64 fp32 each) in MFMA
//      accumulators while it sweeps every query row that selected the block.  The rows are found by scanning the range
//      lists (lane = row) and compacted in ascending t, so the summation order is fixed.  Rows are processed two at a
//      time in round 1 (2 x 6 heads = 12 of the 16 MFMA rows); round 2 lays the (row, head) slots end to end, every tile full:
//      S = Q.K^T, dP = dO.V^T (16x16x32), then dV += P^T.dO and
//      dK += dS^T.Q as 16x16x16 MFMAs whose A operand is taken straight from the S/dP accumulators (their row index,
//      the (query,head) slot, is the contraction index) and whose B operand is read transposed from LDS.
#include "nsa_common.hpp"
#include "sel_attn_params.hpp"

namespace nsa {

typedef __attribute__((ext_vector_type(4))) short s16x4;

template <typename T>
struct BwdT;
template <>
struct BwdT<__bf16> {
    using x8 = bf16x8;
    using x4 = bf16x4;
    __device__ static f32x4 mma32(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    __device__ static f32x4 mma16(x4 a, x4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
    }
};
template <>
struct BwdT<_Float16> {
    using x8 = f16x8;
    using x4 = f16x4;
    __device__ static f32x4 mma32(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    __device__ static f32x4 mma16(x4 a, x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }
};

template <typename X4>
__device__ __forceinline__ X4 tr_read(const unsigned char *p) {
    const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p);
    return __builtin_bit_cast(X4, r);
}

constexpr int BD = 64;           // head dim handled here
constexpr int BROWB = BD * 2;    // 128-B rows
__device__ __forceinline__ int bswz_row(int r) { return r & 7; }          // 16-B piece XOR for b128 row reads
__device__ __forceinline__ int bswz_tr(int r) { return (r >> 1) & 3; }    // 32-B chunk XOR for transposed reads
__device__ __forceinline__ uint32_t off_row_img(int r, int piece) { return r * BROWB + ((piece ^ bswz_row(r)) << 4); }
__device__ __forceinline__ uint32_t off_tr_img(int r, int piece) {
    return r * BROWB + ((((piece >> 1) ^ bswz_tr(r)) << 5) | ((piece & 1) << 4));
}

// ------------------------------------------------------------------------------------------ 1. delta
// Dv = 64: 8 lanes per (row, head), 16 bytes per lane -- a wave reads 8 whole rows of O and of dO per instruction (one thread
// per row made every lane walk its own 128-byte row: 0.9 TB/s)
template <typename T>
__global__ __launch_bounds__(256) void bwd_delta_kernel(const T *__restrict__ O, const T *__restrict__ dO, float *__restrict__ delta,
                                                         int64_t n_rows, int Dv) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3;  // (row, head)
    const int sub = threadIdx.x & 7;
    float acc = 0.f;
    if (i < n_rows) {
        const u32x4 a = *(const u32x4 *)(O + i * Dv + 8 * sub), b = *(const u32x4 *)(dO + i * Dv + 8 * sub);
        const T *pa = (const T *)&a, *pb = (const T *)&b;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc = fmaf(Elt<T>::to_f(pa[k]), Elt<T>::to_f(pb[k]), acc);
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (i < n_rows && sub == 0) delta[i] = acc;
}

// ------------------------------------------------------------------------------------------ 2. dQ (query-major)
template <typename T>
__global__ __launch_bounds__(256, 2) void bwd_dq_kernel(SelAttnBwdParams P, const float *__restrict__ delta, int map_mode) {
    using M = BwdT<T>;
    using x8 = typename M::x8;
    using x4 = typename M::x4;
    constexpr int TILE = 32 * BROWB;  // 4 KiB
    constexpr int SEGB = ((SEG_INTS * 4 + 15) / 16) * 16;
    constexpr int WAVE_LDS = 3 * TILE + SEGB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id();
    const int wave = uniform((int)(threadIdx.x >> 6));
    int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (map_mode != 0) {  // same (b,g)-major / XCD-aware order as the forward kernel
        const int W = (P.S + 3) >> 2;
        int bg, tc;
        if (map_mode == 2) {
            const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
            bg = (idx / W) * 8 + xcd;
            tc = idx % W;
        } else {
            bg = blockIdx.x / W;
            tc = blockIdx.x % W;
        }
        const int t = 4 * tc + wave;
        if (t >= P.S) return;
        row = ((int64_t)(bg / P.G) * P.S + t) * P.G + (bg % P.G);
    }
    if (row >= P.R) return;
    unsigned char *k_row = smem + (size_t)wave * WAVE_LDS;  // K, row image
    unsigned char *k_tr = k_row + TILE;                     // K, transposable image
    unsigned char *v_row = k_tr + TILE;                     // V, row image
    int *seg = (int *)(v_row + TILE);

    const int h = P.h;
    const int g = (int)(row % P.G);
    const int b = (int)(row / ((int64_t)P.G * P.S));
    const unsigned char *Kb = (const unsigned char *)((const T *)P.K + (int64_t)b * P.ksb + (int64_t)g * P.ksg);
    const unsigned char *Vb = (const unsigned char *)((const T *)P.V + (int64_t)b * P.vsb + (int64_t)g * P.vsg);
    const int64_t krowb = P.kss * 2, vrowb = P.vss * 2;
    int nseg;
    const int L = normalise_ranges(P.ranges + row * (int64_t)P.n * 2, P.n, P.S_kv, seg, &nseg);
    const int rho = lane & 15, q = lane >> 4;

    // B operands held for the whole row: Q^T and dO^T fragments (head = column), per-head lse (log2 domain) and delta
    x8 qf[2], dof[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        u32x4 a = {0u, 0u, 0u, 0u}, c = {0u, 0u, 0u, 0u};
        if (rho < h) {
            a = *(const u32x4 *)((const T *)P.Q + (row * h + rho) * (int64_t)BD + 32 * s + 8 * q);
            c = *(const u32x4 *)((const T *)P.dO + (row * h + rho) * (int64_t)BD + 32 * s + 8 * q);
        }
        qf[s] = __builtin_bit_cast(x8, a);
        dof[s] = __builtin_bit_cast(x8, c);
    }
    const float lse2 = (rho < h && L > 0) ? P.lse[row * h + rho] * LOG2E : 0.f;
    const float dlt = (rho < h) ? delta[row * h + rho] : 0.f;
    const float c2 = P.scale * LOG2E;

    const int ld_row = lane >> 3, ld_piece = lane & 7;  // 8 rows x 8 pieces per wave-wide 16-B load
    uint32_t kd_row[4], kd_tr[4], vd_row[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 8 * i + ld_row;
        kd_row[i] = (uint32_t)(ld_row * krowb + ((ld_piece ^ bswz_row(r)) << 4));
        kd_tr[i] = (uint32_t)(ld_row * krowb + (((((ld_piece >> 1) ^ bswz_tr(r)) << 1) | (ld_piece & 1)) << 4));
        vd_row[i] = (uint32_t)(ld_row * vrowb + ((ld_piece ^ bswz_row(r)) << 4));
    }
    uint32_t rd_row[2], rd_tr[4];
#pragma unroll
    for (int s = 0; s < 2; ++s) rd_row[s] = off_row_img(rho, 4 * s + q);
    {
        const int qq = rho >> 2, pp = rho & 3, r = 4 * q + qq;
#pragma unroll
        for (int m = 0; m < 4; ++m) rd_tr[m] = r * BROWB + ((m ^ bswz_tr(r)) << 5) + 8 * pp;
    }
    auto make_rsrc = [&](const unsigned char *base, int64_t bytes) {
        const uint64_t a = (uint64_t)base;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), (short)0,
                                                 __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
    };
    [[maybe_unused]] const auto krs = make_rsrc(Kb, (int64_t)(P.S_kv - 1) * krowb + BROWB);
    [[maybe_unused]] const auto vrs = make_rsrc(Vb, (int64_t)(P.S_kv - 1) * vrowb + BROWB);
    [[maybe_unused]] const int krowb32 = uniform((int)krowb), vrowb32 = uniform((int)vrowb);
    [[maybe_unused]] const int kstep = uniform(8 * (int)krowb), vstep = uniform(8 * (int)vrowb);

    int it_seg = -1, it_start = 0, it_len = 0, it_pos = 0;
    auto next_tile = [&](int &tok0, int &nvalid) -> bool {
        while (true) {
            if (it_pos < it_len) {
                tok0 = it_start + it_pos;
                nvalid = min(32, it_len - it_pos);
                it_pos += 32;
                return true;
            }
            if (++it_seg >= nseg) return false;
            it_start = uniform(seg[2 * it_seg]);
            it_len = uniform(seg[2 * it_seg + 3]) - uniform(seg[2 * it_seg + 1]);
            it_pos = 0;
        }
    };
    auto issue_dma = [&](int tok0, int nvalid) {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef __attribute__((address_space(3))) void lds_void;
        const int ks = uniform(tok0 * krowb32), vs = uniform(tok0 * vrowb32);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // tail tiles: rows past the segment re-read its last row (their dS is masked to zero below)
            const int rowshift = (nvalid == 32) ? 0 : (min(8 * i + ld_row, nvalid - 1) - ld_row);
            const int kso = (nvalid == 32) ? ks + i * kstep : ks, vso = (nvalid == 32) ? vs + i * vstep : vs;
            const uint32_t kadd = (nvalid == 32) ? 0u : (uint32_t)(rowshift * krowb32), vadd = (nvalid == 32) ? 0u : (uint32_t)(rowshift * vrowb32);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (lds_void *)(k_row + i * 1024), 16, kd_row[i] + kadd, kso, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (lds_void *)(k_tr + i * 1024), 16, kd_tr[i] + kadd, kso, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (lds_void *)(v_row + i * 1024), 16, vd_row[i] + vadd, vso, 0, 0);
        }
#else
        (void)tok0;
        (void)nvalid;
#endif
    };

    f32x4 dq[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) dq[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int tok0 = 0, nvalid = 0;
    bool have = next_tile(tok0, nvalid);
    if (have) issue_dma(tok0, nvalid);
    while (have) {
        const int cur_nvalid = nvalid;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        x8 kfr[2][2], vfr[2][2];
        x4 ktr[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                kfr[u][s] = *(const x8 *)(k_row + rd_row[s] + u * 16 * BROWB);
                vfr[u][s] = *(const x8 *)(v_row + rd_row[s] + u * 16 * BROWB);
            }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int m = 0; m < 4; ++m) ktr[u][m] = tr_read<x4>(k_tr + rd_tr[m] + u * 16 * BROWB);
        have = next_tile(tok0, nvalid);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (have) issue_dma(tok0, nvalid);

        f32x4 sacc[2], pacc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            sacc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            pacc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                sacc[u] = M::mma32(kfr[u][s], qf[s], sacc[u]);
                pacc[u] = M::mma32(vfr[u][s], dof[s], pacc[u]);
            }
        }
        x8 dsf;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float p = __builtin_amdgcn_exp2f(fmaf(sacc[u][j], c2, -lse2));
                if (16 * u + 4 * q + j >= cur_nvalid) p = 0.f;
                dsf[4 * u + j] = Elt<T>::from_f(p * (pacc[u][j] - dlt) * P.scale);
            }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            x8 a;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = ktr[0][m][j];
                a[4 + j] = ktr[1][m][j];
            }
            dq[m] = M::mma32(a, dsf, dq[m]);
        }
    }
    if (rho < h) {
        T *dQr = (T *)P.dQ + (row * (int64_t)h + rho) * BD;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            x4 ov;
#pragma unroll
            for (int j = 0; j < 4; ++j) ov[j] = Elt<T>::from_f(dq[m][j]);
            *(x4 *)(dQr + 16 * m + 4 * q) = ov;
        }
    }
}

// ------------------------------------------------------------------------------------------ 2b. dQ, several rows per wave
// Query-tile form of the dQ kernel (see sel_attn_rows_mfma.hip): one wave owns TPW = 16/h consecutive rows of one (b,g) (h = 6: two
// rows = 12 of the 16 MFMA columns) and walks the UNION of their selected 32-key tiles, so the three LDS images of a tile (K rows,
// K transposable, V rows: 12 KB written and read back) are built once for the rows that share it.  Tile schedule and ownership
// come from per-row tile bitmaps (`touch` / `full`, LDS atomics); a slot whose row did not select a key gets p = 0.
__device__ __forceinline__ unsigned bwd_bit_span(int lo, int hi) {  // bits lo..hi inclusive
    const unsigned up = hi >= 31 ? 0xffffffffu : ((1u << (hi + 1)) - 1u);
    return up & ~((1u << lo) - 1u);
}
__device__ __forceinline__ void bwd_lds_or(unsigned *p, unsigned v) {
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    __hip_atomic_fetch_or((lds_u32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

template <typename T>
__global__ __launch_bounds__(256, 2) void bwd_dq_rows_kernel(SelAttnBwdParams P, const float *__restrict__ delta, int map_mode, int tpw, int NW,
                                                          int wave_lds) {
    using M = BwdT<T>;
    using x8 = typename M::x8;
    using x4 = typename M::x4;
    constexpr int TILE = 32 * BROWB;  // 4 KiB
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id();
    const int wave = uniform((int)(threadIdx.x >> 6));
    const int h = P.h, n = P.n;
    const int ngrp = (P.S + tpw - 1) / tpw;
    const int nbg = (int)(P.R / P.S);
    const int W4 = (ngrp + 3) >> 2;
    int bg, tc;
    if (map_mode == 2) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        bg = (idx / W4) * 8 + xcd;
        tc = idx % W4;
    } else {
        bg = blockIdx.x / W4;
        tc = blockIdx.x % W4;
    }
    const int grp = 4 * tc + wave;
    if (grp >= ngrp || bg >= nbg) return;
    const int b = bg / P.G, g = bg - b * P.G;
    const int tw0 = grp * tpw, ntok = min(tpw, P.S - tw0);

    // ONE K image (round 4): the chunk-swizzled image of the transposing reads also serves the b128 row-fragment reads without bank
    // conflicts (see bwd_dkdv_kernel), so K is fetched once per tile, not twice: 8 instead of 12 LDS-DMA instructions per 32 keys
    unsigned char *k_tr = smem + (size_t)wave * wave_lds;   // K, transposable image (row fragments are read from it too)
    unsigned char *v_row = k_tr + TILE;                     // V, row image
    int *rg = (int *)(v_row + TILE);                        // [tpw][n][2]
    unsigned *fullw = (unsigned *)(rg + ((2 * tpw * n + 3) & ~3));
    unsigned *touchw = fullw + tpw * NW;
    unsigned *kmask = touchw + tpw * NW;  // [16]

    for (int i = lane; i < 2 * tpw * NW + 16; i += 64) fullw[i] = 0u;
    for (int r = 0; r < ntok; ++r) {
        const int64_t row = ((int64_t)b * P.S + tw0 + r) * P.G + g;
        if (lane < n) {
            const int32_t *in = P.ranges + (row * n + lane) * 2;
            const int s = min(max(in[0], 0), P.S_kv);
            const int e = min(max(in[1], s), P.S_kv);
            rg[2 * (r * n + lane)] = s;
            rg[2 * (r * n + lane) + 1] = e;
        }
    }
    wave_lds_fence();
    for (int p = lane; p < ntok * n; p += 64) {
        const int r = p / n;
        const int s = rg[2 * p], e = rg[2 * p + 1];
        if (e > s) {
            const int ta = s >> 5, tb = (e - 1) >> 5;
            const int fa = (s + 31) >> 5, fb = (e >> 5) - 1;
            for (int w = ta >> 5; w <= (tb >> 5); ++w) {
                const int base = 32 * w;
                bwd_lds_or(&touchw[r * NW + w], bwd_bit_span(max(ta, base) - base, min(tb, base + 31) - base));
                const int flo = max(fa, base), fhi = min(fb, base + 31);
                if (flo <= fhi) bwd_lds_or(&fullw[r * NW + w], bwd_bit_span(flo - base, fhi - base));
            }
        }
    }
    wave_lds_fence();
    unsigned u0 = 0u, u1 = 0u;
    for (int r = 0; r < ntok; ++r) {
        if (lane < NW) u0 |= touchw[r * NW + lane];
        if (lane + 64 < NW) u1 |= touchw[r * NW + 64 + lane];
    }
    unsigned long long nz0 = __ballot(u0 != 0u), nz1 = __ballot(u1 != 0u);

    const int rho = lane & 15, q = lane >> 4;
    const int tok = rho / h, head = rho - tok * h;
    const bool used = tok < ntok;
    const int64_t orow = used ? ((((int64_t)b * P.S + tw0 + tok) * P.G + g) * h + head) : -1;  // row * h + head

    x8 qf[2], dof[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        u32x4 a = {0u, 0u, 0u, 0u}, c = {0u, 0u, 0u, 0u};
        if (used) {
            a = *(const u32x4 *)((const T *)P.Q + orow * BD + 32 * s + 8 * q);
            c = *(const u32x4 *)((const T *)P.dO + orow * BD + 32 * s + 8 * q);
        }
        qf[s] = __builtin_bit_cast(x8, a);
        dof[s] = __builtin_bit_cast(x8, c);
    }
    float lse2 = 0.f, dlt = 0.f;
    bool live = false;  // rows without any selected key have lse = -inf: every p is 0
    if (used) {
        const float l = P.lse[orow];
        live = l > -INFINITY;
        lse2 = live ? l * LOG2E : 0.f;
        dlt = delta[orow];
    }
    const float c2 = P.scale * LOG2E;

    const unsigned char *Kb = (const unsigned char *)((const T *)P.K + (int64_t)b * P.ksb + (int64_t)g * P.ksg);
    const unsigned char *Vb = (const unsigned char *)((const T *)P.V + (int64_t)b * P.vsb + (int64_t)g * P.vsg);
    const int64_t krowb = P.kss * 2, vrowb = P.vss * 2;
    const int ld_row = lane >> 3, ld_piece = lane & 7;
    uint32_t kd_tr[4], vd_row[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 8 * i + ld_row;
        kd_tr[i] = (uint32_t)(ld_row * krowb + (((((ld_piece >> 1) ^ bswz_tr(r)) << 1) | (ld_piece & 1)) << 4));
        vd_row[i] = (uint32_t)(ld_row * vrowb + ((ld_piece ^ bswz_row(r)) << 4));
    }
    uint32_t rd_row[2], rd_krow[2], rd_tr[4];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        rd_row[s] = off_row_img(rho, 4 * s + q);
        rd_krow[s] = off_tr_img(rho, 4 * s + q);
    }
    {
        const int qq = rho >> 2, pp = rho & 3, r = 4 * q + qq;
#pragma unroll
        for (int m = 0; m < 4; ++m) rd_tr[m] = r * BROWB + ((m ^ bswz_tr(r)) << 5) + 8 * pp;
    }
    auto make_rsrc = [&](const unsigned char *base, int64_t bytes) {
        const uint64_t a = (uint64_t)base;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), (short)0,
                                                 __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
    };
    [[maybe_unused]] const auto krs = make_rsrc(Kb, (int64_t)(P.S_kv - 1) * krowb + BROWB);
    [[maybe_unused]] const auto vrs = make_rsrc(Vb, (int64_t)(P.S_kv - 1) * vrowb + BROWB);
    [[maybe_unused]] const int krowb32 = uniform((int)krowb), vrowb32 = uniform((int)vrowb);
    [[maybe_unused]] const int kstep = uniform(8 * (int)krowb), vstep = uniform(8 * (int)vrowb);
    auto issue_dma = [&](int tok0) {
#ifdef DQ_NODMA  // ablation (timing only, results meaningless): no K / V tile fetches
        return;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
        typedef __attribute__((address_space(3))) void lds_void;
        const int ks = uniform(tok0 * krowb32), vs = uniform(tok0 * vrowb32);
        const bool whole = tok0 + 32 <= P.S_kv;
        const int last = P.S_kv - 1 - tok0;  // rows past the end of K/V re-read the last row (masked: such a tile is never `full`)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rowshift = whole ? 0 : (min(8 * i + ld_row, last) - ld_row);
            const int kso = whole ? ks + i * kstep : ks, vso = whole ? vs + i * vstep : vs;
            const uint32_t kadd = whole ? 0u : (uint32_t)(rowshift * krowb32), vadd = whole ? 0u : (uint32_t)(rowshift * vrowb32);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (lds_void *)(k_tr + i * 1024), 16, kd_tr[i] + kadd, kso, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (lds_void *)(v_row + i * 1024), 16, vd_row[i] + vadd, vso, 0, 0);
        }
#else
        (void)tok0;
#endif
    };

    int iw = 0;
    unsigned ibits = 0u;
    auto next_tile = [&]() -> int {
        if (ibits == 0u) {
            if (nz0) {
                iw = __builtin_ctzll(nz0);
                nz0 &= nz0 - 1ull;
                ibits = (unsigned)__builtin_amdgcn_readlane((int)u0, iw);
            } else if (nz1) {
                const int w = __builtin_ctzll(nz1);
                nz1 &= nz1 - 1ull;
                ibits = (unsigned)__builtin_amdgcn_readlane((int)u1, w);
                iw = 64 + w;
            } else {
                return -1;
            }
        }
        const int bit = __builtin_ctz(ibits);
        ibits &= ibits - 1u;
        return 32 * iw + bit;
    };

    f32x4 dq[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) dq[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int cur = next_tile();
    if (cur >= 0) issue_dma(32 * cur);
    int cw = -1;
    unsigned fw = 0u, tw = 0u;
    const unsigned rowbit = live ? (1u << tok) : 0u;
    while (cur >= 0) {
        const int nxt = next_tile();
        const int tok0 = 32 * cur;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        x8 kfr[2][2], vfr[2][2];
        x4 ktr[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                kfr[u][s] = *(const x8 *)(k_tr + rd_krow[s] + u * 16 * BROWB);
                vfr[u][s] = *(const x8 *)(v_row + rd_row[s] + u * 16 * BROWB);
            }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int m = 0; m < 4; ++m) ktr[u][m] = tr_read<x4>(k_tr + rd_tr[m] + u * 16 * BROWB);
        if ((cur >> 5) != cw) {  // lane r < ntok caches row r's bitmap words of the current 32-tile group
            cw = cur >> 5;
            if (lane < ntok) {
                fw = fullw[lane * NW + cw];
                tw = touchw[lane * NW + cw];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (nxt >= 0) issue_dma(32 * nxt);
        const unsigned fullm = (unsigned)__ballot((fw >> (cur & 31)) & 1u);
        const unsigned touchm = (unsigned)__ballot((tw >> (cur & 31)) & 1u);
        const unsigned partm = touchm & ~fullm;
        if (partm) {  // partially covered by some row: that row's 32-key mask from its ranges
            if (lane < 16) kmask[lane] = 0u;
            wave_lds_fence();
            for (int p = lane; p < ntok * n; p += 64) {
                const int r = p / n;
                const int lo = max(rg[2 * p], tok0) - tok0, hi = min(rg[2 * p + 1], tok0 + 32) - tok0;
                if (((partm >> r) & 1u) && hi > lo) bwd_lds_or(&kmask[r], bwd_bit_span(lo, hi - 1));
            }
            wave_lds_fence();
        }
        const bool on = (fullm & rowbit) != 0u;
        unsigned km = on ? 0xffffffffu : 0u;
        if (partm) {
            if (partm & rowbit) km = kmask[tok];
            km >>= 4 * q;  // bit 16u + j = key 16u + 4q + j of the tile
        }

#ifdef DQ_NOCOMPUTE  // ablation (timing only): tile walk and fetches without the products
        cur = nxt;
        continue;
#endif
        f32x4 sacc[2], pacc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            sacc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            pacc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                sacc[u] = M::mma32(kfr[u][s], qf[s], sacc[u]);
                pacc[u] = M::mma32(vfr[u][s], dof[s], pacc[u]);
            }
        }
        x8 dsf;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float p = __builtin_amdgcn_exp2f(fmaf(sacc[u][j], c2, -lse2));
                if (partm == 0u) p = on ? p : 0.f;  // every slot all-on or all-off: one predicate per lane (wave-uniform branch)
                else if (!((km >> (16 * u + j)) & 1u)) p = 0.f;
                dsf[4 * u + j] = Elt<T>::from_f(p * (pacc[u][j] - dlt) * P.scale);
            }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            x8 a;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = ktr[0][m][j];
                a[4 + j] = ktr[1][m][j];
            }
            dq[m] = M::mma32(a, dsf, dq[m]);
        }
        cur = nxt;
    }
    if (used) {
        T *dQr = (T *)P.dQ + orow * BD;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            x4 ov;
#pragma unroll
            for (int j = 0; j < 4; ++j) ov[j] = Elt<T>::from_f(dq[m][j]);
            *(x4 *)(dQr + 16 * m + 4 * q) = ov;
        }
    }
}

// ------------------------------------------------------------------------------------------ 3. dK / dV (key-block-major)
constexpr int KB_NCT = 8;
#ifdef NSA_DBG_WGTIME
__device__ unsigned long long g_dbg[4 * 65536];
#endif  // column tiles (of 16 (query,head) slots) staged per round

template <typename T>
__global__ __launch_bounds__(256, 2) void bwd_dkdv_kernel(SelAttnBwdParams P, const float *__restrict__ delta, float *__restrict__ part,
                                                        const unsigned long long *__restrict__ hitmap, const unsigned long long *__restrict__ fullmap,
                                                        int *__restrict__ flags, int rows_per_split, int nkb, int nbg, int nsplit) {
    using M = BwdT<T>;
    using x8 = typename M::x8;
    using x4 = typename M::x4;
    // Workgroup -> (key block j, bg, row split z).  Workgroups go round-robin over the 8 XCDs by linear id; all key blocks
    // of one (bg, z) pair are put on ONE XCD (its 512 Q/dO rows stay in that L2), and the pairs are dealt evenly over the
    // XCDs (every pair carries about the same number of hits, while key block 0 alone is hit by every row: a j-major
    // order leaves one XCD with several times the work of the others).
    int j, bg, zsp;
    {
        const int w = blockIdx.x, xcd = w & 7, slot = w >> 3;
        const int pair = (slot / nkb) * 8 + xcd;
        j = slot % nkb;
        if (pair >= nbg * nsplit) return;
        zsp = pair / nbg;
        bg = pair - zsp * nbg;
    }
    constexpr int SLOTS = 16 * KB_NCT;        // (query,head) slots per round
    constexpr int IMG = SLOTS * BROWB;        // bytes of one staged image
    constexpr int NLD = SLOTS * 8 / 256;      // 16-byte pieces per thread and image
    // ONE image per staged operand (round 4): the chunk-swizzled image the transposing reads need serves the b128 row-fragment reads too, free
    // of bank conflicts (every 16-lane service group of ds_read_b128 touches 64 distinct banks: checked exhaustively) -- rounds 1-3 staged
    // Q and dO twice (row image + transposable image: 64 KiB of LDS writes per round instead of 32)
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * IMG];  // the K/V block images alias the staging images (only read before the loop)
    __shared__ __attribute__((aligned(16))) float s_lse2[SLOTS], s_delta[SLOTS];
    __shared__ __attribute__((aligned(16))) unsigned long long s_mask[SLOTS];
    __shared__ int s_tilefull[KB_NCT];  // every used slot of the tile covers all 64 keys of the block: the mask test is skipped
    __shared__ int s_t[512];
    __shared__ unsigned long long s_m[512];
    unsigned char *k_img = lds, *v_img = lds + 64 * BROWB;
    unsigned char *q_img = lds, *do_img = lds + IMG;

#ifdef NSA_DBG_WGTIME
    const unsigned long long dbg_t0 = wall_clock64();
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
    const int rho = lane & 15, q = lane >> 4;
    const int b = bg / P.G, g = bg % P.G;
    const int h = P.h;
    const int key0 = 64 * j;
    const T *Kb = (const T *)P.K + (int64_t)b * P.ksb + (int64_t)g * P.ksg;
    const T *Vb = (const T *)P.V + (int64_t)b * P.vsb + (int64_t)g * P.vsg;
    const float c2 = P.scale * LOG2E;

    // ---- K_j, V_j -> LDS (row images); rows past S_kv re-read the last row (never covered by a mask)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = tid + 256 * i, r = p >> 3, pc = p & 7;
        const int64_t kr = min(key0 + r, P.S_kv - 1);
        *(u32x4 *)(k_img + off_row_img(r, pc)) = *(const u32x4 *)(Kb + kr * P.kss + pc * 8);
        *(u32x4 *)(v_img + off_row_img(r, pc)) = *(const u32x4 *)(Vb + kr * P.vss + pc * 8);
    }
    __syncthreads();
    // B operands of S = Q.K^T and dP = dO.V^T for this wave's 16 keys: loop invariant
    x8 kB[2], vB[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        kB[s] = *(const x8 *)(k_img + off_row_img(16 * wave + rho, 4 * s + q));
        vB[s] = *(const x8 *)(v_img + off_row_img(16 * wave + rho, 4 * s + q));
    }
    __syncthreads();
    f32x4 dK[4], dV[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        dK[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        dV[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    uint32_t rd_row[2], rd_tr[4];
#pragma unroll
    for (int s = 0; s < 2; ++s) rd_row[s] = off_tr_img(rho, 4 * s + q);
    {
        const int qq = rho >> 2, pp = rho & 3, r = 4 * q + qq;
#pragma unroll
        for (int n = 0; n < 4; ++n) rd_tr[n] = r * BROWB + ((n ^ bswz_tr(r)) << 5) + 8 * pp;
    }

    // query rows are split over gridDim.z workgroups per key block (block 0 and the local blocks are selected by every
    // row: without the split their workgroups are a long serial tail); the partial sums are added by a second kernel
    // in fixed order, so the result stays bitwise reproducible
    const int row_begin = zsp * rows_per_split, row_end = min(P.S, row_begin + rows_per_split);
    const int wpb = (P.S + 63) >> 6;  // hit-map words per (bg, key block)
    const unsigned long long *hm = hitmap + ((int64_t)bg * nkb + j) * wpb;
    const unsigned long long *hf = fullmap + ((int64_t)bg * nkb + j) * wpb;
    // Hits are collected over 256-row chunks until at least 256 are pending (far-away key blocks see a handful of hits per
    // chunk at long context; staging rounds want to be full)
    int total_hits = 0, pend = 0;
    for (int base = row_begin; base < row_end; base += 256) {
        // ---- which of these 256 query rows selected keys of this block: 4 words of the hit map (lane = row)
        unsigned long long hw[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int wi = (base >> 6) + w;
            hw[w] = wi < wpb ? hm[wi] : 0ull;
        }
        const int nnew = __popcll(hw[0]) + __popcll(hw[1]) + __popcll(hw[2]) + __popcll(hw[3]);  // uniform over the workgroup
        total_hits += nnew;
        int off = pend;
        pend += nnew;
        if (pend == 0 || (pend < 256 && base + 256 < row_end && nnew == 0)) continue;
        unsigned long long hitb = 0ull;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) off += __popcll(hw[w]);
            if (w == wave) hitb = hw[w];
        }
        const int wi_own = (base >> 6) + wave;
        const unsigned long long fullb = wi_own < wpb ? hf[wi_own] : 0ull;  // rows whose hit is one range over the whole block: mask known
        if ((hitb >> lane) & 1ull) {
            const int t = base + tid;
            unsigned long long mask = ~0ull;
            if (!((fullb >> lane) & 1ull)) {  // partial coverage (or several ranges inside the block): the key mask from the row's ranges
                mask = 0ull;
                const int32_t *rg = P.ranges + (((int64_t)b * P.S + t) * P.G + g) * (int64_t)P.n * 2;
                for (int i = 0; i < P.n; ++i) {
                    int s0 = min(max(rg[2 * i], 0), P.S_kv), e0 = min(max(rg[2 * i + 1], 0), P.S_kv);
                    const int lo = max(s0, key0) - key0, hi = min(e0, key0 + 64) - key0;
                    if (hi > lo) mask |= ((hi - lo == 64) ? ~0ull : ((1ull << (hi - lo)) - 1ull)) << lo;
                }
            }
            const int pos = off + __popcll(hitb & ((1ull << lane) - 1ull));
            s_t[pos] = t;
            s_m[pos] = mask;
        }
        if (pend < 256 && base + 256 < row_end) continue;
        const int nhit = pend;
        pend = 0;
        __syncthreads();

        // ---- process the hit rows, KB_NCT tiles of 16 (row, head) slots per staging round.  The global loads of
        // round r+1 (Q, dO rows, lse, delta) are issued into registers before the MFMAs of round r.
        // slots are laid end to end: slot = h * (row of the round) + head, so a round holds SLOTS / h rows and every 16-slot tile is
        // full (two rows x 6 heads per tile left a quarter of the MFMA rows idle; a slot is one independent (row, head) pair, its tile is
        // only the contraction group of the dV / dK products)
        const int rows_per_round = SLOTS / h;
        u32x4 qa[NLD], da[NLD];
        float l2n = 0.f, dln = 0.f;
        unsigned long long mkn = 0ull;
        auto fetch_round = [&](int r0) {
#ifdef DKDV_NOFETCH  // ablation (timing only): no global loads of Q / dO rows
            return;
#endif
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int p = tid + 256 * i, slot = p >> 3, pc = p & 7;
                const int qi = slot / h, hh = slot - qi * h;
                const int li = r0 + qi;
                qa[i] = (u32x4){0u, 0u, 0u, 0u};
                da[i] = (u32x4){0u, 0u, 0u, 0u};
                if (qi < rows_per_round && li < nhit) {
                    const int64_t rrow = ((int64_t)b * P.S + s_t[li]) * P.G + g;
                    qa[i] = *(const u32x4 *)((const T *)P.Q + (rrow * h + hh) * (int64_t)BD + pc * 8);
                    da[i] = *(const u32x4 *)((const T *)P.dO + (rrow * h + hh) * (int64_t)BD + pc * 8);
                }
            }
            l2n = 0.f;
            dln = 0.f;
            mkn = 0ull;
            if (tid < SLOTS) {
                const int qi = tid / h, hh = tid - qi * h;
                const int li = r0 + qi;
                if (qi < rows_per_round && li < nhit) {
                    const int64_t rrow = ((int64_t)b * P.S + s_t[li]) * P.G + g;
                    l2n = P.lse[rrow * h + hh] * LOG2E;
                    dln = delta[rrow * h + hh];
                    mkn = s_m[li];
                }
            }
        };
        if (nhit > 0) fetch_round(0);
        for (int r0 = 0; r0 < nhit; r0 += rows_per_round) {
            // stage Q and dO rows of every slot twice (row image + transposable image), plus lse / delta / mask per slot
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int p = tid + 256 * i, slot = p >> 3, pc = p & 7;
                *(u32x4 *)(q_img + off_tr_img(slot, pc)) = qa[i];
                *(u32x4 *)(do_img + off_tr_img(slot, pc)) = da[i];
            }
            if (tid < SLOTS) {
                s_lse2[tid] = l2n;
                s_delta[tid] = dln;
                s_mask[tid] = mkn;
                // slots past the last hit row carry zero Q / dO rows, lse2 = delta = 0: p = 1 there multiplies zeros -- they count as full
                const int qi = tid / h;
                const bool fullslot = mkn == ~0ull || !(qi < rows_per_round && r0 + qi < nhit);
                const unsigned long long fb = __ballot(fullslot);  // waves 0 and 1: 64 slots = 4 tiles each
                if (lane < 4) s_tilefull[4 * wave + lane] = ((fb >> (16 * lane)) & 0xffffull) == 0xffffull;
            }
            if (r0 + rows_per_round < nhit) fetch_round(r0 + rows_per_round);
            __syncthreads();
            // column tiles go in PAIRS: the dV / dK products contract over the 32 (row, head) slots of two tiles with ONE 16x16x32 MFMA per
            // 16 output columns (round 4: the 16x16x16 form they used costs the same issue cycles for half the work --
            // tools/ubench/issue_rates.hip: 19-20 reported cycles either way; the MFMA's k index is only a label, so k = 8q + r is slot
            // 4q + r of the first tile and k = 8q + 4 + r the same slot of the second, on both operands).  A second tile past the last
            // hit row is all zero rows (p = 1 times zero dO, dS = 0): it adds nothing.
#ifdef DKDV_NOCOMPUTE  // ablation (timing only, results meaningless): staging and bookkeeping without the tile loop
            const int ntile = 0;
#else
            const int ntile = min(KB_NCT, (min(nhit - r0, rows_per_round) * h + 15) >> 4);
#endif
            for (int ct = 0; ct < ntile; ct += 2) {
                x8 pa8, dsa8;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int sbase = 16 * (ct + half);
                    f32x4 S = {0.f, 0.f, 0.f, 0.f}, dP = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        S = M::mma32(*(const x8 *)(q_img + rd_row[s] + sbase * BROWB), kB[s], S);
                        dP = M::mma32(*(const x8 *)(do_img + rd_row[s] + sbase * BROWB), vB[s], dP);
                    }
                    // accumulator rows = slots sbase + 4q + r, column = key 16 wave + rho
                    const f32x4 l2 = *(const f32x4 *)(s_lse2 + sbase + 4 * q);
                    const f32x4 dl = *(const f32x4 *)(s_delta + sbase + 4 * q);
                    // dS carries no `scale` here: dK is multiplied once when it is written (MFMA and VALU issue add up: every VALU instruction
                    // of this loop is paid in full).  Most tiles are fully covered (whole selection blocks): no per-key mask test for them.
                    if (uniform(s_tilefull[ct + half])) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float p = __builtin_amdgcn_exp2f(fmaf(S[r], c2, -l2[r]));
                            pa8[4 * half + r] = Elt<T>::from_f(p);
                            dsa8[4 * half + r] = Elt<T>::from_f(p * (dP[r] - dl[r]));
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const bool cov = (s_mask[sbase + 4 * q + r] >> (16 * wave + rho)) & 1ull;
                            const float p = cov ? __builtin_amdgcn_exp2f(fmaf(S[r], c2, -l2[r])) : 0.f;
                            pa8[4 * half + r] = Elt<T>::from_f(p);
                            dsa8[4 * half + r] = Elt<T>::from_f(p * (dP[r] - dl[r]));
                        }
                    }
                }
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const x4 d0 = tr_read<x4>(do_img + rd_tr[n] + 16 * ct * BROWB), d1 = tr_read<x4>(do_img + rd_tr[n] + 16 * (ct + 1) * BROWB);
                    const x4 q0 = tr_read<x4>(q_img + rd_tr[n] + 16 * ct * BROWB), q1 = tr_read<x4>(q_img + rd_tr[n] + 16 * (ct + 1) * BROWB);
                    x8 db, qb;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        db[jj] = d0[jj];
                        db[4 + jj] = d1[jj];
                        qb[jj] = q0[jj];
                        qb[4 + jj] = q1[jj];
                    }
                    dV[n] = M::mma32(pa8, db, dV[n]);
                    dK[n] = M::mma32(dsa8, qb, dK[n]);
                }
            }
            __syncthreads();
        }
    }
    // ---- write this wave's 16 keys: accumulator rows = key 16 wave + 4q + r, column = d 16 n + rho.  A split without
    // hits writes nothing and says so in flags[] (the reduce kernel skips it).
#ifdef NSA_DBG_WGTIME
    if (tid == 0) {
        const int64_t w = ((int64_t)zsp * nbg + bg) * nkb + j;
        if (w < 65536) {
            unsigned hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(hwid));
            g_dbg[4 * w] = dbg_t0;
            g_dbg[4 * w + 1] = wall_clock64();
            g_dbg[4 * w + 2] = total_hits;
            g_dbg[4 * w + 3] = hwid;
        }
    }
#endif
    if (nsplit > 1) {
        if (tid == 0) flags[((int64_t)zsp * nbg + bg) * nkb + j] = total_hits > 0;
        if (total_hits == 0) return;
    }
    const int64_t slab = (int64_t)nbg * P.S_kv * BD;  // floats of one [B*G,S_kv,D] tensor
    float *dKb = (nsplit > 1 ? part + (int64_t)zsp * 2 * slab : P.dK) + ((int64_t)bg * P.S_kv) * BD;
    float *dVb = (nsplit > 1 ? part + (int64_t)zsp * 2 * slab + slab : P.dV) + ((int64_t)bg * P.S_kv) * BD;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int key = key0 + 16 * wave + 4 * q + r;
        if (key < P.S_kv) {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                dKb[(int64_t)key * BD + 16 * n + rho] = dK[n][r] * P.scale;
                dVb[(int64_t)key * BD + 16 * n + rho] = dV[n][r];
            }
        }
    }
}

// sum the row-split partials [ns][2][slab] into dK / dV in ascending split order (splits without hits are skipped)
__global__ __launch_bounds__(256) void bwd_reduce_kernel(const float *__restrict__ part, const int *__restrict__ flags,
                                                          float *__restrict__ dK, float *__restrict__ dV, int64_t slab, int ns, int S_kv,
                                                          int nkb) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= 2 * slab) return;
    const int64_t e = i < slab ? i : i - slab;
    const int64_t krow = e / BD, bg = krow / S_kv;
    const int jb = (int)(krow - bg * S_kv) >> 6;
    const int64_t nfl = slab / BD / S_kv * nkb;  // flags per split
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < ns; ++z)
        if (flags[z * nfl + bg * nkb + jb]) acc += *(const f32x4 *)(part + (int64_t)z * 2 * slab + i);
    float *dst = i < slab ? dK + i : dV + (i - slab);
    *(f32x4 *)dst = acc;
}

// Inverted index of the selection: hitmap[bg][key block j][t / 64] bit (t % 64) = row t of (b,g) has a range reaching into
// keys [64 j, 64 j + 64).  One wave = the 64 rows of one word: lanes that reach the same block are gathered with a ballot, so
// a word gets one OR per (range slot, block) instead of one per row; OR-ing makes the result independent of the order.
// fullmap (same shape, round 4): bit = ONE range of the row covers all 64 keys of the block.  The dK/dV kernel then knows the row's key mask
// of the block (all ones) without reading the row's ranges -- every hit of a batched selection (whole blocks only), all but the clamped last
// block of a sequential one, the inner blocks of a band.
__global__ __launch_bounds__(256) void bwd_hitmap_kernel(const int32_t *__restrict__ ranges, unsigned long long *__restrict__ hitmap,
                                                          unsigned long long *__restrict__ fullmap, int64_t nwords, int S, int G, int n, int S_kv,
                                                          int nkb) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= nwords) return;
    const int wpb = (S + 63) >> 6;
    const int64_t bg = wid / wpb;
    const int tw = (int)(wid - bg * wpb), t = 64 * tw + lane;
    const int64_t b = bg / G, g = bg - b * G;
    const int32_t *rg = ranges + ((b * S + min(t, S - 1)) * G + g) * (int64_t)n * 2;
    unsigned long long *w = hitmap + bg * nkb * wpb + tw;
    unsigned long long *wf = fullmap + bg * nkb * wpb + tw;
    for (int i = 0; i < n; ++i) {
        const int s0 = min(max(rg[2 * i], 0), S_kv), e0 = min(max(rg[2 * i + 1], 0), S_kv);
        int j = s0 >> 6;
        const int jend = (t < S && e0 > s0) ? (e0 - 1) >> 6 : -1;
        for (;;) {
            const unsigned long long todo = __ballot(j <= jend);
            if (todo == 0ull) break;
            const int src = __ffsll((long long)todo) - 1;
            const int j0 = __shfl(j, src);
            const bool mine = j <= jend && j == j0;
            const unsigned long long m = __ballot(mine);
            const unsigned long long mf = __ballot(mine && s0 <= 64 * j0 && e0 >= 64 * j0 + 64);
            if (lane == src) {
                atomicOr(w + (int64_t)j0 * wpb, m);
                if (mf) atomicOr(wf + (int64_t)j0 * wpb, mf);
            }
            if (mine) ++j;
        }
    }
}

// ------------------------------------------------------------------------------------------ host
static int dkdv_splits(int S) {
    int ns = (S + 511) / 512;
    return ns < 1 ? 1 : (ns > 16 ? 16 : ns);
}

bool sel_attn_bwd_mfma_supported(int dtype, int h, int Dk, int Dv) {
    return (dtype == NSA_DT_BF16 || dtype == NSA_DT_F16) && Dk == 64 && Dv == 64 && h >= 1 && h <= 16;
}

// workspace layout: delta [R*h] f32 | hit map [nbg][nkb][ceil(S/64)] u64 | full map (same shape) | flags [ns][nbg][nkb] i32 | ns partial [dK|dV] slabs
struct BwdWs {
    size_t hitmap, fullmap, flags, part, total, hitmap_bytes;
    int ns, nkb;
};
static BwdWs bwd_ws_layout(int64_t R, int h, int S, int64_t nbg, int S_kv) {
    BwdWs w;
    w.ns = dkdv_splits(S);
    w.nkb = (S_kv + 63) / 64;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    w.hitmap = up(sizeof(float) * (size_t)R * h);
    w.hitmap_bytes = sizeof(unsigned long long) * (size_t)nbg * w.nkb * ((S + 63) / 64);
    w.fullmap = w.hitmap + up(w.hitmap_bytes);
    w.flags = w.fullmap + up(w.hitmap_bytes);
    w.part = w.flags + up(sizeof(int) * (size_t)w.ns * nbg * w.nkb);
    w.total = w.part + (w.ns > 1 ? sizeof(float) * (size_t)w.ns * 2 * (size_t)nbg * S_kv * BD : 0);
    return w;
}
size_t sel_attn_bwd_mfma_workspace(int64_t R, int h, int S, int64_t nbg, int S_kv) { return bwd_ws_layout(R, h, S, nbg, S_kv).total; }

template <typename T>
static int launch_bwd_t(const SelAttnBwdParams &P, float *delta, hipStream_t st) {
    const int64_t nrh = P.R * P.h;
    if (!P.skip_delta_dq) {
        hipLaunchKernelGGL(bwd_delta_kernel<T>, dim3((unsigned)((nrh * 8 + 255) / 256)), dim3(256), 0, st, (const T *)P.O, (const T *)P.dO,
                           delta, nrh, P.Dv);
        NSA_LAUNCH_CHECK("bwd_delta");
    }
    int map_mode = 0;
    unsigned grid = (unsigned)((P.R + 3) / 4);
    if (P.S >= 16) {
        const int64_t nbg = P.R / P.S, W = (P.S + 3) / 4;
        if (nbg * W < ((int64_t)1 << 31)) {
            map_mode = (nbg % 8 == 0) ? 2 : 1;
            grid = (unsigned)(nbg * W);
        }
    }
    constexpr size_t lds = 4 * (3 * 32 * BROWB + ((SEG_INTS * 4 + 15) / 16) * 16);
    if (!P.skip_delta_dq) {
        // rows of one wave share the K/V tile images when 16/h >= 2 rows fit the MFMA columns (NSA_HIP_SEL_ROWS=0: one row per wave)
        int tpw = 16 / P.h;
        if (tuning(TUNE_SEL_ROWS) == 0) tpw = 1;
        const int nw = ((P.S_kv + 31) / 32 + 31) / 32;
        if (tpw >= 2 && P.S >= 2 * tpw && P.n >= 1 && P.n <= 64 && nw <= 128) {
            const int rg_ints = (2 * tpw * P.n + 3) & ~3, bm_ints = (2 * tpw * nw + 16 + 3) & ~3;
            const int wave_lds = 2 * 32 * BROWB + 4 * (rg_ints + bm_ints);
            const int64_t nbg2 = P.R / P.S, ngrp = (P.S + tpw - 1) / tpw, W4 = (ngrp + 3) / 4;
            NSA_CHECK_ARG(nbg2 * W4 < ((int64_t)1 << 31) && 4 * (size_t)wave_lds <= 160 * 1024, "bwd_dq_rows: launch too large");
            if (4 * (size_t)wave_lds > 64 * 1024)
                NSA_HIP_TRY(hipFuncSetAttribute((const void *)bwd_dq_rows_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * wave_lds));
            hipLaunchKernelGGL(bwd_dq_rows_kernel<T>, dim3((unsigned)(nbg2 * W4)), dim3(256), 4 * (size_t)wave_lds, st, P, (const float *)delta,
                               (nbg2 % 8 == 0) ? 2 : 1, tpw, nw, wave_lds);
            NSA_LAUNCH_CHECK("bwd_dq_rows");
        } else {
            hipLaunchKernelGGL(bwd_dq_kernel<T>, dim3(grid), dim3(256), lds, st, P, (const float *)delta, map_mode);
            NSA_LAUNCH_CHECK("bwd_dq");
        }
    }
    const int64_t nbg = (int64_t)(P.R / P.S);
    NSA_CHECK_ARG(nbg <= 65535, "bwd: B*G too large for one launch");
    const BwdWs W = bwd_ws_layout(P.R, P.h, P.S, nbg, P.S_kv);
    const int ns = W.ns;
    const int rows_per_split = ((P.S + ns - 1) / ns + 255) / 256 * 256;
    unsigned char *ws = (unsigned char *)delta;
    unsigned long long *hitmap = (unsigned long long *)(ws + W.hitmap);
    unsigned long long *fullmap = (unsigned long long *)(ws + W.fullmap);
    int *flags = (int *)(ws + W.flags);
    float *part = (float *)(ws + W.part);
    NSA_HIP_TRY(hipMemsetAsync(hitmap, 0, W.fullmap - W.hitmap + W.hitmap_bytes, st));  // both maps (adjacent)
    const int64_t nwords = nbg * ((P.S + 63) / 64);
    hipLaunchKernelGGL(bwd_hitmap_kernel, dim3((unsigned)((nwords + 3) / 4)), dim3(256), 0, st, P.ranges, hitmap, fullmap, nwords, P.S, P.G,
                       P.n, P.S_kv, W.nkb);
    NSA_LAUNCH_CHECK("bwd_hitmap");
    const int64_t ngrid = ((nbg * ns + 7) / 8) * 8 * W.nkb;
    NSA_CHECK_ARG(ngrid < ((int64_t)1 << 31), "bwd: too many key-block workgroups for one launch");
    hipLaunchKernelGGL(bwd_dkdv_kernel<T>, dim3((unsigned)ngrid), dim3(256), 0, st, P, (const float *)delta, part,
                       (const unsigned long long *)hitmap, (const unsigned long long *)fullmap, flags, rows_per_split, W.nkb, (int)nbg, ns);
    NSA_LAUNCH_CHECK("bwd_dkdv");
    if (ns > 1) {
        const int64_t slab = nbg * P.S_kv * BD;
        hipLaunchKernelGGL(bwd_reduce_kernel, dim3((unsigned)((2 * slab / 4 + 255) / 256)), dim3(256), 0, st, (const float *)part,
                           (const int *)flags, P.dK, P.dV, slab, ns, P.S_kv, W.nkb);
        NSA_LAUNCH_CHECK("bwd_reduce");
    }
    return NSA_OK;
}

#ifdef NSA_DBG_WGTIME
extern "C" __attribute__((visibility("default"))) int nsa_dbg_read(void *host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_dbg), sizeof(unsigned long long) * 4 * 65536);
}
#endif
int launch_sel_attn_bwd_mfma(const SelAttnBwdParams &P, int dtype, float *delta_ws, hipStream_t st) {
    NSA_CHECK_ARG(sel_attn_bwd_mfma_supported(dtype, P.h, P.Dk, P.Dv), "bwd MFMA: unsupported dtype/h/D");
    NSA_CHECK_ARG(P.kss % 8 == 0 && P.vss % 8 == 0 && P.ksb % 8 == 0 && P.vsb % 8 == 0 && P.ksg % 8 == 0 && P.vsg % 8 == 0,
                  "bwd MFMA: K/V strides must be multiples of 8 elements");
    NSA_CHECK_ARG((int64_t)P.S_kv * P.kss * 2 < ((int64_t)1 << 31) && (int64_t)P.S_kv * P.vss * 2 < ((int64_t)1 << 31),
                  "bwd MFMA: one (b,g) K/V slab must be smaller than 2 GiB");
    if (dtype == NSA_DT_BF16) return launch_bwd_t<__bf16>(P, delta_ws, st);
    return launch_bwd_t<_Float16>(P, delta_ws, st);
}

int launch_bwd_delta(const void *O, const void *dO, float *delta, int64_t n_rows, int Dv, int dtype, hipStream_t st) {
    NSA_CHECK_ARG(Dv == 64 && (dtype == NSA_DT_BF16 || dtype == NSA_DT_F16), "bwd_delta: bf16/f16 with Dv = 64 only");
    const dim3 grid((unsigned)((n_rows * 8 + 255) / 256));
    if (dtype == NSA_DT_BF16) hipLaunchKernelGGL(bwd_delta_kernel<__bf16>, grid, dim3(256), 0, st, (const __bf16 *)O, (const __bf16 *)dO, delta, n_rows, Dv);
    else hipLaunchKernelGGL(bwd_delta_kernel<_Float16>, grid, dim3(256), 0, st, (const _Float16 *)O, (const _Float16 *)dO, delta, n_rows, Dv);
    NSA_LAUNCH_CHECK("bwd_delta");
    return NSA_OK;
}

// the band as one [lo, hi) range per (b,t,g) row: input of the selection backward kernels
__global__ __launch_bounds__(256) void band_ranges_kernel(int32_t *__restrict__ ranges, int64_t R, int S, int G, int S_kv, int t0, int a, int dd,
                                                          int c, int w) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= R) return;
    const int t = (int)((row / G) % S);
    const int hi = band_hi(t0, a, dd, c, S_kv, t);
    ranges[2 * row] = max(0, hi - w);
    ranges[2 * row + 1] = hi;
}

int launch_band_ranges(int32_t *ranges, int B, int S, int G, int S_kv, int t0, int a, int dd, int c, int w, hipStream_t st) {
    const int64_t R = (int64_t)B * S * G;
    if (R == 0) return NSA_OK;
    hipLaunchKernelGGL(band_ranges_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, ranges, R, S, G, S_kv, t0, a, dd, c, w);
    NSA_LAUNCH_CHECK("band_ranges");
    return NSA_OK;
}

}  // namespace nsa
