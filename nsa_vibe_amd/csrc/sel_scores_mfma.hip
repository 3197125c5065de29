// Fused selection scorer on MFMA (gfx950): Q, K_cmp -> p_grp without ever materialising p_cmp.
//
// Reference chain: compute_pcmp_all (selection_scorer.py:42-61) -> map_pcmp_to_pslc_batched (:89-116)
// -> .sum(dim=3) (nsa_attention.py:1091).  This kernel covers the default block geometry
// l = 2d, l' = 4d (m7c: 32/16/64), for which Eq.9 is the closed-form 5-tap stencil
//     p_slc[j] = 1/2 p[4j-1] + p[4j] + p[4j+1] + p[4j+2] + 1/2 p[4j+3]      (taps outside [0,S_cmp) dropped)
// (the generic CSC path in sel_scores.hip covers every other geometry).
//
// Mapping.  One workgroup = 4 waves = QW consecutive queries of one (b,g).  The (query, head) pairs
// are the 16 columns of 16x16x32 MFMA tiles (floor(16/h) queries per tile, NT tiles per wave); the
// compressed keys are the MFMA rows:
//     S^T[cmp, (query,head)] = K_cmp tile [64 x D] . Q^T [D x 16]
// so for a fixed column the 4 accumulator registers of lane group q are the 4 consecutive compressed
// columns 4j..4j+3 of ONE selection block j: the stencil is in-lane except the 1/2 p[4j-1] tap
// (one lane rotation by 16), and the Eq.10 head sum is a 16-lane segmented add.
// The K_cmp tile is staged once per workgroup in LDS (XOR-swizzled 16-B pieces, register-staged
// double buffer) and shared by all 4*NT column tiles.
// The softmax is over ALL S_cmp columns (the reference normalises over future compressed tokens
// too), so two sweeps are needed: sweep 1 = row max / row sum (per-lane online, merged once at the
// end), sweep 2 = recompute logits, normalise, stencil, head sum, store.  With causal_skip the
// second sweep stops at the last selection block any query of the workgroup may select
// ((j+1) l' <= t+1); blocks beyond are never read by the selector (masked to -inf there).
#include "nsa_common.hpp"

namespace nsa {

struct ScoresMfmaParams {
    const void *Q;   // [B,S,G,h,D]
    const void *Kc;  // [B,G,S_cmp,D] strided
    float *p_grp;    // [B,S,G,S_sel]
    int B, S, G, h, S_cmp, S_sel;
    int64_t csb, csg, css;
    float scale;
    int causal_skip;
    int d_stride;  // the compression stride d (tokens); l' = 4d
};

template <typename T>
struct MfmaS;
template <>
struct MfmaS<__bf16> {
    using x8 = bf16x8;
    __device__ static f32x4 mma(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <>
struct MfmaS<_Float16> {
    using x8 = f16x8;
    __device__ static f32x4 mma(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

// HC: heads per group as a compile-time constant (0 = read P.h at run time).  The Eq.10 head sum is h - 1 DPP adds per
// (16-row sub-tile, column tile); with a run-time h every one of the 15 possible adds is its own two-instruction basic block
// behind a scalar compare and branch (16 blocks x 16 sub-tiles per K_cmp tile: the second sweep cost 2.4x the first per tile,
// profiles/r01 e_pmc: 543 vs 222 VALU per wave and tile), with HC the sum is straight-line code.
template <typename T, int D, int NT, int HC>
__global__ __launch_bounds__(256) void scores_mfma_kernel(ScoresMfmaParams P) {
    using M = MfmaS<T>;
    using x8 = typename M::x8;
    constexpr int ROWB = D * 2;
    constexpr int PIECES = D / 8;
    constexpr int KSTEPS = D / 32;
    constexpr int TILE_ROWS = 64;
    constexpr int TILE_BYTES = TILE_ROWS * ROWB;
    constexpr int LD_PER_THREAD = TILE_ROWS * PIECES / 256;  // 16-B pieces per thread per tile
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * TILE_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = uniform(tid >> 6);
    const int rho = lane & 15, q = lane >> 4;
    const int h = HC ? HC : P.h;
    const int QPT = 16 / h;              // queries per 16-column tile
    const int QW = 4 * NT * QPT;         // queries per workgroup
    const int bg = blockIdx.y;
    const int b = bg / P.G, g = bg % P.G;
    // late query tiles first: with causal_skip the second sweep of a tile grows with its position (a tile at the end of a 64k sequence does
    // 1.5x the work of the first one), and the dispatcher hands out workgroups in index order -- longest first keeps the tail short
    const int t0 = (P.causal_skip ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x) * QW;
    const T *Kc = (const T *)P.Kc + (int64_t)b * P.csb + (int64_t)g * P.csg;
    const float c2 = P.scale * LOG2E;

    // ---- Q^T fragments of this wave's NT column tiles
    x8 qf[NT][KSTEPS];
    int tq[NT];  // query of this lane's column (or -1)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int qi = rho / h;  // query within the tile
        const int hh = rho % h;
        const int t = t0 + (wave * NT + n) * QPT + qi;
        const bool ok = qi < QPT && t < P.S;
        tq[n] = ok ? t : -1;
        const T *qr = (const T *)P.Q + ((((int64_t)b * P.S + (ok ? t : 0)) * P.G + g) * h + hh) * (int64_t)D;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            u32x4 raw = {0u, 0u, 0u, 0u};
            if (ok) raw = *(const u32x4 *)(qr + 32 * s + 8 * q);
            qf[n][s] = __builtin_bit_cast(x8, raw);
        }
    }

    const int ntiles = (P.S_cmp + TILE_ROWS - 1) / TILE_ROWS;
    // staging: thread -> (row, piece) pairs
    u32x4 stg[LD_PER_THREAD];
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int i = 0; i < LD_PER_THREAD; ++i) {
            const int p = tid + 256 * i;
            const int r = p / PIECES, pc = p % PIECES;
            const int row = min(tile * TILE_ROWS + r, P.S_cmp - 1);
            stg[i] = *(const u32x4 *)(Kc + (int64_t)row * P.css + pc * 8);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LD_PER_THREAD; ++i) {
            const int p = tid + 256 * i;
            const int r = p / PIECES, pc = p % PIECES;
            *(u32x4 *)(lds + buf * TILE_BYTES + r * ROWB + ((pc ^ (r & (PIECES - 1))) << 4)) = stg[i];
        }
    };
    // S^T accumulators of one 64-row tile for all NT column tiles
    auto compute_tile = [&](int buf, f32x4 (&acc)[4][NT]) {
        const unsigned char *base = lds + buf * TILE_BYTES;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = 16 * u + rho;
            x8 a[KSTEPS];
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) a[s] = *(const x8 *)(base + r * ROWB + (((4 * s + q) ^ (r & (PIECES - 1))) << 4));
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KSTEPS; ++s) c = M::mma(a[s], qf[n][s], c);
                acc[u][n] = c;
            }
        }
    };

    // ================= sweep 1: row max and row sum (per-lane online, exp2 domain) =================
    float mrun[NT], lrun[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        mrun[n] = -INFINITY;
        lrun[n] = 0.f;
    }
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int tile = 0; tile < ntiles; ++tile) {
        const int buf = tile & 1;
        if (tile + 1 < ntiles) load_tile(tile + 1);
        f32x4 acc[4][NT];
        compute_tile(buf, acc);
        const int rows_valid = P.S_cmp - tile * TILE_ROWS;  // rows >= this are padding
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            // common path: every lane keeps its own reference max and only accumulates sum(exp2(x - m)); no max, no
            // mask, no rescale.  All 16 exponents are <= 12 whenever the sum stays <= 2^12, so the sum itself is the
            // test: a larger (or inf/nan: first tile, m = -inf) sum, or a padded last tile, takes the exact slow path.
            float sum = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) sum += __builtin_amdgcn_exp2f(fmaf(acc[u][n][j], c2, -mrun[n]));
            if (__any(!(sum <= 4096.f)) || rows_valid < TILE_ROWS) {
                asm volatile("; sweep-1 slow path" ::: "memory");
                float v[16];
                float mx = -INFINITY;
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = 16 * u + 4 * q + j;
                        const float x = (r < rows_valid) ? acc[u][n][j] * c2 : -INFINITY;
                        v[4 * u + j] = x;
                        mx = fmaxf(mx, x);
                    }
                const float mnew = fmaxf(mrun[n], mx);
                sum = 0.f;
                if (mnew > -INFINITY) {  // a lane group may see only padding rows in the last tile
#pragma unroll
                    for (int i = 0; i < 16; ++i) sum += __builtin_amdgcn_exp2f(v[i] - mnew);
                    lrun[n] = lrun[n] * __builtin_amdgcn_exp2f(mrun[n] - mnew);
                    mrun[n] = mnew;
                }
            }
            lrun[n] += sum;
        }
        if (tile + 1 < ntiles) store_tile(buf ^ 1);
        __syncthreads();
    }
    // merge the 4 lane groups of each column
    float mlog[NT];  // m + log2(l): p = exp2(s*c2 - mlog)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        float m = fmaxf(mrun[n], __shfl_xor(mrun[n], 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float l = (mrun[n] > -INFINITY) ? lrun[n] * __builtin_amdgcn_exp2f(mrun[n] - m) : 0.f;
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        mlog[n] = m + __builtin_amdgcn_logf(l);
    }

    // ================= sweep 2: normalise, Eq.9 stencil, Eq.10 head sum, store =================
    const int l_sel = 4 * P.d_stride;
    int jlast = P.S_sel - 1;  // last selection block this workgroup has to produce
    if (P.causal_skip) {
        const int t_last = min(t0 + QW, P.S) - 1;
        jlast = min(jlast, (t_last + 1) / l_sel - 1);
    }
    const int tiles2 = (jlast < 0) ? 0 : min(ntiles, (4 * jlast + 3) / TILE_ROWS + 1);
    float rot_prev[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) rot_prev[n] = 0.f;
    if (tiles2 > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    for (int tile = 0; tile < tiles2; ++tile) {
        const int buf = tile & 1;
        if (tile + 1 < tiles2) load_tile(tile + 1);
        f32x4 acc[4][NT];
        compute_tile(buf, acc);
        const int rows_valid = P.S_cmp - tile * TILE_ROWS;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = 16 * tile + 4 * u + q;  // selection block of this lane group
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                float p[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) p[k] = __builtin_amdgcn_exp2f(fmaf(acc[u][n][k], c2, -mlog[n]));
                if (rows_valid < TILE_ROWS) {  // padded last tile only (wave uniform)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (16 * u + 4 * q + k >= rows_valid) p[k] = 0.f;
                }
                // 1/2 p[4j-1]: register 3 of the previous lane group (previous sub-tile for q == 0)
                const float rot = __shfl(p[3], (lane + 48) & 63, 64);
                const float tapm1 = (q == 0) ? rot_prev[n] : rot;
                rot_prev[n] = rot;
                float slc = fmaf(0.5f, tapm1, p[0]);
                slc += p[1];
                slc += p[2];
                slc = fmaf(0.5f, p[3], slc);
                // Eq.10: heads of one query are h consecutive lanes of the 16-lane row; ascending-h sum with DPP row
                // shifts (lane i reads lane i+k of its row), no LDS crossbar traffic
                float grp = slc;
#define NSA_HS(K) \
    if (K < h) grp += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, slc), 0x100 | K, 0xf, 0xf, true));
                NSA_HS(1) NSA_HS(2) NSA_HS(3) NSA_HS(4) NSA_HS(5) NSA_HS(6) NSA_HS(7) NSA_HS(8)
                NSA_HS(9) NSA_HS(10) NSA_HS(11) NSA_HS(12) NSA_HS(13) NSA_HS(14) NSA_HS(15)
#undef NSA_HS
                if (tq[n] >= 0 && (rho % h) == 0 && j <= jlast)
                    P.p_grp[(((int64_t)b * P.S + tq[n]) * P.G + g) * (int64_t)P.S_sel + j] = grp;
            }
        }
        if (tile + 1 < tiles2) store_tile(buf ^ 1);
        __syncthreads();
    }
}

// ---- host -----------------------------------------------------------------------------------
bool scores_mfma_supported(int dtype, int h, int Dk, int l, int d, int l_sel) {
    return (dtype == NSA_DT_BF16 || dtype == NSA_DT_F16) && (Dk == 64 || Dk == 128) && h >= 1 && h <= 16 && d > 0 && l == 2 * d &&
           l_sel == 4 * d;
}

template <typename T, int D>
static int launch_scores_t(const ScoresMfmaParams &P, hipStream_t st) {
    constexpr int NT = 4;
    const int QPT = 16 / P.h;
    const int QW = 4 * NT * QPT;
    dim3 grid((unsigned)((P.S + QW - 1) / QW), (unsigned)(P.B * P.G));
    switch (P.h) {  // the common group sizes get straight-line head sums
        case 6: hipLaunchKernelGGL((scores_mfma_kernel<T, D, NT, 6>), grid, dim3(256), 0, st, P); break;
        case 4: hipLaunchKernelGGL((scores_mfma_kernel<T, D, NT, 4>), grid, dim3(256), 0, st, P); break;
        case 8: hipLaunchKernelGGL((scores_mfma_kernel<T, D, NT, 8>), grid, dim3(256), 0, st, P); break;
        default: hipLaunchKernelGGL((scores_mfma_kernel<T, D, NT, 0>), grid, dim3(256), 0, st, P); break;
    }
    NSA_LAUNCH_CHECK("scores_mfma");
    return NSA_OK;
}

int launch_sel_scores_mfma(const void *Q, const void *Kc, float *p_grp, int B, int S, int G, int h, int Dk, int S_cmp,
                           int64_t csb, int64_t csg, int64_t css, int S_sel, int d_stride, int dtype, float scale,
                           int causal_skip, hipStream_t st) {
    NSA_CHECK_ARG(css % 8 == 0 && csb % 8 == 0 && csg % 8 == 0 && ((uintptr_t)Q % 16 == 0) && ((uintptr_t)Kc % 16 == 0),
                  "scores_mfma: Q/K_cmp must be 16-byte aligned with strides that are multiples of 8 elements");
    NSA_CHECK_ARG((int64_t)B * G <= 65535, "scores_mfma: B*G too large for one launch");
    NSA_CHECK_ARG(S_cmp >= 1, "scores_mfma: S_cmp must be >= 1");
    // blocks the second sweep does not visit (causal skip, or selection blocks without any compressed row) are zero -- unless the caller
    // asked for causal_skip == 2: it then reads only entries with (j+1) l' <= t+1 (what both selectors do), all of which are written
    if (causal_skip != 2) NSA_HIP_TRY(hipMemsetAsync(p_grp, 0, sizeof(float) * (size_t)B * S * G * S_sel, st));
    ScoresMfmaParams P{Q, Kc, p_grp, B, S, G, h, S_cmp, S_sel, csb, csg, css, scale, causal_skip, d_stride};
    if (dtype == NSA_DT_BF16) return Dk == 64 ? launch_scores_t<__bf16, 64>(P, st) : launch_scores_t<__bf16, 128>(P, st);
    return Dk == 64 ? launch_scores_t<_Float16, 64>(P, st) : launch_scores_t<_Float16, 128>(P, st);
}

}  // namespace nsa
