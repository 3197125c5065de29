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
#include "sel_scores_mfma.hpp"
#include "sel_select_row.hpp"

namespace nsa {

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
// FLAT (HC = 6 with NT = 3): the 48 columns of a wave's three tiles are the 8 x 6 (query, head) pairs of 8 queries laid end to end, instead
// of 2 queries (12 of 16 columns) per tile and four tiles: a quarter fewer MFMAs, exponentials and adds for the same 8 queries.  On gfx950
// MFMA and VALU issue do not overlap on a SIMD (tools/ubench/issue_rates.hip: 4 MFMA + 8 FMA take 90 cycles, 67 + 31 apart), so the time of
// this kernel is the SUM of its MFMA and VALU cycles and every unused column costs both.  Queries 2 and 5 of the eight straddle two tiles:
// their head sum takes the tail of one tile's shifted sums and the head of the next one's (one more DPP add each).
template <typename T, int D, int NT, int HC>
__global__ __launch_bounds__(256, 2) void scores_mfma_kernel(ScoresMfmaParams P) {
    constexpr bool FLAT = (HC == 6 && NT == 3);
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
    const int QPT = 16 / h;                          // queries per 16-column tile
    const int QPW = FLAT ? 8 : NT * QPT;             // queries per wave
    const int QW = 4 * QPW;                          // queries per workgroup
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
        const int col = FLAT ? 16 * n + rho : rho;  // FLAT: column of the wave's 48, else column of the tile
        const int qi = col / h;                      // query within the wave (FLAT) / the tile
        const int hh = col % h;
        const int t = FLAT ? t0 + wave * QPW + qi : t0 + (wave * NT + n) * QPT + qi;
        const bool ok = (FLAT || qi < QPT) && t < P.S;
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
    // full tiles: one scalar base per tile + a lane-constant 32-bit offset (the per-lane 64-bit row arithmetic with its clamp was 16 VALU
    // instructions per tile and wave in a kernel bound by instruction issue); only the last, padded tile clamps its rows
    unsigned toff[LD_PER_THREAD];
#pragma unroll
    for (int i = 0; i < LD_PER_THREAD; ++i) {
        const int p = tid + 256 * i;
        toff[i] = (unsigned)((p / PIECES) * (int)P.css + (p % PIECES) * 8) * (unsigned)sizeof(T);
    }
    const bool small_stride = P.css * (int64_t)TILE_ROWS * (int64_t)sizeof(T) < ((int64_t)1 << 31);
    auto load_tile = [&](int tile) {
        if (small_stride && (tile + 1) * TILE_ROWS <= P.S_cmp) {
            const unsigned char *base = (const unsigned char *)(Kc + (int64_t)tile * TILE_ROWS * P.css);  // wave uniform
#pragma unroll
            for (int i = 0; i < LD_PER_THREAD; ++i) stg[i] = *(const u32x4 *)(base + toff[i]);
            return;
        }
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
            // packed fp32: one v_pk_fma_f32 forms two exponents and one v_pk_add_f32 adds two terms (the exp itself stays scalar): 4 VALU
            // instructions per two logits instead of 6 in a kernel bound by VALU issue.  Two partial sums, added at the end.
            f32x2 sum2 = {0.f, 0.f};
            const f32x2 c22 = {c2, c2}, nm2 = {-mrun[n], -mrun[n]};
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const f32x2 x2 = {acc[u][n][j], acc[u][n][j + 1]};
                    const f32x2 t2 = __builtin_elementwise_fma(x2, c22, nm2);
                    const f32x2 e2 = {__builtin_amdgcn_exp2f(t2[0]), __builtin_amdgcn_exp2f(t2[1])};
                    sum2 += e2;
                }
            float sum = sum2[0] + sum2[1];
            if (__any(!(sum <= 4096.f)) || rows_valid < TILE_ROWS) {
                asm volatile("; sweep-1 slow path" ::: "memory");
                int rv = rows_valid;  // (same reason as for the logits below: the 16 row masks belong to this block)
                asm volatile("" : "+s"(rv));
                float v[16];
                float mx = -INFINITY;
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = 16 * u + 4 * q + j;
                        // the logit enters this block through an (empty) asm: without it the compiler evaluates the masks, products and maxima
                        // of this rare path ahead of the branch, on every tile (90 of 255 VALU instructions of the common path)
                        float a = acc[u][n][j];
                        asm volatile("" : "+v"(a));
                        const float x = (r < rv) ? a * c2 : -INFINITY;
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
    // output addressing: one scalar base per sequence + a 32-bit element offset per column tile (the host sets big_out when S G S_sel
    // reaches 2^31 elements per sequence: 64-bit offsets then)
    float *pg_b = P.p_grp + (int64_t)b * P.S * P.G * P.S_sel;
    unsigned poff[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) poff[n] = (unsigned)((max(tq[n], 0) * P.G + g) * P.S_sel);
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
            float slcs[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                float p[4];
#pragma unroll
                for (int k = 0; k < 4; k += 2) {
                    const f32x2 x2 = {acc[u][n][k], acc[u][n][k + 1]}, c22 = {c2, c2}, nm2 = {-mlog[n], -mlog[n]};
                    const f32x2 t2 = __builtin_elementwise_fma(x2, c22, nm2);
                    p[k] = __builtin_amdgcn_exp2f(t2[0]);
                    p[k + 1] = __builtin_amdgcn_exp2f(t2[1]);
                }
                if (rows_valid < TILE_ROWS) {  // padded last tile only (wave uniform).  rv: the row masks stay inside this block (left to
                                               // the compiler they are formed and applied on every tile: 8 of 36 VALU per sub-tile)
                    int rv = rows_valid;
                    asm volatile("" : "+s"(rv));
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (16 * u + 4 * q + k >= rv) p[k] = 0.f;
                }
                // 1/2 p[4j-1]: register 3 of the previous lane group (previous sub-tile for q == 0)
                const float rot = __shfl(p[3], (lane + 48) & 63, 64);
                const float tapm1 = (q == 0) ? rot_prev[n] : rot;
                rot_prev[n] = rot;
                float slc = fmaf(0.5f, tapm1, p[0]);
                slc += p[1];
                slc += p[2];
                slc = fmaf(0.5f, p[3], slc);
                slcs[n] = slc;
            }
            // Eq.10: heads of one query are h consecutive columns: ascending-h sum with DPP row shifts (lane i reads lane i+k of its row, 0
            // past the row's end), no LDS crossbar traffic.  v_add_f32 with the shift on its first operand: one instruction per head (the
            // compiler keeps a v_mov_b32_dpp and an add apart).  s_nop 1: a DPP read of a VGPR the previous VALU instruction wrote needs two
            // wait states, and inline assembly is outside the compiler's hazard bookkeeping.  Outputs are early-clobber: they must not
            // share a register with an input a later step still reads.
            float grps[NT];
#define NSA_HS1 "s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define NSA_HSA(K) "v_add_f32_dpp %0, %1, %0 row_shl:" #K " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define NSA_HSB(K) "s_nop 1\n\tv_add_f32_dpp %0, %1, %2 row_shl:" #K " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            if constexpr (FLAT) {
                // columns 16 n + rho = 6 query + head.  Query starts: tile 0 lanes 0, 6, 12 (12: heads 0-3 here, 4-5 = lanes 0-1 of tile 1);
                // tile 1 lanes 2, 8, 14 (14: heads 0-1 here, 2-5 = lanes 0-3 of tile 2); tile 2 lanes 4, 10.
                float s0, t1, s1, t2, s2;
                asm volatile(NSA_HS1 NSA_HSA(2) NSA_HSA(3) NSA_HSA(4) NSA_HSA(5) : "=&v"(s0) : "v"(slcs[0]));
                asm volatile(NSA_HS1 : "=&v"(t1) : "v"(slcs[1]));                                            // lane 0: heads 4 + 5 of query 2
                asm volatile(NSA_HSB(2) NSA_HSA(3) NSA_HSA(4) NSA_HSA(5) : "=&v"(s1) : "v"(slcs[1]), "v"(t1));
                asm volatile(NSA_HS1 NSA_HSA(2) NSA_HSA(3) : "=&v"(t2) : "v"(slcs[2]));                      // lane 0: heads 2 .. 5 of query 5
                asm volatile(NSA_HSB(4) NSA_HSA(5) : "=&v"(s2) : "v"(slcs[2]), "v"(t2));
                // the straddling queries: lane 12 of tile 0 adds lane 0 of t1, lane 14 of tile 1 adds lane 0 of t2 (other lanes: + 0 or unused)
                asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 row_shr:12 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" : "=&v"(grps[0]) : "v"(t1), "v"(s0));
                asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 row_shr:14 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" : "=&v"(grps[1]) : "v"(t2), "v"(s1));
                grps[2] = s2;
            } else {
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const float slc = slcs[n];
                    float grp = slc;
                    if constexpr (HC == 4) asm volatile(NSA_HS1 NSA_HSA(2) NSA_HSA(3) : "=&v"(grp) : "v"(slc));
                    else if constexpr (HC == 6) asm volatile(NSA_HS1 NSA_HSA(2) NSA_HSA(3) NSA_HSA(4) NSA_HSA(5) : "=&v"(grp) : "v"(slc));
                    else if constexpr (HC == 8) asm volatile(NSA_HS1 NSA_HSA(2) NSA_HSA(3) NSA_HSA(4) NSA_HSA(5) NSA_HSA(6) NSA_HSA(7) : "=&v"(grp) : "v"(slc));
                    else {
#define NSA_HS(K) \
    if (K < h) grp += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, slc), 0x100 | K, 0xf, 0xf, true));
                        NSA_HS(1) NSA_HS(2) NSA_HS(3) NSA_HS(4) NSA_HS(5) NSA_HS(6) NSA_HS(7) NSA_HS(8)
                        NSA_HS(9) NSA_HS(10) NSA_HS(11) NSA_HS(12) NSA_HS(13) NSA_HS(14) NSA_HS(15)
#undef NSA_HS
                    }
                    grps[n] = grp;
                }
            }
#undef NSA_HSB
#undef NSA_HSA
#undef NSA_HS1
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const bool first_head = FLAT ? ((16 * n + rho) % h) == 0 : (rho % h) == 0;
                if (tq[n] >= 0 && first_head && j <= jlast)
                {
                    if (P.big_out) P.p_grp[(((int64_t)b * P.S + tq[n]) * P.G + g) * (int64_t)P.S_sel + j] = grps[n];
                    else pg_b[poff[n] + (unsigned)j] = grps[n];
                }
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
    const bool flat = P.h == 6 && tuning(TUNE_SCORES_FORM) != 0;  // 8 queries per wave on 3 column tiles instead of 4 (A/B switch: 0 = 4 tiles)
    const int QW = flat ? 32 : 4 * NT * (16 / P.h);
    dim3 grid((unsigned)((P.S + QW - 1) / QW), (unsigned)(P.B * P.G));
    if (flat) {
        hipLaunchKernelGGL((scores_mfma_kernel<T, D, 3, 6>), grid, dim3(256), 0, st, P);
    } else {
        switch (P.h) {  // the common group sizes get straight-line head sums
            case 6: hipLaunchKernelGGL((scores_mfma_kernel<T, D, NT, 6>), grid, dim3(256), 0, st, P); break;
            case 4: hipLaunchKernelGGL((scores_mfma_kernel<T, D, NT, 4>), grid, dim3(256), 0, st, P); break;
            case 8: hipLaunchKernelGGL((scores_mfma_kernel<T, D, NT, 8>), grid, dim3(256), 0, st, P); break;
            default: hipLaunchKernelGGL((scores_mfma_kernel<T, D, NT, 0>), grid, dim3(256), 0, st, P); break;
        }
    }
    NSA_LAUNCH_CHECK("scores_mfma");
    return NSA_OK;
}

// sel / sel_done: the caller wants the top-n ranges of every row too; *sel_done says whether this launch produced them (the 32x32x16 form
// selects inside the launch) or the caller has to run the select kernel behind it
int launch_sel_scores_mfma(const void *Q, const void *Kc, float *p_grp, int B, int S, int G, int h, int Dk, int S_cmp,
                           int64_t csb, int64_t csg, int64_t css, int S_sel, int d_stride, int dtype, float scale,
                           int causal_skip, hipStream_t st, const SelectParams *sel, int *sel_done) {
    if (sel_done) *sel_done = 0;
    NSA_CHECK_ARG(css % 8 == 0 && csb % 8 == 0 && csg % 8 == 0 && ((uintptr_t)Q % 16 == 0) && ((uintptr_t)Kc % 16 == 0),
                  "scores_mfma: Q/K_cmp must be 16-byte aligned with strides that are multiples of 8 elements");
    NSA_CHECK_ARG((int64_t)B * G <= 65535, "scores_mfma: B*G too large for one launch");
    NSA_CHECK_ARG(S_cmp >= 1, "scores_mfma: S_cmp must be >= 1");

    // blocks the second sweep does not visit (causal skip, or selection blocks without any compressed row) are zero -- unless the caller
    // asked for causal_skip == 2: it then reads only entries with (j+1) l' <= t+1 (what both selectors do), all of which are written
    if (causal_skip != 2) NSA_HIP_TRY(hipMemsetAsync(p_grp, 0, sizeof(float) * (size_t)B * S * G * S_sel, st));
    ScoresMfmaParams P{Q, Kc, p_grp, B, S, G, h, S_cmp, S_sel, csb, csg, css, scale, causal_skip, d_stride,
                       (int64_t)S * G * S_sel < ((int64_t)1 << 31) ? 0 : 1};
    const int form = tuning(TUNE_SCORES_FORM);
    if ((form < 0 || form == 2) && scores_mfma32_supported(P, Dk)) {
        // in the launch only where it pays (same box, tools/bench_scores_select.py: 64k x 4 3,537 -> 3,433 us, 64k x 1 905 -> 890, but 32k x 2
        // 509 -> 521, 16k x 2 161 -> 180, 4k x 8 75 -> 87): a wave's 16 rows are 16 dependent chains of ~2 us behind its sweeps, which only
        // a long sweep (a workgroup's lifetime grows with S_cmp) and many workgroups per CU hide.  TUNE_SCORES_SELECT: -1 this rule, 0 never, 1 always
        const int mode = tuning(TUNE_SCORES_SELECT);
        const bool fuse = sel && sel_done && mode != 0 && (mode > 0 || S_cmp >= 3072) && scores_mfma32_select_supported(P, Dk, *sel);
        if (fuse) *sel_done = 1;
        return launch_scores_mfma32(P, dtype, st, fuse ? sel : nullptr);
    }
    if (dtype == NSA_DT_BF16) return Dk == 64 ? launch_scores_t<__bf16, 64>(P, st) : launch_scores_t<__bf16, 128>(P, st);
    return Dk == 64 ? launch_scores_t<_Float16, 64>(P, st) : launch_scores_t<_Float16, 128>(P, st);
}

}  // namespace nsa
