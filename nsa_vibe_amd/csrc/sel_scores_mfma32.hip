// Fused selection scorer on 32x32x16 MFMA tiles (gfx950), h = 6 heads per group and D = 64: Q, K_cmp -> p_grp.
//
// Same reference chain and the same two sweeps as sel_scores_mfma.hip (compute_pcmp_all, selection_scorer.py:42-61 ->
// map_pcmp_to_pslc_batched, :89-116 -> .sum(dim=3), nsa_attention.py:1091; Eq.9 as the 5-tap stencil of l = 2d, l' = 4d).  What changes is
// the tile shape and, in the second sweep, which operand supplies the MFMA rows.
//
// Cost model (tools/ubench/issue_rates.hip, profiles/r03/issue_rates.txt and scorer_notes.txt; cycles of one SIMD at the nominal clock, two or
// more waves to pick from): v_fma / v_add / v_mul 2.6, v_pk_fma_f32 4.9-5.8 (two plain ops cost the same), DPP add 4.2, v_exp_f32 8.2,
// MFMA 32x32x16 33, MFMA 16x16x32 16.3 (the same per flop).  Matrix and vector instructions of a SIMD do NOT run side by side to any
// useful degree, whatever the tile shape, the number of resident waves (2 or 3) or their relative phase: with 3 of the 4 k-steps removed this
// kernel loses exactly the time of the removed MFMAs, with v_exp_f32 replaced by a multiply exactly the difference of their costs.  The time
// of a scorer is the SUM of its instruction costs; this form is the one with the smallest sum found:
// * 32x32x16 tiles: the same matrix time, half the LDS reads per query, and 16 accumulator registers of a lane in ONE row or column group.
// * One wave = 16 queries = 96 (query, head) pairs = three 32-wide tiles, no idle column; one workgroup = 4 waves = 64 queries sharing the
//   K_cmp tile in LDS.
// * Sweep 1 (row max / row sum): S^T = K_cmp tile [32 x D] . Q^T [D x 32]: compressed keys are the MFMA rows, a lane owns ONE (query, head)
//   column per tile, so the running reference max and sum are one register each per tile.  Per logit: v_fma, v_exp, v_add.
// * Sweep 2 (normalise, Eq.10 head sum, Eq.9 stencil, store): the operands swap, S = Q [32 x D] . K_cmp^T [D x 32]: a lane owns one
//   COMPRESSED KEY (column) and 16 (query, head) rows per tile.  The head sum is then plain adds inside the lane (the 6 rows of a query are
//   4 + 2 registers of the two lane halves: two v_permlane32_swap per 4 queries carry the 2-row parts across), and the stencil runs on the
//   head-summed value -- once per query instead of once per head -- with its lane movement done by the LDS: the 8 sums of a lane are written
//   as [query][column] rows, read back as one ds_read_b128 (the 4 columns of a block) + one ds_read_b32 (column 4j - 1) per (query, block),
//   and 4 plain ops finish Eq.9; the stores go out per two half tiles, 64 aligned bytes per query and instruction.  LDS instructions issue beside the vector pipe (LDS ~30 % busy here), DPP ones on it: 6 DPP operations per
//   (query, 8 blocks) measured 13.8 ms at 64k x 16 where this form measures 13.3 (same box).
//   The summation order differs from the 16x16 form (heads first, then taps): same fp32 arithmetic, last-bit differences.
// Per 32 compressed rows x 96 pairs and wave: sweep 1 = 12 MFMA + 48 x (fma, exp, add) ~ 1,100 cycles, sweep 2 = 12 MFMA + 48 x (fma, exp) +
// ~60 head-sum + 8 stencil operations ~ 1,200; floor of the two-sweep algorithm (MFMA + exp + one fma per logit, nothing else) ~ 950 each.
// Ablation switches (timing only, results meaningless; tools/ablate_scorer.sh): SC32_KSTEPS=n keeps n of the 4 k-steps, SC32_NOEXP replaces
// v_exp_f32 by a multiply, SC32_NOSYNC drops staging and barriers, SC32_SWEEP1 stops after the first sweep.
#include "sel_scores_mfma.hpp"
#include "sel_select_row.hpp"
#ifdef SC32_NOEXP  // ablation: a full-rate VALU op in place of v_exp_f32
#define SC32_EXP(x) ((x) * 0.001f)
#else
#define SC32_EXP(x) __builtin_amdgcn_exp2f(x)
#endif
#ifdef SC32_KSTEPS  // ablation build: only the first SC32_KSTEPS of the 4 k-steps (timing only, results meaningless)
#define SC32_MMA(a, b, c) (s < SC32_KSTEPS ? M::mma(a, b, c) : c)
#else
#define SC32_MMA(a, b, c) M::mma(a, b, c)
#endif

namespace nsa {

template <typename T>
struct Mfma32;
template <>
struct Mfma32<__bf16> {
    using x8 = bf16x8;
    __device__ static f32x16 mma(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <>
struct Mfma32<_Float16> {
    using x8 = f16x8;
    __device__ static f32x16 mma(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// Lane layout of one 32x32x16 MFMA (tools/ubench/probe_lanes.hip): A row and B column = lane & 31, both with k = 8 (lane >> 5) + e;
// acc[i] of a lane = C[8 (i >> 2) + 4 (lane >> 5) + (i & 3)][lane & 31].
// SEL (round 4): the wave also runs the top-n selection of its 16 query rows (select_topn_row_auto: the select kernel's own row function, so
// the ranges are those of the separate launch bit for bit) right after its second sweep.  The scores it reads are the ones it has just stored
// (wave-private rows; its stores drained first): L2 hits instead of the 4.3 GB HBM read of a select launch at 64k x 16, and the selector's
// scalar chains run beside the other waves' matrix / vector work -- this kernel leaves the CU's scalar unit idle, the select kernel is bound by it.
template <typename T, bool SEL>
__global__ __launch_bounds__(256, 3) void scores_mfma32_kernel(ScoresMfmaParams P, SelectParams SP) {
    using M = Mfma32<T>;
    using x8 = typename M::x8;
    constexpr int D = 64, HC = 6;
    constexpr int ROWB = D * 2;
    constexpr int TILE_ROWS = 64;              // K_cmp rows per LDS tile (one barrier per tile; 128 and 256 measured slower: registers)
    constexpr int NH = TILE_ROWS / 32;         // 32-row halves per tile
    constexpr int NLD = TILE_ROWS / 32;        // 16-B pieces per thread per tile
    constexpr int TILE_BYTES = TILE_ROWS * ROWB;
    constexpr int QPW = 16, QW = 64;  // queries per wave / workgroup
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * TILE_BYTES];
    __shared__ __attribute__((aligned(16))) float mlg[4][96];
    // Eq.9 staging, per wave 16 rows (8 head-summed registers x 2 lane halves) of a 128-column ring (four half tiles = two output pairs)
    __shared__ __attribute__((aligned(16))) float stn[4][16][128];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = uniform(tid >> 6);
    const int r = lane & 31, half = lane >> 5;
    const int bg = blockIdx.y;
    const int b = bg / P.G, g = bg % P.G;
    // late query tiles first (longest second sweep first), as in the 16x16 form
    const int t0 = (P.causal_skip ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x) * QW;
    const int tw = t0 + wave * QPW;
    const T *Kc = (const T *)P.Kc + (int64_t)b * P.csb + (int64_t)g * P.csg;
    const float c2 = P.scale * LOG2E;

    // ---- Q fragments: row / column 32 n + r of the wave's 96 (query, head) pairs, k-step s = dims 16 s + 8 half ..
    x8 qf[3][4];
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        // pair of row R = 32 n + r: the 48 rows a lane half owns in the second sweep (rows 8 gq + 4 half + e of every 32: bit 2 of R = the
        // half) are 8 COMPLETE queries in the lane's register order -- index within the half idx = 4 (R >> 3) + (R & 3) -> query 8 half + idx / 6,
        // head idx % 6 -- so the Eq.10 head sum needs no exchange between the lane halves (rows as R / 6, R % 6: two permlane swaps and two
        // DPP adds per four queries, 60 vector operations per half tile against 40 now)
        const int R = 32 * n + r;
        const int ridx = ((R >> 3) << 2) | (R & 3);
        const int t = tw + 8 * ((R >> 2) & 1) + ridx / HC, hh = ridx % HC;
        const bool ok = t < P.S;
        const T *qr = (const T *)P.Q + ((((int64_t)b * P.S + (ok ? t : 0)) * P.G + g) * HC + hh) * (int64_t)D;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x4 raw = {0u, 0u, 0u, 0u};
            if (ok) raw = *(const u32x4 *)(qr + 16 * s + 8 * half);
            qf[n][s] = __builtin_bit_cast(x8, raw);
        }
    }

    const int ntiles = (P.S_cmp + TILE_ROWS - 1) / TILE_ROWS;
    // staging of one 64-row K_cmp tile: thread -> 2 (row, 16-B piece) pairs.  LDS piece swizzle (row >> 1) & 7: the 16 rows x one piece a
    // quarter-wave reads, and the 2 rows x 8 pieces it writes, both cover the 64 banks exactly once.
    u32x4 stg[NLD];
    unsigned toff[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int p = tid + 256 * i;
        toff[i] = (unsigned)((p >> 3) * (int)P.css + (p & 7) * 8) * (unsigned)sizeof(T);
    }
    const bool small_stride = P.css * (int64_t)TILE_ROWS * (int64_t)sizeof(T) < ((int64_t)1 << 31);
    auto load_tile = [&](int tile) {
        if (small_stride && (tile + 1) * TILE_ROWS <= P.S_cmp) {
            const unsigned char *base = (const unsigned char *)(Kc + (int64_t)tile * TILE_ROWS * P.css);  // wave uniform
#pragma unroll
            for (int i = 0; i < NLD; ++i) stg[i] = *(const u32x4 *)(base + toff[i]);
            return;
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int p = tid + 256 * i;
            const int row = min(tile * TILE_ROWS + (p >> 3), P.S_cmp - 1);
            stg[i] = *(const u32x4 *)(Kc + (int64_t)row * P.css + (p & 7) * 8);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int p = tid + 256 * i;
            const int row = p >> 3, pc = p & 7;
            *(u32x4 *)(lds + buf * TILE_BYTES + row * ROWB + ((pc ^ ((row >> 1) & 7)) << 4)) = stg[i];
        }
    };
    // K fragment of the 32-row half hf of a tile: row 32 hf + r, dims 16 s + 8 half .. (piece 2 s + half)
    unsigned koff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) koff[s] = (unsigned)(r * ROWB + (((2 * s + half) ^ ((r >> 1) & 7)) << 4));
    auto load_kf = [&](int buf, int hf, x8 (&kf)[4]) {
        const unsigned char *base = lds + buf * TILE_BYTES + hf * (32 * ROWB);
#pragma unroll
        for (int s = 0; s < 4; ++s) kf[s] = *(const x8 *)(base + koff[s]);
    };
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // ================= sweep 1: row max and row sum per (query, head) column (per-lane online, exp2 domain) =================
    float mrun[3], lrun[3];
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        mrun[n] = -INFINITY;
        lrun[n] = 0.f;
    }
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int tile = 0; tile < ntiles; ++tile) {
#ifdef SC32_NOSYNC  // ablation: no staging, no barriers (every tile reads buffer 0)
        const int buf = 0;
#else
        const int buf = tile & 1;
#endif
#ifndef SC32_NOSYNC
        if (tile + 1 < ntiles) load_tile(tile + 1);
#endif
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) {
            const int rows_valid = P.S_cmp - tile * TILE_ROWS - 32 * hf;  // rows of this half >= this are padding
            if (rows_valid <= 0) break;
            x8 kf[4];
            load_kf(buf, hf, kf);
            f32x16 acc[3];
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[n] = zero16;
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int n = 0; n < 3; ++n) acc[n] = SC32_MMA(kf[s], qf[n][s], acc[n]);
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                // common path (as in the 16x16 form): the lane keeps its reference max and only accumulates sum(exp2(x - m)); all 16 exponents
                // are <= 12 whenever the sum stays <= 2^12, so the sum is the test; a larger (or inf / nan: m = -inf) sum or a padded half
                // takes the exact slow path
                // plain v_fma_f32 / v_add_f32, two partial sums (a packed fp32 op costs what its two halves cost: 4.9-5.8 against 2 x 2.6 cycles)
                float s0 = 0.f, s1 = 0.f;
                const float nm = -mrun[n];
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    s0 += SC32_EXP(__builtin_fmaf(acc[n][i], c2, nm));
                    s1 += SC32_EXP(__builtin_fmaf(acc[n][i + 1], c2, nm));
                }
                float sum = s0 + s1;
                if (__any(!(sum <= 4096.f)) || rows_valid < 32) {
                    asm volatile("; sweep-1 slow path" ::: "memory");
                    int rv = rows_valid;  // the row masks belong to this block
                    asm volatile("" : "+s"(rv));
                    float v[16];
                    float mx = -INFINITY;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int row = 8 * (i >> 2) + 4 * half + (i & 3);
                        float a = acc[n][i];  // through an (empty) asm: keeps the masks, products and maxima of this rare path behind the branch
                        asm volatile("" : "+v"(a));
                        const float x = (row < rv) ? a * c2 : -INFINITY;
                        v[i] = x;
                        mx = fmaxf(mx, x);
                    }
                    const float mnew = fmaxf(mrun[n], mx);
                    sum = 0.f;
                    if (mnew > -INFINITY) {  // a lane may see only padding rows
#pragma unroll
                        for (int i = 0; i < 16; ++i) sum += __builtin_amdgcn_exp2f(v[i] - mnew);
                        lrun[n] = lrun[n] * __builtin_amdgcn_exp2f(mrun[n] - mnew);
                        mrun[n] = mnew;
                    }
                }
                lrun[n] += sum;
            }
        }
#ifndef SC32_NOSYNC
        if (tile + 1 < ntiles) store_tile(buf ^ 1);
        __syncthreads();
#endif
    }
    // merge the two lane halves of each column; m + log2(l) goes through LDS to the lanes that own the pair as a ROW in sweep 2
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        const float m = xor32_max(mrun[n]);
        float l = (mrun[n] > -INFINITY) ? lrun[n] * __builtin_amdgcn_exp2f(mrun[n] - m) : 0.f;
        l = xor32_add(l);
        if (half == 0) mlg[wave][32 * n + r] = -(m + __builtin_amdgcn_logf(l));  // p = exp2(s c2 + mlg)
    }
    wave_lds_fence();
    // the 48 constants of a lane (rows 32 n + 8 gq + 4 half + e) are read from LDS where they are used: kept in registers they are the
    // difference between 3 and 2 waves per SIMD (broadcast reads, 12 ds_read_b128 per half tile)

    // ================= sweep 2: normalise, Eq.10 head sum (in lane), Eq.9 stencil (along the lanes), store =================
    const int l_sel = 4 * P.d_stride;
    int jlast = P.S_sel - 1;  // last selection block this workgroup has to produce
    if (P.causal_skip) {
        const int t_last = min(t0 + QW, P.S) - 1;
        jlast = min(jlast, (t_last + 1) / l_sel - 1);
    }
    const int nhalf2 = (jlast < 0) ? 0 : min(NH * ntiles, (4 * jlast + 3) / 32 + 1);  // 32-row halves (8 selection blocks each) to visit
#ifdef SC32_SWEEP1  // ablation: first sweep only
    const int tiles2 = 0;
#else
    const int tiles2 = (nhalf2 + NH - 1) / NH;
#endif
    // the scores of this sequence as a buffer: the stores take a 32-bit lane offset + a scalar offset (no 64-bit address arithmetic on the
    // vector pipe), and rows past the end of the sequence fall outside the buffer
    float *pg_b = P.p_grp + (int64_t)b * P.S * P.G * P.S_sel;
    const __amdgpu_buffer_rsrc_t pg_rs = [&] {
        const uint64_t a = (uint64_t)pg_b;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), (short)0,
                                                 __builtin_amdgcn_readfirstlane((int)((unsigned)P.S * (unsigned)P.G * (unsigned)P.S_sel * 4u)), 0x00020000);
    }();
    const bool all_rows = t0 + QW <= P.S;  // otherwise the stores check their query (last workgroup of a sequence)
    // Eq.9 through LDS (no VALU work for the lane movement: LDS instructions issue beside the vector pipe).  The 8 head-summed registers of a
    // half tile are written as rows [register z + 8 half][column mod 128].  After every second half tile (64 columns = 16 blocks) lane r
    // reads, for register z' = 2 rd + (r >> 4) (rd = 0 .. 3) and block jb = r & 15 of the pair, the four columns of the block with one
    // ds_read_b128 and column 4 jb - 1 with a ds_read_b32 (the ring makes the last column of the previous pair the neighbour of the first), and
    // stores: 16 consecutive lanes write the 64 contiguous, 64-byte aligned bytes of one query (8-block runs per half tile were 32 bytes and
    // left the L2 with half-written sectors: 6.0 GB of writes for 4.3 GB of scores at 64k x 16).
    float *st_w = &stn[wave][8 * half][0];
    const int q2 = r >> 4, jb = r & 15;
    // query of (rd, this lane) = tw + 8 half + 2 rd + q2 (ring row 8 half + z holds query 8 half + z of the wave): byte offset of its block jb in
    // the sequence's p_grp for rd = 0, then a uniform stride
    const unsigned poff0 = ((unsigned)(((tw + q2 + 8 * half) * P.G + g) * P.S_sel) + (unsigned)jb) * 4u;
    const unsigned rdstride = (unsigned)(2 * P.G * P.S_sel) * 4u;
    if (lane < 16) stn[wave][lane][127] = 0.f;  // column -1 of the first half tile: outside [0, S_cmp), dropped
    if (tiles2 > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    for (int tile = 0; tile < tiles2; ++tile) {
#ifdef SC32_NOSYNC  // ablation: no staging, no barriers (every tile reads buffer 0)
        const int buf = 0;
#else
        const int buf = tile & 1;
#endif
#ifndef SC32_NOSYNC
        if (tile + 1 < tiles2) load_tile(tile + 1);
#endif
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) {
            const int hfi = NH * tile + hf;
            if (hfi >= nhalf2) break;
            x8 kf[4];
            load_kf(buf, hf, kf);
            f32x16 acc[3];
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[n] = zero16;
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int n = 0; n < 3; ++n) acc[n] = SC32_MMA(qf[n][s], kf[s], acc[n]);
            // p of this lane's compressed key for its 48 (query, head) rows, as 12 chunks of 4 consecutive rows: chunk k = 4 n + gq
            float p[12][4];
            int mlo = 4 * half;
            asm volatile("" : "+v"(mlo));  // (an opaque OFFSET: the loads stay inside the loop and stay LDS reads -- an opaque pointer made them flat loads)
            const float *mlp = &mlg[wave][mlo];
#pragma unroll
            for (int n = 0; n < 3; ++n)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f32x4 mlv = *(const f32x4 *)(mlp + 32 * n + 8 * gq);
#pragma unroll
                    for (int e = 0; e < 4; ++e) p[4 * n + gq][e] = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[n][4 * gq + e], c2, mlv[e]));
                }
            const int cols_valid = P.S_cmp - 32 * hfi;
            if (cols_valid < 32) {  // padded last half only (wave uniform): compressed keys past the end contribute nothing
                int cv = cols_valid;
                asm volatile("" : "+s"(cv));
                const bool dead = r >= cv;
#pragma unroll
                for (int k = 0; k < 12; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (dead) p[k][e] = 0.f;
            }
            // Eq.10: the lane's 48 values in register order (chunk k, element e) are rows idx = 4 k + e of its half = queries idx / 6 (see the Q
            // fragments): six consecutive values per query, summed as ((a + b) + (c + d)) + (e + f)
            float zs[8];
#pragma unroll
            for (int z = 0; z < 8; ++z) {
                auto pf = [&](int i) -> float { return p[(6 * z + i) >> 2][(6 * z + i) & 3]; };
                zs[z] = ((pf(0) + pf(1)) + (pf(2) + pf(3))) + (pf(4) + pf(5));
            }
            {
#pragma unroll
                for (int z = 0; z < 8; ++z) st_w[128 * z + 32 * (hfi & 3) + r] = zs[z];
                wave_lds_fence();
                if ((hfi & 1) || hfi == nhalf2 - 1) {  // a pair of half tiles is complete (or the sweep ends on a single one)
                    const int pr = hfi >> 1, base = 64 * (pr & 1);
                    const bool mine = 16 * pr + jb <= jlast;
#pragma unroll
                    for (int rd = 0; rd < 4; ++rd) {
                        const float *row = st_w + 128 * (2 * rd + q2);
                        const f32x4 x = *(const f32x4 *)(row + base + 4 * jb);
                        const float xm1 = row[(base + 4 * jb + 127) & 127];
                        // (1/2 x[-1] + x[0]) + x[1] + x[2] + 1/2 x[3]: the order of the 16x16 form, on head sums
                        float y = __builtin_fmaf(0.5f, xm1, x[0]);
                        y += x[1];
                        y += x[2];
                        y = __builtin_fmaf(0.5f, x[3], y);
                        if (mine && (all_rows || tw + 8 * half + 2 * rd + q2 < P.S))
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), pg_rs, (int)poff0, (int)(64u * (unsigned)pr + (unsigned)rd * rdstride), 0);
                    }
                    wave_lds_fence();  // the next pair overwrites the other half of the ring only after these reads
                }
            }
        }
#ifndef SC32_NOSYNC
        if (tile + 1 < tiles2) store_tile(buf ^ 1);
        __syncthreads();
#endif
    }
    if constexpr (SEL) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's score stores have reached L2
        int *sc = (int *)&stn[wave][0][0];                 // run-extraction scratch (the stencil ring is dead)
        // the wave's 16 rows one after the other, the next row's scores fetched while this one is selected (a row is a chain of dependent
        // steps behind its loads: without the overlap the epilogue held the workgroup's registers and LDS for 16 L2 round trips more)
        const int nq = min(QPW, P.S - tw);
        const int sh = SP.l_sel_shift;
        auto fetch = [&](int qi, float (&dst)[16]) {
            const int t = SP.t0 + tw + min(qi, nq - 1);
            const int nvalid = min(SP.S_sel, sh >= 0 ? (t + 1) >> sh : (t + 1) / SP.l_sel);
            const int jmax = max(nvalid - 1, 0);
            const float *pr = SP.p_grp + (((int64_t)b * P.S + tw + min(qi, nq - 1)) * P.G + g) * (int64_t)SP.S_sel;
#pragma unroll
            for (int c = 0; c < 16; ++c) dst[c] = pr[min(lane + 64 * c, jmax)];
        };
        auto select = [&](int qi, const float (&src)[16]) {
            const int t = uniform(tw + qi);
            const int64_t row = ((int64_t)b * P.S + t) * P.G + g;
            select_topn_row_auto<16>(SP, SP.p_grp + row * (int64_t)SP.S_sel, SP.t0 + t, SP.out + row * (int64_t)SP.W * 2, sc, src);
        };
        float va[16], vb[16];
        if (nq > 0) fetch(0, va);
        for (int qi = 0; qi < nq; qi += 2) {
            fetch(qi + 1, vb);
            select(qi, va);
            if (qi + 1 < nq) {
                fetch(qi + 2, va);
                select(qi + 1, vb);
            }
        }
    }
}

// ---- host -----------------------------------------------------------------------------------
bool scores_mfma32_supported(const ScoresMfmaParams &P, int Dk) {
    // 32-bit byte offsets into one sequence's p_grp
    return P.h == 6 && Dk == 64 && !P.big_out && (int64_t)P.S * P.G * P.S_sel * 4 < ((int64_t)1 << 32);
}

bool scores_mfma32_select_supported(const ScoresMfmaParams &P, int Dk, const SelectParams &SP) {
    return scores_mfma32_supported(P, Dk) && P.S_sel <= 1024 && SP.W >= 1 && SP.W <= 64 && SP.t_rows == nullptr;
}

int launch_scores_mfma32(const ScoresMfmaParams &P, int dtype, hipStream_t st, const SelectParams *sel) {
    dim3 grid((unsigned)((P.S + 63) / 64), (unsigned)(P.B * P.G));
    if (sel) {
        if (dtype == NSA_DT_BF16) hipLaunchKernelGGL((scores_mfma32_kernel<__bf16, true>), grid, dim3(256), 0, st, P, *sel);
        else hipLaunchKernelGGL((scores_mfma32_kernel<_Float16, true>), grid, dim3(256), 0, st, P, *sel);
    } else {
        const SelectParams none{};
        if (dtype == NSA_DT_BF16) hipLaunchKernelGGL((scores_mfma32_kernel<__bf16, false>), grid, dim3(256), 0, st, P, none);
        else hipLaunchKernelGGL((scores_mfma32_kernel<_Float16, false>), grid, dim3(256), 0, st, P, none);
    }
    NSA_LAUNCH_CHECK("scores_mfma32");
    return NSA_OK;
}

}  // namespace nsa
