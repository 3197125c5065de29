// One decode step of the selected branch in ONE launch (round 3): logits -> softmax statistics -> Eq.9 / Eq.10 -> sequential top-n ->
// selection attention, for the default block geometry (l = 2d, l' = 4d = 64: Eq.9 is the 5-tap stencil of SURVEY.md 8(a) A3).
// Replaces, per step: compute_pcmp_all -> map_pcmp_to_pslc_batched -> sum(dim=3) -> select_topn_ranges -> the selection executor
// (nsa/core/nsa_attention.py:651-672, 704-830; nsa/core/selection_scorer.py:42-61, 89-116, 124-249).
//
// Why a new kernel (profiles/r02/m_decode_timeline.txt): the round-2 kernel spent 6.5 of its 20 us (B = 64, 16k context) in a chain of
// on-chip phases between its two memory phases -- logits -> LDS -> barrier -> statistics -> barrier -> taps -> barrier -> head sums ->
// barrier -> top-n -> ranges -> barrier -> segment table / scan -> gather -- with HBM idle all the while, at one workgroup per CU.  Here:
//   * the logits never leave the registers: the MFMA accumulators of a wave hold 4 consecutive compressed rows = ONE selection block
//     per lane group, so Eq.9 is in-lane except one lane rotation and Eq.10 is a DPP row-shift add (the prefill scorer's trick,
//     sel_scores_mfma.hip); after ONE barrier (the per-chunk (max, sum) records and the halo logits of the chunk edges through LDS)
//     every wave forms the per-head log-sum-exp for itself -- in the butterfly order of wave_max / wave_sum, so p_grp has the bits of the
//     three-kernel route (decode_logits_mfma_kernel -> decode_pgrp_kernel) -- and writes its blocks' group scores;
//   * the selector leaves the picked BLOCKS in LDS (sel_select_row.hpp), wave e mod NW gathers block e: no range -> segment -> chunk
//     translation on the critical path; the merged ranges (the path's second result) are stored while the gather is in flight;
//   * NW = 8 (two workgroups per CU) from 257 rows on: one row's chain overlaps another row's memory phases (sel_attn_decode.hpp);
//   * long contexts / few rows: the K_cmp sweep of a row is SPLIT over NS workgroups (all on one XCD: they meet in its L2); each leaves
//     its scaled logits and chunk records in the workspace and takes a ticket, the LAST one to arrive carries on with the row (no
//     workgroup ever waits for another: no co-residency assumption, nothing to dead-lock) -- early rows reach their chain and gather
//     while later rows still sweep, which is the overlap the single-workgroup form cannot have at 64k.
// Results: ranges identical, O bit-identical to the three separate launches (tests: test_fused_decode_step,
// test_fused_decode_scorer_equals_three_kernel_route).
#include <mutex>
#include <unordered_map>

#define DEC_TS(i)  // (the shared device functions' own stamps belong to the round-2 kernel's timeline: this kernel stamps with DS_TS)
#include "nsa_internal.hpp"
#include "sel_attn_decode.hpp"
#include "sel_attn_params.hpp"
#include "sel_select_row.hpp"
#include "band_attn_fwd_body.hpp"

namespace nsa {

#ifdef NSA_DEC_TS  // make TIMELINE=1: s_memrealtime stamps of the workgroup(s) of the middle row, read back by nsa_debug_read_ts2 (tools/decode_timeline.py)
static __device__ long long g_ts2[32];
#define DS_TS(i) do { if (ts_on && threadIdx.x == 0) g_ts2[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DS_TS(i)
#endif

struct DecStepParams {
    const void *Q;   // [R,h,64]
    const void *Kc;  // [B,G,S_cmp,64] strided
    float *part_g;   // SPLIT: [R][h][64][2] per-chunk (max, sum exp2)
    float *halo_g;   // SPLIT: [R][64][16] scaled logit of every chunk's last row, per head
    float *pg_g;     // SPLIT: [R][2048] group scores of the row's blocks
    int *cnt;        // SPLIT: [2][R] arrivals (records published) and tickets (scores published); zero between launches
    int R, G, h, S_cmp, S_sel, NS, nchunk, cpg, t_token, spin;
    int64_t csb, csg, css;
    float c2;
};

constexpr int DSTEP_CH = 128;                  // chunks of 64 compressed rows per row of the step: contexts up to 128k tokens (S_cmp <= 8192)
constexpr int DSTEP_PART = 16 * DSTEP_CH * 2;  // floats: [head][chunk][2]
constexpr int DSTEP_HALO = DSTEP_CH * 16;      // floats: [chunk][head] scaled logit of the chunk's last row
constexpr int DSTEP_TAIL = 1024 + 4 * (128 + 68 + 4);  // bytes behind the V tiles: mlw [16][16] f32 | scr | list | misc
static size_t dstep_lds(int nw) { return (size_t)nw * DEC_ATT_TILE + DSTEP_TAIL + (nw == 16 ? 3 * DEC_ATT_TILE : 0); }

// workgroup barrier for data exchanged through LDS only: waits for this wave's LDS operations, NOT for its vector memory operations
// (__syncthreads() drains vmcnt too, which would stall the three prefetching waves -- and with them the whole workgroup -- until their
// forced blocks have landed, moving that traffic back onto the critical path)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Workgroups behind the step's own (BP.n_sel ...) carry the sliding and the compressed branch of a layer step (nsa_layer_decode_step): the two
// are independent of the selected branch, and as a launch of their own they cost the step ~7 us of kernel plus a launch gap on a chain of
// five dependent launches.  Every wave of such a workgroup is one (row, key split) unit of band_attn_body's split form (wave-private LDS
// in the V tile of the wave, no workgroup barrier); they are dispatched behind the step's own workgroups, so a team of the split form finds
// its members resident as before.
template <typename T, int NW>
__device__ __forceinline__ bool decode_band_workgroup(const DecBandPair &BP) {
    if (blockIdx.x < BP.n_sel) return false;
    const unsigned bb = blockIdx.x - BP.n_sel;
    if (bb < BP.n_w) band_attn_body<T, 64, 1, true, 1>(BP.w, bb, NW, &BP.mg, true);
    else band_attn_body<T, 64, 1, true, 1>(BP.c, bb - BP.n_w, NW, &BP.mg, false);
    return true;
}

// CPW = chunks of 64 compressed rows per wave.  2: the logits of both stay in the MFMA accumulators (contexts to 32 NW chunks).  4 (round 4,
// unsplit only): a row of up to 4 NW chunks in ONE workgroup -- 64k contexts on 16 waves, 32k on 8 -- for batches whose rows times a team's
// workgroups do not fit the chip together (B >= 128 at 64k: a team must be co-resident, R NS <= slots): the first chunk's logits stay in
// registers, those of the later three wait in LDS ([slot][u][head][q] f32x4, 256 h bytes per chunk: each lane reads back exactly what it
// wrote, so no barrier guards them) until the row's log-sum-exp is known.  Two chunks (16 KiB per wave) are in flight throughout.  Same
// arithmetic on the same values: p_grp, ranges and O have the bits of every other form.
template <typename T, int NW, bool SPLIT, int HC, int CPW = 2>
__global__ __launch_bounds__(NW * 64, 4) void decode_step_kernel(DecStepParams P, SelectParams SP, int cand, DecAttnArgs AT, DecBandPair BP) {
    static_assert(CPW == 2 || (CPW == 4 && !SPLIT), "four chunks per wave: unsplit form only");
    if (decode_band_workgroup<T, NW>(BP)) return;
    using M = MfmaT<T>;
    using x8 = typename M::x8;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // the score phases live where the V tiles of the gather will be: part | halo | pg.  PREF (16 waves, unsplit): waves 0, 14 and 15 fetch
    // the row's forced blocks (0, t//64 - 1, t//64: known from t alone, selection_scorer.py:159-170) at kernel start, V into their own
    // tiles, K into three tiles behind the tail -- so the score data starts at tile 1
    constexpr bool PREF = NW == 16 && !SPLIT;
    float *part = (float *)(lds + (PREF ? DEC_ATT_TILE : 0));
    float *halo = part + DSTEP_PART;
    float *pg = halo + DSTEP_HALO;  // [S_sel]
    float *mlw = (float *)(lds + NW * DEC_ATT_TILE);  // [NW][16] per-wave copy of the per-head log-sum-exp
    int *scr = (int *)(mlw + NW * 16);                // [128] run extraction of the selector
    int *list = scr + 128;                            // [68] picked blocks, ascending
    int *misc = list + 68;                            // [0] number of picked blocks, [1] ticket, [2] team assembled
    [[maybe_unused]] unsigned char *ktiles = lds + NW * DEC_ATT_TILE + DSTEP_TAIL;  // PREF: [3] K images of the prefetched blocks

    const int lane = lane_id(), wave = uniform((int)(threadIdx.x >> 6)), rho = lane & 15, q = lane >> 4;
    const int h = P.h;
    auto xl_slot = [&](int k) -> float * {  // CPW = 4: [3 NW slots][4][h][4] f32x4 logits of the chunks beyond a wave's first, behind pg
        return pg + ((P.S_sel + 3) & ~3) + ((size_t)((k - 1) * NW + wave) * 4 * h) * 16;
    };
    int row = blockIdx.x, sp = 0;
    if constexpr (SPLIT) {  // the NS workgroups of a row sit on one XCD (workgroups go round-robin over the 8 XCDs)
        const int per = 8 * P.NS, grp = blockIdx.x / per, rem = blockIdx.x - grp * per;
        row = grp * 8 + (rem & 7);
        sp = rem >> 3;
        if (row >= P.R) return;
    }
    [[maybe_unused]] const bool ts_on = row == P.R / 2;
    DS_TS(0);
    const int g = row % P.G;
    const int64_t b = row / P.G;
    const int hc = min(rho, h - 1);  // head of this lane's column (columns >= h repeat the last head: their results are never used)
    // ---- phase 1: logits of this workgroup's chunks (64 compressed rows each = MFMA rows; heads = columns), up to two per wave,
    // all loads out at once
    x8 qf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) qf[s] = *(const x8 *)((const T *)P.Q + ((int64_t)row * h + hc) * 64 + 32 * s + 8 * q);
    const T *kb = (const T *)P.Kc + b * P.csb + (int64_t)g * P.csg;
    const int c_lo = SPLIT ? sp * P.cpg : 0, c_hi = SPLIT ? min(P.nchunk, c_lo + P.cpg) : P.nchunk;
    constexpr int REGC = CPW == 2 ? 2 : 1;  // chunks of a wave whose logits stay in registers
    x8 a[2][4][2];
    f32x4 acc[REGC][4];
    auto load_chunk = [&](int c, x8 (&dst)[4][2]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = min(c * 64 + 16 * u + rho, P.S_cmp - 1);
#pragma unroll
            for (int s = 0; s < 2; ++s) dst[u][s] = *(const x8 *)(kb + (int64_t)r * P.css + 32 * s + 8 * q);
        }
    };
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = c_lo + wave + NW * k;
        if (c < c_hi) load_chunk(c, a[k]);
    }
    // (behind the K_cmp loads in program order: vector memory operations complete in order, the scores must not wait for these)
    DecPrefetch pre{-1, nullptr};
    if constexpr (PREF) {
        const int cb = P.t_token >> 6, t_end = min(P.t_token + 1, AT.S_kv);
        const int slot = wave == 0 ? 0 : wave - 13;  // waves 14, 15 -> K tiles 1, 2
        if ((wave == 0 || wave >= 14) && cb >= 2 && cb < P.S_sel && AT.kss == 64) {
            const int blk = wave == 0 ? 0 : cb - (15 - wave);
            pre.tok0 = 64 * blk;
            pre.ktile = ktiles + slot * DEC_ATT_TILE;
            decode_prefetch_chunk<T>(AT, row, pre.tok0, min(64, t_end - pre.tok0), lds + wave * DEC_ATT_TILE, ktiles + slot * DEC_ATT_TILE);
        }
    }

    for (int i = threadIdx.x; i < P.S_sel; i += NW * 64) pg[i] = 0.f;  // blocks without a compressed row keep a zero score
    // scaled logits of chunk c (MFMA rows = 64 compressed rows, columns = heads) and the chunk's (max, sum exp2) per head: the arithmetic of
    // decode_logits_mfma_kernel
    auto chunk_logits = [&](int c, const x8 (&src)[4][2], f32x4 (&z)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f32x4 zz = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) zz = M::mma(src[u][s], qf[s], zz);
            z[u] = zz;
        }
    };
    auto chunk_stats = [&](int c, f32x4 (&z)[4]) {
        float m = -INFINITY;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = z[u][j] * P.c2;
                z[u][j] = v;
                if (c * 64 + 16 * u + 4 * q + j < P.S_cmp) m = fmaxf(m, v);
            }
        m = xor32_max(xor16_max(m));  // (= the xor-16, xor-32 shuffle steps of decode_logits_mfma_kernel: same bits)
        float l = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c * 64 + 16 * u + 4 * q + j < P.S_cmp) l += __builtin_amdgcn_exp2f(z[u][j] - m);
        l = xor32_add(xor16_add(l));
        if constexpr (SPLIT) {
            if (rho < h) {  // write-through (sc1) stores: no release fence needed (cdna guide, Guideline 16 R1)
                if (q == 0)
                    __hip_atomic_store((unsigned long long *)(P.part_g + (((int64_t)row * h + rho) * DSTEP_CH + c) * 2),
                                       ((unsigned long long)__float_as_uint(l) << 32) | __float_as_uint(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (q == 3)
                    __hip_atomic_store((unsigned *)(P.halo_g + ((int64_t)row * DSTEP_CH + c) * 16 + rho), __float_as_uint(z[3][3]), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            if (rho < h) {
                if (q == 0) *(f32x2 *)(part + (rho * DSTEP_CH + c) * 2) = (f32x2){m, l};
                if (q == 3) halo[c * 16 + rho] = z[3][3];  // row 64 c + 63: the half tap of the next chunk's first block
            }
        }
    };
    if constexpr (CPW == 2) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int c = c_lo + wave + NW * k;
            if (c < c_hi) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 2; ++s) z = M::mma(a[k][u][s], qf[s], z);
                    acc[k][u] = z;
                }
                chunk_stats(c, acc[k]);
            }
        }
    } else {
        // chunk 0 -> registers, chunks 1 .. 3 -> LDS; the loads of chunk k + 2 go out as soon as the MFMAs of chunk k have read its fragments
#pragma unroll
        for (int k = 0; k < CPW; ++k) {
            const int c = wave + NW * k;
            if (c < c_hi) {
                f32x4 z[4];
                chunk_logits(c, a[k & 1], z);
                if (k + 2 < CPW && c + 2 * NW < c_hi) load_chunk(c + 2 * NW, a[k & 1]);
                chunk_stats(c, z);
                if (k == 0) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[0][u] = z[u];
                } else if (rho < h) {
                    float *slot = xl_slot(k);
#pragma unroll
                    for (int u = 0; u < 4; ++u) *(f32x4 *)(slot + ((u * h + rho) * 4 + q) * 4) = z[u];
                }
            }
        }
    }
    DS_TS(1);
    if constexpr (SPLIT) {
        // The NS workgroups of the row meet ONCE: each has published the (max, sum) records and the edge logits of its chunks -- a few
        // hundred bytes, write-through (sc1) stores, every storing wave drained, then ONE lane's agent-scope add behind the workgroup
        // barrier (the hand-off form of the CDNA guide, Guideline 16 R1) -- and waits until all NS have (one lane polls with sc1 loads and
        // s_sleep; the others wait at the barrier it then joins).  The logits themselves never leave the registers: with the row's
        // log-sum-exp every workgroup scores its OWN blocks.  (First form of this kernel: the last arriver re-read all logits from the
        // workspace -- 98 KB per row at 64k, 3.3 us at the ~30 GB/s one CU pulls from memory, and scored all 64 chunks alone.)
        // The wait pays off while the row's workgroups are resident together: the host launches this form only while R NS workgroups fit
        // the chip at once.  It is bounded (P.spin polls), and a workgroup whose team does not assemble in time does the missing work itself
        // (below): no workgroup ever depends on another one being scheduled.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        DS_TS(2);
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(P.cnt + row, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int it = 0;
            while (__hip_atomic_load(P.cnt + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < P.NS && it < P.spin) {
                __builtin_amdgcn_s_sleep(2);
                ++it;
            }
            misc[2] = __hip_atomic_load(P.cnt + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= P.NS ? 1 : 0;
        }
        __syncthreads();
        DS_TS(3);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // (no instruction: keeps the compiler from moving the loads below above this point)
        if (misc[2]) {
            // the row's records and edge logits -> LDS, one sc1 load per thread
            for (int i = threadIdx.x; i < h * DSTEP_CH; i += NW * 64) {
                const int hh = i / DSTEP_CH, idx = i % DSTEP_CH;
                if (idx < P.nchunk) {
                    const unsigned long long r = __hip_atomic_load((const unsigned long long *)(P.part_g + (((int64_t)row * h + hh) * DSTEP_CH + idx) * 2), __ATOMIC_RELAXED,
                                                                   __HIP_MEMORY_SCOPE_AGENT);
                    *(f32x2 *)(part + (hh * DSTEP_CH + idx) * 2) = (f32x2){__uint_as_float((unsigned)r), __uint_as_float((unsigned)(r >> 32))};
                }
            }
            for (int i = threadIdx.x; i < P.nchunk * 16; i += NW * 64)
                if ((i & 15) < h) halo[i] = __uint_as_float(__hip_atomic_load((const unsigned *)(P.halo_g + (int64_t)row * DSTEP_CH * 16 + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        } else {
            // The team did not assemble within the poll budget (its other workgroups are not resident yet: a second stream or process holds
            // the CUs).  Nobody waits for anybody here: this workgroup forms the records and edge logits of ALL chunks of the row itself --
            // the same instructions on the same data, so the same bits as its team mates publish -- and carries on.  Slower, never stuck.
            for (int c = wave; c < P.nchunk; c += NW) {
                x8 b2[4][2];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = min(c * 64 + 16 * u + rho, P.S_cmp - 1);
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) b2[u][s2] = *(const x8 *)(kb + (int64_t)r * P.css + 32 * s2 + 8 * q);
                }
                f32x4 z[4];
                float m = -INFINITY;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    z[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) z[u] = M::mma(b2[u][s2], qf[s2], z[u]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = z[u][j] * P.c2;
                        z[u][j] = v;
                        if (c * 64 + 16 * u + 4 * q + j < P.S_cmp) m = fmaxf(m, v);
                    }
                }
                m = xor32_max(xor16_max(m));
                float l = 0.f;
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (c * 64 + 16 * u + 4 * q + j < P.S_cmp) l += __builtin_amdgcn_exp2f(z[u][j] - m);
                l = xor32_add(xor16_add(l));
                if (rho < h) {
                    if (q == 0) *(f32x2 *)(part + (rho * DSTEP_CH + c) * 2) = (f32x2){m, l};
                    if (q == 3) halo[c * 16 + rho] = z[3][3];
                }
            }
        }
        __syncthreads();
        DS_TS(4);
    } else {
        lds_barrier();
        DS_TS(4);
    }

    // ---- phase 2a: per-head log-sum-exp of the row's logits from the chunk records, by every wave for itself.  One 16-lane row of the
    // wave per head, lane i holds records i, i+16, i+32, i+48: ((a_i + a_{i+32}) + (a_{i+16} + a_{i+48})) then xor 8, 4, 2, 1 is the
    // summation tree of wave_sum over 64 lanes (decode_pgrp_kernel / the round-2 fused kernel): same bits.
    const int nr = (P.nchunk + 15) >> 4;  // records per lane that exist at all (a context of 16k has 16 chunks: one; 128k: eight)
    for (int h0 = 0; h0 < h; h0 += 4) {
        const int hh = h0 + q;
        float mv[8], lv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = rho + 16 * i;
            mv[i] = -INFINITY;
            lv[i] = 0.f;
            if (i < nr && hh < h && idx < P.nchunk) {
                const f32x2 r = *(const f32x2 *)(part + (hh * DSTEP_CH + idx) * 2);
                mv[i] = r[0];
                lv[i] = r[1];
            }
        }
        float m = fmaxf(fmaxf(fmaxf(mv[0], mv[1]), fmaxf(mv[2], mv[3])), fmaxf(fmaxf(mv[4], mv[5]), fmaxf(mv[6], mv[7])));
        m = fmaxf(m, row_ror<8>(m));  // butterfly over the 16 lanes of the row by DPP rotations (nsa_common.hpp: no LDS crossbar on the chain)
        m = fmaxf(m, row_ror<4>(m));
        m = fmaxf(m, row_ror<2>(m));
        m = fmaxf(m, row_ror<1>(m));
        float av[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // (a record that does not exist adds +0: exact, so skipping its exponential changes nothing)
            av[i] = (i < nr && rho + 16 * i < P.nchunk) ? lv[i] * __builtin_amdgcn_exp2f(mv[i] - m) : 0.f;
            // beyond 64 chunks the three-kernel route lets lane L add records L and L + 64 before the butterfly: the same sum here
            if (i + 4 < nr && rho + 16 * (i + 4) < P.nchunk) av[i] += lv[i + 4] * __builtin_amdgcn_exp2f(mv[i + 4] - m);
        }
        float s = (av[0] + av[2]) + (av[1] + av[3]);
        s += row_ror<8>(s);
        s += row_ror<4>(s);
        s += row_ror<2>(s);
        s += row_ror<1>(s);
        if (rho == 0 && hh < h) mlw[wave * 16 + hh] = m + __builtin_amdgcn_logf(s);
    }
    wave_lds_fence();
    const float ml = mlw[wave * 16 + hc];
    DS_TS(5);

    // ---- phase 2b: p = exp2(x - ml), Eq.9 stencil (1/2, 1, 1, 1, 1/2 over rows 4j-1 .. 4j+3, ascending), Eq.10 head sum (ascending h)
    auto blocks_of_chunk = [&](int c, const f32x4 (&v)[4], float hprev) {
        float rot_prev = c > 0 ? __builtin_amdgcn_exp2f(hprev - ml) : 0.f;  // p of row 64 c - 1 (a chunk that exists has its predecessor complete)
        const bool tail = c * 64 + 64 > P.S_cmp;  // only the row's last chunk can reach past S_cmp (wave uniform): the others need no row masks
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float p[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) p[k] = __builtin_amdgcn_exp2f(v[u][k] - ml);
            if (tail) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (c * 64 + 16 * u + 4 * q + k >= P.S_cmp) p[k] = 0.f;
            }
            const float rot = __shfl(p[3], (lane + 48) & 63, 64);  // row 4j - 1 = last row of the previous lane group (previous sub-tile for q = 0)
            const float tapm1 = (q == 0) ? rot_prev : rot;
            rot_prev = rot;
            float slc = fmaf(0.5f, tapm1, p[0]);  // (0.5 x is exact: the fma rounds once, like the separate product and sum of the reference order)
            slc += p[1];
            slc += p[2];
            slc = fmaf(0.5f, p[3], slc);
            // heads of the group = columns 0 .. h-1 of this 16-lane row: lane 0 collects them in ascending order by DPP row shifts.  HC = the
            // group size at compile time (straight-line code); HC = 0: any h <= 16 -- columns >= h enter as zeros (p >= 0: adding +0 is exact),
            // so the 15 shifted adds run unconditionally (a scalar branch per possible head cost 2 us per step at 64k)
            float grp;
#define NSA_DS_HS(K) grp += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, src), 0x100 | K, 0xf, 0xf, true));
            if constexpr (HC == 6) {
                const float src = slc;
                grp = slc;
                NSA_DS_HS(1) NSA_DS_HS(2) NSA_DS_HS(3) NSA_DS_HS(4) NSA_DS_HS(5)
            } else {
                const float src = rho < h ? slc : 0.f;
                grp = src;
                NSA_DS_HS(1) NSA_DS_HS(2) NSA_DS_HS(3) NSA_DS_HS(4) NSA_DS_HS(5) NSA_DS_HS(6) NSA_DS_HS(7) NSA_DS_HS(8)
                NSA_DS_HS(9) NSA_DS_HS(10) NSA_DS_HS(11) NSA_DS_HS(12) NSA_DS_HS(13) NSA_DS_HS(14) NSA_DS_HS(15)
            }
#undef NSA_DS_HS
            const int j = 16 * c + 4 * u + q;
            if (rho == 0 && j < P.S_sel) {
                if constexpr (SPLIT) __hip_atomic_store((unsigned *)(P.pg_g + (int64_t)row * 2048 + j), __float_as_uint(grp), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else pg[j] = grp;
            }
        }
    };
#pragma unroll
    for (int k = 0; k < CPW; ++k) {
        const int c = c_lo + wave + NW * k;
        if (c < c_hi) {
            if (k < REGC) {
                blocks_of_chunk(c, acc[k], c > 0 ? halo[(c - 1) * 16 + hc] : 0.f);
            } else {
                f32x4 z[4];  // lanes of columns >= h read the last head's values, as their accumulators would hold them (never used)
                const float *slot = xl_slot(k);
#pragma unroll
                for (int u = 0; u < 4; ++u) z[u] = *(const f32x4 *)(slot + ((u * h + hc) * 4 + q) * 4);
                blocks_of_chunk(c, z, halo[(c - 1) * 16 + hc]);
            }
        }
    }
    if constexpr (SPLIT) {
        // second meeting, nobody waits: the scores of this workgroup's blocks are published (sc1, drained), one lane takes a ticket; the
        // workgroup that draws the last one reads the row's scores (<= 4 KB) and carries on with the row, the others are done
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const int tk = __hip_atomic_fetch_add(P.cnt + P.R + row, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tk == P.NS - 1) {  // every workgroup of the row is past the first meeting: both words go back to zero for the next launch
                __hip_atomic_store(P.cnt + row, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(P.cnt + P.R + row, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            misc[1] = tk;
        }
        __syncthreads();
        if (misc[1] != P.NS - 1) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int j = threadIdx.x; j < min(P.S_sel, 16 * P.nchunk); j += NW * 64)
            pg[j] = __uint_as_float(__hip_atomic_load((const unsigned *)(P.pg_g + (int64_t)row * 2048 + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    DS_TS(6);
    lds_barrier();
    DS_TS(7);

    // ---- phase 3: top-n + forced blocks (one wave); the picked blocks go to LDS for the gather, the merged ranges to the caller.
    // (Tried and dropped, profiles/r03/decode_notes.txt: the selection as a rank problem over the whole workgroup -- every block's rank by
    // all-pairs compares of 64-bit (key, ~index) images, three short phases instead of the threshold search's dependent rounds.  All-pairs
    // over 256 slots is 65,536 compares = ~1 us of VALU issue on one CU even at full rate, and v_cmp_gt_u64 is not full rate: 2.1 + 0.7 us
    // at 16k context against 2.8 us for this wave's search, 5.9 against 4.5 us at 64k.)
    int rs = 0, re = 0;
    if (wave == 0) {
        int nb = 0;
        switch (cand) {
            case 1: select_topn_row_regs<1>(SP, pg, P.t_token, rs, re, scr, list, &nb); break;
            case 2: select_topn_row_regs<2>(SP, pg, P.t_token, rs, re, scr, list, &nb); break;
            case 4: select_topn_row_regs<4>(SP, pg, P.t_token, rs, re, scr, list, &nb); break;
            case 8: select_topn_row_regs<8>(SP, pg, P.t_token, rs, re, scr, list, &nb); break;
            case 16: select_topn_row_regs<16>(SP, pg, P.t_token, rs, re, scr, list, &nb); break;
            default: select_topn_row_regs<32, true>(SP, pg, P.t_token, rs, re, scr, list, &nb); break;
        }
        if (lane == 0) misc[0] = nb;
    }
    DS_TS(8);
    lds_barrier();  // the scores are dead from here on: their space holds the V tiles
    if (wave == 0 && lane < SP.W) {
        int32_t *out = SP.out + (int64_t)row * SP.W * 2;
        out[2 * lane] = rs;
        out[2 * lane + 1] = re;
    }
    const int NC = uniform(misc[0]);
    // ---- phase 4: selection attention over the picked blocks (wave e mod NW gathers block e; partials merged through LDS)
    const ListChunks ch{list, min(P.t_token + 1, AT.S_kv)};
    DS_TS(9);
    decode_attend_chunks<T, NW>(AT, row, ch, NC, lds, qf, pre);
    DS_TS(10);
}

// ---- one-pass form (round 4): many rows at a long context ------------------------------------------------------------------------
// B >= 128 sequences at 64k: a row's 64 chunks neither stay in the accumulators of one workgroup (2 per wave) nor can R teams of
// workgroups be resident together, and one 16-wave workgroup per CU with the later logits in LDS (CPW = 4 above) leaves every CU's
// chain -- log-sum-exp, scores, top-n: ~11 us of 50 per row, HBM idle -- exposed, all CUs in step (profiles/r04/decode_cold_notes.txt).
// Here a row is ONE 8-wave workgroup again (two rows per CU: one row's chain under the other's sweep / gather) at up to 8 chunks per wave:
// what a wave keeps of a chunk is not its 16 logits per lane but the Eq.9 sums of their exponentials RELATIVE TO THE CHUNK'S OWN MAXIMUM,
//     E[h, j] = 1/2 e(4j-1) + e(4j) + e(4j+1) + e(4j+2) + 1/2 e(4j+3),   e(i) = exp2(x[h, i] - m_c[h])
// -- 4 registers per chunk, and the exponentials are the ones the chunk's (max, sum) record needs anyway: ONE exponential per logit instead
// of two.  With the row's log-sum-exp ml[h] the score of block j is  sum_h E[h, j] exp2(m_c[h] - ml[h])  (+ the half tap of the previous
// chunk's last row for a chunk's first block, from the edge logits every form publishes).  Same (max, sum) records, same ml, same selector
// and gather as the other forms; the scores carry one more rounding (<= 2 ulp against exp2(x - ml) summed directly), so the RANGES are those
// of the exact forms wherever the 13th / 14th ranking keys are further apart than that -- the contract for scores computed from Q / K
//  -- and exact-tie order is only guaranteed by the exact forms.  The plan picks this form only where they do not apply.
template <typename T, int NW, int HC>
__global__ __launch_bounds__(NW * 64, 4) void decode_step_onepass_kernel(DecStepParams P, SelectParams SP, int cand, DecAttnArgs AT, DecBandPair BP) {
    if (decode_band_workgroup<T, NW>(BP)) return;
    using M = MfmaT<T>;
    using x8 = typename M::x8;
    constexpr int CPW = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr bool PREF = NW == 16;
    float *part = (float *)(lds + (PREF ? DEC_ATT_TILE : 0));
    float *halo = part + DSTEP_PART;
    float *pg = halo + DSTEP_HALO;  // [S_sel]
    float *mlw = (float *)(lds + NW * DEC_ATT_TILE);
    int *scr = (int *)(mlw + NW * 16);
    int *list = scr + 128;
    int *misc = list + 68;
    [[maybe_unused]] unsigned char *ktiles = lds + NW * DEC_ATT_TILE + DSTEP_TAIL;

    const int lane = lane_id(), wave = uniform((int)(threadIdx.x >> 6)), rho = lane & 15, q = lane >> 4;
    const int h = P.h;
    const int row = blockIdx.x;
    [[maybe_unused]] const bool ts_on = row == P.R / 2;
    DS_TS(0);
    const int g = row % P.G;
    const int64_t b = row / P.G;
    const int hc = min(rho, h - 1);
    x8 qf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) qf[s] = *(const x8 *)((const T *)P.Q + ((int64_t)row * h + hc) * 64 + 32 * s + 8 * q);
    const T *kb = (const T *)P.Kc + b * P.csb + (int64_t)g * P.csg;
    const int nchunk = P.nchunk;
    x8 a[2][4][2];
    auto load_chunk = [&](int c, x8 (&dst)[4][2]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = min(c * 64 + 16 * u + rho, P.S_cmp - 1);
#pragma unroll
            for (int s = 0; s < 2; ++s) dst[u][s] = *(const x8 *)(kb + (int64_t)r * P.css + 32 * s + 8 * q);
        }
    };
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (wave + NW * k < nchunk) load_chunk(wave + NW * k, a[k]);
    DecPrefetch pre{-1, nullptr};
    if constexpr (PREF) {
        const int cb = P.t_token >> 6, t_end = min(P.t_token + 1, AT.S_kv);
        const int slot = wave == 0 ? 0 : wave - 13;
        if ((wave == 0 || wave >= 14) && cb >= 2 && cb < P.S_sel && AT.kss == 64) {
            const int blk = wave == 0 ? 0 : cb - (15 - wave);
            pre.tok0 = 64 * blk;
            pre.ktile = ktiles + slot * DEC_ATT_TILE;
            decode_prefetch_chunk<T>(AT, row, pre.tok0, min(64, t_end - pre.tok0), lds + wave * DEC_ATT_TILE, ktiles + slot * DEC_ATT_TILE);
        }
    }
    for (int i = threadIdx.x; i < P.S_sel; i += NW * 64) pg[i] = 0.f;

    // ---- phase 1: per chunk the (max, sum exp2) record, the edge logit, and the Eq.9 sums of the chunk's blocks relative to its maximum
    float E[CPW][4];
#pragma unroll
    for (int k = 0; k < CPW; ++k) {
        const int c = wave + NW * k;
#pragma unroll
        for (int u = 0; u < 4; ++u) E[k][u] = 0.f;
        if (c < nchunk) {
            f32x4 z[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                f32x4 zz = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) zz = M::mma(a[k & 1][u][s], qf[s], zz);
                z[u] = zz;
            }
            if (k + 2 < CPW && c + 2 * NW < nchunk) load_chunk(c + 2 * NW, a[k & 1]);  // two chunks (16 KiB per wave) in flight throughout
            float m = -INFINITY;
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = z[u][j] * P.c2;
                    z[u][j] = v;
                    if (c * 64 + 16 * u + 4 * q + j < P.S_cmp) m = fmaxf(m, v);
                }
            m = xor32_max(xor16_max(m));
            const float edge = z[3][3];
            float l = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float e = (c * 64 + 16 * u + 4 * q + j < P.S_cmp) ? __builtin_amdgcn_exp2f(z[u][j] - m) : 0.f;
                    z[u][j] = e;
                    l += e;  // (a masked row adds +0: exact -- the sum has the bits of the other forms' record)
                }
            l = xor32_add(xor16_add(l));
            if (rho < h) {
                if (q == 0) *(f32x2 *)(part + (rho * DSTEP_CH + c) * 2) = (f32x2){m, l};
                if (q == 3) halo[c * 16 + rho] = edge;
            }
            float rot_prev = 0.f;  // (the half tap of row 64 c - 1 belongs to another chunk's scale: added in phase 2b)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float rot = __shfl(z[u][3], (lane + 48) & 63, 64);
                const float tapm1 = (q == 0) ? rot_prev : rot;
                rot_prev = rot;
                float slc = fmaf(0.5f, tapm1, z[u][0]);
                slc += z[u][1];
                slc += z[u][2];
                E[k][u] = fmaf(0.5f, z[u][3], slc);
            }
        }
    }
    DS_TS(1);
#ifdef DSTEP_SWEEP_ONLY  // ablation build (timing only, results meaningless): the K_cmp sweep alone -- B = 256 @64k cold: 46.7 of 75.5 us
    if (E[0][0] == 12345.f) pg[0] = E[7][3];
    return;
#endif
    lds_barrier();
    DS_TS(4);
    // ---- phase 2a: per-head log-sum-exp from the chunk records (the arithmetic of decode_step_kernel: same bits)
    const int nr = (nchunk + 15) >> 4;
    for (int h0 = 0; h0 < h; h0 += 4) {
        const int hh = h0 + q;
        float mv[8], lv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = rho + 16 * i;
            mv[i] = -INFINITY;
            lv[i] = 0.f;
            if (i < nr && hh < h && idx < nchunk) {
                const f32x2 r = *(const f32x2 *)(part + (hh * DSTEP_CH + idx) * 2);
                mv[i] = r[0];
                lv[i] = r[1];
            }
        }
        float m = fmaxf(fmaxf(fmaxf(mv[0], mv[1]), fmaxf(mv[2], mv[3])), fmaxf(fmaxf(mv[4], mv[5]), fmaxf(mv[6], mv[7])));
        m = fmaxf(m, row_ror<8>(m));
        m = fmaxf(m, row_ror<4>(m));
        m = fmaxf(m, row_ror<2>(m));
        m = fmaxf(m, row_ror<1>(m));
        float av[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            av[i] = (i < nr && rho + 16 * i < nchunk) ? lv[i] * __builtin_amdgcn_exp2f(mv[i] - m) : 0.f;
            if (i + 4 < nr && rho + 16 * (i + 4) < nchunk) av[i] += lv[i + 4] * __builtin_amdgcn_exp2f(mv[i + 4] - m);
        }
        float s = (av[0] + av[2]) + (av[1] + av[3]);
        s += row_ror<8>(s);
        s += row_ror<4>(s);
        s += row_ror<2>(s);
        s += row_ror<1>(s);
        if (rho == 0 && hh < h) mlw[wave * 16 + hh] = m + __builtin_amdgcn_logf(s);
    }
    wave_lds_fence();
    const float ml = mlw[wave * 16 + hc];
    DS_TS(5);
    // ---- phase 2b: scores of the wave's blocks: E scaled to the row's normaliser, the previous chunk's half tap, Eq.10 head sum (ascending h)
#pragma unroll
    for (int k = 0; k < CPW; ++k) {
        const int c = wave + NW * k;
        if (c < nchunk) {
            const float f = __builtin_amdgcn_exp2f(part[(hc * DSTEP_CH + c) * 2] - ml);
            const float hp = c > 0 ? 0.5f * __builtin_amdgcn_exp2f(halo[(c - 1) * 16 + hc] - ml) : 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float slc = E[k][u] * f;
                if (u == 0 && q == 0) slc += hp;
                float grp;
#define NSA_DS_HS(K) grp += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, src), 0x100 | K, 0xf, 0xf, true));
                if constexpr (HC == 6) {
                    const float src = slc;
                    grp = slc;
                    NSA_DS_HS(1) NSA_DS_HS(2) NSA_DS_HS(3) NSA_DS_HS(4) NSA_DS_HS(5)
                } else {
                    const float src = rho < h ? slc : 0.f;
                    grp = src;
                    NSA_DS_HS(1) NSA_DS_HS(2) NSA_DS_HS(3) NSA_DS_HS(4) NSA_DS_HS(5) NSA_DS_HS(6) NSA_DS_HS(7) NSA_DS_HS(8)
                    NSA_DS_HS(9) NSA_DS_HS(10) NSA_DS_HS(11) NSA_DS_HS(12) NSA_DS_HS(13) NSA_DS_HS(14) NSA_DS_HS(15)
                }
#undef NSA_DS_HS
                const int j = 16 * c + 4 * u + q;
                if (rho == 0 && j < P.S_sel) pg[j] = grp;
            }
        }
    }
    DS_TS(6);
    lds_barrier();
    DS_TS(7);
    // ---- phase 3 / 4: top-n + forced blocks (one wave), then the gather over the picked blocks -- decode_step_kernel's
    int rs = 0, re = 0;
    if (wave == 0) {
        int nb = 0;
        switch (cand) {
            case 1: select_topn_row_regs<1>(SP, pg, P.t_token, rs, re, scr, list, &nb); break;
            case 2: select_topn_row_regs<2>(SP, pg, P.t_token, rs, re, scr, list, &nb); break;
            case 4: select_topn_row_regs<4>(SP, pg, P.t_token, rs, re, scr, list, &nb); break;
            case 8: select_topn_row_regs<8>(SP, pg, P.t_token, rs, re, scr, list, &nb); break;
            case 16: select_topn_row_regs<16>(SP, pg, P.t_token, rs, re, scr, list, &nb); break;
            default: select_topn_row_regs<32, true>(SP, pg, P.t_token, rs, re, scr, list, &nb); break;
        }
        if (lane == 0) misc[0] = nb;
    }
    DS_TS(8);
    lds_barrier();
    if (wave == 0 && lane < SP.W) {
        int32_t *out = SP.out + (int64_t)row * SP.W * 2;
        out[2 * lane] = rs;
        out[2 * lane + 1] = re;
    }
    const int NC = uniform(misc[0]);
    const ListChunks ch{list, min(P.t_token + 1, AT.S_kv)};
    DS_TS(9);
    decode_attend_chunks<T, NW>(AT, row, ch, NC, lds, qf, pre);
    DS_TS(10);
}

// ---- host ------------------------------------------------------------------------------------------------------------------------
// arrival tickets of the split form: owned by the library, one zeroed array per (device, stream) -- launches of one stream are ordered
// and every launch leaves its tickets at zero, launches of different streams never share an array
static int *decode_tickets(hipStream_t st, int64_t rows) {
    static std::mutex mu;
    static std::unordered_map<uint64_t, std::pair<int *, int64_t>> tab;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    const uint64_t key = ((uint64_t)(uintptr_t)st << 8) ^ (uint64_t)dev;
    std::lock_guard<std::mutex> lk(mu);
    auto it = tab.find(key);
    if (it != tab.end() && it->second.second >= rows) return it->second.first;
    const int64_t n = rows < 4096 ? 4096 : rows * 2;
    int *p = nullptr;
    if (hipMalloc(&p, sizeof(int) * n) != hipSuccess) return nullptr;
    // zeroed ON THE LAUNCH STREAM: the fill is ordered before the first kernel that reads the tickets whatever kind of stream `st` is (a plain
    // hipMemset runs on the null stream, which a non-blocking stream does not wait for).  Once per stream; an outgrown array stays
    // allocated -- a launch may still use it
    if (hipMemsetAsync(p, 0, sizeof(int) * n, st) != hipSuccess) {
        (void)hipFree(p);
        return nullptr;
    }
    tab[key] = {p, n};
    return p;
}

size_t decode_step_workspace(int64_t R, int h, int S_cmp) {  // split form: chunk records, edge logits, group scores of every row
    (void)S_cmp;
    return sizeof(float) * (size_t)R * ((size_t)h * DSTEP_CH * 2 + DSTEP_CH * 16 + 2048);
}

static int device_cu_count() {
    static std::mutex mu;
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    std::lock_guard<std::mutex> lk(mu);
    if (cus[dev] == 0) {
        int n = 0;
        cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cus[dev];
}

// bytes of score data the unsplit kernel keeps in the V-tile area, and what that area holds
static size_t dstep_score_bytes(int nw, int h, int S_sel, int cpw) {
    return sizeof(float) * (DSTEP_PART + DSTEP_HALO + (size_t)((S_sel + 3) & ~3)) + (cpw == 4 ? (size_t)3 * nw * 256 * h : 0);
}
static size_t dstep_score_room(int nw) { return (size_t)(nw == 16 ? 13 : nw) * DEC_ATT_TILE; }

// waves per row workgroup, workgroups per row (1 = an unsplit kernel) and the kernel form; false = this shape is not for the one-launch step.
// form 0: logits in the accumulators (two chunks per wave; a long row as a team of NS workgroups that meet inside the launch: R * NS of them
// must be resident together); 1: one 16- or 8-wave workgroup per row with four chunks per wave, the later chunks' logits in LDS (exact: the
// bits of form 0); 2: the one-pass form, eight chunks per wave (decode_step_onepass_kernel: scores within 2 ulp of the exact forms).
// TUNE_DECODE_WIDE: -1 = form 2 where a team would not fit the chip (measured cold, profiles/r04/decode_cold_forms.txt: B=256@64k 75.5 us
// against 98.4 for form 1 and 98.1 for the round-2 two-launch route; B=128@64k 47.4 / 52.3 / 56.8), 0 = never, 1 / 2 = that form wherever
// the row fits it.
static bool decode_step_plan(int64_t R, int nchunk, int h, int S_sel, int *nw_out, int *ns_out, int *form_out) {
    const int nw = dec_att_waves(R);
    const int64_t slots = (int64_t)device_cu_count() * (nw == 16 ? 1 : 2);  // 1024-thread workgroups hold a CU each (LDS), 512-thread ones share it
    const int mode = tuning(TUNE_DECODE_SPLIT), wide = tuning(TUNE_DECODE_WIDE);
    int ns = (nchunk + 2 * nw - 1) / (2 * nw);  // at most two chunks per wave (the accumulators of a wave's chunks stay in registers)
    const bool wide_fits = nchunk > 2 * nw && nchunk <= 4 * nw && dstep_score_bytes(nw, h, S_sel, 4) <= dstep_score_room(nw);
    const bool onepass_fits = nchunk <= 8 * nw && dstep_score_bytes(nw, h, S_sel, 2) <= dstep_score_room(nw);
    const bool no_team = ns > 1 && R * ns > slots;
    *nw_out = nw;
    *ns_out = 1;
    if (wide_fits && wide == 1) {
        *form_out = 1;
        return true;
    }
    if (onepass_fits && (wide == 2 || (wide < 0 && no_team))) {
        *form_out = 2;
        return true;
    }
    if (no_team) return false;
    if (mode > 0) {
        int want = mode < ns ? ns : (mode > nchunk ? nchunk : mode);
        while (want > ns && R * want > slots) --want;
        ns = want;
    } else {
        // measured rule (profiles/r03/decode_sweep_split_grid.txt).  A team costs two hand-offs (~2.5 us): a row that fits one workgroup
        // (<= 2 chunks per wave) is only split when every workgroup then gets 8 chunks (64 KiB: contexts from 32k on, few rows) -- at 16k
        // the unsplit form wins at every batch size, at 32k with 128 rows too.  A row that must be split anyway gets one chunk per wave
        // if the chip holds that many workgroups.
        const int ns1 = (nchunk + nw - 1) / nw;  // one chunk per wave
        if (ns >= 2 && ns1 > ns && R * ns1 <= slots) ns = ns1;
        const int t8 = nchunk / 8;  // workgroups of 8 chunks
        if (nchunk >= 32 && t8 > ns && R * t8 <= slots) ns = t8;
    }
    *ns_out = ns;
    *form_out = 0;
    return true;
}

bool decode_step_supported(int64_t R, int dtype, int h, int Dk, int Dv, int S_cmp, int S_sel, int S_kv, int l, int d, int l_sel, int n_top, int t_token,
                           int64_t kcb, int64_t kcg, int64_t kcs, int64_t ksb, int64_t ksg, int64_t kss, int64_t vsb, int64_t vsg, int64_t vss,
                           const void *Q, const void *Kc, const void *K, const void *V) {
    if (tuning(TUNE_DECODE_STEP) == 0 || tuning(TUNE_DECODE_UNFUSED) > 0) return false;
    int nw, ns, form;
    if (R < 1 || S_cmp < 1 || S_sel < 1 || h < 1 || h > 16 || !decode_step_plan(R, (S_cmp + 63) / 64, h, S_sel, &nw, &ns, &form)) return false;
    return (dtype == NSA_DT_BF16 || dtype == NSA_DT_F16) && Dk == 64 && Dv == 64 && h >= 1 && h <= 16 && d > 0 && l == 2 * d && l_sel == 4 * d &&
           l_sel == 64 && S_cmp >= 1 && S_cmp <= 64 * DSTEP_CH && S_sel >= 1 && S_sel <= 2048 && n_top >= 3 && n_top <= 64 && t_token >= 0 &&
           S_kv >= t_token + 1 && (int64_t)S_kv * 128 < ((int64_t)1 << 31) && kcs % 8 == 0 && kcb % 8 == 0 && kcg % 8 == 0 &&
           ((uintptr_t)Kc % 16 == 0) && sel_attn_decode_wg_supported(dtype, h, Dk, Dv, n_top, ksb, ksg, kss, vsb, vsg, vss, Q, K, V);
}

int launch_decode_step(const void *Q, const void *Kc, const void *K, const void *V, void *O, int32_t *ranges_out, int B, int G, int h, int S_cmp,
                       int S_sel, int S_kv, int n_top, int t_token, int64_t kcb, int64_t kcg, int64_t kcs, int64_t ksb, int64_t ksg, int64_t kss,
                       int64_t vsb, int64_t vsg, int64_t vss, int dtype, float scale, void *ws, size_t ws_bytes, hipStream_t st, const DecBandPair *band) {
    const int64_t R = (int64_t)B * G;
    NSA_CHECK_ARG(R >= 1 && R <= (1 << 24), "decode step: bad row count");
    const int nchunk = ((S_cmp + 63) / 64);
    int nw = 16, ns = 1, form = 0;
    NSA_CHECK_ARG(decode_step_plan(R, nchunk, h, S_sel, &nw, &ns, &form), "decode step: shape not covered (decode_step_supported)");
    const int cpw = form == 1 ? 4 : 2;
    DecStepParams P{Q, Kc, nullptr, nullptr, nullptr, nullptr, (int)R, G, h, S_cmp, S_sel, ns, nchunk, (nchunk + ns - 1) / ns, t_token, 0, kcb, kcg, kcs, scale * LOG2E};
    P.spin = tuning(TUNE_DECODE_TEAM_SPIN) >= 0 ? tuning(TUNE_DECODE_TEAM_SPIN) : 512;  // polls of ~0.5-1 us each before a workgroup goes on alone
    if (ns > 1) {
        NSA_CHECK_ARG(ws && ws_bytes >= decode_step_workspace(R, h, S_cmp) && ((uintptr_t)ws % 16 == 0), "decode step: workspace too small");
        P.part_g = (float *)ws;
        P.halo_g = P.part_g + (size_t)R * h * DSTEP_CH * 2;
        P.pg_g = P.halo_g + (size_t)R * DSTEP_CH * 16;
        P.cnt = decode_tickets(st, 2 * R);
        NSA_CHECK_ARG(P.cnt != nullptr, "decode step: could not allocate the arrival tickets");
    }
    SelectParams SP{};
    if (int rc = select_params_sequential(&SP, S_sel, 64, n_top, 1, 2, n_top)) return rc;
    SP.out = ranges_out;
    SP.R = R;
    SP.S = 1;
    SP.G = G;
    SP.t0 = t_token;
    const int c = (S_sel + 63) / 64;
    const int cand = c <= 1 ? 1 : c <= 2 ? 2 : c <= 4 ? 4 : c <= 8 ? 8 : c <= 16 ? 16 : 32;
    const DecAttnArgs AT{Q, K, V, O, G, h, S_kv, n_top, ksb, ksg, kss, vsb, vsg, vss, scale * LOG2E};
    void (*k)(DecStepParams, SelectParams, int, DecAttnArgs, DecBandPair);
    const bool bf = dtype == NSA_DT_BF16, split = ns > 1;
#define NSA_DSK(NW_, SP_, HC_) (bf ? decode_step_kernel<__bf16, NW_, SP_, HC_> : decode_step_kernel<_Float16, NW_, SP_, HC_>)
#define NSA_DSK4(NW_, HC_) (bf ? decode_step_kernel<__bf16, NW_, false, HC_, 4> : decode_step_kernel<_Float16, NW_, false, HC_, 4>)
#define NSA_DSK1(NW_, HC_) (bf ? decode_step_onepass_kernel<__bf16, NW_, HC_> : decode_step_onepass_kernel<_Float16, NW_, HC_>)
    if (form == 2) k = h == 6 ? (nw == 16 ? NSA_DSK1(16, 6) : NSA_DSK1(8, 6)) : (nw == 16 ? NSA_DSK1(16, 0) : NSA_DSK1(8, 0));
    else if (cpw == 4) k = h == 6 ? (nw == 16 ? NSA_DSK4(16, 6) : NSA_DSK4(8, 6)) : (nw == 16 ? NSA_DSK4(16, 0) : NSA_DSK4(8, 0));
    else if (h == 6) k = nw == 16 ? (split ? NSA_DSK(16, true, 6) : NSA_DSK(16, false, 6)) : (split ? NSA_DSK(8, true, 6) : NSA_DSK(8, false, 6));
    else k = nw == 16 ? (split ? NSA_DSK(16, true, 0) : NSA_DSK(16, false, 0)) : (split ? NSA_DSK(8, true, 0) : NSA_DSK(8, false, 0));
#undef NSA_DSK1
#undef NSA_DSK4
#undef NSA_DSK
    const size_t lds = dstep_lds(nw);
    // the score data sits in V tiles 1 .. (the prefetching waves of the 16-wave form own tiles 0, 14, 15)
    NSA_CHECK_ARG(dstep_score_bytes(nw, h, S_sel, cpw) <= dstep_score_room(nw), "decode step: S_sel too large");
    {  // raise the dynamic-LDS limit once per kernel (the runtime call costs about a millisecond)
        static std::mutex mu;
        static void *raised[32] = {};
        std::lock_guard<std::mutex> lk(mu);
        bool done = false;
        for (void *r : raised) done |= (r == (void *)k);
        if (!done) {
            NSA_HIP_TRY(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            for (void *&r : raised)
                if (!r) {
                    r = (void *)k;
                    break;
                }
        }
    }
    int64_t grid = ns > 1 ? ((R + 7) / 8) * 8 * ns : R;
    DecBandPair BP{};
    BP.n_sel = 0xffffffffu;
    if (band) {  // the layer step's sliding + compressed branches on workgroups behind the step's own (decode_band_workgroup)
        BP = *band;
        int64_t waves[2];
        NSA_CHECK_ARG(BP.w.Dk == 64 && BP.w.Dv == 64 && band_dual_plan(&BP.w, &BP.c, dtype, waves), "decode step: band branches not in split form");
        if (BP.mg.on) {  // splits merged by the workgroup that holds them: a unit = nsplit consecutive waves of one workgroup
            NSA_CHECK_ARG(BP.w.S == 1 && BP.c.S == 1 && BP.mg.gates && BP.w.O && BP.c.O && h <= 16, "decode step: band merge needs S = 1, outputs and a gate buffer");
            int nsm = 1;
            while (2 * nsm <= BP.w.nsplit && 2 * nsm <= nw) nsm *= 2;
            BP.w.nsplit = BP.c.nsplit = nsm;
            waves[0] = waves[1] = R * nsm;
        }
        const int64_t gw = (waves[0] + nw - 1) / nw, gc = (waves[1] + nw - 1) / nw;
        NSA_CHECK_ARG(grid + gw + gc < ((int64_t)1 << 31), "decode step: too many workgroups");
        BP.n_sel = (unsigned)grid;
        BP.n_w = (unsigned)gw;
        grid += gw + gc;
    }
    static_assert(2 * Geo<64>::TILE_BYTES <= DEC_ATT_TILE, "a band wave keeps its K and V tile in the wave's V tile of the step");
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(nw * 64), lds, st, P, SP, cand, AT, BP);
    NSA_LAUNCH_CHECK("decode_step");
    return NSA_OK;
}

}  // namespace nsa

#ifdef NSA_DEC_TS
extern "C" __attribute__((visibility("default"))) int nsa_debug_read_ts2(long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(nsa::g_ts2), sizeof(long long) * 32);
}
#endif
