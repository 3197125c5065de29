// Top-n selection of one row by one wave (shared by select_topn_kernel and the fused decode scorer).
#pragma once
#include <type_traits>

#include "nsa_common.hpp"

#ifndef DEC_TS
#define DEC_TS(i)
#endif
namespace nsa {

struct SelectParams {
    const float *p_grp;     // [R,S_sel]
    const int32_t *t_rows;  // [R] or null
    int32_t *out;           // [R,W,2]
    int64_t R;
    int S, G, t0, S_sel, l_sel, n_top, force_init, force_local, mode, W;
    int k_actual;      // picks per row
    int n_forced;      // forced entries used (sequential: all; batched: kept columns, maybe truncated)
    unsigned keepmask; // batched: bit i = sorted forced column i is kept
    int all_valid;     // batched with n_top >= S_sel: select every valid block
    int l_sel_shift;   // log2(l_sel) when l_sel is a power of two, else -1 (the block of a token without an integer division)
    int forced_plain;  // batched: every forced column is kept and used (S > 2 l'), the forced set is {0} + (cblk - force_local, cblk]
};

// runs of a selected-block bitmap (wd[c] = blocks 64 c .. 64 c + 63, wave uniform) -> token ranges, by all lanes at once: a lane that starts /
// ends a run takes the run's index from the bits below it and drops the token bound into scr[2 index (+1)]; lane i then holds run i in
// (my_s, my_e) ([0, 0) beyond the last run).  scr: 128 ints of LDS private to the wave.
template <int CAND>
__device__ __forceinline__ void extract_runs_lanes(const unsigned long long (&wd)[CAND], const int l_sel, const int t, int *scr, int &my_s, int &my_e) {
    const int lane = lane_id();
    my_s = 0;
    my_e = 0;
    int n_s = 0, n_e = 0;  // runs started / ended in the words so far (wave uniform)
#pragma unroll
    for (int c = 0; c < CAND; ++c) {
        const unsigned long long w = wd[c];
        if (w == 0ull) continue;
        const unsigned long long below = c > 0 ? (wd[c > 0 ? c - 1 : 0] >> 63) : 0ull;             // block 64c - 1 selected
        const unsigned long long above = c + 1 < CAND ? (wd[c + 1 < CAND ? c + 1 : c] << 63) : 0ull;  // block 64c + 64 selected
        const unsigned long long sm = w & ~((w << 1) | below);  // blocks that start a run
        const unsigned long long em = w & ~((w >> 1) | above);  // blocks that end one
        const int blk = 64 * c + lane;
        // (the scalar masks predicate the lanes directly; the run counts so far enter the bit counts as their start value)
        if (__builtin_amdgcn_inverse_ballot_w64(sm))
            scr[2 * (int)__builtin_amdgcn_mbcnt_hi((unsigned)(sm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)sm, (unsigned)n_s))] = blk * l_sel;
        if (__builtin_amdgcn_inverse_ballot_w64(em))
            scr[2 * (int)__builtin_amdgcn_mbcnt_hi((unsigned)(em >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)em, (unsigned)n_e)) + 1] =
                min((blk + 1) * l_sel, t + 1);
        n_s += __popcll(sm);
        n_e += __popcll(em);
    }
    if (lane < n_s) {  // the i-th start pairs with the i-th end: runs are disjoint and ascending
        my_s = scr[2 * lane];
        my_e = scr[2 * lane + 1];
    }
}

// Threshold (radix) select of the k_actual largest keys of a row spread over the wave -- lane l holds the order-preserving integer images
// u[c] of candidates lane + 64 c (0 = not a candidate) -- ties at the threshold to the lowest (c, lane), i.e. (key desc, index asc) when the
// candidates are laid out in ascending index order.  Returns the lane's pick mask (bit c).  Shared by the flat selector below and the two
// stages of the hierarchical one (select_topn_hier_*).
template <int CAND, bool SORT_ALL = false>
__device__ __forceinline__ typename std::conditional<(CAND <= 32), unsigned, unsigned long long>::type threshold_select(const unsigned (&u)[CAND],
                                                                                                                     const int k_actual) {
    using mask_t = typename std::conditional<(CAND <= 32), unsigned, unsigned long long>::type;
    const int lane = lane_id();
    mask_t sel = 0;
    constexpr bool SORTED = CAND <= 16 || SORT_ALL;  // beyond 1024 blocks the sorted copy does not fit the fused kernels' register budget
    unsigned w[CAND];  // per-lane descending copy (Batcher's odd-even merge sort, CAND a power of two)
#pragma unroll
    for (int c = 0; c < CAND; ++c) w[c] = u[c];
    if constexpr (SORTED) {
#pragma unroll
        for (int pp = 1; pp < CAND; pp <<= 1)
#pragma unroll
            for (int kk = pp; kk >= 1; kk >>= 1)
#pragma unroll
                for (int jj = kk % pp; jj + kk < CAND; jj += 2 * kk)
#pragma unroll
                    for (int ii = 0; ii < kk; ++ii)
                        if (ii + jj + kk < CAND && (ii + jj) / (2 * pp) == (ii + jj + kk) / (2 * pp)) {
                            const unsigned a = w[ii + jj], b = w[ii + jj + kk];
                            w[ii + jj] = max(a, b);
                            w[ii + jj + kk] = min(a, b);
                        }
    }
    DEC_TS(22);
    // counts are capped just above k: the search needs "at least k", and "exactly k" ends it early
    auto count_ge = [&](unsigned cand, int cap) -> int {
        int cnt = 0;
        if constexpr (SORTED && CAND >= 2) {
            // slots 0 and 1 in straight-line code (most rounds end there: one exit test instead of four), then slot by slot
            const unsigned long long b0 = __ballot(w[0] >= cand), b1 = __ballot(w[1] >= cand);
            cnt = __popcll(b0) + __popcll(b1);
            if (b1 == 0ull || cnt > cap) return cnt;  // (b0 == 0 implies b1 == 0: the slots are sorted per lane)
#pragma unroll
            for (int c = 2; c < CAND; ++c) {
                const unsigned long long b = __ballot(w[c] >= cand);
                if (b == 0ull) break;
                cnt += __popcll(b);
                if (cnt > cap) break;  // exact while <= cap
            }
            return cnt;
        }
#pragma unroll
        for (int c = 0; c < CAND; ++c) {
            const unsigned long long b = __ballot(w[c] >= cand);
            if (SORTED && b == 0ull) break;
            cnt += __popcll(b);
            if (SORTED && cnt > cap) break;  // exact while <= cap
        }
        return cnt;
    };
    const int k_eff = min(k_actual, count_ge(1u, k_actual));  // valid candidates, as far as they matter
    if (k_eff > 0) {
        // MSB-first search for the k-th largest key T.  A prefix with EXACTLY k keys at or above it ends the search: those k keys
        // are the picks whatever the remaining bits are (no tie can straddle the cut) -- typically after ~20 of the 32 rounds.
        DEC_TS(23);
        unsigned T = 0;
        bool exact = false;
        int bit0 = 31;
        {
            // the first nine rounds (sign + exponent of the k-th key) in at most two probes: the k-th largest key cannot exceed the
            // largest one, M; if at least k keys share M's top nine bits that IS the prefix of the k-th key, otherwise it is tried one
            // exponent lower (scores of one row rarely spread over more than two octaves), otherwise the search starts from the top bit.
            // Any prefix P with count(>= P) >= k > count(>= P + 2^23) is the one the bit-by-bit search arrives at: same T, same picks.
            unsigned lm = w[0];
            if constexpr (!SORTED) {
#pragma unroll
                for (int c = 1; c < CAND; ++c) lm = max(lm, w[c]);
            }
            const unsigned p1 = wave_max_u32(lm) & 0xFF800000u;
            if (p1 != 0u) {
                const int c1 = count_ge(p1, k_eff);
                if (c1 >= k_eff) {
                    T = p1;
                    bit0 = 22;
                    exact = c1 == k_eff;
                } else if (p1 > 0x00800000u) {
                    const unsigned p2 = p1 - 0x00800000u;
                    const int c2 = count_ge(p2, k_eff);
                    if (c2 >= k_eff) {
                        T = p2;
                        bit0 = 22;
                        exact = c2 == k_eff;
                    }
                }
            }
        }
        for (int bit = bit0; bit >= 0 && !exact; --bit) {
            const unsigned cand = T | (1u << bit);
            const int c = count_ge(cand, k_eff);
            if (c >= k_eff) T = cand;
            if (c == k_eff) {
                exact = true;
                break;
            }
        }
        DEC_TS(24);
        if (exact) {
#pragma unroll
            for (int c = 0; c < CAND; ++c)
                if (u[c] >= T) sel |= (mask_t)1 << c;
        } else {
            int cnt_gt = 0;  // keys above T: fewer than k_eff by construction
#pragma unroll
            for (int c = 0; c < CAND; ++c) {
                const unsigned long long b = __ballot(w[c] > T);
                if (SORTED && b == 0ull) break;
                cnt_gt += __popcll(b);
            }
            int remaining = k_eff - cnt_gt;  // >= 1 slots for the keys equal to T, lowest index first
            const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
            for (int c = 0; c < CAND; ++c) {  // ascending c then ascending lane = ascending block index
                if (__ballot(u[c] >= T) == 0ull) continue;
                const bool eq = u[c] == T;
                const unsigned long long em = __ballot(eq);
                const int take = min(__popcll(em), remaining);
                if (u[c] > T || (eq && __popcll(em & lt_mask) < take)) sel |= (mask_t)1 << c;
                remaining -= take;
            }
        }
    }
    return sel;
}

// one wave: top-n + forced + merge of ONE row.  p = the row's S_sel group scores (global or LDS).  Lane i < min(W, 64) returns
// range i in (my_s, my_e); ranges beyond the emitted runs are [0, 0).
// scr: 128 ints of LDS private to the wave, or null.  With it the runs of the selected bitmap are extracted by all lanes at once (a lane
// that starts / ends a run knows the run's index from the bits below it and drops the token bound into slot [index]); without it one
// scalar loop walks the runs (about 30 dependent scalar instructions per run: 1.5 us of a decode step).
// SORT_ALL: keep the per-lane sorted copy also beyond 16 candidates per lane (the select kernel has the registers for it; the fused
// kernels, at 1024 threads or 250 VGPRs, do not)
// blk_list (LDS, >= n_top + 3 ints, wave private until the caller publishes it) / nblk: the selected blocks themselves, ascending, and
// their count -- what a consumer that walks blocks (the fused decode step) needs, before and independent of the run extraction.
template <int CAND, bool SORT_ALL = false>
// pre (optional): the row's raw scores already in registers, pre[c] = p[min(lane + 64 c, max(nvalid - 1, 0))] -- a caller that selects
// several rows one after the other fetches the next row's while this one is processed (the scorer's epilogue)
__device__ __forceinline__ void select_topn_row_regs(const SelectParams &P, const float *p, const int t, int &my_s, int &my_e,
                                                     int *scr = nullptr, int *blk_list = nullptr, int *nblk = nullptr, const float *pre = nullptr) {
    const int lane = lane_id();
    const int l_sel = P.l_sel, S_sel = P.S_sel;
    const int sh = P.l_sel_shift;
    // (t is the same for every lane: one row per wave.  Saying so keeps the slot tests below on the scalar unit)
    const int nvalid_blocks = uniform(min(S_sel, sh >= 0 ? (t + 1) >> sh : (t + 1) / l_sel));  // blocks j with (j+1)*l' <= t+1

    using mask_t = typename std::conditional<(CAND <= 32), unsigned, unsigned long long>::type;
    float key[CAND];
    mask_t selbits = 0;  // bit c = block lane + 64 c selected
    // ALL slot loads go out before the first key is formed: unconditional, the index clamped to the last valid block (an entry the scorer
    // has written even with causal_skip; a row without any valid block reads entry 0 and masks it).  The first form guarded every slot with
    // a scalar branch -- no load crossed a branch, so a row paid one dependent memory round trip PER SLOT (8 on average at 64k): the
    // select kernel ran at the latency of ~8 serial HBM reads per row, 57 % of its wave cycles waiting (profiles/r02/pmc_select_S65536_B1.txt).
    const int jmax = max(nvalid_blocks - 1, 0);
    float vraw[CAND];
#pragma unroll
    for (int c = 0; c < CAND; ++c) vraw[c] = pre ? pre[c] : p[min(lane + 64 * c, jmax)];
#pragma unroll
    for (int c = 0; c < CAND; ++c) {
        const int j = lane + 64 * c;
        const float k = __fsub_rn(vraw[c], __fmul_rn((float)j, 1e-8f));
        key[c] = j < nvalid_blocks ? k : -INFINITY;
    }
    if (P.all_valid) {
#pragma unroll
        for (int c = 0; c < CAND; ++c) selbits |= lane + 64 * c < nvalid_blocks ? (mask_t)1 << c : (mask_t)0;
    }

    DEC_TS(20);
    if (!P.all_valid) {
        // ---- forced blocks
        const int cblk = uniform(max(sh >= 0 ? t >> sh : t / l_sel, 0));
        const int nf_all = (P.force_init ? 1 : 0) + P.force_local;
        if (P.mode == NSA_SEL_SEQUENTIAL || P.forced_plain) {
            // every entry of the forced list [0 (init)] + [max(cblk - a, 0) : a = force_local-1 .. 0] is taken: block j is forced iff it is
            // the initial block or lies in (cblk - force_local, cblk] (the clamp to 0 is the case force_local > cblk).  Sequential mode
            // keeps a forced block valid or not, batched mode drops the invalid ones (the partial current block).  The window spans one
            // or two slots: the others skip it on the scalar unit.
            const int keep_to = P.mode == NSA_SEL_SEQUENTIAL ? S_sel : nvalid_blocks;
            const int lo = max(cblk - P.force_local + 1, 0), hi = min(cblk, S_sel - 1), hi_sel = min(hi, keep_to - 1);
            if (P.force_local > 0 && lo <= hi) {
#pragma unroll
                for (int c = 0; c < CAND; ++c) {
                    if (64 * c > hi || 64 * c + 63 < lo) continue;
                    const int j = lane + 64 * c;
                    const bool f = (unsigned)(j - lo) <= (unsigned)(hi - lo);
                    key[c] = f ? -INFINITY : key[c];  // excluded from top-k
                    selbits |= (f && j <= hi_sel) ? (mask_t)1 << c : (mask_t)0;
                }
            }
            if (P.force_init && S_sel > 0) {
                key[0] = lane == 0 ? -INFINITY : key[0];
                selbits |= (lane == 0 && keep_to > 0) ? (mask_t)1 : (mask_t)0;
            }
        } else {
            int used = 0;
            for (int i = 0; i < nf_all; ++i) {
                // sorted forced list: [0 (init)] then max(cblk - a, 0) with a descending to 0
                int f;
                if (P.force_init && i == 0) f = 0;
                else f = max(cblk - (nf_all - 1 - i), 0);
                if (!((P.keepmask >> i) & 1u)) continue;
                if (used >= P.n_forced) break;
                ++used;
                if (f >= S_sel) continue;
                const bool valid = f < nvalid_blocks;
                if ((f & 63) == lane) {
                    const int c = f >> 6;
#pragma unroll
                    for (int cc = 0; cc < CAND; ++cc)
                        if (cc == c) {
                            key[cc] = -INFINITY;                       // excluded from top-k
                            if (valid) selbits |= (mask_t)1 << cc;  // batched drops invalid picks
                        }
                }
            }
        }
        // ---- top-k picks: threshold (radix) select on an order-preserving integer image of the key.
        // 32 rounds of {compare, ballot, popcount} find the k-th largest key T; everything above T is picked and the
        // remaining slots go to the keys equal to T in ascending index order -- exactly (key desc, idx asc), with no
        // cross-lane data movement (the iterative arg-max needed 12 dependent ds_bpermute per pick).
        // The rounds COUNT on a per-lane sorted copy of the keys (descending): a round asks "are there >= k keys >= cand", so it
        // walks the sorted slots until the count reaches k or a slot has no key >= cand -- one or two ballots instead of CAND
        // (k = 13 of 1024 at S = 64k: slot 0, the 64 lane maxima, settles most rounds).  The picks are then made on the
        // original keys, slot by slot in ascending index order, skipping slots without a key >= T.
        DEC_TS(21);
        unsigned u[CAND];
#pragma unroll
        for (int c = 0; c < CAND; ++c) {
            const unsigned bits = __float_as_uint(key[c]);
            const bool ok = key[c] > -INFINITY;  // forced / masked / NaN candidates never compete
            u[c] = ok ? ((bits & 0x80000000u) ? ~bits : (bits | 0x80000000u)) : 0u;  // valid keys map to >= 0x00800000
        }
        selbits |= threshold_select<CAND, SORT_ALL>(u, P.k_actual);
    }

    DEC_TS(25);
    // ---- run extraction
    my_s = 0;
    my_e = 0;
    if (scr) {
        unsigned long long wd[CAND];
#pragma unroll
        for (int c = 0; c < CAND; ++c) wd[c] = __ballot((selbits >> c) & 1u);
        if (blk_list) {  // the picked blocks in ascending order: a set lane drops its block id at the number of set bits below it
            int nb = 0;
#pragma unroll
            for (int c = 0; c < CAND; ++c) {
                const unsigned long long w = wd[c];
                if (w == 0ull) continue;
                if (__builtin_amdgcn_inverse_ballot_w64(w))
                    blk_list[(int)__builtin_amdgcn_mbcnt_hi((unsigned)(w >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)w, (unsigned)nb))] = 64 * c + lane;
                nb += __popcll(w);
            }
            *nblk = nb;
        }
        extract_runs_lanes<CAND>(wd, l_sel, t, scr, my_s, my_e);
        return;
    }
    int nrun = 0;
    int cur_s = -1, cur_e = -1;  // pending run (block ids), wave uniform
    auto emit = [&]() {
        if (nrun == lane) {
            my_s = cur_s * l_sel;
            my_e = min((cur_e + 1) * l_sel, t + 1);
        }
        ++nrun;
    };
#pragma unroll
    for (int c = 0; c < CAND; ++c) {
        unsigned long long w = __ballot((selbits >> c) & 1u);
        while (w) {  // maximal runs of ones of this word, ascending
            const int sb = __builtin_ctzll(w);
            const unsigned long long inv = ~(w >> sb);
            const int len = inv ? __builtin_ctzll(inv) : 64;
            const unsigned long long ones = (len == 64) ? ~0ull : ((1ull << len) - 1ull);
            w &= ~(ones << sb);
            const int s_blk = 64 * c + sb, e_blk = s_blk + len - 1;
            if (cur_s >= 0 && s_blk == cur_e + 1) {
                cur_e = e_blk;  // run continues across the word boundary
            } else {
                if (cur_s >= 0) emit();
                cur_s = s_blk;
                cur_e = e_blk;
            }
        }
    }
    if (cur_s >= 0) emit();
}

// same, storing the row's [W,2] ranges
template <int CAND>
__device__ __forceinline__ void select_topn_row(const SelectParams &P, const float *p, const int t, int32_t *out, int *scr = nullptr,
                                                const float *pre = nullptr) {
    const int lane = lane_id();
    int my_s, my_e;
    select_topn_row_regs<CAND, (CAND <= 32)>(P, p, t, my_s, my_e, scr, nullptr, nullptr, pre);
    if (lane < P.W) {
        out[2 * lane] = my_s;
        out[2 * lane + 1] = my_e;
    }
    // rows wider than a wave (W > 64 only when n_top >= S_sel > 64: a single run) -> zero the rest
    for (int i = lane + 64; i < P.W; i += 64) {
        out[2 * i] = 0;
        out[2 * i + 1] = 0;
    }
}

// one wave, one row, the instantiation picked per row: only blocks up to the current one can be valid or forced, so an early row of a long
// sequence runs the instantiation of a short one (the per-lane sort, the key transform and the pick masks scale with the slot count: at 64k
// the average row needs 10.7 of the 16 slots' worth of work).  t must be wave uniform.
template <int CAND>
__device__ __forceinline__ void select_topn_row_auto(const SelectParams &P, const float *p, const int t, int32_t *out, int *sc,
                                                     const float *pre = nullptr) {
    const int sh = P.l_sel_shift;
    const int cblk = max(sh >= 0 ? t >> sh : t / P.l_sel, 0);
    const int need = (min(cblk + 1, P.S_sel) + 63) >> 6;  // slots that hold a block <= cblk
    if constexpr (CAND >= 2) {
        if (need <= 1) return select_topn_row<1>(P, p, t, out, sc, pre);
    }
    if constexpr (CAND >= 4) {
        if (need <= 2) return select_topn_row<2>(P, p, t, out, sc, pre);
    }
    if constexpr (CAND >= 8) {
        if (need <= 4) return select_topn_row<4>(P, p, t, out, sc, pre);
    }
    if constexpr (CAND >= 16) {
        if (need <= 8) return select_topn_row<8>(P, p, t, out, sc, pre);
    }
    if constexpr (CAND >= 32) {
        if (need <= 16) return select_topn_row<16>(P, p, t, out, sc, pre);
    }
    select_topn_row<CAND>(P, p, t, out, sc, pre);
}

// host: SelectParams of the sequential selector (decode / per-row prefill), see sel_select.hip
int select_params_sequential(SelectParams *P, int S_sel, int l_sel, int n_top, int force_init, int force_local, int W);
int select_params_fill(SelectParams *P, int S_sel, int l_sel, int n_top, int force_init, int force_local, int mode, int S_total, int W);

}  // namespace nsa
