// Block form of the MFMA selection attention (prefill / training forward; bf16 / f16, Dk = Dv = 64).
//
// Why: the query-tile kernel (sel_attn_rows_mfma.hip, NT = 1) is bound by two things its structure fixes (PMC r01/n_*): ~21
// VALU+SALU instructions per MFMA (per-32-key-tile schedule walk, ballots, masks, DMA set-up) and the CU's vector-memory path
// (64 B/clk/CU): every row pair pulls each of its tiles through L1 -> LDS on its own, 10 GB per launch at S=4096, B=8.
// Here a wave owns NT column tiles of 16/h consecutive rows each (h = 6: NT = 4 -> 8 rows) and walks the UNION of their selected
// 64-key blocks (= the selection block l' of the reference geometry):
//   * a block is brought into wave-private LDS ONCE by LDS-DMA (16 KiB, 16 wave-instructions) and its K / V^T fragments are read
//     into registers once; every column tile whose rows selected the block then runs 8 + 8 MFMAs from those registers, tiles
//     none of whose rows touch the block are skipped by one scalar test.  At S = 4096 the union of 8 rows is ~0.4 x the sum of
//     their selections: that factor comes off the L1 -> LDS traffic and off the LDS reads;
//   * per-block overhead (schedule walk, two ballots, DMA issue, waits) is paid per 64 keys x up to NT tiles, not per 32 keys;
//   * a slot whose row did not select the block gets -inf through the softmax offset (ONE v_cndmask per tile; x = s*c2 - inf);
//     only blocks a row covers PARTIALLY (the causal clamp at t+1 of the sequential selector, unaligned user ranges, the tail
//     of K/V) take the masked path with a per-row 64-bit key mask rebuilt from the row's ranges.
// Semantics are those of every selection executor here: union of the clamped ranges, end <= start ignored, empty row -> zeros
// (attention_kernels.py:705-772).  Softmax (exp2 domain, deferred max), S^T / O^T formulation, swizzled LDS image and epilogue
// are those of sel_attn_rows_mfma.hip; outputs agree with it to rounding (same products, the key order inside a block is the same).
#include "attn_mfma_tiles.hpp"
#include "sel_select_row.hpp"

namespace nsa {

namespace blk {
constexpr int D = 64, ROWB = 128, KS = 2, MT = 4;
constexpr int BLK = 64;                 // keys per block
constexpr int TILE_BYTES = BLK * ROWB;  // 8 KiB per operand
__device__ __forceinline__ unsigned span(int lo, int hi) {  // bits lo..hi inclusive, 0 <= lo <= hi <= 31
    const unsigned up = hi >= 31 ? 0xffffffffu : ((1u << (hi + 1)) - 1u);
    return up & ~((1u << lo) - 1u);
}
__device__ __forceinline__ void lds_or(unsigned *p, unsigned v) {
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    __hip_atomic_fetch_or((lds_u32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
// one v_max3_f32 (fmaxf makes hipcc canonicalise each MFMA-derived operand with a v_max x,x first: 16 extra VALU per tile)
__device__ __forceinline__ float max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
}  // namespace blk

// RS: row sums of P on the matrix pipe (l += ones . P^T: 2 MFMAs per tile instead of 16 v_add; the pipe is ~18 % busy, the VALU is the
// bound) -- the sum then runs over the bf16 P the PV product uses, and every lane holds the whole column sum (no cross-lane step at the end)
// FLAT (h = 6, NT = 3): the 48 columns of the wave's three tiles are the 8 x 6 (row, head) pairs of its 8 query rows laid end to end, instead of
// 2 rows (12 of 16 columns) per tile and four tiles.  MFMA and VALU issue do not overlap on a gfx950 SIMD (tools/ubench/issue_rates.hip), so
// the idle quarter of every tile cost both its MFMAs and its softmax arithmetic.  Rows 2 and 5 straddle two tiles: every column keeps its own
// running max / sum / output (a column is one (row, head) pair either way), only the ownership masks see the difference.
// KSPLIT (round 2; by row position since round 4): at 64k a (b,g)'s K/V is 16 MiB against 4 MiB of L2 per XCD and the plain walk is bound by
// L2-miss traffic.  A row at position t touches keys [0, t] only, so the keys are split by where the row sits: rows below T1 are walked whole
// and write O directly, rows in [T1, T2) by TWO workgroups -- the even / odd 8-block (128 KiB) stripes of the keys -- and rows from T2 on by
// FOUR (stripe index mod 4).  All eight XCDs work on the same walk (pair, zone, class) at the same time, every XCD on every eighth workgroup of
// it: balanced for any number of pairs, each XCD keeps its own copy of the walk's key region.  A split row leaves per class (m, l, s) in fp32
// and its normalised O / s in f16 (2^-11 relative: below the rounding of the output dtype; s = 1 unless the class's output leaves the f16
// range, then a power of two) per (row, head); a second small launch merges the 2 / 4 records of a row in class order (bitwise reproducible).
// Defaults T1 = 32k, T2 = off, measured (profiles/r04/key_split_zones.txt): the L2 hit rate follows the region size as the gather
// microbenchmark says (0.71 unsplit -> 0.88 two classes above 16k -> 0.94 four classes above 32k, memory-side reads 13.1 -> 5.5 -> 2.3 GB at
// 64k x 4), but the launch does not: from two classes on it is bound by the CU's L2 -> LDS path with ONE 16 KiB block in flight per wave
// (8 x 16 KiB / ~1.1 us), and every further class costs its set-up and its records -- four classes above 32k are 5 % SLOWER than two, rows
// below 32k are best left whole.  (Round 2 split every row two ways, each (pair, half) on its own XCD: 15.8 ms at 64k x 16, now 15.5.)
template <typename T, int NT, bool RS, bool FLAT, bool KSPLIT = false>
__global__ __launch_bounds__(256, 2) void sel_attn_blocks_mfma_kernel(SelAttnParams P, SelectParams SP, int cand) {
    static_assert(!FLAT || NT == 3, "the flat layout is 3 tiles of 16 columns = 8 rows x 6 heads");
    using namespace blk;
    using M = MfmaT<T>;
    using x8 = typename M::x8;
    using x4 = typename M::x4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id();
    const int wave = uniform((int)(threadIdx.x >> 6));
    const int h = P.h, NW = P.nw, n = P.n;
    const int tpt = P.tpw;                 // rows per column tile (16 / h)
    const int tpw = FLAT ? 8 : NT * tpt;   // rows per wave
    const int ngrp = (P.S + tpw - 1) / tpw;  // row groups per (b,g)
    const int nbg = (int)(P.R / P.S);
    const int W4 = (ngrp + 3) >> 2;  // workgroups per (b,g)
    int bg, tc;
    // KSPLIT: zone of the workgroup's rows (0 unsplit, 1 two key classes, 2 four) and its class, from blockIdx alone (decoded again in the
    // epilogue: nothing of it stays live across the block loop, whose scalar registers are all taken).  Workgroups go round-robin over the
    // 8 XCDs: XCD x takes every eighth workgroup of each walk, walks in the order (pair; zone 0, zone 1 class 0, 1, zone 2 class 0 .. 3), so
    // the chip gathers from one <= 4 MiB key region at a time
    auto ks_decode = [&](int &bg_, int &tc_, int &zone, int &cls) -> bool {
        int bid = blockIdx.x;
        asm volatile("" : "+s"(bid));  // (opaque: the compiler must not keep the first decode's results alive for the later ones)
        const int xcd = bid & 7, idx = bid >> 3;
        const int per = P.ks_wa + 2 * P.ks_wb + 4 * P.ks_wc;
        bg_ = idx / per;
        int rem = idx - bg_ * per, i, w0, w1;
        zone = 0, cls = 0;
        if (rem < P.ks_wa) {
            i = rem, w0 = 0, w1 = P.ks_w1;
        } else if (rem < P.ks_wa + 2 * P.ks_wb) {
            rem -= P.ks_wa;
            zone = 1, cls = rem / P.ks_wb, i = rem - cls * P.ks_wb, w0 = P.ks_w1, w1 = P.ks_w2;
        } else {
            rem -= P.ks_wa + 2 * P.ks_wb;
            zone = 2, cls = rem / P.ks_wc, i = rem - cls * P.ks_wc, w0 = P.ks_w2, w1 = 0x7fffffff;
        }
        tc_ = w0 + 8 * i + xcd;
        return tc_ < w1;
    };
    if constexpr (KSPLIT) {
        int zone, cls;
        if (!ks_decode(bg, tc, zone, cls) || tc >= W4) return;
    } else if (P.map_mode == 2) {  // whole (b,g) pairs per XCD (workgroups go round-robin over the 8 XCDs)
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

    unsigned char *kl = smem + (size_t)wave * P.wave_lds;
    unsigned char *vl = kl + TILE_BYTES;
    int *rg = (int *)(vl + TILE_BYTES);                              // [tpw][n][2] clamped ranges
    unsigned *fullw = (unsigned *)(rg + ((2 * tpw * n + 3) & ~3));   // [tpw][NW] blocks a single range of the row covers completely
    unsigned *touchw = fullw + tpw * NW;                             // [tpw][NW] blocks with at least one selected key
    unsigned *kmask = touchw + tpw * NW;                             // [tpw][2]  64-bit key mask of a partially covered block

    // ---- per-slot constants; the Q loads go out first so that their latency runs under the range / bitmap work below
    const int rho = lane & 15, q = lane >> 4;
    // column -> (row of the wave, head): FLAT column 16 nn + rho = 6 row + head; else rho = h tsub + head with row = nn tpt + tsub
    auto col_row = [&](int nn) -> int { return FLAT ? (16 * nn + rho) / 6 : nn * tpt + rho / h; };
    auto col_head = [&](int nn) -> int { return FLAT ? (16 * nn + rho) % 6 : rho % h; };
    auto col_used = [&](int nn) -> bool { return FLAT ? col_row(nn) < ntok : (rho / h < tpt && col_row(nn) < ntok); };
    // bit of the column's row in the ownership masks (0 for an unused column): a few VALU ops where it is needed instead of NT registers
    const unsigned usedmask = span(0, ntok - 1);
    auto rowbit = [&](int nn) -> unsigned { return col_used(nn) ? (1u << col_row(nn)) : 0u; };
    unsigned nmask[NT];   // rows of column tile nn (wave uniform)
    x8 qf[NT][KS];
#pragma unroll
    for (int nn = 0; nn < NT; ++nn) {
        const int tok = col_row(nn), head = col_head(nn);
        const bool used = col_used(nn);
        const int r_lo = FLAT ? (16 * nn) / 6 : nn * tpt, r_hi = min(FLAT ? (16 * nn + 15) / 6 : nn * tpt + tpt - 1, ntok - 1);
        nmask[nn] = r_lo <= r_hi ? span(r_lo, r_hi) : 0u;
        const int64_t orow = (((int64_t)b * P.S + tw0 + (used ? tok : 0)) * P.G + g) * h + head;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            u32x4 raw = {0u, 0u, 0u, 0u};
            if (used) raw = *(const u32x4 *)((const T *)P.Q + orow * D + 32 * s + 8 * q);
            qf[nn][s] = __builtin_bit_cast(x8, raw);
        }
    }
    (void)usedmask;

    // ---- (1) bitmaps cleared, ranges of the rows -> LDS (from the fused selector or the ranges tensor)
    for (int i = lane; i < 2 * tpw * NW + 2 * tpw; i += 64) fullw[i] = 0u;
    auto mark_blocks = [&](int r, int s, int e) {  // one (row, range) pair per lane -> the row's block bitmaps
        if (e > s) {
            const int ta = s >> 6, tb = (e - 1) >> 6;          // blocks touched
            const int fa = (s + 63) >> 6, fb = (e >> 6) - 1;   // blocks covered completely: fa..fb
            for (int w = ta >> 5; w <= (tb >> 5); ++w) {
                const int base = 32 * w;
                lds_or(&touchw[r * NW + w], span(max(ta, base) - base, min(tb, base + 31) - base));
                const int flo = max(fa, base), fhi = min(fb, base + 31);
                if (flo <= fhi) lds_or(&fullw[r * NW + w], span(flo - base, fhi - base));
            }
        }
    };
    if (P.fuse_select) {
        for (int r = 0; r < ntok; ++r) {
            const int64_t row = ((int64_t)b * P.S + tw0 + r) * P.G + g;
            int s = 0, e = 0;
            const int t = SP.t_rows ? SP.t_rows[row] : SP.t0 + tw0 + r;
            const float *pg = SP.p_grp + row * (int64_t)SP.S_sel;
            switch (cand) {
                case 1: select_topn_row_regs<1>(SP, pg, t, s, e, (int *)kl); break;
                case 2: select_topn_row_regs<2>(SP, pg, t, s, e, (int *)kl); break;
                case 4: select_topn_row_regs<4>(SP, pg, t, s, e, (int *)kl); break;
                case 8: select_topn_row_regs<8>(SP, pg, t, s, e, (int *)kl); break;
                default: select_topn_row_regs<16>(SP, pg, t, s, e, (int *)kl); break;
            }
            if (lane < n) {
                int32_t *out = SP.out + row * (int64_t)n * 2;
                out[2 * lane] = s;
                out[2 * lane + 1] = e;
                s = min(max(s, 0), P.S_kv);
                e = min(max(e, s), P.S_kv);
                rg[2 * (r * n + lane)] = s;
                rg[2 * (r * n + lane) + 1] = e;
            }
        }
        wave_lds_fence();
        // ---- (2) block bitmaps: one (row, range) pair per lane
        for (int p = lane; p < ntok * n; p += 64) mark_blocks(p / n, rg[2 * p], rg[2 * p + 1]);
    } else {
        // the ranges of ALL rows of the wave in one round of loads, one (row, range) pair per lane (a load per row, one row after the other,
        // put 8 dependent memory round trips in front of every wave: the setup was 15 % of the kernel at S = 4096), and straight on into
        // the bitmaps (the cleared words are visible: same wave, LDS operations complete in order)
        wave_lds_fence();
        for (int p = lane; p < ntok * n; p += 64) {
            const int r = p / n, i = p - r * n;
            const int64_t row = ((int64_t)b * P.S + tw0 + r) * P.G + g;
            const int32_t *in = P.ranges + (row * n + i) * 2;
            int s = in[0], e = in[1];
            s = min(max(s, 0), P.S_kv);
            e = min(max(e, s), P.S_kv);
            rg[2 * p] = s;
            rg[2 * p + 1] = e;
            mark_blocks(r, s, e);
        }
    }
    wave_lds_fence();
    // union over the rows = the block schedule; kept in a register (lane w holds word w, NW <= 64)
    unsigned u0 = 0u;
    for (int r = 0; r < ntok; ++r)
        if (lane < NW) u0 |= touchw[r * NW + lane];
    if constexpr (KSPLIT) {  // stripes of 8 blocks, dealt to the classes in turn: every class sees the same causal shape
        int bg_, tc_, zone, cls;
        ks_decode(bg_, tc_, zone, cls);
        if (zone == 1) u0 &= 0x00FF00FFu << (8 * cls);
        else if (zone == 2) u0 &= 0x000000FFu << (8 * cls);
    }
    unsigned long long nz0 = __ballot(u0 != 0u);

    const unsigned char *Kb = (const unsigned char *)((const T *)P.K + (int64_t)b * P.ksb + (int64_t)g * P.ksg);
    const unsigned char *Vb = (const unsigned char *)((const T *)P.V + (int64_t)b * P.vsb + (int64_t)g * P.vsg);
    const int64_t krowb = P.kss * 2, vrowb = P.vss * 2;
    auto make_rsrc = [&](const unsigned char *base, int64_t bytes) {
        const uint64_t a = (uint64_t)base;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), (short)0,
                                                 __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
    };
    [[maybe_unused]] const auto krs = make_rsrc(Kb, (int64_t)(P.S_kv - 1) * krowb + ROWB);
    [[maybe_unused]] const auto vrs = make_rsrc(Vb, (int64_t)(P.S_kv - 1) * vrowb + ROWB);
    [[maybe_unused]] const int krowb32 = uniform((int)krowb), vrowb32 = uniform((int)vrowb);
    // one wave-wide 16-B load covers 8 rows x 128 B; the XOR swizzle sits on the SOURCE side (the DMA destination is lane-linear),
    // and for 8-row pieces it depends on the lane only: swz_k(8 i + r) = r & 7, swz_v(8 i + r) = (r >> 1) & 3
    const int ld_row = lane >> 3, ld_piece = lane & 7;
    [[maybe_unused]] const uint32_t kdma = (uint32_t)(ld_row * krowb + ((ld_piece ^ (ld_row & 7)) << 4));
    [[maybe_unused]] const uint32_t vdma = (uint32_t)(ld_row * vrowb + (((((ld_piece >> 1) ^ ((ld_row >> 1) & 3)) << 1) | (ld_piece & 1)) << 4));
    uint32_t krd0[KS], vrd0[MT];
#pragma unroll
    for (int s = 0; s < KS; ++s) krd0[s] = rho * ROWB + (((4 * s + q) ^ (rho & 7)) << 4);
    {
        const int qq = rho >> 2, pp = rho & 3, r = 4 * q + qq;
#pragma unroll
        for (int m = 0; m < MT; ++m) vrd0[m] = r * ROWB + ((m ^ ((r >> 1) & 3)) << 5) + 8 * pp;
    }
    // a block of 64 keys starting at tok0 -> wave-private LDS; rows past the end of K/V re-read the last row (they are masked:
    // a block reaching past S_kv is never `full`)
    auto issue_dma = [&](int tok0) {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef __attribute__((address_space(3))) void lds_void;
        const int ks = uniform(tok0 * krowb32), vs = uniform(tok0 * vrowb32);
        if (tok0 + BLK <= P.S_kv) {
            // K/V rows are contiguous (128 B apart: checked by the host), so 8 rows = 1 KiB on both sides: the instruction's immediate
            // offset (added to the memory address AND to the LDS address) steps through four pieces per M0 / soffset setting
#define NSA_BLK_DMA(RS, LP, VO, SO, I) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, (lds_void *)((LP) + ((I) >> 2) * 4096), 16, VO, (SO) + ((I) >> 2) * 4096, ((I)&3) * 1024, 0)
            NSA_BLK_DMA(krs, kl, kdma, ks, 0); NSA_BLK_DMA(krs, kl, kdma, ks, 1); NSA_BLK_DMA(krs, kl, kdma, ks, 2); NSA_BLK_DMA(krs, kl, kdma, ks, 3);
            NSA_BLK_DMA(krs, kl, kdma, ks, 4); NSA_BLK_DMA(krs, kl, kdma, ks, 5); NSA_BLK_DMA(krs, kl, kdma, ks, 6); NSA_BLK_DMA(krs, kl, kdma, ks, 7);
            NSA_BLK_DMA(vrs, vl, vdma, vs, 0); NSA_BLK_DMA(vrs, vl, vdma, vs, 1); NSA_BLK_DMA(vrs, vl, vdma, vs, 2); NSA_BLK_DMA(vrs, vl, vdma, vs, 3);
            NSA_BLK_DMA(vrs, vl, vdma, vs, 4); NSA_BLK_DMA(vrs, vl, vdma, vs, 5); NSA_BLK_DMA(vrs, vl, vdma, vs, 6); NSA_BLK_DMA(vrs, vl, vdma, vs, 7);
#undef NSA_BLK_DMA
        } else {
            const int last = P.S_kv - 1 - tok0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int rc = min(8 * i + ld_row, last);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (lds_void *)(kl + i * 1024), 16,
                                                         rc * krowb32 + ((ld_piece ^ (ld_row & 7)) << 4), ks, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (lds_void *)(vl + i * 1024), 16,
                                                         rc * vrowb32 + (((((ld_piece >> 1) ^ ((ld_row >> 1) & 3)) << 1) | (ld_piece & 1)) << 4), vs, 0, 0);
            }
        }
#else
        (void)tok0;
#endif
    };

    f32x4 o[NT][MT];
    float mrun[NT], lrun[NT];
#pragma unroll
    for (int nn = 0; nn < NT; ++nn) {
#pragma unroll
        for (int m = 0; m < MT; ++m) o[nn][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // unused slots carry +inf so that their scores never trigger the max-raising path
        mrun[nn] = rowbit(nn) ? -INFINITY : INFINITY;
        lrun[nn] = 0.f;
    }
    const float c2 = P.scale * LOG2E;
    x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = Elt<T>::from_f(1.f);

    // ---- (3) block schedule = set bits of the union bitmap, ascending (wave-uniform scalars)
    int iw = 0;
    unsigned ibits = 0u;
    auto next_blk = [&]() -> int {
        if (ibits == 0u) {
            if (!nz0) return -1;
            iw = __builtin_ctzll(nz0);
            nz0 &= nz0 - 1ull;
            ibits = (unsigned)__builtin_amdgcn_readlane((int)u0, iw);
        }
        const int bit = __builtin_ctz(ibits);
        ibits &= ibits - 1u;
        return 32 * iw + bit;
    };
    int cur = next_blk();
    if (cur >= 0) issue_dma(BLK * cur);
    int cw = -1;
    unsigned fw = 0u, tw = 0u;

    while (cur >= 0) {
        const int nxt = next_blk();
        const int tok0 = BLK * cur;
        x8 kfr[4][KS];
        x8 va[MT][2];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // LDS-DMA completion is a vmcnt event
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int s = 0; s < KS; ++s) kfr[u][s] = *(const x8 *)(kl + krd0[s] + u * 16 * ROWB);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const x4 lo = M::tr(vl + vrd0[m] + hf * 32 * ROWB), hi = M::tr(vl + vrd0[m] + (hf * 32 + 16) * ROWB);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    va[m][hf][j] = lo[j];
                    va[m][hf][4 + j] = hi[j];
                }
            }
        // ownership of this block: bit r of fullm / touchm = row r covers it completely / selected at least one of its keys
        if ((cur >> 5) != cw) {  // lane r < ntok caches row r's bitmap words of the current 32-block group
            cw = cur >> 5;
            if (lane < ntok) {
                fw = fullw[lane * NW + cw];
                tw = touchw[lane * NW + cw];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments are in registers: the buffers may be refilled
        __builtin_amdgcn_sched_barrier(0);
        if (nxt >= 0) issue_dma(BLK * nxt);
        const unsigned fullm = (unsigned)__ballot((fw >> (cur & 31)) & 1u);
        const unsigned touchm = (unsigned)__ballot((tw >> (cur & 31)) & 1u);
        const unsigned partm = touchm & ~fullm;
        if (partm) {  // partially covered by some row: rebuild that row's 64-key mask from its ranges
            if (lane < 2 * tpw) kmask[lane] = 0u;
            wave_lds_fence();
            for (int p = lane; p < ntok * n; p += 64) {
                const int r = p / n;
                const int lo = max(rg[2 * p], tok0) - tok0, hi = min(rg[2 * p + 1], tok0 + BLK) - tok0;
                if (((partm >> r) & 1u) && hi > lo) {
                    if (lo < 32) lds_or(&kmask[2 * r], span(lo, min(hi, 32) - 1));
                    if (hi > 32) lds_or(&kmask[2 * r + 1], span(max(lo, 32) - 32, hi - 33));
                }
            }
            wave_lds_fence();
        }

#pragma unroll
        for (int nn = 0; nn < NT; ++nn) {
            if (!(touchm & nmask[nn])) continue;  // no row of this column tile selected a key of the block
            f32x4 sacc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                sacc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) sacc[u] = M::mma(kfr[u][s], qf[nn][s], sacc[u]);
            }
            const unsigned rbit = rowbit(nn);
            const bool on = (fullm & rbit) != 0u;
            float x[16];
            unsigned vbits;  // bit 4u+j: key 16u + 4q + j of the block is selected by this slot's row
            float tmax;
            if (!(partm & nmask[nn])) {
                // every slot is all-on or all-off: the softmax offset carries the mask (s*c2 - inf = -inf)
                vbits = on ? 0xffffu : 0u;
                const float mneg = on ? -mrun[nn] : -INFINITY;
                const f32x2 c22 = {c2, c2}, mn2 = {mneg, mneg};  // two exponents per v_pk_fma_f32 (4.9 cycles against 2 x 3.9)
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 4; j += 2) {
                        const f32x2 s2 = {sacc[u][j], sacc[u][j + 1]};
                        const f32x2 t2 = __builtin_elementwise_fma(s2, c22, mn2);
                        x[4 * u + j] = t2[0];
                        x[4 * u + j + 1] = t2[1];
                    }
            } else {
                unsigned lo32 = on ? 0xffffffffu : 0u, hi32 = lo32;
                if (partm & rbit) {
                    lo32 = kmask[2 * col_row(nn)];
                    hi32 = kmask[2 * col_row(nn) + 1];
                }
                lo32 >>= 4 * q;
                hi32 >>= 4 * q;
                vbits = (lo32 & 0xfu) | ((lo32 >> 12) & 0xf0u) | ((hi32 & 0xfu) << 8) | ((hi32 >> 4) & 0xf000u);
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = ((vbits >> i) & 1u) ? fmaf(sacc[i >> 2][i & 3], c2, -mrun[nn]) : -INFINITY;
            }
            tmax = max3(max3(max3(x[0], x[1], x[2]), max3(x[3], x[4], x[5]), max3(x[6], x[7], x[8])),
                        max3(max3(x[9], x[10], x[11]), x[12], x[13]), max3(x[14], x[15], x[15]));
            if (__any(!(tmax <= RESCALE_THR))) {
                float vmax = -INFINITY;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float v = ((vbits >> i) & 1u) ? sacc[i >> 2][i & 3] * c2 : -INFINITY;
                    x[i] = v;
                    vmax = fmaxf(vmax, v);
                }
                vmax = fmaxf(vmax, __shfl_xor(vmax, 16, 64));
                vmax = fmaxf(vmax, __shfl_xor(vmax, 32, 64));
                const float mnew = fmaxf(mrun[nn], vmax);
                const float msub = (mnew == -INFINITY) ? 0.f : mnew;  // nothing valid seen yet: keep x = -inf, p = 0
                const float alpha = (mnew == mrun[nn]) ? 1.f : __builtin_amdgcn_exp2f(mrun[nn] - msub);
                mrun[nn] = mnew;
                lrun[nn] *= alpha;
#pragma unroll
                for (int m = 0; m < MT; ++m) o[nn][m] *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] -= msub;
            }
            float psum = 0.f;
            x8 pf[2];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float pe = __builtin_amdgcn_exp2f(x[i]);
                if constexpr (!RS) psum += pe;
                pf[i >> 3][i & 7] = Elt<T>::from_f(pe);
            }
            if constexpr (RS) {
                f32x4 ls = M::mma(ones, pf[0], (f32x4){0.f, 0.f, 0.f, 0.f});
                ls = M::mma(ones, pf[1], ls);
                psum = ls[0];
            }
            lrun[nn] += psum;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                o[nn][m] = M::mma(va[m][0], pf[0], o[nn][m]);
                o[nn][m] = M::mma(va[m][1], pf[1], o[nn][m]);
            }
        }
        cur = nxt;
    }

    // ---- epilogue
    [[maybe_unused]] int kzone = 0, ksp = 0;
    if constexpr (KSPLIT) {
        int bg_, tc_;
        ks_decode(bg_, tc_, kzone, ksp);
    }
#pragma unroll
    for (int nn = 0; nn < NT; ++nn) {
        float ltot = lrun[nn];
        if constexpr (!RS) {  // per-lane partial sums over the lane's 16 keys of every block -> column sum
            ltot += __shfl_xor(ltot, 16, 64);
            ltot += __shfl_xor(ltot, 32, 64);
        }
        if (!rowbit(nn)) continue;
        const int64_t orow = (((int64_t)b * P.S + tw0 + col_row(nn)) * P.G + g) * h + col_head(nn);
        const float inv = ltot > 0.f ? 1.f / ltot : 0.f;
        if (KSPLIT && kzone > 0) {  // partial record of this class: ml[rec] = (m, l, s, -), po[rec][64] = O / (l s) in f16
            // s = 1 unless the half's output leaves the f16 range (bf16 inputs with |V| > 32768): then the power of two that brings its
            // largest element back below 2^15 -- the merge multiplies it back in, so the record never overflows where the plain walk is finite
            float amax = 0.f;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fabsf(o[nn][m][j] * inv));
            amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
            amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
            float sc = 1.f, isc = 1.f;
            if (amax > 32768.f && amax < INFINITY) {
                const int e = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 127 - 14;  // amax / 2^e < 2^15
                sc = __uint_as_float((unsigned)(e + 127) << 23);
                isc = __uint_as_float((unsigned)(127 - e) << 23);
            }
            // records: zone 1 rows first ([bg][t - r1][head][2]), then zone 2 ([bg][t - r2][head][4]); the (m, l, s) quadruples, then the f16 rows
            const int nbg_ = (int)(P.R / P.S), t = tw0 + col_row(nn), hd = col_head(nn);
            const int64_t n1 = (int64_t)nbg_ * (P.ks_r2 - P.ks_r1) * h * 2, nrec = n1 + (int64_t)nbg_ * (P.S - P.ks_r2) * h * 4;
            const int64_t rec = kzone == 1 ? (((int64_t)bg * (P.ks_r2 - P.ks_r1) + (t - P.ks_r1)) * h + hd) * 2 + ksp
                                           : n1 + (((int64_t)bg * (P.S - P.ks_r2) + (t - P.ks_r2)) * h + hd) * 4 + ksp;
            float *ml = (float *)P.part;
            _Float16 *po = (_Float16 *)(ml + nrec * 4) + rec * 64;
            if (q == 0) *(f32x4 *)(ml + rec * 4) = (f32x4){mrun[nn], ltot, sc, 0.f};
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                f16x4 ov;
#pragma unroll
                for (int j = 0; j < 4; ++j) ov[j] = (_Float16)(o[nn][m][j] * inv * isc);
                *(f16x4 *)(po + 16 * m + 4 * q) = ov;
            }
            continue;
        }
        T *Or = (T *)P.O + orow * D;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            x4 ov;
#pragma unroll
            for (int j = 0; j < 4; ++j) ov[j] = Elt<T>::from_f(o[nn][m][j] * inv);
            *(x4 *)(Or + 16 * m + 4 * q) = ov;
        }
        if (P.lse && q == 0) P.lse[orow] = ltot > 0.f ? (mrun[nn] + __builtin_amdgcn_logf(ltot)) * LN2 : -INFINITY;
    }
}

// merge of the 2 / 4 class records of a split row, in class order: thread = ((pair, row >= r1, head), 8 output elements)
template <typename T>
__global__ __launch_bounds__(256) void sel_attn_ksplit_combine_kernel(const float *__restrict__ ml, const _Float16 *__restrict__ po, T *__restrict__ O,
                                                                      float *__restrict__ lse, int nbg, int S, int G, int h, int r1, int r2) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3;
    const int oct = threadIdx.x & 7;
    const int64_t per = (int64_t)(S - r1) * h;
    if (i >= per * nbg) return;
    const int bg = (int)(i / per);
    const int rem = (int)(i - bg * per), t = r1 + rem / h, hd = rem % h;
    const int64_t n1 = (int64_t)nbg * (r2 - r1) * h * 2;
    const int k = t < r2 ? 2 : 4;
    const int64_t rec = t < r2 ? (((int64_t)bg * (r2 - r1) + (t - r1)) * h + hd) * 2 : n1 + (((int64_t)bg * (S - r2) + (t - r2)) * h + hd) * 4;
    f32x4 r[4];
    float M = -INFINITY;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        r[c] = c < k ? *(const f32x4 *)(ml + (rec + c) * 4) : (f32x4){-INFINITY, 0.f, 1.f, 0.f};  // (m, l, s, -) of every class
        M = fmaxf(M, r[c][0]);
    }
    float w[4], L = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        w[c] = r[c][1] > 0.f ? r[c][1] * __builtin_amdgcn_exp2f(r[c][0] - M) : 0.f;
        if (c < k) L += w[c];
    }
    typedef typename MfmaT<T>::x8 x8;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c < k) {
            // (s = 1 in every ordinary record: x * 1.0f is exact, the result is that of the unscaled merge bit for bit)
            const float a = L > 0.f ? (w[c] / L) * r[c][2] : 0.f;
            const f16x8 p = *(const f16x8 *)(po + (rec + c) * 64 + 8 * oct);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = c == 0 ? (float)p[j] * a : acc[j] + (float)p[j] * a;
        }
    }
    x8 out;
#pragma unroll
    for (int j = 0; j < 8; ++j) out[j] = Elt<T>::from_f(acc[j]);
    const int b = bg / G, g = bg - b * G;
    const int64_t orow = (((int64_t)b * S + t) * G + g) * h + hd;
    *(x8 *)(O + orow * 64 + 8 * oct) = out;
    if (lse && oct == 0) lse[orow] = L > 0.f ? (M + __builtin_amdgcn_logf(L)) * LN2 : -INFINITY;
}

// ---- host side ----------------------------------------------------------------------------
// key-split form wanted for this shape?  TUNE_SEL_KSPLIT: -1 = by the K/V footprint of a (b,g) pair against the 4 MiB of L2 per XCD, 0 never,
// 1 always.  The zones of the form (rows walked whole / in two / in four key classes) follow the rows' positions, so a short context is all
// zone 0 and costs nothing but the different workgroup order; the rule below is where the measured gain starts (profiles/r04/key_split_zones.txt).
static bool ksplit_rule(int dtype, int h, int Dk, int Dv, int S, int S_kv, int n, int64_t R) {
    const int mode = tuning(TUNE_SEL_KSPLIT);
    if (mode == 0 || S <= 0 || Dv != 64 || Dk != 64 || !(dtype == NSA_DT_BF16 || dtype == NSA_DT_F16)) return false;
    if (mode > 0) return true;
    (void)n, (void)R;
    return S_kv >= 32768;  // (at 32k no row is split yet: the shared-walk workgroup order alone is 0.96 x the plain order; 0.89 x at 48k, 0.80-0.82 x at 64k)
}
// zone boundaries in rows of a (b,g), multiples of the rows of one workgroup: rows whose position (row + S_kv - S: the rows are the last S
// positions of the context) is below T1 are walked whole, below T2 in two key classes, the others in four
static void ksplit_zones(int h, int S, int S_kv, int *r1, int *r2) {
    const int rpw = 4 * 4 * (16 / h);  // rows per workgroup of the NT = 4 form
    const int t1 = tuning(TUNE_SEL_KSPLIT_T1) >= 0 ? tuning(TUNE_SEL_KSPLIT_T1) : 32768;
    const int t2 = tuning(TUNE_SEL_KSPLIT_T2) >= 0 ? tuning(TUNE_SEL_KSPLIT_T2) : (1 << 30);
    const int off = S_kv > S ? S_kv - S : 0;
    auto rows_below = [&](int t) {
        int64_t r = (int64_t)t - off;
        r = r < 0 ? 0 : (r > S ? S : r);
        r = (r + rpw - 1) / rpw * rpw;
        return (int)(r > S ? (S + rpw - 1) / rpw * rpw : r);
    };
    *r1 = rows_below(t1);
    *r2 = rows_below(t2 > t1 ? t2 : t1);
}
size_t sel_attn_ksplit_workspace(int dtype, int h, int Dk, int Dv, int S, int S_kv, int n, int64_t R) {
    if (!ksplit_rule(dtype, h, Dk, Dv, S, S_kv, n, R) || h < 1 || h > 16) return 0;
    int r1, r2;
    ksplit_zones(h, S, S_kv, &r1, &r2);
    const int64_t nbg = R / S;
    const int64_t rows2 = r2 > S ? 0 : S - r2, rows1 = (r2 > S ? S : r2) - (r1 > S ? S : r1);
    const size_t nrec = (size_t)nbg * h * (2 * (size_t)(rows1 > 0 ? rows1 : 0) + 4 * (size_t)rows2);
    return nrec * (4 * sizeof(float) + 64 * sizeof(_Float16)) + 16;  // (16: a launch without any split row still gets a non-null workspace)
}

// Column tiles per wave for a shape, 0 = not covered (the query-tile kernel takes it).  TUNE_SEL_BLOCKS: -1 auto, 0 off, N forces NT = N.
int sel_attn_blocks_nt(int dtype, int h, int Dk, int Dv, int S, int S_kv, int n, int64_t R, int64_t kss, int64_t vss) {
    if (!(dtype == NSA_DT_BF16 || dtype == NSA_DT_F16) || Dk != 64 || Dv != 64 || h < 1 || h > 16) return 0;
    if (n < 1 || n > 64 || S_kv < 1 || S_kv > 131072) return 0;
    if (kss != 64 || vss != 64) return 0;  // the DMA steps through rows 128 B apart (rows of a [.., S, 64] cache are)
    const int mode = tuning(TUNE_SEL_BLOCKS);
    if (mode == 0) return 0;
    // automatic choice: the block form wherever it applies.  Same-box A/B (profiles/r02/c_kernel_form_sweep_3.txt): 5-25 % faster than the
    // query-tile pairs from S = 1k to 64k.  (Before the selector's run extraction went lane-parallel the pairs still won by 5 % between
    // S_kv = 8k and 48k, c_kernel_form_sweep_2.txt: the fused selector is a larger share of a wave that owns 8 rows.)
    const int tpt = 16 / h;
    int nt = (mode == 1 || mode == 2 || mode == 4) ? mode : 4;
    while (nt > 1 && nt * tpt > 32) nt >>= 1;  // ownership masks are 32 bits wide
    if (nt * tpt > 32) return 0;
    while (nt > 1 && S < 2 * nt * tpt) nt >>= 1;  // short sequences: do not leave most of a wave's tiles empty
    if (S < tpt) return 0;
    (void)R;
    return nt;
}

template <typename T, int NT, bool FLAT, bool KSPLIT = false>
static int launch_blocks_t(const SelAttnParams &P0, hipStream_t st) {
    SelAttnParams P = P0;
    const int tpt = 16 / P.h, tpw = FLAT ? 8 : NT * tpt;
    P.tpw = tpt;
    P.nw = ((P.S_kv + 63) / 64 + 31) / 32;
    P.nsplit = 1;
    P.part = KSPLIT ? (float *)P0.ks_ws : nullptr;
    const int rg_ints = (2 * tpw * P.n + 3) & ~3;
    const int bm_ints = (2 * tpw * P.nw + 2 * tpw + 3) & ~3;
    P.wave_lds = 2 * blk::TILE_BYTES + 4 * (rg_ints + bm_ints);
    const size_t lds = 4 * (size_t)P.wave_lds;
    NSA_CHECK_ARG(lds <= 160 * 1024, "sel_attn_blocks: %zu B of LDS needed", lds);
    const int64_t nbg = P.R / P.S;
    const int64_t ngrp = (P.S + tpw - 1) / tpw;
    const int64_t W4 = (ngrp + 3) / 4;
    int64_t grid = nbg * W4;
    if constexpr (KSPLIT) {
        int r1, r2;
        ksplit_zones(P.h, P.S, P.S_kv, &r1, &r2);
        const int rpw = 4 * tpw;
        const int64_t wA = r1 / rpw < W4 ? r1 / rpw : W4, wB = r2 / rpw < W4 ? r2 / rpw : W4;
        P.ks_w1 = (int)wA;
        P.ks_w2 = (int)wB;
        P.ks_r1 = (int)(wA * rpw < P.S ? wA * rpw : P.S);
        P.ks_r2 = (int)(wB * rpw < P.S ? wB * rpw : P.S);
        P.ks_wa = (int)((wA + 7) / 8);
        P.ks_wb = (int)((wB - wA + 7) / 8);
        P.ks_wc = (int)((W4 - wB + 7) / 8);
        grid = 8 * nbg * ((int64_t)P.ks_wa + 2 * P.ks_wb + 4 * P.ks_wc);
    }
    NSA_CHECK_ARG(grid < (int64_t)1 << 31, "sel_attn_blocks: grid too large");
    P.map_mode = (nbg % 8 == 0) ? 2 : 1;
    SelectParams SP{};
    int cand = 0;
    if (P.fuse_select) {
        NSA_CHECK_ARG(P.select != nullptr, "fused selection without selector parameters");
        SP = *(const SelectParams *)P.select;
        const int c = (SP.S_sel + 63) / 64;
        NSA_CHECK_ARG(c <= 16 && SP.W <= 64 && SP.W == P.n, "fused selection: S_sel <= 1024 and at most 64 ranges per row");
        cand = c <= 1 ? 1 : c <= 2 ? 2 : c <= 4 ? 4 : c <= 8 ? 8 : 16;
    }
    void (*k)(SelAttnParams, SelectParams, int) = tuning(TUNE_SEL_ROWSUM) ? sel_attn_blocks_mfma_kernel<T, NT, true, FLAT, KSPLIT>
                                                                          : sel_attn_blocks_mfma_kernel<T, NT, false, FLAT, KSPLIT>;
    if (lds > 64 * 1024) NSA_HIP_TRY(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(256), lds, st, P, SP, cand);
    NSA_LAUNCH_CHECK("sel_attn_blocks_mfma");
    if constexpr (KSPLIT) {
        const int64_t items = nbg * (int64_t)(P.S - P.ks_r1) * P.h;  // (row, head) pairs with records
        if (items > 0) {
            const int64_t nrec = nbg * P.h * (2 * (int64_t)(P.ks_r2 - P.ks_r1) + 4 * (int64_t)(P.S - P.ks_r2));
            const float *ml = (const float *)P.part;
            hipLaunchKernelGGL(sel_attn_ksplit_combine_kernel<T>, dim3((unsigned)((items * 8 + 255) / 256)), dim3(256), 0, st, ml,
                               (const _Float16 *)(ml + nrec * 4), (T *)P.O, P.lse, (int)nbg, P.S, P.G, P.h, P.ks_r1, P.ks_r2);
            NSA_LAUNCH_CHECK("sel_attn_ksplit_combine");
        }
    }
    return NSA_OK;
}

int launch_sel_attn_blocks_mfma(const SelAttnParams &P, int dtype, int nt, hipStream_t st) {
    // 8 rows per wave with h = 6: three fully used column tiles instead of four with 12 of 16 columns.  It pays while the rows of a wave share
    // their blocks (every tile of a block is computed anyway: -10 % at 2k, -8 % at 4k); once a block belongs to one row (S_kv >> n l') the
    // four-tile form computes the one tile of that row and skips three, the flat form needs two tiles for the straddling rows 2 and 5
    // (+3 % at 16k, +4..10 % at 64k; same-box A/B, profiles/r02/k_flat_columns.txt).  TUNE_SEL_FLAT: -1 this rule, 0 never, 1 always.
    const int fmode = tuning(TUNE_SEL_FLAT);
    const bool flat = nt == 4 && P.h == 6 && (fmode > 0 || (fmode < 0 && (int64_t)P.S_kv <= (int64_t)384 * P.n));
    // key halves on different XCDs (see the kernel): needs the caller's workspace, no fused selector, and the walk of a row group long enough
    const size_t ks_need = (nt == 4 && !flat && !P.fuse_select) ? sel_attn_ksplit_workspace(dtype, P.h, P.Dk, P.Dv, P.S, P.S_kv, P.n, P.R) : 0;
    if (ks_need > 0 && P.ks_ws && P.ks_bytes >= ks_need)
        return dtype == NSA_DT_BF16 ? launch_blocks_t<__bf16, 4, false, true>(P, st) : launch_blocks_t<_Float16, 4, false, true>(P, st);
    if (dtype == NSA_DT_BF16) {
        if (flat) return launch_blocks_t<__bf16, 3, true>(P, st);
        if (nt == 4) return launch_blocks_t<__bf16, 4, false>(P, st);
        if (nt == 2) return launch_blocks_t<__bf16, 2, false>(P, st);
        return launch_blocks_t<__bf16, 1, false>(P, st);
    }
    if (flat) return launch_blocks_t<_Float16, 3, true>(P, st);
    if (nt == 4) return launch_blocks_t<_Float16, 4, false>(P, st);
    if (nt == 2) return launch_blocks_t<_Float16, 2, false>(P, st);
    return launch_blocks_t<_Float16, 1, false>(P, st);
}

}  // namespace nsa
