// Query-tile form of the MFMA selection attention (prefill / training forward; bf16 / f16, Dk = Dv in {64, 128}, h <= 8 for row sharing).
//
// sel_attn_mfma.hip gives every query row its own wave: the h heads of the row fill h of the 16 MFMA columns and every
// selected K/V tile is fetched once PER ROW.  Neighbouring rows of one (b,g) select largely the same blocks (block 0, the
// local blocks, and -- while t < n*l' -- every block), so here ONE wave owns TPW = 48/h consecutive rows (h = 6: 8 rows x 6
// heads = 48 slots = 3 column tiles, every MFMA column used) and walks the UNION of their selected 32-key tiles: a tile
// is brought into LDS once for all TPW rows and each row masks the keys it did not select.  At S = 4096 the union of 8
// rows is ~0.4x the sum of their selections, which is the factor the L2 gather shrinks by.
//
// Per wave: (1) the rows' ranges go to LDS (from the fused selector, sel_select_row.hpp, or from the ranges tensor);
// (2) two bitmaps per row over 32-key tiles -- `touch` (some key of the tile selected) and `full` (a single range covers
// the whole tile) -- are built with LDS atomics, one (row, range) pair per lane; their OR over the rows is the tile
// schedule; (3) tiles are walked in ascending order with one tile of LDS-DMA prefetch.  A tile that every row covers in
// full takes the mask-free path of the band kernel; otherwise each slot applies a 32-bit key mask (all ones, zero, or
// -- for the few partially covered tiles, e.g. the causal clamp at t+1 -- a mask rebuilt from the row's ranges), and
// column tiles none of whose rows touch the key tile are skipped.  Union semantics of overlapping ranges
// (attention_kernels.py:721-732) fall out of the bitmaps: no sorting / merging of ranges is needed.
// Softmax (deferred max), S^T / O^T formulation, tile geometry and the epilogue are those of band_attn_mfma.hip.
#include <stdlib.h>

#include "attn_mfma_tiles.hpp"
#include "sel_select_row.hpp"

namespace nsa {

__device__ __forceinline__ unsigned bit_span(int lo, int hi) {  // bits lo..hi inclusive, 0 <= lo <= hi <= 31
    const unsigned up = hi >= 31 ? 0xffffffffu : ((1u << (hi + 1)) - 1u);
    return up & ~((1u << lo) - 1u);
}

// one v_max3_f32 (fmaxf makes hipcc canonicalise every MFMA-derived operand with a v_max x,x first)
__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ void lds_or(unsigned *p, unsigned v) {  // ds_or_b32: stays in the wave's in-order DS queue
    typedef __attribute__((address_space(3))) unsigned lds_u32;
    __hip_atomic_fetch_or((lds_u32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// RS: row sums of P on the matrix pipe (l += ones . P^T, one MFMA per tile instead of 8 v_add; see sel_attn_blocks_mfma.hip)
template <typename T, int D, int NT, bool RS>
__global__ __launch_bounds__(256, 2) void sel_attn_rows_mfma_kernel(SelAttnParams P, SelectParams SP, int cand) {
    using M = MfmaT<T>;
    using G_ = Geo<D>;
    using x8 = typename M::x8;
    using x4 = typename M::x4;
    constexpr int KS = G_::KSTEPS, MT = G_::MT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id();
    const int wave = uniform((int)(threadIdx.x >> 6));
    const int tpw = P.tpw, h = P.h, NW = P.nw, n = P.n;
    const int ngrp = (P.S + tpw - 1) / tpw;  // row groups per (b,g)
    const int nbg = (int)(P.R / P.S);
    const int W4 = (ngrp + 3) >> 2;  // workgroups per (b,g)
    int bg, tc;
    if (P.map_mode == 2) {  // whole (b,g) pairs per XCD (workgroups go round-robin over the 8 XCDs)
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
    unsigned char *vl = kl + G_::TILE_BYTES;
    int *rg = (int *)(vl + G_::TILE_BYTES);                   // [tpw][n][2] clamped ranges
    unsigned *fullw = (unsigned *)(rg + ((2 * tpw * n + 3) & ~3));  // [tpw][NW]
    unsigned *touchw = fullw + tpw * NW;                      // [tpw][NW]
    unsigned *kmask = touchw + tpw * NW;                      // [16]

    // ---- per-slot constants; the Q loads are issued first so that their latency runs under the range / bitmap work below
    const int rho = lane & 15, q = lane >> 4;
    int tokn[NT];
    unsigned rowbit[NT];  // bit of the slot's row in the ownership masks, 0 for an unused slot
    int64_t orow[NT];  // (row * h + head) of the slot, -1 = unused slot
    unsigned nmask[NT];  // rows that have a slot in column tile n (wave uniform)
    x8 qf[NT][KS];
#pragma unroll
    for (int nn = 0; nn < NT; ++nn) {
        const int slot = 16 * nn + rho, tok = slot / h, head = slot - tok * h;
        const bool used = tok < ntok;
        tokn[nn] = tok;
        rowbit[nn] = used ? (1u << tok) : 0u;
        orow[nn] = used ? ((((int64_t)b * P.S + tw0 + tok) * P.G + g) * h + head) : -1;
        const int r_lo = (16 * nn) / h, r_hi = min((16 * nn + 15) / h, ntok - 1);
        nmask[nn] = r_lo <= r_hi ? bit_span(r_lo, r_hi) : 0u;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            u32x4 raw = {0u, 0u, 0u, 0u};
            if (used) raw = *(const u32x4 *)((const T *)P.Q + orow[nn] * D + 32 * s + 8 * q);
            qf[nn][s] = __builtin_bit_cast(x8, raw);
        }
    }
    const unsigned usedmask = bit_span(0, ntok - 1);

    // ---- (1) bitmaps cleared, ranges of the rows -> LDS
    for (int i = lane; i < 2 * tpw * NW + 16; i += 64) fullw[i] = 0u;
    for (int r = 0; r < ntok; ++r) {
        const int64_t row = ((int64_t)b * P.S + tw0 + r) * P.G + g;
        int s = 0, e = 0;
        if (P.fuse_select) {
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
            }
        } else if (lane < n) {
            const int32_t *in = P.ranges + (row * n + lane) * 2;
            s = in[0];
            e = in[1];
        }
        if (lane < n) {
            s = min(max(s, 0), P.S_kv);
            e = min(max(e, s), P.S_kv);
            rg[2 * (r * n + lane)] = s;
            rg[2 * (r * n + lane) + 1] = e;
        }
    }
    wave_lds_fence();

    // ---- (2) tile bitmaps: one (row, range) pair per lane
    for (int p = lane; p < ntok * n; p += 64) {
        const int r = p / n;
        const int s = rg[2 * p], e = rg[2 * p + 1];
        if (e > s) {
            const int ta = s >> 5, tb = (e - 1) >> 5;      // tiles touched
            const int fa = (s + 31) >> 5, fb = (e >> 5) - 1;  // tiles covered completely: fa..fb
            for (int w = ta >> 5; w <= (tb >> 5); ++w) {
                const int base = 32 * w;
                lds_or(&touchw[r * NW + w], bit_span(max(ta, base) - base, min(tb, base + 31) - base));
                const int flo = max(fa, base), fhi = min(fb, base + 31);
                if (flo <= fhi) lds_or(&fullw[r * NW + w], bit_span(flo - base, fhi - base));
            }
        }
    }
    wave_lds_fence();
    // union over the rows = the tile schedule; kept in registers (lane w holds word w, NW <= 128): walking it costs no LDS round trips
    unsigned u0 = 0u, u1 = 0u;
    for (int r = 0; r < ntok; ++r) {
        if (lane < NW) u0 |= touchw[r * NW + lane];
        if (lane + 64 < NW) u1 |= touchw[r * NW + 64 + lane];
    }
    unsigned long long nz0 = __ballot(u0 != 0u), nz1 = __ballot(u1 != 0u);

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
    [[maybe_unused]] const int kstep = uniform(G_::RPI * (int)krowb), vstep = uniform(G_::RPI * (int)vrowb);
    const int ld_row = lane / G_::PIECES, ld_piece = lane % G_::PIECES;
    uint32_t kdma[G_::NLD], vdma[G_::NLD];
#pragma unroll
    for (int i = 0; i < G_::NLD; ++i) {
        const int r = i * G_::RPI + ld_row;
        kdma[i] = (uint32_t)(ld_row * krowb + ((ld_piece ^ G_::swz_k(r)) << 4));
        vdma[i] = (uint32_t)(ld_row * vrowb + (((((ld_piece >> 1) ^ G_::swz_v(r)) << 1) | (ld_piece & 1)) << 4));
    }
    uint32_t krd0[KS], vrd0[MT];
#pragma unroll
    for (int s = 0; s < KS; ++s) krd0[s] = rho * G_::ROWB + (((4 * s + q) ^ G_::swz_k(rho)) << 4);
    {
        const int qq = rho >> 2, pp = rho & 3, r = 4 * q + qq;
#pragma unroll
        for (int m = 0; m < MT; ++m) vrd0[m] = r * G_::ROWB + ((m ^ G_::swz_v(r)) << 5) + 8 * pp;
    }
    // a tile of 32 keys starting at tok0 -> wave-private LDS (swizzle applied on the source side); rows past the end of K/V
    // re-read the last row (they are masked: a tile reaching past S_kv is never `full`)
    auto issue_dma = [&](int tok0) {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef __attribute__((address_space(3))) void lds_void;
        const int ks = uniform(tok0 * krowb32), vs = uniform(tok0 * vrowb32);
        if (tok0 + 32 <= P.S_kv) {
#pragma unroll
            for (int i = 0; i < G_::NLD; ++i) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (lds_void *)(kl + i * 1024), 16, kdma[i], ks + i * kstep, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (lds_void *)(vl + i * 1024), 16, vdma[i], vs + i * vstep, 0, 0);
            }
        } else {
            const int last = P.S_kv - 1 - tok0;
#pragma unroll
            for (int i = 0; i < G_::NLD; ++i) {
                const int r = i * G_::RPI + ld_row;
                const int rc = min(r, last);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (lds_void *)(kl + i * 1024), 16,
                                                         rc * krowb32 + ((ld_piece ^ G_::swz_k(r)) << 4), ks, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (lds_void *)(vl + i * 1024), 16,
                                                         rc * vrowb32 + (((((ld_piece >> 1) ^ G_::swz_v(r)) << 1) | (ld_piece & 1)) << 4), vs, 0, 0);
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
        // unused slots carry +inf so that their (zero) scores never trigger the max-raising path
        mrun[nn] = orow[nn] >= 0 ? -INFINITY : INFINITY;
        lrun[nn] = 0.f;
    }
    const float c2 = P.scale * LOG2E;
    x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = Elt<T>::from_f(1.f);

    // ---- (3) tile schedule = set bits of the union bitmap, ascending (wave-uniform scalars)
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
    int cur = next_tile();
    if (cur >= 0) issue_dma(32 * cur);
    int cw = -1;
    unsigned fw = 0u, tw = 0u;

    while (cur >= 0) {
        const int nxt = next_tile();
        const int tok0 = 32 * cur;
        x8 kfr[2][KS];
        x8 va[MT];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // LDS-DMA completion is a vmcnt event
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s = 0; s < KS; ++s) kfr[u][s] = *(const x8 *)(kl + krd0[s] + u * 16 * G_::ROWB);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const x4 lo = M::tr(vl + vrd0[m]), hi = M::tr(vl + vrd0[m] + 16 * G_::ROWB);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                va[m][j] = lo[j];
                va[m][4 + j] = hi[j];
            }
        }
        // ownership of this tile: bit r of fullm / touchm = row r covers it completely / selected at least one of its keys
        if ((cur >> 5) != cw) {  // lane r < ntok caches row r's bitmap words of the current 32-tile group (refreshed once per word)
            cw = cur >> 5;
            if (lane < ntok) {
                fw = fullw[lane * NW + cw];
                tw = touchw[lane * NW + cw];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments are in registers: the buffers may be refilled
        __builtin_amdgcn_sched_barrier(0);
        if (nxt >= 0) issue_dma(32 * nxt);
        const unsigned fullm = (unsigned)__ballot((fw >> (cur & 31)) & 1u);
        const unsigned touchm = (unsigned)__ballot((tw >> (cur & 31)) & 1u);
        const unsigned partm = touchm & ~fullm;

        if (NT > 1 && fullm == usedmask) {  // (NT = 1 keeps ONE path: the predicate costs 8 v_cndmask, a second path costs registers)
            float x[NT][8];
            // ---- every row selected the whole tile: mask-free path, straight-line over the column tiles
            f32x4 sacc[NT][2];
#pragma unroll
            for (int nn = 0; nn < NT; ++nn)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    sacc[nn][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < KS; ++s) sacc[nn][u] = M::mma(kfr[u][s], qf[nn][s], sacc[nn][u]);
                }
            float tmax = -INFINITY;
#pragma unroll
            for (int nn = 0; nn < NT; ++nn) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[nn][4 * u + j] = fmaf(sacc[nn][u][j], c2, -mrun[nn]);
                tmax = fmaxf(tmax, fmaxf(fmaxf(fmaxf(x[nn][0], x[nn][1]), fmaxf(x[nn][2], x[nn][3])),
                                         fmaxf(fmaxf(x[nn][4], x[nn][5]), fmaxf(x[nn][6], x[nn][7]))));
            }
            if (__any(!(tmax <= RESCALE_THR))) {
#pragma unroll
                for (int nn = 0; nn < NT; ++nn) {
                    float vmax = -INFINITY;
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float v = orow[nn] >= 0 ? sacc[nn][u][j] * c2 : -INFINITY;
                            x[nn][4 * u + j] = v;
                            vmax = fmaxf(vmax, v);
                        }
                    vmax = fmaxf(vmax, __shfl_xor(vmax, 16, 64));
                    vmax = fmaxf(vmax, __shfl_xor(vmax, 32, 64));
                    const float mnew = fmaxf(mrun[nn], vmax);
                    const float msub = (mnew == -INFINITY) ? 0.f : mnew;
                    const float alpha = (mnew == mrun[nn]) ? 1.f : __builtin_amdgcn_exp2f(mrun[nn] - msub);
                    mrun[nn] = mnew;
                    lrun[nn] *= alpha;
#pragma unroll
                    for (int m = 0; m < MT; ++m) o[nn][m] *= alpha;
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[nn][j] -= msub;
                }
            }
#pragma unroll
            for (int nn = 0; nn < NT; ++nn) {
                float psum = 0.f;
                x8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float pe = __builtin_amdgcn_exp2f(x[nn][j]);
                    if constexpr (!RS) psum += pe;
                    pf[j] = Elt<T>::from_f(pe);
                }
                if constexpr (RS) psum = M::mma(ones, pf, (f32x4){0.f, 0.f, 0.f, 0.f})[0];
                lrun[nn] += psum;
#pragma unroll
                for (int m = 0; m < MT; ++m) o[nn][m] = M::mma(va[m], pf, o[nn][m]);
            }
        } else {
            // ---- mixed ownership: per-slot key masks; column tiles none of whose rows touch the tile are skipped
            if (partm) {  // partially covered by some row: rebuild that row's 32-key mask from its ranges
                if (lane < 16) kmask[lane] = 0u;
                wave_lds_fence();
                for (int p = lane; p < ntok * n; p += 64) {
                    const int r = p / n;
                    const int lo = max(rg[2 * p], tok0) - tok0, hi = min(rg[2 * p + 1], tok0 + 32) - tok0;
                    if (((partm >> r) & 1u) && hi > lo) lds_or(&kmask[r], bit_span(lo, hi - 1));
                }
                wave_lds_fence();
            }
            // one column tile at a time (short live ranges: the mixed path must not cost the kernel its second wave per SIMD)
#pragma unroll
            for (int nn = 0; nn < NT; ++nn) {
                if (!(touchm & nmask[nn])) continue;
                const bool on = (fullm & rowbit[nn]) != 0u;
                unsigned km = on ? 0xffffffffu : 0u;
                if (partm) {
                    if (partm & rowbit[nn]) km = kmask[tokn[nn]];
                    km >>= 4 * q;  // bit 16u + j of km = key 16u + 4q + j of the tile
                }
                f32x4 sacc[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    sacc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < KS; ++s) sacc[u] = M::mma(kfr[u][s], qf[nn][s], sacc[u]);
                }
                float xs[8];
                float tmax = -INFINITY;
                if (partm == 0u) {  // every slot is all-on or all-off: the softmax offset carries the mask (s*c2 - inf = -inf)
                    const float mneg = on ? -mrun[nn] : -INFINITY;
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int j = 0; j < 4; ++j) xs[4 * u + j] = fmaf(sacc[u][j], c2, mneg);
                    tmax = max3f(max3f(xs[0], xs[1], xs[2]), max3f(xs[3], xs[4], xs[5]), max3f(xs[6], xs[7], xs[7]));
                } else {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float v = ((km >> (16 * u + j)) & 1u) ? fmaf(sacc[u][j], c2, -mrun[nn]) : -INFINITY;
                            xs[4 * u + j] = v;
                            tmax = fmaxf(tmax, v);
                        }
                }
                if (__any(!(tmax <= RESCALE_THR))) {
                    float vmax = -INFINITY;
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float v = ((km >> (16 * u + j)) & 1u) ? sacc[u][j] * c2 : -INFINITY;
                            xs[4 * u + j] = v;
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
                    for (int j = 0; j < 8; ++j) xs[j] -= msub;
                }
                float psum = 0.f;
                x8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float pe = __builtin_amdgcn_exp2f(xs[j]);
                    if constexpr (!RS) psum += pe;
                    pf[j] = Elt<T>::from_f(pe);
                }
                if constexpr (RS) psum = M::mma(ones, pf, (f32x4){0.f, 0.f, 0.f, 0.f})[0];
                lrun[nn] += psum;
#pragma unroll
                for (int m = 0; m < MT; ++m) o[nn][m] = M::mma(va[m], pf, o[nn][m]);
            }
        }
        cur = nxt;
    }

    // ---- epilogue
#pragma unroll
    for (int nn = 0; nn < NT; ++nn) {
        float ltot = lrun[nn];
        if constexpr (!RS) {
            ltot += __shfl_xor(ltot, 16, 64);
            ltot += __shfl_xor(ltot, 32, 64);
        }
        if (orow[nn] < 0) continue;
        const float inv = ltot > 0.f ? 1.f / ltot : 0.f;
        T *Or = (T *)P.O + orow[nn] * D;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            x4 ov;
#pragma unroll
            for (int j = 0; j < 4; ++j) ov[j] = Elt<T>::from_f(o[nn][m][j] * inv);
            *(x4 *)(Or + 16 * m + 4 * q) = ov;
        }
        if (P.lse && q == 0) P.lse[orow[nn]] = ltot > 0.f ? (mrun[nn] + __builtin_amdgcn_logf(ltot)) * LN2 : -INFINITY;
    }
}

// ---- host side ----------------------------------------------------------------------------
// Rows per wave for a shape (0 = not covered: use the one-row-per-wave kernel).  *nt = column tiles per wave.
//   NT = 1: 16/h rows share one MFMA column tile (h = 6: a pair of rows, 12 of 16 columns used).  A union tile costs what a tile
//           of the one-row kernel costs, so the gain is the union/sum ratio of the rows' selections -- never a loss.
//   NT = 3: 48/h rows; pays off only while (nearly) every row selects (nearly) every tile, i.e. short contexts (S_kv <= ~n*l').
int sel_attn_rows_tpw(int dtype, int h, int Dk, int Dv, int S, int S_kv, int n, int64_t R, int *nt) {
    *nt = 1;
    if (!(dtype == NSA_DT_BF16 || dtype == NSA_DT_F16) || Dk != Dv || (Dk != 64 && Dk != 128) || h < 1 || h > 16) return 0;
    if (n < 1 || n > 64 || S_kv < 1 || S_kv > 131072) return 0;
    const int mode = tuning(TUNE_SEL_ROWS);  // 0 = off, 1 = pairs (NT 1), 3 = query tiles (NT 3); -1 = automatic
    if (mode == 0) return 0;
    int want_nt = 1;
    if (Dk == 64 && (mode == 3 || (mode < 0 && S_kv <= 1536 && R >= 16384 && h >= 3))) want_nt = 3;
    if (want_nt == 3 && h < 3) want_nt = 1;
    const int tpw = (16 * want_nt) / h;
    if (tpw < 2 || S < 2 * tpw) return 0;
    *nt = want_nt;
    return tpw;
}

template <typename T, int D, int NT>
static int launch_rows_t(const SelAttnParams &P0, int tpw, hipStream_t st) {
    SelAttnParams P = P0;
    P.tpw = tpw;
    P.nw = ((P.S_kv + 31) / 32 + 31) / 32;
    P.nsplit = 1;
    P.part = nullptr;
    const int rg_ints = (2 * tpw * P.n + 3) & ~3;
    const int bm_ints = (2 * tpw * P.nw + 16 + 3) & ~3;
    P.wave_lds = 2 * Geo<D>::TILE_BYTES + 4 * (rg_ints + bm_ints);
    const size_t lds = 4 * (size_t)P.wave_lds;
    NSA_CHECK_ARG(lds <= 160 * 1024, "sel_attn_rows: %zu B of LDS needed", lds);
    const int64_t nbg = P.R / P.S;
    const int64_t ngrp = (P.S + tpw - 1) / tpw;
    const int64_t W4 = (ngrp + 3) / 4;
    NSA_CHECK_ARG(nbg * W4 < (int64_t)1 << 31, "sel_attn_rows: grid too large");
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
    void (*k)(SelAttnParams, SelectParams, int) =
        tuning(TUNE_SEL_ROWSUM) ? sel_attn_rows_mfma_kernel<T, D, NT, true> : sel_attn_rows_mfma_kernel<T, D, NT, false>;
    if (lds > 64 * 1024) NSA_HIP_TRY(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)(nbg * W4)), dim3(256), lds, st, P, SP, cand);
    NSA_LAUNCH_CHECK("sel_attn_rows_mfma");
    return NSA_OK;
}

int launch_sel_attn_rows_mfma(const SelAttnParams &P, int dtype, int tpw, int nt, hipStream_t st) {
    if (P.Dk == 128) return dtype == NSA_DT_BF16 ? launch_rows_t<__bf16, 128, 1>(P, tpw, st) : launch_rows_t<_Float16, 128, 1>(P, tpw, st);
    if (nt == 3) return dtype == NSA_DT_BF16 ? launch_rows_t<__bf16, 64, 3>(P, tpw, st) : launch_rows_t<_Float16, 64, 3>(P, tpw, st);
    return dtype == NSA_DT_BF16 ? launch_rows_t<__bf16, 64, 1>(P, tpw, st) : launch_rows_t<_Float16, 64, 1>(P, tpw, st);
}

}  // namespace nsa
