// MFMA selection-attention forward for gfx950 (bf16 / f16, Dk = Dv = D in {64,128}, h <= 16).
//
// Mapping (one wave64 = one query row (b,t,g); a 256-thread workgroup = 4 consecutive tokens of one
// (b,g), so the forced blocks -- block 0 and the local blocks -- are shared through the CU's L1):
//   * The h query heads of the GQA group are the 16 columns of a 16x16x32 MFMA (columns >= h are
//     zero padding), so every K/V byte fetched is shared by all heads of the group -- K/V are read
//     once per group, never per head (the reference's Triton kernels re-read them per head,
//     nsa/kernels/triton_sel_kernel/sel_fwd.py:143-238).
//   * The selected ranges are normalised to sorted disjoint segments (union semantics of
//     attention_kernels.py:721-732) and walked in 32-token tiles; tail tiles are masked.
//   * S^T[key, head] = K_tile[32 x D] . Q^T[D x 16]  : A = K rows from LDS (ds_read_b128, XOR
//     swizzled 16-B pieces), B = Q^T fragments held in registers for the whole row.
//   * softmax in fp32, exp2 domain.  Keys live in the accumulator registers and the 4 lane groups,
//     so the row sum needs no cross-lane work until the epilogue.  The running max is only raised
//     when a tile exceeds it by more than RESCALE_THR (log2 units): in the steady state a tile costs
//     8 fma + 8 exp + 8 add + 4 cvt per lane and no cross-lane instruction, and the O accumulators are
//     not touched by the VALU at all.  exp2 arguments stay <= RESCALE_THR, so p <= 256 (bf16 P, fp32 l,O).
//   * O^T[dv, head] += V^T[dv x 32 keys] . P^T[32 keys x 16] : B = P^T taken straight from the S^T
//     accumulators (cvt to bf16, no lane movement: the k index of the PV MFMA is mapped onto the
//     accumulator's key order), A = V^T read with ds_read_b64_tr_b16 (hardware transpose) from a
//     row-major, XOR-swizzled V tile.
//   * HBM/L2 -> LDS staging is register staged with one tile of prefetch: the global loads of tile
//     i+1 are issued before the MFMAs of tile i.  Every global load instruction covers whole 128-B
//     (D=64) / 256-B (D=128) rows (lanes row-linear, 16 B per lane; fragment-shaped K loads straight to
//     registers measured 15-20 % slower).  Full tiles use a scalar tile base + per-lane 32-bit offsets
//     computed once per row (no per-tile address arithmetic on the VALU).
// All LDS is wave private: no workgroup barrier anywhere in the kernel.
#include <stdlib.h>

#include "attn_mfma_tiles.hpp"
#include "sel_select_row.hpp"

namespace nsa {

// STAGE 0: tiles are staged global -> VGPR -> ds_write_b128 -> LDS (one tile of register prefetch).
// STAGE 1: tiles are written into LDS by LDS-DMA (buffer_load_dwordx4 ... lds, no VGPR / ds_write on the path); the
//          XOR swizzle is applied on the per-lane SOURCE offset because the DMA destination is lane-linear.
template <typename T, int D, bool SPLIT, int STAGE>
__global__ __launch_bounds__(256, 2) void sel_attn_fwd_mfma_kernel(SelAttnParams P, SelectParams SP, int cand) {
    using M = MfmaT<T>;
    using G_ = Geo<D>;
    using x8 = typename M::x8;
    using x4 = typename M::x4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id();
    const int wave = uniform((int)(threadIdx.x >> 6));
    const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
    const int nsplit = SPLIT ? P.nsplit : 1;
    int64_t row = SPLIT ? wid / nsplit : wid;
    const int sp = SPLIT ? (int)(wid % nsplit) : 0;
    if (!SPLIT && P.map_mode != 0) {
        // workgroup = 4 consecutive tokens of ONE (b,g).  map_mode 2 (B*G % 8 == 0): workgroups are dealt
        // round-robin over the 8 XCDs (blockIdx % 8 labels the XCD group), so give every XCD whole (b,g) pairs
        // and walk them one after the other: the K/V of one (b,g) (1 MiB at S=4096) then stays in that XCD's
        // 4 MiB L2.  Placement only changes speed, never results.
        const int W = (P.S + 3) >> 2;  // workgroups per (b,g)
        int bg, tc;
        if (P.map_mode == 2) {
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

    unsigned char *kl = smem + (size_t)wave * G_::WAVE_LDS;
    unsigned char *vl = kl + G_::TILE_BYTES;
    int *seg = (int *)(vl + G_::TILE_BYTES);

    const int h = P.h;
    const int g = (int)(row % P.G);
    const int b = (int)(row / ((int64_t)P.G * P.S));
    const unsigned char *Kb = (const unsigned char *)((const T *)P.K + (int64_t)b * P.ksb + (int64_t)g * P.ksg);
    const unsigned char *Vb = (const unsigned char *)((const T *)P.V + (int64_t)b * P.vsb + (int64_t)g * P.vsg);
    const T *Qr = (const T *)P.Q + row * (int64_t)h * D;
    const int64_t krowb = P.kss * 2, vrowb = P.vss * 2;  // row pitch in bytes

    int nseg, L;
    if (!SPLIT && P.fuse_select) {
        // top-n selection of this row from its group scores, in the attention kernel itself: the selector is VALU work of a
        // few hundred instructions per row, the attention is bound by the L2 gather -- run inside this launch it costs little
        // and the separate select kernel (8 % of the hot-path step) disappears.  Same arithmetic as select_topn_kernel.
        const int t = SP.t_rows ? SP.t_rows[row] : SP.t0 + (int)((row / P.G) % P.S);
        const float *pg = SP.p_grp + row * (int64_t)SP.S_sel;
        int ms = 0, me = 0;
        switch (cand) {
            case 1: select_topn_row_regs<1>(SP, pg, t, ms, me, (int *)kl); break;
            case 2: select_topn_row_regs<2>(SP, pg, t, ms, me, (int *)kl); break;
            case 4: select_topn_row_regs<4>(SP, pg, t, ms, me, (int *)kl); break;
            case 8: select_topn_row_regs<8>(SP, pg, t, ms, me, (int *)kl); break;
            default: select_topn_row_regs<16>(SP, pg, t, ms, me, (int *)kl); break;
        }
        if (lane < SP.W) {
            int32_t *out = SP.out + row * (int64_t)SP.W * 2;
            out[2 * lane] = ms;
            out[2 * lane + 1] = me;
        }
        L = normalise_ranges_lanes(ms, me, SP.W, P.S_kv, seg, &nseg);
    } else {
        L = normalise_ranges(P.ranges + row * (int64_t)P.n * 2, P.n, P.S_kv, seg, &nseg);
    }

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

    // ---- per-lane constants: global byte offsets inside a full tile, LDS write/read offsets.  Wherever the XOR
    // swizzle term does not depend on the unrolled index the offset is written as base + compile-time constant so
    // it folds into the instruction's immediate offset field (one VGPR instead of one per unrolled access).
    const int ld_row = lane / G_::PIECES;    // row within one wave-wide load
    const int ld_piece = lane % G_::PIECES;  // 16-B piece within the row
    const uint32_t koff0 = (uint32_t)(ld_row * krowb + ld_piece * 16);
    const uint32_t voff0 = (uint32_t)(ld_row * vrowb + ld_piece * 16);
    // LDS-DMA: lane l of load i lands at tile + i*1 KiB + 16 l, i.e. row i*RPI + ld_row, stored piece ld_piece; it must
    // fetch the source piece that the swizzled image keeps at that position (the swizzles are involutions)
    uint32_t kdma[G_::NLD], vdma[G_::NLD];
#pragma unroll
    for (int i = 0; i < G_::NLD; ++i) {
        const int r = i * G_::RPI + ld_row;
        kdma[i] = (uint32_t)(ld_row * krowb + ((ld_piece ^ G_::swz_k(r)) << 4));
        vdma[i] = (uint32_t)(ld_row * vrowb + (((((ld_piece >> 1) ^ G_::swz_v(r)) << 1) | (ld_piece & 1)) << 4));
    }
    constexpr bool WR_IMM = (G_::RPI % G_::PIECES) == 0 && (G_::RPI % 8) == 0;  // swizzle of row i*RPI+ld_row == swizzle of ld_row
    uint32_t kwr[G_::NLD], vwr[G_::NLD];
#pragma unroll
    for (int i = 0; i < G_::NLD; ++i) {
        const int r = i * G_::RPI + ld_row;
        const int rs = WR_IMM ? ld_row : r;
        kwr[i] = i * G_::RPI * G_::ROWB + ld_row * G_::ROWB + ((ld_piece ^ G_::swz_k(rs)) << 4);
        vwr[i] = i * G_::RPI * G_::ROWB + ld_row * G_::ROWB + ((((ld_piece >> 1) ^ G_::swz_v(rs)) << 5) | ((ld_piece & 1) << 4));
    }
    // reads: row 16u + rho (K) / 16u + 4q + qq (V): both swizzles are invariant under +16 rows
    uint32_t krd0[G_::KSTEPS], vrd0[G_::MT];
#pragma unroll
    for (int s = 0; s < G_::KSTEPS; ++s) krd0[s] = rho * G_::ROWB + (((4 * s + q) ^ G_::swz_k(rho)) << 4);
    {
        const int qq = rho >> 2, pp = rho & 3, r = 4 * q + qq;
#pragma unroll
        for (int m = 0; m < G_::MT; ++m) vrd0[m] = r * G_::ROWB + ((m ^ G_::swz_v(r)) << 5) + 8 * pp;
    }

    // ---- tile iterator over the segments (all wave-uniform scalars)
    int it_seg = -1, it_start = 0, it_len = 0, it_pos = 0, it_cnt = 0;
    auto next_tile = [&](int &tok0, int &nvalid) -> bool {
        while (true) {
            if (it_pos < it_len) {
                tok0 = it_start + it_pos;
                nvalid = min(32, it_len - it_pos);
                it_pos += 32;
                if (SPLIT) {
                    const bool mine = (it_cnt % nsplit) == sp;
                    ++it_cnt;
                    if (!mine) continue;
                }
                return true;
            }
            if (++it_seg >= nseg) return false;
            it_start = uniform(seg[2 * it_seg]);
            it_len = uniform(seg[2 * it_seg + 3]) - uniform(seg[2 * it_seg + 1]);
            it_pos = 0;
        }
    };

    // K/V of this (b,g) as buffer resources: loads are `buffer_load_dwordx4 v, voffset(VGPR), srsrc, soffset(SGPR)` --
    // the per-tile part of the address is scalar, the per-lane part a loop-invariant VGPR: no VALU address math.
    // The descriptor is built from wave-uniform values only (readfirstlane'd halves), so no waterfall loop is emitted.
    typedef __attribute__((ext_vector_type(4))) unsigned int bu32x4;
    auto make_rsrc = [&](const unsigned char *base, int64_t bytes) {
        const uint64_t a = (uint64_t)base;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), (short)0,
                                                 __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
    };
    const auto krs = make_rsrc(Kb, (int64_t)(P.S_kv - 1) * krowb + G_::ROWB);
    const auto vrs = make_rsrc(Vb, (int64_t)(P.S_kv - 1) * vrowb + G_::ROWB);
    // readfirstlane pins these in SGPRs (a soffset the compiler keeps in a VGPR costs a waterfall loop per load)
    const int krowb32 = uniform((int)krowb), vrowb32 = uniform((int)vrowb);
    const int kstep = uniform(G_::RPI * (int)krowb), vstep = uniform(G_::RPI * (int)vrowb);
    u32x4 kreg[G_::NLD], vreg[G_::NLD];
    auto issue_loads = [&](int tok0, int nvalid) {
        const int ks = uniform(tok0 * krowb32), vs = uniform(tok0 * vrowb32);  // scalar byte offsets of the tile
        if (nvalid == 32) {
#pragma unroll
            for (int i = 0; i < G_::NLD; ++i) {
                kreg[i] = __builtin_bit_cast(u32x4, (bu32x4)__builtin_amdgcn_raw_buffer_load_b128(krs, koff0, ks + i * kstep, 0));
                vreg[i] = __builtin_bit_cast(u32x4, (bu32x4)__builtin_amdgcn_raw_buffer_load_b128(vrs, voff0, vs + i * vstep, 0));
            }
        } else {  // tail tile: clamp the row so nothing outside the segment is touched
#pragma unroll
            for (int i = 0; i < G_::NLD; ++i) {
                const int r = min(i * G_::RPI + ld_row, nvalid - 1);
                kreg[i] = __builtin_bit_cast(u32x4, (bu32x4)__builtin_amdgcn_raw_buffer_load_b128(krs, r * krowb32 + ld_piece * 16, ks, 0));
                vreg[i] = __builtin_bit_cast(u32x4, (bu32x4)__builtin_amdgcn_raw_buffer_load_b128(vrs, r * vrowb32 + ld_piece * 16, vs, 0));
            }
        }
    };

    auto issue_dma = [&](int tok0, int nvalid) {
#if defined(__HIP_DEVICE_COMPILE__)  // hipcc's host pass drops the whole kernel stub if it sees this builtin
        typedef __attribute__((address_space(3))) void lds_void;
        const int ks = uniform(tok0 * krowb32), vs = uniform(tok0 * vrowb32);
        if (nvalid == 32) {
#pragma unroll
            for (int i = 0; i < G_::NLD; ++i) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (lds_void *)(kl + i * 1024), 16, kdma[i], ks + i * kstep, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (lds_void *)(vl + i * 1024), 16, vdma[i], vs + i * vstep, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < G_::NLD; ++i) {
                const int r = i * G_::RPI + ld_row;
                const int rc = min(r, nvalid - 1);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (lds_void *)(kl + i * 1024), 16,
                                                         rc * krowb32 + ((ld_piece ^ G_::swz_k(r)) << 4), ks, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (lds_void *)(vl + i * 1024), 16,
                                                         rc * vrowb32 + (((((ld_piece >> 1) ^ G_::swz_v(r)) << 1) | (ld_piece & 1)) << 4), vs, 0, 0);
            }
        }
#else
        (void)tok0;
        (void)nvalid;
#endif
    };

    f32x4 o[G_::MT];
#pragma unroll
    for (int m = 0; m < G_::MT; ++m) o[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrun = -INFINITY, lrun = 0.f;
    const float c2 = P.scale * LOG2E;

    int tok0 = 0, nvalid = 0;
    bool have = next_tile(tok0, nvalid);
    if (have) {
        if (STAGE == 1) issue_dma(tok0, nvalid);
        else issue_loads(tok0, nvalid);
    }

    while (have) {
        const int cur_nvalid = nvalid;
        x8 kfr[2][G_::KSTEPS];
        x4 vfr[2][G_::MT];
        if (STAGE == 1) {
            // the DMA of this tile was issued one iteration ago; LDS-DMA completion is a vmcnt event
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int s = 0; s < G_::KSTEPS; ++s) kfr[u][s] = *(const x8 *)(kl + krd0[s] + u * 16 * G_::ROWB);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int m = 0; m < G_::MT; ++m) vfr[u][m] = M::tr(vl + vrd0[m] + u * 16 * G_::ROWB);
            have = next_tile(tok0, nvalid);
            // every fragment is in registers before the tile buffer is handed back to the DMA engine
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (have) issue_dma(tok0, nvalid);
        } else {
            // ---- stage registers -> LDS (swizzled)
#pragma unroll
            for (int i = 0; i < G_::NLD; ++i) {
                *(u32x4 *)(kl + kwr[i]) = kreg[i];
                *(u32x4 *)(vl + vwr[i]) = vreg[i];
            }
            // ---- prefetch the next tile while this one is consumed
            have = next_tile(tok0, nvalid);
            if (have) issue_loads(tok0, nvalid);
            wave_lds_fence();
        }

        // ---- S^T = K . Q^T
        f32x4 sacc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            sacc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < G_::KSTEPS; ++s) {
                if (STAGE == 1) sacc[u] = M::mma(kfr[u][s], qf[s], sacc[u]);
                else sacc[u] = M::mma(*(const x8 *)(kl + krd0[s] + u * 16 * G_::ROWB), qf[s], sacc[u]);
            }
        }
        // ---- softmax, exp2 domain, deferred max; key of sacc[u][j] is 16u + 4q + j
        float x[8];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) x[4 * u + j] = fmaf(sacc[u][j], c2, -mrun);
        const float tmax = fmaxf(fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])), fmaxf(fmaxf(x[4], x[5]), fmaxf(x[6], x[7])));
        // slow path: the running max must be raised (always for the first tile: mrun = -inf -> x = +inf), or the
        // tile is a masked tail tile (the mask is only applied here, the common path carries no select)
        if (__any(!(tmax <= RESCALE_THR)) || cur_nvalid != 32) {
            asm volatile("; slow path (keeps the compiler from if-converting it into the common path)" ::: "memory");
            float vmax = -INFINITY;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = (16 * u + 4 * q + j < cur_nvalid) ? sacc[u][j] * c2 : -INFINITY;
                    x[4 * u + j] = v;
                    vmax = fmaxf(vmax, v);
                }
            vmax = fmaxf(vmax, __shfl_xor(vmax, 16, 64));
            vmax = fmaxf(vmax, __shfl_xor(vmax, 32, 64));
            const float mnew = fmaxf(mrun, vmax);  // finite: every tile has at least one valid key
            const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
            mrun = mnew;
            lrun *= alpha;
#pragma unroll
            for (int m = 0; m < G_::MT; ++m) o[m] *= alpha;
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] -= mnew;
        }
        float psum = 0.f;
        x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float pe = __builtin_amdgcn_exp2f(x[j]);
            psum += pe;
            pf[j] = Elt<T>::from_f(pe);
        }
        lrun += psum;

        // ---- O^T += V^T . P^T   (k index j<4 -> key 4q+j, j>=4 -> key 16+4q+(j-4))
#pragma unroll
        for (int m = 0; m < G_::MT; ++m) {
            x4 lo, hi;
            if (STAGE == 1) {
                lo = vfr[0][m];
                hi = vfr[1][m];
            } else {
                lo = M::tr(vl + vrd0[m]);
                hi = M::tr(vl + vrd0[m] + 16 * G_::ROWB);
            }
            x8 a;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = lo[j];
                a[4 + j] = hi[j];
            }
            o[m] = M::mma(a, pf, o[m]);
        }
        if (STAGE == 0) wave_lds_fence();  // LDS tile is rewritten at the top of the next iteration
    }

    // ---- epilogue
    float ltot = lrun + __shfl_xor(lrun, 16, 64);
    ltot += __shfl_xor(ltot, 32, 64);
    if (SPLIT) {
        // partial record [row][sp][head][PART_PAD + D]: m (log2 domain), l, unnormalised O
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
    unsigned grid = (unsigned)((waves + 3) / 4);
    P.map_mode = 0;
    if (!split && P.S >= 16) {
        const int64_t nbg = P.R / P.S;  // B*G
        const int64_t W = (P.S + 3) / 4;
        if (nbg * W < (int64_t)1 << 31) {
            P.map_mode = (nbg % 8 == 0) ? 2 : 1;
            grid = (unsigned)(nbg * W);
        }
    }
    if (const int m = tuning(TUNE_ATTN_MAP); m >= 0) {  // A/B switch for measurements: force a mapping
        if (!split && m == 0) {
            P.map_mode = 0;
            grid = (unsigned)((waves + 3) / 4);
        }
        if (!split && m == 1 && P.map_mode == 2) P.map_mode = 1;
    }
    const int stage = tuning(TUNE_ATTN_STAGE);  // A/B switch: 0 = register staging, 1 = LDS-DMA  // default: LDS-DMA (measured 3-5 % faster than register staging at S<=16k)
    void (*k)(SelAttnParams, SelectParams, int) = nullptr;
    if (split && stage == 1) k = sel_attn_fwd_mfma_kernel<T, D, true, 1>;
    else if (split) k = sel_attn_fwd_mfma_kernel<T, D, true, 0>;
    else if (stage == 1) k = sel_attn_fwd_mfma_kernel<T, D, false, 1>;
    else k = sel_attn_fwd_mfma_kernel<T, D, false, 0>;
    if (lds > 64 * 1024) NSA_HIP_TRY(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SelectParams SP{};
    int cand = 0;
    if (P.fuse_select) {
        NSA_CHECK_ARG(!split && P.select != nullptr, "fused selection needs the non-split route");
        SP = *(const SelectParams *)P.select;
        const int c = (SP.S_sel + 63) / 64;
        NSA_CHECK_ARG(c <= 16 && SP.W <= 64, "fused selection: S_sel <= 1024 and at most 64 ranges per row");
        cand = c <= 1 ? 1 : c <= 2 ? 2 : c <= 4 ? 4 : c <= 8 ? 8 : 16;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, P, SP, cand);
    NSA_LAUNCH_CHECK("sel_attn_fwd_mfma");
    if (split && !P.defer_combine) {
        hipLaunchKernelGGL((sel_attn_combine_kernel<T, D>), dim3((unsigned)((P.R * P.h + 3) / 4)), dim3(256), 0, st, P);
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
    NSA_CHECK_ARG((int64_t)P.S_kv * P.kss * 2 < ((int64_t)1 << 31) && (int64_t)P.S_kv * P.vss * 2 < ((int64_t)1 << 31),
                  "MFMA kernel: one (b,g) K/V slab must be smaller than 2 GiB (buffer addressing)");
    if (!(P.part != nullptr && P.nsplit > 1)) {  // many rows: rows of a wave share the K/V tiles they both selected
        if (tuning(TUNE_SEL_ROWS) < 0) {  // block form (64-key blocks, several column tiles per wave) unless a row form is forced
            const int nb = sel_attn_blocks_nt(dtype, P.h, P.Dk, P.Dv, P.S, P.S_kv, P.n, P.R, P.kss, P.vss);
            if (nb > 0) return launch_sel_attn_blocks_mfma(P, dtype, nb, st);
        }
        int nt = 1;
        const int tpw = sel_attn_rows_tpw(dtype, P.h, P.Dk, P.Dv, P.S, P.S_kv, P.n, P.R, &nt);
        if (tpw > 0) return launch_sel_attn_rows_mfma(P, dtype, tpw, nt, st);
    }
    if (dtype == NSA_DT_BF16) return P.Dk == 64 ? launch_mfma_t<__bf16, 64>(P, st) : launch_mfma_t<__bf16, 128>(P, st);
    return P.Dk == 64 ? launch_mfma_t<_Float16, 64>(P, st) : launch_mfma_t<_Float16, 128>(P, st);
}

}  // namespace nsa
