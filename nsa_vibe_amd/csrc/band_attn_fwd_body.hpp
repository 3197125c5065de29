// Forward body of the MFMA band attention (band_attn_mfma.hip has the description of the mapping): a device function of the workgroup's
// linear index so that other launches can carry band work on some of their workgroups -- the decode step kernels of sel_decode_fused.hip
// run the sliding and the compressed branch of a layer step on workgroups behind those of the selected branch.
#pragma once
#include "attn_mfma_tiles.hpp"
#include "layer_gate.hpp"

namespace nsa {

// STAGE 1: tiles come by LDS-DMA; STAGE 0: register staging (global loads issued behind the fragment reads, ds_write at the top of the
// next iteration): the compute-bound NT = 3 form spends ~19 % of a tile issuing its 8 DMA instructions, plain loads issue faster.
template <typename T, int D, int NT, bool SPLIT, int STAGE>
// WPB: waves of a workgroup that take band work (SPLIT form: every wave is its own unit; the plain kernels run 4)
// mg (SPLIT, S = 1 only): BandMergeArgs of sel_attn_params.hpp -- merge the unit's splits here; eval_gates: this launch's branch also
// evaluates the gate probabilities of its rows
__device__ __forceinline__ void band_attn_body(const BandAttnParams &P, const unsigned bid, const int wpb = 4, const BandMergeArgs *mg = nullptr,
                                               const bool eval_gates = false) {
    using M = MfmaT<T>;
    using G_ = Geo<D>;
    using x8 = typename M::x8;
    using x4 = typename M::x4;
    constexpr int KS = G_::KSTEPS, MT = G_::MT;  // QK k-steps, 16-row dv tiles of PV
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id();
    const int wave = uniform((int)(threadIdx.x >> 6));
    const int tpw = P.tpw, h = P.h;
    const int ngrp = (P.S + tpw - 1) / tpw;  // token groups per (b,g)
    const int nbg = P.B * P.G;
    int bg, grp, sp = 0;
    if (SPLIT) {
        const int64_t wid = (int64_t)bid * wpb + wave;
        const int64_t gi = wid / P.nsplit;
        sp = (int)(wid - gi * P.nsplit);
        bg = (int)(gi / ngrp);
        grp = (int)(gi - (int64_t)bg * ngrp);
        if (bg >= nbg) return;
    } else {
        const int W = (ngrp + 3) >> 2;  // workgroups per (b,g)
        int tc;
        if (P.map_mode == 2) {  // whole (b,g) pairs per XCD (workgroups go round-robin over the 8 XCDs)
            const int xcd = bid & 7, idx = bid >> 3;
            bg = (idx / W) * 8 + xcd;
            tc = idx % W;
        } else {
            bg = bid / W;
            tc = bid % W;
        }
        grp = 4 * tc + wave;
        if (grp >= ngrp || bg >= nbg) return;
    }
    const int b = bg / P.G, g = bg - b * P.G;
    const int tw0 = grp * tpw, ntok = min(tpw, P.S - tw0);

    unsigned char *kl = smem + (size_t)wave * (2 * G_::TILE_BYTES);
    unsigned char *vl = kl + G_::TILE_BYTES;

    // ---- key interval of the wave and of every slot (hi and lo are non-decreasing in t)
    const int hi_min = band_hi(P.t0, P.a, P.dd, P.c, P.S_kv, tw0);
    const int hi_max = band_hi(P.t0, P.a, P.dd, P.c, P.S_kv, tw0 + ntok - 1);
    const int klo = max(0, hi_min - P.w), lo_max = max(0, hi_max - P.w);
    const int rho = lane & 15, q = lane >> 4;
    int hi_s[NT], lo_s[NT];
    int64_t orow[NT];  // (row * h + head) of the slot, -1 = unused slot
    x8 qf[NT][KS];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int slot = 16 * n + rho, tok = slot / h, head = slot - tok * h;
        const bool used = tok < ntok;
        const int t = tw0 + tok;
        hi_s[n] = used ? band_hi(P.t0, P.a, P.dd, P.c, P.S_kv, t) : 0;
        lo_s[n] = max(0, hi_s[n] - P.w);
        orow[n] = used ? ((((int64_t)b * P.S + t) * P.G + g) * h + head) : -1;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            u32x4 raw = {0u, 0u, 0u, 0u};
            if (used) raw = *(const u32x4 *)((const T *)P.Q + orow[n] * D + 32 * s + 8 * q);
            qf[n][s] = __builtin_bit_cast(x8, raw);
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
    // a tile of 32 keys starting at tok0 -> wave-private LDS (swizzle applied on the source side); rows past the end of
    // K/V re-read the last row (they are masked: only boundary tiles reach past hi)
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

    typedef __attribute__((ext_vector_type(4))) unsigned int bu32x4;
    u32x4 kreg[G_::NLD], vreg[G_::NLD];
    uint32_t kwr[G_::NLD], vwr[G_::NLD];
#pragma unroll
    for (int i = 0; i < G_::NLD; ++i) {
        const int r = i * G_::RPI + ld_row;
        kwr[i] = r * G_::ROWB + ((ld_piece ^ G_::swz_k(r)) << 4);
        vwr[i] = r * G_::ROWB + ((((ld_piece >> 1) ^ G_::swz_v(r)) << 5) | ((ld_piece & 1) << 4));
    }
    auto issue_loads = [&](int tok0) {
        const int ks = uniform(tok0 * krowb32), vs = uniform(tok0 * vrowb32);
        const int last = P.S_kv - 1 - tok0;
#pragma unroll
        for (int i = 0; i < G_::NLD; ++i) {
            const int rc = min(i * G_::RPI + ld_row, last);
            kreg[i] = __builtin_bit_cast(u32x4, (bu32x4)__builtin_amdgcn_raw_buffer_load_b128(krs, rc * krowb32 + ld_piece * 16, ks, 0));
            vreg[i] = __builtin_bit_cast(u32x4, (bu32x4)__builtin_amdgcn_raw_buffer_load_b128(vrs, rc * vrowb32 + ld_piece * 16, vs, 0));
        }
    };

    f32x4 o[NT][MT];
    float mrun[NT], lrun[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
#pragma unroll
        for (int m = 0; m < MT; ++m) o[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // unused slots carry +inf so that their (zero) scores never trigger the max-raising path
        mrun[n] = orow[n] >= 0 ? -INFINITY : INFINITY;
        lrun[n] = 0.f;
    }
    const float c2 = P.scale * LOG2E;

    // tiles of this wave (SPLIT: a contiguous share of them)
    const int ntile_all = hi_max > klo ? (hi_max - klo + 31) >> 5 : 0;
    int tile = 0, tile_end = ntile_all;
    if (SPLIT) {
        tile = (int)(((int64_t)ntile_all * sp) / P.nsplit);
        tile_end = (int)(((int64_t)ntile_all * (sp + 1)) / P.nsplit);
    }
    if (tile < tile_end) {
        if (STAGE == 1) issue_dma(klo + 32 * tile);
        else issue_loads(klo + 32 * tile);
    }

    for (; tile < tile_end; ++tile) {
        const int tok0 = klo + 32 * tile;
        x8 kfr[2][KS];
        x8 va[MT];
        if (STAGE == 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // LDS-DMA completion is a vmcnt event
        } else {
#pragma unroll
            for (int i = 0; i < G_::NLD; ++i) {
                *(u32x4 *)(kl + kwr[i]) = kreg[i];
                *(u32x4 *)(vl + vwr[i]) = vreg[i];
            }
            wave_lds_fence();
        }
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
        if (STAGE == 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments are in registers: the buffers may be refilled
            __builtin_amdgcn_sched_barrier(0);
            if (tile + 1 < tile_end) issue_dma(tok0 + 32);
        } else {
            if (tile + 1 < tile_end) issue_loads(tok0 + 32);
            wave_lds_fence();  // the fragment reads above are ordered before next iteration's ds_writes
        }

        const bool interior = tok0 >= lo_max && tok0 + 32 <= hi_min;  // every key valid for every slot
        // S^T of all column tiles first, ONE slow-path decision per key tile: the common path below is straight-line code, so
        // the softmax VALU work of one column tile is scheduled against the MFMAs of its neighbours
        f32x4 sacc[NT][2];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                sacc[n][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) sacc[n][u] = M::mma(kfr[u][s], qf[n][s], sacc[n][u]);
            }
        // key of sacc[n][u][j] = tok0 + 16u + 4q + j
        float x[NT][8];
        float tmax = -INFINITY;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) x[n][4 * u + j] = fmaf(sacc[n][u][j], c2, -mrun[n]);
            tmax = fmaxf(tmax, fmaxf(fmaxf(fmaxf(x[n][0], x[n][1]), fmaxf(x[n][2], x[n][3])),
                                     fmaxf(fmaxf(x[n][4], x[n][5]), fmaxf(x[n][6], x[n][7]))));
        }
        if (!interior || __any(!(tmax <= RESCALE_THR))) {
            asm volatile("; slow path: masks and/or raise the running max" ::: "memory");
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                float vmax = -INFINITY;
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int key = tok0 + 16 * u + 4 * q + j;
                        const float v = (key >= lo_s[n] && key < hi_s[n]) ? sacc[n][u][j] * c2 : -INFINITY;
                        x[n][4 * u + j] = v;
                        vmax = fmaxf(vmax, v);
                    }
                vmax = fmaxf(vmax, __shfl_xor(vmax, 16, 64));
                vmax = fmaxf(vmax, __shfl_xor(vmax, 32, 64));
                const float mnew = fmaxf(mrun[n], vmax);
                const float msub = (mnew == -INFINITY) ? 0.f : mnew;  // nothing valid seen yet: keep x = -inf, p = 0
                const float alpha = (mnew == mrun[n]) ? 1.f : __builtin_amdgcn_exp2f(mrun[n] - msub);
                mrun[n] = mnew;
                lrun[n] *= alpha;
#pragma unroll
                for (int m = 0; m < MT; ++m) o[n][m] *= alpha;
#pragma unroll
                for (int j = 0; j < 8; ++j) x[n][j] -= msub;
            }
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            float psum = 0.f;
            x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float pe = __builtin_amdgcn_exp2f(x[n][j]);
                psum += pe;
                pf[j] = Elt<T>::from_f(pe);
            }
            lrun[n] += psum;
#pragma unroll
            for (int m = 0; m < MT; ++m) o[n][m] = M::mma(va[m], pf, o[n][m]);
        }
    }

    // ---- epilogue
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        float ltot = lrun[n] + __shfl_xor(lrun[n], 16, 64);
        ltot += __shfl_xor(ltot, 32, 64);
        if (orow[n] < 0) continue;
        if (SPLIT && mg && mg->on) {
            // the record of this split goes to the wave's own tile space (its K / V tiles are consumed): [head][PART_PAD + D]
            float *pr = (float *)kl + (int)(orow[n] % h) * (D + PART_PAD);
            if (q == 0) {
                pr[0] = mrun[n];
                pr[1] = ltot;
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) *(f32x4 *)(pr + PART_PAD + 16 * m + 4 * q) = o[n][m];
        } else if (SPLIT) {
            // partial record [row][sp][head][PART_PAD + D] in the layout of the selection kernel's combine pass
            const int64_t row = orow[n] / h;
            const int head = (int)(orow[n] - row * h);
            float *pr = P.part + ((row * P.nsplit + sp) * (int64_t)h + head) * (D + PART_PAD);
            if (q == 0) {
                pr[0] = mrun[n];
                pr[1] = ltot;
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) *(f32x4 *)(pr + PART_PAD + 16 * m + 4 * q) = o[n][m];
        } else {
            const float inv = ltot > 0.f ? 1.f / ltot : 0.f;
            T *Or = (T *)P.O + orow[n] * D;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                x4 ov;
#pragma unroll
                for (int j = 0; j < 4; ++j) ov[j] = Elt<T>::from_f(o[n][m][j] * inv);
                *(x4 *)(Or + 16 * m + 4 * q) = ov;
            }
            if (P.lse && q == 0) P.lse[orow[n]] = ltot > 0.f ? (mrun[n] + __builtin_amdgcn_logf(ltot)) * LN2 : -INFINITY;
        }
    }
    if constexpr (SPLIT && NT == 1 && D == 64) {
        if (mg && mg->on) {
            // ---- merge of the unit's nsplit records (waves wave - sp .. of this workgroup), heads dealt over the unit's waves; one lane per
            // column.  The decode finish kernel's arithmetic (layer_fused.hip: 16 clamped steps, weights of the missing splits 0).
            // (every live wave of the workgroup arrives here exactly once: waves beyond the last unit have returned above -- a terminated
            // wave no longer counts for s_barrier -- and a wave without tiles or without used slots still runs through the epilogue)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const int ns = P.nsplit;
            const float *u0 = (const float *)(smem + (size_t)(wave - sp) * (2 * G_::TILE_BYTES));
            constexpr int SST = 2 * G_::TILE_BYTES / 4;  // floats between the records of consecutive splits
            const int64_t row = ((int64_t)b * P.S + tw0) * P.G + g;
            for (int hh = sp; hh < h; hh += ns) {
                const float *base = u0 + hh * (D + PART_PAD);
                const float mv = lane < ns ? base[lane * SST] : -INFINITY;
                const float lv = lane < ns ? base[lane * SST + 1] : 0.f;
                float pv[16];
#pragma unroll
                for (int s = 0; s < 16; ++s) pv[s] = base[min(s, ns - 1) * SST + PART_PAD + lane];
                const float mmax = wave_max(mv);
                const float w = (mv == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mv - mmax);
                const float ltot = wave_sum(lv * w);
                float acc = 0.f;
#pragma unroll
                for (int s = 0; s < 16; ++s) acc = fmaf(pv[s], __shfl(w, s, 64), acc);
                ((T *)P.O)[(row * h + hh) * D + lane] = Elt<T>::from_f(acc * (ltot > 0.f ? 1.f / ltot : 0.f));
            }
            if (eval_gates && sp == ns - 1) {  // (the unit's last wave: the one with the fewest heads to merge)
                float *sqp = (float *)(kl + 3 * G_::TILE_BYTES / 2);  // 256 floats behind this wave's record (h <= 16: 4352 bytes)
                const T *Qr = (const T *)P.Q + row * h * D;
                float pr[3];
                if (mg->Hd <= 32 && h <= 8) {
                    GateFast<T> gf;
                    gf.load(Qr, h, mg->Hd, mg->gw1, mg->gb1, mg->gw2, mg->gb2);
                    gf.compute(h, mg->Hd, mg->tau, sqp, pr);
                } else {
                    gate_probs<T>(Qr, h, D, mg->Hd, mg->gw1, mg->gb1, mg->gw2, mg->gb2, mg->tau, sqp, pr);
                }
                if (lane < 3) mg->gates[row * 3 + lane] = lane == 0 ? pr[0] : (lane == 1 ? pr[1] : pr[2]);
            }
        }
    }
}

}  // namespace nsa
