// Decode form of the selection attention: ONE workgroup of NW waves per query row (b,g), S = 1.
//
// A decode step reads every selected K/V row exactly once (n*l' rows per (b,g)): it is bound by HBM latency and bandwidth, not by
// any pipe.  The waves of the row's workgroup take the 64-key chunks of the row's selected union (chunk e -> wave e mod NW) and
// issue ALL loads of a chunk at once -- K straight into registers in MFMA A-fragment shape (8 global_load_dwordx4: the GEMV /
// decode form of the CDNA guide), V by 8 LDS-DMA pieces into the wave's 8 KiB of LDS for the transposing reads -- so a chunk costs
// one memory round trip; the per-wave (m, l, O^T) partials are merged through LDS behind one barrier, in wave order (bitwise
// reproducible), and the row's O is written once.  No workspace, no second launch.
//   NW = 16 (1024 threads, one workgroup per CU): a row's 16 chunks are all in flight at once -- few rows (R <= CUs);
//   NW = 8  (512 threads, 128 VGPRs, 64 KiB of V tiles: two workgroups per CU): a wave walks its two chunks with the K rows of the
//           second one prefetched behind the first; one row's top-n / merge phases overlap the other row's gather -- many rows.
// The chunks come from a provider: the block list the fused decode step's selector leaves in LDS (sel_decode_fused.hip) or the
// normalised segments of arbitrary ranges (standalone launch: sel_attn_decode.hip).  For selector output both enumerate the same
// chunks in the same order, so the two routes give the same bits.
// Semantics: union of the clamped ranges (normalise_ranges_lanes), end <= start ignored, empty row -> zeros
// (attention_kernels.py:705-772).  bf16 / f16, Dk = Dv = 64, h <= 16, V rows contiguous (128 B apart).
#pragma once
#include "attn_mfma_tiles.hpp"

namespace nsa {
#ifndef DEC_TS
#ifdef NSA_DEC_TS
static __device__ long long g_dec_ts[64];
#define DEC_TS(i) do { if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) g_dec_ts[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DEC_TS(i)
#endif
#endif

struct DecAttnArgs {
    const void *Q;  // [R,h,64]
    const void *K;  // [B,G,S_kv,64] strided
    const void *V;
    void *O;        // [R,h,64]
    int G, h, S_kv, n;
    int64_t ksb, ksg, kss, vsb, vsg, vss;
    float c2;  // scale * log2(e)
};

constexpr int DEC_ATT_TILE = 64 * 128;                         // V chunk per wave
constexpr int DEC_ATT_TAIL = ((SEG_INTS * 4 + 15) / 16) * 16;  // sorted segments of the row (standalone launch)
constexpr int dec_att_lds(int nw) { return nw * DEC_ATT_TILE + DEC_ATT_TAIL; }

// waves per row workgroup for a launch of `rows` rows: 16 while every row can have a CU of its own, 8 beyond (two rows per CU)
int dec_att_waves(int64_t rows);

// ---- chunk providers: chunk e (ascending token order) -> first key and number of keys (1..64)
// blocks picked by the selector, in LDS (sorted, disjoint, 64 keys each; the last one is cut at t_end = min(t + 1, S_kv))
struct ListChunks {
    const int *list;
    int t_end;
    __device__ __forceinline__ void get(int e, int &tok0, int &len) const {
        tok0 = uniform(list[e]) << 6;
        len = min(64, t_end - tok0);
    }
};
// sorted disjoint segments in registers: lane i holds segment i (start, length), cb = chunks before it
struct SegChunks {
    int nseg, sstart, slen, cb, nck;
    __device__ __forceinline__ void get(int c, int &tok0, int &len) const {
        const int lane = lane_id();
        const unsigned long long hit = __ballot(lane < nseg && c >= cb && c < cb + nck);
        const int sl = __builtin_ctzll(hit);
        const int s0 = __builtin_amdgcn_readlane(sstart, sl), sn = __builtin_amdgcn_readlane(slen, sl), c0 = __builtin_amdgcn_readlane(cb, sl);
        tok0 = s0 + 64 * (c - c0);
        len = min(64, s0 + sn - tok0);  // keys of this chunk, >= 1
    }
};

// ---- which chunks a wave gathers, and in which order.  NW = 8: e = wave, wave + 8, ...  NW = 16 ("edge rule", from 3 chunks on): the first
// chunk goes to wave 0 and the last two to waves 14 and 15, the chunks between them (p = e - 1) go round the waves in the order
// 1 .. 13, 0, 14, 15; a wave takes its edge chunk first, then its middle chunks in ascending order.  For selector output the edge chunks
// are the forced blocks (0, t//l' - 1, t//l'), known from t alone: the fused decode step lets those three waves fetch them at kernel start,
// behind the whole scoring chain (sel_decode_fused.hip).  A pure function of (e, NC): the standalone launch follows the same rule, so
// the partial records of a row are merged in the same grouping on both routes.
template <int NW>
__device__ __forceinline__ int dec_first_chunk(int w, int NC) {
    if (NW != 16 || NC < 3) return w < NC ? w : NC;
    if (w == 0) return 0;
    if (w == 14) return NC - 2;
    if (w == 15) return NC - 1;
    return w <= NC - 3 ? w : NC;  // middle waves 1 .. 13: p = w - 1
}
template <int NW>
__device__ __forceinline__ int dec_next_chunk(int cur, int w, int NC) {
    if (NW != 16 || NC < 3) return cur + NW;
    const bool edge = (w == 0 && cur == 0) || (w == 14 && cur == NC - 2) || (w == 15 && cur == NC - 1);
    const int iw = (w >= 1 && w <= 13) ? w - 1 : (w == 0 ? 13 : w);
    const int p = edge ? iw : cur - 1 + 16;
    return p < NC - 3 ? p + 1 : NC;
}

// a chunk fetched ahead of time (fused decode step): K rows by LDS-DMA into an 8 KiB tile of their own (XOR-swizzled 16-B pieces, the
// image of sel_attn_blocks_mfma.hip), V rows into the wave's V tile as always.  Needs contiguous K rows (kss = 64).
struct DecPrefetch {
    int tok0;                    // first key of the chunk the wave fetched, -1 = none
    const unsigned char *ktile;  // its K image in LDS
};
template <typename T>
__device__ __forceinline__ void decode_prefetch_chunk(const DecAttnArgs &A, int64_t row, int tok0, int len, unsigned char *vtile, unsigned char *ktile) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) void lds_void;
    const int lane = lane_id();
    const int g = (int)(row % A.G);
    const int64_t b = row / A.G;
    const uint64_t ka = (uint64_t)((const T *)A.K + b * A.ksb + (int64_t)g * A.ksg), va = (uint64_t)((const T *)A.V + b * A.vsb + (int64_t)g * A.vsg);
    const uint32_t ka_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ka), ka_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ka >> 32));
    const uint32_t va_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)va), va_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(va >> 32));
    const int bytes = __builtin_amdgcn_readfirstlane((int)((int64_t)A.S_kv * 128));
    const auto krs = __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)ka_hi << 32) | (uint64_t)ka_lo), (short)0, bytes, 0x00020000);
    const auto vrs = __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)va_hi << 32) | (uint64_t)va_lo), (short)0, bytes, 0x00020000);
    const int ld_row = lane >> 3, ld_piece = lane & 7;
    const uint32_t ksw = (uint32_t)((ld_piece ^ (ld_row & 7)) << 4);
    const uint32_t vsw = (uint32_t)(((((ld_piece >> 1) ^ ((ld_row >> 1) & 3)) << 1) | (ld_piece & 1)) << 4);
    const int so = uniform(tok0 * 128);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int rc = min(8 * i + ld_row, len - 1);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (lds_void *)(ktile + i * 1024), 16, rc * 128 + ksw, so, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int rc = min(8 * i + ld_row, len - 1);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (lds_void *)(vtile + i * 1024), 16, rc * 128 + vsw, so, 0, 0);
    }
#else
    (void)A, (void)row, (void)tok0, (void)len, (void)vtile, (void)ktile;
#endif
}

// vt: NW * DEC_ATT_TILE bytes of LDS, 16-byte aligned (V tiles, then the partial records); every thread of the NW * 64-thread workgroup must
// call (one workgroup barrier inside).  qf_in: the row's Q^T fragments if the caller holds them already (lane (rho, q):
// Q[head min(rho, h-1)][32 s + 8 q ..], s = 0, 1).  NC chunks; chunks with len <= 0 must not occur.
template <typename T, int NW, typename CH>
__device__ __forceinline__ void decode_attend_chunks(const DecAttnArgs &A, int64_t row, const CH &ch, const int NC, unsigned char *vt,
                                                     const typename MfmaT<T>::x8 *qf_in = nullptr, const DecPrefetch pre = DecPrefetch{-1, nullptr}) {
    using M = MfmaT<T>;
    using x8 = typename M::x8;
    using x4 = typename M::x4;
    constexpr int ROWB = 128;
    constexpr bool PF = NW < 16;  // K rows of the wave's next chunk fetched behind the current one (the register budget of 512 threads allows it)
    const int lane = lane_id();
    const int wave = uniform((int)(threadIdx.x >> 6));
    const int rho = lane & 15, q = lane >> 4;
    const int h = A.h;
    const int g = (int)(row % A.G);
    const int64_t b = row / A.G;
    unsigned char *vl = vt + wave * DEC_ATT_TILE;

    // Q^T fragments (B operand): column = head (columns >= h repeat the last head: their results are never stored)
    x8 qf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
        qf[s] = qf_in ? qf_in[s] : __builtin_bit_cast(x8, *(const u32x4 *)((const T *)A.Q + (row * h + min(rho, h - 1)) * 64 + 32 * s + 8 * q));

    const unsigned char *Kb = (const unsigned char *)((const T *)A.K + b * A.ksb + (int64_t)g * A.ksg);
    const unsigned char *Vb = (const unsigned char *)((const T *)A.V + b * A.vsb + (int64_t)g * A.vsg);
    const int64_t krowb = A.kss * 2;
    const int ld_row = lane >> 3, ld_piece = lane & 7;
    [[maybe_unused]] const uint32_t vsw = (uint32_t)(((((ld_piece >> 1) ^ ((ld_row >> 1) & 3)) << 1) | (ld_piece & 1)) << 4);
    uint32_t vrd0[4];
    {
        const int qq = rho >> 2, pp = rho & 3, r = 4 * q + qq;
#pragma unroll
        for (int m = 0; m < 4; ++m) vrd0[m] = r * ROWB + ((m ^ ((r >> 1) & 3)) << 5) + 8 * pp;
    }
    // (readfirstlane returns a SIGNED int: the halves go through uint32_t, or a low half >= 2^31 sign-extends into the high one --
    // tests/test_hip_descriptor_bit31.py runs every descriptor-building kernel on such addresses)
    const uint64_t va = (uint64_t)Vb;
    const uint32_t va_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)va), va_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(va >> 32));
    [[maybe_unused]] const auto vrs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(((uint64_t)va_hi << 32) | (uint64_t)va_lo), (short)0, __builtin_amdgcn_readfirstlane((int)((int64_t)(A.S_kv - 1) * ROWB + ROWB)), 0x00020000);

    // every load of a chunk goes out at once: V by LDS-DMA, K straight to registers (rows past the chunk re-read its last row)
    auto issue_v = [&](int tok0, int len) {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef __attribute__((address_space(3))) void lds_void;
        const int vs = uniform(tok0 * ROWB);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int rc = min(8 * i + ld_row, len - 1);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (lds_void *)(vl + i * 1024), 16, rc * ROWB + vsw, vs, 0, 0);
        }
#else
        (void)tok0, (void)len;
#endif
    };
    auto load_k = [&](int tok0, int len, x8 (&k)[4][2]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kr = tok0 + min(16 * u + rho, len - 1);
#pragma unroll
            for (int s = 0; s < 2; ++s) k[u][s] = __builtin_bit_cast(x8, *(const u32x4 *)(Kb + (int64_t)kr * krowb + 64 * s + 16 * q));
        }
    };

    f32x4 o[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) o[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mrun = -INFINITY, lrun = 0.f;

    int cur = dec_first_chunk<NW>(wave, NC), tok0 = 0, len = 0;
    x8 kfr[4][2];
    if (cur < NC) {
        ch.get(cur, tok0, len);
        if (pre.tok0 == tok0 && pre.tok0 >= 0) {
            // this chunk went out at kernel start (K and V by LDS-DMA): both have long landed; the K fragments come from the LDS image
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int s = 0; s < 2; ++s) kfr[u][s] = *(const x8 *)(pre.ktile + (16 * u + rho) * ROWB + (((4 * s + q) ^ (rho & 7)) << 4));
        } else {
            load_k(tok0, len, kfr);  // K first: loads complete in order, and the scores need K before the P V product needs V
            issue_v(tok0, len);
        }
    }
    while (cur < NC) {
        const int nxt = dec_next_chunk<NW>(cur, wave, NC);
        const bool hn = nxt < NC;
        int ntok0 = 0, nlen = 0;
        [[maybe_unused]] x8 kn[4][2];
        if (hn) {
            ch.get(nxt, ntok0, nlen);
            if constexpr (PF) load_k(ntok0, nlen, kn);
        }
        f32x4 sacc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            sacc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) sacc[u] = M::mma(kfr[u][s], qf[s], sacc[u]);
        }
        float x[16];
        float vmax = -INFINITY;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = (16 * u + 4 * q + j < len) ? sacc[u][j] * A.c2 : -INFINITY;
                x[4 * u + j] = v;
                vmax = fmaxf(vmax, v);
            }
        vmax = xor32_max(xor16_max(vmax));
        const float mnew = fmaxf(mrun, vmax);  // finite: the chunk has at least one key
        const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);  // 0 for the first chunk
        mrun = mnew;
        float psum = 0.f;
        x8 pf[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float pe = __builtin_amdgcn_exp2f(x[i] - mnew);
            psum += pe;
            pf[i >> 3][i & 7] = Elt<T>::from_f(pe);
        }
        lrun = lrun * alpha + psum;
        // the V pieces have landed (LDS-DMA completion is a vmcnt event; loads complete in order: the 8 prefetched K loads may stay out)
        if (PF && hn) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            o[m] *= alpha;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const x4 lo = M::tr(vl + vrd0[m] + hf * 32 * ROWB), hi = M::tr(vl + vrd0[m] + (hf * 32 + 16) * ROWB);
                x8 vf;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    vf[j] = lo[j];
                    vf[4 + j] = hi[j];
                }
                o[m] = M::mma(vf, pf[hf], o[m]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // transposing reads done before the next chunk's DMA overwrites the tile
        __builtin_amdgcn_sched_barrier(0);
        if (hn) {
            if constexpr (PF) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int s = 0; s < 2; ++s) kfr[u][s] = kn[u][s];
            } else {
                load_k(ntok0, nlen, kfr);
            }
            issue_v(ntok0, nlen);
        }
        cur = nxt;
        tok0 = ntok0;
        len = nlen;
    }
    // ---- partial record of this wave over its own V tile: m[16] | l[16] | o[16 slots][64]
    const float ltot = xor32_add(xor16_add(lrun));
    float *pm = (float *)vl, *pl = pm + 16, *po = pm + 32;
    if (q == 0) {
        pm[rho] = mrun;
        pl[rho] = ltot;
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) *(f32x4 *)(po + rho * 64 + 16 * m + 4 * q) = o[m];
    __syncthreads();
    // ---- merge: thread (head, d) walks the NW partial records in wave order (fixed order: bitwise reproducible)
    for (int tid = threadIdx.x; tid < h * 64; tid += NW * 64) {
        const int hh = tid >> 6, d = tid & 63;
        float mm = -INFINITY;
#pragma unroll
        for (int w = 0; w < NW; ++w) mm = fmaxf(mm, ((const float *)(vt + w * DEC_ATT_TILE))[hh]);
        float acc = 0.f, l = 0.f;
        if (mm > -INFINITY) {
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const float *pw = (const float *)(vt + w * DEC_ATT_TILE);
                const float wgt = __builtin_amdgcn_exp2f(pw[hh] - mm);  // exp2(-inf) = 0 for a wave without a chunk
                l = fmaf(pw[16 + hh], wgt, l);
                acc = fmaf(pw[32 + hh * 64 + d], wgt, acc);
            }
        }
        ((T *)A.O)[(row * h + hh) * 64 + d] = Elt<T>::from_f(l > 0.f ? acc / l : 0.f);
    }
}

// arbitrary ranges (lanes i < n of wave 0 pass range i, unclamped): sorted union -> chunks.  lds: dec_att_lds(NW) bytes.
template <typename T, int NW>
__device__ __forceinline__ void decode_attend_row(const DecAttnArgs &A, int64_t row, int rs, int re, unsigned char *lds) {
    const int lane = lane_id();
    const int wave = uniform((int)(threadIdx.x >> 6));
    int *seg = (int *)(lds + NW * DEC_ATT_TILE);
    if (wave == 0) {
        int ns;
        const int total = normalise_ranges_lanes(rs, re, A.n, A.S_kv, seg, &ns);
        if (lane == 0) {
            seg[SEG_INTS - 2] = ns;
            seg[SEG_INTS - 1] = total;
        }
    }
    __syncthreads();
    SegChunks ch;
    ch.nseg = uniform(seg[SEG_INTS - 2]);  // wave uniform by construction: tell the compiler (scalar loop control)
    ch.sstart = 0;
    ch.slen = 0;
    if (lane < ch.nseg) {
        ch.sstart = seg[2 * lane];
        ch.slen = seg[2 * lane + 3] - seg[2 * lane + 1];
    }
    ch.nck = (ch.slen + 63) >> 6;
    int inc = ch.nck;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
    }
    ch.cb = inc - ch.nck;
    const int NC = uniform(__shfl(inc, 63, 64));
    decode_attend_chunks<T, NW>(A, row, ch, NC, lds);
}

}  // namespace nsa
