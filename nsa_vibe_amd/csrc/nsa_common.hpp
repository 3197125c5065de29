// Shared device/host helpers for libnsa_sel_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nsa_sel_hip.h"

namespace nsa {

// ---- error plumbing (host) -------------------------------------------------------------
void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what);

#define NSA_CHECK_ARG(cond, ...)          \
    do {                                  \
        if (!(cond)) {                    \
            ::nsa::set_error(__VA_ARGS__); \
            return NSA_ERR_INVALID;       \
        }                                 \
    } while (0)

#define NSA_HIP_TRY(expr)                                                   \
    do {                                                                    \
        hipError_t _e = (expr);                                             \
        if (_e != hipSuccess) return ::nsa::hip_fail(_e, #expr);            \
    } while (0)

#define NSA_LAUNCH_CHECK(name)                                              \
    do {                                                                    \
        hipError_t _e = hipGetLastError();                                  \
        if (_e != hipSuccess) return ::nsa::hip_fail(_e, name " launch");   \
    } while (0)

// ---- measurement / A-B switches (host) -----------------------------------------------------
// Process-wide integer switches, seeded ONCE from the environment (NSA_HIP_<NAME>) when the library is first asked and
// changed afterwards only through nsa_hip_set_tuning (include/nsa_sel_hip.h): no getenv on the launch path.
enum Tune {
    TUNE_SEL_ROWS = 0,      // NSA_HIP_SEL_ROWS: selection forward/backward form, -1 auto, 0 one row per wave, 1 pairs, 3 48-slot tiles
    TUNE_ATTN_MAP,          // NSA_HIP_ATTN_MAP: one-row kernel workgroup mapping, -1 auto
    TUNE_ATTN_STAGE,        // NSA_HIP_ATTN_STAGE: 0 register staging, 1 LDS-DMA (default)
    TUNE_BAND_STAGE,        // NSA_HIP_BAND_STAGE: same for the band kernel
    TUNE_DECODE_UNFUSED,    // NSA_HIP_DECODE_UNFUSED: 1 = three-kernel decode scorer route, 0 = fused whenever it fits, -1 auto
    TUNE_SEL_BLOCKS,        // NSA_HIP_SEL_BLOCKS: 64-key block form of the selection forward, -1 auto, 0 off, N = row pairs per wave
    TUNE_DECODE_WG,         // NSA_HIP_DECODE_WG: decode attention as one workgroup per row (+ fused into the decode scorer), -1 auto, 0 off
    TUNE_SEL_ROWSUM,        // NSA_HIP_SEL_ROWSUM: block-form forward, row sums of P by MFMA (1) or by v_add (0)
    TUNE_DECODE_STENCIL,    // NSA_HIP_DECODE_STENCIL: fused decode kernel, 1 = closed-form Eq.9 taps for l = 2d, l' = 4d; 0 = always the CSC
    TUNE_SEL_FUSE,          // NSA_HIP_SEL_FUSE: nsa_sel_select_attn_fwd, 1 = the selector runs inside the attention launch, 0 = two launches
    TUNE_SCORES_FORM,       // NSA_HIP_SCORES_FORM: fused scorer with h = 6: -1 / 2 = 32x32x16 tiles, 16 queries per wave (D = 64), 1 = 16x16x32 tiles with 8 queries on 3 column tiles, 0 = on 4 tiles
    TUNE_SEL_FLAT,          // NSA_HIP_SEL_FLAT: block-form attention with h = 6, 1 = 8 rows on 3 column tiles (no idle columns), 0 = on 4 tiles, -1 = by context length
    TUNE_SEL_KSPLIT,        // NSA_HIP_SEL_KSPLIT: block-form attention with the keys of a pair split over two XCD groups: -1 by shape, 0 never, 1 always
    TUNE_DECODE_STOP,       // NSA_HIP_DECODE_STOP: measurement aid, the fused decode kernel returns after phase N (1 logits, 2 scores, 3 top-n); 0 = run all
    TUNE_DECODE_WAVES,      // NSA_HIP_DECODE_WAVES: waves per row workgroup of the decode kernels, -1 by the number of rows, 8 or 16
    TUNE_DECODE_SPLIT,      // NSA_HIP_DECODE_SPLIT: fused decode step, workgroups that share the logits phase of one row, -1 by shape, N forces N
    TUNE_DECODE_STEP,       // NSA_HIP_DECODE_STEP: 1 = the one-launch decode step of sel_decode_fused.hip wherever it applies (default), 0 = the round-2 kernels
    TUNE_DECODE_TEAM_SPIN,  // NSA_HIP_DECODE_TEAM_SPIN: split decode step, polls a workgroup waits for its team before it goes on alone; -1 = 512, 0 = never waits
    TUNE_DECODE_WIDE,       // NSA_HIP_DECODE_WIDE: one-launch decode step, a long row in ONE workgroup: 1 = four chunks per wave with the later chunks' logits in LDS (exact), 2 = the one-pass form (eight chunks per wave, scores within 2 ulp), wherever the row fits; -1 = only where a team of workgroups would not fit the chip; 0 = never
    TUNE_SEL_KSPLIT_T1,     // NSA_HIP_SEL_KSPLIT_T1 / _T2: key-split attention, rows from this position on (position = row + S_kv - S) are split 2-way / 4-way; -1 = 32768 / never
    TUNE_SEL_KSPLIT_T2,
    TUNE_SCORES_SELECT,     // NSA_HIP_SCORES_SELECT: nsa_sel_scores_select, 1 = the top-n selection of a query tile always runs inside the scorer launch (32x32x16 form), 0 = always its own launch, -1 = inside from 48k contexts on (S_cmp >= 3072: where it is faster)
    TUNE_DECODE_BAND,       // NSA_HIP_DECODE_BAND: layer decode step, the sliding + compressed branches: 0 = their own launch; 1 = on the launch of the selected branch's one-launch decode step, splits merged by the finish kernel; 2 = the same with the splits merged by the workgroup that holds them (branch outputs and gates final); 3 / -1 (default) = 2 + with <= 8 rows the gate mix is the A operand of the output projection (three launches per step)
    TUNE_COUNT
};
int tuning(Tune t);

// ---- vector types ----------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

constexpr int WAVE = 64;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

// ---- element traits --------------------------------------------------------------------
template <typename T>
struct Elt;
template <>
struct Elt<float> {
    static constexpr int dt = NSA_DT_F32;
    __device__ static float to_f(float x) { return x; }
    __device__ static float from_f(float x) { return x; }
};
template <>
struct Elt<__bf16> {
    static constexpr int dt = NSA_DT_BF16;
    __device__ static float to_f(__bf16 x) { return (float)x; }
    __device__ static __bf16 from_f(float x) { return (__bf16)x; }
};
template <>
struct Elt<_Float16> {
    static constexpr int dt = NSA_DT_F16;
    __device__ static float to_f(_Float16 x) { return (float)x; }
    __device__ static _Float16 from_f(float x) { return (_Float16)x; }
};

// ---- wave helpers ----------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ int uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
// a kernel argument (or any wave-uniform value) materialised in scalar registers HERE: its scalar load is issued at this point of the
// program instead of right before its first use
template <typename V>
__device__ __forceinline__ void pin_sgpr(V &x) {
    static_assert(sizeof(V) == 4 || sizeof(V) == 8, "pin_sgpr: 32- or 64-bit values");
    asm volatile("" : "+s"(x));
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// ---- cross-lane steps without the LDS crossbar (gfx950) --------------------------------------------------------------------------------
// __shfl_xor(v, 16 / 32) goes through ds_bpermute (an LDS-crossbar round trip, ~100 cycles, on a dependent chain); v_permlane16_swap /
// v_permlane32_swap exchange whole rows / halves of two registers in one VALU instruction.  With both operands = v, every lane ends up with
// (value of the even row or lower half, value of the odd row or upper half) of its pair: one more op gives the xor-16 / xor-32 butterfly step.
// IEEE add and max are commutative, so the result has the bits of `v op __shfl_xor(v, 16 / 32)`.
__device__ __forceinline__ float xor16_max(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor16_add(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_add(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// lane i reads lane (i + N) mod 16 of its 16-lane row (DPP row_ror).  For a value that is already symmetric under xor of every higher bit
// of the lane index (the state after the earlier steps of a butterfly reduction) this IS the xor-N partner.
template <int N>
__device__ __forceinline__ float row_ror(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 | N, 0xf, 0xf, false));
}

// maximum of an unsigned value over the wave (uniform result): four DPP rotations inside the 16-lane rows, then the row and half swaps
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#define NSA_ROR_MAX(N) v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x120 | N, 0xf, 0xf, false));
    NSA_ROR_MAX(8) NSA_ROR_MAX(4) NSA_ROR_MAX(2) NSA_ROR_MAX(1)
#undef NSA_ROR_MAX
    const auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    v = max(a[0], a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (unsigned)__builtin_amdgcn_readfirstlane((int)max(b[0], b[1]));
}

// make LDS traffic of this wave visible to its own later LDS reads (wave-private regions only:
// DS operations of one wave execute in order, the fence only pins the compiler).
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------------------
// Range normalisation (one wave per query row).
//   Reference semantics (attention_kernels.py:721-732): clamp start/end to [0,S_kv], allowed =
//   union of the ranges; entries with end <= start contribute nothing (every other executor and
//   sel_cuda.cpp:16-18 skip them).  The union is produced as sorted, disjoint segments so the
//   kernels can walk tokens in ascending order without touching a token twice.
//   seg[] lives in wave-private LDS: seg[2*i] = start, seg[2*i+1] = exclusive prefix length,
//   seg[2*nseg+1] holds the total.  Supports n <= 64 ranges.
// Returns the total number of selected tokens L (wave uniform); *nseg_out = number of segments
// (including empty ones; empty segments have equal consecutive prefix offsets).
// ---------------------------------------------------------------------------------------
// core: lane i < n holds range i as (s, e) (unclamped)
__device__ __forceinline__ int normalise_ranges_lanes(int s, int e, int n, int S_kv, int *seg, int *nseg_out) {
    const int lane = lane_id();
    if (lane < n) {
        s = min(max(s, 0), S_kv);
        e = min(max(e, 0), S_kv);
    } else {
        s = 0;
        e = 0;
    }
    const bool valid = e > s;
    // rank sort by (start, lane) among valid entries
    int rank = 0;
    for (int j = 0; j < n; ++j) {
        int sj = __shfl(s, j, 64);
        int ej = __shfl(e, j, 64);
        bool vj = ej > sj;
        rank += (vj && (sj < s || (sj == s && j < lane))) ? 1 : 0;
    }
    const int nvalid = __popcll(__ballot(valid));
    // scatter to sorted order through LDS (reuse seg as scratch: [0..63] starts, [64..127] ends)
    if (valid) {
        seg[rank] = s;
        seg[64 + rank] = e;
    }
    wave_lds_fence();
    int ss = 0, se = 0;
    if (lane < nvalid) {
        ss = seg[lane];
        se = seg[64 + lane];
    }
    wave_lds_fence();
    // exclusive running max of the ends -> effective start
    int run = (lane < nvalid) ? se : 0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int v = __shfl_up(run, o, 64);
        if (lane >= o) run = max(run, v);
    }
    int prev = __shfl_up(run, 1, 64);
    if (lane == 0) prev = 0;
    int es = max(ss, prev);
    int len = (lane < nvalid) ? max(se - es, 0) : 0;
    int inc = len;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
    }
    const int total = __shfl(inc, 63, 64);
    if (lane < nvalid) {
        seg[2 * lane] = es;
        seg[2 * lane + 1] = inc - len;
    }
    if (lane == 0) {
        seg[2 * nvalid] = 0;
        seg[2 * nvalid + 1] = total;
    }
    wave_lds_fence();
    *nseg_out = nvalid;
    return total;
}
__device__ __forceinline__ int normalise_ranges(const int32_t *__restrict__ rg, int n, int S_kv, int *seg, int *nseg_out) {
    const int lane = lane_id();
    int s = 0, e = 0;
    if (lane < n) {
        s = rg[2 * lane];
        e = rg[2 * lane + 1];
    }
    return normalise_ranges_lanes(s, e, n, S_kv, seg, nseg_out);
}
constexpr int SEG_INTS = 132;  // LDS ints needed by normalise_ranges per wave (>= 2*64+2, and 128 scratch)

}  // namespace nsa
