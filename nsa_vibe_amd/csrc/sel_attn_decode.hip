// Decode form of the selection attention as its own launch (one 1024-thread workgroup per query row): see sel_attn_decode.hpp.
// Used by the decode routes that do not run the fused scorer+selector+attention kernel (contexts beyond its LDS budget, other
// score geometries, the plain nsa_sel_attn_fwd call with S = 1).
#include "sel_attn_decode.hpp"

namespace nsa {

template <typename T, int NW>
__global__ __launch_bounds__(NW * 64, 4) void sel_attn_decode_wg_kernel(DecAttnArgs A, const int32_t *__restrict__ ranges) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dlds[];
    const int64_t row = blockIdx.x;
    const int lane = lane_id();
    int rs = 0, re = 0;
    if (threadIdx.x < 64 && lane < A.n) {
        rs = ranges[(row * A.n + lane) * 2];
        re = ranges[(row * A.n + lane) * 2 + 1];
    }
    decode_attend_row<T, NW>(A, row, rs, re, dlds);
}

// 16 waves per row while every row can have a CU to itself (all 16 chunks of a selector row in flight at once); beyond that 8 waves
// and 64 KiB of V tiles, so that two rows share a CU and one row's merge / set-up overlaps the other's gather.  The fused decode step
// (sel_decode_fused.hip) follows the same rule: a row's chunks go to the same waves on both routes, and the outputs agree bit for bit.
int dec_att_waves(int64_t rows) {
    const int mode = tuning(TUNE_DECODE_WAVES);
    if (mode == 8 || mode == 16) return mode;
    return rows <= 256 ? 16 : 8;
}

bool sel_attn_decode_wg_supported(int dtype, int h, int Dk, int Dv, int n, int64_t ksb, int64_t ksg, int64_t kss, int64_t vsb, int64_t vsg,
                                  int64_t vss, const void *Q, const void *K, const void *V) {
    if (tuning(TUNE_DECODE_WG) == 0) return false;
    // 16-byte global loads and LDS-DMA pieces at K + b ksb + g ksg (+ row kss): every stride a multiple of 8 elements
    return (dtype == NSA_DT_BF16 || dtype == NSA_DT_F16) && Dk == 64 && Dv == 64 && h >= 1 && h <= 16 && n >= 1 && n <= 64 && vss == 64 &&
           kss % 8 == 0 && ksb % 8 == 0 && ksg % 8 == 0 && vsb % 8 == 0 && vsg % 8 == 0 && ((uintptr_t)Q % 16 == 0) && ((uintptr_t)K % 16 == 0) && ((uintptr_t)V % 16 == 0);
}

int launch_sel_attn_decode_wg(const void *Q, const void *K, const void *V, const int32_t *ranges, void *O, int64_t R, int G, int h, int S_kv, int n,
                              int64_t ksb, int64_t ksg, int64_t kss, int64_t vsb, int64_t vsg, int64_t vss, int dtype, float scale, hipStream_t st) {
    NSA_CHECK_ARG(R >= 1 && R < ((int64_t)1 << 31), "sel_attn_decode: bad row count");
    NSA_CHECK_ARG((int64_t)S_kv * 128 < ((int64_t)1 << 31), "sel_attn_decode: one (b,g) V slab must be smaller than 2 GiB (buffer addressing)");
    DecAttnArgs A{Q, K, V, O, G, h, S_kv, n, ksb, ksg, kss, vsb, vsg, vss, scale * LOG2E};
    const int nw = dec_att_waves(R);
    void (*k)(DecAttnArgs, const int32_t *) =
        nw == 16 ? (dtype == NSA_DT_BF16 ? sel_attn_decode_wg_kernel<__bf16, 16> : sel_attn_decode_wg_kernel<_Float16, 16>)
                 : (dtype == NSA_DT_BF16 ? sel_attn_decode_wg_kernel<__bf16, 8> : sel_attn_decode_wg_kernel<_Float16, 8>);
    static void *raised[4] = {nullptr, nullptr, nullptr, nullptr};  // raise the dynamic-LDS limit once per kernel (the runtime call costs about a millisecond)
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
    hipLaunchKernelGGL(k, dim3((unsigned)R), dim3(nw * 64), dec_att_lds(nw), st, A, ranges);
    NSA_LAUNCH_CHECK("sel_attn_decode_wg");
    return NSA_OK;
}

}  // namespace nsa
