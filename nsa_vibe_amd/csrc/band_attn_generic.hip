// Generic (VALU, any dtype, Dk/Dv <= 256) band attention forward: the route for fp32 and for head sizes the MFMA kernel
// does not cover.  One wave per (b, t, g, head); lanes = 64 keys of a chunk for the scores, = output columns for P.V.
#include "nsa_common.hpp"
#include "sel_attn_params.hpp"

namespace nsa {

template <typename T>
__global__ __launch_bounds__(256) void band_attn_fwd_generic_kernel(BandAttnParams P) {
    __shared__ float sq[4][256];
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
    const int64_t nrh = (int64_t)P.B * P.S * P.G * P.h;
    if (wid >= nrh) return;
    const int64_t row = wid / P.h;
    const int g = (int)(row % P.G);
    const int64_t bt = row / P.G;
    const int t = (int)(bt % P.S), b = (int)(bt / P.S);
    const int hi = band_hi(P.t0, P.a, P.dd, P.c, P.S_kv, t), lo = max(0, hi - P.w);
    const T *Qr = (const T *)P.Q + wid * P.Dk;
    const T *Kb = (const T *)P.K + (int64_t)b * P.ksb + (int64_t)g * P.ksg;
    const T *Vb = (const T *)P.V + (int64_t)b * P.vsb + (int64_t)g * P.vsg;
    for (int d = lane; d < P.Dk; d += 64) sq[wave][d] = Elt<T>::to_f(Qr[d]);
    wave_lds_fence();
    float m = -INFINITY, l = 0.f, acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = lo; k0 < hi; k0 += 64) {
        const int key = k0 + lane;
        const bool valid = key < hi;
        float s = -INFINITY;
        if (valid) {
            const T *kr = Kb + (int64_t)key * P.kss;
            float dot = 0.f;
            for (int d = 0; d < P.Dk; ++d) dot = fmaf(sq[wave][d], Elt<T>::to_f(kr[d]), dot);
            s = dot * P.scale;
        }
        const float mnew = fmaxf(m, wave_max(s));  // finite: lane 0 of every chunk is valid
        const float alpha = expf(m - mnew);
        const float p = valid ? expf(s - mnew) : 0.f;
        l = l * alpha + wave_sum(p);
        m = mnew;
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] *= alpha;
        const int nk = min(64, hi - k0);
        for (int kk = 0; kk < nk; ++kk) {
            const float pk = __shfl(p, kk, 64);
            const T *vr = Vb + (int64_t)(k0 + kk) * P.vss;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int d = lane + 64 * c;
                if (d < P.Dv) acc[c] = fmaf(pk, Elt<T>::to_f(vr[d]), acc[c]);
            }
        }
    }
    const float inv = l > 0.f ? 1.f / l : 0.f;
    T *Or = (T *)P.O + wid * P.Dv;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int d = lane + 64 * c;
        if (d < P.Dv) Or[d] = Elt<T>::from_f(acc[c] * inv);
    }
    if (P.lse && lane == 0) P.lse[wid] = l > 0.f ? m + logf(l) : -INFINITY;
}

int launch_band_attn_fwd_generic(const BandAttnParams &P, int dtype, hipStream_t st) {
    NSA_CHECK_ARG(P.Dk <= 256 && P.Dv <= 256, "band_attn: Dk/Dv up to 256 supported");
    const int64_t nrh = (int64_t)P.B * P.S * P.G * P.h;
    NSA_CHECK_ARG((nrh + 3) / 4 < ((int64_t)1 << 31), "band_attn: too many rows for one launch");
    const dim3 grid((unsigned)((nrh + 3) / 4)), block(256);
    if (dtype == NSA_DT_F32) hipLaunchKernelGGL(band_attn_fwd_generic_kernel<float>, grid, block, 0, st, P);
    else if (dtype == NSA_DT_BF16) hipLaunchKernelGGL(band_attn_fwd_generic_kernel<__bf16>, grid, block, 0, st, P);
    else hipLaunchKernelGGL(band_attn_fwd_generic_kernel<_Float16>, grid, block, 0, st, P);
    NSA_LAUNCH_CHECK("band_attn_fwd_generic");
    return NSA_OK;
}

}  // namespace nsa
