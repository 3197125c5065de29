// Lane-movement and MFMA-layout probe for gfx950 (run on the GPU box): which lane each cross-lane primitive reads, and where the
// 32x32x16 bf16 MFMA expects / delivers its operands.  hipcc --offload-arch=gfx950 -O2 -o probe_lanes probe_lanes.hip && ./probe_lanes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void k(int *out, float *mo) {
    const int lane = threadIdx.x;
    int v = lane;
    out[0 * 64 + lane] = __builtin_amdgcn_ds_swizzle(v, 0xC000 | (0 << 10) | (1 << 5));   // rotate mode, dir 0, by 1
    out[1 * 64 + lane] = __builtin_amdgcn_ds_swizzle(v, 0xC000 | (1 << 10) | (1 << 5));   // rotate mode, dir 1, by 1
    out[2 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x142, 0xa, 0x1, false);      // row_bcast:15, rows 1 and 3, bank 0
    out[3 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x111, 0xf, 0xe, false);      // row_shr:1, banks 1-3
    const auto s = __builtin_amdgcn_permlane32_swap(v, v + 100, false, false);             // a = lane, b = lane + 100
    out[4 * 64 + lane] = s[0];
    out[5 * 64 + lane] = s[1];
    out[6 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x103, 0xf, 0xf, true);       // row_shl:3, bound_ctrl
    // masked identity DPP add: rows 2,3 only
    float d = 1000.f + lane, a = (float)lane, b = 0.5f;
    asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 quad_perm:[0,1,2,3] row_mask:0xc bank_mask:0xf" : "+v"(d) : "v"(a), "v"(b));
    out[7 * 64 + lane] = (int)(d * 2.f);
    // MFMA 32x32x16: A[m][k] = (m == 3 && k == 9) ? 1 : 0 style probes are awkward; use A[m][k] = m (k == 0 only), B[k][n] = n (k == 0 only) + 1
    bf16x8 fa, fb;
    const int half = lane >> 5, r = lane & 31;
    for (int i = 0; i < 8; ++i) {
        const int kk = 8 * half + i;
        fa[i] = (__bf16)((kk == 0) ? (float)r : 0.f);        // A[m = r][k = 0] = r
        fb[i] = (__bf16)((kk == 0) ? (float)(r + 1) : 0.f);  // B[k = 0][n = r] = r + 1
    }
    f32x16 c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, c, 0, 0, 0);
    for (int i = 0; i < 16; ++i) mo[i * 64 + lane] = c[i];  // expect m * (n + 1)
}

int main() {
    int *out;
    float *mo;
    (void)hipMalloc(&out, 8 * 64 * 4);
    (void)hipMalloc(&mo, 16 * 64 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, mo);
    int h[8 * 64];
    float m[16 * 64];
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipMemcpy(m, mo, sizeof(m), hipMemcpyDeviceToHost);
    const char *names[8] = {"swizzle rot dir0 by1", "swizzle rot dir1 by1", "dpp row_bcast15 rm=a bm=1", "dpp row_shr1 bm=e", "permlane32_swap[0]", "permlane32_swap[1]",
                            "dpp row_shl3 bc", "fmac_dpp identity rm=c (x2)"};
    for (int t = 0; t < 8; ++t) {
        printf("%-28s:", names[t]);
        for (int l = 0; l < 64; ++l) printf(" %d", h[t * 64 + l]);
        printf("\n");
    }
    // MFMA layout: value m*(n+1) -> recover (m, n) per (i, lane)
    int bad = 0;
    for (int i = 0; i < 16; ++i)
        for (int l = 0; l < 64; ++l) {
            const int n = l & 31, mrow = 8 * (i >> 2) + 4 * (l >> 5) + (i & 3);
            if (m[i * 64 + l] != (float)(mrow * (n + 1))) ++bad;
        }
    printf("mfma 32x32x16: acc[i] of lane l = C[8 (i>>2) + 4 (l>>5) + (i&3)][l&31] with A row = l&31, B column = l&31, k = 8 (l>>5) + e: %s (%d mismatches)\n",
           bad ? "NO" : "yes", bad);
    return 0;
}
