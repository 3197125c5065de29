// Where do the 4 waves of a 256-thread workgroup land?  (gfx950, HW_ID register: wave slot [3:0], SIMD [5:4], CU [11:8], SE [15:13]; XCC_ID separately)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(unsigned *out) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    __shared__ float buf[64];
    // stay resident a little so that a second workgroup shares the CU
    float x = threadIdx.x;
    for (int i = 0; i < 20000; ++i) x = x * 1.0001f + 0.5f;
    if (x == 12345.f) buf[0] = x;
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
}
int main() {
    const int nwg = 1024;
    unsigned *out;
    (void)hipMalloc(&out, nwg * 4 * 2 * 4);
    hipLaunchKernelGGL(k, dim3(nwg), dim3(256), 0, 0, out);
    static unsigned h[nwg * 8];
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    int distinct4 = 0;
    for (int w = 0; w < nwg; ++w) {
        unsigned mask = 0;
        for (int i = 0; i < 4; ++i) mask |= 1u << ((h[(w * 4 + i) * 2] >> 4) & 3);
        if (mask == 0xf) ++distinct4;
    }
    printf("workgroups whose 4 waves sit on 4 different SIMDs: %d of %d\n", distinct4, nwg);
    for (int w = 0; w < 6; ++w) {
        printf("wg %d:", w);
        for (int i = 0; i < 4; ++i) {
            const unsigned hw = h[(w * 4 + i) * 2];
            printf("  [xcc %u se %u cu %u simd %u slot %u]", h[(w * 4 + i) * 2 + 1] & 0xf, (hw >> 13) & 7, (hw >> 8) & 0xf, (hw >> 4) & 3, hw & 0xf);
        }
        printf("\n");
    }
    return 0;
}
