// v_exp_f32 against v_exp_legacy_f32 on gfx950: issue rate (cycles per wave instruction, 4 waves per SIMD) and accuracy against exp2 in double.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
template <int LEG>
__global__ __launch_bounds__(256) void rate(float *out) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 0.1f, a2 = a0 + 0.2f, a3 = a0 + 0.3f, a4 = a0 + 0.4f, a5 = a0 + 0.5f, a6 = a0 + 0.6f, a7 = a0 + 0.7f;
    for (int it = 0; it < 2000; ++it) {
        if (LEG)
            asm volatile("v_exp_legacy_f32 %0, %0\n v_exp_legacy_f32 %1, %1\n v_exp_legacy_f32 %2, %2\n v_exp_legacy_f32 %3, %3\n v_exp_legacy_f32 %4, %4\n v_exp_legacy_f32 %5, %5\n v_exp_legacy_f32 %6, %6\n v_exp_legacy_f32 %7, %7\n"
                         "v_exp_legacy_f32 %0, %0\n v_exp_legacy_f32 %1, %1\n v_exp_legacy_f32 %2, %2\n v_exp_legacy_f32 %3, %3\n v_exp_legacy_f32 %4, %4\n v_exp_legacy_f32 %5, %5\n v_exp_legacy_f32 %6, %6\n v_exp_legacy_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        else
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ void acc(const float *x, float *y0, float *y1, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float a = x[i], b, c;
    asm volatile("v_exp_f32 %0, %1" : "=v"(b) : "v"(a));
    asm volatile("v_exp_legacy_f32 %0, %1" : "=v"(c) : "v"(a));
    y0[i] = b;
    y1[i] = c;
}
int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate * 1e-6;
    float *out;
    (void)hipMalloc(&out, 4 * 256 * p.multiProcessorCount * 4);
    for (int leg = 0; leg < 2; ++leg) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        for (int r = 0; r < 2; ++r) {
            (void)hipEventRecord(e0);
            if (leg) hipLaunchKernelGGL(rate<1>, dim3(p.multiProcessorCount * 4), dim3(256), 0, 0, out);
            else hipLaunchKernelGGL(rate<0>, dim3(p.multiProcessorCount * 4), dim3(256), 0, 0, out);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
        }
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.2f cycles per wave instruction (4 waves per SIMD, nominal %.2f GHz)\n", leg ? "v_exp_legacy_f32" : "v_exp_f32", ms * 1e-3 * ghz * 1e9 / (2000.0 * 16 * 4), ghz);
    }
    const int n = 1 << 20;
    std::vector<float> hx(n), h0(n), h1(n);
    for (int i = 0; i < n; ++i) hx[i] = -40.0f + 52.0f * (float)i / n;  // the range of a softmax exponent
    float *dx, *d0, *d1;
    (void)hipMalloc(&dx, n * 4); (void)hipMalloc(&d0, n * 4); (void)hipMalloc(&d1, n * 4);
    (void)hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(acc, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, n);
    (void)hipMemcpy(h0.data(), d0, n * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(h1.data(), d1, n * 4, hipMemcpyDeviceToHost);
    double e0m = 0, e1m = 0, s0 = 0, s1 = 0;
    for (int i = 0; i < n; ++i) {
        const double ref = std::exp2((double)hx[i]);
        const double r0 = (h0[i] - ref) / ref, r1 = (h1[i] - ref) / ref;
        e0m = std::fmax(e0m, std::fabs(r0)); e1m = std::fmax(e1m, std::fabs(r1));
        s0 += r0; s1 += r1;
    }
    printf("relative error against exp2 (double), x in [-40, 12]: v_exp_f32 max %.3g mean %.3g; v_exp_legacy_f32 max %.3g mean %.3g\n", e0m, s0 / n, e1m, s1 / n);
    return 0;
}
