// Does hipExtAnyOrderLaunch clear the AQL barrier bit on gfx950?  Two kernels that each spin for ~T us on one workgroup, in ONE stream:
// serial they take 2T, overlapped T.  A third, ordinary launch behind them must still see both results (it checks the flags they set).
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/any_order tools/ubench/any_order.hip && tools/ubench/any_order
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                 \
    do {                                                                      \
        hipError_t e_ = (x);                                                  \
        if (e_ != hipSuccess) {                                               \
            printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            exit(1);                                                          \
        }                                                                     \
    } while (0)

__global__ void spin_kernel(int *flag, long long ticks, int value) {
    const long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) *flag = value;
}
__global__ void check_kernel(const int *a, const int *b, int *out, int va, int vb) {
    if (threadIdx.x == 0) *out = (*a == va && *b == vb) ? 1 : 0;
}

int main() {
    int *flags;
    CK(hipMalloc(&flags, 3 * sizeof(int)));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const long long ticks = 2000;  // 20 us
    for (int mode = 0; mode < 2; ++mode) {
        float best = 1e9f;
        int okall = 1;
        for (int it = 0; it < 20; ++it) {
            CK(hipMemsetAsync(flags, 0, 3 * sizeof(int), st));
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, flags, ticks, it + 1);
            if (mode == 0) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, flags + 1, ticks, it + 101);
            else hipExtLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, flags + 1, ticks, it + 101);
            hipLaunchKernelGGL(check_kernel, dim3(1), dim3(64), 0, st, flags, flags + 1, flags + 2, it + 1, it + 101);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            int ok;
            CK(hipMemcpy(&ok, flags + 2, sizeof(int), hipMemcpyDeviceToHost));
            okall &= ok;
            if (ms < best) best = ms;
        }
        printf("%s second launch: two 20 us kernels + check = %.1f us (best of 20), check kernel saw both results: %s\n",
               mode == 0 ? "ordinary " : "any-order", best * 1e3f, okall ? "yes" : "NO");
    }
    return 0;
}
