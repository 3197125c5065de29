// Issue-rate microbenchmark for gfx950 (run on the GPU box):  hipcc --offload-arch=gfx950 -O2 -o issue_rates issue_rates.hip && ./issue_rates
// Each kernel runs ITERS x UNROLL copies of one instruction pattern per wave; W waves per SIMD (one 256-thread workgroup = one wave per SIMD,
// W workgroups per CU).  Reported: SIMD cycles per pattern instance = time x clock x (1 / (ITERS x UNROLL x W)), i.e. the reciprocal issue rate the
// SIMD sustains with W waves to pick from.  Patterns: fma, pk_fma, exp, mfma 16x16x32 / 32x32x16 bf16, and MFMA + VALU mixes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int ITERS = 2000;

#define REP8(X) X X X X X X X X
#define REP2(X) X X

template <int PAT>
__global__ __launch_bounds__(256) void k(float *out) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = 1.0001f, c0 = 0.5f;
    f32x4 m0 = {0, 0, 0, 0}, m1 = m0, m2 = m0, m3 = m0;
    f32x16 w0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, w1 = w0;
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {1.0001f, 1.0001f}, pc = {0.5f, 0.5f};
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(float)(threadIdx.x & 7); fb[i] = (__bf16)1.0f; }
    for (int it = 0; it < ITERS; ++it) {
        if (PAT == 0) {  // 8 independent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(c0));
        } else if (PAT == 1) {  // 8 independent v_exp_f32
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n"
                         "v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (PAT == 2) {  // 4 independent v_pk_fma_f32 (8 fmas)
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
        } else if (PAT == 3) {  // 4 independent MFMA 16x16x32 bf16
            m0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, m0, 0, 0, 0);
            m1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, m1, 0, 0, 0);
            m2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, m2, 0, 0, 0);
            m3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, m3, 0, 0, 0);
        } else if (PAT == 4) {  // 4 MFMA + 8 fma (independent of the MFMA results)
            m0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, m0, 0, 0, 0);
            asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n" : "+v"(a0), "+v"(a1) : "v"(b0), "v"(c0));
            m1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, m1, 0, 0, 0);
            asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n" : "+v"(a2), "+v"(a3) : "v"(b0), "v"(c0));
            m2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, m2, 0, 0, 0);
            asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n" : "+v"(a4), "+v"(a5) : "v"(b0), "v"(c0));
            m3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, m3, 0, 0, 0);
            asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n" : "+v"(a6), "+v"(a7) : "v"(b0), "v"(c0));
        } else if (PAT == 5) {  // 4 MFMA + 16 fma
            m0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, m0, 0, 0, 0);
            asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n" : "+v"(a0), "+v"(a1) : "v"(b0), "v"(c0));
            m1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, m1, 0, 0, 0);
            asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n" : "+v"(a2), "+v"(a3) : "v"(b0), "v"(c0));
            m2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, m2, 0, 0, 0);
            asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n" : "+v"(a4), "+v"(a5) : "v"(b0), "v"(c0));
            m3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, m3, 0, 0, 0);
            asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n" : "+v"(a6), "+v"(a7) : "v"(b0), "v"(c0));
        } else if (PAT == 6) {  // 4 MFMA + 8 exp
            m0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, m0, 0, 0, 0);
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n" : "+v"(a0), "+v"(a1));
            m1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, m1, 0, 0, 0);
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n" : "+v"(a2), "+v"(a3));
            m2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, m2, 0, 0, 0);
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n" : "+v"(a4), "+v"(a5));
            m3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, m3, 0, 0, 0);
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n" : "+v"(a6), "+v"(a7));
        } else if (PAT == 7) {  // 8 fma + 8 exp interleaved (independent)
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_exp_f32 %4, %4\n v_fma_f32 %1, %1, %8, %9\n v_exp_f32 %5, %5\n v_fma_f32 %2, %2, %8, %9\n v_exp_f32 %6, %6\n"
                         "v_fma_f32 %3, %3, %8, %9\n v_exp_f32 %7, %7\n v_fma_f32 %0, %0, %8, %9\n v_exp_f32 %4, %4\n v_fma_f32 %1, %1, %8, %9\n v_exp_f32 %5, %5\n"
                         "v_fma_f32 %2, %2, %8, %9\n v_exp_f32 %6, %6\n v_fma_f32 %3, %3, %8, %9\n v_exp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(c0));
        } else if (PAT == 8) {  // the scorer's sweep-1 shape per MFMA pair: 2 dependent MFMA, then fma -> exp -> add on the 4 results
            m0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, (f32x4){0, 0, 0, 0}, 0, 0, 0);
            m0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa, m0, 0, 0, 0);
            m1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, (f32x4){0, 0, 0, 0}, 0, 0, 0);
            m1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa, m1, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) a0 += __builtin_amdgcn_exp2f(fmaf(m0[j], b0, c0));
#pragma unroll
            for (int j = 0; j < 4; ++j) a1 += __builtin_amdgcn_exp2f(fmaf(m1[j], b0, c0));
        } else if (PAT == 9) {  // 2 independent MFMA 32x32x16 bf16 (the flops of 4 x 16x16x32)
            w0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, w0, 0, 0, 0);
            w1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, w1, 0, 0, 0);
        } else if (PAT == 10) {  // 2 MFMA 32x32x16 + 8 fma
            w0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, w0, 0, 0, 0);
            asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(c0));
            w1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, w1, 0, 0, 0);
            asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n" : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(c0));
        } else if (PAT == 11) {  // 2 MFMA 32x32x16 + 16 fma
            w0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, w0, 0, 0, 0);
            asm volatile(REP2("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(c0));
            w1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, w1, 0, 0, 0);
            asm volatile(REP2("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n") : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(c0));
        } else if (PAT == 12) {  // 2 MFMA 32x32x16 + 8 exp
            w0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, w0, 0, 0, 0);
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            w1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, w1, 0, 0, 0);
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n" : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (PAT == 13) {  // 2 MFMA 32x32x16 + 32 fma (VALU-heavy mix: does the matrix time hide?)
            w0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, w0, 0, 0, 0);
            asm volatile(REP2(REP2("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(c0));
            w1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, w1, 0, 0, 0);
            asm volatile(REP2(REP2("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")) : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(c0));
        } else if (PAT == 14) {  // 4 MFMA 16x16x32 + 32 fma
            m0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, m0, 0, 0, 0);
            asm volatile(REP2("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(c0));
            m1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, m1, 0, 0, 0);
            asm volatile(REP2("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n") : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(c0));
            m2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, m2, 0, 0, 0);
            asm volatile(REP2("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(c0));
            m3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, m3, 0, 0, 0);
            asm volatile(REP2("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n") : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(c0));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + m0[0] + m1[1] + m2[2] + m3[3] + p0[0] + p1[1] + p2[0] + p3[1] + w0[0] + w1[5];
}

template <int PAT>
static void run(const char *name, int per_iter, float *out, double ghz, int ncu) {
    for (int W = 1; W <= 4; ++W) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<PAT>, dim3(ncu * W), dim3(256), 0, 0, out);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<PAT>, dim3(ncu * W), dim3(256), 0, 0, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double cyc = ms * 1e-3 * ghz * 1e9 / ((double)ITERS * W);
        printf("%-44s W=%d: %8.1f cycles per iteration of one wave-slot (%d instr) = %6.2f cycles/instr\n", name, W, cyc, per_iter, cyc / per_iter);
    }
}

int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate * 1e-6;
    printf("%s, %d CUs, %.2f GHz (reported)\n", p.name, p.multiProcessorCount, ghz);
    float *out;
    (void)hipMalloc(&out, sizeof(float) * 256 * p.multiProcessorCount * 4);
    const int n = p.multiProcessorCount;
    run<0>("8 x v_fma_f32", 8, out, ghz, n);
    run<1>("8 x v_exp_f32", 8, out, ghz, n);
    run<2>("4 x v_pk_fma_f32", 4, out, ghz, n);
    run<3>("4 x mfma 16x16x32 bf16", 4, out, ghz, n);
    run<4>("4 x mfma + 8 x fma", 12, out, ghz, n);
    run<5>("4 x mfma + 16 x fma", 20, out, ghz, n);
    run<6>("4 x mfma + 8 x exp", 12, out, ghz, n);
    run<7>("8 x fma + 8 x exp", 16, out, ghz, n);
    run<8>("2 x (2 dep. mfma) + 8 x (fma, exp, add)", 28, out, ghz, n);
    run<9>("2 x mfma 32x32x16 bf16", 2, out, ghz, n);
    run<10>("2 x mfma32 + 8 x fma", 10, out, ghz, n);
    run<11>("2 x mfma32 + 16 x fma", 18, out, ghz, n);
    run<12>("2 x mfma32 + 8 x exp", 10, out, ghz, n);
    run<13>("2 x mfma32 + 32 x fma", 34, out, ghz, n);
    run<14>("4 x mfma16 + 32 x fma", 36, out, ghz, n);
    return 0;
}
