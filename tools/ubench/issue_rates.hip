// Issue-rate microbenchmark for gfx950 (run on the GPU box):  hipcc --offload-arch=gfx950 -O2 -mllvm -amdgpu-mfma-vgpr-form=1 -o issue_rates issue_rates.hip && ./issue_rates
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
    f32x16 w0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, w1 = w0, w2 = w0;
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {1.0001f, 1.0001f}, pc = {0.5f, 0.5f};
    bf16x8 fa, fb, fb1, fb2;  // distinct B operands per accumulator chain (identical MFMAs would be merged by the compiler)
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(float)(threadIdx.x & 7); fb[i] = (__bf16)1.0f; fb1[i] = (__bf16)(float)(i & 1); fb2[i] = (__bf16)(float)(threadIdx.x & 1); }
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
        } else if (PAT == 17) {  // 2 MFMA 32x32x16 + 32 pk_fma
            w0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, w0, 0, 0, 0);
            asm volatile(REP2(REP2("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n")) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
            w1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, w1, 0, 0, 0);
            asm volatile(REP2(REP2("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n")) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
        } else if (PAT == 18) {  // 32 pk_fma alone
            asm volatile(REP2(REP2("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n")) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
            asm volatile(REP2(REP2("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n")) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
        } else if (PAT == 19) {  // 2 MFMA 32x32x16 + 32 asm volatile(REP2(REP2("v_add_f32_dpp %0, %1, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %2, %1 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %3, %2 row_shl:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %0, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)); adds
            w0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, w0, 0, 0, 0);
            asm volatile(REP2(REP2("v_add_f32_dpp %0, %1, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %2, %1 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %3, %2 row_shl:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %0, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            w1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, w1, 0, 0, 0);
            asm volatile(REP2(REP2("v_add_f32_dpp %0, %1, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %2, %1 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %3, %2 row_shl:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %0, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else if (PAT == 20) {  // 32 asm volatile(REP2(REP2("v_add_f32_dpp %0, %1, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %2, %1 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %3, %2 row_shl:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %0, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)); adds alone
            asm volatile(REP2(REP2("v_add_f32_dpp %0, %1, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %2, %1 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %3, %2 row_shl:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %0, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            asm volatile(REP2(REP2("v_add_f32_dpp %0, %1, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %2, %1 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %3, %2 row_shl:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %0, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else if (PAT == 21) {  // 2 MFMA 32x32x16 + 32 exp
            w0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, w0, 0, 0, 0);
            asm volatile(REP2(REP2("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            w1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, w1, 0, 0, 0);
            asm volatile(REP2(REP2("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else if (PAT == 22) {  // 32 exp alone
            asm volatile(REP2(REP2("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            asm volatile(REP2(REP2("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else if (PAT == 23) {  // burst + dependent VALU: 12 MFMA 32x32x16 (3 fresh accumulators x 4 k-steps), then 192 fma that read the 48 results
            f32x16 z0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, z1 = z0, z2 = z0;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                z0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, z0, 0, 0, 0);
                z1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb1, z1, 0, 0, 0);
                z2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb2, z2, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(z0[i]), "v"(b0));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(z1[i]), "v"(b0));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a2) : "v"(z2[i]), "v"(b0));
                }
        } else if (PAT == 24) {  // same, software-pipelined: the fmas read the results of the PREVIOUS iteration's burst
            f32x16 z0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, z1 = z0, z2 = z0;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                z0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, z0, 0, 0, 0);
                z1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb1, z1, 0, 0, 0);
                z2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb2, z2, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(w0[i]), "v"(b0));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(w1[i]), "v"(b0));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a2) : "v"(w2[i]), "v"(b0));
                }
            w0 = z0; w1 = z1; w2 = z2;
        } else if (PAT == 25) {  // burst in CHAIN order: 3 x (4 dependent MFMA 32x32x16 back to back), then 192 fma on the results
            f32x16 z0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, z1 = z0, z2 = z0;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) z0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, z0, 0, 0, 0);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) z1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb1, z1, 0, 0, 0);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) z2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb2, z2, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(z0[i]), "v"(b0));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(z1[i]), "v"(b0));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a2) : "v"(z2[i]), "v"(b0));
                }
        } else if (PAT == 26) {  // ONE dependent chain of 16 v_fma_f32 (latency, not throughput)
            asm volatile(REP8("v_fma_f32 %0, %0, %1, %2\n") REP8("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a0) : "v"(b0), "v"(c0));
        } else if (PAT == 27) {  // one dependent chain fma -> exp -> add, 5 times (15 instr)
            asm volatile(REP2(REP2("v_fma_f32 %1, %0, %2, %3\n v_exp_f32 %1, %1\n v_add_f32 %0, %0, %1\n")) "v_fma_f32 %1, %0, %2, %3\n v_exp_f32 %1, %1\n v_add_f32 %0, %0, %1\n" : "+v"(a0), "+v"(a1) : "v"(b0), "v"(c0));
        } else if (PAT == 28) {  // two independent chains of 8 v_fma_f32, interleaved
            asm volatile(REP8("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n") : "+v"(a0), "+v"(a1) : "v"(b0), "v"(c0));
        } else if (PAT == 29) {  // ONE chain of 4 dependent MFMA 32x32x16
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n" : "+v"(w0) : "v"(fa), "v"(fb));
        } else if (PAT == 33) {  // 4 independent MFMA 16x16x16 bf16 (the legacy half-K form the backward's dV / dK products use)
            typedef __attribute__((ext_vector_type(4))) short s16x4;
            const s16x4 ha = __builtin_bit_cast(s16x4, (f32x4){a0, a1, 0, 0}.xy), hb = __builtin_bit_cast(s16x4, (f32x4){a2, a3, 0, 0}.xy);
            m0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ha, hb, m0, 0, 0, 0);
            m1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(hb, ha, m1, 0, 0, 0);
            m2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ha, ha, m2, 0, 0, 0);
            m3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(hb, hb, m3, 0, 0, 0);
        } else if (PAT == 30) {  // ONE chain of 4 dependent MFMA 16x16x32
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n" : "+v"(m0) : "v"(fa), "v"(fb));
        } else if (PAT == 31) {  // three chains of 4 dependent MFMA 32x32x16, chain after chain (12 MFMA)
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n" : "+v"(w0) : "v"(fa), "v"(fb));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n" : "+v"(w1) : "v"(fa), "v"(fb1));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n" : "+v"(w2) : "v"(fa), "v"(fb2));
        } else if (PAT == 32) {  // the same 12 MFMA, the three chains interleaved
            asm volatile(REP2(REP2("v_mfma_f32_32x32x16_bf16 %0, %3, %4, %0\n v_mfma_f32_32x32x16_bf16 %1, %3, %5, %1\n v_mfma_f32_32x32x16_bf16 %2, %3, %6, %2\n")) : "+v"(w0), "+v"(w1), "+v"(w2) : "v"(fa), "v"(fb), "v"(fb1), "v"(fb2));
        } else if (PAT == 15) {  // burst form: 12 MFMA 32x32x16 back to back (3 accumulators x 4 k-steps), then 192 fma
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                w0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, w0, 0, 0, 0);
                w1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, w1, 0, 0, 0);
                w2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, w2, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 12; ++q) { asm volatile(REP2(REP2("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(c0)); }
        } else if (PAT == 16) {  // the same 12 MFMA + 192 fma, one MFMA per 16 fma
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                w0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, w0, 0, 0, 0);
                { asm volatile(REP2(REP2("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(c0)); }
                w1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, w1, 0, 0, 0);
                { asm volatile(REP2(REP2("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(c0)); }
                w2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, w2, 0, 0, 0);
                { asm volatile(REP2(REP2("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(c0)); }
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + m0[0] + m1[1] + m2[2] + m3[3] + p0[0] + p1[1] + p2[0] + p3[1] + w0[0] + w1[5] + w2[7];
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
    run<33>("4 x mfma 16x16x16 bf16 (legacy half-K form)", 4, out, ghz, n);
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
    run<15>("burst: 12 x mfma32, then 192 x fma", 204, out, ghz, n);
    run<16>("12 x (mfma32 + 16 x fma)", 204, out, ghz, n);
    run<23>("burst 12 x mfma32, then 192 x fma on results", 204, out, ghz, n);
    run<24>("same, fmas on the previous burst", 204, out, ghz, n);
    run<25>("chain-order burst (3 x 4 dependent), then fma", 204, out, ghz, n);
    run<26>("one chain of 16 dependent fma", 16, out, ghz, n);
    run<27>("one chain 5 x (fma -> exp -> add)", 15, out, ghz, n);
    run<28>("two chains of 8 dependent fma", 16, out, ghz, n);
    run<29>("one chain of 4 dependent mfma32", 4, out, ghz, n);
    run<30>("one chain of 4 dependent mfma16", 4, out, ghz, n);
    run<31>("3 chains x 4 dependent mfma32, chain after chain", 12, out, ghz, n);
    run<32>("3 chains x 4 mfma32, interleaved", 12, out, ghz, n);
    run<17>("2 x mfma32 + 32 x pk_fma", 34, out, ghz, n);
    run<18>("32 x pk_fma", 32, out, ghz, n);
    run<19>("2 x mfma32 + 32 x add_dpp", 34, out, ghz, n);
    run<20>("32 x add_dpp", 32, out, ghz, n);
    run<21>("2 x mfma32 + 32 x exp", 34, out, ghz, n);
    run<22>("32 x exp", 32, out, ghz, n);
    return 0;
}
