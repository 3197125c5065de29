// How fast can one CU gather 16 KB blocks (8 KB of K + 8 KB of V, as the 64k selection attention does) out of a 512 MB region, as a function of
// the bytes it keeps in flight?  Each wave loops over random blocks; per round it has, in flight at once,
//   mode 0: one block by LDS-DMA (16 KB)                         -- the attention kernel today (8 waves per CU: 128 KB per CU)
//   mode 1: one block by LDS-DMA + one block by global loads into 64 VGPRs (32 KB)
//   mode 2: two blocks by global loads into 128 VGPRs (32 KB), no LDS
// and consumes them with a few ALU operations.  Workgroups of 256 threads, W per CU (LDS request padded to fix W).
//   hipcc --offload-arch=gfx950 -O2 -o gather_depth gather_depth.hip && ./gather_depth
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE>
__global__ __launch_bounds__(256, 2) void gather(const unsigned char *__restrict__ base, const unsigned *__restrict__ blocks, int rounds, unsigned nblocks,
                                                 unsigned *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char *my = smem + wave * 16384;
    const unsigned gw = blockIdx.x * 4 + wave;
    u32x4 acc = {0, 0, 0, 0};
    typedef __attribute__((address_space(3))) void lds_void;
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, (short)0, (int)0x7fffffff, 0x00020000);
    for (int r = 0; r < rounds; ++r) {
        const unsigned b0 = __builtin_amdgcn_readfirstlane(blocks[(gw * 131u + r * 2u) % nblocks]);
        const unsigned b1 = __builtin_amdgcn_readfirstlane(blocks[(gw * 131u + r * 2u + 1u) % nblocks]);
        u32x4 reg[16];
        if (MODE == 0 || MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)(my + i * 1024), 16, lane * 16, (int)(b0 * 16384u + i * 1024), 0, 0);
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) reg[i] = *(const u32x4 *)(base + (size_t)b1 * 16384u + i * 1024 + lane * 16);
        }
        u32x4 reg2[16];
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) reg2[i] = *(const u32x4 *)(base + (size_t)b0 * 16384u + i * 1024 + lane * 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE == 0 || MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc += *(const u32x4 *)(my + i * 1024 + lane * 16);
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc += reg[i];
        }
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc += reg2[i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int MODE>
static void run(const char *name, int W, const unsigned char *base, const unsigned *blocks, unsigned nblocks, unsigned *out, int ncu) {
    const int rounds = 400;
    const size_t lds = W == 1 ? 90 * 1024 : 65536;  // 1 or 2 workgroups per CU
    (void)hipFuncSetAttribute((const void *)gather<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(gather<MODE>, dim3(ncu * W), dim3(256), lds, 0, base, blocks, rounds, nblocks, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double bytes = (double)ncu * W * 4 * rounds * (MODE == 0 ? 16384.0 : 32768.0);
    printf("%-44s %d workgroup(s) per CU: %7.3f ms, %6.2f TB/s = %6.1f GB/s per CU (in flight per CU: %d KB)\n", name, W, ms, bytes / ms * 1e-9, bytes / ms * 1e-6 / ncu,
           W * 4 * (MODE == 0 ? 16 : 32));
}

int main(int argc, char **argv) {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    const size_t region_mb = argc > 1 ? (size_t)atoi(argv[1]) : 512;
    const size_t region = region_mb << 20;
    const unsigned nblk = (unsigned)(region / 16384);
    printf("region %zu MB\n", region_mb);
    unsigned char *base;
    (void)hipMalloc(&base, region);
    (void)hipMemset(base, 1, region);
    std::vector<unsigned> hb(1 << 20);
    unsigned x = 12345;
    for (auto &v : hb) { x = x * 1664525u + 1013904223u; v = (x >> 8) % nblk; }
    unsigned *blocks, *out;
    (void)hipMalloc(&blocks, hb.size() * 4);
    (void)hipMemcpy(blocks, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&out, (size_t)ncu * 2 * 256 * 4);
    for (int W = 1; W <= 2; ++W) {
        run<0>("LDS-DMA, one block per wave", W, base, blocks, (unsigned)hb.size(), out, ncu);
        run<1>("LDS-DMA + one block into registers", W, base, blocks, (unsigned)hb.size(), out, ncu);
        run<2>("two blocks into registers", W, base, blocks, (unsigned)hb.size(), out, ncu);
    }
    return 0;
}
