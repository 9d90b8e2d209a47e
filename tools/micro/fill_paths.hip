// How fast can ONE CU bring L2-resident bytes on chip, by path?  512-thread workgroups, one per CU (128 KB of LDS each, like the 256-row GEMM):
//   dma   : global_load_lds_dwordx4 only (1 KiB per wave-instruction), counted vmcnt, 8 in flight per wave
//   reg   : global_load_dwordx4 into VGPRs + ds_write_b128, 8 in flight per wave
//   half  : every wave alternates the two (half of the bytes by each path)
// Source: a 2 MB window per XCD-group of workgroups, re-read many times (L2 hits), 128-byte rows of 8 lanes x 16 B like the GEMM's form-K pieces.
// Build: hipcc --offload-arch=gfx950 -O3 -w tools/micro/fill_paths.hip -o tools/micro/fill_paths.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, long window, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const char* base = src + (long)(blockIdx.x & 7) * window;            // workgroups of one XCD share a window
    char* my = lds + w * 16384;                                          // 16 KiB ring per wave
    u4 r[8];
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const long off = (((long)it * 8 + j) * 8 + w) * 1024 % window;
            const char* p = base + off + lane * 16;
            const bool dma = MODE == 0 || (MODE == 2 && (j & 1) == 0);
            if (dma) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p, (__attribute__((address_space(3))) void*)(my + j * 1024), 16, 0, 0);
            else r[j] = *(const u4*)p;
        }
        if (MODE != 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (MODE == 1 || (j & 1)) *(u4*)(my + j * 1024 + lane * 16) = r[j];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += *(unsigned*)(my + ((it * 4 + lane * 16) & 8191));
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
int main() {
    const long window = 2 << 20;
    char* d; unsigned* s;
    hipMalloc(&d, 8 * window); hipMemset(d, 1, 8 * window); hipMalloc(&s, 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000;
    auto run = [&](auto kern, const char* name, int wgs) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), 131072, 0, d, window, 50, s);
        hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), 131072, 0, d, window, iters, s);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double bytes = (double)wgs * iters * 8 * 8 * 1024;
        printf("%-46s %3d WGs: %7.2f ms  %6.1f GB/s per CU  %5.1f B/clk/CU at 2.1 GHz  %5.2f TB/s chip\n", name, wgs, ms, bytes / wgs / ms / 1e6, bytes / wgs / ms / 1e6 / 2.1, bytes / ms / 1e9);
    };
    for (int wgs : {32, 256}) {
        run(k<0>, "LDS-DMA only (global_load_lds_dwordx4)", wgs);
        run(k<1>, "registers only (global_load_dwordx4 + ds_write)", wgs);
        run(k<2>, "half and half", wgs);
    }
    return 0;
}
