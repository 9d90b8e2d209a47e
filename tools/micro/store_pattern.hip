// Per-CU store rate of a 256 x 256 bf16 output tile by store-instruction shape (one 512-thread workgroup per CU, as the 256-row GEMM's
// epilogue): how many CONTIGUOUS bytes of one output row a wave's global_store_dwordx4 covers -- 64 (the epilogue of round 2: 16 rows
// x 64 B), 128, 256 or 512 (a whole tile row).  Build: hipcc --offload-arch=gfx950 -O3 -w tools/micro/store_pattern.hip -o tools/micro/store_pattern.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
template <int S, bool NT>
__global__ __launch_bounds__(512) void k(char* out, long ld_bytes, int tiles_per_wg, int ntile_cols) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int R = 1024 / S, CB = 512 / S;                  // rows per instruction, column blocks per tile
    const int row_in = (lane * 16) / S, col_in = (lane * 16) % S;
    u4 v = {(unsigned)lane, (unsigned)w, blockIdx.x, 7u};
    for (int t = 0; t < tiles_per_wg; ++t) {
        const long tile = (long)blockIdx.x * tiles_per_wg + t;
        char* base = out + (tile / ntile_cols) * 256 * ld_bytes + (tile % ntile_cols) * 512;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int b = w * 16 + i, rb = b / CB, cb = b % CB;
            u4* p = (u4*)(base + (long)(rb * R + row_in) * ld_bytes + cb * S + col_in);
            if (NT) __builtin_nontemporal_store(v, p); else *p = v;
        }
    }
}
int main() {
    const int ncols = 12, tiles_per_wg = 8, nwg = 256;       // N = 3072: 12 column tiles
    const long ld = 3072 * 2, rows = (long)(nwg * tiles_per_wg + ncols - 1) / ncols * 256;
    char* d; hipMalloc(&d, rows * ld);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](auto kern, const char* name, int wgs) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), 0, 0, d, ld, tiles_per_wg, ncols);
        hipEventRecord(a);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), 0, 0, d, ld, tiles_per_wg, ncols);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double us = ms * 100, bytes = (double)wgs * tiles_per_wg * 131072;
        printf("%-34s %3d WGs: %8.1f us  %6.2f us per 128-KB tile per CU  %6.1f GB/s per CU  %5.2f TB/s chip\n", name, wgs, us, us / tiles_per_wg, 131072.0 * tiles_per_wg / us / 1e3, bytes / us / 1e6);
    };
    for (int wgs : {64, 256}) {
        run(k<64, true>, "16 rows x   64 B per instr, nt", wgs);
        run(k<128, true>, " 8 rows x  128 B per instr, nt", wgs);
        run(k<256, true>, " 4 rows x  256 B per instr, nt", wgs);
        run(k<512, true>, " 2 rows x  512 B per instr, nt", wgs);
        run(k<64, false>, "16 rows x   64 B per instr, plain", wgs);
        run(k<128, false>, " 8 rows x  128 B per instr, plain", wgs);
        run(k<512, false>, " 2 rows x  512 B per instr, plain", wgs);
    }
    return 0;
}
