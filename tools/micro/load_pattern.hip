// Does the access SHAPE of the space-attention kernels' operand loads cap their bandwidth?  Each wave reads the q, k and v rows of one
// (sample, head, frame) item out of the packed [B * N, 2304] bf16 projection -- 37 rows x 128 bytes per operand, 4608 bytes apart --
//   shape 0  "fragment-shaped": lane (row = lane & 15, chunk = lane >> 4): 16 rows x 64 B per instruction (what a 16x16x32 MFMA operand wants)
//   shape 1  "full lines":       lane (row = lane >> 3, chunk = lane & 7):  8 rows x 128 B per instruction
// and writes one value per wave.  hipcc --offload-arch=gfx950 -O3 tools/micro/load_pattern.hip -o tools/micro/load_pattern.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned short u16;
template <int SHAPE>
__global__ __launch_bounds__(256) void rd(const uint4* __restrict__ qkv, float* __restrict__ out, int B, int H, int F, int R, int N) {
    const int lane = threadIdx.x & 63, item = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int f = item % F, h = (item / F) % H, b = item / (F * H);
    if (b >= B) return;
    const long row0 = (long)b * N + 1 + (long)f * R;
    unsigned acc = 0;
#pragma unroll
    for (int op = 0; op < 3; ++op) {
        const char* base = (const char*)qkv + row0 * 4608 + op * 1536 + h * 128;
        if (SHAPE == 0) {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                int r = 16 * t + (lane & 15); r = r < R ? r : R - 1;
                const uint4 a = *(const uint4*)(base + (long)r * 4608 + 16 * (lane >> 4));
                const uint4 c = *(const uint4*)(base + (long)r * 4608 + 64 + 16 * (lane >> 4));
                acc += a.x ^ a.y ^ a.z ^ a.w ^ c.x ^ c.y ^ c.z ^ c.w;
            }
        } else {
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                int r = 8 * t + (lane >> 3); r = r < R ? r : R - 1;
                const uint4 a = *(const uint4*)(base + (long)r * 4608 + 16 * (lane & 7));
                acc += a.x ^ a.y ^ a.z ^ a.w;
            }
        }
    }
    if (acc == 0x12345678u) out[item] = 1.f;
}
int main() {
    const int B = 64, H = 12, F = 8, R = 36, N = 1 + F * R;
    const size_t bytes = (size_t)B * N * 4608;
    uint4* d; float* o;
    hipMalloc(&d, bytes); hipMalloc(&o, (size_t)B * H * F * 4);
    hipMemset(d, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int items = B * H * F;
    for (int shape = 0; shape < 2; ++shape)
        for (int rep = 0; rep < 2; ++rep) {
            for (int w = 0; w < 3; ++w) { if (shape == 0) rd<0><<<items / 4, 256>>>(d, o, B, H, F, R, N); else rd<1><<<items / 4, 256>>>(d, o, B, H, F, R, N); }
            hipEventRecord(e0);
            const int n = 50;
            for (int w = 0; w < n; ++w) { if (shape == 0) rd<0><<<items / 4, 256>>>(d, o, B, H, F, R, N); else rd<1><<<items / 4, 256>>>(d, o, B, H, F, R, N); }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double us = 1e3 * ms / n, useful = (double)items * 3 * R * 128;
            printf("shape %d (%s): %7.1f us per pass over q, k, v of %d items  %.2f TB/s of useful bytes (%.0f MB; the tensor is %.0f MB)\n", shape,
                   shape ? "8 rows x 128 B per instruction" : "16 rows x 64 B per instruction", us, items, useful / us / 1e6, useful / 1e6, bytes / 1e6);
        }
    return 0;
}
