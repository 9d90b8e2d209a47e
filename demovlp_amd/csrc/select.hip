// Region select (K1; data_loader/WebVid_dataset.py:231-283 + :151-228): per frame, order regions by detection
// confidence (descending), keep the first R, edge-pad short frames with the last kept row, build the 0/1 mask and the
// 6-d box geometry.  Pure HBM gather: one workgroup per (clip, frame); ranks come from an all-pairs comparison in LDS
// (Nraw <= 1024); rows move one wave per row, 16-byte loads / 8-byte stores, two rows in flight per wave.  Indices are bit-exact vs numpy for distinct confidences;
// ties resolve as a stable ascending sort reversed (larger index first).
#include "common.h"

constexpr int SEL_FEAT = 2048, SEL_OUT = 2054, SEL_MAXN = 1024;

__global__ __launch_bounds__(256) void region_select_kernel(int F, int Nraw, int R, const float* __restrict__ feats, const float* __restrict__ bbox,
                                                            const float* __restrict__ conf, const float* __restrict__ wh,
                                                            const int* __restrict__ nvalid, float* __restrict__ obj, float* __restrict__ mask,
                                                            int* __restrict__ order, int* __restrict__ lens) {
    __shared__ float c[SEL_MAXN];
    __shared__ int ord[SEL_MAXN];
    const int64_t bf = blockIdx.x;
    const int n = nvalid ? nvalid[bf] : Nraw;
    const float* cf = conf + bf * Nraw;
    for (int t = threadIdx.x; t < n; t += blockDim.x) c[t] = cf[t];
    __syncthreads();
    for (int t = threadIdx.x; t < n; t += blockDim.x) {
        const float me = c[t];
        int rank = 0;
        for (int u = 0; u < n; ++u) rank += (c[u] > me) || (c[u] == me && u > t);
        ord[rank] = t;
    }
    __syncthreads();
    const int keep = n < R ? n : R;
    if (threadIdx.x == 0) lens[bf] = keep;
    const float iw = wh[bf * 2], ih = wh[bf * 2 + 1];
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        mask[bf * R + r] = r < keep ? 1.f : 0.f;
        order[bf * R + r] = r < keep ? ord[r] : -1;
    }
    // One wave per output row, two rows per wave in flight: a lane reads 8 x 16 bytes of each source row (8192-byte rows, 16-byte
    // aligned) -- sixteen loads outstanding before the first store -- and writes them as 8-byte pieces (destination rows are 8216
    // bytes: 8-byte aligned only).  (One row at a time per workgroup with 8-byte accesses left the gather latency-bound.)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int r0 = wid; r0 < R; r0 += 8) {
        const int r1 = r0 + 4;
        const int s0 = ord[r0 < keep ? r0 : keep - 1];                    // np.pad(..., 'edge')
        const int s1 = ord[(r1 < R ? r1 : r0) < keep ? (r1 < R ? r1 : r0) : keep - 1];
        const f32x4* f0 = (const f32x4*)(feats + (bf * Nraw + s0) * SEL_FEAT);
        const f32x4* f1 = (const f32x4*)(feats + (bf * Nraw + s1) * SEL_FEAT);
        f32x4 v[2][8];                                                    // 2 rows x 2048 floats / 64 lanes
#pragma unroll
        for (int u = 0; u < 8; ++u) { v[0][u] = __builtin_nontemporal_load(f0 + lane + 64 * u); v[1][u] = __builtin_nontemporal_load(f1 + lane + 64 * u); }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = q == 0 ? r0 : r1;
            if (r >= R) break;
            float* orow = obj + (bf * R + r) * SEL_OUT;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float2* d = (float2*)(orow + 4 * (lane + 64 * u));
                d[0] = make_float2(v[q][u][0], v[q][u][1]); d[1] = make_float2(v[q][u][2], v[q][u][3]);
            }
            if (lane == 0) {
                const float* b = bbox + (bf * Nraw + (q == 0 ? s0 : s1)) * 4;
                const float sw = __fdiv_rn(b[2] - b[0], iw), sh = __fdiv_rn(b[3] - b[1], ih);
                const float sx = __fdiv_rn(b[0], iw), sy = __fdiv_rn(b[1], ih);
                orow[SEL_FEAT + 0] = sx; orow[SEL_FEAT + 1] = sy;
                orow[SEL_FEAT + 2] = __fadd_rn(sx, sw); orow[SEL_FEAT + 3] = __fadd_rn(sy, sh);
                orow[SEL_FEAT + 4] = sw; orow[SEL_FEAT + 5] = sh;
            }
        }
    }
}

// feats [B*F][Nraw][2048], bbox [B*F][Nraw][4], conf [B*F][Nraw], wh [B*F][2] (image w,h), nvalid [B*F] or null
// -> obj [B*F][R][2054], mask [B*F][R] (1/0), order [B*F][R] (source index, -1 = pad), lens [B*F]
extern "C" int dvlp_region_select(int64_t BF, int64_t F, int64_t Nraw, int64_t R, const float* feats, const float* bbox, const float* conf,
                                  const float* wh, const int* nvalid, float* obj, float* mask, int* order, int* lens, void* stream) {
    dvlp_clear_status();
    if (BF <= 0 || Nraw <= 0 || Nraw > SEL_MAXN || R <= 0) return DVLP_ERR_SHAPE;
    hipLaunchKernelGGL(region_select_kernel, dim3((unsigned)BF), dim3(256), 0, (hipStream_t)stream, (int)F, (int)Nraw, (int)R, feats, bbox,
                       conf, wh, nvalid, obj, mask, order, lens);
    return dvlp_launch_status();
}
