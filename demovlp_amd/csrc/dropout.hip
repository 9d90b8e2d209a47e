// Dropout of the text tower.  The reference leaves DistilBERT in train mode (model/model.py:29-30), so every training step
// applies HuggingFace DistilBERT's three dropouts (config.dropout = config.attention_dropout = 0.1):
//   embeddings            LayerNorm(word + position) -> dropout
//   self-attention        softmax(scores) -> dropout -> . V                 (probabilities, before the context product)
//   feed-forward          lin2(gelu(lin1(x))) -> dropout -> + residual
// Masks are Philox4x32-10 streams keyed by (seed, step offset, site, element): the state {seed_lo, seed_hi, offset} lives in
// DEVICE memory and dvlp_dropout_advance bumps the offset once per forward, so a captured hipGraph of the training step
// draws fresh masks at every replay.  Forward kernels also store the keep bytes (1 byte per element; the backward re-applies
// them), which keeps forward/backward consistent whatever happens to the state in between.
#include "common.h"

__global__ void dropout_advance_kernel(uint32_t* state) { state[2] += 1u; }

template <typename T>
__global__ __launch_bounds__(256) void dropout_fwd_kernel(int64_t n4, const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                                                          uint8_t* __restrict__ keep, uint32_t thr, float scale, const uint32_t* __restrict__ state, uint32_t site) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    uint32_t u[4];
    philox4x32_10((uint32_t)i, (uint32_t)(i >> 32), site, state[2], state[0], state[1], u);
    uint32_t kb = 0;
    float o[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const bool k = u[t] >= thr;
        kb |= (k ? 1u : 0u) << (8 * t);
        o[t] = (k ? to_f(x[4 * i + t]) * scale : 0.f) + (res ? to_f(res[4 * i + t]) : 0.f);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) y[4 * i + t] = from_f<T>(o[t]);
    *(uint32_t*)(keep + 4 * i) = kb;
}

template <typename T>
__global__ __launch_bounds__(256) void dropout_bwd_kernel(int64_t n4, const T* __restrict__ dy, const uint8_t* __restrict__ keep, float scale, T* __restrict__ dx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const uint32_t kb = *(const uint32_t*)(keep + 4 * i);
#pragma unroll
    for (int t = 0; t < 4; ++t) dx[4 * i + t] = from_f<T>(((kb >> (8 * t)) & 1u) ? to_f(dy[4 * i + t]) * scale : 0.f);
}

// keep[bh][q][key] and keepT[bh][key][q] (row stride Ns = N rounded up to 16, pads 0): one thread per 4 x 4 block (4 queries x 4
// keys = 4 Philox calls), so both orientations leave as 4-byte words -- byte-wide scattered writes of the transposed copy cost 5x
__global__ __launch_bounds__(256) void dropout_attn_mask_kernel(int64_t BH, int N, int Ns, uint8_t* __restrict__ keep, uint8_t* __restrict__ keepT,
                                                                uint32_t thr, const uint32_t* __restrict__ state, uint32_t site) {
    const int ng = Ns / 4;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= BH * ng * ng) return;
    const int jg = (int)(idx % ng), qg = (int)((idx / ng) % ng);
    const int64_t bh = idx / ((int64_t)ng * ng);
    const uint32_t off = state[2], s0 = state[0], s1 = state[1];
    uint32_t kq[4], kt[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int q = 4 * qg + a;
        uint32_t u[4] = {0u, 0u, 0u, 0u};
        if (q < N) philox4x32_10((uint32_t)(bh * N + q), (uint32_t)jg, site, off, s0, s1, u);
        uint32_t kb = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint32_t k = (q < N && 4 * jg + t < N && u[t] >= thr) ? 1u : 0u;
            kb |= k << (8 * t);
            kt[t] |= k << (8 * a);
        }
        kq[a] = kb;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        *(uint32_t*)(keep + (bh * Ns + 4 * qg + a) * Ns + 4 * jg) = kq[a];
        *(uint32_t*)(keepT + (bh * Ns + 4 * jg + a) * Ns + 4 * qg) = kt[a];
    }
}

static inline uint32_t drop_threshold(float p) { const double t = (double)p * 4294967296.0; return t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t; }

extern "C" int dvlp_dropout_advance(void* state, void* stream) {
    dvlp_clear_status();
    hipLaunchKernelGGL(dropout_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (uint32_t*)state);
    return dvlp_launch_status();
}

extern "C" int dvlp_dropout_fwd(int dtype, int64_t n, const void* x, const void* res, void* y, void* keep, float p, const void* state, int site, void* stream) {
    dvlp_clear_status();
    if (n <= 0 || n % 4 || p < 0.f || p >= 1.f) return DVLP_ERR_SHAPE;
    const int64_t n4 = n / 4;
    const dim3 grid((unsigned)cdiv(n4, 256)), block(256);
    const uint32_t thr = drop_threshold(p);
    const float scale = 1.f / (1.f - p);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DVLP_F32) hipLaunchKernelGGL(dropout_fwd_kernel<float>, grid, block, 0, st, n4, (const float*)x, (const float*)res, (float*)y, (uint8_t*)keep, thr, scale, (const uint32_t*)state, (uint32_t)site);
    else if (dtype == DVLP_BF16) hipLaunchKernelGGL(dropout_fwd_kernel<bf16>, grid, block, 0, st, n4, (const bf16*)x, (const bf16*)res, (bf16*)y, (uint8_t*)keep, thr, scale, (const uint32_t*)state, (uint32_t)site);
    else return DVLP_ERR_DTYPE;
    return dvlp_launch_status();
}

extern "C" int dvlp_dropout_bwd(int dtype, int64_t n, const void* dy, const void* keep, float p, void* dx, void* stream) {
    dvlp_clear_status();
    if (n <= 0 || n % 4 || p < 0.f || p >= 1.f) return DVLP_ERR_SHAPE;
    const int64_t n4 = n / 4;
    const dim3 grid((unsigned)cdiv(n4, 256)), block(256);
    const float scale = 1.f / (1.f - p);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DVLP_F32) hipLaunchKernelGGL(dropout_bwd_kernel<float>, grid, block, 0, st, n4, (const float*)dy, (const uint8_t*)keep, scale, (float*)dx);
    else if (dtype == DVLP_BF16) hipLaunchKernelGGL(dropout_bwd_kernel<bf16>, grid, block, 0, st, n4, (const bf16*)dy, (const uint8_t*)keep, scale, (bf16*)dx);
    else return DVLP_ERR_DTYPE;
    return dvlp_launch_status();
}

extern "C" int dvlp_dropout_attn_mask(int64_t BH, int64_t N, float p, const void* state, int site, void* keep, void* keepT, void* stream) {
    dvlp_clear_status();
    if (BH <= 0 || N <= 0 || p < 0.f || p >= 1.f) return DVLP_ERR_SHAPE;
    const int Ns = (int)((N + 15) / 16 * 16);
    const int64_t total = BH * (Ns / 4) * (Ns / 4);
    hipLaunchKernelGGL(dropout_attn_mask_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, BH, (int)N, Ns, (uint8_t*)keep,
                       (uint8_t*)keepT, drop_threshold(p), (const uint32_t*)state, (uint32_t)site);
    return dvlp_launch_status();
}

// the device Philox against the published known-answer vectors (tests/test_gpu_round2.py): out[4] <- philox(ctr[4], key[2])
__global__ void philox_kat_kernel(const uint32_t* ck, uint32_t* out) {
    uint32_t u[4];
    philox4x32_10(ck[0], ck[1], ck[2], ck[3], ck[4], ck[5], u);
    out[0] = u[0]; out[1] = u[1]; out[2] = u[2]; out[3] = u[3];
}
extern "C" int dvlp_philox_kat(const void* ctr_key, void* out, void* stream) {
    dvlp_clear_status();
    hipLaunchKernelGGL(philox_kat_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (const uint32_t*)ctr_key, (uint32_t*)out);
    return dvlp_launch_status();
}
