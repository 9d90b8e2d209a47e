// Fused HF-AdamW over a flat parameter buffer (K15; transformers.optimization.AdamW 4.10.0 restated from its
// documented update -- parity unpinned): eps is added to sqrt(v) BEFORE the bias-correction
// scaling, decoupled weight decay after the update.  One pass over p/g/m/v; optionally also emits the bf16 shadow
// copy the MFMA kernels read, so the weights are not re-read for the cast.
#include "common.h"

__global__ void adamw_kernel(int64_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             float lr, float b1, float b2, float eps, float wd, float step_size, float grad_scale, bf16* __restrict__ shadow) {
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
        if (i + 3 < n) {
            float4 pp = *(float4*)(p + i), gg = *(const float4*)(g + i), mm = *(float4*)(m + i), vv = *(float4*)(v + i);
            float* P = (float*)&pp; float* G = (float*)&gg; float* M = (float*)&mm; float* V = (float*)&vv;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float gr = G[j] * grad_scale;
                M[j] = M[j] * b1 + gr * (1.f - b1);
                V[j] = V[j] * b2 + gr * gr * (1.f - b2);
                P[j] = P[j] - step_size * (M[j] / (sqrtf(V[j]) + eps));
                if (wd > 0.f) P[j] = P[j] - P[j] * lr * wd;
            }
            *(float4*)(p + i) = pp; *(float4*)(m + i) = mm; *(float4*)(v + i) = vv;
            if (shadow) { bf16x4 s; s[0] = (bf16)P[0]; s[1] = (bf16)P[1]; s[2] = (bf16)P[2]; s[3] = (bf16)P[3]; *(bf16x4*)(shadow + i) = s; }
        } else {
            for (int64_t k = i; k < n; ++k) {
                const float gr = g[k] * grad_scale;
                m[k] = m[k] * b1 + gr * (1.f - b1);
                v[k] = v[k] * b2 + gr * gr * (1.f - b2);
                float x = p[k] - step_size * (m[k] / (sqrtf(v[k]) + eps));
                if (wd > 0.f) x = x - x * lr * wd;
                p[k] = x;
                if (shadow) shadow[k] = (bf16)x;
            }
        }
    }
}

// step is 1-based.  grad_scale lets the caller fold a gradient average (1/world_size) into the update.
extern "C" int dvlp_adamw_step(int64_t n, float* p, const float* g, float* m, float* v, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int64_t step, float grad_scale, void* bf16_shadow, void* stream) {
    dvlp_clear_status();
    if (n <= 0 || step <= 0) return DVLP_ERR_SHAPE;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr * sqrt(bc2) / bc1);
    int64_t blocks = cdiv(n, 1024); if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n, p, g, m, v, lr, beta1, beta2, eps,
                       weight_decay, step_size, grad_scale, (bf16*)bf16_shadow);
    return dvlp_launch_status();
}

// Device-resident hyper-parameters: hyper = {lr, beta1, beta2, eps, weight_decay, grad_scale, step, step_size} (fp32; step is an
// exact integer below 2^24).  The step counter advances ON THE DEVICE, so a hipGraph that captured one optimisation step replays
// with the right bias correction every time, and the host changes lr / grad_scale by writing the buffer (no re-capture).
__global__ void adamw_prep_kernel(float* __restrict__ hyper) {
    const double step = (double)hyper[6] + 1.0;
    const double bc1 = 1.0 - pow((double)hyper[1], step), bc2 = 1.0 - pow((double)hyper[2], step);
    hyper[6] = (float)step;
    hyper[7] = (float)((double)hyper[0] * sqrt(bc2) / bc1);
}
__global__ void adamw_dev_kernel(int64_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                 const float* __restrict__ hyper, bf16* __restrict__ shadow) {
    adamw_dev_elements(n, p, g, m, v, hyper, shadow, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}

extern "C" int dvlp_adamw_step_dev(int64_t n, float* p, const float* g, float* m, float* v, float* hyper, void* bf16_shadow, void* stream) {
    dvlp_clear_status();
    if (n <= 0 || !hyper) return DVLP_ERR_SHAPE;
    int64_t blocks = cdiv(n, 1024); if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(adamw_prep_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, hyper);
    hipLaunchKernelGGL(adamw_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n, p, g, m, v, (const float*)hyper, (bf16*)bf16_shadow);
    return dvlp_launch_status();
}

// The same update in two parts, for a step whose optimizer work is spread over the backward pass: dvlp_adamw_prep_dev advances the
// device step counter / step size ONCE (before the first partial update of the step), dvlp_adamw_range_dev updates one range of the
// flat buffers with the hyper-parameters as they stand (trainer.FusedAdamW.begin_overlapped: a layer's weights are updated on the
// gradient side stream as soon as its weight gradients are final -- the HBM-bound update hides behind the MFMA-bound backward).
extern "C" int dvlp_adamw_prep_dev(float* hyper, void* stream) {
    dvlp_clear_status();
    if (!hyper) return DVLP_ERR_SHAPE;
    hipLaunchKernelGGL(adamw_prep_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, hyper);
    return dvlp_launch_status();
}
extern "C" int dvlp_adamw_range_dev(int64_t n, float* p, const float* g, float* m, float* v, const float* hyper, void* bf16_shadow, void* stream) {
    dvlp_clear_status();
    if (n <= 0 || !hyper) return DVLP_ERR_SHAPE;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15 || ((uintptr_t)bf16_shadow & 7)) return DVLP_ERR_SHAPE;   // vector accesses
    int64_t blocks = cdiv(n, 1024); if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(adamw_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n, p, g, m, v, hyper, (bf16*)bf16_shadow);
    return dvlp_launch_status();
}
