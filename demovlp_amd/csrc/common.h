// Shared device helpers for the DemoVLP gfx950 kernels.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { DVLP_F32 = 0, DVLP_BF16 = 1 };
enum { DVLP_OK = 0, DVLP_ERR_DTYPE = -1, DVLP_ERR_SHAPE = -2, DVLP_ERR_LAUNCH = -3, DVLP_ERR_UNSUPPORTED = -4 };

// epilogue flags of dvlp_gemm (keep in sync with include/demovlp_hip.h)
enum {
    EPI_GELU = 1,       // aux <- pre-activation, C <- gelu_erf(v)
    EPI_GELU_BWD = 2,   // v *= gelu'(aux)
    EPI_RELU_BWD = 4,   // v = aux > 0 ? v : 0
    EPI_ACCUM = 8,      // C += v
    EPI_OUT_F32 = 32,   // C is float* regardless of the compute dtype (weight gradients go straight to fp32)
    EPI_LEAKY = 16,     // v = v > 0 ? v : 0.1 v   (LeakyReLU(0.1), model/loss.py:236)
};

template <typename T> __device__ __forceinline__ float to_f(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v) { return (T)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * expf(-0.5f * x * x);
}

// hipGetLastError() is sticky per thread and also reports errors left behind by OTHER libraries' benign failed calls
// (e.g. a failed attribute query inside the framework), so judge a launch by the error state it changes: clear before
// launching (dvlp_clear_status) and read after (dvlp_launch_status).
extern int g_dvlp_last_hip_error;      // defined in gemm.hip; read through dvlp_last_error_string()
static inline void dvlp_clear_status() { (void)hipGetLastError(); }
static inline int dvlp_launch_status() {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return DVLP_OK;
    g_dvlp_last_hip_error = (int)e;
    return DVLP_ERR_LAUNCH;
}
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
