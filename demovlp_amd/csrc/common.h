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

// Developer switches (dvlp_dev_*: A/B measurements, timing ablations, forced code paths; include/demovlp_hip_dev.h).  Only the
// -DDVLP_DEV build (libdemovlp_hip_dev.so) exports their setters; in the product library a setter is an unreferenced internal
// function, nothing can write the switch and the compiler folds it to its default.
#ifdef DVLP_DEV
#define DVLP_DEV_API extern "C"
#else
#define DVLP_DEV_API [[maybe_unused]] static
#endif

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

// Wave-wide (64-lane) all-reduce on the VALU: four DPP butterfly steps inside each 16-lane row (quad_perm x2,
// row_half_mirror, row_mirror), then the four rows through v_permlane16_swap / v_permlane32_swap (no scalar round trip:
// the v_readlane form cost ~0.6 % of the step in the reduction-heavy loss kernels).  8 VALU instructions; the
// ds_bpermute-based __shfl_xor ladder these replace cost six dependent LDS round trips.
template <int CTRL>
__device__ __forceinline__ float dvlp_dpp(float x) {
    // old = 0 with bound_ctrl: every pattern used here has a valid source for every lane, and in this form the compiler folds the
    // move into the consuming add (v_add_f32_dpp) instead of emitting v_mov_b32_dpp + v_add_f32
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dvlp_dpp<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dvlp_dpp<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dvlp_dpp<0x141>(v);     // row_half_mirror
    v += dvlp_dpp<0x140>(v);     // row_mirror
    // the four 16-lane rows: v_permlane16_swap / v_permlane32_swap of the value with itself (xor-16, xor-32 steps), all VALU
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dvlp_dpp<0xB1>(v));
    v = fmaxf(v, dvlp_dpp<0x4E>(v));
    v = fmaxf(v, dvlp_dpp<0x141>(v));
    v = fmaxf(v, dvlp_dpp<0x140>(v));
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
// all-reduce over the four lanes {c, c+16, c+32, c+48} that share lane&15 (the MFMA attention kernels' per-query
// statistics in the "S^T" layout): v_permlane16_swap / v_permlane32_swap of a value with itself leave the pair's two
// members in the two results, so one swap + one add is an xor-16 (xor-32) all-reduce step.
__device__ __forceinline__ float col4_sum(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
__device__ __forceinline__ float col4_max(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
// all-reduce inside each aligned group of 8 lanes (a 64-channel row read as 8 lanes x 8 channels)
__device__ __forceinline__ float oct_sum(float v) {
    v += dvlp_dpp<0xB1>(v); v += dvlp_dpp<0x4E>(v); v += dvlp_dpp<0x141>(v);
    return v;
}
// all-reduce over the eight lanes {c, c+8, ..., c+56} that share lane&7: row_ror:8 is the xor-8 step inside a 16-lane row
__device__ __forceinline__ float stride8_sum(float v) {
    v += dvlp_dpp<0x128>(v);
    return col4_sum(v);
}
// all-reduce inside each 16-lane row only (the MFMA attention kernels' per-query statistics in the "S" layout)
__device__ __forceinline__ float row16_sum(float v) {
    v += dvlp_dpp<0xB1>(v); v += dvlp_dpp<0x4E>(v); v += dvlp_dpp<0x141>(v); v += dvlp_dpp<0x140>(v);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dvlp_dpp<0xB1>(v)); v = fmaxf(v, dvlp_dpp<0x4E>(v)); v = fmaxf(v, dvlp_dpp<0x141>(v)); v = fmaxf(v, dvlp_dpp<0x140>(v));
    return v;
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * expf(-0.5f * x * x);
}

// GELU / GELU' of the bf16 GEMM epilogues (every bf16 kernel, vector and scalar paths alike), two elements at a time
// (v_pk_fma_f32).  History: libm erff / expf cost ~100 VALU per element (the fc1 backward GEMM spent as long in its epilogue as in
// its MFMAs); an Abramowitz-Stegun rational erf with one shared v_exp_f32 and one v_rcp_f32 (round 1) still was 2 quarter-rate +
// ~14 full-rate instructions per element, and with one workgroup per CU nothing overlaps them -- the fc1 forward / fc2 backward
// products spent ~25 % of their time there.  Odd minimax polynomials in z = clamp(x / A, -1, 1)
// (tools/fit_gelu_poly.py, errors include the fp32 Horner evaluation):
//   Phi(x)   = 0.5 + z P(z^2), A = 4.0, |err| <= 6e-6 inside [-4, 4] (gelu: 2.3e-5), tails pinned at Phi(+-4)
//   gelu'(x) = 0.5 + z Q(z^2), A = 4.5, |err| <= 1.8e-4
// both far below the bf16 resolution of the values they produce.  The fp32 kernels keep erff / expf (1e-4 parity path).
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr float GELU_PHI_C[9] = {1.595690840e+00f, -4.250278161e+00f, 1.011453720e+01f, -1.857818477e+01f, 2.592817434e+01f, -2.640316869e+01f, 1.822784375e+01f, -7.519930336e+00f, 1.385288103e+00f};
constexpr float GELU_DGELU_C[10] = {3.589317633e+00f, -2.414135244e+01f, 1.081506568e+02f, -3.294218669e+02f, 7.116908195e+02f, -1.091321666e+03f, 1.154870954e+03f, -7.967612061e+02f, 3.208072879e+02f, -5.696292942e+01f};
template <int NC>
__device__ __forceinline__ f32x2 odd_poly2(const float (&c)[NC], f32x2 z) {
    const f32x2 s = z * z;
    f32x2 acc = {c[NC - 1], c[NC - 1]};
#pragma unroll
    for (int k = NC - 2; k >= 0; --k) acc = __builtin_elementwise_fma(acc, s, (f32x2){c[k], c[k]});
    return __builtin_elementwise_fma(z, acc, (f32x2){0.5f, 0.5f});
}
__device__ __forceinline__ f32x2 gelu_poly2(f32x2 x) {
    const f32x2 z = {__builtin_amdgcn_fmed3f(x[0] * 0.25f, -1.f, 1.f), __builtin_amdgcn_fmed3f(x[1] * 0.25f, -1.f, 1.f)};
    return x * odd_poly2(GELU_PHI_C, z);
}
__device__ __forceinline__ f32x2 gelu_grad_poly2(f32x2 x) {
    constexpr float ia = 1.f / 4.5f;
    const f32x2 z = {__builtin_amdgcn_fmed3f(x[0] * ia, -1.f, 1.f), __builtin_amdgcn_fmed3f(x[1] * ia, -1.f, 1.f)};
    return odd_poly2(GELU_DGELU_C, z);
}

// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11): counter-based, so a dropout mask is a
// pure function of (seed, step offset, site, element index) -- reproducible on the host (oracle/restatement.py:philox4x32_10,
// pinned by the published known-answer vectors) and independent of launch geometry.
__host__ __device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
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


// One HF-AdamW pass over elements [0, n) by thread `tid` of `nthreads` (4 elements per thread and iteration; hyper = {lr, beta1, beta2, eps,
// weight_decay, grad_scale, step, step_size} in device memory).  Shared by adamw_dev_kernel (csrc/optim.hip) and by the spare workgroups of the
// grouped weight-gradient launch (csrc/gemm.hip: the previous layer's update rides on the CUs that launch leaves idle).
__device__ __forceinline__ void adamw_dev_elements(int64_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   const float* __restrict__ hyper, bf16* __restrict__ shadow, int64_t tid, int64_t nthreads) {
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], grad_scale = hyper[5], step_size = hyper[7];
    for (int64_t i = tid * 4; i < n; i += nthreads * 4) {
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
