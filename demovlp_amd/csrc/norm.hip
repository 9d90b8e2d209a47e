// LayerNorm forward/backward and column reductions (bias gradients).  HBM-bound: one wave per row, 8/16-byte vector
// accesses, statistics in fp32, two-stage deterministic column reductions (no float atomics).
#include "common.h"

constexpr int LN_MAXC = 4;   // D = 256 * NC, NC <= 4  (D = 768 on this path)

template <typename T> struct Vec4;
template <> struct Vec4<float> {
    static __device__ __forceinline__ void load(const float* p, float (&o)[4]) { float4 v = *(const float4*)p; o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
    static __device__ __forceinline__ void store(float* p, const float (&o)[4]) { *(float4*)p = make_float4(o[0], o[1], o[2], o[3]); }
};
template <> struct Vec4<bf16> {
    static __device__ __forceinline__ void load(const bf16* p, float (&o)[4]) { bf16x4 v = *(const bf16x4*)p; o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2]; o[3] = (float)v[3]; }
    static __device__ __forceinline__ void store(bf16* p, const float (&o)[4]) { bf16x4 v; v[0] = (bf16)o[0]; v[1] = (bf16)o[1]; v[2] = (bf16)o[2]; v[3] = (bf16)o[3]; *(bf16x4*)p = v; }
};

// y = (x - mean) * rstd * gamma + beta ; optional y_relu = max(y, 0)
template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(int64_t M, int NC, const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, T* __restrict__ y, T* __restrict__ y_relu,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int D = NC * 256;
    float v[LN_MAXC][4];
    float s = 0.f;
    for (int c = 0; c < NC; ++c) {
        Vec4<T>::load(x + row * D + c * 256 + lane * 4, v[c]);
        s += v[c][0] + v[c][1] + v[c][2] + v[c][3];
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = v[c][j] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    for (int c = 0; c < NC; ++c) {
        float g[4], b[4], o[4];
        Vec4<float>::load(gamma + c * 256 + lane * 4, g);
        Vec4<float>::load(beta + c * 256 + lane * 4, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (v[c][j] - mean) * rstd * g[j] + b[j];
        Vec4<T>::store(y + row * D + c * 256 + lane * 4, o);
        if (y_relu) {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j], 0.f);
            Vec4<T>::store(y_relu + row * D + c * 256 + lane * 4, o);
        }
    }
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// bf16, D = 768 (every LayerNorm of both towers): HALF a wave per row, three 16-byte loads per lane, so a wave keeps two rows (3 KB)
// in flight with a third of the load instructions -- the row-per-wave form above reaches 4.3 TB/s on the [18496, 768] tokens
// (one workgroup = 8 rows).
__device__ __forceinline__ float ln_half_sum(float v) {
    v = row16_sum(v);
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__global__ __launch_bounds__(256) void ln_fwd768_bf16_kernel(int64_t M, const bf16* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, bf16* __restrict__ y, bf16* __restrict__ y_relu,
                                                             float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    const int lane = threadIdx.x & 63, hl = lane & 31;
    int64_t row = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 6) * 2 + (lane >> 5);
    const bool live = row < M;
    row = live ? row : M - 1;
    const bf16* xr = x + row * 768;
    float v[3][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const bf16x8 t = *(const bf16x8*)(xr + c * 256 + hl * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[c][j] = (float)t[j]; s += v[c][j]; }
    }
    const float mean = ln_half_sum(s) * (1.f / 768.f);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = v[c][j] - mean; q += d * d; }
    const float rstd = rsqrtf(ln_half_sum(q) * (1.f / 768.f) + eps);
    if (!live) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float4 g0 = *(const float4*)(gamma + c * 256 + hl * 8), g1 = *(const float4*)(gamma + c * 256 + hl * 8 + 4);
        const float4 b0 = *(const float4*)(beta + c * 256 + hl * 8), b1 = *(const float4*)(beta + c * 256 + hl * 8 + 4);
        const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        bf16x8 o, orl;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float t = (v[c][j] - mean) * rstd * g[j] + b[j]; o[j] = (bf16)t; orl[j] = (bf16)fmaxf(t, 0.f); }
        *(bf16x8*)(y + row * 768 + c * 256 + hl * 8) = o;
        if (y_relu) *(bf16x8*)(y_relu + row * 768 + c * 256 + hl * 8) = orl;
    }
    if (hl == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) [+ dres],  g = dy * gamma ;  partial dgamma/dbeta per block.
template <typename T, int CS, int NCT>
__global__ __launch_bounds__(256) void ln_bwd_kernel(int64_t M, int NC_, const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                     const float* __restrict__ rstd_in, const T* __restrict__ dres, T* __restrict__ dx,
                                                     float* __restrict__ partial /* [grid][2 + CS][D] */) {
    constexpr int cs = CS;      // compile-time: a run-time test inside the row loop cost the kernel ~15 %
    constexpr int NC = NCT;     // compile-time too: sized for D = 1024 the per-lane arrays pushed the CS variant past 128 VGPRs
    (void)NC_;                  // (3 waves per SIMD instead of 4: its 1024 workgroups then needed a second round)
    // cs: also emit the column sums of the OUTPUT dx (third plane): dx feeds a Linear whose bias gradient is exactly that sum,
    // which saves a separate pass over dx (values are summed as stored, i.e. after rounding to T, like that pass would)
    extern __shared__ __attribute__((aligned(16))) float red_[];      // [4 waves][2 + cs planes][D]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int D = NC * 256;
    float dg[NCT][4], db[NCT][4], gm[NCT][4], dc[NCT][4];
    for (int c = 0; c < NC; ++c) {
        Vec4<float>::load(gamma + c * 256 + lane * 4, gm[c]);
#pragma unroll
        for (int j = 0; j < 4; ++j) { dg[c][j] = 0.f; db[c][j] = 0.f; dc[c][j] = 0.f; }
    }
    for (int64_t row = (int64_t)blockIdx.x * 4 + wid; row < M; row += (int64_t)gridDim.x * 4) {
        const float mean = mean_in[row], rstd = rstd_in[row];
        float xh[NCT][4], g[NCT][4];
        float s1 = 0.f, s2 = 0.f;
        for (int c = 0; c < NC; ++c) {
            float d[4];
            Vec4<T>::load(dy + row * D + c * 256 + lane * 4, d);
            Vec4<T>::load(x + row * D + c * 256 + lane * 4, xh[c]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xh[c][j] = (xh[c][j] - mean) * rstd;
                g[c][j] = d[j] * gm[c][j];
                s1 += g[c][j];
                s2 += g[c][j] * xh[c][j];
                dg[c][j] += d[j] * xh[c][j];
                db[c][j] += d[j];
            }
        }
        s1 = wave_sum(s1) / (float)D;
        s2 = wave_sum(s2) / (float)D;
        for (int c = 0; c < NC; ++c) {
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = rstd * (g[c][j] - s1 - xh[c][j] * s2);
            if (dres) {
                float r[4];
                Vec4<T>::load(dres + row * D + c * 256 + lane * 4, r);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] += r[j];
            }
            Vec4<T>::store(dx + row * D + c * 256 + lane * 4, o);
            if (cs) {
#pragma unroll
                for (int j = 0; j < 4; ++j) dc[c][j] += to_f(from_f<T>(o[j]));
            }
        }
    }
    const int planes = 2 + (cs ? 1 : 0);
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = c * 256 + lane * 4 + j;
            red_[(wid * planes + 0) * D + col] = dg[c][j]; red_[(wid * planes + 1) * D + col] = db[c][j];
            if (cs) red_[(wid * planes + 2) * D + col] = dc[c][j];
        }
    __syncthreads();
    for (int i = threadIdx.x; i < planes * D; i += 256)
        partial[(int64_t)blockIdx.x * planes * D + i] = red_[i] + red_[planes * D + i] + red_[2 * planes * D + i] + red_[3 * planes * D + i];
}

// out[g][c] (+)= sum_p in[(g*P + p)*C + c].  32 columns x 8 partial-sum lanes per workgroup: the P-loop is split
// 8 ways so a [512 x 768] partial buffer reduces in ~64 dependent loads per thread instead of 512.
__global__ __launch_bounds__(256) void reduce_groups_kernel(int64_t P, int64_t C, const float* __restrict__ in, float* __restrict__ out, int accumulate) {
    __shared__ float red[8][33];
    const int cl = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int64_t c = (int64_t)blockIdx.x * 32 + cl;
    const int64_t g = blockIdx.y;
    float s = 0.f;
    if (c < C)
        for (int64_t p = q; p < P; p += 8) s += in[(g * P + p) * C + c];
    red[q][cl] = s;
    __syncthreads();
    if (q == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][cl];
        out[g * C + c] = accumulate ? out[g * C + c] + t : t;
    }
}

// stage 1 of a (grouped) column sum.  Row m of group g lives at x + g*gstride + (m / inner)*ostride + (m % inner)*ld.
// partial[g][p][n] = sum over the p-th row chunk of x[row][n].
// A workgroup covers 256 columns x 8 rows per step: each thread owns 8 consecutive columns (one 16-byte load for bf16,
// two for fp32) of every 8th row, then the 8 row-lanes are combined through LDS.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(int64_t M, int64_t N, const T* __restrict__ x, int64_t ld, int64_t inner, int64_t ostride,
                                                     int64_t gstride, int64_t rows_per, float* __restrict__ partial, int vec,
                                                     unsigned* __restrict__ counters, float* __restrict__ out, int accumulate) {
    __shared__ float red[8][256 + 8];
    __shared__ int s_last;
    const int cc = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int64_t n = (int64_t)blockIdx.x * 256 + cc * 8;
    const int64_t g = blockIdx.z;
    const int64_t m0 = (int64_t)blockIdx.y * rows_per, m1 = m0 + rows_per < M ? m0 + rows_per : M;
    const T* base = x + g * gstride;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (n < N) {
        const bool full = vec && n + 7 < N;
        for (int64_t m = m0 + rl; m < m1; m += 8) {
            const T* row = base + (m / inner) * ostride + (m % inner) * ld + n;
            if (full) {
                if (sizeof(T) == 2) {
                    const bf16x8 v = *(const bf16x8*)row;
#pragma unroll
                    for (int t = 0; t < 8; ++t) acc[t] += (float)v[t];
                } else {
                    const float4 a = *(const float4*)row, b = *(const float4*)((const float*)row + 4);
                    acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w; acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
                }
            } else {
#pragma unroll
                for (int t = 0; t < 8; ++t) if (n + t < N) acc[t] += to_f(row[t]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) red[rl][cc * 8 + t] = acc[t];
    __syncthreads();
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col < N) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += red[k][threadIdx.x];
        partial[(g * gridDim.y + blockIdx.y) * N + col] = s;
    }
    if (!counters) return;
    // Fused second stage (saves a launch per bias gradient): the LAST row-chunk workgroup of this (group, column block)
    // to arrive sums the partials.  Placement-independent hand-off: every storing wave drains its stores, the workgroup
    // barrier, then one lane releases at agent scope and takes a ticket; the last arriver acquires at agent scope before
    // any lane reads the other workgroups' partials (cdna_hip_programming.md Guideline 16, counter form).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned* cnt = counters + g * gridDim.x + blockIdx.x;
        const unsigned ticket = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = ticket == gridDim.y - 1;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // re-arm for the next call
        }
        s_last = last;
    }
    __syncthreads();
    if (s_last && col < N) {
        float t = 0.f;
        const float* src = partial + g * gridDim.y * N + col;
        for (unsigned p = 0; p < gridDim.y; ++p) t += src[(int64_t)p * N];
        out[g * N + col] = accumulate ? out[g * N + col] + t : t;
    }
}

static int g_ln_wide = 1;        // bf16 D = 768 forward: 1 = half-wave-per-row kernel with 16-byte accesses, 0 = the generic kernel (A/B, tests)
DVLP_DEV_API int dvlp_dev_layernorm_wide(int on) { g_ln_wide = on; return DVLP_OK; }
extern "C" int dvlp_layernorm_fwd(int dtype, int64_t M, int64_t D, const void* x, const float* gamma, const float* beta, float eps,
                                  void* y, void* y_relu, float* mean, float* rstd, void* stream) {
    dvlp_clear_status();
    if (D % 256 || D > 256 * LN_MAXC || M <= 0) return DVLP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)cdiv(M, 4)), block(256);
    if (dtype == DVLP_F32) hipLaunchKernelGGL(ln_fwd_kernel<float>, grid, block, 0, st, M, (int)(D / 256), (const float*)x, gamma, beta, eps, (float*)y, (float*)y_relu, mean, rstd);
    else if (dtype == DVLP_BF16 && D == 768 && g_ln_wide && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)y_relu) & 15) == 0)
        hipLaunchKernelGGL(ln_fwd768_bf16_kernel, dim3((unsigned)cdiv(M, 8)), block, 0, st, M, (const bf16*)x, gamma, beta, eps, (bf16*)y, (bf16*)y_relu, mean, rstd);
    else if (dtype == DVLP_BF16) hipLaunchKernelGGL(ln_fwd_kernel<bf16>, grid, block, 0, st, M, (int)(D / 256), (const bf16*)x, gamma, beta, eps, (bf16*)y, (bf16*)y_relu, mean, rstd);
    else return DVLP_ERR_DTYPE;
    return dvlp_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// Deferred second-stage reductions.  Every bias / LayerNorm-parameter gradient is a two-stage column sum whose second
// stage is a ~10 us launch over a few KB; a training step has ~125 of them, each also paying a ~2 us kernel boundary.
// With dvlp_reduce_defer() the first stage writes its partials into a caller-owned arena instead and queues the second
// stage; dvlp_reduce_flush() runs ALL queued second stages as one launch (before the optimizer / the tail gradient
// bucket needs them).  Only calls that pass accumulate | 2 ("result not read before the flush") are queued.
// ------------------------------------------------------------------------------------------------------------------
#include <mutex>
#include <vector>
struct RItem { const float* in; float* out; int64_t P, C, stride; int accumulate; int blk0; };
// The items travel to the kernel BY VALUE, a batch per launch (kernel arguments are part of a captured graph node): a device-side table
// would be shared by every captured graph and every eager step -- a step of another shape rewrites it under the graphs that point at
// it -- and uploading it needs a copy and a stream synchronisation, neither of which may happen during a capture.
struct RItemC { const float* in; float* out; int P, C, stride, blk0acc; };      // 32 bytes; blk0acc = 2 * (first block within the batch) + accumulate
constexpr int RB_MAX = 96;
struct RBatch { int n, pad; RItemC it[RB_MAX]; };                               // 3 080 bytes of kernel arguments
struct RDefer {
    std::mutex mu;
    float* ws = nullptr; int64_t ws_floats = 0, used = 0;
    int64_t cap_items = 0;
    std::vector<RItem> cur;
    int nblk = 0;
};
static RDefer g_rd;

extern "C" int dvlp_reduce_defer(void* workspace, int64_t workspace_bytes, void* table, int64_t table_bytes) {
    std::lock_guard<std::mutex> lk(g_rd.mu);
    g_rd.ws = (float*)workspace; g_rd.ws_floats = workspace ? workspace_bytes / 4 : 0; g_rd.used = 0;
    (void)table;                                         // (a device-side item table is no longer used: see RBatch)
    g_rd.cap_items = workspace ? (table_bytes > 0 ? table_bytes / 48 : 4096) : 0;      // bound on the queue length, as the caller sized it
    g_rd.cur.clear(); g_rd.nblk = 0;
    return DVLP_OK;
}
// reserve `floats` of partial space for a deferrable call; nullptr: not enabled / full -> the caller reduces immediately
static float* rd_reserve(int64_t floats, int64_t nitems) {
    if (!g_rd.ws || g_rd.used + floats > g_rd.ws_floats || (int64_t)g_rd.cur.size() + nitems > g_rd.cap_items) return nullptr;
    float* p = g_rd.ws + g_rd.used;
    g_rd.used += (floats + 63) / 64 * 64;
    return p;
}
static void rd_push(const float* in, float* out, int64_t P, int64_t C, int64_t stride, int accumulate) {
    g_rd.cur.push_back(RItem{in, out, P, C, stride, accumulate, g_rd.nblk});
    g_rd.nblk += (int)cdiv(C, 64);
}

// used by the GEMM's fused column sums (gemm.hip): reserve partial space / queue a reduction under the queue's lock
float* dvlp_rd_reserve_push(int64_t P, int64_t C, float* out) {
    std::lock_guard<std::mutex> lk(g_rd.mu);
    float* p = rd_reserve(P * C, 1);
    if (p) rd_push(p, out, P, C, C, 0);
    return p;
}

__global__ __launch_bounds__(256) void reduce_batched_kernel(const RBatch b) {
    __shared__ float red[4][64];
    // which item owns this block: binary search over the items' first-block prefix (<= 7 steps, wave-uniform, scalar loads of the arguments)
    int lo = 0, hi = b.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((b.it[mid].blk0acc >> 1) <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const float* __restrict__ in = b.it[lo].in;
    float* __restrict__ out = b.it[lo].out;
    const int P = b.it[lo].P, C = b.it[lo].C, blk0 = b.it[lo].blk0acc >> 1, accumulate = b.it[lo].blk0acc & 1;
    const int64_t stride = b.it[lo].stride;
    // 64 columns (a 256-byte piece of each partial row) x 4 row slices per workgroup
    const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t c = (int64_t)((int)blockIdx.x - blk0) * 64 + cl;
    float s = 0.f;
    if (c < C) {
        // eight independent partial rows in flight per thread (a plain loop was one dependent load after another: 144 us for the step's
        // ~130 queued reductions, most of them 1024 LayerNorm partial rows deep)
        float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int64_t p = q;
        for (; p + 28 < P; p += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s8[u] += in[(p + 4 * u) * stride + c];
        }
        for (; p < P; p += 4) s8[0] += in[p * stride + c];
        s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    }
    red[q][cl] = s;
    __syncthreads();
    if (q == 0 && c < C) {
        const float t = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
        out[c] = accumulate ? out[c] + t : t;
    }
}

extern "C" int dvlp_reduce_flush(void* stream) {
    std::lock_guard<std::mutex> lk(g_rd.mu);
    if (g_rd.cur.empty()) { g_rd.used = 0; return DVLP_OK; }
    dvlp_clear_status();
    hipStream_t st = (hipStream_t)stream;
    // batches of up to RB_MAX items, each one launch with its items in the kernel arguments (a training step queues ~130: two launches)
    for (size_t i0 = 0; i0 < g_rd.cur.size(); i0 += RB_MAX) {
        RBatch b{};
        const size_t n = g_rd.cur.size() - i0 < (size_t)RB_MAX ? g_rd.cur.size() - i0 : (size_t)RB_MAX;
        const int base = g_rd.cur[i0].blk0;
        int nblk = 0;
        b.n = (int)n;
        for (size_t k = 0; k < n; ++k) {
            const RItem& r = g_rd.cur[i0 + k];
            if (r.P > INT32_MAX || r.C > INT32_MAX || r.stride > INT32_MAX) return DVLP_ERR_SHAPE;
            b.it[k] = RItemC{r.in, r.out, (int)r.P, (int)r.C, (int)r.stride, 2 * (r.blk0 - base) + (r.accumulate ? 1 : 0)};
            nblk = r.blk0 - base + (int)cdiv(r.C, 64);
        }
        hipLaunchKernelGGL(reduce_batched_kernel, dim3((unsigned)nblk), dim3(256), 0, st, b);
    }
    g_rd.cur.clear(); g_rd.used = 0; g_rd.nblk = 0;
    return dvlp_launch_status();
}

// workspace: fp32 [(dvlp_layernorm_bwd_blocks(M) + 1) * 2 * D].  dgamma/dbeta are overwritten (accumulate=0) or added to.
// (1024 workgroups = 4 per CU; 768 / 512 -- fewer partial rows for the deferred reduction -- measured equal / +0.1 ms per step)
extern "C" int64_t dvlp_layernorm_bwd_blocks(int64_t M) { const int64_t b = cdiv(M, 4); return b < 1024 ? b : 1024; }

extern "C" int dvlp_colsum(int dtype, int64_t M, int64_t N, const void* x, int64_t ld, int64_t inner, int64_t ostride, int64_t groups,
                           int64_t gstride, float* out, float* workspace, int accumulate, void* stream);

extern "C" int dvlp_layernorm_bwd(int dtype, int64_t M, int64_t D, const void* dy, const void* x, const float* gamma, const float* mean,
                                  const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta, float* workspace,
                                  int accumulate, float* dx_colsum, void* stream) {
    dvlp_clear_status();
    if (D % 256 || D > 256 * LN_MAXC || M <= 0) return DVLP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t nb = dvlp_layernorm_bwd_blocks(M);
    dim3 grid((unsigned)nb), block(256);
    std::unique_lock<std::mutex> lk(g_rd.mu);
    // dx_colsum (the bias gradient of the Linear that dx feeds) rides along only on the deferred path: a third partial plane
    const int cs = (dx_colsum && (accumulate & 2)) ? 1 : 0;
    float* dws = (accumulate & 2) ? rd_reserve(nb * (2 + cs) * D, 2 + cs) : nullptr;
    const int csk = dws ? cs : 0;
    if (dws) workspace = dws;
    const size_t ldsb = (size_t)4 * (2 + csk) * D * sizeof(float);
#define LNB_(T_, CS_, NC_) hipLaunchKernelGGL((ln_bwd_kernel<T_, CS_, NC_>), grid, block, ldsb, st, M, NC_, (const T_*)dy, (const T_*)x, gamma, mean, rstd, \
                                              (const T_*)dres, (T_*)dx, workspace)
#define LNB(T_, CS_) do { switch ((int)(D / 256)) { case 1: LNB_(T_, CS_, 1); break; case 2: LNB_(T_, CS_, 2); break; case 3: LNB_(T_, CS_, 3); break; \
                                                    default: LNB_(T_, CS_, 4); break; } } while (0)
    if (dtype == DVLP_F32) { if (csk) LNB(float, 1); else LNB(float, 0); }
    else if (dtype == DVLP_BF16) { if (csk) LNB(bf16, 1); else LNB(bf16, 0); }
    else return DVLP_ERR_DTYPE;
#undef LNB
#undef LNB_
    // partial layout [nb][2 (+1)][D]: reduce the planes separately (stride = planes * D between blocks)
    if (dws) {
        const int64_t pl = (2 + csk) * D;
        if (dbeta == dgamma + D) rd_push(dws, dgamma, nb, 2 * D, pl, accumulate & 1);
        else { rd_push(dws, dgamma, nb, D, pl, accumulate & 1); rd_push(dws + D, dbeta, nb, D, pl, accumulate & 1); }
        if (csk) rd_push(dws + 2 * D, dx_colsum, nb, D, pl, 0);
        return dvlp_launch_status();
    }
    lk.unlock();
    accumulate &= 1;
    if (dbeta == dgamma + D) {   // contiguous destination: one launch
        hipLaunchKernelGGL(reduce_groups_kernel, dim3((unsigned)cdiv(2 * D, 32), 1), dim3(256), 0, st, nb, 2 * D, workspace, dgamma, accumulate);
    } else {
        hipLaunchKernelGGL(reduce_groups_kernel, dim3((unsigned)cdiv(2 * D, 32), 1), dim3(256), 0, st, nb, 2 * D, workspace, workspace + nb * 2 * D, 0);
        hipLaunchKernelGGL(reduce_groups_kernel, dim3((unsigned)cdiv(D, 32), 1), dim3(256), 0, st, (int64_t)1, D, workspace + nb * 2 * D, dgamma, accumulate);
        hipLaunchKernelGGL(reduce_groups_kernel, dim3((unsigned)cdiv(D, 32), 1), dim3(256), 0, st, (int64_t)1, D, workspace + nb * 2 * D + D, dbeta, accumulate);
    }
    if (int rc = dvlp_launch_status()) return rc;
    // not deferred: the column sums of dx are a plain second pass (the LayerNorm workspace is free again in stream order)
    if (dx_colsum) return dvlp_colsum(dtype, M, D, dx, D, M, 0, 1, 0, dx_colsum, workspace, 0, stream);
    return DVLP_OK;
}

// out[g][n] (+)= sum_m x_g[m][n]   (bias / embedding-table gradients).  Row m of group g is at
// x + g*gstride + (m / inner)*ostride + (m % inner)*ld  (plain [M, ld] matrix: inner = M, groups = 1).
// workspace: fp32 [groups * dvlp_colsum_chunks(M) * N] (+ optional counters, see dvlp_colsum_counters)
// Zero-initialised device counters (one per (group, 256-column block)) enabling the fused single-launch reduction of
// dvlp_colsum; caller-owned, must stay zero between calls (the kernel re-arms them).  NULL: two-launch path.
static unsigned* g_colsum_counters = nullptr;
static int64_t g_colsum_ncounters = 0;
extern "C" int dvlp_colsum_counters(void* ptr, int64_t count) { g_colsum_counters = (unsigned*)ptr; g_colsum_ncounters = ptr ? count : 0; return DVLP_OK; }

extern "C" int64_t dvlp_colsum_chunks(int64_t M) { const int64_t c = cdiv(M, 64); return c < 192 ? c : 192; }

extern "C" int dvlp_colsum(int dtype, int64_t M, int64_t N, const void* x, int64_t ld, int64_t inner, int64_t ostride, int64_t groups,
                           int64_t gstride, float* out, float* workspace, int accumulate, void* stream) {
    dvlp_clear_status();
    if (M <= 0 || N <= 0 || groups <= 0 || inner <= 0) return DVLP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t P = dvlp_colsum_chunks(M), rows_per = cdiv(M, P);
    dim3 grid((unsigned)cdiv(N, 256), (unsigned)P, (unsigned)groups), block(256);
    std::unique_lock<std::mutex> lk(g_rd.mu);
    float* dws = (accumulate & 2) ? rd_reserve(groups * P * N, groups) : nullptr;
    if (dws) workspace = dws;
    else lk.unlock();
    const int acc1 = accumulate & 1;
    const int64_t al = dtype == DVLP_F32 ? 4 : 8;      // elements per 16 bytes
    const int vec = (ld % al == 0) && (ostride % al == 0) && (gstride % al == 0) && ((uintptr_t)x % 16 == 0);
    unsigned* cnt = !dws && g_colsum_counters && groups * cdiv(N, 256) <= g_colsum_ncounters ? g_colsum_counters : nullptr;
    if (dtype == DVLP_F32) hipLaunchKernelGGL(colsum_kernel<float>, grid, block, 0, st, M, N, (const float*)x, ld, inner, ostride, gstride, rows_per, workspace, vec, cnt, out, acc1);
    else if (dtype == DVLP_BF16) hipLaunchKernelGGL(colsum_kernel<bf16>, grid, block, 0, st, M, N, (const bf16*)x, ld, inner, ostride, gstride, rows_per, workspace, vec, cnt, out, acc1);
    else return DVLP_ERR_DTYPE;
    if (dws) {
        for (int64_t g = 0; g < groups; ++g) rd_push(dws + g * P * N, out + g * N, P, N, N, acc1);
        return dvlp_launch_status();
    }
    if (!cnt) hipLaunchKernelGGL(reduce_groups_kernel, dim3((unsigned)cdiv(N, 32), (unsigned)groups), dim3(256), 0, st, P, N, workspace, out, acc1);
    return dvlp_launch_status();
}
