// Loss heads on [B,B] similarity matrices: sim_matrix + NormSoftmaxLoss (K9/K10; model/model.py:582-590,
// model/loss.py:126-138) and RWALoss's softmax-KL tail (K12; model/loss.py:105-116), forward AND analytic backward in
// one launch.  B is the per-rank batch (64) or the gathered batch (<= 1024): latency-bound, one 1024-thread workgroup,
// phases separated by workgroup barriers, all math fp32.
#include "common.h"

constexpr int LD_ = 256;   // embedding dim

template <typename T> __device__ __forceinline__ void l4(const T* p, float (&o)[4]);
template <> __device__ __forceinline__ void l4<float>(const float* p, float (&o)[4]) { float4 v = *(const float4*)p; o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
template <> __device__ __forceinline__ void l4<bf16>(const bf16* p, float (&o)[4]) { bf16x4 v = *(const bf16x4*)p; o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2]; o[3] = (float)v[3]; }
template <typename T> __device__ __forceinline__ void s4(T* p, const float (&o)[4]);
template <> __device__ __forceinline__ void s4<float>(float* p, const float (&o)[4]) { *(float4*)p = make_float4(o[0], o[1], o[2], o[3]); }
template <> __device__ __forceinline__ void s4<bf16>(bf16* p, const float (&o)[4]) { bf16x4 v; v[0] = (bf16)o[0]; v[1] = (bf16)o[1]; v[2] = (bf16)o[2]; v[3] = (bf16)o[3]; *(bf16x4*)p = v; }

struct LossArgs {
    const void *gt, *go;     // global text / object embeddings [Bt][256], [Bo][256]
    const float* xs;         // local scores [Bo][Bt] (row = video) or null
    float *sim, *dsim;       // [Bt][Bo] out / scratch
    void *dgt, *dgo;         // grads (compute dtype)
    float* dxs;              // [Bo][Bt] grad of the total loss wrt xs
    float* losses;           // [3] total, global, local
    int B;
    float temperature, lam;
    int use_global, use_local, stages;   // stages: 1 = sim forward, 2 = losses + dsim/dxs, 4 = embedding grads from dsim
    int staged;              // both embedding matrices fit in LDS (B <= 64): rows are read from there
    int mf;                  // bf16, B in {32, 64}: the B x B x 256 products run on the matrix cores (see loss_kernel)
};

__device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float s = 0.f;
    for (int w = 0; w < nw; ++w) s += red[w];
    return s;
}

template <typename T>
__global__ __launch_bounds__(1024) void loss_kernel(LossArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int B = a.B, lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    float* nt = sm; float* no = nt + B; float* lse_r = no + B; float* lse_c = lse_r + B; float* red = lse_c + B;
    float* rnt = red + 32; float* rno = rnt + B;          // reciprocal norms (the loops below would otherwise divide per entry)
    float* E = rno + B;                       // staged: [2B][256] fp32 copies of gt (rows 0..B-1) and go (rows B..2B-1)
    const T* gt = (const T*)a.gt; const T* go = (const T*)a.go;
    // row r of the stacked (gt | go) matrix, 4 channels per lane: from LDS when staged -- this is a ONE-workgroup kernel, so
    // every row re-read from global memory inside the B x B loops below was a full, unhidden memory round trip
    auto ldrow = [&](int r, float (&v)[4]) {
        if (a.staged) { const float4 t = *(const float4*)&E[r * LD_ + lane * 4]; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
        else l4<T>((r < B ? gt + (int64_t)r * LD_ : go + (int64_t)(r - B) * LD_) + lane * 4, v);
    };
    float gl = 0.f, ll = 0.f;
    if (a.use_global && (a.stages & 5)) {
        // norms, clamped as in sim_matrix: a / max(|a|, 1e-8)
        for (int r = wid; r < 2 * B; r += nw) {
            float v[4];
            l4<T>((r < B ? gt + (int64_t)r * LD_ : go + (int64_t)(r - B) * LD_) + lane * 4, v);
            if (a.staged) *(float4*)&E[r * LD_ + lane * 4] = make_float4(v[0], v[1], v[2], v[3]);
            const float n = sqrtf(wave_sum(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]));
            if (lane == 0) { (r < B ? nt[r] : no[r - B]) = fmaxf(n, 1e-8f); (r < B ? rnt[r] : rno[r - B]) = 1.f / fmaxf(n, 1e-8f); }
        }
        __syncthreads();
    }
    // Matrix-core form of the three products (bf16 embeddings, B a multiple of 32).  As one-wave-per-entry dot products they read
    // two 1-KB rows from LDS per entry -- 8 MB per product through one CU's LDS port, ~60 of this launch's 82 us at B = 64.
    // sim = gt go^T: both operands are plain 16-byte row reads (k = channel).  The two gradient products contract over ROWS (k = o or t):
    // the A operand is dsim (or its transpose) times the other side's reciprocal norm, split into a bf16 head and a bf16 remainder (two
    // MFMAs: ~2^-17 relative, fp32 for this purpose), the B operand gathers eight rows' values of one channel.
    if constexpr (sizeof(T) == 2) {
        if (a.use_global && (a.stages & 1) && a.mf) {
            const int nT = B >> 4;
            for (int tile = wid; tile < nT * nT; tile += nw) {
                const int ti = tile / nT, oi = tile % nT;
                const bf16* ar = (const bf16*)gt + (int64_t)(ti * 16 + (lane & 15)) * LD_ + (lane >> 4) * 8;
                const bf16* br = (const bf16*)go + (int64_t)(oi * 16 + (lane & 15)) * LD_ + (lane >> 4) * 8;
                bf16x8 af[8], bq[8];
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) { af[kk] = *(const bf16x8*)(ar + kk * 32); bq[kk] = *(const bf16x8*)(br + kk * 32); }
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kk], bq[kk], acc, 0, 0, 0);
                const int o = oi * 16 + (lane & 15);                 // lane holds D[text row 4 (lane >> 4) + r][object row lane & 15]
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int t = ti * 16 + 4 * (lane >> 4) + r; a.sim[t * B + o] = acc[r] * (rnt[t] * rno[o]); }
            }
            __syncthreads();
        }
    }
    if (a.use_global && (a.stages & 1) && !a.mf) {
        // sim[t][o]: one wave per entry
        for (int e = wid; e < B * B; e += nw) {
            const int t = e / B, o = e % B;
            float x[4], y[4];
            ldrow(t, x); ldrow(B + o, y);
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) d += x[c] * y[c];
            d = wave_sum(d) * (rnt[t] * rno[o]);
            if (lane == 0) a.sim[e] = d;
        }
        __syncthreads();
    }
    if (a.use_global && (a.stages & 2)) {
        // log-sum-exp of rows and columns of sim / temperature
        for (int r = wid; r < 2 * B; r += nw) {
            const bool row = r < B; const int idx = row ? r : r - B;
            float m = -INFINITY;
            for (int k = lane; k < B; k += 64) m = fmaxf(m, (row ? a.sim[idx * B + k] : a.sim[k * B + idx]) / a.temperature);
            m = wave_max(m);
            float s = 0.f;
            for (int k = lane; k < B; k += 64) s += expf((row ? a.sim[idx * B + k] : a.sim[k * B + idx]) / a.temperature - m);
            s = wave_sum(s);
            if (lane == 0) (row ? lse_r[idx] : lse_c[idx]) = m + logf(s);
        }
        __syncthreads();
        float part = 0.f;
        for (int e = threadIdx.x; e < B * B; e += blockDim.x) {
            const int t = e / B, o = e % B;
            const float z = a.sim[e] / a.temperature;
            const float dl = (expf(z - lse_r[t]) + expf(z - lse_c[o]) - (t == o ? 2.f : 0.f)) / ((float)B * a.temperature);
            a.dsim[e] = dl;
            if (t == o) part -= (z - lse_r[t]) + (z - lse_c[o]);
        }
        gl = block_sum(part, red) / (float)B;
        __syncthreads();
    }
    if (a.use_global && (a.stages & 4)) {
        // d/d normalised rows, then through a / max(|a|, eps)
        float* Ds = E + 2 * B * LD_;                 // staged: dsim [B][B] (the row loop below re-read it from global memory entry by entry)
        if (a.staged || a.mf) {
            for (int e = threadIdx.x; e < B * B; e += blockDim.x) Ds[e] = a.dsim[e];
            __syncthreads();
        }
        if constexpr (sizeof(T) == 2) {
            if (a.mf) {
                // wave = one 16-channel tile of both gradient matrices; the un-normalised sums land in E ([2B][256], free in this mode)
                const int nK = B >> 5, g8 = (lane >> 4) * 8;
                for (int ci = wid; ci < LD_ / 16; ci += nw) {
                    const int c = ci * 16 + (lane & 15);
                    bf16x8 gof[2], gtf[2];
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            if (kk >= nK) break;
                            gof[kk][i] = ((const bf16*)go)[(int64_t)(kk * 32 + g8 + i) * LD_ + c];
                            gtf[kk][i] = ((const bf16*)gt)[(int64_t)(kk * 32 + g8 + i) * LD_ + c];
                        }
                    for (int rt = 0; rt < (B >> 4); ++rt) {
                        const int rr = rt * 16 + (lane & 15);
                        f32x4 acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk) {
                            if (kk >= nK) break;
                            const int k0 = kk * 32 + g8;
                            bf16x8 h1, l1, h2, l2;
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const float w1 = Ds[rr * B + k0 + i] * rno[k0 + i];       // d/d text row rr: over object rows
                                const float w2 = Ds[(k0 + i) * B + rr] * rnt[k0 + i];     // d/d object row rr: over text rows
                                h1[i] = (bf16)w1; l1[i] = (bf16)(w1 - (float)h1[i]);
                                h2[i] = (bf16)w2; l2[i] = (bf16)(w2 - (float)h2[i]);
                            }
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h1, gof[kk], acc1, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l1, gof[kk], acc1, 0, 0, 0);
                            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h2, gtf[kk], acc2, 0, 0, 0);
                            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l2, gtf[kk], acc2, 0, 0, 0);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = rt * 16 + 4 * (lane >> 4) + r;
                            E[i * LD_ + c] = acc1[r]; E[(B + i) * LD_ + c] = acc2[r];
                        }
                    }
                }
                __syncthreads();
            }
        }
        for (int r = wid; r < 2 * B; r += nw) {
            const bool row = r < B; const int idx = row ? r : r - B;
            const float* ron = row ? rno : rnt;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.mf) {
                const float4 t4 = *(const float4*)&E[r * LD_ + lane * 4];
                acc[0] = t4.x; acc[1] = t4.y; acc[2] = t4.z; acc[3] = t4.w;
            } else
            for (int k = 0; k < B; ++k) {
                const float w = (a.staged ? (row ? Ds[idx * B + k] : Ds[k * B + idx]) : (row ? a.dsim[idx * B + k] : a.dsim[k * B + idx])) * ron[k];
                float y[4];
                ldrow(row ? B + k : k, y);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] += w * y[c];
            }
            float x[4];
            ldrow(r, x);
            const float n = row ? nt[idx] : no[idx];
            float o4[4];
            if (n > 1e-8f) {
                const float proj = wave_sum(acc[0] * x[0] + acc[1] * x[1] + acc[2] * x[2] + acc[3] * x[3]) / (n * n);
#pragma unroll
                for (int c = 0; c < 4; ++c) o4[c] = (acc[c] - x[c] * proj) / n;
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) o4[c] = acc[c] / 1e-8f;
            }
            s4<T>((T*)(row ? a.dgt : a.dgo) + (int64_t)idx * LD_ + lane * 4, o4);
        }
    }
    if (a.use_local && (a.stages & 2)) {
        // RWA: p = softmax(lam * xs, dim=1); L_i = sum_j p (log p - log(eye + 1e-6)); dL_i/dz_k = p_k ((log p_k - c_k) - L_i)
        float part = 0.f;
        for (int i = wid; i < B; i += nw) {
            float m = -INFINITY;
            for (int k = lane; k < B; k += 64) m = fmaxf(m, a.xs[i * B + k] * a.lam);
            m = wave_max(m);
            float s = 0.f;
            for (int k = lane; k < B; k += 64) s += expf(a.xs[i * B + k] * a.lam - m);
            const float lse = m + logf(wave_sum(s));
            float Li = 0.f;
            for (int k = lane; k < B; k += 64) {
                const float lp = a.xs[i * B + k] * a.lam - lse;
                Li += expf(lp) * (lp - logf((k == i ? 1.f : 0.f) + 1e-6f));
            }
            Li = wave_sum(Li);
            for (int k = lane; k < B; k += 64) {
                const float lp = a.xs[i * B + k] * a.lam - lse;
                a.dxs[i * B + k] = a.lam / (float)B * expf(lp) * ((lp - logf((k == i ? 1.f : 0.f) + 1e-6f)) - Li);
            }
            if (lane == 0) part += Li;
        }
        ll = block_sum(part, red) / (float)B;
    }
    if (threadIdx.x == 0 && (a.stages & 2)) { a.losses[0] = gl + ll; a.losses[1] = gl; a.losses[2] = ll; }
}

static int g_loss_mfma = 1;
DVLP_DEV_API int dvlp_dev_loss_mfma(int on) { g_loss_mfma = on; return DVLP_OK; }
// sim / dsim: fp32 [B*B] each.  stages (bitmask): 1 = sim_matrix forward (gt, go -> sim); 2 = losses from sim / xs, with
// dsim = d global / d sim and dxs = d local / d xs; 4 = sim_matrix backward (dsim -> dgt, dgo).  7 = everything in one launch.
extern "C" int dvlp_global_local_loss(int dtype, int64_t B, int64_t d, const void* gt, const void* go, const float* xs, float temperature,
                                      float lam, int use_global, int use_local, int stages, float* sim, float* dsim, void* dgt, void* dgo,
                                      float* dxs, float* losses, void* stream) {
    dvlp_clear_status();
    if (d != LD_ || B <= 0 || B > 2048) return DVLP_ERR_SHAPE;
    if (use_local && !xs) return DVLP_ERR_SHAPE;
    const int mf = (dtype == DVLP_BF16 && (B == 32 || B == 64) && g_loss_mfma) ? 1 : 0;
    const int staged = (B <= 64 && !mf) ? 1 : 0;
    LossArgs a{gt, go, xs, sim, dsim, dgt, dgo, dxs, losses, (int)B, temperature, lam, use_global, use_local, stages, staged, mf};
    const size_t lds = (size_t)(6 * B + 32 + ((staged || mf) ? 2 * B * LD_ + B * B : 0)) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    { static bool once = false; if (!once) { once = true;
        (void)hipFuncSetAttribute((const void*)loss_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)loss_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); } }
    if (dtype == DVLP_F32) hipLaunchKernelGGL(loss_kernel<float>, dim3(1), dim3(1024), lds, st, a);
    else if (dtype == DVLP_BF16) hipLaunchKernelGGL(loss_kernel<bf16>, dim3(1), dim3(1024), lds, st, a);
    else return DVLP_ERR_DTYPE;
    return dvlp_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// Rectangular sim_matrix (model/model.py:582-590 takes any [N,d] x [M,d]; the validation path hands it the whole eval
// set, trainer/trainer_dist.py:369): rows are normalised here into fp32, the [N,M] product (and the two backward
// products) run on the exact-fp32 GEMM, and the normalisation's backward is the second kernel.  One wave per row.
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rownorm_fwd_kernel(int64_t M, const T* __restrict__ x, float* __restrict__ xn, float* __restrict__ norm) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;
    float v[4];
    l4<T>(x + r * LD_ + lane * 4, v);
    const float n = sqrtf(wave_sum(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]));
    const float rn = 1.f / fmaxf(n, 1e-8f);
    *(float4*)(xn + r * LD_ + lane * 4) = make_float4(v[0] * rn, v[1] * rn, v[2] * rn, v[3] * rn);
    if (lane == 0) norm[r] = n;
}

template <typename T>
__global__ __launch_bounds__(256) void rownorm_bwd_kernel(int64_t M, const T* __restrict__ x, const float* __restrict__ norm,
                                                          const float* __restrict__ dxn, T* __restrict__ dx) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;
    float v[4], o[4];
    l4<T>(x + r * LD_ + lane * 4, v);
    const float4 g4 = *(const float4*)(dxn + r * LD_ + lane * 4);
    const float g[4] = {g4.x, g4.y, g4.z, g4.w};
    const float n = norm[r];
    if (n > 1e-8f) {
        const float proj = wave_sum(g[0] * v[0] + g[1] * v[1] + g[2] * v[2] + g[3] * v[3]) / (n * n);
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = (g[c] - v[c] * proj) / n;
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = g[c] / 1e-8f;
    }
    s4<T>(dx + r * LD_ + lane * 4, o);
}

extern "C" int dvlp_rownorm_fwd(int dtype, int64_t M, int64_t d, const void* x, float* xn, float* norm, void* stream) {
    dvlp_clear_status();
    if (d != LD_ || M <= 0) return DVLP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)cdiv(M, 4));
    if (dtype == DVLP_F32) hipLaunchKernelGGL(rownorm_fwd_kernel<float>, grid, dim3(256), 0, st, M, (const float*)x, xn, norm);
    else if (dtype == DVLP_BF16) hipLaunchKernelGGL(rownorm_fwd_kernel<bf16>, grid, dim3(256), 0, st, M, (const bf16*)x, xn, norm);
    else return DVLP_ERR_DTYPE;
    return dvlp_launch_status();
}

extern "C" int dvlp_rownorm_bwd(int dtype, int64_t M, int64_t d, const void* x, const float* norm, const float* dxn, void* dx, void* stream) {
    dvlp_clear_status();
    if (d != LD_ || M <= 0) return DVLP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)cdiv(M, 4));
    if (dtype == DVLP_F32) hipLaunchKernelGGL(rownorm_bwd_kernel<float>, grid, dim3(256), 0, st, M, (const float*)x, norm, dxn, (float*)dx);
    else if (dtype == DVLP_BF16) hipLaunchKernelGGL(rownorm_bwd_kernel<bf16>, grid, dim3(256), 0, st, M, (const bf16*)x, norm, dxn, (bf16*)dx);
    else return DVLP_ERR_DTYPE;
    return dvlp_launch_status();
}
