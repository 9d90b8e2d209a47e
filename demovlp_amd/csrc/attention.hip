// Attention for both towers (K4 and the attention part of K8), forward and backward.
//
// mode 0 "space"  (model/object_transformer.py:152-196 + :91-97, restated as a block-structured mask):
//     tokens n = 0 (CLS), 1 + f*R + r.  A frame-f query attends {CLS} U {frame f} (R+1 keys); the CLS query attends
//     all N keys.  Additive key mask 0 / -100.  One workgroup per (batch, head, frame) stages that frame's R+1 K/V
//     rows in LDS once; one extra workgroup per (batch, head) handles the CLS query across all keys.
// mode 1 "full"   (DistilBERT self-attention): every query attends all N keys; additive key mask 0 / -inf.
//
// head_dim = 64 = one wavefront: a wave owns one query row; scores live one key per lane (two for > 64 keys), the
// softmax max/sum are wave shuffles, and the P.V product runs with lane = output channel.  Arithmetic is fp32 for
// both storage dtypes.  q is pre-scaled by head_dim^-1/2 exactly as the reference does (q *= scale, :160).
//
// Backward recomputes P from q/k (tiles are tiny) and is deterministic: per-(b,h,frame) partials for the shared CLS
// key go to a workspace and are summed by the CLS workgroup of a second launch.
#include "common.h"

constexpr int HD = 64;         // head dim
constexpr int KMAX = 128;      // max keys per LDS-staged segment
constexpr int KP = HD + 1;     // padded LDS row (floats)
constexpr int QC_MAX = 48;     // queries per chunk in the backward segment kernel

struct AttnArgs {
    const void *q, *k, *v;     // [B, N, ld] with head h at columns [h*64, h*64+64)
    int64_t ld;
    const float* addmask;      // [B, N] additive key mask
    void* out;                 // fwd: [B, N, ldo]
    const void* dout;          // bwd
    int64_t ldo;
    void *dq, *dk, *dv;        // bwd: same layout as q/k/v (ld = ldd)
    int64_t ldd;
    float* ws;                 // bwd space mode: [B, H, F, 2, 64] partial dK/dV of the CLS key
    int B, N, H, F, R, mode;
    float scale;
};

__device__ __forceinline__ float lane_bcast(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }

template <typename T>
__device__ __forceinline__ void load_seg_rows(float* dst /*[n][KP]*/, const T* __restrict__ src, int64_t ld, int64_t brow0, int h, int nk,
                                              int incl0, int q0, float mul) {
    for (int idx = threadIdx.x; idx < nk * HD; idx += blockDim.x) {
        const int j = idx >> 6, d = idx & 63;
        const int row = incl0 ? (j == 0 ? 0 : q0 + j - 1) : q0 + j;
        dst[j * KP + d] = to_f(src[(brow0 + row) * ld + h * HD + d]) * mul;
    }
}

// scores of one query (pre-scaled, one channel per lane) against the staged keys: key j on lane j (and j+64)
__device__ __forceinline__ void seg_scores(float qv, const float* Ks, const float* ms, int nk, int lane, float& s0, float& s1) {
    s0 = 0.f; s1 = 0.f;
    const int j0 = lane < nk ? lane : 0, j1 = lane + 64 < nk ? lane + 64 : 0;
#pragma unroll 8
    for (int d = 0; d < HD; ++d) {
        const float qd = lane_bcast(qv, d);
        s0 += qd * Ks[j0 * KP + d];
        s1 += qd * Ks[j1 * KP + d];
    }
    s0 = lane < nk ? s0 + ms[lane] : -INFINITY;
    s1 = lane + 64 < nk ? s1 + ms[lane + 64] : -INFINITY;
}

// ------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int seg = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int nseg = a.mode == 0 ? a.F : 1;
    const int64_t brow0 = (int64_t)b * a.N;
    const T* q = (const T*)a.q; const T* k = (const T*)a.k; const T* v = (const T*)a.v; T* out = (T*)a.out;
    if (seg < nseg) {
        const int incl0 = a.mode == 0 ? 1 : 0, R = a.R, q0 = a.mode == 0 ? 1 + seg * R : 0, nk = R + incl0;
        float* Ks = sm; float* Vs = Ks + nk * KP; float* ms = Vs + nk * KP;
        load_seg_rows<T>(Ks, k, a.ld, brow0, h, nk, incl0, q0, 1.f);
        load_seg_rows<T>(Vs, v, a.ld, brow0, h, nk, incl0, q0, 1.f);
        for (int j = threadIdx.x; j < nk; j += blockDim.x) ms[j] = a.addmask[brow0 + (incl0 ? (j == 0 ? 0 : q0 + j - 1) : q0 + j)];
        __syncthreads();
        for (int i = wid; i < R; i += 4) {
            const int64_t qrow = brow0 + q0 + i;
            const float qv = to_f(q[qrow * a.ld + h * HD + lane]) * a.scale;
            float s0, s1;
            seg_scores(qv, Ks, ms, nk, lane, s0, s1);
            const float m = wave_max(fmaxf(s0, s1));
            const float e0 = lane < nk ? expf(s0 - m) : 0.f, e1 = lane + 64 < nk ? expf(s1 - m) : 0.f;
            const float inv = 1.f / wave_sum(e0 + e1);
            float o = 0.f;
            for (int j = 0; j < nk; ++j) {
                const float pj = j < 64 ? lane_bcast(e0, j) : lane_bcast(e1, j - 64);
                o += pj * Vs[j * KP + lane];
            }
            out[qrow * a.ldo + h * HD + lane] = from_f<T>(o * inv);
        }
    } else {
        // CLS query (space mode): all N keys, key chunks of 64 dealt round-robin to the 4 waves, online softmax
        float* red = sm;   // [4][66]: per-wave (m, l, o[64])
        const float qv = to_f(q[brow0 * a.ld + h * HD + lane]) * a.scale;
        float m = -INFINITY, l = 0.f, o = 0.f;
        for (int c0 = wid * 64; c0 < a.N; c0 += 256) {
            const int key = c0 + lane;
            float s = -INFINITY;
            if (key < a.N) {
                const T* kr = k + (brow0 + key) * a.ld + h * HD;
                s = 0.f;
#pragma unroll 8
                for (int d = 0; d < HD; ++d) s += lane_bcast(qv, d) * to_f(kr[d]);
                s += a.addmask[brow0 + key];
            }
            const float mn = fmaxf(m, wave_max(s));
            const float alpha = expf(m - mn);
            const float p = key < a.N ? expf(s - mn) : 0.f;
            l = l * alpha + wave_sum(p);
            o *= alpha;
            const int cnt = a.N - c0 < 64 ? a.N - c0 : 64;
            for (int j = 0; j < cnt; ++j) o += lane_bcast(p, j) * to_f(v[(brow0 + c0 + j) * a.ld + h * HD + lane]);
            m = mn;
        }
        red[wid * 66 + 2 + lane] = o;
        if (lane == 0) { red[wid * 66] = m; red[wid * 66 + 1] = l; }
        __syncthreads();
        if (wid == 0) {
            float M = -INFINITY;
            for (int w = 0; w < 4; ++w) M = fmaxf(M, red[w * 66]);
            float L = 0.f, O = 0.f;
            for (int w = 0; w < 4; ++w) {
                const float mw = red[w * 66];
                const float f = mw == -INFINITY ? 0.f : expf(mw - M);
                L += red[w * 66 + 1] * f;
                O += red[w * 66 + 2 + lane] * f;
            }
            out[brow0 * a.ldo + h * HD + lane] = from_f<T>(O / L);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// backward, launch 1: per-segment workgroups (frame queries / all queries in full mode)
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_seg_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int seg = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int64_t brow0 = (int64_t)b * a.N;
    const T* q = (const T*)a.q; const T* k = (const T*)a.k; const T* v = (const T*)a.v; const T* dout = (const T*)a.dout;
    T* dq = (T*)a.dq; T* dk = (T*)a.dk; T* dv = (T*)a.dv;
    const int incl0 = a.mode == 0 ? 1 : 0, R = a.R, q0 = a.mode == 0 ? 1 + seg * R : 0, nk = R + incl0;
    const int QC = R < QC_MAX ? R : QC_MAX, nkp = nk + 1;
    float* Ks = sm; float* Vs = Ks + nk * KP; float* ms = Vs + nk * KP;
    float* Qs = ms + KMAX; float* dOs = Qs + QC * KP; float* Ps = dOs + QC * KP; float* dSs = Ps + QC * nkp;
    load_seg_rows<T>(Ks, k, a.ld, brow0, h, nk, incl0, q0, 1.f);
    load_seg_rows<T>(Vs, v, a.ld, brow0, h, nk, incl0, q0, 1.f);
    for (int j = threadIdx.x; j < nk; j += blockDim.x) ms[j] = a.addmask[brow0 + (incl0 ? (j == 0 ? 0 : q0 + j - 1) : q0 + j)];
    float accK[KMAX * HD / 256], accV[KMAX * HD / 256];
#pragma unroll
    for (int t = 0; t < KMAX * HD / 256; ++t) { accK[t] = 0.f; accV[t] = 0.f; }

    for (int c0 = 0; c0 < R; c0 += QC) {
        const int nq = R - c0 < QC ? R - c0 : QC;
        __syncthreads();   // previous chunk's phase B done with Qs/dOs/Ps/dSs (and K/V staged on first pass)
        load_seg_rows<T>(Qs, q, a.ld, brow0, h, nq, 0, q0 + c0, a.scale);
        load_seg_rows<T>(dOs, dout, a.ldo, brow0, h, nq, 0, q0 + c0, 1.f);
        __syncthreads();
        // phase A: one wave per query -> P, dS rows and dq
        for (int i = wid; i < nq; i += 4) {
            const float qv = Qs[i * KP + lane], dov = dOs[i * KP + lane];
            float s0, s1;
            seg_scores(qv, Ks, ms, nk, lane, s0, s1);
            const float m = wave_max(fmaxf(s0, s1));
            const float e0 = lane < nk ? expf(s0 - m) : 0.f, e1 = lane + 64 < nk ? expf(s1 - m) : 0.f;
            const float inv = 1.f / wave_sum(e0 + e1);
            const float p0 = e0 * inv, p1 = e1 * inv;
            float dp0 = 0.f, dp1 = 0.f;
            const int j0 = lane < nk ? lane : 0, j1 = lane + 64 < nk ? lane + 64 : 0;
#pragma unroll 8
            for (int d = 0; d < HD; ++d) {
                const float g = lane_bcast(dov, d);
                dp0 += g * Vs[j0 * KP + d];
                dp1 += g * Vs[j1 * KP + d];
            }
            const float Dsum = wave_sum(p0 * dp0 + p1 * dp1);
            const float ds0 = p0 * (dp0 - Dsum), ds1 = p1 * (dp1 - Dsum);
            if (lane < nk) { Ps[i * nkp + lane] = p0; dSs[i * nkp + lane] = ds0; }
            if (lane + 64 < nk) { Ps[i * nkp + lane + 64] = p1; dSs[i * nkp + lane + 64] = ds1; }
            float g = 0.f;
            for (int j = 0; j < nk; ++j) {
                const float dsj = j < 64 ? lane_bcast(ds0, j) : lane_bcast(ds1, j - 64);
                g += dsj * Ks[j * KP + lane];
            }
            dq[(brow0 + q0 + c0 + i) * a.ldd + h * HD + lane] = from_f<T>(g * a.scale);
        }
        __syncthreads();
        // phase B: thread-per-(key, channel) accumulation of dK, dV over this chunk's queries
#pragma unroll
        for (int t = 0; t < KMAX * HD / 256; ++t) {
            const int idx = threadIdx.x + 256 * t;
            const int j = idx >> 6, d = idx & 63;
            if (j < nk) {
                float ak = 0.f, av = 0.f;
                for (int i = 0; i < nq; ++i) {
                    ak += dSs[i * nkp + j] * Qs[i * KP + d];
                    av += Ps[i * nkp + j] * dOs[i * KP + d];
                }
                accK[t] += ak; accV[t] += av;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < KMAX * HD / 256; ++t) {
        const int idx = threadIdx.x + 256 * t;
        const int j = idx >> 6, d = idx & 63;
        if (j < nk) {
            if (incl0 && j == 0) {
                float* w = a.ws + ((((int64_t)b * a.H + h) * a.F + seg) * 2) * HD;
                w[d] = accK[t]; w[HD + d] = accV[t];
            } else {
                const int64_t row = brow0 + (incl0 ? q0 + j - 1 : q0 + j);
                dk[row * a.ldd + h * HD + d] = from_f<T>(accK[t]);
                dv[row * a.ldd + h * HD + d] = from_f<T>(accV[t]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// backward, launch 2 (space mode): the CLS query's contributions + the CLS key's totals.  One workgroup per (b, h).
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_cls_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int h = blockIdx.x, b = blockIdx.y;
    const int64_t brow0 = (int64_t)b * a.N;
    const int N = a.N;
    const T* q = (const T*)a.q; const T* k = (const T*)a.k; const T* v = (const T*)a.v; const T* dout = (const T*)a.dout;
    T* dq = (T*)a.dq; T* dk = (T*)a.dk; T* dv = (T*)a.dv;
    float* S = sm; float* DP = S + N; float* red = DP + N;   // red: [4][64] + scalars
    const float qv = to_f(q[brow0 * a.ld + h * HD + lane]) * a.scale;
    const float dov = to_f(dout[brow0 * a.ldo + h * HD + lane]);
    for (int j = wid; j < N; j += 4) {
        const float kv = to_f(k[(brow0 + j) * a.ld + h * HD + lane]), vv = to_f(v[(brow0 + j) * a.ld + h * HD + lane]);
        const float s = wave_sum(qv * kv), dp = wave_sum(dov * vv);
        if (lane == 0) { S[j] = s + a.addmask[brow0 + j]; DP[j] = dp; }
    }
    __syncthreads();
    // every wave redundantly reduces the (small) score vector
    float m = -INFINITY;
    for (int j = lane; j < N; j += 64) m = fmaxf(m, S[j]);
    m = wave_max(m);
    float l = 0.f, pd = 0.f;
    for (int j = lane; j < N; j += 64) { const float e = expf(S[j] - m); l += e; pd += e * DP[j]; }
    l = wave_sum(l); pd = wave_sum(pd);
    const float inv = 1.f / l, Dsum = pd * inv;
    float dqa = 0.f;
    for (int j = wid; j < N; j += 4) {
        const float p = expf(S[j] - m) * inv, ds = p * (DP[j] - Dsum);
        const int64_t off = (brow0 + j) * a.ld + h * HD + lane, offd = (brow0 + j) * a.ldd + h * HD + lane;
        dqa += ds * to_f(k[off]);
        float gk = ds * qv, gv = p * dov;
        if (j == 0) {
            const float* w = a.ws + (((int64_t)b * a.H + h) * a.F) * 2 * HD;
            for (int f = 0; f < a.F; ++f) { gk += w[f * 2 * HD + lane]; gv += w[f * 2 * HD + HD + lane]; }
        } else {
            gk += to_f(dk[offd]); gv += to_f(dv[offd]);
        }
        dk[offd] = from_f<T>(gk); dv[offd] = from_f<T>(gv);
    }
    red[wid * 64 + lane] = dqa;
    __syncthreads();
    if (wid == 0) dq[brow0 * a.ldd + h * HD + lane] = from_f<T>((red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane]) * a.scale);
}

static int attn_check(const AttnArgs& a) {
    if (a.B <= 0 || a.H <= 0 || a.N <= 0) return DVLP_ERR_SHAPE;
    if (a.mode == 0) { if (a.N != 1 + a.F * a.R || a.R + 1 > KMAX) return DVLP_ERR_SHAPE; }
    else if (a.mode == 1) { if (a.R != a.N || a.N > KMAX) return DVLP_ERR_SHAPE; }
    else return DVLP_ERR_UNSUPPORTED;
    return DVLP_OK;
}

extern "C" int dvlp_attention_fwd(int dtype, int mode, int64_t B, int64_t N, int64_t H, int64_t F, int64_t R, const void* q, const void* k,
                                  const void* v, int64_t ld, const float* addmask, void* out, int64_t ldo, float scale, void* stream) {
    dvlp_clear_status();
    AttnArgs a{};
    a.q = q; a.k = k; a.v = v; a.ld = ld; a.addmask = addmask; a.out = out; a.ldo = ldo;
    a.B = (int)B; a.N = (int)N; a.H = (int)H; a.F = (int)F; a.R = (int)R; a.mode = mode; a.scale = scale;
    if (int rc = attn_check(a)) return rc;
    const int nseg = mode == 0 ? (int)F : 1, nk = (int)R + (mode == 0 ? 1 : 0);
    size_t lds = (size_t)(2 * nk * KP + KMAX) * sizeof(float);
    if (lds < 4 * 66 * sizeof(float)) lds = 4 * 66 * sizeof(float);
    dim3 grid((unsigned)(nseg + (mode == 0 ? 1 : 0)), (unsigned)H, (unsigned)B), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DVLP_F32) {
        static bool once = false; if (!once) { once = true; (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
        hipLaunchKernelGGL(attn_fwd_kernel<float>, grid, block, lds, st, a);
    } else if (dtype == DVLP_BF16) {
        static bool once = false; if (!once) { once = true; (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
        hipLaunchKernelGGL(attn_fwd_kernel<bf16>, grid, block, lds, st, a);
    } else return DVLP_ERR_DTYPE;
    return dvlp_launch_status();
}

// workspace (space mode only): fp32 [B*H*F*2*64]
extern "C" int dvlp_attention_bwd(int dtype, int mode, int64_t B, int64_t N, int64_t H, int64_t F, int64_t R, const void* q, const void* k,
                                  const void* v, int64_t ld, const float* addmask, const void* dout, int64_t ldo, void* dq, void* dk,
                                  void* dv, int64_t ldd, float* workspace, float scale, void* stream) {
    dvlp_clear_status();
    AttnArgs a{};
    a.q = q; a.k = k; a.v = v; a.ld = ld; a.addmask = addmask; a.dout = dout; a.ldo = ldo; a.dq = dq; a.dk = dk; a.dv = dv; a.ldd = ldd;
    a.ws = workspace;
    a.B = (int)B; a.N = (int)N; a.H = (int)H; a.F = (int)F; a.R = (int)R; a.mode = mode; a.scale = scale;
    if (int rc = attn_check(a)) return rc;
    if (mode == 0 && !workspace) return DVLP_ERR_SHAPE;
    const int nseg = mode == 0 ? (int)F : 1, nk = (int)R + (mode == 0 ? 1 : 0);
    const int QC = (int)R < QC_MAX ? (int)R : QC_MAX;
    const size_t lds = (size_t)(2 * nk * KP + KMAX + 2 * QC * KP + 2 * QC * (nk + 1)) * sizeof(float);
    if (lds > 160 * 1024) return DVLP_ERR_SHAPE;
    dim3 grid((unsigned)nseg, (unsigned)H, (unsigned)B), block(256);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds2 = (size_t)(2 * N + 4 * 64 + 8) * sizeof(float);
    dim3 grid2((unsigned)H, (unsigned)B);
    if (dtype == DVLP_F32) {
        static bool once = false; if (!once) { once = true; (void)hipFuncSetAttribute((const void*)attn_bwd_seg_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
        hipLaunchKernelGGL(attn_bwd_seg_kernel<float>, grid, block, lds, st, a);
        if (mode == 0) hipLaunchKernelGGL(attn_bwd_cls_kernel<float>, grid2, block, lds2, st, a);
    } else if (dtype == DVLP_BF16) {
        static bool once = false; if (!once) { once = true; (void)hipFuncSetAttribute((const void*)attn_bwd_seg_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
        hipLaunchKernelGGL(attn_bwd_seg_kernel<bf16>, grid, block, lds, st, a);
        if (mode == 0) hipLaunchKernelGGL(attn_bwd_cls_kernel<bf16>, grid2, block, lds2, st, a);
    } else return DVLP_ERR_DTYPE;
    return dvlp_launch_status();
}
