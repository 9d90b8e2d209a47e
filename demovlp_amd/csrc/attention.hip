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
    int seg_begin;             // first segment index served by blockIdx.x = 0 (CLS-only launches start at nseg)
    int abl;                   // TIMING-ONLY ablation bits (tools/attn_bench.py): 0 in production
    // attention-probability dropout (mode 1, HF MultiHeadSelfAttention: weights = dropout(softmax(scores))): keep bytes in both
    // orientations, [b*H + h][q][key] and [b*H + h][key][q], row stride Ns; the product P V uses P keep kscale.  null = off.
    const uint8_t *keep, *keepT;
    float kscale;
    int Ns;
    // space mode, bf16, CLS query folded into the frame waves (round 3): the CLS query rides along as query row R of every frame
    // tile in the FORWARD too -- each wave leaves the CLS row's softmax over its own keys (normalised output [64] + (max, sum)) in
    // `cls_o` / `cls_st`, attn_fwd_cls_combine_kernel merges the F partials flash-style and keeps the global (max, sum) in
    // `cls_stats` [B, H, 4]; the backward reads them back (no statistics pass), takes D = <dO_cls, O_cls> from the forward output
    // `fwd_out`, and leaves per-frame partials of dq_cls in `dq_ws` for the closing launch.
    float *cls_o, *cls_st, *cls_stats, *dq_ws;
    const void* fwd_out;
    int64_t ld_fo;
    // merged backward: partial column sums of dq | dk | dv as stored (the gradient of the packed qkv bias): [B*F + B][3*H*64] fp32, one row
    // per (b, frame) written by that frame's 12 head waves + one row per b for the CLS token's rows; summed by the deferred-reduction queue
    float* csum;
};
// Optional extras of dvlp_attention_fwd_ex / dvlp_attention_bwd_ex (include/demovlp_hip.h: dvlp_attn_ext), handed to the call that consumes
// them.  (Rounds 2-3 armed these through thread-local "next call" setters: an exception between the two calls left a stale device
// pointer armed for an unrelated launch.)
struct dvlp_attn_ext {
    const void* keep; const void* keepT; float keep_scale;      // mode 1: dropout of the attention probabilities (dvlp_dropout_attn_mask)
    float* colsum;                                              // backward, mode 0: column sums of dq | dk | dv wanted here (or NULL)
    int colsum_fused;                                           // out: the backward queued them (else the caller sums the columns)
    int folded;                                                 // out: the forward folded the CLS query and filled cls_stats
};
// four consecutive keep bytes as multipliers
__device__ __forceinline__ void keep4(const uint8_t* p, float scale, float (&m)[4]) {
    const uint32_t kb = *(const uint32_t*)p;
#pragma unroll
    for (int t = 0; t < 4; ++t) m[t] = ((kb >> (8 * t)) & 1u) ? scale : 0.f;
}
static int g_attn_abl = 0;
static int g_attn_lean = 1;        // space-mode bf16, CLS folded: 1 = the round-5 kernels (buffer addressing, swapped output products), 0 = rounds 3-4 (A/B, tests)
DVLP_DEV_API int dvlp_dev_attention_lean(int on) { g_attn_lean = on; return DVLP_OK; }
static int g_attn_merged = 1;      // space-mode bf16 backward: 1 = one-pass form, 0 = the three-launch form (A/B, tests)
DVLP_DEV_API int dvlp_dev_attention_bwd_variant(int merged) { g_attn_merged = merged; return DVLP_OK; }
DVLP_DEV_API int dvlp_dev_attention_ablate(int bits) { g_attn_abl = bits; return DVLP_OK; }

__device__ __forceinline__ float lane_bcast(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }

// 8 consecutive channels of a row as floats (one 16-byte read in bf16, two in fp32)
__device__ __forceinline__ void load8(const float* p, float (&o)[8]) {
    const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16* p, float (&o)[8]) {
    const bf16x8 v = *(const bf16x8*)p;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)v[e];
}
__device__ __forceinline__ void store8(float* p, const float (&o)[8]) {
    *(float4*)p = make_float4(o[0], o[1], o[2], o[3]); *(float4*)(p + 4) = make_float4(o[4], o[5], o[6], o[7]);
}
__device__ __forceinline__ void store8(bf16* p, const float (&o)[8]) {
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16)o[e];
    *(bf16x8*)p = v;
}

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
    const int seg = blockIdx.x + a.seg_begin, h = blockIdx.y, b = blockIdx.z;
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
            float d0 = e0, d1 = e1;                                  // numerators entering P V (dropped when dropout is on)
            if (a.keep && a.mode == 1) {
                const uint8_t* kr = a.keep + (((int64_t)b * a.H + h) * a.Ns + i) * a.Ns;
                d0 = (lane < nk && kr[lane]) ? e0 * a.kscale : 0.f;
                d1 = (lane + 64 < nk && kr[lane + 64 < a.Ns ? lane + 64 : 0]) ? e1 * a.kscale : 0.f;
            }
            float o = 0.f;
            for (int j = 0; j < nk; ++j) {
                const float pj = j < 64 ? lane_bcast(d0, j) : lane_bcast(d1, j - 64);
                o += pj * Vs[j * KP + lane];
            }
            out[qrow * a.ldo + h * HD + lane] = from_f<T>(o * inv);
        }
    } else {
        // CLS query (space mode): all N keys.  A wave reads EIGHT key rows per instruction (lane = key 8g + (lane>>3), channels
        // 8 (lane&7) .. +7, 16 bytes each), so a wave's ~72 keys are ~9 loads that are all in flight at once; a score is an
        // 8-lane DPP reduction.  (One key per instruction with a wave-wide reduction per key was latency-bound: 40 us.)
        constexpr int UG = 5;                   // key groups per unrolled step
        float* S = sm;                          // [N] scores
        float* red = sm + a.N;                  // [4][64] partial outputs
        const int kl = lane >> 3, c8 = lane & 7;
        float q8[8];
        load8(q + brow0 * a.ld + h * HD + c8 * 8, q8);
#pragma unroll
        for (int e = 0; e < 8; ++e) q8[e] *= a.scale;
        const int ngroups = (a.N + 7) / 8;
        for (int g0 = wid; g0 < ngroups; g0 += 4 * UG) {
            float k8[UG][8];
#pragma unroll
            for (int u = 0; u < UG; ++u) {
                int j = 8 * (g0 + 4 * u) + kl;
                j = j < a.N ? j : a.N - 1;
                load8(k + (brow0 + j) * a.ld + h * HD + c8 * 8, k8[u]);
            }
#pragma unroll
            for (int u = 0; u < UG; ++u) {
                const int j = 8 * (g0 + 4 * u) + kl;
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) d += q8[e] * k8[u][e];
                d = oct_sum(d);
                if (c8 == 0 && j < a.N) S[j] = d + a.addmask[brow0 + j];
            }
        }
        __syncthreads();
        float m = -INFINITY;
        for (int j = lane; j < a.N; j += 64) m = fmaxf(m, S[j]);
        m = wave_max(m);
        float l = 0.f;
        for (int j = lane; j < a.N; j += 64) l += expf(S[j] - m);
        l = wave_sum(l);
        float o8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int g0 = wid; g0 < ngroups; g0 += 4 * UG) {
            float v8[UG][8];
#pragma unroll
            for (int u = 0; u < UG; ++u) {
                int j = 8 * (g0 + 4 * u) + kl;
                j = j < a.N ? j : a.N - 1;
                load8(v + (brow0 + j) * a.ld + h * HD + c8 * 8, v8[u]);
            }
#pragma unroll
            for (int u = 0; u < UG; ++u) {
                const int j = 8 * (g0 + 4 * u) + kl;
                const float pj = j < a.N ? expf(S[j] - m) : 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) o8[e] += pj * v8[u][e];
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) o8[e] = stride8_sum(o8[e]);
        if (kl == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[wid * 64 + c8 * 8 + e] = o8[e];
        }
        __syncthreads();
        if (wid == 0) out[brow0 * a.ldo + h * HD + lane] = from_f<T>((red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane]) / l);
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
            float k0m = 1.f, k1m = 1.f;                              // dropout multipliers of this query's probabilities
            if (a.keep && a.mode == 1) {
                const uint8_t* kr = a.keep + (((int64_t)b * a.H + h) * a.Ns + (c0 + i)) * a.Ns;
                k0m = (lane < nk && kr[lane]) ? a.kscale : 0.f;
                k1m = (lane + 64 < nk && kr[lane + 64 < a.Ns ? lane + 64 : 0]) ? a.kscale : 0.f;
                dp0 *= k0m; dp1 *= k1m;                              // d softmax = keep scale * d(dropped weights)
            }
            const float Dsum = wave_sum(p0 * dp0 + p1 * dp1);
            const float ds0 = p0 * (dp0 - Dsum), ds1 = p1 * (dp1 - Dsum);
            if (lane < nk) { Ps[i * nkp + lane] = p0 * k0m; dSs[i * nkp + lane] = ds0; }       // dV takes the DROPPED weights
            if (lane + 64 < nk) { Ps[i * nkp + lane + 64] = p1 * k1m; dSs[i * nkp + lane + 64] = ds1; }
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
    float* S = sm; float* DP = S + N; float* red = DP + N;   // red: [4][64]
    // Eight key rows per load instruction (lane = key 8g + (lane>>3), channels 8 (lane&7) .. +7), a wave's whole share of the
    // keys in flight at once, scores by 8-lane DPP reductions -- see the forward CLS branch.
    constexpr int UG = 5;
    const int kl = lane >> 3, c8 = lane & 7;
    float q8[8], do8[8];
    load8(q + brow0 * a.ld + h * HD + c8 * 8, q8);
    load8(dout + brow0 * a.ldo + h * HD + c8 * 8, do8);
#pragma unroll
    for (int e = 0; e < 8; ++e) q8[e] *= a.scale;
    const int ngroups = (N + 7) / 8;
    for (int g0 = wid; g0 < ngroups; g0 += 4 * UG) {
        float k8[UG][8], v8[UG][8];
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            int j = 8 * (g0 + 4 * u) + kl;
            j = j < N ? j : N - 1;
            load8(k + (brow0 + j) * a.ld + h * HD + c8 * 8, k8[u]);
            load8(v + (brow0 + j) * a.ld + h * HD + c8 * 8, v8[u]);
        }
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            const int j = 8 * (g0 + 4 * u) + kl;
            float sc = 0.f, dp = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { sc += q8[e] * k8[u][e]; dp += do8[e] * v8[u][e]; }
            sc = oct_sum(sc); dp = oct_sum(dp);
            if (c8 == 0 && j < N) { S[j] = sc + a.addmask[brow0 + j]; DP[j] = dp; }
        }
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
    float dq8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int g0 = wid; g0 < ngroups; g0 += 4 * UG) {
        float k8[UG][8], gk[UG][8], gv[UG][8];
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            int j = 8 * (g0 + 4 * u) + kl;
            j = j < N ? j : N - 1;
            load8(k + (brow0 + j) * a.ld + h * HD + c8 * 8, k8[u]);
            if (j == 0) {
                // the CLS key: totals of the per-frame partials left by launch 1
#pragma unroll
                for (int e = 0; e < 8; ++e) { gk[u][e] = 0.f; gv[u][e] = 0.f; }
                const float* w = a.ws + (((int64_t)b * a.H + h) * a.F) * 2 * HD;
                for (int f = 0; f < a.F; ++f) {
                    float t0[8], t1[8];
                    load8(w + f * 2 * HD + c8 * 8, t0); load8(w + f * 2 * HD + HD + c8 * 8, t1);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { gk[u][e] += t0[e]; gv[u][e] += t1[e]; }
                }
            } else {
                load8(dk + (brow0 + j) * a.ldd + h * HD + c8 * 8, gk[u]);
                load8(dv + (brow0 + j) * a.ldd + h * HD + c8 * 8, gv[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            const int j = 8 * (g0 + 4 * u) + kl;
            if (j < N) {
                const float pj = expf(S[j] - m) * inv, ds = pj * (DP[j] - Dsum);
#pragma unroll
                for (int e = 0; e < 8; ++e) { dq8[e] += ds * k8[u][e]; gk[u][e] += ds * q8[e]; gv[u][e] += pj * do8[e]; }
                store8(dk + (brow0 + j) * a.ldd + h * HD + c8 * 8, gk[u]);
                store8(dv + (brow0 + j) * a.ldd + h * HD + c8 * 8, gv[u]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) dq8[e] = stride8_sum(dq8[e]);
    if (kl == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[wid * 64 + c8 * 8 + e] = dq8[e];
    }
    __syncthreads();
    if (wid == 0) dq[brow0 * a.ldd + h * HD + lane] = from_f<T>((red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane]) * a.scale);
}


// ==================================================================================================================
// bf16 MFMA path (v_mfma_f32_16x16x32_bf16).  One WAVE owns one (batch, head, frame) [space] or one (batch, head,
// query-tile group) [full]; tiles are 16 x 16 with the 64-wide head dim as two k-steps.
//
// Trick that keeps P in registers between the two products: compute S^T = K Q^T (A = K rows, B = Q rows, both plain
// 16-byte global row reads), so a lane holds S^T[key = 16 kt + 4 (lane>>4) + r][q = lane&15]: its softmax column is
// (lane&15) and the row reductions are two xor-shuffles (16, 32).  The same registers, converted to bf16, ARE the A
// operand of O = P V if the contraction index of each 32-wide k-step is permuted as
//     k = 8 g + j  <->  key = 16 kt_(j>>2) + 4 g + (j&3)        (g = lane>>4)
// and the B operand (V) is read from a row-major LDS tile with the transposing read, whose 4-row blocks are exactly
// rows 16 kt + 4 g + {0..3}.  Backward uses the same rule in both orientations (S^T for dQ, S for dK/dV).
// ==================================================================================================================
constexpr int VLD = 72;    // LDS tile row stride in elements (144 B: 16-byte aligned rows, spreads the tr-read banks)

struct Seg {
    int mode, R, L, f, qbase;
    int cls_q = 0;             // space mode: query row R of the tile is the CLS token (merged backward)
    __device__ __forceinline__ int tok_q(int i) const {
        if (mode == 0) return i < R ? 1 + f * R + i : (cls_q && i == R ? 0 : -1);
        const int t = qbase + i;
        return t < L ? t : -1;
    }
    __device__ __forceinline__ int tok_k(int j) const {
        if (mode == 0) return j == 0 ? 0 : (j <= R ? f * R + j : -1);
        return j < L ? j : -1;
    }
};

__device__ __forceinline__ bf16x8 tr_pair(unsigned a0, unsigned a1) {
    bf16x4 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(lo), "=&v"(hi) : "v"(a0), "v"(a1) : "memory");
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)p; }

// copy ROWS token rows (64 bf16 of head h each) into a row-major LDS tile; rows without a token are zero
template <int ROWS>
__device__ __forceinline__ void stage_tile(bf16* Ts, const bf16* __restrict__ src, int64_t brow0, int64_t ld, int h, const Seg& sg, bool keys, int lane) {
#pragma unroll
    for (int it = 0; it < ROWS / 8; ++it) {
        const int row = it * 8 + (lane >> 3), ch = lane & 7;
        const int tok = keys ? sg.tok_k(row) : sg.tok_q(row);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (tok >= 0) v = *(const uint4*)(src + (brow0 + tok) * ld + h * HD + ch * 8);
        *(uint4*)&Ts[row * VLD + ch * 8] = v;
    }
}

// row fragments (A or B operand of a "X rows . Y rows" product): lane -> row 16 t + (lane&15), k = 32 ks + 8 (lane>>4) ..+7
template <int NT>
__device__ __forceinline__ void load_row_frags(bf16x8 (&f)[NT][2], const bf16* __restrict__ src, int64_t brow0, int64_t ld, int h, const Seg& sg, bool keys, int lane) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int tok = keys ? sg.tok_k(16 * t + (lane & 15)) : sg.tok_q(16 * t + (lane & 15));
        tok = tok < 0 ? 0 : tok;
        const bf16* p = src + (brow0 + tok) * ld + h * HD + 8 * (lane >> 4);
        f[t][0] = *(const bf16x8*)p;
        f[t][1] = *(const bf16x8*)(p + 32);
    }
}

// write row fragments (load_row_frags layout) into a row-major LDS tile and zero rows [16 NT, 16 NTP): the tile the
// transposing reads need is built from registers the wave already holds instead of a second trip to global memory
// (each extra load -> wait -> use phase costs a wave ~3 us of exposed latency, and these kernels are nothing but such phases)
template <int NT, int NTP>
__device__ __forceinline__ void put_row_frags(bf16* Ts, const bf16x8 (&f)[NT][2], int lane) {
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        *(bf16x8*)&Ts[(16 * t + c) * VLD + 8 * g] = f[t][0];
        *(bf16x8*)&Ts[(16 * t + c) * VLD + 32 + 8 * g] = f[t][1];
    }
    const bf16x8 z = {};
#pragma unroll
    for (int t = NT; t < NTP; ++t) {
        *(bf16x8*)&Ts[(16 * t + c) * VLD + 8 * g] = z;
        *(bf16x8*)&Ts[(16 * t + c) * VLD + 32 + 8 * g] = z;
    }
}

// Store NT accumulator tiles (rows 16 t + 4 g + r, channel 16 dt + c per lane: the MFMA C layout) to token rows of a
// [.., ld] global tensor through the wave's LDS tile, so that global memory sees 16-byte stores of whole 128-byte head rows
// instead of 2-byte ones.  `keys`: rows are keys (row 0 = the shared CLS key, skipped here) else queries.
template <int NT>
__device__ __forceinline__ void emit_rows(bf16* Ts, const f32x4 (&acc)[NT][4], float mul, bf16* __restrict__ dst, int64_t brow0, int64_t ld, int h,
                                          const Seg& sg, bool keys, int lane) {
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) Ts[(16 * t + 4 * g + r) * VLD + 16 * dt + c] = (bf16)(acc[t][dt][r] * mul);
#pragma unroll
    for (int it = 0; it < NT * 2; ++it) {
        const int row = it * 8 + (lane >> 3), ch = lane & 7;
        const int tok = keys ? (row == 0 ? -1 : sg.tok_k(row)) : sg.tok_q(row);
        if (tok >= 0) *(uint4*)(dst + (brow0 + tok) * ld + h * HD + ch * 8) = *(const uint4*)&Ts[row * VLD + ch * 8];
    }
}

// column sums (64 channels of head h) of the rows emit_rows just wrote from the wave's LDS tile -- the bf16 values as stored -- into dst[64]
template <int NT>
__device__ __forceinline__ void tile_colsum(const bf16* Ts, const Seg& sg, bool keys, int lane, float* __restrict__ dst) {
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < NT * 2; ++it) {
        const int row = it * 8 + (lane >> 3), ch = lane & 7;
        const int tok = keys ? (row == 0 ? -1 : sg.tok_k(row)) : sg.tok_q(row);
        if (tok >= 0) {
            const bf16x8 v = *(const bf16x8*)&Ts[row * VLD + ch * 8];
#pragma unroll
            for (int e = 0; e < 8; ++e) a8[e] += (float)v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) a8[e] = stride8_sum(a8[e]);
    if ((lane >> 3) == 0) {
        *(float4*)(dst + 8 * lane) = make_float4(a8[0], a8[1], a8[2], a8[3]);
        *(float4*)(dst + 8 * lane + 4) = make_float4(a8[4], a8[5], a8[6], a8[7]);
    }
}

__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
    bf16x8 r;
    r[0] = (bf16)a[0]; r[1] = (bf16)a[1]; r[2] = (bf16)a[2]; r[3] = (bf16)a[3];
    r[4] = (bf16)b[0]; r[5] = (bf16)b[1]; r[6] = (bf16)b[2]; r[7] = (bf16)b[3];
    return r;
}

// (space mode, NKT <= 3: four waves per SIMD; the 7 / 8-tile text form would spill there.)  The key mask is one 4-byte load per lane
// (key = lane) fanned out by ds_bpermute instead of twelve loads per lane (round 5: 116 -> 112 registers, 31.8 -> 30 us at the bench shape).
template <int NQT, int NKT, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE))) void mattn_fwd_kernel(AttnArgs a, int items, int qgroups) {
    extern __shared__ __attribute__((aligned(16))) char smraw[];
    constexpr int NKTP = (NKT + 1) & ~1;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, g = lane >> 4, c = lane & 15, qq = (lane >> 2) & 3, pp = lane & 3;
    const int item = blockIdx.x * 4 + wid;
    if (item >= items) return;                                   // wave-uniform: EXEC stays full for the tr-reads
    const bool fold = a.mode == 0 && a.cls_o != nullptr;        // wave-uniform: the CLS query is query row R of this frame's tile
    const int cq = a.R >> 4, cc = a.R & 15;                      // its tile and column in the S^T layout
    const bf16* q = (const bf16*)a.q; const bf16* k = (const bf16*)a.k; const bf16* v = (const bf16*)a.v; bf16* out = (bf16*)a.out;
    bf16* Vs = (bf16*)smraw + wid * (NKTP * 16 * VLD);
    bf16x8 qf[NQT][2], vfr[NKT][2];
    float mlane = 0.f;                                           // additive mask of key `lane` of the item's segment (-inf: no such key)
    Seg sg{a.mode, a.R, a.N, 0, 0};
    int h, b;
    if (a.mode == 0) { sg.f = item % a.F; h = (item / a.F) % a.H; b = item / (a.F * a.H); }
    else { sg.qbase = (item % qgroups) * 16 * NQT; h = (item / qgroups) % a.H; b = item / (qgroups * a.H); }
    Seg sge = sg;                                                // rows emitted to `out`: the frame's regions only
    sg.cls_q = fold ? 1 : 0;
    const int64_t brow0 = (int64_t)b * a.N;
    load_row_frags<NQT>(qf, q, brow0, a.ld, h, sg, false, lane);
    load_row_frags<NKT>(vfr, v, brow0, a.ld, h, sg, true, lane);      // one load phase: V goes to LDS from registers below
    if constexpr (NKT <= 4) { const int tok = sg.tok_k(lane); mlane = tok >= 0 ? a.addmask[brow0 + tok] : -INFINITY; }
    {
        f32x4 st[NKT][NQT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            int tokk = sg.tok_k(16 * kt + c);                    // K straight from memory, a tile at a time
            tokk = tokk < 0 ? 0 : tokk;
            const bf16* pk = k + (brow0 + tokk) * a.ld + h * HD + 8 * g;
            const bf16x8 k0 = *(const bf16x8*)pk, k1 = *(const bf16x8*)(pk + 32);
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf[qt][0], acc, 0, 0, 0);
                st[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf[qt][1], acc, 0, 0, 0);
            }
        }
        put_row_frags<NKT, NKTP>(Vs, vfr, lane);
        float mkc[NKT][4];                                       // key 16 kt + 4 g + r: one 4-byte load per lane fanned out (twelve loads per lane before)
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (NKT <= 4) mkc[kt][r] = __shfl(mlane, 16 * kt + 4 * g + r, 64);
                else { const int tok = sg.tok_k(16 * kt + 4 * g + r); mkc[kt][r] = tok >= 0 ? a.addmask[brow0 + tok] : -INFINITY; }
            }
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
            float m = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    st[kt][qt][r] = st[kt][qt][r] * a.scale + mkc[kt][r];
                    // CLS query x CLS key belongs to frame 0's partial only
                    if (kt == 0 && r == 0 && fold && qt == cq && c == cc && g == 0 && sg.f != 0) st[kt][qt][r] = -INFINITY;
                    m = fmaxf(m, st[kt][qt][r]);
                }
            m = col4_max(m);
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { st[kt][qt][r] = __expf(st[kt][qt][r] - m); sum += st[kt][qt][r]; }
            sum = col4_sum(sum);
            if (fold && qt == cq && c == cc && g == 0) { a.cls_st[2 * (int64_t)item] = m; a.cls_st[2 * (int64_t)item + 1] = sum; }
            const float inv = 1.f / sum;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) st[kt][qt][r] *= inv;
            if (a.keep && a.mode == 1) {                                 // weights = dropout(softmax(scores)) before the context product
                int qi = sg.qbase + 16 * qt + c;
                qi = qi < a.N ? qi : a.N - 1;
                const uint8_t* kr = a.keep + (((int64_t)b * a.H + h) * a.Ns + qi) * a.Ns;
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) {
                    const uint32_t kw = *(const uint32_t*)(kr + 16 * kt + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) st[kt][qt][r] *= ((kw >> (8 * r)) & 1u) ? a.kscale : 0.f;
                }
            }
        }
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        bf16x8 pa[NKTP / 2][NQT];                                // P as the context product's A operand (bf16): half the registers of `st`
        f32x4 o[NQT][4];
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[qt][dt] = zero4;
#pragma unroll
        for (int s = 0; s < NKTP / 2; ++s) {
            const int kt0 = 2 * s, kt1 = 2 * s + 1;
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt) pa[s][qt] = pack8(st[kt0][qt], kt1 < NKT ? st[kt1 < NKT ? kt1 : kt0][qt] : zero4);     // (a k-step at a time: the text form has no room for all of it)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 vb = tr_pair(lds_addr(&Vs[(16 * kt0 + 4 * g + qq) * VLD + 16 * dt + 4 * pp]),
                                          lds_addr(&Vs[(16 * kt1 + 4 * g + qq) * VLD + 16 * dt + 4 * pp]));
#pragma unroll
                for (int qt = 0; qt < NQT; ++qt) o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[s][qt], vb, o[qt][dt], 0, 0, 0);
            }
        }
        if (fold) {          // the CLS row's partial (normalised over this frame's keys) stays fp32: merged by attn_fwd_cls_combine_kernel
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * qt + 4 * g + r == a.R) {
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt) a.cls_o[(int64_t)item * HD + 16 * dt + c] = o[qt][dt][r];
                    }
        }
        emit_rows<NQT>(Vs, o, 1.f, out, brow0, a.ldo, h, sge, false, lane);
    }
}

// ==================================================================================================================
// Round 5 -- space attention, bf16, the "lean" forward.  Round 4 read this kernel as latency-bound at 3.6 TB/s (one load -> compute -> store
// pass per wave).  Two experiments say otherwise (profiles/r5_attention_experiments.txt): a persistent form with the NEXT item's operands in
// flight through the whole current item is no faster (27.5-31 us against 26-30), at two waves per SIMD as at four, with the round-4 stream
// as with this one; and halving the instruction stream (1 719 -> 905 instructions per 37 x 37 x 64 item, 42 of them MFMAs) buys 15 %.
// The first round of 4 096 waves asks for 75 MB at once and the memory system delivers this access shape -- 16 rows x 64 bytes per load
// instruction, 128-byte head slices 4.6 KB apart -- at ~4.3 TB/s whatever is in flight: the kernel is memory-bound at the rate of its
// access shape, and what the instruction diet removes is the tail behind the last round's loads.  The diet:
//   * operands through buffer loads: one descriptor per tensor, the (batch, head) part of the address in the scalar offset, a 32-bit
//     per-lane row offset (one multiply-add per row tile);
//   * the context product with its operands swapped (O^T = V^T P^T): a lane then holds FOUR CONSECUTIVE CHANNELS of one query row, so the
//     output is staged with one 8-byte LDS store per 16 x 16 tile (12 instead of 48 two-byte ones) and the CLS partial leaves as four
//     16-byte stores under one test;
//   * the four transposing reads of a k-step pair issued together, one wait.
// Same arithmetic per output element as mattn_fwd_kernel (the swapped product sums the same terms in the same order).
// ==================================================================================================================
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x8 buf_row16(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// the B (or A) operands of four 16-channel tiles of one 32-deep k-step from a row-major LDS tile: eight transposing reads, one wait
__device__ __forceinline__ void tr_quad(bf16x8 (&o)[4], unsigned a0, unsigned a1) {
    bf16x4 l0, h0, l1, h1, l2, h2, l3, h3;
    asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %9\n\t"
                 "ds_read_b64_tr_b16 %2, %8 offset:32\n\tds_read_b64_tr_b16 %3, %9 offset:32\n\t"
                 "ds_read_b64_tr_b16 %4, %8 offset:64\n\tds_read_b64_tr_b16 %5, %9 offset:64\n\t"
                 "ds_read_b64_tr_b16 %6, %8 offset:96\n\tds_read_b64_tr_b16 %7, %9 offset:96\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1), "=&v"(l2), "=&v"(h2), "=&v"(l3), "=&v"(h3) : "v"(a0), "v"(a1) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    o[0] = __builtin_shufflevector(l0, h0, 0, 1, 2, 3, 4, 5, 6, 7);
    o[1] = __builtin_shufflevector(l1, h1, 0, 1, 2, 3, 4, 5, 6, 7);
    o[2] = __builtin_shufflevector(l2, h2, 0, 1, 2, 3, 4, 5, 6, 7);
    o[3] = __builtin_shufflevector(l3, h3, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ u32x2_t pack4(const f32x4& a) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t lo = {(bf16)a[0], (bf16)a[1]}, hi = {(bf16)a[2], (bf16)a[3]};
    return (u32x2_t){__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
}

// One WORKGROUP per (batch, head): eight waves, wave w takes frames w, w + 8, ...  The CLS query's per-frame partials (normalised output over
// the frame's keys + (max, sum)) therefore meet in LDS: after one barrier wave 0 merges them flash-style, in frame order -- exactly what
// attn_fwd_cls_combine_kernel did from global memory one launch later (bit-equal; that 64-thread launch, its ~1.5 us boundary and the partials'
// round trip through memory are gone).  The padded fourth key tile of a 37-key frame is one zero tile shared by the workgroup, so a wave's
// tile is 48 rows: 8 x 6.9 KB + 2.3 KB + F x 264 B per workgroup, two workgroups per CU.
constexpr int SATTN_WAVES = 8;
template <int NT>
__global__ __launch_bounds__(64 * SATTN_WAVES) __attribute__((amdgpu_waves_per_eu(4))) void sattn_fwd_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smraw[];
    constexpr int NTP = (NT + 1) & ~1;
    constexpr int TILE = NT * 16 * VLD;                          // elements per wave tile (no padding rows: see the zero tile)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, g = lane >> 4, c = lane & 15, qq = (lane >> 2) & 3, pp = lane & 3;
    const int bh = (int)blockIdx.x, h = bh % a.H, b = bh / a.H;
    const int R = a.R;
    const int cq = R >> 4, cc = R & 15;
    const int ldb = (int)a.ld * 2, ldob = (int)a.ldo * 2;
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.q), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.k), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.v), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, 0x7fffffff, 0x00020000);
    bf16* Vs = (bf16*)smraw + wid * TILE;
    bf16* Zt = (bf16*)smraw + SATTN_WAVES * TILE;                // 16 zero rows: the second half of an odd tile count's last k-step
    const bool merge_here = gridDim.y == 1;                      // every frame of this (batch, head) is in this workgroup
    float* slot_o = (float*)(Zt + 16 * VLD);                     // [8][64] the CLS row's partial outputs
    float* slot_st = slot_o + SATTN_WAVES * HD;                  // [8][2]  (max, sum) over the frame's keys
    if (NT != NTP) {
        for (int i = threadIdx.x; i < 16 * VLD / 8; i += blockDim.x) *(uint4*)&Zt[8 * i] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
    }
    const int soff = (b * a.N * (int)a.ld + h * HD) * 2, soffo = (b * a.N * (int)a.ldo + h * HD) * 2;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int f = (int)blockIdx.y * SATTN_WAVES + wid;
    if (f < a.F) {                                               // (wave-uniform: EXEC stays full for the tr-reads)
        const int fR = f * R;
        // one load phase: q (rows = the frame's regions, + the CLS token as row R), k / v (row 0 = the CLS token, rows 1..R the regions);
        // rows past the end re-read token 0 (finite values; their scores are masked / their outputs not written)
        bf16x8 qf[NT][2], kf[NT][2], vfr[NT][2];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int i = 16 * t + c;
            const int vq = (i < R ? 1 + fR + i : 0) * ldb + 16 * g, vk = (i >= 1 && i <= R ? fR + i : 0) * ldb + 16 * g;
            qf[t][0] = buf_row16(rq, vq, soff); qf[t][1] = buf_row16(rq, vq + 64, soff);
            kf[t][0] = buf_row16(rk, vk, soff); kf[t][1] = buf_row16(rk, vk + 64, soff);
            vfr[t][0] = buf_row16(rv, vk, soff); vfr[t][1] = buf_row16(rv, vk + 64, soff);
        }
        const float mlane = lane <= R ? a.addmask[(int64_t)b * a.N + (lane == 0 ? 0 : fR + lane)] : -INFINITY;      // additive mask of key `lane`
        if (a.abl & 8) {     // TIMING ONLY (tools/attn_bench.py): the kernel's memory traffic without its arithmetic -- every operand loaded, the output rows stored
            bf16x8 acc8 = vfr[0][0];
#pragma unroll
            for (int t = 0; t < NT; ++t) { acc8 = acc8 + qf[t][0] + qf[t][1] + kf[t][0] + kf[t][1] + vfr[t][0] + vfr[t][1]; }
#pragma unroll
            for (int it = 0; it < NT * 2; ++it) {
                const int row = it * 8 + (lane >> 3), ch = lane & 7;
                if (row < R) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc8), ro, (1 + fR + row) * ldob + ch * 16, soffo, 0);
            }
            if (mlane == 123.4567f) a.cls_stats[0] = 1.f;
            return;
        }
        f32x4 st[NT][NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int qt = 0; qt < NT; ++qt) {
                const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][0], qf[qt][0], zero4, 0, 0, 0);
                st[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][1], qf[qt][1], acc, 0, 0, 0);
            }
        put_row_frags<NT, NT>(Vs, vfr, lane);
        float mk[NT][4];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mk[kt][r] = __shfl(mlane, 16 * kt + 4 * g + r, 64);
        bf16x8 pa[NTP / 2][NT];
#pragma unroll
        for (int qt = 0; qt < NT; ++qt) {
            float m = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    st[kt][qt][r] = st[kt][qt][r] * a.scale + mk[kt][r];
                    // CLS query x CLS key belongs to frame 0's partial only
                    if (kt == 0 && r == 0 && qt == cq && c == cc && g == 0 && f != 0) st[kt][qt][r] = -INFINITY;
                    m = fmaxf(m, st[kt][qt][r]);
                }
            m = col4_max(m);
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { st[kt][qt][r] = __expf(st[kt][qt][r] - m); sum += st[kt][qt][r]; }
            sum = col4_sum(sum);
            if (qt == cq && c == cc && g == 0) { slot_st[2 * wid] = m; slot_st[2 * wid + 1] = sum; }
            const float inv = 1.f / sum;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) st[kt][qt][r] *= inv;
#pragma unroll
            for (int s = 0; s < NTP / 2; ++s) pa[s][qt] = pack8(st[2 * s][qt], 2 * s + 1 < NT ? st[2 * s + 1 < NT ? 2 * s + 1 : 2 * s][qt] : zero4);
        }
        // O^T = V^T P^T: o[qt][dt][r] = O[query 16 qt + c][channel 16 dt + 4 g + r]
        f32x4 o[NT][4];
#pragma unroll
        for (int qt = 0; qt < NT; ++qt)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[qt][dt] = zero4;
#pragma unroll
        for (int s = 0; s < NTP / 2; ++s) {
            bf16x8 vb[4];
            const bf16* hi = 2 * s + 1 < NT ? &Vs[(32 * s + 16 + 4 * g + qq) * VLD + 4 * pp] : &Zt[(4 * g + qq) * VLD + 4 * pp];
            tr_quad(vb, lds_addr(&Vs[(32 * s + 4 * g + qq) * VLD + 4 * pp]), lds_addr(hi));
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int qt = 0; qt < NT; ++qt) o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb[dt], pa[s][qt], o[qt][dt], 0, 0, 0);
        }
        // the CLS row's partial (normalised over this frame's keys) stays fp32, in this wave's LDS slot
#pragma unroll
        for (int qt = 0; qt < NT; ++qt)
            if (qt == cq && c == cc) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    *(float4*)&slot_o[wid * HD + 16 * dt + 4 * g] = make_float4(o[qt][dt][0], o[qt][dt][1], o[qt][dt][2], o[qt][dt][3]);
            }
        // stage the frame's rows in the tile (8-byte pieces), leave as whole 128-byte head rows
#pragma unroll
        for (int qt = 0; qt < NT; ++qt)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) *(u32x2_t*)&Vs[(16 * qt + c) * VLD + 16 * dt + 4 * g] = pack4(o[qt][dt]);
#pragma unroll
        for (int it = 0; it < NT * 2; ++it) {
            const int row = it * 8 + (lane >> 3), ch = lane & 7;
            if (row < R)
                __builtin_amdgcn_raw_buffer_store_b128(*(const u32x4_t*)&Vs[row * VLD + ch * 8], ro, (1 + fR + row) * ldob + ch * 16, soffo, 0);
        }
    }
    if (!merge_here) {   // more than eight frames: this wave's partial goes to global memory for the combine launch
        if (f < a.F) {
            const int64_t item = (int64_t)bh * a.F + f;
            a.cls_o[item * HD + lane] = slot_o[wid * HD + lane];
            if (lane < 2) a.cls_st[2 * item + lane] = slot_st[2 * wid + lane];
        }
        return;
    }
    __syncthreads();
    if (wid == 0) {      // the CLS query's output: out = sum_f w_f o_f, w_f = l_f e^(m_f - M) / L; (M, L) kept for the backward
        float M = -INFINITY;
        for (int w8 = 0; w8 < a.F; ++w8) M = fmaxf(M, slot_st[2 * w8]);
        float L = 0.f, acc = 0.f;
        for (int w8 = 0; w8 < a.F; ++w8) { const float w = slot_st[2 * w8 + 1] * __expf(slot_st[2 * w8] - M); L += w; acc += w * slot_o[w8 * HD + lane]; }
        ((bf16*)a.out)[(int64_t)b * a.N * a.ldo + h * HD + lane] = (bf16)(acc / L);
        if (lane == 0) { float* s4 = a.cls_stats + (int64_t)bh * 4; s4[0] = M; s4[1] = L; s4[2] = 0.f; s4[3] = 0.f; }
    }
}

// CLS query, forward: merge the F per-frame partials of one (b, h) -- out = sum_f w_f o_f, w_f = l_f e^(m_f - M) / L -- and keep the
// global softmax statistics (M, L) for the backward
__global__ __launch_bounds__(64) void attn_fwd_cls_combine_kernel(AttnArgs a) {
    const int lane = threadIdx.x, h = blockIdx.x, b = blockIdx.y;
    const int64_t bh = (int64_t)b * a.H + h;
    const float* st = a.cls_st + bh * a.F * 2;
    const float* o = a.cls_o + bh * a.F * HD;
    float M = -INFINITY;
    for (int f = 0; f < a.F; ++f) M = fmaxf(M, st[2 * f]);
    float L = 0.f, acc = 0.f;
    for (int f = 0; f < a.F; ++f) { const float w = st[2 * f + 1] * __expf(st[2 * f] - M); L += w; acc += w * o[f * HD + lane]; }
    ((bf16*)a.out)[(int64_t)b * a.N * a.ldo + h * HD + lane] = (bf16)(acc / L);
    if (lane == 0) { float* s4 = a.cls_stats + bh * 4; s4[0] = M; s4[1] = L; s4[2] = 0.f; s4[3] = 0.f; }
}

// backward, space mode: one wave per (b, h, frame).  Layout 1 (S^T) -> dQ; layout 2 (S) -> dK, dV.  P is recomputed.
template <int NQT, int NKT, int PART>
__global__ __launch_bounds__(256) void mattn_bwd_space_kernel(AttnArgs a, int items) {
    extern __shared__ __attribute__((aligned(16))) char smraw[];
    constexpr int NQTP = (NQT + 1) & ~1, NKTP = (NKT + 1) & ~1, TROWS = 16 * (NQTP > NKTP ? NQTP : NKTP);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, g = lane >> 4, c = lane & 15, qq = (lane >> 2) & 3, pp = lane & 3;
    const int item = blockIdx.x * 4 + wid;
    if (item >= items) return;
    Seg sg{0, a.R, a.N, item % a.F, 0};
    const int h = (item / a.F) % a.H, b = item / (a.F * a.H);
    const int64_t brow0 = (int64_t)b * a.N;
    const bf16* q = (const bf16*)a.q; const bf16* k = (const bf16*)a.k; const bf16* v = (const bf16*)a.v; const bf16* dout = (const bf16*)a.dout;
    bf16* dq = (bf16*)a.dq; bf16* dk = (bf16*)a.dk; bf16* dv = (bf16*)a.dv;
    bf16* Ts = (bf16*)smraw + wid * (TROWS * VLD);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // ---------------- layout 1: keys on (g, r), queries on lane&15 -> dQ ----------------
    if (PART == 0) {
        bf16x8 qf[NQT][2], gf[NQT][2], kf[NKT][2], vf[NKT][2];
        load_row_frags<NQT>(qf, q, brow0, a.ld, h, sg, false, lane);
        load_row_frags<NQT>(gf, dout, brow0, a.ldo, h, sg, false, lane);
        load_row_frags<NKT>(kf, k, brow0, a.ld, h, sg, true, lane);
        load_row_frags<NKT>(vf, v, brow0, a.ld, h, sg, true, lane);
        f32x4 st[NKT][NQT], dp[NKT][NQT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt) {
                f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][0], qf[qt][0], zero4, 0, 0, 0);
                st[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][1], qf[qt][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[kt][0], gf[qt][0], zero4, 0, 0, 0);
                dp[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[kt][1], gf[qt][1], acc, 0, 0, 0);
            }
        float mk[NKT][4];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int tok = sg.tok_k(16 * kt + 4 * g + r); mk[kt][r] = tok >= 0 ? a.addmask[brow0 + tok] : -INFINITY; }
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt) {
            float m = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { st[kt][qt][r] = st[kt][qt][r] * a.scale + mk[kt][r]; m = fmaxf(m, st[kt][qt][r]); }
            m = col4_max(m);
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { st[kt][qt][r] = __expf(st[kt][qt][r] - m); sum += st[kt][qt][r]; }
            sum = col4_sum(sum);
            const float inv = 1.f / sum;
            float D = 0.f;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { st[kt][qt][r] *= inv; D += st[kt][qt][r] * dp[kt][qt][r]; }
            D = col4_sum(D);
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) st[kt][qt][r] = st[kt][qt][r] * (dp[kt][qt][r] - D);      // dS^T
        }
        put_row_frags<NKT, NKTP>(Ts, kf, lane);
        f32x4 acc[NQT][4];
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) acc[qt][dt] = zero4;
#pragma unroll
        for (int s = 0; s < NKTP / 2; ++s) {
            const int kt0 = 2 * s, kt1 = 2 * s + 1;
            bf16x8 da[NQT];
#pragma unroll
            for (int qt = 0; qt < NQT; ++qt) da[qt] = pack8(st[kt0][qt], kt1 < NKT ? st[kt1 < NKT ? kt1 : kt0][qt] : zero4);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 kb = tr_pair(lds_addr(&Ts[(16 * kt0 + 4 * g + qq) * VLD + 16 * dt + 4 * pp]),
                                          lds_addr(&Ts[(16 * kt1 + 4 * g + qq) * VLD + 16 * dt + 4 * pp]));
#pragma unroll
                for (int qt = 0; qt < NQT; ++qt) acc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da[qt], kb, acc[qt][dt], 0, 0, 0);
            }
        }
        emit_rows<NQT>(Ts, acc, a.scale, dq, brow0, a.ldd, h, sg, false, lane);
    }
    // ---------------- layout 2: queries on (g, r), keys on lane&15 -> dV, dK ----------------
    if (PART == 1) {
        bf16x8 qf[NQT][2], gf[NQT][2], kf[NKT][2], vf[NKT][2];
        load_row_frags<NQT>(qf, q, brow0, a.ld, h, sg, false, lane);
        load_row_frags<NQT>(gf, dout, brow0, a.ldo, h, sg, false, lane);
        load_row_frags<NKT>(kf, k, brow0, a.ld, h, sg, true, lane);
        load_row_frags<NKT>(vf, v, brow0, a.ld, h, sg, true, lane);
        f32x4 s2[NQT][NKT], dp[NQT][NKT];
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[qt][0], kf[kt][0], zero4, 0, 0, 0);
                s2[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[qt][1], kf[kt][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[qt][0], vf[kt][0], zero4, 0, 0, 0);
                dp[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[qt][1], vf[kt][1], acc, 0, 0, 0);
            }
        float mk[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) { const int tok = sg.tok_k(16 * kt + c); mk[kt] = tok >= 0 ? a.addmask[brow0 + tok] : -INFINITY; }
#pragma unroll
        for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool qok = sg.tok_q(16 * qt + 4 * g + r) >= 0;
                float m = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) { s2[qt][kt][r] = s2[qt][kt][r] * a.scale + mk[kt]; m = fmaxf(m, s2[qt][kt][r]); }
                m = row16_max(m);
                float sum = 0.f;
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) { s2[qt][kt][r] = (a.abl & 2) ? s2[qt][kt][r] - m : __expf(s2[qt][kt][r] - m); sum += s2[qt][kt][r]; }
                sum = row16_sum(sum);
                const float inv = qok ? 1.f / sum : 0.f;                    // padded query rows contribute nothing
                float D = 0.f;
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) { s2[qt][kt][r] *= inv; D += s2[qt][kt][r] * dp[qt][kt][r]; }
                D = row16_sum(D);
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) dp[qt][kt][r] = s2[qt][kt][r] * (dp[qt][kt][r] - D);   // dS
            }
        if (a.abl & 4) { if (dp[0][0][0] == 123.4567f) dk[0] = (bf16)1.f; return; }
        // dV[key][d] = sum_q P[q][key] dO[q][d]
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 0) put_row_frags<NQT, NQTP>(Ts, gf, lane); else put_row_frags<NQT, NQTP>(Ts, qf, lane);
            f32x4 acc[NKT][4];
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) acc[kt][dt] = zero4;
#pragma unroll
            for (int s = 0; s < NQTP / 2; ++s) {
                const int qt0 = 2 * s, qt1 = 2 * s + 1;
                bf16x8 pa[NKT];
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) {
                    const f32x4& lo = pass == 0 ? s2[qt0][kt] : dp[qt0][kt];
                    const f32x4& hi = qt1 < NQT ? (pass == 0 ? s2[qt1 < NQT ? qt1 : qt0][kt] : dp[qt1 < NQT ? qt1 : qt0][kt]) : zero4;
                    pa[kt] = pack8(lo, hi);
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const bf16x8 xb = tr_pair(lds_addr(&Ts[(16 * qt0 + 4 * g + qq) * VLD + 16 * dt + 4 * pp]),
                                              lds_addr(&Ts[(16 * qt1 + 4 * g + qq) * VLD + 16 * dt + 4 * pp]));
#pragma unroll
                    for (int kt = 0; kt < NKT; ++kt) acc[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[kt], xb, acc[kt][dt], 0, 0, 0);
                }
            }
            const float mul = pass == 0 ? 1.f : a.scale;
            bf16* dst = pass == 0 ? dv : dk;
            if (a.abl & 1) { if (acc[0][0][0] == 123.4567f) dst[0] = (bf16)1.f; continue; }
            if (g == 0) {          // key 0 of the frame = the shared CLS key: fp32 partial for the second launch
                float* w = a.ws + ((((int64_t)b * a.H + h) * a.F + sg.f) * 2 + (pass == 0 ? 1 : 0)) * HD + c;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) w[16 * dt] = acc[0][dt][0] * mul;
            }
            emit_rows<NKT>(Ts, acc, mul, dst, brow0, a.ldd, h, sg, true, lane);
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Space-mode backward in ONE pass over q, k, v, dO (bf16).  The three-launch form above reads them twice in the two frame
// launches and then read-modify-writes every dK / dV row in the CLS launch: ~625 MB per layer for 200 MB of useful traffic,
// and the kernels are HBM-bound.  Here:
//   launch A (attn_bwd_cls_pre_kernel, one workgroup per (b, h)): the CLS query's softmax statistics over all N keys
//            (max, sum, D = sum_j p_j <dO_cls, v_j>) -> stats[b][h][0..2], and dq_cls;
//   launch B (this kernel, one wave per (b, h, frame)): layout 2 only.  The CLS query rides along as query row R of every
//            frame tile, normalised with the GLOBAL statistics of launch A instead of the tile's own, so its contributions
//            to dK / dV of the frame's keys come out of the same MFMAs (its product with the CLS key is kept in frame 0
//            only).  dQ = dS K goes through the wave's LDS tile: dS (C layout) is written as bf16 and read back as the A
//            operand, K is read with the transposing read.  Per-frame partials of the shared CLS key go to `ws` as before;
//   launch C (attn_bwd_cls_post_kernel): sums those partials into dK / dV of the CLS key.
// ------------------------------------------------------------------------------------------------------------------
// (two waves per SIMD: left alone the compiler takes 328 VGPRs for NT = 3 -- one wave per SIMD, 102 us per layer.  Rounds 1-4 re-read Q and
//  K from global memory for the second / third stage instead of holding them: 244 registers, 62 us, three exposed load latencies per wave.)
// Round 5: (1) TWO LDS tiles per wave (18 KB x 8 waves fit the CU at two waves per SIMD): dO and Q are parked in them as soon as they
// arrive, K follows when the first tile is free -- no operand is fetched twice and none is held in registers across a stage; (2) P and dS
// are kept as the bf16 operands the three later products consume (48 registers instead of 72).  62 -> 56 us per layer at the bench shape.
template <int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void mattn_bwd_space_merged_kernel(AttnArgs a, int items, const float* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) char smraw[];
    constexpr int NTP = (NT + 1) & ~1, TROWS = 16 * NTP, NS = NTP / 2;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, g = lane >> 4, c = lane & 15, qq = (lane >> 2) & 3, pp = lane & 3;
    const int item = blockIdx.x * 4 + wid;
    if (item >= items) return;
    const bf16* q = (const bf16*)a.q; const bf16* k = (const bf16*)a.k; const bf16* v = (const bf16*)a.v; const bf16* dout = (const bf16*)a.dout;
    bf16* dq = (bf16*)a.dq; bf16* dk = (bf16*)a.dk; bf16* dv = (bf16*)a.dv;
    bf16* T1 = (bf16*)smraw + wid * (2 * TROWS * VLD);
    bf16* T2 = T1 + TROWS * VLD;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const Seg sg{0, a.R, a.N, item % a.F, 0, 1};                // queries: the frame's R regions + the CLS query as row R
    const Seg sgp{0, a.R, a.N, item % a.F, 0, 0};               // the same without the CLS row (dQ rows written here)
    const int h = (item / a.F) % a.H, b = item / (a.F * a.H);
    const int64_t brow0 = (int64_t)b * a.N;
    bf16x8 qf[NT][2], gf[NT][2], kf[NT][2], vf[NT][2];
    load_row_frags<NT>(qf, q, brow0, a.ld, h, sg, false, lane);
    load_row_frags<NT>(gf, dout, brow0, a.ldo, h, sg, false, lane);
    load_row_frags<NT>(kf, k, brow0, a.ld, h, sg, true, lane);
    load_row_frags<NT>(vf, v, brow0, a.ld, h, sg, true, lane);
    float mlane;
    { const int tok = sg.tok_k(lane); mlane = tok >= 0 ? a.addmask[brow0 + tok] : -INFINITY; }
    const float* st3 = stats + ((int64_t)b * a.H + h) * 4;
    const float m_cls = st3[0], il_cls = 1.f / st3[1];
    float D_cls = st3[2];
    if (a.fwd_out) {     // statistics saved by the forward: D = sum_j p_j <dO_cls, v_j> = <dO_cls, O_cls>
        const float pd = (float)((const bf16*)a.fwd_out)[brow0 * a.ld_fo + h * HD + lane] * (float)dout[brow0 * a.ldo + h * HD + lane];
        D_cls = wave_sum(pd);
    }
    {
        put_row_frags<NT, NTP>(T1, gf, lane);                    // dO -> tile 1 (dV pass), Q -> tile 2 (dK pass)
        put_row_frags<NT, NTP>(T2, qf, lane);
        f32x4 s2[NT][NT], dp[NT][NT];
#pragma unroll
        for (int qt = 0; qt < NT; ++qt)
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[qt][0], vf[kt][0], zero4, 0, 0, 0);
                dp[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[qt][1], vf[kt][1], acc, 0, 0, 0);
            }
#pragma unroll
        for (int qt = 0; qt < NT; ++qt)
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[qt][0], kf[kt][0], zero4, 0, 0, 0);
                s2[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[qt][1], kf[kt][1], acc, 0, 0, 0);
            }
        float mk[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) mk[kt] = __shfl(mlane, 16 * kt + c, 64);
#pragma unroll
        for (int qt = 0; qt < NT; ++qt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qi = 16 * qt + 4 * g + r;
                const bool qok = sg.tok_q(qi) >= 0, is_cls = qi == a.R;
                float m = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) { s2[qt][kt][r] = s2[qt][kt][r] * a.scale + mk[kt]; m = fmaxf(m, s2[qt][kt][r]); }
                m = row16_max(m);
                if (is_cls) m = m_cls;                                       // the CLS query's softmax spans all N keys
                float sum = 0.f;
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) {
                    float e = __expf(s2[qt][kt][r] - m);
                    if (is_cls && kt == 0 && c == 0 && sg.f != 0) e = 0.f;    // CLS query x CLS key: counted once, in frame 0
                    s2[qt][kt][r] = e; sum += e;
                }
                sum = row16_sum(sum);
                const float inv = is_cls ? il_cls : (qok ? 1.f / sum : 0.f);      // padded query rows contribute nothing
                float D = 0.f;
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) { s2[qt][kt][r] *= inv; D += s2[qt][kt][r] * dp[qt][kt][r]; }
                D = row16_sum(D);
                if (is_cls) D = D_cls;
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) dp[qt][kt][r] = s2[qt][kt][r] * (dp[qt][kt][r] - D);   // dS
            }
        // P and dS as the bf16 A operands of the key-major products (k index = query: tiles 2 s, 2 s + 1)
        bf16x8 pP[NS][NT], pS[NS][NT];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                const int qt0 = 2 * s, qt1 = 2 * s + 1;
                pP[s][kt] = pack8(s2[qt0][kt], qt1 < NT ? s2[qt1 < NT ? qt1 : qt0][kt] : zero4);
                pS[s][kt] = pack8(dp[qt0][kt], qt1 < NT ? dp[qt1 < NT ? qt1 : qt0][kt] : zero4);
            }
        // dV[key][d] = sum_q P[q][key] dO[q][d];  dK[key][d] = scale * sum_q dS[q][key] Q[q][d]
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            bf16* Ts = pass == 0 ? T1 : T2;
            f32x4 acc[NT][4];
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) acc[kt][dt] = zero4;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int qt0 = 2 * s, qt1 = 2 * s + 1;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const bf16x8 xb = tr_pair(lds_addr(&Ts[(16 * qt0 + 4 * g + qq) * VLD + 16 * dt + 4 * pp]),
                                              lds_addr(&Ts[(16 * qt1 + 4 * g + qq) * VLD + 16 * dt + 4 * pp]));
#pragma unroll
                    for (int kt = 0; kt < NT; ++kt) acc[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pass == 0 ? pP[s][kt] : pS[s][kt], xb, acc[kt][dt], 0, 0, 0);
                }
            }
            const float mul = pass == 0 ? 1.f : a.scale;
            bf16* dst = pass == 0 ? dv : dk;
            if (g == 0) {          // key 0 of the frame = the shared CLS key: fp32 partial for launch C
                float* w = a.ws + ((((int64_t)b * a.H + h) * a.F + sg.f) * 2 + (pass == 0 ? 1 : 0)) * HD + c;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) w[16 * dt] = acc[0][dt][0] * mul;
            }
            emit_rows<NT>(Ts, acc, mul, dst, brow0, a.ldd, h, sg, true, lane);
            if (a.csum) tile_colsum<NT>(Ts, sg, true, lane, a.csum + ((int64_t)b * a.F + sg.f) * (3 * a.H * HD) + (pass == 0 ? 2 : 1) * a.H * HD + h * HD);
            if (pass == 0) put_row_frags<NT, NTP>(T1, kf, lane);  // tile 1 is free: K (held in registers so far) moves in, read back transposed for dQ
        }
        // dQ[q][d] = scale * sum_key dS[q][key] K[key][d]: K through tile 1 (transposing read, natural key order), dS through tile 2
        bf16x8 kb[NS][4];
#pragma unroll
        for (int ks = 0; ks < NS; ++ks)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                kb[ks][dt] = tr_pair(lds_addr(&T1[(32 * ks + 8 * g + qq) * VLD + 16 * dt + 4 * pp]),
                                     lds_addr(&T1[(32 * ks + 8 * g + 4 + qq) * VLD + 16 * dt + 4 * pp]));
        // (columns >= 16 NT of the rows written next keep finite values of the tile's previous use; they meet the zero K rows of the padded k-step)
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int qt = 2 * s + (e >> 2), r = e & 3;
                    if (qt < NT) T2[(16 * qt + 4 * g + r) * VLD + 16 * kt + c] = pS[s][kt][e];
                }
        f32x4 dqa[NT][4];
#pragma unroll
        for (int qt = 0; qt < NT; ++qt)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dqa[qt][dt] = zero4;
#pragma unroll
        for (int ks = 0; ks < NS; ++ks)
#pragma unroll
            for (int qt = 0; qt < NT; ++qt) {
                const bf16x8 dsf = *(const bf16x8*)&T2[(16 * qt + c) * VLD + 32 * ks + 8 * g];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) dqa[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsf, kb[ks][dt], dqa[qt][dt], 0, 0, 0);
            }
        if (a.dq_ws) {       // dq of the CLS query: this frame's share (summed by attn_bwd_cls_post_kernel)
#pragma unroll
            for (int qt = 0; qt < NT; ++qt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * qt + 4 * g + r == a.R) {
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt) a.dq_ws[(int64_t)item * HD + 16 * dt + c] = dqa[qt][dt][r] * a.scale;
                    }
        }
        emit_rows<NT>(T1, dqa, a.scale, dq, brow0, a.ldd, h, sgp, false, lane);
        if (a.csum) tile_colsum<NT>(T1, sgp, false, lane, a.csum + ((int64_t)b * a.F + sgp.f) * (3 * a.H * HD) + h * HD);
    }
}

// Round 5 -- space attention backward, bf16, the "lean" form of the kernel above (same products, same sums; see sattn_fwd_kernel for the
// three ideas).  Every output product runs with its operands swapped (dV^T = dO^T P, dK^T = Q^T dS, dQ^T = K^T dS^T), so a lane holds four
// consecutive channels of one key / query row: staging is one 8-byte LDS store per 16 x 16 tile, the CLS key's and the CLS query's
// partials leave as 16-byte stores from the four lanes that hold them.  dS reaches the dQ product through the tile TRANSPOSED ([key][query],
// 8-byte pieces straight from the packed registers, read back with the transposing read) instead of 36 two-byte stores.
template <int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void sattn_bwd_kernel(AttnArgs a, int items, const float* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) char smraw[];
    constexpr int NTP = (NT + 1) & ~1, TROWS = 16 * NTP, NS = NTP / 2;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, g = lane >> 4, c = lane & 15, qq = (lane >> 2) & 3, pp = lane & 3;
    const int item = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wid));
    if (item >= items) return;
    const int R = a.R;
    const int cq = R >> 4, cc = R & 15;
    const int ldb = (int)a.ld * 2, ldob = (int)a.ldo * 2, lddb = (int)a.ldd * 2;
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.q), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.k), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.v), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dout), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdq = __builtin_amdgcn_make_buffer_rsrc(a.dq, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdk = __builtin_amdgcn_make_buffer_rsrc(a.dk, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdv = __builtin_amdgcn_make_buffer_rsrc(a.dv, 0, 0x7fffffff, 0x00020000);
    bf16* T1 = (bf16*)smraw + wid * (2 * TROWS * VLD);
    bf16* T2 = T1 + TROWS * VLD;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // one load phase: q / dO rows = the frame's regions + the CLS token as row R; k / v rows = CLS token, regions; rows past the end re-read
    // token 0 (finite; masked below)
    const int f = item % a.F, h = (item / a.F) % a.H, b = item / (a.F * a.H), fR = f * R;
    const int soff = (b * a.N * (int)a.ld + h * HD) * 2, soffo = (b * a.N * (int)a.ldo + h * HD) * 2, soffd = (b * a.N * (int)a.ldd + h * HD) * 2;
    bf16x8 qf[NT][2], gf[NT][2], kf[NT][2], vf[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int i = 16 * t + c;
        const int tq = i < R ? 1 + fR + i : 0, tk = i >= 1 && i <= R ? fR + i : 0;
        qf[t][0] = buf_row16(rq, tq * ldb + 16 * g, soff); qf[t][1] = buf_row16(rq, tq * ldb + 16 * g + 64, soff);
        gf[t][0] = buf_row16(rg, tq * ldob + 16 * g, soffo); gf[t][1] = buf_row16(rg, tq * ldob + 16 * g + 64, soffo);
        kf[t][0] = buf_row16(rk, tk * ldb + 16 * g, soff); kf[t][1] = buf_row16(rk, tk * ldb + 16 * g + 64, soff);
        vf[t][0] = buf_row16(rv, tk * ldb + 16 * g, soff); vf[t][1] = buf_row16(rv, tk * ldb + 16 * g + 64, soff);
    }
    const float mlane = lane <= R ? a.addmask[(int64_t)b * a.N + (lane == 0 ? 0 : fR + lane)] : -INFINITY;      // additive mask of key `lane`
    const float* st3 = stats + ((int64_t)b * a.H + h) * 4;
    const float m_cls = st3[0], il_cls = 1.f / st3[1];
    float D_cls = st3[2];
    if (a.fwd_out) {     // statistics saved by the forward: D = sum_j p_j <dO_cls, v_j> = <dO_cls, O_cls>
        const float pd = (float)((const bf16*)a.fwd_out)[(int64_t)b * a.N * a.ld_fo + h * HD + lane] * (float)((const bf16*)a.dout)[(int64_t)b * a.N * a.ldo + h * HD + lane];
        D_cls = wave_sum(pd);
    }
    put_row_frags<NT, NTP>(T1, gf, lane);                        // dO -> tile 1 (dV pass), Q -> tile 2 (dK pass)
    put_row_frags<NT, NTP>(T2, qf, lane);
    f32x4 s2[NT][NT], dp[NT][NT];
#pragma unroll
    for (int qt = 0; qt < NT; ++qt)
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[qt][0], vf[kt][0], zero4, 0, 0, 0);
            dp[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[qt][1], vf[kt][1], acc, 0, 0, 0);
        }
#pragma unroll
    for (int qt = 0; qt < NT; ++qt)
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[qt][0], kf[kt][0], zero4, 0, 0, 0);
            s2[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[qt][1], kf[kt][1], acc, 0, 0, 0);
        }
    float mk[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) mk[kt] = __shfl(mlane, 16 * kt + c, 64);
#pragma unroll
    for (int qt = 0; qt < NT; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qi = 16 * qt + 4 * g + r;
            const bool qok = qi <= R, is_cls = qi == R;
            float m = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) { s2[qt][kt][r] = s2[qt][kt][r] * a.scale + mk[kt]; m = fmaxf(m, s2[qt][kt][r]); }
            m = row16_max(m);
            if (is_cls) m = m_cls;                                       // the CLS query's softmax spans all N keys
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                float e = __expf(s2[qt][kt][r] - m);
                if (is_cls && kt == 0 && c == 0 && f != 0) e = 0.f;       // CLS query x CLS key: counted once, in frame 0
                s2[qt][kt][r] = e; sum += e;
            }
            sum = row16_sum(sum);
            const float inv = is_cls ? il_cls : (qok ? 1.f / sum : 0.f);      // padded query rows contribute nothing
            float D = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) { s2[qt][kt][r] *= inv; D += s2[qt][kt][r] * dp[qt][kt][r]; }
            D = row16_sum(D);
            if (is_cls) D = D_cls;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) dp[qt][kt][r] = s2[qt][kt][r] * (dp[qt][kt][r] - D);   // dS
        }
    // P and dS as bf16 operands of the key-major products (k index = query: tiles 2 s, 2 s + 1)
    bf16x8 pP[NS][NT], pS[NS][NT];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            const int qt0 = 2 * s, qt1 = 2 * s + 1;
            pP[s][kt] = pack8(s2[qt0][kt], qt1 < NT ? s2[qt1 < NT ? qt1 : qt0][kt] : zero4);
            pS[s][kt] = pack8(dp[qt0][kt], qt1 < NT ? dp[qt1 < NT ? qt1 : qt0][kt] : zero4);
        }
    // dV^T[d][key] = sum_q dO[q][d] P[q][key];  dK^T[d][key] = scale * sum_q Q[q][d] dS[q][key]: acc[kt][dt][r] = row (key 16 kt + c), channel 16 dt + 4 g + r
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        bf16* Ts = pass == 0 ? T1 : T2;
        f32x4 acc[NT][4];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) acc[kt][dt] = zero4;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            bf16x8 xb[4];
            tr_quad(xb, lds_addr(&Ts[(32 * s + 4 * g + qq) * VLD + 4 * pp]), lds_addr(&Ts[(32 * s + 16 + 4 * g + qq) * VLD + 4 * pp]));
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) acc[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb[dt], pass == 0 ? pP[s][kt] : pS[s][kt], acc[kt][dt], 0, 0, 0);
        }
        const float mul = pass == 0 ? 1.f : a.scale;
        if (c == 0) {          // key 0 of the frame = the shared CLS key: fp32 partial for the closing launch
            float* w = a.ws + ((((int64_t)b * a.H + h) * a.F + f) * 2 + (pass == 0 ? 1 : 0)) * HD + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) *(float4*)&w[16 * dt] = make_float4(acc[0][dt][0] * mul, acc[0][dt][1] * mul, acc[0][dt][2] * mul, acc[0][dt][3] * mul);
        }
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) *(u32x2_t*)&Ts[(16 * kt + c) * VLD + 16 * dt + 4 * g] = pack4(acc[kt][dt] * mul);
#pragma unroll
        for (int it = 0; it < NT * 2; ++it) {
            const int row = it * 8 + (lane >> 3), ch = lane & 7;
            if (row >= 1 && row <= R)
                __builtin_amdgcn_raw_buffer_store_b128(*(const u32x4_t*)&Ts[row * VLD + ch * 8], pass == 0 ? rdv : rdk, (fR + row) * lddb + ch * 16, soffd, 0);
        }
        if (a.csum) {
            const Seg sg{0, R, a.N, f, 0, 1};
            tile_colsum<NT>(Ts, sg, true, lane, a.csum + ((int64_t)b * a.F + f) * (3 * a.H * HD) + (pass == 0 ? 2 : 1) * a.H * HD + h * HD);
        }
        if (pass == 0) put_row_frags<NT, NTP>(T1, kf, lane);    // tile 1 is free: K (held in registers so far) moves in, read back transposed for dQ
    }
    // dQ^T[d][q] = scale * sum_key K[key][d] dS[q][key]: K^T from tile 1, dS^T written to tile 2 as [key][query] and read back transposed
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            const u32x4_t w = __builtin_bit_cast(u32x4_t, pS[s][kt]);
            *(u32x2_t*)&T2[(16 * kt + c) * VLD + 32 * s + 4 * g] = (u32x2_t){w[0], w[1]};
            if (2 * s + 1 < NT) *(u32x2_t*)&T2[(16 * kt + c) * VLD + 32 * s + 16 + 4 * g] = (u32x2_t){w[2], w[3]};
        }
    f32x4 dqa[NT][4];
#pragma unroll
    for (int qt = 0; qt < NT; ++qt)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dqa[qt][dt] = zero4;
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) {
        bf16x8 kb[4], dsb[4];
        tr_quad(kb, lds_addr(&T1[(32 * ks + 8 * g + qq) * VLD + 4 * pp]), lds_addr(&T1[(32 * ks + 8 * g + 4 + qq) * VLD + 4 * pp]));
        tr_quad(dsb, lds_addr(&T2[(32 * ks + 8 * g + qq) * VLD + 4 * pp]), lds_addr(&T2[(32 * ks + 8 * g + 4 + qq) * VLD + 4 * pp]));
#pragma unroll
        for (int qt = 0; qt < NT; ++qt)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dqa[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kb[dt], dsb[qt], dqa[qt][dt], 0, 0, 0);
    }
    if (a.dq_ws) {       // dq of the CLS query: this frame's share (summed by attn_bwd_cls_post_kernel)
#pragma unroll
        for (int qt = 0; qt < NT; ++qt)
            if (qt == cq && c == cc) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    *(float4*)&a.dq_ws[(int64_t)item * HD + 16 * dt + 4 * g] =
                        make_float4(dqa[qt][dt][0] * a.scale, dqa[qt][dt][1] * a.scale, dqa[qt][dt][2] * a.scale, dqa[qt][dt][3] * a.scale);
            }
    }
#pragma unroll
    for (int qt = 0; qt < NT; ++qt)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) *(u32x2_t*)&T1[(16 * qt + c) * VLD + 16 * dt + 4 * g] = pack4(dqa[qt][dt] * a.scale);
#pragma unroll
    for (int it = 0; it < NT * 2; ++it) {
        const int row = it * 8 + (lane >> 3), ch = lane & 7;
        if (row < R)
            __builtin_amdgcn_raw_buffer_store_b128(*(const u32x4_t*)&T1[row * VLD + ch * 8], rdq, (1 + fR + row) * lddb + ch * 16, soffd, 0);
    }
    if (a.csum) {
        const Seg sgp{0, R, a.N, f, 0, 0};
        tile_colsum<NT>(T1, sgp, false, lane, a.csum + ((int64_t)b * a.F + f) * (3 * a.H * HD) + h * HD);
    }
}

// launch A of the merged backward: CLS-query statistics and dq_cls.  stats[b][h] = (max, sum, D, -)
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_cls_pre_kernel(AttnArgs a, float* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int h = blockIdx.x, b = blockIdx.y;
    const int64_t brow0 = (int64_t)b * a.N;
    const int N = a.N;
    const T* q = (const T*)a.q; const T* k = (const T*)a.k; const T* v = (const T*)a.v; const T* dout = (const T*)a.dout;
    T* dq = (T*)a.dq;
    float* S = sm; float* DP = S + N; float* red = DP + N;
    constexpr int UG = 5;
    const int kl = lane >> 3, c8 = lane & 7;
    float q8[8], do8[8];
    load8(q + brow0 * a.ld + h * HD + c8 * 8, q8);
    load8(dout + brow0 * a.ldo + h * HD + c8 * 8, do8);
#pragma unroll
    for (int e = 0; e < 8; ++e) q8[e] *= a.scale;
    const int ngroups = (N + 7) / 8;
    for (int g0 = wid; g0 < ngroups; g0 += 4 * UG) {
        float k8[UG][8], v8[UG][8];
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            int j = 8 * (g0 + 4 * u) + kl;
            j = j < N ? j : N - 1;
            load8(k + (brow0 + j) * a.ld + h * HD + c8 * 8, k8[u]);
            load8(v + (brow0 + j) * a.ld + h * HD + c8 * 8, v8[u]);
        }
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            const int j = 8 * (g0 + 4 * u) + kl;
            float sc = 0.f, dp = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { sc += q8[e] * k8[u][e]; dp += do8[e] * v8[u][e]; }
            sc = oct_sum(sc); dp = oct_sum(dp);
            if (c8 == 0 && j < N) { S[j] = sc + a.addmask[brow0 + j]; DP[j] = dp; }
        }
    }
    __syncthreads();
    float m = -INFINITY;
    for (int j = lane; j < N; j += 64) m = fmaxf(m, S[j]);
    m = wave_max(m);
    float l = 0.f, pd = 0.f;
    for (int j = lane; j < N; j += 64) { const float e = expf(S[j] - m); l += e; pd += e * DP[j]; }
    l = wave_sum(l); pd = wave_sum(pd);
    const float inv = 1.f / l, Dsum = pd * inv;
    if (threadIdx.x == 0) { float* s3 = stats + ((int64_t)b * a.H + h) * 4; s3[0] = m; s3[1] = l; s3[2] = Dsum; s3[3] = 0.f; }
    float dq8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int g0 = wid; g0 < ngroups; g0 += 4 * UG) {
        float k8[UG][8];
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            int j = 8 * (g0 + 4 * u) + kl;
            j = j < N ? j : N - 1;
            load8(k + (brow0 + j) * a.ld + h * HD + c8 * 8, k8[u]);
        }
#pragma unroll
        for (int u = 0; u < UG; ++u) {
            const int j = 8 * (g0 + 4 * u) + kl;
            if (j < N) {
                const float ds = expf(S[j] - m) * inv * (DP[j] - Dsum);
#pragma unroll
                for (int e = 0; e < 8; ++e) dq8[e] += ds * k8[u][e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) dq8[e] = stride8_sum(dq8[e]);
    if (kl == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[wid * 64 + c8 * 8 + e] = dq8[e];
    }
    __syncthreads();
    if (wid == 0) dq[brow0 * a.ldd + h * HD + lane] = from_f<T>((red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane]) * a.scale);
}

// launch C of the merged backward: dK / dV of the shared CLS key = sum of the per-frame partials
template <typename T>
__global__ __launch_bounds__(64) void attn_bwd_cls_post_kernel(AttnArgs a) {
    const int lane = threadIdx.x, h = blockIdx.x, b = blockIdx.y;
    const float* w = a.ws + (((int64_t)b * a.H + h) * a.F) * 2 * HD;
    float gk = 0.f, gv = 0.f;
    for (int f = 0; f < a.F; ++f) { gk += w[f * 2 * HD + lane]; gv += w[f * 2 * HD + HD + lane]; }
    const int64_t off = (int64_t)b * a.N * a.ldd + h * HD + lane;
    ((T*)a.dk)[off] = from_f<T>(gk);
    ((T*)a.dv)[off] = from_f<T>(gv);
    float gq = 0.f;
    if (a.dq_ws) {
        const float* wq = a.dq_ws + (((int64_t)b * a.H + h) * a.F) * HD;
        for (int f = 0; f < a.F; ++f) gq += wq[f * HD + lane];
        ((T*)a.dq)[off] = from_f<T>(gq);
    }
    if (a.csum) {        // the CLS token's rows of dq | dk | dv (as stored): partial row B*F + b
        float* cr = a.csum + ((int64_t)a.B * a.F + b) * (3 * a.H * HD) + h * HD + lane;
        cr[0] = to_f(from_f<T>(gq)); cr[a.H * HD] = to_f(from_f<T>(gk)); cr[2 * a.H * HD] = to_f(from_f<T>(gv));
    }
}

// backward, full (text) mode: one workgroup per (b, h); K, Q and dO tiles are staged once in LDS and shared; wave w owns
// query tiles {2w, 2w+1} for dQ (layout 1) and key tiles {2w, 2w+1} for dK / dV (layout 2).  NT = ceil(L / 16) <= 8.
template <int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void mattn_bwd_full_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smraw[];
    constexpr int NTP = (NT + 1) & ~1, ROWS = 16 * NTP;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, g = lane >> 4, c = lane & 15, qq = (lane >> 2) & 3, pp = lane & 3;
    const int h = blockIdx.x, b = blockIdx.y;
    Seg sg{1, a.R, a.N, 0, 0};
    const int64_t brow0 = (int64_t)b * a.N;
    const bf16* q = (const bf16*)a.q; const bf16* k = (const bf16*)a.k; const bf16* v = (const bf16*)a.v; const bf16* dout = (const bf16*)a.dout;
    bf16* dq = (bf16*)a.dq; bf16* dk = (bf16*)a.dk; bf16* dv = (bf16*)a.dv;
    bf16* Ks = (bf16*)smraw; bf16* Qs = Ks + ROWS * VLD; bf16* Gs = Qs + ROWS * VLD; bf16* Vs = Gs + ROWS * VLD;
    float* stats = (float*)(Vs + ROWS * VLD);      // [ROWS][3]: row max, 1/row sum, D = sum_k P dP  (written by layout 1)
    // cooperative staging: wave w copies rows [w*ROWS/4, (w+1)*ROWS/4) of each tile
    {
        const Seg all = sg;
#pragma unroll
        for (int it = 0; it < ROWS / 32; ++it) {
            const int row = wid * (ROWS / 4) + it * 8 + (lane >> 3), ch = lane & 7;
            const int tok = all.tok_k(row);
            uint4 kv = make_uint4(0u, 0u, 0u, 0u), qv = kv, gv = kv, vv = kv;
            if (tok >= 0) {
                kv = *(const uint4*)(k + (brow0 + tok) * a.ld + h * HD + ch * 8);
                qv = *(const uint4*)(q + (brow0 + tok) * a.ld + h * HD + ch * 8);
                gv = *(const uint4*)(dout + (brow0 + tok) * a.ldo + h * HD + ch * 8);
                vv = *(const uint4*)(v + (brow0 + tok) * a.ld + h * HD + ch * 8);
            }
            *(uint4*)&Ks[row * VLD + ch * 8] = kv; *(uint4*)&Qs[row * VLD + ch * 8] = qv; *(uint4*)&Gs[row * VLD + ch * 8] = gv;
            *(uint4*)&Vs[row * VLD + ch * 8] = vv;
        }
    }
    __syncthreads();
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // row fragments of tile t (16 rows, both k-steps) straight from the LDS tiles (row-major, 16-byte reads)
    auto rowfrag = [&](const bf16* T, int t, int ks) { return *(const bf16x8*)&T[(16 * t + c) * VLD + 32 * ks + 8 * g]; };
    // (V is staged like the others: fetching its fragments from global memory inside the tile loops put a memory round trip
    //  in front of every MFMA group)
    auto vfrag = [&](int t, int ks) { return rowfrag(Vs, t, ks); };
    // ---------------- layout 1: this wave's query tiles -> dQ ----------------
#pragma unroll 1
    for (int qi = 0; qi < 2; ++qi) {
        const int qt = 2 * wid + qi;
        if (qt >= NT) break;
        const bf16x8 q0 = rowfrag(Qs, qt, 0), q1 = rowfrag(Qs, qt, 1), g0 = rowfrag(Gs, qt, 0), g1 = rowfrag(Gs, qt, 1);
        f32x4 st[NT], dp[NT];
        // keep bytes of (query 16 qt + c, keys 16 kt + 4 g .. + 3): fetched ahead of the products that hide the round trip, and held
        // as packed words (float multipliers here cost a wave of occupancy)
        uint32_t kw[NT];
        if (a.keep) {
            int qi = 16 * qt + c;
            qi = qi < a.N ? qi : a.N - 1;
            const uint8_t* kr = a.keep + (((int64_t)b * a.H + h) * a.Ns + qi) * a.Ns;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) kw[kt] = *(const uint32_t*)(kr + 16 * kt + 4 * g);
        } else {
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) kw[kt] = 0x01010101u;
        }
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowfrag(Ks, kt, 0), q0, zero4, 0, 0, 0);
            st[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowfrag(Ks, kt, 1), q1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfrag(kt, 0), g0, zero4, 0, 0, 0);
            dp[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfrag(kt, 1), g1, acc, 0, 0, 0);
        }
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tok = sg.tok_k(16 * kt + 4 * g + r);
                st[kt][r] = st[kt][r] * a.scale + (tok >= 0 ? a.addmask[brow0 + tok] : -INFINITY);
                m = fmaxf(m, st[kt][r]);
            }
        m = col4_max(m);
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { st[kt][r] = __expf(st[kt][r] - m); sum += st[kt][r]; }
        sum = col4_sum(sum);
        const float inv = 1.f / sum;
        float D = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dp[kt][r] *= ((kw[kt] >> (8 * r)) & 1u) ? a.kscale : 0.f;   // d softmax = keep scale * d(dropped weights); 1 when off
                st[kt][r] *= inv; D += st[kt][r] * dp[kt][r];
            }
        D = col4_sum(D);
        if (g == 0) { float* sp = stats + (16 * qt + c) * 3; sp[0] = m; sp[1] = inv; sp[2] = D; }
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[kt][r] = st[kt][r] * (dp[kt][r] - D);
        f32x4 acc[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
        for (int s = 0; s < NTP / 2; ++s) {
            const int kt0 = 2 * s, kt1 = 2 * s + 1;
            const bf16x8 da = pack8(st[kt0], kt1 < NT ? st[kt1 < NT ? kt1 : kt0] : zero4);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 kb = tr_pair(lds_addr(&Ks[(16 * kt0 + 4 * g + qq) * VLD + 16 * dt + 4 * pp]),
                                          lds_addr(&Ks[(16 * kt1 + 4 * g + qq) * VLD + 16 * dt + 4 * pp]));
                acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, kb, acc[dt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tok = sg.tok_q(16 * qt + 4 * g + r);
            if (tok >= 0) {
                bf16* orow = dq + (brow0 + tok) * a.ldd + h * HD + c;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) orow[16 * dt] = (bf16)(acc[dt][r] * a.scale);
            }
        }
    }
    __syncthreads();
    // ---------------- layout 2: this wave's key tiles -> dV, dK ----------------
#pragma unroll 1
    for (int ki = 0; ki < 2; ++ki) {
        const int kt = 2 * wid + ki;
        if (kt >= NT) break;
        const bf16x8 k0 = rowfrag(Ks, kt, 0), k1 = rowfrag(Ks, kt, 1), v0 = vfrag(kt, 0), v1 = vfrag(kt, 1);
        const int ktok = sg.tok_k(16 * kt + c);
        const float mk = ktok >= 0 ? a.addmask[brow0 + ktok] : -INFINITY;
        f32x4 s2[NT], dp[NT];
        uint32_t kw[NT];                                         // keepT bytes of (key 16 kt + c, queries 16 qt + 4 g .. + 3)
        if (a.keepT) {
            int kk = 16 * kt + c;
            kk = kk < a.N ? kk : a.N - 1;
            const uint8_t* kr = a.keepT + (((int64_t)b * a.H + h) * a.Ns + kk) * a.Ns;
#pragma unroll
            for (int qt = 0; qt < NT; ++qt) kw[qt] = *(const uint32_t*)(kr + 16 * qt + 4 * g);
        } else {
#pragma unroll
            for (int qt = 0; qt < NT; ++qt) kw[qt] = 0x01010101u;
        }
#pragma unroll
        for (int qt = 0; qt < NT; ++qt) {
            f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowfrag(Qs, qt, 0), k0, zero4, 0, 0, 0);
            s2[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowfrag(Qs, qt, 1), k1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowfrag(Gs, qt, 0), v0, zero4, 0, 0, 0);
            dp[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowfrag(Gs, qt, 1), v1, acc, 0, 0, 0);
        }
        // row statistics of query q = 16 qt + 4 g + r (over ALL keys) come from layout 1 through LDS
#pragma unroll
        for (int qt = 0; qt < NT; ++qt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qi = 16 * qt + 4 * g + r;
                const bool qok = sg.tok_q(qi) >= 0;
                const float* sp = stats + qi * 3;
                const float p = qok ? __expf(s2[qt][r] * a.scale + mk - sp[0]) * sp[1] : 0.f;
                const float km = ((kw[qt] >> (8 * r)) & 1u) ? a.kscale : 0.f;   // dropout multiplier, 1 when off
                dp[qt][r] = p * (dp[qt][r] * km - sp[2]);        // dS[q][key]
                s2[qt][r] = p * km;                              // dV takes the dropped weights
            }
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const bf16* Ts = pass == 0 ? Gs : Qs;
            f32x4 acc[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
            for (int s = 0; s < NTP / 2; ++s) {
                const int qt0 = 2 * s, qt1 = 2 * s + 1;
                const f32x4& lo = pass == 0 ? s2[qt0] : dp[qt0];
                const f32x4& hi = qt1 < NT ? (pass == 0 ? s2[qt1 < NT ? qt1 : qt0] : dp[qt1 < NT ? qt1 : qt0]) : zero4;
                const bf16x8 pa = pack8(lo, hi);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const bf16x8 xb = tr_pair(lds_addr(&Ts[(16 * qt0 + 4 * g + qq) * VLD + 16 * dt + 4 * pp]),
                                              lds_addr(&Ts[(16 * qt1 + 4 * g + qq) * VLD + 16 * dt + 4 * pp]));
                    acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, xb, acc[dt], 0, 0, 0);
                }
            }
            const float mul = pass == 0 ? 1.f : a.scale;
            bf16* dst = pass == 0 ? dv : dk;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tok = sg.tok_k(16 * kt + 4 * g + r);
                if (tok >= 0) {
                    bf16* orow = dst + (brow0 + tok) * a.ldd + h * HD + c;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) orow[16 * dt] = (bf16)(acc[dt][r] * mul);
                }
            }
        }
    }
}

static int attn_check(const AttnArgs& a) {
    if (a.B <= 0 || a.H <= 0 || a.N <= 0) return DVLP_ERR_SHAPE;
    if (a.mode == 0) { if (a.N != 1 + a.F * a.R || a.R + 1 > KMAX) return DVLP_ERR_SHAPE; }
    else if (a.mode == 1) { if (a.R != a.N || a.N > KMAX) return DVLP_ERR_SHAPE; }
    else return DVLP_ERR_UNSUPPORTED;
    return DVLP_OK;
}

float* dvlp_rd_reserve_push(int64_t P, int64_t C, float* out);      // norm.hip: deferred-reduction queue
// (dvlp_attn_ext::colsum: the backward (mode 0) also queues the column sums of dq | dk | dv ([3*H*64] fp32: the packed qkv bias gradient)
//  when it can -- bf16 one-pass form with forward statistics, deferred reductions enabled -- and says so in colsum_fused)
static int g_attn_fold = 1;        // space-mode bf16: 1 = CLS query folded into the frame waves when the caller passes workspaces, 0 = separate launches (A/B, tests)
DVLP_DEV_API int dvlp_dev_attention_cls_fold(int on) { g_attn_fold = on; return DVLP_OK; }

// `workspace` (B*H*F*66 floats) + `cls_stats` (B*H*4 floats), both optional: space mode, bf16 -- the CLS query is folded into the
// frame waves and its global softmax statistics are kept in `cls_stats` for dvlp_attention_bwd.  ext->folded says whether that
// happened (shape outside the fold's range, fp32, or the fold switched off: `cls_stats` is left untouched).
extern "C" int dvlp_attention_fwd_ex(int dtype, int mode, int64_t B, int64_t N, int64_t H, int64_t F, int64_t R, const void* q, const void* k,
                                     const void* v, int64_t ld, const float* addmask, void* out, int64_t ldo, float scale, float* workspace,
                                     float* cls_stats, dvlp_attn_ext* ext, void* stream) {
    dvlp_clear_status();
    if (ext) ext->folded = 0;
    AttnArgs a{};
    a.abl = g_attn_abl;
    a.q = q; a.k = k; a.v = v; a.ld = ld; a.addmask = addmask; a.out = out; a.ldo = ldo;
    a.B = (int)B; a.N = (int)N; a.H = (int)H; a.F = (int)F; a.R = (int)R; a.mode = mode; a.scale = scale;
    a.keep = (mode == 1 && ext) ? (const uint8_t*)ext->keep : nullptr; a.keepT = (mode == 1 && ext) ? (const uint8_t*)ext->keepT : nullptr;
    a.kscale = (mode == 1 && ext && ext->keep) ? ext->keep_scale : 1.f; a.Ns = (int)((N + 15) / 16 * 16);
    if (int rc = attn_check(a)) return rc;
    const int nseg = mode == 0 ? (int)F : 1, nk = (int)R + (mode == 0 ? 1 : 0);
    size_t lds = (size_t)(2 * nk * KP + KMAX) * sizeof(float);
    if (mode == 0 && lds < (size_t)(N + 256) * sizeof(float)) lds = (size_t)(N + 256) * sizeof(float);     // CLS workgroup: scores [N] + partial outputs
    dim3 grid((unsigned)(nseg + (mode == 0 ? 1 : 0)), (unsigned)H, (unsigned)B), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DVLP_F32) {
        static bool once = false; if (!once) { once = true; (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
        hipLaunchKernelGGL(attn_fwd_kernel<float>, grid, block, lds, st, a);
    } else if (dtype == DVLP_BF16) {
        static bool once = false; if (!once) { once = true; (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
        if (ld % 8 || ldo % 8) return DVLP_ERR_SHAPE;
        // MFMA path for the frame / full segments; the single CLS query per (b, h) stays on the streaming VALU workgroup
        bool done = false;
#define MFWD(NQT_, NKT_, ITEMS, QG) do { const int items_ = (int)(ITEMS); const size_t l_ = (size_t)4 * (((NKT_ + 1) & ~1) * 16 * VLD) * sizeof(bf16); \
            constexpr int wpe_ = NKT_ <= 3 ? 4 : 2; \
            (void)hipFuncSetAttribute((const void*)mattn_fwd_kernel<NQT_, NKT_, wpe_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            hipLaunchKernelGGL((mattn_fwd_kernel<NQT_, NKT_, wpe_>), dim3((unsigned)cdiv(items_, 4)), block, l_, st, a, items_, (int)(QG)); done = true; } while (0)
        if (mode == 0) {
            const int nqt = (int)cdiv(R, 16), nkt = (int)cdiv(R + 1, 16);
            // fold: the CLS row needs a free query slot in the frame tile (R + 1 <= 16 nqt)
            if (ext && workspace && cls_stats && g_attn_fold && nqt == nkt && nqt <= 3) {     // (ext: the caller must be able to learn that it happened)
                a.cls_o = workspace; a.cls_st = workspace + B * H * F * HD; a.cls_stats = cls_stats;
            }
            const bool small = (int64_t)B * N * (ld > ldo ? ld : ldo) * 2 < (int64_t)0x7fffff00;       // 32-bit byte offsets inside a tensor
            if (g_attn_lean && a.cls_o && small) {          // (the fold guarantees nqt == nkt <= 3; of the ablation bits this kernel knows only 8)
                const size_t l_ = (size_t)(SATTN_WAVES * nkt * 16 * VLD + 16 * VLD) * sizeof(bf16) + (size_t)SATTN_WAVES * (HD + 2) * sizeof(float);
                const unsigned chunks = (unsigned)cdiv(F, SATTN_WAVES);
#define SFWD(NT_) hipLaunchKernelGGL((sattn_fwd_kernel<NT_>), dim3((unsigned)(B * H), chunks), dim3(64 * SATTN_WAVES), l_, st, a)
                if (nkt == 3) SFWD(3); else if (nkt == 2) SFWD(2); else SFWD(1);
#undef SFWD
                // <= 8 frames: merged inside the kernel; more: the combine launch over the per-frame partials
                if (chunks > 1) hipLaunchKernelGGL(attn_fwd_cls_combine_kernel, dim3((unsigned)H, (unsigned)B), dim3(64), 0, st, a);
                if (ext) ext->folded = 1;
                return dvlp_launch_status();
            }
            else if (nqt == 3 && nkt == 3) MFWD(3, 3, B * H * F, 1);
            else if (nqt == 2 && nkt == 2) MFWD(2, 2, B * H * F, 1);
            else if (nqt == 1 && nkt == 1) MFWD(1, 1, B * H * F, 1);
            else if (nqt == 1 && nkt == 2) MFWD(1, 2, B * H * F, 1);
            else if (nqt == 2 && nkt == 3) MFWD(2, 3, B * H * F, 1);
            if (done && a.cls_o) { hipLaunchKernelGGL(attn_fwd_cls_combine_kernel, dim3((unsigned)H, (unsigned)B), dim3(64), 0, st, a); if (ext) ext->folded = 1; }
            else if (done) {   // CLS query on the streaming VALU workgroup
                AttnArgs c = a; c.seg_begin = nseg;
                hipLaunchKernelGGL(attn_fwd_kernel<bf16>, dim3(1, (unsigned)H, (unsigned)B), block, lds, st, c);
            }
        } else {
            const int nkt = (int)cdiv(N, 16);
            if (nkt == 7) MFWD(2, 7, B * H * 4, 4);
            else if (nkt == 8) MFWD(2, 8, B * H * 4, 4);
            else if (nkt <= 3) MFWD(3, 3, B * H, 1);
        }
#undef MFWD
        if (!done) hipLaunchKernelGGL(attn_fwd_kernel<bf16>, grid, block, lds, st, a);
    } else return DVLP_ERR_DTYPE;
    return dvlp_launch_status();
}

// workspace (space mode only): fp32 [B*H*(F*3*64 + 4)]; `fwd_out` / `cls_stats`: the forward's output and the CLS statistics it saved
// (dvlp_attention_fwd with workspaces), both optional
extern "C" int dvlp_attention_bwd_ex(int dtype, int mode, int64_t B, int64_t N, int64_t H, int64_t F, int64_t R, const void* q, const void* k,
                                     const void* v, int64_t ld, const float* addmask, const void* dout, int64_t ldo, void* dq, void* dk,
                                     void* dv, int64_t ldd, float* workspace, float scale, const void* fwd_out, int64_t ld_fwd_out,
                                     const float* cls_stats, dvlp_attn_ext* ext, void* stream) {
    dvlp_clear_status();
    AttnArgs a{};
    a.abl = g_attn_abl;
    a.q = q; a.k = k; a.v = v; a.ld = ld; a.addmask = addmask; a.dout = dout; a.ldo = ldo; a.dq = dq; a.dk = dk; a.dv = dv; a.ldd = ldd;
    a.ws = workspace;
    a.B = (int)B; a.N = (int)N; a.H = (int)H; a.F = (int)F; a.R = (int)R; a.mode = mode; a.scale = scale;
    a.keep = (mode == 1 && ext) ? (const uint8_t*)ext->keep : nullptr; a.keepT = (mode == 1 && ext) ? (const uint8_t*)ext->keepT : nullptr;
    a.kscale = (mode == 1 && ext && ext->keep) ? ext->keep_scale : 1.f; a.Ns = (int)((N + 15) / 16 * 16);
    float* const csum_dst = ext ? ext->colsum : nullptr;
    if (ext) ext->colsum_fused = 0;
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
        bool done = false;
        // one-pass form (CLS query rides along in the frame tiles): needs R + 1 query rows in <= 3 tiles
        // (timing ablations 1 / 2 / 4 exist in the three-launch kernel only: a run that sets them must not silently time an unablated one-pass kernel)
        if (mode == 0 && ld % 8 == 0 && ldo % 8 == 0 && ldd % 8 == 0 && R + 1 <= 48 && g_attn_merged && !(a.abl & 7)) {
            const int nt = (int)cdiv(R + 1, 16), items = (int)(B * H * F);
            float* stats = workspace + B * H * F * 2 * HD;
            if (fwd_out && cls_stats && g_attn_fold) {      // statistics from the folded forward: no statistics pass, dq_cls from per-frame partials
                stats = const_cast<float*>(cls_stats);
                a.fwd_out = fwd_out; a.ld_fo = ld_fwd_out; a.dq_ws = workspace + B * H * F * 2 * HD + B * H * 4;
                if (csum_dst) {      // the CLS rows reach the partial buffer through the closing launch, which needs dq_cls: this form only
                    a.csum = dvlp_rd_reserve_push(B * F + B, 3 * H * HD, csum_dst);
                    if (ext) ext->colsum_fused = a.csum != nullptr;
                }
            } else hipLaunchKernelGGL(attn_bwd_cls_pre_kernel<bf16>, grid2, block, lds2, st, a, stats);
#define MMRG(NT_) do { const size_t l_ = (size_t)4 * 2 * 16 * ((NT_ + 1) & ~1) * VLD * sizeof(bf16); \
                (void)hipFuncSetAttribute((const void*)mattn_bwd_space_merged_kernel<NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                hipLaunchKernelGGL((mattn_bwd_space_merged_kernel<NT_>), dim3((unsigned)cdiv(items, 4)), block, l_, st, a, items, (const float*)stats); } while (0)
#define SBWD(NT_) do { const size_t l_ = (size_t)4 * 2 * 16 * ((NT_ + 1) & ~1) * VLD * sizeof(bf16); \
                (void)hipFuncSetAttribute((const void*)sattn_bwd_kernel<NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                hipLaunchKernelGGL((sattn_bwd_kernel<NT_>), dim3((unsigned)cdiv(items, 4)), block, l_, st, a, items, (const float*)stats); } while (0)
            const int64_t ldmax = ld > ldd ? (ld > ldo ? ld : ldo) : (ldd > ldo ? ldd : ldo);
            const bool small = (int64_t)B * N * ldmax * 2 < (int64_t)0x7fffff00;       // 32-bit byte offsets inside a tensor
            if (g_attn_lean && a.fwd_out && small) { if (nt == 3) SBWD(3); else if (nt == 2) SBWD(2); else SBWD(1); }
            else if (nt == 3) MMRG(3); else if (nt == 2) MMRG(2); else MMRG(1);
#undef SBWD
#undef MMRG
            hipLaunchKernelGGL(attn_bwd_cls_post_kernel<bf16>, grid2, dim3(64), 0, st, a);
            return dvlp_launch_status();
        }
        if (mode == 0 && ld % 8 == 0 && ldo % 8 == 0 && ldd % 8 == 0) {
            const int nqt = (int)cdiv(R, 16), nkt = (int)cdiv(R + 1, 16);
            const int items = (int)(B * H * F);
#define MBWD(NQT_, NKT_) do { const int tr_ = 16 * ((((NQT_ + 1) & ~1) > ((NKT_ + 1) & ~1)) ? ((NQT_ + 1) & ~1) : ((NKT_ + 1) & ~1)); \
            hipLaunchKernelGGL((mattn_bwd_space_kernel<NQT_, NKT_, 0>), dim3((unsigned)cdiv(items, 4)), block, (size_t)4 * tr_ * VLD * sizeof(bf16), st, a, items); \
            hipLaunchKernelGGL((mattn_bwd_space_kernel<NQT_, NKT_, 1>), dim3((unsigned)cdiv(items, 4)), block, (size_t)4 * tr_ * VLD * sizeof(bf16), st, a, items); done = true; } while (0)
            if (nqt == 3 && nkt == 3) MBWD(3, 3);
            else if (nqt == 2 && nkt == 2) MBWD(2, 2);
            else if (nqt == 1 && nkt == 1) MBWD(1, 1);
            else if (nqt == 1 && nkt == 2) MBWD(1, 2);
            else if (nqt == 2 && nkt == 3) MBWD(2, 3);
#undef MBWD
        }
        if (mode == 1 && ld % 8 == 0 && ldo % 8 == 0 && ldd % 8 == 0) {
            const int nt = (int)cdiv(N, 16);
#define MFULL(NT_) do { const size_t l_ = (size_t)4 * 16 * ((NT_ + 1) & ~1) * VLD * sizeof(bf16) + (size_t)16 * ((NT_ + 1) & ~1) * 3 * sizeof(float); \
            (void)hipFuncSetAttribute((const void*)mattn_bwd_full_kernel<NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            hipLaunchKernelGGL((mattn_bwd_full_kernel<NT_>), dim3((unsigned)H, (unsigned)B), block, l_, st, a); done = true; } while (0)
            if (nt == 7) MFULL(7);
            else if (nt == 8) MFULL(8);
            else if (nt == 3) MFULL(3);
#undef MFULL
        }
        if (!done) hipLaunchKernelGGL(attn_bwd_seg_kernel<bf16>, grid, block, lds, st, a);
        if (mode == 0) hipLaunchKernelGGL(attn_bwd_cls_kernel<bf16>, grid2, block, lds2, st, a);
    } else return DVLP_ERR_DTYPE;
    return dvlp_launch_status();
}

// the plain forms: no extras
extern "C" int dvlp_attention_fwd(int dtype, int mode, int64_t B, int64_t N, int64_t H, int64_t F, int64_t R, const void* q, const void* k,
                                  const void* v, int64_t ld, const float* addmask, void* out, int64_t ldo, float scale, float* workspace,
                                  float* cls_stats, void* stream) {
    // without an extension struct the caller cannot learn whether the fold happened: the plain form never folds (cls_stats untouched)
    return dvlp_attention_fwd_ex(dtype, mode, B, N, H, F, R, q, k, v, ld, addmask, out, ldo, scale, workspace, cls_stats, nullptr, stream);
}
extern "C" int dvlp_attention_bwd(int dtype, int mode, int64_t B, int64_t N, int64_t H, int64_t F, int64_t R, const void* q, const void* k,
                                  const void* v, int64_t ld, const float* addmask, const void* dout, int64_t ldo, void* dq, void* dk,
                                  void* dv, int64_t ldd, float* workspace, float scale, const void* fwd_out, int64_t ld_fwd_out,
                                  const float* cls_stats, void* stream) {
    // the plain forward never folds the CLS query (it cannot report whether `cls_stats` were written), so statistics handed to the plain
    // backward cannot have come from it: take the two-pass form, which recomputes them, whatever the caller passed
    (void)fwd_out; (void)ld_fwd_out; (void)cls_stats;
    return dvlp_attention_bwd_ex(dtype, mode, B, N, H, F, R, q, k, v, ld, addmask, dout, ldo, dq, dk, dv, ldd, workspace, scale, nullptr, 0, nullptr,
                                 nullptr, stream);
}
