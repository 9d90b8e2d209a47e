// Fused per-pair local-loss kernels (K11 "xattn_pair_fused" of SURVEY.md section 2.2; model/loss.py:209-330 restated per
// section 8(a) row A10), bf16 MFMA path.  ONE workgroup per (video i, caption j) pair keeps everything on chip:
//
//   S = LeakyReLU(C^ Q^T)            MFMA, rows of S owned by waves (accumulators never leave registers)
//   both focal softmaxes               VALU on the accumulators; the column direction exchanges 3 x [8][W] floats through LDS
//   P1' [W][G], P2' [G][W]             written ONCE to LDS as bf16 -- the B operands of the two context products
//   wc^T = C^^T P1'^T, wc2^T = Q^^T P2'^T   MFMA, waves own 32-channel slices of d; only |wc|^2, |wc2|^2 are kept
//   cosines                            <raw, ctx> = (|raw| + eps) * sum P' S_pre  comes straight from the softmax pass
//                                      (C = C^ (|C| + eps)), so the raw embeddings are never re-read
//
// and writes one scalar.  The round-1 structure (xattn.hip; still used for fp32 parity runs and for F*R > 288) streamed
// S, P1, P2, wc, wc2 (+ their gradients) through HBM between 9 batched GEMMs and 6 VALU kernels: ~2 GB of workspace and
// ~4.5 ms per step at B = 64; this keeps O(B (G + W) d) operands + O(B^2) results.
//
// The backward recomputes S and the softmaxes from the same operands (nothing but the inputs is saved), then runs the six
// gradient products per pair on chip; the per-image / per-caption accumulations go through fp32 atomics in HBM
// (dC^_i: 64 adders per address, dQ^_j: 64 adders).
//
// MFMA layouts (v_mfma_f32_16x16x32_bf16): lane l holds A[row l&15][k = 8 (l>>4) + t], B[k = 8 (l>>4) + t][col l&15], t < 8;
// D[row 4 (l>>4) + r][col l&15], r < 4.
#include "common.h"

constexpr int FD = 256;          // projection_dim
constexpr int FW_MAXB = 7;       // word blocks of 16 (W <= 112)
constexpr int FG_SLOTS = 3;      // region blocks of 16 per wave (8 waves: G <= 384 by registers, <= 288 by LDS)
constexpr int FQS = FD + 8;      // LDS row stride of the staged Q^ (elements): 528 B rows -> conflict-free 16-byte fragment reads

struct FusedArgs {
    const bf16 *chat, *chatT, *qhat, *qhatT;   // [Bi][Gr][256], [Bi][256][Gk], [Bj][Wp][256], [Bj][256][Wk]
    const float *nc, *nq;                      // |C_raw| [Bi][G], |Q_raw| [Bj][W]
    const float *mimg, *mcap;                  // additive masks [Bi][G], [Bj][W]
    float* scores;                             // [Bi][Bj]
    int Bi, Bj, G, W, Gr, Gk, Wp, Wk;
    float lam;
    int gate;
    int stop;                                  // TIMING-ONLY ablation: stop after phase n (0 = run everything)
};

typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
// two fp32 values as one register of bf16 (round to nearest even: v_cvt_pk_bf16_f32) and back
__device__ __forceinline__ unsigned pk2(float lo, float hi) { bf16x2 p; p[0] = (bf16)lo; p[1] = (bf16)hi; return __builtin_bit_cast(unsigned, p); }
__device__ __forceinline__ float pk_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float pk_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// hat rows (x / (|x| + 1e-8), loss.py:333-338) in both orientations + |x|.  One wave per padded row; pad rows / pad
// columns are written as zeros so the MFMA operands can be read without bounds checks.
__global__ __launch_bounds__(256) void xf_prep_kernel(int outer, int inner, int rows_p, int cols_p, const bf16* __restrict__ raw,
                                                      bf16* __restrict__ hat, bf16* __restrict__ hatT, float* __restrict__ nrm) {
    const int lane = threadIdx.x & 63;
    const int rmax = rows_p > cols_p ? rows_p : cols_p;
    const int64_t idx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= (int64_t)outer * rmax) return;
    const int o = (int)(idx / rmax), r = (int)(idx % rmax);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < inner) {
        const bf16x4 x = *(const bf16x4*)(raw + ((int64_t)o * inner + r) * FD + lane * 4);
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = (float)x[t];
        const float n = sqrtf(wave_sum(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]));
        if (lane == 0) nrm[(int64_t)o * inner + r] = n;
        const float inv = 1.f / (n + 1e-8f);
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] *= inv;
    }
    bf16x4 h;
#pragma unroll
    for (int t = 0; t < 4; ++t) h[t] = (bf16)v[t];
    if (r < rows_p) *(bf16x4*)(hat + ((int64_t)o * rows_p + r) * FD + lane * 4) = h;
    if (r < cols_p) {
#pragma unroll
        for (int t = 0; t < 4; ++t) hatT[((int64_t)o * FD + lane * 4 + t) * cols_p + r] = h[t];
    }
}

// LDS map of the forward kernel (bytes)
struct FLds { int qs, p2, p1, xa, xb, xc, xd, dotg, dotw, nw1, nw2, total; };
static __host__ __device__ inline FLds flds(int Gr, int Gk, int Wp, int Wk) {
    FLds L;
    const int p2b = Gr * (Wk + 8) * 2, qsb = Wp * FQS * 2, p1b = Wp * (Gk + 8) * 2;
    L.qs = 0; L.p2 = 0;                                   // Q^ is dead (barrier) before P2' is written over it
    L.p1 = (p2b > qsb ? p2b : qsb);
    L.p1 = (L.p1 + 255) & ~255;
    int o = L.p1 + ((p1b + 255) & ~255);
    L.xa = o; o += 8 * 112 * 4; L.xb = o; o += 8 * 112 * 4; L.xc = o; o += 8 * 112 * 4; L.xd = o; o += 8 * 112 * 4;
    L.dotg = o; o += 384 * 4; L.dotw = o; o += 112 * 4; L.nw1 = o; o += 112 * 4; L.nw2 = o; o += 384 * 4;
    L.total = o;
    return L;
}

// sum over the 8 waves' partials of column c (c = 16 nb + (lane & 15)) for every word block
__device__ __forceinline__ void xch_put(float* x, int wid, int lane, const float (&v)[FW_MAXB], int nwb) {
    if ((lane >> 4) == 0) {
#pragma unroll
        for (int nb = 0; nb < FW_MAXB; ++nb) if (nb < nwb) x[wid * 112 + nb * 16 + lane] = v[nb];
    }
}
__device__ __forceinline__ void xch_get(const float* x, int lane, float (&v)[FW_MAXB], int nwb) {
#pragma unroll
    for (int nb = 0; nb < FW_MAXB; ++nb) {
        float t = 0.f;
        if (nb < nwb) {
#pragma unroll
            for (int w = 0; w < 8; ++w) t += x[w * 112 + nb * 16 + (lane & 15)];
        }
        v[nb] = t;
    }
}

// S tiles of this wave's region blocks and both focal softmaxes.
// On return: P1' is in LDS [w][g] (row stride Gk + 8), P2' in LDS [g][w] (row stride Wk + 8), dotg[g] = sum_w P2' S_pre,
// dotw[w] = sum_g P1' S_pre.  S (post-LeakyReLU), rn, cn stay in R for the caller.
struct PairRegs {
    f32x4 S[FG_SLOTS][FW_MAXB];     // LeakyReLU(S_pre)
    float rn[FG_SLOTS][4];          // 1 / (|S_row| + eps)
    float cn[FW_MAXB];              // 1 / (|S_col| + eps)   (column 16 nb + (lane & 15))
};

__device__ __forceinline__ void pair_forward(const FusedArgs& a, char* smem, int i, int j, PairRegs& R) {
    unsigned E[FG_SLOTS][FW_MAXB][2];   // softmax numerators of the pass in flight as packed bf16 pairs (they end as bf16 MFMA operands)
    const int tid = threadIdx.x, lane = tid & 63, lq = lane >> 4, lc = lane & 15;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbg = a.Gr >> 4, nwb = a.Wp >> 4;
    const FLds L = flds(a.Gr, a.Gk, a.Wp, a.Wk);
    bf16* Qs = (bf16*)(smem + L.qs);
    bf16* P1s = (bf16*)(smem + L.p1);
    bf16* P2s = (bf16*)(smem + L.p2);
    float *xa = (float*)(smem + L.xa), *xb = (float*)(smem + L.xb), *xc = (float*)(smem + L.xc), *xd = (float*)(smem + L.xd);
    float *dotg = (float*)(smem + L.dotg), *dotw = (float*)(smem + L.dotw);
    const int p1s = a.Gk + 8, p2s = a.Wk + 8;

    // ---- C^ fragments of this wave's region blocks (wid, wid + 8, wid + 16) are requested first: their L2 latency runs under the
    //      staging of Q^_j [Wp][256] into LDS (16-byte pieces, padded rows)
    const bf16* ch = a.chat + (int64_t)i * a.Gr * FD;
    const float* mimg = a.mimg + (int64_t)i * a.G;
    const float* mcap = a.mcap + (int64_t)j * a.W;
    bf16x8 af[FG_SLOTS][8];
    float mi[FG_SLOTS][4];               // additive region mask; -1e4 for pad rows (exp -> 0)
#pragma unroll
    for (int s = 0; s < FG_SLOTS; ++s) {
        const int rb = wid + 8 * s;
        const bf16* ap = ch + (int64_t)((rb < nbg ? rb : 0) * 16 + lc) * FD + 8 * lq;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) af[s][ks] = *(const bf16x8*)(ap + ks * 32);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int g = rb * 16 + lq * 4 + r;
            mi[s][r] = (rb < nbg && g < a.G) ? mimg[g] : -1e4f;
        }
    }
    {
        const bf16* q = a.qhat + (int64_t)j * a.Wp * FD;
        for (int p = tid; p < a.Wp * 32; p += blockDim.x) {
            const int r = p >> 5, c = (p & 31) * 8;
            *(bf16x8*)(Qs + r * FQS + c) = *(const bf16x8*)(q + r * FD + c);
        }
    }
    __syncthreads();

    // ---- S = LeakyReLU(C^ Q^T)
#pragma unroll
    for (int s = 0; s < FG_SLOTS; ++s) {
        const int rb = wid + 8 * s;
#pragma unroll
        for (int nb = 0; nb < FW_MAXB; ++nb) R.S[s][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (rb < nbg) {
#pragma unroll
            for (int nb = 0; nb < FW_MAXB; ++nb) {
                if (nb < nwb) {
                    f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
                    const bf16* bp = Qs + (nb * 16 + lc) * FQS + 8 * lq;
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) c = mfma16(af[s][ks], *(const bf16x8*)(bp + ks * 32), c);
#pragma unroll
                    for (int r = 0; r < 4; ++r) c[r] = fmaxf(c[r], 0.1f * c[r]);             // LeakyReLU(0.1), loss.py:236
                    R.S[s][nb] = c;
                }
            }
        }
    }
    if (a.stop == 1) return;
    // validity of this lane's columns, caption mask
    float mc[FW_MAXB];                   // additive word mask; -1e4 for pad columns
#pragma unroll
    for (int nb = 0; nb < FW_MAXB; ++nb) {
        const int w = nb * 16 + lc;
        mc[nb] = (nb < nwb && w < a.W) ? mcap[w] : -1e4f;
    }

    // ---- reciprocal row / column norms of S (loss.py:238 in both directions); pad rows / columns of S are exact zeros
    float csq[FW_MAXB];
#pragma unroll
    for (int nb = 0; nb < FW_MAXB; ++nb) csq[nb] = 0.f;
    const float L2E = 1.4426950408889634f * a.lam;        // exp(lam x) = exp2(L2E x)
    float k1[FG_SLOTS][4];                                // per region: exponent scale lam log2(e) / (|S_row| + eps)
#pragma unroll
    for (int s = 0; s < FG_SLOTS; ++s) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float q = 0.f;
#pragma unroll
            for (int nb = 0; nb < FW_MAXB; ++nb) { const float v = R.S[s][nb][r]; q += v * v; csq[nb] += v * v; }
            R.rn[s][r] = 1.f / (sqrtf(row16_sum(q)) + 1e-8f);
            k1[s][r] = L2E * R.rn[s][r];
            mi[s][r] = L2E * (mi[s][r] - 1.f);            // exponent offset: lam log2(e) (m_img - 1); z <= lam, so no max pass is needed
        }
    }
#pragma unroll
    for (int nb = 0; nb < FW_MAXB; ++nb) csq[nb] = col4_sum(csq[nb]);
    xch_put(xa, wid, lane, csq, nwb);
    __syncthreads();                                   // (also: every wave is done reading Q^ from LDS)
    xch_get(xa, lane, csq, nwb);
#pragma unroll
    for (int nb = 0; nb < FW_MAXB; ++nb) R.cn[nb] = 1.f / (sqrtf(csq[nb]) + 1e-8f);
    if (a.stop == 2) return;

    // ---- image -> text: for each word, softmax over regions of lam (S rn + m_img) (loss.py:241-259), in exp2 form
    float cs[FW_MAXB];
#pragma unroll
    for (int nb = 0; nb < FW_MAXB; ++nb) cs[nb] = 0.f;
#pragma unroll
    for (int s = 0; s < FG_SLOTS; ++s)
#pragma unroll
        for (int nb = 0; nb < FW_MAXB; ++nb) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned u = pk2(__builtin_amdgcn_exp2f(fmaf(R.S[s][nb][2 * h], k1[s][2 * h], mi[s][2 * h])),
                                       __builtin_amdgcn_exp2f(fmaf(R.S[s][nb][2 * h + 1], k1[s][2 * h + 1], mi[s][2 * h + 1])));
                E[s][nb][h] = u; cs[nb] += pk_lo(u) + pk_hi(u);
            }
            __builtin_amdgcn_sched_barrier(0);            // keep the unrolled tiles from being interleaved (register pressure)
        }
#pragma unroll
    for (int nb = 0; nb < FW_MAXB; ++nb) cs[nb] = col4_sum(cs[nb]);
    xch_put(xb, wid, lane, cs, nwb);
    __syncthreads();
    xch_get(xb, lane, cs, nwb);
    // focal gate (loss.py:274-283): H = [P G - sum P > 0], P = e / sum e.  sum P is sum e * (1 / sum e) (1 within an ulp), so the
    // gate is a per-word threshold on e:  e > sum P / (G / sum e)
    float thr[FW_MAXB], gs[FW_MAXB], gd[FW_MAXB];
    const float fG = (float)a.G, fW = (float)a.W;
#pragma unroll
    for (int nb = 0; nb < FW_MAXB; ++nb) {
        const float inv = cs[nb] > 0.f ? 1.f / cs[nb] : 0.f;
        thr[nb] = a.gate ? (cs[nb] * inv) / (inv * fG) : -1.f;
        gs[nb] = 0.f; gd[nb] = 0.f;
    }
#pragma unroll
    for (int s = 0; s < FG_SLOTS; ++s)
#pragma unroll
        for (int nb = 0; nb < FW_MAXB; ++nb) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned u = E[s][nb][h];
                const float e0 = pk_lo(u), e1 = pk_hi(u);
                const unsigned keep = (e0 > thr[nb] ? 0x0000ffffu : 0u) | (e1 > thr[nb] ? 0xffff0000u : 0u);
                const unsigned v = u & keep;
                const float s0 = R.S[s][nb][2 * h], s1 = R.S[s][nb][2 * h + 1];
                E[s][nb][h] = v;
                gs[nb] += pk_lo(v) + pk_hi(v);
                gd[nb] = fmaf(pk_lo(v), fminf(s0, 10.f * s0), fmaf(pk_hi(v), fminf(s1, 10.f * s1), gd[nb]));     // S_pre: undo LeakyReLU(0.1)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
    for (int nb = 0; nb < FW_MAXB; ++nb) { gs[nb] = col4_sum(gs[nb]); gd[nb] = col4_sum(gd[nb]); }
    xch_put(xc, wid, lane, gs, nwb);
    xch_put(xd, wid, lane, gd, nwb);
    __syncthreads();
    xch_get(xc, lane, gs, nwb);
    xch_get(xd, lane, gd, nwb);
#pragma unroll
    for (int nb = 0; nb < FW_MAXB; ++nb) {
        gs[nb] = gs[nb] > 0.f ? 1.f / gs[nb] : 0.f;        // P1' = H e / sum(H e)
        if (wid == 0 && lq == 0 && nb < nwb) dotw[nb * 16 + lc] = gd[nb] * gs[nb];
    }
    // P1' -> LDS [w][g], 4 consecutive regions per 8-byte write
#pragma unroll
    for (int s = 0; s < FG_SLOTS; ++s) {
        const int rb = wid + 8 * s;
        if (rb < nbg) {
#pragma unroll
            for (int nb = 0; nb < FW_MAXB; ++nb) {
                if (nb < nwb) {
                    uint2 o;
                    o.x = pk2(pk_lo(E[s][nb][0]) * gs[nb], pk_hi(E[s][nb][0]) * gs[nb]);
                    o.y = pk2(pk_lo(E[s][nb][1]) * gs[nb], pk_hi(E[s][nb][1]) * gs[nb]);
                    *(uint2*)(P1s + (nb * 16 + lc) * p1s + rb * 16 + lq * 4) = o;
                }
            }
        }
    }
    // zero the k padding of P1' (regions Gr .. Gk-1), if any
    if (a.Gk > a.Gr) {
        for (int p = tid; p < a.Wp * (a.Gk - a.Gr); p += blockDim.x) P1s[(p / (a.Gk - a.Gr)) * p1s + a.Gr + p % (a.Gk - a.Gr)] = (bf16)0.f;
    }
    if (a.stop == 3) return;

    // ---- text -> image: for each region, softmax over words of lam (S cn + m_cap); rows live inside the wave
    float kc[FW_MAXB];
#pragma unroll
    for (int nb = 0; nb < FW_MAXB; ++nb) { kc[nb] = L2E * R.cn[nb]; mc[nb] = L2E * (mc[nb] - 1.f); }
#pragma unroll
    for (int s = 0; s < FG_SLOTS; ++s) {
        const int rb = wid + 8 * s;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float sum = 0.f;
            float e[FW_MAXB];
#pragma unroll
            for (int nb = 0; nb < FW_MAXB; ++nb) {
                e[nb] = __builtin_amdgcn_exp2f(fmaf(R.S[s][nb][r], kc[nb], mc[nb]));
                sum += e[nb];
            }
            sum = row16_sum(sum);
            const float inv = sum > 0.f ? 1.f / sum : 0.f;
            const float th = a.gate ? (sum * inv) / (inv * fW) : -1.f;
            float g1 = 0.f, g2 = 0.f;
#pragma unroll
            for (int nb = 0; nb < FW_MAXB; ++nb) {
                const float he = e[nb] > th ? e[nb] : 0.f;
                float sv = R.S[s][nb][r];
                asm volatile("" : "+v"(sv));                 // opaque copy: without it S_pre is shared with the image->text pass and 84 more registers stay live
                const float spre = fminf(sv, 10.f * sv);
                e[nb] = he;
                g1 += he; g2 = fmaf(he, spre, g2);
            }
            g1 = row16_sum(g1); g2 = row16_sum(g2);
            const float is = g1 > 0.f ? 1.f / g1 : 0.f;
            const int g = rb * 16 + lq * 4 + r;
            if (rb < nbg) {
                if (lc == 0) dotg[g] = g2 * is;
#pragma unroll
                for (int nb = 0; nb < FW_MAXB; ++nb)
                    if (nb < nwb) P2s[g * p2s + nb * 16 + lc] = (bf16)(e[nb] * is);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // zero the k padding of P2' (words Wp .. Wk-1)
    if (a.Wk > a.Wp) {
        for (int p = tid; p < a.Gr * (a.Wk - a.Wp); p += blockDim.x) P2s[(p / (a.Wk - a.Wp)) * p2s + a.Wp + p % (a.Wk - a.Wp)] = (bf16)0.f;
    }
}

__global__ __launch_bounds__(512) void xf_fwd_kernel(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, lq = lane >> 4, lc = lane & 15;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = blockIdx.x, i = blockIdx.y;
    const int nbg = a.Gr >> 4, nwb = a.Wp >> 4;
    const FLds L = flds(a.Gr, a.Gk, a.Wp, a.Wk);
    float *dotg = (float*)(smem + L.dotg), *dotw = (float*)(smem + L.dotw), *nw1 = (float*)(smem + L.nw1), *nw2 = (float*)(smem + L.nw2);
    const bf16* P1s = (const bf16*)(smem + L.p1);
    const bf16* P2s = (const bf16*)(smem + L.p2);
    const int p1s = a.Gk + 8, p2s = a.Wk + 8;
    for (int t = tid; t < 112; t += blockDim.x) nw1[t] = 0.f;
    for (int t = tid; t < 384; t += blockDim.x) nw2[t] = 0.f;
    {
        PairRegs R;
        pair_forward(a, smem, i, j, R);
    }
    // A operands of the two context products (this wave's 32 channels of C^^T and Q^^T) are requested BEFORE the barrier that
    // publishes P1' / P2': their L2 latency runs under the wait for the slowest wave
    constexpr int MAXKS = 9;                               // Gk <= 288
    const int nks1 = a.Gk >> 5, nks2 = a.Wk >> 5;
    bf16x8 ca[2][MAXKS], qa[2][4];
    {
        const bf16* ct = a.chatT + ((int64_t)i * FD + wid * 32 + lc) * a.Gk + 8 * lq;
        const bf16* qt = a.qhatT + ((int64_t)j * FD + wid * 32 + lc) * a.Wk + 8 * lq;
#pragma unroll
        for (int ks = 0; ks < MAXKS; ++ks) {
            const int k = ks < nks1 ? ks : 0;
            ca[0][ks] = *(const bf16x8*)(ct + k * 32); ca[1][ks] = *(const bf16x8*)(ct + (int64_t)16 * a.Gk + k * 32);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k = ks < nks2 ? ks : 0;
            qa[0][ks] = *(const bf16x8*)(qt + k * 32); qa[1][ks] = *(const bf16x8*)(qt + (int64_t)16 * a.Wk + k * 32);
        }
    }
    __syncthreads();
    if (a.stop >= 1 && a.stop <= 4) { if (tid == 0) a.scores[(int64_t)i * a.Bj + j] = dotg[0] + dotw[0]; return; }

    // ---- |wc_w|^2: wc^T [d][w] = C^^T [d][g] P1'^T; wave wid owns channels 32 wid .. 32 wid + 31
    {
        f32x4 acc[2][FW_MAXB];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int nb = 0; nb < FW_MAXB; ++nb) acc[db][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < MAXKS; ++ks) {
            if (ks < nks1) {
#pragma unroll
                for (int nb = 0; nb < FW_MAXB; ++nb) {
                    if (nb < nwb) {
                        const bf16x8 b = *(const bf16x8*)(P1s + (nb * 16 + lc) * p1s + ks * 32 + 8 * lq);
                        acc[0][nb] = mfma16(ca[0][ks], b, acc[0][nb]);
                        acc[1][nb] = mfma16(ca[1][ks], b, acc[1][nb]);
                    }
                }
            }
        }
#pragma unroll
        for (int nb = 0; nb < FW_MAXB; ++nb) {
            float q = 0.f;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 4; ++r) q += acc[db][nb][r] * acc[db][nb][r];
            q = col4_sum(q);
            if (nb < nwb && lq == 0) atomicAdd(&nw1[nb * 16 + lc], q);
        }
    }
    // ---- |wc2_g|^2: wc2^T [d][g] = Q^^T [d][w] P2'^T, two region blocks per iteration (independent accumulator chains)
    if (a.stop != 5) {
        for (int gb = 0; gb < nbg; gb += 2) {
            const int gb1 = gb + 1 < nbg ? gb + 1 : gb;
            f32x4 c00 = (f32x4){0.f, 0.f, 0.f, 0.f}, c01 = c00, c10 = c00, c11 = c00;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks < nks2) {
                    const bf16x8 b0 = *(const bf16x8*)(P2s + (gb * 16 + lc) * p2s + ks * 32 + 8 * lq);
                    const bf16x8 b1 = *(const bf16x8*)(P2s + (gb1 * 16 + lc) * p2s + ks * 32 + 8 * lq);
                    c00 = mfma16(qa[0][ks], b0, c00); c01 = mfma16(qa[1][ks], b0, c01);
                    c10 = mfma16(qa[0][ks], b1, c10); c11 = mfma16(qa[1][ks], b1, c11);
                }
            }
            float q0 = 0.f, q1 = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) { q0 += c00[r] * c00[r] + c01[r] * c01[r]; q1 += c10[r] * c10[r] + c11[r] * c11[r]; }
            q0 = col4_sum(q0); q1 = col4_sum(q1);
            if (lq == 0) { atomicAdd(&nw2[gb * 16 + lc], q0); if (gb + 1 < nbg) atomicAdd(&nw2[gb1 * 16 + lc], q1); }
        }
    }
    __syncthreads();
    // ---- cosines (loss.py:286-291, 317-327): means over ALL W words / G regions (padded ones included, as the reference)
    float part = 0.f;
    const float* nq = a.nq + (int64_t)j * a.W;
    const float* nc = a.nc + (int64_t)i * a.G;
    for (int w = tid; w < a.W; w += blockDim.x) {
        const float n = nq[w];
        part += dotw[w] * (n + 1e-8f) / fmaxf(n * sqrtf(nw1[w]), 1e-8f) / (float)a.W;
    }
    for (int g = tid; g < a.G; g += blockDim.x) {
        const float n = nc[g];
        part += dotg[g] * (n + 1e-8f) / fmaxf(n * sqrtf(nw2[g]), 1e-8f) / (float)a.G;
    }
    part = wave_sum(part);
    float* red = (float*)(smem + L.xb);
    if (lane == 0) red[wid] = part;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int w = 0; w < 8; ++w) t += red[w];
        a.scores[(int64_t)i * a.Bj + j] = t;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// host side (called from dvlp_xattn_fwd / dvlp_xattn_bwd in xattn.hip)
// ------------------------------------------------------------------------------------------------------------------
static inline int64_t frup(int64_t a, int64_t m) { return (a + m - 1) / m * m; }

struct FusedLayout { int64_t Gr, Gk, Wp, Wk, off_chat, off_chatT, off_qhat, off_qhatT, off_nc, off_nq, total; };
static FusedLayout fused_layout(int64_t Bi, int64_t Bj, int64_t G, int64_t W) {
    FusedLayout F{};
    F.Gr = frup(G, 16); F.Gk = frup(G, 32); F.Wp = frup(W, 16); F.Wk = frup(W, 32);
    int64_t o = 0;
    auto take = [&](int64_t bytes) { int64_t r = o; o += frup(bytes, 256); return r; };
    F.off_chat = take(Bi * F.Gr * FD * 2); F.off_chatT = take(Bi * FD * F.Gk * 2);
    F.off_qhat = take(Bj * F.Wp * FD * 2); F.off_qhatT = take(Bj * FD * F.Wk * 2);
    F.off_nc = take(Bi * G * 4); F.off_nq = take(Bj * W * 4);
    F.total = o;
    return F;
}

static int g_xf_stop = 0;
DVLP_DEV_API int dvlp_dev_xfused_ablate(int stop) { g_xf_stop = stop; return 0; }

bool dvlp_xfused_ok(int64_t G, int64_t W) {
    if (G < 1 || W < 1 || W > 16 * FW_MAXB || G > 16 * 8 * FG_SLOTS) return false;
    const FLds L = flds((int)frup(G, 16), (int)frup(G, 32), (int)frup(W, 16), (int)frup(W, 32));
    return L.total <= 160 * 1024;
}
int64_t dvlp_xfused_workspace_bytes(int64_t Bi, int64_t Bj, int64_t G, int64_t W) { return fused_layout(Bi, Bj, G, W).total; }

int dvlp_xfused_fwd(int64_t Bi, int64_t Bj, int64_t G, int64_t W, const void* Craw, const void* Qraw, const float* mimg, const float* mcap,
                    float lam, int gate, float* scores, void* workspace, hipStream_t st) {
    const FusedLayout F = fused_layout(Bi, Bj, G, W);
    char* ws = (char*)workspace;
    FusedArgs a{};
    a.chat = (const bf16*)(ws + F.off_chat); a.chatT = (const bf16*)(ws + F.off_chatT);
    a.qhat = (const bf16*)(ws + F.off_qhat); a.qhatT = (const bf16*)(ws + F.off_qhatT);
    a.nc = (const float*)(ws + F.off_nc); a.nq = (const float*)(ws + F.off_nq);
    a.mimg = mimg; a.mcap = mcap; a.scores = scores;
    a.Bi = (int)Bi; a.Bj = (int)Bj; a.G = (int)G; a.W = (int)W; a.Gr = (int)F.Gr; a.Gk = (int)F.Gk; a.Wp = (int)F.Wp; a.Wk = (int)F.Wk;
    a.lam = lam; a.gate = gate; a.stop = g_xf_stop;
    const int64_t rc = F.Gr > F.Gk ? F.Gr : F.Gk, rq = F.Wp > F.Wk ? F.Wp : F.Wk;
    hipLaunchKernelGGL(xf_prep_kernel, dim3((unsigned)cdiv(Bi * rc, 4)), dim3(256), 0, st, (int)Bi, (int)G, (int)F.Gr, (int)F.Gk, (const bf16*)Craw,
                       (bf16*)(ws + F.off_chat), (bf16*)(ws + F.off_chatT), (float*)(ws + F.off_nc));
    hipLaunchKernelGGL(xf_prep_kernel, dim3((unsigned)cdiv(Bj * rq, 4)), dim3(256), 0, st, (int)Bj, (int)W, (int)F.Wp, (int)F.Wk, (const bf16*)Qraw,
                       (bf16*)(ws + F.off_qhat), (bf16*)(ws + F.off_qhatT), (float*)(ws + F.off_nq));
    const FLds L = flds(a.Gr, a.Gk, a.Wp, a.Wk);
    { static bool once = false; if (!once) { once = true;
        (void)hipFuncSetAttribute((const void*)xf_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); } }
    hipLaunchKernelGGL(xf_fwd_kernel, dim3((unsigned)Bj, (unsigned)Bi), dim3(512), (size_t)L.total, st, a);
    return 0;
}
