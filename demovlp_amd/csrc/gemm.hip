// GEMM family for the DemoVLP hot path (K3/K5/K6/K7 of SURVEY.md section 2.2 and their backward products).
//
//   C[M,N] = epilogue( alpha * sum_k A(m,k) * B(n,k) )
//
// Operand storage ("form"):   form K: element (r,k) at X[r*ld + k]   (k contiguous; nn.Linear weight / activations)
//                             form R: element (r,k) at X[k*ld + r]   (row index contiguous)
//   forward  y  = x W^T      : A = x  form K, B = W  form K      (transA=0, transB=0)
//   backward dx = dy W       : A = dy form K, B = W  form R      (transA=0, transB=1)
//   backward dW = dy^T x     : A = dy form R, B = x  form R      (transA=1, transB=1)
//
// f32 path : v_mfma_f32_32x32x2_f32 -- exact fp32 products, fp32 accumulate (bitwise an fmaf chain), used for the
//            1e-4 parity runs.  128x128x16 tile, LDS tiles stored k-major so every operand read is conflict-free.
// bf16 path: v_mfma_f32_16x16x32_bf16, fp32 accumulate.  128x128x64 tile, 4 waves (2x2), register-prefetched
//            double-buffered LDS, one barrier per K tile, XCD-aware tile order.  form K operands read fragments with
//            ds_read_b128 from padded row-major tiles; form R operands are staged k-major and read with the gfx950
//            transposing LDS read (ds_read_b64_tr_b16), so no activation transposes are materialised in HBM.
#include "common.h"
#include <vector>
#include <cstdlib>
#include <cstdlib>

struct Epi {
    const float* bias;   // [N] fp32 or null
    const void* res;     // residual [M, ldres] (same dtype as C) or null
    void* aux;           // GELU: pre-activation out; *_BWD: pre-activation in
    int64_t ldres, ldaux;
    int flags;
    float alpha;
    int64_t sA, sB, sC, sRes, sAux;   // batch strides in elements (blockIdx.y = batch index)
    int vec;                          // C / res / aux rows are 16-byte aligned: the bf16 epilogue may use 8-wide accesses
    float* csum;                      // 256-row kernel: [2 * row tiles][N] partial column sums of the stored output, or null
};

// Scalar epilogue.  Deliberately NOT inlined: it is called 64x per thread from the f32 kernel and from the ragged-edge path
// of the bf16 kernel; inlining every copy (each with erff/expf expansions) made the kernels ~230 KB of code, far past the
// instruction cache, and the K loop paid for it in fetch stalls (measured: 10 % MFMA utilisation).
template <typename T>
__device__ __attribute__((noinline)) void epi_store(const Epi& e, T* __restrict__ C, int64_t ldc, int64_t m, int64_t n, float v) {
    v *= e.alpha;
    if (e.bias) v += e.bias[n];
    if (e.flags & EPI_GELU) {
        ((T*)e.aux)[m * e.ldaux + n] = from_f<T>(v);
        v = sizeof(T) == 2 ? gelu_poly2((f32x2){v, 0.f})[0] : gelu_erf(v);       // bf16: the same polynomial as the vector epilogues
    }
    if (e.flags & EPI_LEAKY) v = v > 0.f ? v : 0.1f * v;
    if (e.flags & EPI_GELU_BWD) {
        const float h = to_f(((const T*)e.aux)[m * e.ldaux + n]);
        v *= sizeof(T) == 2 ? gelu_grad_poly2((f32x2){h, 0.f})[0] : gelu_erf_grad(h);
    }
    if (e.flags & EPI_RELU_BWD) v = to_f(((const T*)e.aux)[m * e.ldaux + n]) > 0.f ? v : 0.f;
    if (e.res) v += to_f(((const T*)e.res)[m * e.ldres + n]);
    if (e.flags & EPI_OUT_F32) {
        float* Cf = (float*)C;
        if (e.flags & EPI_ACCUM) v += Cf[m * ldc + n];
        Cf[m * ldc + n] = v;
        return;
    }
    if (e.flags & EPI_ACCUM) v += to_f(C[m * ldc + n]);
    C[m * ldc + n] = from_f<T>(v);
}

// 8 consecutive columns of one output row (bf16 kernel's LDS-staged epilogue): vector loads of bias / residual / aux and
// one 16-byte (bf16) or two 16-byte (fp32-out) stores per lane instead of eight scattered 2-byte ones.
__device__ __forceinline__ void epi_store8(const Epi& e, bf16* __restrict__ C, int64_t ldc, int64_t m, int64_t n, float (&v)[8], int64_t N) {
    if (!e.vec || n + 7 >= N) {
        const Epi ec = e;       // the out-of-line callee takes an address: hand it a copy so the caller's Epi stays in SGPRs
#pragma unroll
        for (int t = 0; t < 8; ++t) if (n + t < N) epi_store<bf16>(ec, C, ldc, m, n + t, v[t]);
        return;
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] *= e.alpha;
    if (e.bias) {
        const float4 b0 = *(const float4*)(e.bias + n), b1 = *(const float4*)(e.bias + n + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
    }
    if (e.flags & EPI_GELU) {
        bf16x8 pre;
#pragma unroll
        for (int t = 0; t < 8; ++t) pre[t] = (bf16)v[t];
#pragma unroll
        for (int t = 0; t < 8; t += 2) { const f32x2 y = gelu_poly2((f32x2){v[t], v[t + 1]}); v[t] = y[0]; v[t + 1] = y[1]; }   // as in epi_store8_pre
        *(bf16x8*)((bf16*)e.aux + m * e.ldaux + n) = pre;
    }
    if (e.flags & EPI_LEAKY) {
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = v[t] > 0.f ? v[t] : 0.1f * v[t];
    }
    if (e.flags & (EPI_GELU_BWD | EPI_RELU_BWD)) {
        const bf16x8 a = *(const bf16x8*)((const bf16*)e.aux + m * e.ldaux + n);
        if (e.flags & EPI_GELU_BWD) {
#pragma unroll
            for (int t = 0; t < 8; t += 2) {
                const f32x2 d = gelu_grad_poly2((f32x2){(float)a[t], (float)a[t + 1]});
                v[t] *= d[0]; v[t + 1] *= d[1];
            }
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = (float)a[t] > 0.f ? v[t] : 0.f;
        }
    }
    if (e.res) {
        const bf16x8 r = *(const bf16x8*)((const bf16*)e.res + m * e.ldres + n);
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] += (float)r[t];
    }
    if (e.flags & EPI_OUT_F32) {
        float* Cf = (float*)C + m * ldc + n;
        if (e.flags & EPI_ACCUM) {
            const float4 c0 = *(const float4*)Cf, c1 = *(const float4*)(Cf + 4);
            v[0] += c0.x; v[1] += c0.y; v[2] += c0.z; v[3] += c0.w; v[4] += c1.x; v[5] += c1.y; v[6] += c1.z; v[7] += c1.w;
        }
        *(float4*)Cf = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)(Cf + 4) = make_float4(v[4], v[5], v[6], v[7]);
        return;
    }
    bf16* Cp = C + m * ldc + n;
    if (e.flags & EPI_ACCUM) {
        const bf16x8 c = *(const bf16x8*)Cp;
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] += (float)c[t];
    }
    bf16x8 o;
#pragma unroll
    for (int t = 0; t < 8; ++t) o[t] = (bf16)v[t];
    __builtin_nontemporal_store(o, (bf16x8*)Cp);      // streaming: keep the operand panels in L2 (see epi_store8_pre)
}

// epi_store8 for a whole, aligned 8-column group with the operands it would load (bias, residual, aux-in) already in
// registers: the 256-row kernel runs one workgroup per CU, so nothing hides a dependent load inside its store loop.
// `fl` / `has_res` are e.flags / (e.res != null), or compile-time constants in the kernels specialised on the epilogue kind.
// NOAB: alpha == 1 and the bias is already inside the accumulators (persistent form: they are initialised with it)
template <bool NOAB = false>
__device__ __forceinline__ void epi_store8_pre(const Epi& e, const int fl, const bool has_res, bf16* __restrict__ C, int64_t ldc, int64_t m, int64_t n,
                                               float (&v)[8], const float4 b0, const float4 b1, const bf16x8 r, const bf16x8 a) {
    if constexpr (!NOAB) {
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] *= e.alpha;
        if (e.bias) { v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w; }
    }
    if (fl & EPI_GELU) {
        bf16x8 pre;
#pragma unroll
        for (int t = 0; t < 8; ++t) pre[t] = (bf16)v[t];
#pragma unroll
        for (int t = 0; t < 8; t += 2) { const f32x2 y = gelu_poly2((f32x2){v[t], v[t + 1]}); v[t] = y[0]; v[t + 1] = y[1]; }
        __builtin_nontemporal_store(pre, (bf16x8*)((bf16*)e.aux + m * e.ldaux + n));
    }
    if (fl & EPI_LEAKY) {
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = v[t] > 0.f ? v[t] : 0.1f * v[t];
    }
    if (fl & EPI_GELU_BWD) {
#pragma unroll
        for (int t = 0; t < 8; t += 2) {
            const f32x2 d = gelu_grad_poly2((f32x2){(float)a[t], (float)a[t + 1]});
            v[t] *= d[0]; v[t + 1] *= d[1];
        }
    }
    if (fl & EPI_RELU_BWD) {
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = (float)a[t] > 0.f ? v[t] : 0.f;
    }
    if (has_res) {
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] += (float)r[t];
    }
    if (fl & EPI_OUT_F32) {
        float* Cf = (float*)C + m * ldc + n;
        if (fl & EPI_ACCUM) {
            const float4 c0 = *(const float4*)Cf, c1 = *(const float4*)(Cf + 4);
            v[0] += c0.x; v[1] += c0.y; v[2] += c0.z; v[3] += c0.w; v[4] += c1.x; v[5] += c1.y; v[6] += c1.z; v[7] += c1.w;
        }
        *(float4*)Cf = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)(Cf + 4) = make_float4(v[4], v[5], v[6], v[7]);
        return;
    }
    bf16* Cp = C + m * ldc + n;
    if (fl & EPI_ACCUM) {
        const bf16x8 c = *(const bf16x8*)Cp;
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] += (float)c[t];
    }
    bf16x8 o;
#pragma unroll
    for (int t = 0; t < 8; ++t) { o[t] = (bf16)v[t]; v[t] = (float)o[t]; }     // v <- the values as stored (column-sum fusion)
    // streaming store: a 256 x 256 tile's outputs (128-256 KB per CU per round) would otherwise push the operand panels that the
    // neighbouring column tiles are about to re-read out of the 4 MiB L2
    if (fl & (64 << 24)) *(bf16x8*)Cp = o;            // timing ablation: ordinary store
    else __builtin_nontemporal_store(o, (bf16x8*)Cp);
}

// XCD-aware remap: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD a contiguous run of tiles
// (neighbouring tiles share an A panel -> L2 hits).  Bijective for any grid size.
__device__ __forceinline__ int64_t xcd_remap(int64_t bid, int64_t nwg) {
    const int64_t q = nwg / 8, r = nwg % 8, xcd = bid % 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
}
// 32-bit forms for the 256-row kernel: its blocks are few and short-lived (12 K tiles at K = 768), and the 64-bit divisions
// of the generic forms sit in front of the first LDS-DMA of every block
__device__ __forceinline__ int xcd_remap32(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
__device__ __forceinline__ void tile_of32(int wg, int ntm, int ntn, int& tm, int& tn) {
    const int per_group = 8 * ntn;                      // GROUP_M = 8
    const int grp = wg / per_group, first = grp * 8;
    const int gsz = ntm - first < 8 ? ntm - first : 8;
    const int in = wg - grp * per_group;
    tn = in / gsz;
    tm = first + in - tn * gsz;
}

// Tile order inside that run: groups of GROUP_M row panels, column-major inside a group, so the ~64 workgroups resident
// on one XCD (32 CUs x 2) cover an ~8 x 8 patch of tiles: 16 operand panels (3 MB at K = 768) stay in the 4 MiB L2
// instead of the 20-27 panels a row-major sweep of an 18-24 tile wide output touches.
constexpr int GROUP_M = 8;
__device__ __forceinline__ void tile_of(int64_t wg, int64_t ntm, int64_t ntn, int64_t& tm, int64_t& tn) {
    const int64_t per_group = GROUP_M * ntn;
    const int64_t grp = wg / per_group, first = grp * GROUP_M;
    const int64_t gsz = ntm - first < GROUP_M ? ntm - first : GROUP_M;
    const int64_t in = wg % per_group;
    tm = first + in % gsz;
    tn = in / gsz;
}

// ------------------------------------------------------------------------------------------------------------------
// f32 kernel
// ------------------------------------------------------------------------------------------------------------------
constexpr int F_BM = 128, F_BN = 128, F_BK = 16, F_LD = F_BM + 4;

template <bool FORM_R>
__device__ __forceinline__ void f32_stage(float (*S)[F_LD], const float* __restrict__ X, int64_t ld, int64_t r0, int64_t k0,
                                          int64_t R, int64_t K, bool vec_ok, int tid) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = tid + 256 * i;
        if (FORM_R) {  // X[k*ld + r]: float4 along r
            const int k = p >> 5, rq = (p & 31) * 4;
            const int64_t gk = k0 + k, gr = r0 + rq;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gk < K) {
                const float* src = X + gk * ld + gr;
                if (vec_ok && gr + 3 < R) v = *(const float4*)src;
                else {
                    if (gr + 0 < R) v.x = src[0];
                    if (gr + 1 < R) v.y = src[1];
                    if (gr + 2 < R) v.z = src[2];
                    if (gr + 3 < R) v.w = src[3];
                }
            }
            *(float4*)&S[k][rq] = v;
        } else {       // X[r*ld + k]: float4 along k
            const int row = p >> 2, kq = (p & 3) * 4;
            const int64_t gr = r0 + row, gk = k0 + kq;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gr < R) {
                const float* src = X + gr * ld + gk;
                if (vec_ok && gk + 3 < K) v = *(const float4*)src;
                else {
                    if (gk + 0 < K) v.x = src[0];
                    if (gk + 1 < K) v.y = src[1];
                    if (gk + 2 < K) v.z = src[2];
                    if (gk + 3 < K) v.w = src[3];
                }
            }
            S[kq + 0][row] = v.x; S[kq + 1][row] = v.y; S[kq + 2][row] = v.z; S[kq + 3][row] = v.w;
        }
    }
}

template <bool A_R, bool B_R>
__global__ __launch_bounds__(256) void gemm_f32_kernel(int64_t M, int64_t N, int64_t K, const float* __restrict__ A, int64_t lda,
                                                       const float* __restrict__ B, int64_t ldb, float* __restrict__ C, int64_t ldc,
                                                       Epi e, int a_vec, int b_vec, int64_t ntn) {
    __shared__ __attribute__((aligned(16))) float As[F_BK][F_LD];
    __shared__ __attribute__((aligned(16))) float Bs[F_BK][F_LD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t wg = xcd_remap(blockIdx.x, gridDim.x);
    int64_t tm_, tn_;
    tile_of(wg, gridDim.x / ntn, ntn, tm_, tn_);
    const int64_t m0 = tm_ * F_BM, n0 = tn_ * F_BN;
    A += blockIdx.y * e.sA; B += blockIdx.y * e.sB; C += blockIdx.y * e.sC;
    if (e.res) e.res = (const float*)e.res + blockIdx.y * e.sRes;
    if (e.aux) e.aux = (float*)e.aux + blockIdx.y * e.sAux;
    const int wm = (wid >> 1) * 64, wn = (wid & 1) * 64;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int64_t k0 = 0; k0 < K; k0 += F_BK) {
        f32_stage<A_R>(As, A, lda, m0, k0, M, K, a_vec, tid);
        f32_stage<B_R>(Bs, B, ldb, n0, k0, N, K, b_vec, tid);
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < F_BK; kk += 2) {
            const int kr = kk + (lane >> 5), c = lane & 31;
            float a[2], b[2];
            a[0] = As[kr][wm + c]; a[1] = As[kr][wm + 32 + c];
            b[0] = Bs[kr][wn + c]; b[1] = Bs[kr][wn + 32 + c];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D map of 32x32: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int64_t n = n0 + wn + j * 32 + (lane & 31);
                if (m < M && n < N) epi_store<float>(e, C, ldc, m, n, acc[i][j][r]);
            }
}

// ------------------------------------------------------------------------------------------------------------------
// bf16 kernel
// ------------------------------------------------------------------------------------------------------------------
constexpr int H_BM = 128, H_BN = 128, H_BK = 64;
constexpr int H_LDK = H_BK;        // form K tile: [128 rows][64 k], 128-B rows, 16-B chunk c stored at slot c ^ (row & 7):
                                   // the 16 lanes of a ds_read_b128 group then hit 16 distinct 4-bank slots (conflict-free)
constexpr int H_LDR = H_BM + 8;    // form R tile: [64 k][128 rows] padded to 136 elements (272 B rows, 16-B aligned)
constexpr int H_TILE = (H_BM * H_LDK > H_BK * H_LDR ? H_BM * H_LDK : H_BK * H_LDR);   // elements per operand tile

struct Stage4 { uint4 v[4]; };

// global -> registers (4 x 16 B per thread per operand), SAFE form: out-of-range rows / k are zero-filled element-wise.
// Only used for operands whose row count is not a multiple of 8 (divergent branches in the K loop cost ~2x).
template <bool FORM_R>
__device__ __forceinline__ void h_load_safe(Stage4& s, const bf16* __restrict__ X, int64_t ld, int64_t r0, int64_t k0, int64_t R,
                                            int64_t K, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = tid + 256 * i;
        int64_t gr, gk;
        bool ok;
        const bf16* src;
        if (FORM_R) { const int k = p >> 4, rq = (p & 15) * 8; gk = k0 + k; gr = r0 + rq; ok = gk < K && gr + 7 < R; src = X + gk * ld + gr; }
        else        { const int row = p >> 3, kq = (p & 7) * 8; gr = r0 + row; gk = k0 + kq; ok = gr < R && gk + 7 < K; src = X + gr * ld + gk; }
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (ok) v = *(const uint4*)src;
        else {
            bf16 t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                bool in;
                if (FORM_R) in = gk < K && gr + j < R; else in = gr < R && gk + j < K;
                t[j] = in ? src[j] : (bf16)0.f;
            }
            v = *(uint4*)t;
        }
        s.v[i] = v;
    }
}

// FAST form: no per-lane branches in the K loop.  Rows past the edge are CLAMPED onto the last valid row / 8-row chunk
// (they only feed output rows >= M or columns >= N, which are never stored); only the K tail needs zeros, and that is a
// workgroup-uniform case taken for the last K tile alone.
template <bool FORM_R>
__device__ __forceinline__ void h_load_fast(Stage4& s, const bf16* __restrict__ X, int64_t ld, int64_t r0, int64_t k0, int64_t R,
                                            int64_t K, int tid, bool ktail) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = tid + 256 * i;
        if (FORM_R) {
            const int k = p >> 4, rq = (p & 15) * 8;
            int64_t gr = r0 + rq;
            gr = gr > R - 8 ? R - 8 : gr;
            int64_t gk = k0 + k;
            if (!ktail) s.v[i] = *(const uint4*)(X + gk * ld + gr);
            else {
                const bool in = gk < K;
                gk = in ? gk : K - 1;
                uint4 v = *(const uint4*)(X + gk * ld + gr);
                s.v[i] = in ? v : make_uint4(0u, 0u, 0u, 0u);
            }
        } else {
            const int row = p >> 3, kq = (p & 7) * 8;
            int64_t gr = r0 + row;
            gr = gr > R - 1 ? R - 1 : gr;
            int64_t gk = k0 + kq;
            if (!ktail) s.v[i] = *(const uint4*)(X + gr * ld + gk);
            else {
                // K is a multiple of 8 on this path, so an 8-element chunk is entirely inside or entirely outside
                const bool in = gk < K;
                gk = in ? gk : K - 8;
                uint4 v = *(const uint4*)(X + gr * ld + gk);
                s.v[i] = in ? v : make_uint4(0u, 0u, 0u, 0u);
            }
        }
    }
}

template <bool FORM_R>
__device__ __forceinline__ void h_store(bf16* __restrict__ S, const Stage4& s, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = tid + 256 * i;
        if (FORM_R) { const int k = p >> 4, rq = (p & 15) * 8; *(uint4*)&S[k * H_LDR + rq] = s.v[i]; }
        else        { const int row = p >> 3, ch = p & 7; *(uint4*)&S[row * H_LDK + ((ch ^ (row & 7)) << 3)] = s.v[i]; }
    }
}

// Four MFMA 16x16x32 operand fragments (rows [rbase + 16 i, +16), i = 0..3; k in [ks, ks+32)) of one LDS tile.
// lane l holds X[row = l&15][k = 8*(l>>4) + j], j = 0..7.
static_assert(H_LDR == 136, "immediate offsets of the transposing reads below assume 272-byte k-rows");
template <bool FORM_R>
__device__ __forceinline__ void h_frags(bf16x8 (&f)[4], const bf16* __restrict__ S, int rbase, int ks, int lane) {
    if (!FORM_R) {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = *(const bf16x8*)&S[(rbase + 16 * i + (lane & 15)) * H_LDK + ((((ks >> 3) + (lane >> 4)) ^ (lane & 7)) << 3)];
    } else {
        // ds_read_b64_tr_b16: per 16-lane group, lane 4q+p supplies the address of block row q, columns 4p..4p+3 of a
        // 4-row x 16-column block of 16-bit elements; lane i receives column i, rows 0..3.  Block rows = k, columns = r.
        // Group g = lane>>4 needs k = ks + 8g + {0..3} (lo) and + {4..7} (hi = +4 k-rows = +1088 B); tile i is +32 B.
        // All eight reads are issued back to back and retired by ONE wait.
        const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
        const unsigned a0 = (unsigned)(uintptr_t)&S[(ks + 8 * g + q) * H_LDR + rbase + 4 * pp];
        bf16x4 l0, h0, l1, h1, l2, h2, l3, h3;
        asm volatile(
            "ds_read_b64_tr_b16 %0, %8\n\t"
            "ds_read_b64_tr_b16 %1, %8 offset:1088\n\t"
            "ds_read_b64_tr_b16 %2, %8 offset:32\n\t"
            "ds_read_b64_tr_b16 %3, %8 offset:1120\n\t"
            "ds_read_b64_tr_b16 %4, %8 offset:64\n\t"
            "ds_read_b64_tr_b16 %5, %8 offset:1152\n\t"
            "ds_read_b64_tr_b16 %6, %8 offset:96\n\t"
            "ds_read_b64_tr_b16 %7, %8 offset:1184\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1), "=&v"(l2), "=&v"(h2), "=&v"(l3), "=&v"(h3) : "v"(a0) : "memory");
        f[0] = __builtin_shufflevector(l0, h0, 0, 1, 2, 3, 4, 5, 6, 7);
        f[1] = __builtin_shufflevector(l1, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        f[2] = __builtin_shufflevector(l2, h2, 0, 1, 2, 3, 4, 5, 6, 7);
        f[3] = __builtin_shufflevector(l3, h3, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

template <bool A_R, bool B_R, bool SAFE>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(int64_t M, int64_t N, int64_t K, const bf16* __restrict__ A, int64_t lda,
                                                        const bf16* __restrict__ B, int64_t ldb, bf16* __restrict__ C, int64_t ldc,
                                                        Epi e, int64_t ntn, int64_t kchunk, float* __restrict__ slab) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16* smem = (bf16*)smem_raw;                        // [2 buffers][A tile | B tile]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t wg = xcd_remap(blockIdx.x, gridDim.x);
    int64_t tm_, tn_;
    tile_of(wg, gridDim.x / ntn, ntn, tm_, tn_);
    const int64_t m0 = tm_ * H_BM, n0 = tn_ * H_BN;
    A += blockIdx.y * e.sA; B += blockIdx.y * e.sB; C += blockIdx.y * e.sC * ((e.flags & EPI_OUT_F32) ? 2 : 1);
    if (e.res) e.res = (const bf16*)e.res + blockIdx.y * e.sRes;
    if (e.aux) e.aux = (bf16*)e.aux + blockIdx.y * e.sAux;
    const int wm = (wid >> 1) * 64, wn = (wid & 1) * 64;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // split-K: blockIdx.z owns k in [kbeg, kend); partial tiles go to fp32 slabs and a second kernel applies the epilogue
    const int64_t kbeg = blockIdx.z * kchunk, kend = kbeg + kchunk < K ? kbeg + kchunk : K;
    const int64_t nk = (kend - kbeg + H_BK - 1) / H_BK;
    const bool has_tail = (kend - kbeg) % H_BK != 0;
    Stage4 ra, rb;
    auto load_tile = [&](int64_t kt) {
        const int64_t k0 = kbeg + kt * H_BK;
        if (SAFE) {
            h_load_safe<A_R>(ra, A, lda, m0, k0, M, kend, tid);
            h_load_safe<B_R>(rb, B, ldb, n0, k0, N, kend, tid);
        } else {
            const bool tail = has_tail && kt == nk - 1;
            h_load_fast<A_R>(ra, A, lda, m0, k0, M, kend, tid, tail);
            h_load_fast<B_R>(rb, B, ldb, n0, k0, N, kend, tid, tail);
        }
    };
    load_tile(0);
    h_store<A_R>(smem, ra, tid);
    h_store<B_R>(smem + H_TILE, rb, tid);
    __syncthreads();
    for (int64_t kt = 0; kt < nk; ++kt) {
        const bf16* As = smem + (kt & 1) * 2 * H_TILE;
        const bf16* Bs = As + H_TILE;
        if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
        for (int ks = 0; ks < H_BK; ks += 32) {
            bf16x8 af[4], bfr[4];
            h_frags<A_R>(af, As, wm, ks, lane);
            h_frags<B_R>(bfr, Bs, wn, ks, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) {
            bf16* An = smem + ((kt + 1) & 1) * 2 * H_TILE;
            h_store<A_R>(An, ra, tid);
            h_store<B_R>(An + H_TILE, rb, tid);
        }
        __syncthreads();
    }
    // Epilogue through LDS: the wave parks its 64x64 fp32 tile (C/D map of 16x16: col = lane&15, row = 4*(lane>>4)+reg)
    // in its own 64 x 68 float region (the K-loop buffers are free now), then every lane owns 8 consecutive columns of
    // a row: 16-byte loads of bias/residual/aux and 16-byte stores instead of 64 scattered 2-byte stores per lane.
    float* Ct = (float*)smem_raw + wid * (64 * 68);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) Ct[(16 * i + 4 * (lane >> 4) + r) * 68 + 16 * j + (lane & 15)] = acc[i][j][r];
    // same-wave LDS accesses complete in order: no barrier needed before reading the wave's own region back
#pragma unroll 1
    for (int it = 0; it < 8; ++it) {
        const int row = it * 8 + (lane >> 3), col = (lane & 7) * 8;
        const int64_t m = m0 + wm + row, n = n0 + wn + col;
        const float4 c0 = *(const float4*)&Ct[row * 68 + col], c1 = *(const float4*)&Ct[row * 68 + col + 4];
        float v[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        if (m < M && n < N) {
            if (slab) {
                float* dst = slab + ((int64_t)(blockIdx.y * gridDim.z + blockIdx.z) * M + m) * N + n;
                if ((N & 3) == 0 && n + 7 < N) { *(float4*)dst = c0; *(float4*)(dst + 4) = c1; }
                else {
#pragma unroll
                    for (int t = 0; t < 8; ++t) if (n + t < N) dst[t] = v[t];
                }
            } else epi_store8(e, C, ldc, m, n, v, N);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// bf16 kernel, LDS-DMA variant (K % 64 == 0, row counts of form-R operands % 8 == 0): the default on the hot path.
// Tiles go global -> LDS with global_load_lds_dwordx4 (no VGPR staging, no ds_write), two stages, one barrier per K
// tile.  An LDS-DMA instruction writes 1 KiB linearly (lane i -> base + 16 i), so tiles are UNPADDED and bank conflicts
// are removed by permuting which global 16-byte chunk each lane fetches (swizzle on the source side, same involution on
// the read side):
//   form K tile [128 rows][64 k]  (128-B rows): chunk c of row r lives in slot c ^ (r & 7)          -> ds_read_b128
//   form R tile [64 k][128 rows]  (256-B rows): chunk c of k-row k lives in slot c ^ (2 m(k)),
//                                 m(k) = 4 ((k >> 3) & 1) + (k & 3)                                 -> ds_read_b64_tr_b16
// ------------------------------------------------------------------------------------------------------------------
// Tile geometry of the LDS-DMA kernel: BN = 128, BK = 64, WM waves along M (2 -> BM = 128, 256 threads, 2 workgroups per
// CU; 4 -> BM = 256, 512 threads, 1 workgroup per CU).  The K loop is bound by the L2 -> LDS fill rate (measured ~10 TB/s
// chip-wide: 128x128 tiles give 64 FLOP per filled byte = ~650 TFLOP/s), so the wide tile (96 FLOP/B) is preferred
// whenever the grid still fills the chip.
template <int ROWS> struct GTile { static constexpr int BYTES = ROWS * 128; };

// 16 zero bytes in global memory: the LDS-DMA source of every chunk that lies beyond K in a ragged last K tile (the source
// address of an LDS-DMA is per lane, so zero-filling costs one select, no extra instruction)
__device__ __attribute__((aligned(16))) const unsigned g_zero16[4] = {0u, 0u, 0u, 0u};

// kend < 0: the whole 64-wide K tile is valid; else k >= kend reads zeros (K % 8 == 0 is required)
template <bool FORM_R, int ROWS, int NW, bool TAIL = false>
__device__ __forceinline__ void g_issue(char* lds_tile, const bf16* __restrict__ X, int64_t ld, int64_t r0, int64_t k0, int64_t R,
                                        int wid, int lane, int64_t kend = -1) {
    constexpr int PIECES = ROWS / 8, PER_WAVE = PIECES / NW;       // 1-KiB pieces of the tile
    constexpr int CPR = ROWS / 8;                                    // form R: 16-byte chunks per k-row
    constexpr int KPP = 64 / CPR;                                    // form R: k-rows per piece
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) {
        const int q = wid * PER_WAVE + j;                          // piece index (wave-uniform)
        const bf16* src;
        if (!FORM_R) {
            const int rl = 8 * q + (lane >> 3), c = (lane & 7) ^ (lane >> 3);
            int64_t gr = r0 + rl;
            gr = gr > R - 1 ? R - 1 : gr;
            src = X + gr * ld + k0 + c * 8;
            if (TAIL && k0 + c * 8 >= kend) src = (const bf16*)g_zero16;
        } else {
            const int kl = KPP * q + lane / CPR, m = ((kl >> 3) & 1) * 4 + (kl & 3), c = (lane % CPR) ^ (m << 1);
            int64_t gr = r0 + c * 8;
            gr = gr > R - 8 ? R - 8 : gr;
            src = X + (k0 + kl) * ld + gr;
            if (TAIL && k0 + kl >= kend) src = (const bf16*)g_zero16;
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds_tile + q * 1024), 16, 0, 0);
    }
}

// 8 transposing reads (4 fragments x lo/hi) from 4 per-fragment addresses; hi = +4 k-rows = +HI bytes; no wait (g_wait8)
template <int HI>
__device__ __forceinline__ void g_tr8(bf16x4 (&lo)[4], bf16x4 (&hi)[4], unsigned a0, unsigned a1, unsigned a2, unsigned a3) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %8\n\t"
        "ds_read_b64_tr_b16 %1, %8 offset:%12\n\t"
        "ds_read_b64_tr_b16 %2, %9\n\t"
        "ds_read_b64_tr_b16 %3, %9 offset:%12\n\t"
        "ds_read_b64_tr_b16 %4, %10\n\t"
        "ds_read_b64_tr_b16 %5, %10 offset:%12\n\t"
        "ds_read_b64_tr_b16 %6, %11\n\t"
        "ds_read_b64_tr_b16 %7, %11 offset:%12"
        : "=&v"(lo[0]), "=&v"(hi[0]), "=&v"(lo[1]), "=&v"(hi[1]), "=&v"(lo[2]), "=&v"(hi[2]), "=&v"(lo[3]), "=&v"(hi[3])
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "i"(HI) : "memory");
}
// retire every outstanding LDS read; naming the destinations makes every consumer depend on this statement
__device__ __forceinline__ void g_wait8(bf16x4 (&lo)[4], bf16x4 (&hi)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2]), "+v"(lo[3]), "+v"(hi[3]) :: "memory");
}

template <bool A_R, bool B_R, int WM, int STAGES>
__global__ __launch_bounds__(128 * WM) void gemm_bf16_glds_kernel(int64_t M, int64_t N, int64_t K, const bf16* __restrict__ A, int64_t lda,
                                                                  const bf16* __restrict__ B, int64_t ldb, bf16* __restrict__ C, int64_t ldc,
                                                                  Epi e, int64_t ntn, int64_t kchunk, float* __restrict__ slab) {
    constexpr int NW = 2 * WM, BM = 64 * WM, BN = 128;
    constexpr int A_B = GTile<BM>::BYTES, B_B = GTile<BN>::BYTES, STAGE = A_B + B_B;
    constexpr int A_ROWB = BM * 2, B_ROWB = BN * 2;               // form R: bytes per k-row
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t wg = xcd_remap(blockIdx.x, gridDim.x);
    int64_t tm_, tn_;
    tile_of(wg, gridDim.x / ntn, ntn, tm_, tn_);
    const int64_t m0 = tm_ * BM, n0 = tn_ * BN;
    A += blockIdx.y * e.sA; B += blockIdx.y * e.sB; C += blockIdx.y * e.sC * ((e.flags & EPI_OUT_F32) ? 2 : 1);
    if (e.res) e.res = (const bf16*)e.res + blockIdx.y * e.sRes;
    if (e.aux) e.aux = (bf16*)e.aux + blockIdx.y * e.sAux;
    const int wm = (wid >> 1) * 64, wn = (wid & 1) * 64;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int64_t kbeg = blockIdx.z * kchunk, kend = kbeg + kchunk < K ? kbeg + kchunk : K;
    const int64_t nk = (kend - kbeg + H_BK - 1) / H_BK;          // the last K tile may be ragged (K % 8 == 0): zero-filled
    const bool ragged_k = (kend - kbeg) % H_BK != 0;

    // per-lane LDS read offsets inside a tile (stage / k-step offsets are added below)
    const int g = lane >> 4, r = lane & 15, qq = (lane >> 2) & 3, pp = lane & 3;
    unsigned offA[4], offB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (!A_R) offA[i] = (unsigned)((wm + 16 * i + r) * 128);                                      // + slot*16 per k-step
        else      offA[i] = (unsigned)((8 * g + qq) * A_ROWB + ((((wm >> 3) + 2 * i + (pp >> 1)) ^ ((((g & 1) << 2) + qq) << 1)) << 4) + ((pp & 1) << 3));
        if (!B_R) offB[i] = (unsigned)((wn + 16 * i + r) * 128);
        else      offB[i] = (unsigned)((8 * g + qq) * B_ROWB + ((((wn >> 3) + 2 * i + (pp >> 1)) ^ ((((g & 1) << 2) + qq) << 1)) << 4) + ((pp & 1) << 3));
    }
    const unsigned lds0 = (unsigned)(uintptr_t)smem_raw;

    auto issue = [&](int64_t kt, int st) {
        char* base = smem_raw + st * STAGE;
        if (ragged_k && kt == nk - 1) {                       // wave-uniform: the common path carries no per-lane K test
            g_issue<A_R, BM, NW, true>(base, A, lda, m0, kbeg + kt * H_BK, M, wid, lane, kend);
            g_issue<B_R, BN, NW, true>(base + A_B, B, ldb, n0, kbeg + kt * H_BK, N, wid, lane, kend);
        } else {
            g_issue<A_R, BM, NW>(base, A, lda, m0, kbeg + kt * H_BK, M, wid, lane);
            g_issue<B_R, BN, NW>(base + A_B, B, ldb, n0, kbeg + kt * H_BK, N, wid, lane);
        }
    };
    const int ab = e.flags >> 24;                      // timing ablations (0 in production)
    // DMA instructions one wave issues per K tile (vmcnt bookkeeping of the 3-stage pipeline)
    constexpr int NPW = (BM / 8) / NW + (BN / 8) / NW;
    issue(0, 0);
    if (STAGES == 3) { if (nk > 1) issue(1, 1); }
    else __syncthreads();                              // also drains the LDS-DMA (vmcnt(0))
    for (int64_t kt = 0; kt < nk; ++kt) {
        int cur;
        if (STAGES == 3) {
            // Tiles kt and kt+1 are in flight.  Wait for THIS wave's pieces of tile kt only (vmcnt counts in issue order),
            // then a raw barrier (no vmcnt(0) drain, unlike __syncthreads) makes every wave's pieces of tile kt visible and
            // proves all waves finished reading tile kt-1, whose stage is refilled right after with tile kt+2.
            cur = (int)(kt % 3);
            if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + 2 < nk && !(ab & 1)) issue(kt + 2, (int)((kt + 2) % 3));
        } else {
            cur = (int)(kt & 1);
            if (kt + 1 < nk && !(ab & 1)) issue(kt + 1, cur ^ 1);       // stage cur^1 was last read before the previous barrier
        }
        const unsigned tA = lds0 + cur * STAGE, tB = tA + A_B;
        bf16x8 af[2][4], bfr[2][4];
        bf16x4 alo[2][4], ahi[2][4], blo[2][4], bhi[2][4];
        if (!(ab & 2) || kt == 0) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (A_R) g_tr8<4 * A_ROWB>(alo[s], ahi[s], tA + offA[0] + s * 32 * A_ROWB, tA + offA[1] + s * 32 * A_ROWB, tA + offA[2] + s * 32 * A_ROWB, tA + offA[3] + s * 32 * A_ROWB);
            else {
#pragma unroll
                for (int i = 0; i < 4; ++i) af[s][i] = *(const bf16x8*)(smem_raw + cur * STAGE + offA[i] + ((((4 * s + g) ^ (r & 7))) << 4));
            }
            if (B_R) g_tr8<4 * B_ROWB>(blo[s], bhi[s], tB + offB[0] + s * 32 * B_ROWB, tB + offB[1] + s * 32 * B_ROWB, tB + offB[2] + s * 32 * B_ROWB, tB + offB[3] + s * 32 * B_ROWB);
            else {
#pragma unroll
                for (int i = 0; i < 4; ++i) bfr[s][i] = *(const bf16x8*)(smem_raw + cur * STAGE + A_B + offB[i] + ((((4 * s + g) ^ (r & 7))) << 4));
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (A_R) {
                g_wait8(alo[s], ahi[s]);
#pragma unroll
                for (int i = 0; i < 4; ++i) af[s][i] = __builtin_shufflevector(alo[s][i], ahi[s][i], 0, 1, 2, 3, 4, 5, 6, 7);
            }
            if (B_R) {
                g_wait8(blo[s], bhi[s]);
#pragma unroll
                for (int i = 0; i < 4; ++i) bfr[s][i] = __builtin_shufflevector(blo[s][i], bhi[s][i], 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
        }
        if (!(ab & 4)) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s][i], bfr[s][j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { asm volatile("" :: "v"(af[0][i]), "v"(af[1][i]), "v"(bfr[0][i]), "v"(bfr[1][i])); }
        }
        if (STAGES != 3) __syncthreads();
    }
    if (STAGES == 3) __syncthreads();
    // epilogue through LDS in two 32-row passes (8.7 KB per wave per pass, so 8 waves fit the 96 KB of the wide variant)
    float* Ct = (float*)smem_raw + wid * (32 * 68);
#pragma unroll
    for (int hp = 0; hp < 2; ++hp) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) Ct[(16 * i + 4 * (lane >> 4) + rr) * 68 + 16 * j + (lane & 15)] = acc[2 * hp + i][j][rr];
#pragma unroll 1
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + (lane >> 3), col = (lane & 7) * 8;
            const int64_t m = m0 + wm + 32 * hp + row, n = n0 + wn + col;
            const float4 c0 = *(const float4*)&Ct[row * 68 + col], c1 = *(const float4*)&Ct[row * 68 + col + 4];
            float v[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
            if (m < M && n < N) {
                if (slab) {
                    float* dst = slab + ((int64_t)(blockIdx.y * gridDim.z + blockIdx.z) * M + m) * N + n;
                    if ((N & 3) == 0 && n + 7 < N) { *(float4*)dst = c0; *(float4*)(dst + 4) = c1; }
                    else {
#pragma unroll
                        for (int t = 0; t < 8; ++t) if (n + t < N) dst[t] = v[t];
                    }
                } else epi_store8(e, C, ldc, m, n, v, N);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// 256 x 256 x 64 tile, 8 waves (2 x 4), one workgroup per CU, ping-pong schedule ("p8": 8 phases per two K tiles).
//
// Why: the 128 x 128 kernel above reads 16 KiB of LDS per wave per K tile for 32 MFMAs -- on a CU that is exactly the LDS
// bandwidth (128 B/clk) the MFMA pipes need at full rate, so with fills and barriers on top it settles near a third of
// peak.  Here a wave owns 128 x 64 outputs (24 KiB of LDS reads for 64 MFMAs: 75 % of the LDS budget at full MFMA rate),
// and the two waves that share a SIMD run half a phase apart: while one issues its 16-MFMA cluster the other issues the
// LDS reads of its next cluster and its share of the LDS-DMA prefetch.
//
// Units: a K tile is staged as four 16-KiB half tiles ("units", 128 rows x 64 k, each laid out exactly like a tile of the
// 128-row kernel so the same swizzles and fragment-read formulas apply), in the order they are consumed:
//        kind 0 = A rows 0..127 (A_lo)   1 = B rows 0..127 (B_lo)   2 = B rows 128..255 (B_hi)   3 = A rows 128..255 (A_hi)
// Wave (wr, wc) owns output rows {64 wr + [0,64)} u {128 + 64 wr + [0,64)} and columns {32 wc + [0,32)} u {128 + 32 wc + [0,32)}.
// Phase q of a K tile (16 MFMAs = one 64 x 32 quadrant x K 64):
//        q0: read A_lo, B_lo -> acc[0..3][0..1]     q1: read B_hi -> acc[0..3][2..3]
//        q2: read A_hi       -> acc[4..7][2..3]     q3: (no reads) -> acc[4..7][0..1]
// Global phase P = 4 t + q stages unit u = P + 6 (one unit = 2 LDS-DMA instructions per wave), i.e. 4-5 phases ahead of
// its first read, into the LDS slot whose last read was >= 2 phases ago (kinds 0/1 are last read in q0, 2 in q1, 3 in q2).
//
// Synchronisation (two raw s_barriers per phase, no vmcnt(0) in the steady state):
//   * group wr = 1 runs one barrier behind group wr = 0, so its load segment coincides with the other group's MFMAs;
//   * RAW: every wave waits (counted vmcnt) in phase P for all units <= P + 2, i.e. for what phase P + 1 reads; both groups
//     have passed that wait before the barrier that opens the first read of phase P + 1;
//   * WAR: the reads of phase p are retired by the lgkmcnt(0) that precedes its MFMAs; the slot is refilled in phase p + 2
//     at the earliest, two barriers later for either group.
// ------------------------------------------------------------------------------------------------------------------
constexpr int P_UNIT = 128 * 128;        // bytes of one unit
constexpr int P_BUF = 4 * P_UNIT;        // one K tile
constexpr int P_LDS = 2 * P_BUF;         // 128 KiB

template <int HI>
__device__ __forceinline__ void g_tr4(bf16x4 (&lo)[2], bf16x4 (&hi)[2], unsigned a0, unsigned a1) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %4\n\t"
        "ds_read_b64_tr_b16 %1, %4 offset:%6\n\t"
        "ds_read_b64_tr_b16 %2, %5\n\t"
        "ds_read_b64_tr_b16 %3, %5 offset:%6"
        : "=&v"(lo[0]), "=&v"(hi[0]), "=&v"(lo[1]), "=&v"(hi[1])
        : "v"(a0), "v"(a1), "i"(HI) : "memory");
}
__device__ __forceinline__ void g_wait4(bf16x4 (&lo)[2], bf16x4 (&hi)[2]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]) :: "memory");
}
// the same reads with the unit / k-step displacement as an IMMEDIATE offset: the address registers then stay the 4 (or 2)
// per-lane fragment offsets (per buffer parity) instead of one hoisted VGPR per (parity, unit, k-step, fragment) combination
template <int OFF>
__device__ __forceinline__ void g_tr8i(bf16x4 (&lo)[4], bf16x4 (&hi)[4], unsigned a0, unsigned a1, unsigned a2, unsigned a3) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %8 offset:%12\n\t"
        "ds_read_b64_tr_b16 %1, %8 offset:%13\n\t"
        "ds_read_b64_tr_b16 %2, %9 offset:%12\n\t"
        "ds_read_b64_tr_b16 %3, %9 offset:%13\n\t"
        "ds_read_b64_tr_b16 %4, %10 offset:%12\n\t"
        "ds_read_b64_tr_b16 %5, %10 offset:%13\n\t"
        "ds_read_b64_tr_b16 %6, %11 offset:%12\n\t"
        "ds_read_b64_tr_b16 %7, %11 offset:%13"
        : "=&v"(lo[0]), "=&v"(hi[0]), "=&v"(lo[1]), "=&v"(hi[1]), "=&v"(lo[2]), "=&v"(hi[2]), "=&v"(lo[3]), "=&v"(hi[3])
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "i"(OFF), "i"(OFF + 1024) : "memory");
}
template <int OFF>
__device__ __forceinline__ void g_tr4i(bf16x4 (&lo)[2], bf16x4 (&hi)[2], unsigned a0, unsigned a1) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %4 offset:%6\n\t"
        "ds_read_b64_tr_b16 %1, %4 offset:%7\n\t"
        "ds_read_b64_tr_b16 %2, %5 offset:%6\n\t"
        "ds_read_b64_tr_b16 %3, %5 offset:%7"
        : "=&v"(lo[0]), "=&v"(hi[0]), "=&v"(lo[1]), "=&v"(hi[1])
        : "v"(a0), "v"(a1), "i"(OFF), "i"(OFF + 1024) : "memory");
}
template <int N> __device__ __forceinline__ void p_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
// allow `units` (<= 4) staged units = 2 * units LDS-DMA instructions of this wave to stay in flight
__device__ __forceinline__ void p_wait_units(int units) {
    if (units >= 4) p_vmcnt<8>();
    else if (units == 3) p_vmcnt<6>();
    else if (units == 2) p_vmcnt<4>();
    else if (units == 1) p_vmcnt<2>();
    else p_vmcnt<0>();
}
__device__ __forceinline__ void p_glds(const bf16* src, char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

// ragged / split-K store of one 32-row pass (edge tiles, unaligned outputs, fp32 slabs): rare or cheap, so ONE out-of-line
// copy instead of four inlined ones per kernel (the epilogue code would otherwise dwarf the K loop in the instruction cache)
__device__ __attribute__((noinline)) void p8_store_ragged(const Epi& e, bf16* __restrict__ C, int64_t ldc, const float* Ct, int64_t mrow0, int64_t ncol,
                                                           int64_t M, int64_t N, float* __restrict__ slab_out, int lane) {
    const int col = (lane & 7) * 8, rsub = lane >> 3;
#pragma unroll 1
    for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + rsub;
        const int64_t m = mrow0 + row;
        const float4 c0 = *(const float4*)&Ct[row * 68 + col], c1 = *(const float4*)&Ct[row * 68 + col + 4];
        float v[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        if (m < M && ncol < N) {
            if (slab_out) {
                float* dst = slab_out + m * N + ncol;
                if ((N & 3) == 0 && ncol + 7 < N) { *(float4*)dst = c0; *(float4*)(dst + 4) = c1; }
                else {
#pragma unroll
                    for (int t = 0; t < 8; ++t) if (ncol + t < N) dst[t] = v[t];
                }
            } else epi_store8(e, C, ldc, m, ncol, v, N);
        }
    }
}

// Timeline stamps (tools/p8_timeline.py builds a second library with -DDVLP_STAMP; the product library carries none of this):
// per workgroup {block, HW_ID, XCC_ID, entry, first data landed, K loop done, stores issued, stores acknowledged} on the 100 MHz
// s_memrealtime clock, kept in SGPRs and written once at the very end (a store inside the K loop would count in vmcnt).
#ifdef DVLP_STAMP
__device__ unsigned long long* g_p8_stamp = nullptr;
extern "C" int dvlp_p8_stamp_buffer(void* p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_p8_stamp), &p, sizeof p) == hipSuccess ? DVLP_OK : DVLP_ERR_LAUNCH; }
// timing experiment (P8_ALIAS=1 python tools/p8_timeline.py): every workgroup LOADS the operands of tile (0, 0) (outputs still go to its own
// tile), i.e. the whole chip streams the same two panels out of L2 -- does a full-chip K loop run at the speed it has on 60 CUs once the
// memory system is taken out?  Round 4: no (21.5 vs 21.3 us stamped; 17.0 on 60 CUs): the full-chip K loop is clock-bound, not fill-bound.
__device__ int g_p8_alias = 0;
extern "C" int dvlp_p8_alias(int on) { return hipMemcpyToSymbol(HIP_SYMBOL(g_p8_alias), &on, sizeof on) == hipSuccess ? DVLP_OK : DVLP_ERR_LAUNCH; }
#define P8_STAMP(i) do { if (st) st[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
// per-phase shader-clock stamps inside the K loop (issued where they are taken, consumed at the end of the phase so that no extra wait lands
// between the fragment reads and the barrier): where a phase spends its time -- load segment, first barrier, MFMA cluster, second barrier
#define P8_PH(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define P8_PH_ACC(q) do { if (st) { st[8 + 4 * (q)] += ph1 - ph0; st[9 + 4 * (q)] += ph2 - ph1; st[10 + 4 * (q)] += ph3 - ph2; st[11 + 4 * (q)] += ph4 - ph3; } } while (0)
#else
#define P8_STAMP(i) do { } while (0)
#define P8_PH(v) do { } while (0)
#define P8_PH_ACC(q) do { } while (0)
#endif

// One 256 x 256 output tile at (m0, n0) over K tiles [kbeg, kbeg + 64 nk); `slab_out` non-null: raw fp32 partial (split-K).
// EK = epilogue kind of whole tiles: 0 bias only, 1 bias + residual, 2 GELU forward (pre-activation out), 3 GELU backward
// (pre-activation in), 4 anything (flags read at run time).  Kinds 0-3 are straight-line code: with run-time flag tests around the
// residual / aux loads the compiler closes every step with s_waitcnt vmcnt(0), which also waits for the previous step's stores.
constexpr int P8_EK_ANY = 4;
// MIH: 16-row blocks per wave in the UPPER half of the tile (rows 128..): 4 = 256-row tile; 3 = 224-row tile (the upper unit is still
// staged whole, its last 32 rows are simply not multiplied).  Tile height is a ROUND-QUANTISATION knob: with M = 18496 tokens, 256-row
// tiles give 73 x {3, 9, 12} = 219 / 657 / 876 tiles = 1 / 3 / 4 rounds on 256 CUs for 0.86 / 2.57 / 3.42 rounds of work, 224-row tiles
// give 83 x {3, 9, 12} = 249 / 747 / 996 tiles: the same 1 / 3 / 4 rounds, each 7/8 as long.
template <int MIH> constexpr int p8_tile_rows() { return 128 + 32 * MIH; }
// Persistent form (PERS, gemm_bf16_p8p_kernel): one workgroup walks several tiles.  The unit stream is CONTINUOUS across tiles
// (nk even, so a tile ends on a buffer-parity boundary): the last six phases of a tile already stage units 0..5 of the next one,
// the epilogue's loads / stores are issued behind them and never waited for as a whole, and the next K loop starts on landed data.
// vmcnt retires in issue order, so while the epilogue's operations are younger than the units a wait is for, they are simply ADDED
// to the allowed count (p8_epi_ops: a LOWER bound of what the epilogue issues per wave -- a larger count would wait too little).
// Only whole tiles (no ragged / split-K path: those would pull ~40 SGPRs and scratch into the loop; a scratch reload waits vmcnt(0)).
struct P8Next { bool first; int64_t m0n, n0n; bool prev_counted; };     // (m0n, n0n): the next tile, or this tile again for the last one
                                                                         // (the six units staged for nobody are drained at kernel end)
template <int EK, int MIH> constexpr int p8_epi_ops() { return EK == 0 ? 2 * (4 + MIH) : (EK >= 1 && EK <= 3) ? 4 * (4 + MIH) : 0; }
template <bool A_R, bool B_R, int EK, int MIH = 4, bool PERS = false>
__device__ __forceinline__ void p8_tile(char* smem_raw, int64_t M, int64_t N, const bf16* __restrict__ A, int64_t lda, const bf16* __restrict__ B,
                                        int64_t ldb, bf16* __restrict__ C, int64_t ldc, const Epi& e, int64_t m0, int64_t n0, int64_t kbeg, int nk,
                                        float* __restrict__ slab_out, unsigned long long* st = nullptr, const P8Next nx = P8Next{true, 0, 0, false}) {
    const int tid = threadIdx.x;
    int lane = tid & 63;
    // persistent form: everything derived from the lane id (source pointers, fragment offsets, the epilogue's row / column offsets) is
    // recomputed per tile; an opaque copy keeps the compiler from hoisting those registers out of the tile loop
    if constexpr (PERS) asm volatile("" : "+v"(lane));
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int U = 4 * nk;
    constexpr int EOPS = p8_epi_ops<EK, MIH>();
    static_assert(MIH == 4 || !A_R, "short tiles are only built for row-major (form K) A operands");

    // per-lane source pointers of this wave's two LDS-DMA pieces of every unit kind; they advance one K tile per use
    auto src_ptr = [&](auto form_r, const bf16* X, int64_t ld, int64_t r0, int64_t R, int j) -> const bf16* {
        const int q = wid * 2 + j;                                     // 1-KiB piece of the unit
        if constexpr (!decltype(form_r)::value) {
            const int rl = 8 * q + (lane >> 3), c = (lane & 7) ^ (lane >> 3);
            int64_t gr = r0 + rl;
            gr = gr > R - 1 ? R - 1 : gr;
            return X + gr * ld + kbeg + c * 8;
        } else {
            const int kl = 4 * q + (lane >> 4), m = ((kl >> 3) & 1) * 4 + (kl & 3), c = (lane & 15) ^ (m << 1);
            int64_t gr = r0 + c * 8;
            gr = gr > R - 8 ? R - 8 : gr;
            return X + (kbeg + kl) * ld + gr;
        }
    };
    using FA = std::integral_constant<bool, A_R>;
    using FB = std::integral_constant<bool, B_R>;
    // source cursors of the wave's LDS-DMA pieces: pointers in the one-tile form; in the persistent form 32-bit ELEMENT offsets from A / B
    // (8 registers instead of 16 live across the K loop; the dispatch guarantees the operands span < 2^31 elements)
    using Cur = std::conditional_t<PERS, unsigned, const bf16*>;
    Cur sp[4][2];
    auto cur_of = [&](const bf16* ptr, const bf16* base) -> Cur { if constexpr (PERS) return (unsigned)(ptr - base); else return ptr; };
    auto set_ptrs = [&](int64_t m0_, int64_t n0_) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            sp[0][j] = cur_of(src_ptr(FA{}, A, lda, m0_, M, j), A);
            sp[3][j] = cur_of(src_ptr(FA{}, A, lda, m0_ + 128, M, j), A);
            sp[1][j] = cur_of(src_ptr(FB{}, B, ldb, n0_, N, j), B);
            sp[2][j] = cur_of(src_ptr(FB{}, B, ldb, n0_ + 128, N, j), B);
        }
    };
#ifdef DVLP_STAMP
    if (g_p8_alias) set_ptrs(0, 0); else
#endif
    set_ptrs(m0, n0);          // PERS, follow-on tile: units 0..5 were staged by the previous tile; the cursors are rebuilt and advanced
                               // below (cheaper than keeping them alive across the epilogue)
    const int64_t kstepA = A_R ? H_BK * lda : H_BK, kstepB = B_R ? H_BK * ldb : H_BK;
    auto stage = [&](auto kind_, int par) {
        constexpr int KIND = decltype(kind_)::value;
        char* dst = smem_raw + par * P_BUF + KIND * P_UNIT + wid * 2048;
        const bf16* base = (KIND == 0 || KIND == 3) ? A : B;
        if constexpr (PERS) { p_glds(base + sp[KIND][0], dst); p_glds(base + sp[KIND][1], dst + 1024); }
        else { p_glds(sp[KIND][0], dst); p_glds(sp[KIND][1], dst + 1024); }
        const int64_t ks = (KIND == 0 || KIND == 3) ? kstepA : kstepB;
        sp[KIND][0] += (std::conditional_t<PERS, unsigned, int64_t>)ks; sp[KIND][1] += (std::conditional_t<PERS, unsigned, int64_t>)ks;
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    // prologue FIRST: units 0..5 (tile 0 and the first half of tile 1) are requested before anything else is set up, so the
    // accumulator clears and fragment-offset arithmetic below run under the memory latency instead of in front of it
    auto stage_bias = [&](int64_t n0_) {           // PERS: 256 fp32 bias values of a tile -> this wave's LDS copy (one LDS-DMA instruction)
        if (e.bias) p_glds((const bf16*)(e.bias + n0_ + 4 * lane), smem_raw + P_LDS + wid * 1024);
    };
    // persistent form: EIGHT units (two whole K tiles) are ahead of every tile's first phase -- units 0..5 from the previous tile's last
    // phases, 6 and 7 from behind its K loop -- so that the epilogue's stores, which sit in the vmcnt queue behind them, are first
    // waited for in phase 7 (the first wait that needs a unit younger than they are)
    if (!PERS || nx.first) {
        if constexpr (PERS) stage_bias(n0);
        stage(I0{}, 0); stage(I1{}, 0); stage(I2{}, 0); stage(I3{}, 0);
        if (nk > 1) { stage(I0{}, 1); stage(I1{}, 1); }
        if constexpr (PERS) { stage(I2{}, 1); stage(I3{}, 1); }
    } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            using D = std::conditional_t<PERS, unsigned, int64_t>;
            sp[0][j] += (D)(2 * kstepA); sp[1][j] += (D)(2 * kstepB); sp[2][j] += (D)(2 * kstepB); sp[3][j] += (D)(2 * kstepA);
        }
    }

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // per-lane fragment-read offsets inside a unit
    const int g = lane >> 4, r = lane & 15, qq = (lane >> 2) & 3, pp = lane & 3;
    unsigned aK[2], aKh[2], bK[2], aR[4], bR[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        aK[s] = (unsigned)((64 * wr + r) * 128 + (((4 * s + g) ^ (r & 7)) << 4));
        aKh[s] = (unsigned)((16 * MIH * wr + r) * 128 + (((4 * s + g) ^ (r & 7)) << 4));      // upper half: MIH blocks per wave group
        bK[s] = (unsigned)((32 * wc + r) * 128 + (((4 * s + g) ^ (r & 7)) << 4));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        aR[i] = (unsigned)((8 * g + qq) * 256 + (((8 * wr + 2 * i + (pp >> 1)) ^ ((((g & 1) << 2) + qq) << 1)) << 4) + ((pp & 1) << 3));
#pragma unroll
    for (int j = 0; j < 2; ++j)
        bR[j] = (unsigned)((8 * g + qq) * 256 + (((4 * wc + 2 * j + (pp >> 1)) ^ ((((g & 1) << 2) + qq) << 1)) << 4) + ((pp & 1) << 3));
    const unsigned lds0 = (unsigned)(uintptr_t)smem_raw;
    // transposing reads take (address register + immediate): one register set per buffer parity (the 64 KiB between the
    // two K-tile buffers does not fit the 16-bit offset field)
    unsigned aRp[2][4], bRp[2][2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
#pragma unroll
        for (int i = 0; i < 4; ++i) aRp[par][i] = lds0 + par * P_BUF + aR[i];
#pragma unroll
        for (int j = 0; j < 2; ++j) bRp[par][j] = lds0 + par * P_BUF + bR[j];
    }

    bf16x8 aF[2][4], bL[2][2], bH[2][2];                      // [k-step][fragment]
    bf16x4 alo[2][4], ahi[2][4], bllo[2][2], blhi[2][2], bhlo[2][2], bhhi[2][2];
    auto read_a = [&](auto par_, auto kind_) {
        constexpr int PAR = decltype(par_)::value, KOFF = decltype(kind_)::value * P_UNIT;
        if constexpr (A_R) {
            g_tr8i<KOFF>(alo[0], ahi[0], aRp[PAR][0], aRp[PAR][1], aRp[PAR][2], aRp[PAR][3]);
            g_tr8i<KOFF + 8192>(alo[1], ahi[1], aRp[PAR][0], aRp[PAR][1], aRp[PAR][2], aRp[PAR][3]);
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < (KOFF ? MIH : 4); ++i) aF[s][i] = *(const bf16x8*)(smem_raw + PAR * P_BUF + KOFF + (KOFF ? aKh : aK)[s] + i * 2048);
        }
    };
    auto read_b = [&](auto par_, auto kind_, bf16x8 (&bf)[2][2], bf16x4 (&lo)[2][2], bf16x4 (&hi)[2][2]) {
        constexpr int PAR = decltype(par_)::value, KOFF = decltype(kind_)::value * P_UNIT;
        if constexpr (B_R) {
            g_tr4i<KOFF>(lo[0], hi[0], bRp[PAR][0], bRp[PAR][1]);
            g_tr4i<KOFF + 8192>(lo[1], hi[1], bRp[PAR][0], bRp[PAR][1]);
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[s][j] = *(const bf16x8*)(smem_raw + PAR * P_BUF + KOFF + bK[s] + j * 2048);
        }
    };
    auto land_a = [&]() {
        if (A_R) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                g_wait8(alo[s], ahi[s]);
#pragma unroll
                for (int i = 0; i < 4; ++i) aF[s][i] = __builtin_shufflevector(alo[s][i], ahi[s][i], 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
    };
    auto land_b = [&](bf16x8 (&bf)[2][2], bf16x4 (&lo)[2][2], bf16x4 (&hi)[2][2]) {
        if (B_R) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                g_wait4(lo[s], hi[s]);
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[s][j] = __builtin_shufflevector(lo[s][j], hi[s][j], 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
    };

    // VAR: 0 = one-tile form (run-time tail handling); persistent form, K-tile pairs of a tile: 1 = first (the previous epilogue's
    // operations may sit in the queue), 2 = middle, 3 = last (switches to the next tile's units at phase U - 6 = (q 2, parity 0))
    auto phase = [&](auto q_, auto par_, auto var_, int P) {
        constexpr int Q = decltype(q_)::value, PAR = decltype(par_)::value, VAR = decltype(var_)::value;
        P8_PH(ph0);
        // ---- load segment: fragment reads of this phase, one unit of prefetch, counted wait for what the NEXT phase reads
        if constexpr (Q == 0) { read_b(par_, I1{}, bL, bllo, blhi); read_a(par_, I0{}); }
        if constexpr (Q == 1) read_b(par_, I2{}, bH, bhlo, bhhi);
        if constexpr (Q == 2) read_a(par_, I3{});
        if constexpr (VAR == 0) {
            if (P + 6 < U) stage(std::integral_constant<int, (Q + 2) & 3>{}, Q < 2 ? (PAR ^ 1) : PAR);
            if constexpr (Q != 2) p_wait_units(U - 3 - P);
        } else {
            if constexpr (VAR == 3 && Q == 2 && PAR == 0) { set_ptrs(nx.m0n, nx.n0n); stage_bias(nx.n0n); }
            // (phases 0 and 1 of a tile stage nothing: units 6 and 7 are already on their way)
            if constexpr (!(VAR == 1 && PAR == 0 && Q < 2)) stage(std::integral_constant<int, (Q + 2) & 3>{}, Q < 2 ? (PAR ^ 1) : PAR);
            if constexpr (Q != 2) {
                // four units stay in flight (five / four + the two early ones in phases 0 / 1: 8 is a lower bound there); through phase 5 of
                // a follow-on tile the previous epilogue's operations sit between the units waited for and the youngest ones
                if constexpr (VAR == 1 && !(PAR == 1 && Q == 3) && EOPS > 0) { if (nx.prev_counted) p_vmcnt<8 + EOPS>(); else p_vmcnt<8>(); }
                else p_vmcnt<8>();
            }
        }
        P8_PH(ph1);
        __builtin_amdgcn_s_barrier();
        P8_PH(ph2);
        // ---- MFMA segment
        if constexpr (Q == 0) { land_b(bL, bllo, blhi); land_a(); }
        if constexpr (Q == 1) land_b(bH, bhlo, bhhi);
        if constexpr (Q == 2) land_a();
        constexpr int MI = (Q >= 2) ? 4 : 0, NJ = (Q == 1 || Q == 2) ? 2 : 0;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < (MI ? MIH : 4); ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[MI + i][NJ + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((NJ ? bH : bL)[s][j], aF[s][i], acc[MI + i][NJ + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        P8_PH(ph3);
        __builtin_amdgcn_s_barrier();
        P8_PH(ph4);
        P8_PH_ACC(Q);
    };
    auto pair = [&](auto var_, int t) {
        phase(I0{}, I0{}, var_, 4 * t + 0); phase(I1{}, I0{}, var_, 4 * t + 1); phase(I2{}, I0{}, var_, 4 * t + 2); phase(I3{}, I0{}, var_, 4 * t + 3);
        phase(I0{}, I1{}, var_, 4 * t + 4); phase(I1{}, I1{}, var_, 4 * t + 5); phase(I2{}, I1{}, var_, 4 * t + 6); phase(I3{}, I1{}, var_, 4 * t + 7);
    };

    // wait for units 0 and 1 of the prologue issued at the top
    if (PERS && EOPS > 0 && nx.prev_counted) p_vmcnt<8 + EOPS>();            // units 2..5 and the previous tile's epilogue may still be in flight
    else if (nk > 1) p_vmcnt<8>(); else p_vmcnt<4>();
    if constexpr (PERS) {
        // accumulators start from the bias (alpha == 1): this wave's own 1-KiB copy of the tile's 256 bias values, staged by LDS-DMA ahead
        // of the tile's unit 0 (so the wait above covers it) -- the epilogue then has no load the compiler would close with vmcnt(0)
        if (e.bias) {
            const float* bl = (const float*)(smem_raw + P_LDS + wid * 1024);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 b4 = *(const f32x4*)(bl + 32 * wc + 16 * (j & 1) + 128 * (j >> 1) + 4 * (lane >> 4));
#pragma unroll
                for (int i = 0; i < 4 + MIH; ++i) acc[i][j] = b4;
            }
        }
    }
    __builtin_amdgcn_s_barrier();
    P8_STAMP(1);
    if (wr == 1) __builtin_amdgcn_s_barrier();                 // group 1 runs one barrier behind group 0
    if constexpr (PERS) {                                      // nk even, >= 4
        pair(I1{}, 0);
#pragma unroll 1
        for (int t = 2; t < nk - 2; t += 2) pair(I2{}, t);
        pair(I3{}, nk - 2);
    } else {
        for (int t = 0; t < nk; t += 2) {
            phase(I0{}, I0{}, I0{}, 4 * t + 0); phase(I1{}, I0{}, I0{}, 4 * t + 1); phase(I2{}, I0{}, I0{}, 4 * t + 2); phase(I3{}, I0{}, I0{}, 4 * t + 3);
            if (t + 1 < nk) {
                phase(I0{}, I1{}, I0{}, 4 * t + 4); phase(I1{}, I1{}, I0{}, 4 * t + 5); phase(I2{}, I1{}, I0{}, 4 * t + 6); phase(I3{}, I1{}, I0{}, 4 * t + 7);
            }
        }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();                 // re-align the groups: every LDS read and DMA has retired
    P8_STAMP(2);
    if constexpr (PERS) { stage(I2{}, 1); stage(I3{}, 1); }    // the next tile's units 6 and 7 (their slots were last read two phases ago)

    // Epilogue, per wave and without workgroup barriers.  The products above are issued with the B fragment as the FIRST MFMA
    // operand, so the accumulator tile is C^T: lane (g = lane / 16, r = lane % 16) holds acc[i][j][0..3] = C[row 16 i' + r][columns
    // 16 j' + 4 g + 0..3] -- four CONSECUTIVE columns of one row.  fp32 outputs (split-K slabs) leave as 16-byte stores as they are;
    // for bf16 one v_permlane16_swap per register pairs the two 16-column blocks so that a lane owns 8 consecutive columns
    // (16 bytes) and a store instruction writes 16 rows x 64 contiguous bytes.  No LDS round trip (it was ~20 % of a K = 768 product).
    const int ab = PERS ? 0 : e.flags >> 24;    // timing ablations (0 in production): 8 no stores, 16 no epilogue, 32 stores hit 256 rows only, 64 no nt
    if (ab & 16) { if (acc[0][0][0] == 123.456f && acc[7][3][3] == 1.f) C[0] = (bf16)1.f; return; }
    if (ab & 8) M = 0;
    const int r16 = lane & 15, g4 = lane >> 4;
    auto row_of = [&](int ii) { return m0 + (ii < 4 ? 64 * wr : 128 + 16 * MIH * wr) + 16 * (ii & 3) + r16; };
    if (!PERS && slab_out && n0 + 256 <= N && (N & 3) == 0) {
#pragma unroll
        for (int ii = 0; ii < 4 + MIH; ++ii) {
            const int64_t m = row_of(ii);
            if (m < M) {
                float* dst = slab_out + m * N + n0 + 32 * wc + 4 * g4;
#pragma unroll
                for (int j = 0; j < 4; ++j) *(f32x4*)(dst + 128 * (j >> 1) + 16 * (j & 1)) = acc[ii][j];
            }
        }
        return;
    }
    const bool whole = PERS || (e.vec && n0 + 256 <= N && !slab_out);   // every 8-column group of this block is whole and 16-byte aligned
    if (whole) {
        const int fl = EK == P8_EK_ANY ? e.flags : EK == 2 ? EPI_GELU : EK == 3 ? EPI_GELU_BWD : (e.flags & EPI_LEAKY);   // LeakyReLU: no loads, stays a run-time test
        const bool has_res = EK == P8_EK_ANY ? e.res != nullptr : EK == 1, has_aux = (fl & (EPI_GELU_BWD | EPI_RELU_BWD)) != 0;
        const int64_t cn0 = n0 + 32 * wc + 16 * (g4 & 1) + 8 * (g4 >> 1);      // + 128 hh: this lane's 8 columns after the swap
        float4 bb[2][2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            bb[hh][0] = make_float4(0.f, 0.f, 0.f, 0.f); bb[hh][1] = bb[hh][0];
            if (!PERS && e.bias) { bb[hh][0] = *(const float4*)(e.bias + cn0 + 128 * hh); bb[hh][1] = *(const float4*)(e.bias + cn0 + 128 * hh + 4); }
        }
        float cs[2][8];                                    // e.csum: this lane's column sums over the rows it stores
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int t = 0; t < 8; ++t) cs[hh][t] = 0.f;
        // residual / aux-in rows are requested three row blocks ahead: one workgroup per CU means no other wave hides a dependent
        // load, and vmcnt retires in issue order, so a load waits for every store issued before it
        bf16x8 pr[4][2] = {}, pa[4][2] = {};
        auto preload = [&](int ii, bf16x8 (&r)[2], bf16x8 (&a)[2]) {
            int64_t m = row_of(ii);
            m = m < M ? m : (M > 0 ? M - 1 : 0);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                if (has_res) r[hh] = *(const bf16x8*)((const bf16*)e.res + m * e.ldres + cn0 + 128 * hh);
                if (has_aux) a[hh] = *(const bf16x8*)((const bf16*)e.aux + m * e.ldaux + cn0 + 128 * hh);
            }
        };
        preload(0, pr[0], pa[0]); preload(1, pr[1], pa[1]); preload(2, pr[2], pa[2]);
#pragma unroll
        for (int ii = 0; ii < 4 + MIH; ++ii) {
            if (ii + 3 < 4 + MIH) preload(ii + 3, pr[(ii + 3) & 3], pa[(ii + 3) & 3]);
            const int64_t m = row_of(ii);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                float v[8];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[ii][2 * hh][t]), __float_as_uint(acc[ii][2 * hh + 1][t]), false, false);
                    v[t] = __uint_as_float(sw[0]); v[4 + t] = __uint_as_float(sw[1]);
                }
                if (m < M) {
                    epi_store8_pre<PERS>(e, fl, has_res, C, ldc, (EK == P8_EK_ANY && (ab & 32)) ? (m & 255) : m, cn0 + 128 * hh, v, bb[hh][0], bb[hh][1], pr[ii & 3][hh], pa[ii & 3][hh]);
                    if (!PERS && e.csum) {
#pragma unroll
                        for (int t = 0; t < 8; ++t) cs[hh][t] += v[t];
                    }
                }
            }
        }
        if (!PERS && e.csum) {
            // the 16 lanes of a row group hold the same 8 columns for different rows: combine, one partial row per wave
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                for (int t = 0; t < 8; ++t) cs[hh][t] = row16_sum(cs[hh][t]);
                if (r16 == 0) {
                    float* dst = e.csum + ((m0 / p8_tile_rows<MIH>()) * 2 + wr) * N + cn0 + 128 * hh;
                    *(float4*)dst = make_float4(cs[hh][0], cs[hh][1], cs[hh][2], cs[hh][3]);
                    *(float4*)(dst + 4) = make_float4(cs[hh][4], cs[hh][5], cs[hh][6], cs[hh][7]);
                }
            }
        }
        return;
    }
    // Only the run-time-flag kernel (EK = anything) carries the ragged path: the specialised kinds, the short tiles and the persistent form
    // are dispatched when every tile takes one of the two paths above (N a multiple of 256; aligned outputs unless split).  Round 6: with the
    // ragged path (an out-of-line call that keeps two copies of the epilogue descriptor alive) in every 256-row instantiation, each carried
    // 30-38 spilled SGPRs; the text tower's split-K launches ran on them.
    if constexpr (EK != P8_EK_ANY || MIH != 4 || PERS) return;
    // ragged tiles / unaligned outputs: four 32-row passes through LDS (rows {lo, hi} x {first, second 32}; a pass covers the
    // wave's 32 + 32 columns), scalar epilogue out of line
    constexpr int P_EPW = 32 * 68 * 4;                         // bytes per wave: staged accumulators
    float* Ct = (float*)(smem_raw + wid * P_EPW);
    const int col = (lane & 7) * 8;
    const int64_t ncol = n0 + 32 * wc + (col < 32 ? col : 96 + col);
    auto mrow_of = [&](int hp) { return m0 + 128 * (hp >> 1) + 64 * wr + 32 * (hp & 1); };
#pragma unroll 1
    for (int hp = 0; hp < 4; ++hp) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 t4 = hp == 0 ? acc[i][j] : hp == 1 ? acc[2 + i][j] : hp == 2 ? acc[4 + i][j] : acc[6 + i][j];
                *(f32x4*)&Ct[(16 * i + r16) * 68 + 16 * j + 4 * g4] = t4;
            }
        const Epi ec = e;   // see epi_store8: keep the kernel's own Epi out of scratch
        p8_store_ragged(ec, C, ldc, Ct, mrow_of(hp), ncol, M, N, slab_out, lane);
    }
}
constexpr int P_EPI_LDS = 8 * (32 * 68 * 4);

template <bool A_R, bool B_R, int EK, int MIH = 4>
__global__ __launch_bounds__(512) void gemm_bf16_p8_kernel(int64_t M, int64_t N, int64_t K, const bf16* __restrict__ A, int64_t lda,
                                                           const bf16* __restrict__ B, int64_t ldb, bf16* __restrict__ C, int64_t ldc,
                                                           Epi e, int64_t ntn, int64_t kchunk, float* __restrict__ slab) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
#ifdef DVLP_STAMP
    unsigned long long st[24] = {};
    P8_STAMP(0);
#endif
    const int wg = xcd_remap32((int)blockIdx.x, (int)gridDim.x);
    int tm_, tn_;
    tile_of32(wg, (int)gridDim.x / (int)ntn, (int)ntn, tm_, tn_);
    A += blockIdx.y * e.sA; B += blockIdx.y * e.sB; C += blockIdx.y * e.sC * ((e.flags & EPI_OUT_F32) ? 2 : 1);
    if (e.res) e.res = (const bf16*)e.res + blockIdx.y * e.sRes;
    if (e.aux) e.aux = (bf16*)e.aux + blockIdx.y * e.sAux;
    const int64_t kbeg = blockIdx.z * kchunk, kend = kbeg + kchunk < K ? kbeg + kchunk : K;
    float* slab_out = slab ? slab + (int64_t)(blockIdx.y * gridDim.z + blockIdx.z) * M * N : nullptr;
#ifdef DVLP_STAMP
    p8_tile<A_R, B_R, EK, MIH>(smem_raw, M, N, A, lda, B, ldb, C, ldc, e, (int64_t)tm_ * p8_tile_rows<MIH>(), (int64_t)tn_ * 256, kbeg, (int)((kend - kbeg) / H_BK), slab_out, st);
    P8_STAMP(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    P8_STAMP(4);
    if (g_p8_stamp && (threadIdx.x == 0 || threadIdx.x == 256)) {        // one wave of each wave group (wr = 0, 1)
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* o = g_p8_stamp + 48 * (size_t)blockIdx.x + (threadIdx.x ? 24 : 0);
        o[0] = blockIdx.x; o[1] = hw; o[2] = xcc;
        for (int i = 0; i < 5; ++i) o[3 + i] = st[i];
        for (int i = 8; i < 24; ++i) o[i] = st[i];
    }
#else
    p8_tile<A_R, B_R, EK, MIH>(smem_raw, M, N, A, lda, B, ldb, C, ldc, e, (int64_t)tm_ * p8_tile_rows<MIH>(), (int64_t)tn_ * 256, kbeg, (int)((kend - kbeg) / H_BK), slab_out);
#endif
}

// Persistent form of the kernel above for outputs of more than one round of tiles (one workgroup per CU, workgroup w takes tiles
// w, w + G, ...).  What it removes per tile (tools/p8_timeline.py, K = 768: 18 us of K loop in a 24-32 us tile): the 2 us between
// entry and the first landed data (the next tile's first six units are staged by the previous tile's last phases) and the 0.6-3 us
// between two workgroups on a CU; the epilogue's stores drain under the next tile's first phases.
template <bool A_R, bool B_R, int EK, int MIH>
__global__ __launch_bounds__(512) void gemm_bf16_p8p_kernel(int64_t M, int64_t N, int64_t K, const bf16* __restrict__ A, int64_t lda,
                                                            const bf16* __restrict__ B, int64_t ldb, bf16* __restrict__ C, int64_t ldc,
                                                            Epi e, int ntm, int ntn) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int TH = p8_tile_rows<MIH>();
    const int T = ntm * ntn, G = (int)gridDim.x, w = (int)blockIdx.x;
    const int cnt = (T - w + G - 1) / G;
    const int nk = (int)(K / H_BK);
    int tm_, tn_;
    tile_of32(xcd_remap32(w, T), ntm, ntn, tm_, tn_);
    bool prev_counted = false;
#pragma unroll 1
    for (int i = 0; i < cnt; ++i) {
        int tmn = tm_, tnn = tn_;                                // last tile: stage its own first units again (never read)
        if (i + 1 < cnt) tile_of32(xcd_remap32(w + (i + 1) * G, T), ntm, ntn, tmn, tnn);
        const P8Next nx{i == 0, (int64_t)tmn * TH, (int64_t)tnn * 256, prev_counted};
        p8_tile<A_R, B_R, EK, MIH, true>(smem_raw, M, N, A, lda, B, ldb, C, ldc, e, (int64_t)tm_ * TH, (int64_t)tn_ * 256, 0, nk, nullptr, nullptr, nx);
        prev_counted = (int64_t)tm_ * TH + TH <= M;              // every wave issued its full set of epilogue operations
        tm_ = tmn; tn_ = tnn;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the units staged for nobody must land before this CU's LDS is handed on
}

// Grouped weight gradients: up to P8G_MAX independent dW_p = dY_p^T X_p products (all form R x form R, fp32 out) in ONE
// launch.  A layer's weight gradients are 9-36 output tiles each -- far too few to fill 256 CUs one at a time without
// splitting K 7-28 ways, which costs ~65 MB of fp32 slabs per product.  Together they are ~110 tiles, so a 2-3 way split
// fills the chip and the slab traffic drops ~4x.  Block ranges of the problems start at multiples of 8 so that the
// XCD-aware tile order stays valid inside each problem.
constexpr int P8G_MAX = 8, P8G_MAXP = 64;
struct P8Group {
    int count;
    int blk0[P8G_MAX + 1];                 // first block of each problem (multiples of 8), blk0[count] = grid size
    int ntn[P8G_MAX], tiles[P8G_MAX], S[P8G_MAX];
    int64_t M[P8G_MAX], N[P8G_MAX], K[P8G_MAX], lda[P8G_MAX], ldb[P8G_MAX], kchunk[P8G_MAX];
    const bf16* A[P8G_MAX];
    const bf16* B[P8G_MAX];
    float* C[P8G_MAX];                     // final fp32 destination (S == 1: written directly)
    float* slab[P8G_MAX];                  // S > 1: [S][M][N] partials
    // XCD patches (npatch > 0): patch i = up to 3 x 3 adjacent tiles of ONE K slice of one problem, run by the 9 blocks
    // {8 (9 (i / 8) + w) + i % 8 : w < 9} -- all on XCD i % 8 (blocks are dealt to XCDs round-robin), all resident at once, all walking
    // the same K range: the patch's 3 + 3 operand panels are fetched into that XCD's L2 once and shared, instead of each block
    // streaming its own two (measured with FETCH_SIZE: 1.15 GB -> see profiles per ViT layer; 455 MB are unique)
    int npatch;
    unsigned char pp[P8G_MAXP], pz[P8G_MAXP], ptm[P8G_MAXP], ptn[P8G_MAXP], ppm[P8G_MAXP], ppn[P8G_MAXP];
    // Round 6: an optimizer update riding on the CUs this launch leaves idle (a ViT layer's group is 216 blocks on 256 CUs): blocks
    // [opt_blk0, opt_blk0 + opt_nblk) run one fused HF-AdamW pass over opt_n elements -- the PREVIOUS layer's weights, whose gradients
    // were final before this launch -- instead of a launch of its own between two GEMM launches (dvlp_wgrad_grouped_ex).
    int opt_blk0, opt_nblk;
    int64_t opt_n;
    float *opt_p, *opt_m, *opt_v;
    const float *opt_g, *opt_hyper;
    bf16* opt_shadow;
};
__global__ __launch_bounds__(512) void gemm_bf16_p8_group_kernel(P8Group g, int flags) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    if (g.opt_nblk > 0 && (int)blockIdx.x >= g.opt_blk0) {            // a spare workgroup: its share of the riding optimizer update
        adamw_dev_elements(g.opt_n, g.opt_p, g.opt_g, g.opt_m, g.opt_v, g.opt_hyper, g.opt_shadow,
                           (int64_t)((int)blockIdx.x - g.opt_blk0) * 512 + threadIdx.x, (int64_t)g.opt_nblk * 512);
        return;
    }
    int p = 0, z;
    int64_t tm_, tn_;
    if (g.npatch > 0) {
        const int slot = (int)blockIdx.x >> 3, pi = ((int)blockIdx.x & 7) + 8 * (slot / 9), w = slot % 9;
        if (pi >= g.npatch) return;
        const int pm = g.ppm[pi], pn = g.ppn[pi];
        if (w >= pm * pn) return;                                   // padding block of a partial patch
        p = g.pp[pi]; z = g.pz[pi];
        tm_ = g.ptm[pi] + w % pm; tn_ = g.ptn[pi] + w / pm;
    } else {
#pragma unroll
        for (int i = 1; i < P8G_MAX; ++i) if (i < g.count && (int)blockIdx.x >= g.blk0[i]) p = i;
        const int local = (int)blockIdx.x - g.blk0[p];
        const int nblk = g.tiles[p] * g.S[p];
        if (local >= nblk) return;                                  // padding block
        const int64_t wg = xcd_remap(local, nblk);                  // blk0 is a multiple of 8, so local % 8 still names the XCD
        z = (int)(wg / g.tiles[p]);
        tile_of(wg % g.tiles[p], g.tiles[p] / g.ntn[p], g.ntn[p], tm_, tn_);
    }
    const int64_t kbeg = z * g.kchunk[p], kend = kbeg + g.kchunk[p] < g.K[p] ? kbeg + g.kchunk[p] : g.K[p];
    Epi e{nullptr, nullptr, nullptr, 0, 0, flags | EPI_OUT_F32, 1.0f, 0, 0, 0, 0, 0, 1, nullptr};
    e.vec = (g.N[p] % 4 == 0) ? 1 : 0;
    float* slab_out = g.S[p] > 1 ? g.slab[p] + (int64_t)z * g.M[p] * g.N[p] : nullptr;
    p8_tile<true, true, P8_EK_ANY>(smem_raw, g.M[p], g.N[p], g.A[p], g.lda[p], g.B[p], g.ldb[p], (bf16*)g.C[p], g.N[p], e, tm_ * 256, tn_ * 256, kbeg,
                        (int)((kend - kbeg) / H_BK), slab_out);
}
// sums the slabs of every split problem of a group into its fp32 destination (float4 per thread)
// K tail (Kt[p] = K % 64 rows the 64-deep K tiles of the group kernel do not cover -- 32 frames x 36 regions + CLS = 1153 tokens per sample
// make K = 16 x 1153): the rank-Kt product of the last rows rides in this pass, which already touches every output once.
struct P8GroupReduce { int count; int blk0[P8G_MAX + 1]; int S[P8G_MAX]; int64_t MN[P8G_MAX]; const float* slab[P8G_MAX]; float* C[P8G_MAX];
                       int Kt[P8G_MAX]; int64_t N[P8G_MAX], lda[P8G_MAX], ldb[P8G_MAX]; const bf16* tA[P8G_MAX]; const bf16* tB[P8G_MAX]; };
__global__ __launch_bounds__(256) void p8_group_reduce_kernel(P8GroupReduce g, int accumulate) {
    int p = 0;
#pragma unroll
    for (int i = 1; i < P8G_MAX; ++i) if (i < g.count && (int)blockIdx.x >= g.blk0[i]) p = i;
    const int64_t i4 = ((int64_t)((int)blockIdx.x - g.blk0[p]) * 256 + threadIdx.x) * 4;
    if (i4 >= g.MN[p]) return;
    float4 s = *(const float4*)(g.slab[p] + i4);
    for (int k = 1; k < g.S[p]; ++k) {
        const float4 v = *(const float4*)(g.slab[p] + (int64_t)k * g.MN[p] + i4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (g.Kt[p]) {
        const int64_t m = i4 / g.N[p], n = i4 % g.N[p];          // N is a multiple of 8: the four outputs share a row
        const bf16* ta = g.tA[p] + m; const bf16* tb = g.tB[p] + n;
        for (int k = 0; k < g.Kt[p]; ++k) {
            const float a = (float)ta[(int64_t)k * g.lda[p]];
            const bf16x4 b = *(const bf16x4*)(tb + (int64_t)k * g.ldb[p]);
            s.x += a * (float)b[0]; s.y += a * (float)b[1]; s.z += a * (float)b[2]; s.w += a * (float)b[3];
        }
    }
    if (accumulate) {
        const float4 c = *(const float4*)(g.C[p] + i4);
        s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
    }
    *(float4*)(g.C[p] + i4) = s;
}
constexpr int P_LDS_TOTAL = (P_EPI_LDS > P_LDS ? P_EPI_LDS : P_LDS) + 8192;      // K-loop buffers / epilogue staging + 8 KiB of per-wave bias copies (persistent form)

// second stage of a split-K GEMM: sum the S fp32 slabs of one output and apply the epilogue (8 columns per thread)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(int64_t M, int64_t N, int S, const float* __restrict__ slab, bf16* __restrict__ C,
                                                            int64_t ldc, Epi e) {
    const int64_t n8 = (N + 7) / 8;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * n8) return;
    const int64_t m = idx / n8, n = (idx % n8) * 8;
    const int64_t batch = blockIdx.y;
    C += batch * e.sC * ((e.flags & EPI_OUT_F32) ? 2 : 1);
    if (e.res) e.res = (const bf16*)e.res + batch * e.sRes;
    if (e.aux) e.aux = (bf16*)e.aux + batch * e.sAux;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < S; ++s) {
        const float* src = slab + ((batch * S + s) * M + m) * N + n;
        if ((N & 3) == 0 && n + 7 < N) {
            const float4 a = *(const float4*)src, b = *(const float4*)(src + 4);
            v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) if (n + t < N) v[t] += src[t];
        }
    }
    epi_store8(e, C, ldc, m, n, v, N);
}

// ------------------------------------------------------------------------------------------------------------------
// optional per-launch HIP-event timing (bench.py's roofline figure is measured with these, on the launch stream)
// ------------------------------------------------------------------------------------------------------------------
int g_dvlp_last_hip_error = 0;
extern "C" const char* dvlp_last_error_string() { return hipGetErrorString((hipError_t)g_dvlp_last_hip_error); }

// A/B switch between the LDS-DMA kernel (default) and the register-staged one (tools/gemm_bench.py --variant)
static int g_ablate = 0;         // tools/gemm_bench.py --ablate: 1 skip LDS-DMA issue, 2 skip LDS fragment reads, 4 skip MFMAs (TIMING ONLY)
DVLP_DEV_API int dvlp_dev_gemm_ablate(int bits) { g_ablate = bits; return DVLP_OK; }
static bool g_use_glds = true;
static int g_wide_mode = 0;      // 0: never use the 256-row tile (default: measured no faster on this path's shapes), 1: heuristic, 2: always
DVLP_DEV_API int dvlp_dev_gemm_wide_mode(int mode) { g_wide_mode = mode; return DVLP_OK; }
static int g_wgrad_ride = 1;     // grouped weight gradients: 1 = an optimizer update handed to dvlp_wgrad_grouped_ex rides on the launch's spare workgroups, 0 = its own launch
DVLP_DEV_API int dvlp_dev_wgrad_group_ride(int on) { g_wgrad_ride = on; return DVLP_OK; }
static int g_wgrad_patch = 1;    // grouped weight gradients: 1 = 3 x 3 tile patches pinned to XCDs (operand panels shared through L2), 0 = per-problem tile order
DVLP_DEV_API int dvlp_dev_wgrad_group_patches(int on) { g_wgrad_patch = on; return DVLP_OK; }
static int g_p8_mode = 1;        // 256 x 256 ping-pong kernel: 0 never, 1 where the grid suits it, 2 whenever the operands allow
DVLP_DEV_API int dvlp_dev_gemm_p8_mode(int mode) { g_p8_mode = mode; return DVLP_OK; }
static int g_p8_persist = 1;     // persistent form of the 256-row kernel on multi-round outputs: 0 off, 1 on (default; A/B: tools/p8p_bench.py)
DVLP_DEV_API int dvlp_dev_gemm_p8_persistent(int mode) { g_p8_persist = mode; return DVLP_OK; }
static int g_rb = 1;             // resident-B streaming kernel (csrc/gemm_rb.hip) for batched skinny products: 0 never, 1 where its shapes fit (default)
DVLP_DEV_API int dvlp_dev_gemm_resident_b(int on) { g_rb = on; return DVLP_OK; }
bool dvlp_gemm_rb_try(int transA, int transB, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb, void* C,
                      int64_t ldc, int64_t batch, int64_t sA, int64_t sB, int64_t sC, hipStream_t st);
static int g_p8_short = 1;       // 224-row tiles of the 256-row kernel: 0 never, 1 where they save CU-rounds (default), 2 whenever allowed
DVLP_DEV_API int dvlp_dev_gemm_p8_short_tiles(int mode) { g_p8_short = mode; return DVLP_OK; }
static int g_force_split = 0;    // dvlp_gemm: 0 = automatic K split, > 0 = forced (A/B measurements: tools/gemm_sweep.py)
DVLP_DEV_API int dvlp_dev_gemm_force_split(int s) { g_force_split = s; return DVLP_OK; }
static int g_wgrad_split = 0;    // grouped weight gradients: 0 = automatic uniform K split, > 0 = forced
DVLP_DEV_API int dvlp_dev_wgrad_group_split(int s) { g_wgrad_split = s; return DVLP_OK; }
static int64_t g_splitk_target = 768;     // workgroups a split-K launch aims for (tools/gemm_bench.py --splitk-target)
DVLP_DEV_API int dvlp_dev_gemm_splitk_target(int64_t n) { g_splitk_target = n > 0 ? n : 768; return DVLP_OK; }
DVLP_DEV_API int dvlp_dev_gemm_variant(int use_lds_dma) { g_use_glds = use_lds_dma != 0; return DVLP_OK; }

// caller-provided scratch for split-K slabs (dvlp_set_workspace); nullptr disables splitting
// One scratch buffer per stream (two GEMMs in flight on different streams must not share slabs); stream 0 entry is the
// default for any stream without its own registration.
#include <unordered_map>
#include <mutex>
#include <map>
#include <string>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
struct WsEntry { float* ptr; int64_t bytes; };
static std::unordered_map<void*, WsEntry> g_ws_map;
static std::mutex g_ws_mu;
extern "C" int dvlp_set_workspace_stream(void* stream, void* ptr, int64_t bytes) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    if (ptr) g_ws_map[stream] = WsEntry{(float*)ptr, bytes}; else g_ws_map.erase(stream);
    return DVLP_OK;
}
extern "C" int dvlp_set_workspace(void* ptr, int64_t bytes) { return dvlp_set_workspace_stream(nullptr, ptr, bytes); }
static WsEntry ws_for(void* stream) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    auto it = g_ws_map.find(stream);
    if (it == g_ws_map.end()) it = g_ws_map.find(nullptr);
    return it == g_ws_map.end() ? WsEntry{nullptr, 0} : it->second;
}

// Per-stream hint (caller-registered, like the workspaces): launches on `stream` share the chip with another stream's kernels -- the text
// tower beside the object tower (model.ObjectRelation.parallel_towers).  A co-running launch is not helped by finishing its own grid in fewer
// rounds (the other stream fills the CUs it leaves idle); what counts is CU-time per FLOP, i.e. the tallest tiles: the dispatch then keeps round
// 5's rule (256 / 224 rows, 3-way split of the text tower's K >= 2304 products) instead of the latency planner (p8_plan).  Measured in the
// replayed step, same box, alternating runs: planner on the co-running text tower 17.72 / 17.72 ms against 17.59 / 17.63 with this rule;
// single stream 18.88 / 18.92 with the planner against 18.99 / 19.02 without (profiles/r6_planner_ab.txt).
static std::unordered_map<void*, int> g_stream_hint;
extern "C" int dvlp_stream_hint(void* stream, int co_running) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    if (co_running) g_stream_hint[stream] = co_running; else g_stream_hint.erase(stream);
    return DVLP_OK;
}
static bool stream_co_running(void* stream) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    auto it = g_stream_hint.find(stream);
    return it != g_stream_hint.end() && it->second != 0;
}

struct ProfRec { hipEvent_t a, b; double flops; int64_t M, N, K, batch; int form, flags, kern; };
static bool g_prof = false;
static std::vector<ProfRec> g_recs;

extern "C" int dvlp_prof_enable(int on) {
    g_prof = on != 0;
    return DVLP_OK;
}
// Synchronises; returns the summed duration (ms), flops and count of the GEMM launches recorded since the last call.
extern "C" int dvlp_prof_collect(double* total_ms, double* total_flops, int64_t* count) {
    double ms = 0, fl = 0;
    // DVLP_PROF_REPORT=1: per-shape table on stderr (which GEMMs of the step run below the family average)
    struct Agg { double ms, fl; int64_t n; };
    std::map<std::string, Agg> table;
    const bool report = getenv("DVLP_PROF_REPORT") != nullptr;
    for (auto& r : g_recs) {
        (void)hipEventSynchronize(r.b);
        float t = 0.f;
        (void)hipEventElapsedTime(&t, r.a, r.b);
        ms += t; fl += r.flops;
        if (report) {
            char key[160];
            snprintf(key, sizeof key, "%s M=%6lld N=%5lld K=%6lld b=%4lld epi=%2d kern=%s", r.form == 0 ? "KK" : r.form == 1 ? "KR" : r.form == 3 ? "RR" : "RK",
                     (long long)r.M, (long long)r.N, (long long)r.K, (long long)r.batch, r.flags, r.kern == 3 ? "p8g " : r.kern == 2 ? "p8  " : r.kern == 1 ? "g128" : "reg ");
            Agg& a = table[key];
            a.ms += t; a.fl += r.flops; a.n += 1;
        }
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    if (report) {
        std::vector<std::pair<std::string, Agg>> rows(table.begin(), table.end());
        std::sort(rows.begin(), rows.end(), [](const auto& x, const auto& y) { return x.second.ms > y.second.ms; });
        for (auto& kv : rows)
            fprintf(stderr, "[gemm] %s  n=%5lld  total %9.3f ms  avg %8.1f us  %7.1f TFLOP/s\n", kv.first.c_str(), (long long)kv.second.n, kv.second.ms,
                    1e3 * kv.second.ms / kv.second.n, kv.second.fl / (kv.second.ms * 1e-3) / 1e12);
    }
    *total_ms = ms; *total_flops = fl; *count = (int64_t)g_recs.size();
    g_recs.clear();
    return DVLP_OK;
}

// ---- column sums of a GEMM's output (dvlp_gemm_ex, dvlp_gemm_ext::colsum): the bias gradient of the Linear that consumes it ----
float* dvlp_rd_reserve_push(int64_t P, int64_t C, float* out);      // norm.hip: deferred-reduction queue
extern "C" int dvlp_colsum(int dtype, int64_t M, int64_t N, const void* x, int64_t ld, int64_t inner, int64_t ostride, int64_t groups,
                           int64_t gstride, float* out, float* workspace, int accumulate, void* stream);
extern "C" int64_t dvlp_colsum_chunks(int64_t M);
// Optional extras of dvlp_gemm_ex (include/demovlp_hip.h: dvlp_gemm_ext), handed to the call that consumes them (rounds 2-3 armed the
// column-sum request through a thread-local "next call" setter: an exception between the two calls left a stale pointer armed).
struct dvlp_gemm_ext {
    float* colsum;          // fp32 [N]: column sums of the stored output C wanted here (batch 1, not for fp32 outputs), or NULL
    int colsum_fused;       // out: 1 = queued from inside the GEMM's epilogue (final after the deferred reductions are flushed), 0 = a plain pass ran
};

// Tile height x K split of the 256-column kernel for an under-filled or round-quantised output (round 6).  What a block costs is not
// proportional to its rows: per 64-deep K tile ~ (0.55 + 0.0016 rows) us on a lightly filled chip, up to ~1.35x that when every CU streams
// (clock and L2 share), plus a prologue + epilogue of ~ (6 + 0.028 rows) us; a K split adds the slab round trip and the reduction launch
// (~2.5 us + 1.08 us per million output elements and slice).  Constants fitted to tools/tile_sweep.py on MI355X (profiles/r6_tile_sweep.txt:
// M = 6400 / 7712 / 18496 tokens x the eight forward / dX shapes of a layer); the model only has to RANK the candidates.
struct P8Plan { int mih; int64_t S; };
static P8Plan p8_plan(int64_t M, int64_t N, int64_t K, int64_t ntn8, int ncu, bool may_split, int64_t ws_bytes) {
    const int64_t nk = K / 64;
    P8Plan best{4, 1};
    double best_c = 1e30;
    for (int h = 4; h >= 1; --h) {
        const int64_t rows = 128 + 32 * h, nb1 = cdiv(M, rows) * ntn8;
        for (int64_t S = 1; S <= (may_split ? 4 : 1); ++S) {
            if (S > 1 && (nk < 8 * S || nb1 >= 200 || S * M * N * 4 > ws_bytes)) break;
            const int64_t nb = nb1 * S, rounds = cdiv(nb, ncu);
            const double fill = (double)nb / (double)(rounds * ncu);
            const double crowd = 1.0 + 0.35 * (fill > 0.3 ? (fill - 0.3) / 0.7 : 0.0);
            const double tk = (0.55 + 0.0016 * rows) * crowd, fixed = 6.0 + 0.028 * rows;
            const double c = rounds * ((double)cdiv(nk, S) * tk + fixed) + (S > 1 ? 2.5 + 1.08e-6 * (double)S * (double)M * (double)N : 0.0);
            if (c < best_c * 0.995) { best_c = c; best = P8Plan{h, S}; }        // (ties go to the taller tile / the smaller split: tried first)
        }
    }
    return best;
}

static int gemm_batched_impl(int dtype, int transA, int transB, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
                             const void* B, int64_t ldb, void* C, int64_t ldc, const float* bias, const void* res, int64_t ldres,
                             void* aux, int64_t ldaux, int flags, float alpha, int64_t batch, int64_t strideA, int64_t strideB,
                             int64_t strideC, int64_t strideRes, int64_t strideAux, dvlp_gemm_ext* ext, void* stream) {
    dvlp_clear_status();
    float* const csum_dst = ext ? ext->colsum : nullptr;      // column sums of this GEMM's output requested
    if (ext) ext->colsum_fused = 0;
    bool csum_fused = false;
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0 || batch > 65535) return DVLP_ERR_SHAPE;
    if ((flags & (EPI_GELU | EPI_GELU_BWD | EPI_RELU_BWD)) && !aux) return DVLP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    Epi e{bias, res, aux, ldres, ldaux, flags | (g_ablate << 24), alpha, strideA, strideB, strideC, strideRes, strideAux, 0, nullptr};
    {
        const int64_t cal = (flags & EPI_OUT_F32) ? 4 : 8;      // elements per 16 bytes of C
        bool v = (ldc % cal == 0) && ((uintptr_t)C % 16 == 0) && (strideC % cal == 0);
        if (res) v = v && (ldres % 8 == 0) && ((uintptr_t)res % 16 == 0) && (strideRes % 8 == 0);
        if (aux) v = v && (ldaux % 8 == 0) && ((uintptr_t)aux % 16 == 0) && (strideAux % 8 == 0);
        if (bias) v = v && ((uintptr_t)bias % 16 == 0);
        e.vec = v ? 1 : 0;
    }
    ProfRec rec{};
    if (g_prof) {
        (void)hipEventCreate(&rec.a); (void)hipEventCreate(&rec.b);
        rec.flops = 2.0 * M * N * K * batch; rec.M = M; rec.N = N; rec.K = K; rec.batch = batch; rec.form = transA * 2 + transB; rec.flags = flags; rec.kern = 0;
        (void)hipEventRecord(rec.a, st);
    }
    if (dtype == DVLP_F32) {
        const int64_t ntm = cdiv(M, F_BM), ntn = cdiv(N, F_BN);
        const int a_vec = (lda % 4 == 0) && ((uintptr_t)A % 16 == 0) && (strideA % 4 == 0);
        const int b_vec = (ldb % 4 == 0) && ((uintptr_t)B % 16 == 0) && (strideB % 4 == 0);
        dim3 grid((unsigned)(ntm * ntn), (unsigned)batch), block(256);
#define LAUNCH_F32(AR, BR) hipLaunchKernelGGL((gemm_f32_kernel<AR, BR>), grid, block, 0, st, M, N, K, (const float*)A, lda, \
                                              (const float*)B, ldb, (float*)C, ldc, e, a_vec, b_vec, ntn)
        if (!transA && !transB) LAUNCH_F32(false, false);
        else if (!transA && transB) LAUNCH_F32(false, true);
        else if (transA && transB) LAUNCH_F32(true, true);
        else LAUNCH_F32(true, false);
#undef LAUNCH_F32
    } else if (dtype == DVLP_BF16 && g_rb && !bias && !res && !aux && flags == 0 && alpha == 1.0f && !csum_dst && !g_ablate &&
               dvlp_gemm_rb_try(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, batch, strideA, strideB, strideC, st)) {
        rec.kern = 4;             // batched skinny product with its B operand resident in LDS (the local loss' per-video / per-caption contractions)
    } else if (dtype == DVLP_BF16) {
        // 16-byte vector loads need 8-element-aligned leading dims and base pointers
        if (lda % 8 || ldb % 8 || (uintptr_t)A % 16 || (uintptr_t)B % 16 || strideA % 8 || strideB % 8) return DVLP_ERR_SHAPE;
        const int64_t ntm = cdiv(M, H_BM), ntn = cdiv(N, H_BN);
        const size_t lds = (size_t)4 * H_TILE * sizeof(bf16);
        static_assert(4 * 64 * 68 * sizeof(float) <= (size_t)4 * H_TILE * sizeof(bf16), "epilogue staging must fit the K-loop buffers");
        // the branch-free loader clamps rows: form-K operands need >= 1 row, form-R operands a row count that is a multiple of 8
        const bool safe = (transA ? (M % 8 != 0 || M < 8) : false) || (transB ? (N % 8 != 0 || N < 8) : false) || K < 8 || K % 8 != 0;
        const bool dma = !safe && g_use_glds;                 // (!safe implies K % 8 == 0; a ragged last K tile is zero-filled)
        // 256 x 256 ping-pong kernel: one workgroup per CU, so it wants >= ~a chip of tiles (or a K long enough to split)
        const int64_t ntm8 = cdiv(M, 256), ntn8 = cdiv(N, 256), tiles8 = ntm8 * ntn8 * batch;
        // ... and outputs that fill its tiles: 288 columns would leave the second 256-wide tile 7/8 empty
        const bool fills8 = 10 * M * N >= 8 * (ntm8 * 256) * (ntn8 * 256);
        // ... and a tile count that fills whole rounds of the 256 CUs (300 tiles = 2 rounds for 1.17 rounds of work); a long K
        // (>= 32 K tiles) amortises its fixed costs well enough to win even at 75 tiles
        const bool rounds8 = 10 * tiles8 >= 8 * 256 * cdiv(tiles8, 256);
        // (measured per shape with tools/gemm_sweep.py: 75-tile outputs with K = 2304 / 3072 -- the text tower's fc2, fc1 dX, qkv dX -- run
        //  faster on the 128-row kernel without a K split than on a 3-way split 256-row launch (47 vs 55, 38 vs 50, 48 vs 55 us), and a
        //  3-tile weight gradient -- the two 256-wide projections -- on 384 128-row blocks than on 96 256-row ones)
        // (round 3, in the step: with the text tower beside the object tower, its 75 / 300-tile products on the 256-row kernel -- any round
        //  count -- take 1 % off the step; alone they measured slower, above.  DVLP_P8_MIN_TILES=192 DVLP_P8_ROUNDS=1 restores round 2's rule)
        static const int p8_min_tiles = getenv("DVLP_P8_MIN_TILES") ? atoi(getenv("DVLP_P8_MIN_TILES")) : 64;      // experiments (round 3: 192 -> 64, see below)
        static const bool p8_need_rounds = getenv("DVLP_P8_ROUNDS") ? atoi(getenv("DVLP_P8_ROUNDS")) != 0 : false;
        // (round 6: batched long-K reductions whose output half-fills its tiles -- the local loss' dC^_i [288 x 256] and dQ^_j [104 x 256] products
        //  over K = 6 656 / 18 432, batch 64 -- also run faster here with a 2- / 4-way K split than on the 128-row kernel: 181 -> 132, 162 -> 122,
        //  197 -> 160 us (profiles/r6_loss_reductions_on_p8.txt); a [104 x 104] output does not: 108 -> 132)
        const bool longk8 = batch >= 8 && K >= 4096 && N % 256 == 0 && 10 * M >= 4 * ntm8 * 256;
        const bool p8 = dma && K % H_BK == 0 && g_p8_mode != 0 && (g_p8_mode == 2 || longk8 || (M >= 256 && N >= 256 && fills8 && ((tiles8 >= p8_min_tiles && (rounds8 || !p8_need_rounds)) || (tiles8 >= 8 && tiles8 <= 64 && K >= 4096))));
        // Under-filled grids with a long reduction (weight gradients: 36-144 output tiles, K = B*N tokens) are split along
        // K so that ~3 workgroups (128-row kernel) or 1 workgroup (256-row kernel) land on every CU; partials go through fp32
        // slabs (deterministic, no float atomics).
        int64_t S = 1;
        const int64_t tiles = p8 ? tiles8 : ntm * ntn * batch;
        const WsEntry wse = ws_for(stream);
        float* g_ws = wse.ptr;
        const int64_t g_ws_bytes = wse.bytes;
        static const int ncu8 = [] { int d = 0, n = 256; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n > 0 ? n : 256; }();
        // tile height (p8_tile_rows) and K split of the 256-column kernel, chosen TOGETHER where every tile takes a whole-tile store (row-major
        // A, whole 256-column tiles, aligned outputs; a split launch writes slabs, which any height can): p8_plan above.  M = 18496: 224-row
        // tiles (83 x {3, 9, 12} = 249 / 747 / 996 blocks fill 1 / 3 / 4 rounds of 256 CUs that 256-row tiles leave 14 % empty); M = 6400
        // (the text tower): 160-row tiles (120 / 480 blocks instead of 75 / 300), K >= 2304 as 160 rows x 2 slices instead of 256 x 3.
        int mih8 = 4;
        const bool plan8 = p8 && g_p8_short != 0 && !transA && batch == 1 && N % 256 == 0 && e.vec && (N & 3) == 0;
        const bool co_run = plan8 && g_p8_short == 1 && stream_co_running(stream);
        if (plan8 && g_p8_short == 1 && g_force_split == 0 && !co_run) {
            const P8Plan pl = p8_plan(M, N, K, ntn8, ncu8, g_ws != nullptr && K >= 1024, g_ws_bytes);
            mih8 = pl.mih; S = pl.S;
        } else {
            // (a co-running launch is not split: slabs and a reduction launch buy latency it does not need with CU-time the other stream does --
            //  DVLP_CORUN_SPLIT=1 restores the split for A/B runs)
            static const bool corun_split = getenv("DVLP_CORUN_SPLIT") && atoi(getenv("DVLP_CORUN_SPLIT")) != 0;
            if (g_ws && tiles < (p8 ? 129 : 200) && K >= 1024 && !(co_run && !corun_split)) {
                S = p8 ? 256 / tiles : (g_splitk_target + tiles - 1) / tiles;
                if (S > K / 256) S = K / 256;
                if (S > 32) S = 32;
                while (S > 1 && S * batch * M * N * 4 > g_ws_bytes) --S;
            }
            if (g_force_split > 0 && g_ws) { S = g_force_split; while (S > 1 && S * batch * M * N * 4 > g_ws_bytes) --S; }
            if (plan8) {           // developer switches: a forced height (10 + MIH), "224 wherever allowed" (2), round 5's rule (3)
                if (g_p8_short >= 10) mih8 = g_p8_short - 10 >= 1 && g_p8_short - 10 <= 4 ? g_p8_short - 10 : 4;
                else if (g_p8_short == 2) mih8 = 3;
                else if ((g_p8_short == 3 || co_run) && S == 1 && cdiv(cdiv(M, 224) * ntn8, ncu8) * 224 < cdiv(tiles8, ncu8) * 256) mih8 = 3;
            }
        }
        int64_t kchunk = cdiv(cdiv(K, S), H_BK) * H_BK;
        S = cdiv(K, kchunk);
        float* slab = S > 1 ? g_ws : nullptr;
        dim3 grid((unsigned)(ntm * ntn), (unsigned)batch, (unsigned)S), block(256);
        // > 64 KiB of dynamic LDS must be opted into once per kernel
#define LAUNCH_BF16_(AR, BR, SF) do { static bool once = false; if (!once) { once = true; \
            (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<AR, BR, SF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); } \
        hipLaunchKernelGGL((gemm_bf16_kernel<AR, BR, SF>), grid, block, lds, st, M, N, K, (const bf16*)A, lda, \
                           (const bf16*)B, ldb, (bf16*)C, ldc, e, ntn, kchunk, slab); } while (0)
#define LAUNCH_P8K_(AR, BR, EK, MIH) do { static bool once = false; if (!once) { once = true; \
            (void)hipFuncSetAttribute((const void*)gemm_bf16_p8_kernel<AR, BR, EK, MIH>, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS_TOTAL); } \
        hipLaunchKernelGGL((gemm_bf16_p8_kernel<AR, BR, EK, MIH>), grid8, dim3(512), (size_t)P_LDS_TOTAL, st, M, N, K, (const bf16*)A, lda, \
                           (const bf16*)B, ldb, (bf16*)C, ldc, e, ntn8, kchunk, slab); } while (0)
#define LAUNCH_P8P_(AR, BR, EK, MIH) do { static bool once = false; if (!once) { once = true; \
            (void)hipFuncSetAttribute((const void*)gemm_bf16_p8p_kernel<AR, BR, EK, MIH>, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS_TOTAL); } \
        hipLaunchKernelGGL((gemm_bf16_p8p_kernel<AR, BR, EK, MIH>), dim3((unsigned)ncu8), dim3(512), (size_t)P_LDS_TOTAL, st, M, N, K, (const bf16*)A, lda, \
                           (const bf16*)B, ldb, (bf16*)C, ldc, e, (int)ntm8h, (int)ntn8); } while (0)
#define LAUNCH_P8S_(AR, BR, EK) do { if constexpr (!AR && (EK == 0 || EK == 2)) { if (p8p) { LAUNCH_P8P_(AR, BR, EK, 3); break; } } if constexpr (!AR) { \
        if (mih8 == 3) { LAUNCH_P8K_(AR, BR, EK, 3); break; } if (mih8 == 2) { LAUNCH_P8K_(AR, BR, EK, 2); break; } if (mih8 == 1) { LAUNCH_P8K_(AR, BR, EK, 1); break; } } \
        LAUNCH_P8K_(AR, BR, EK, 4); } while (0)
#define LAUNCH_P8_(AR, BR) do { if (ek8 == 0) LAUNCH_P8S_(AR, BR, 0); else if (ek8 == 1) LAUNCH_P8S_(AR, BR, 1); else if (ek8 == 2) LAUNCH_P8S_(AR, BR, 2); \
        else if (ek8 == 3) LAUNCH_P8S_(AR, BR, 3); else if (mih8 != 4) LAUNCH_P8S_(AR, BR, 0); else LAUNCH_P8K_(AR, BR, P8_EK_ANY, 4); } while (0)
#define LAUNCH_GLDS_(AR, BR) do { if (p8) LAUNCH_P8_(AR, BR); else if (wide) { static bool once = false; if (!once) { once = true; \
            (void)hipFuncSetAttribute((const void*)gemm_bf16_glds_kernel<AR, BR, 4, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (256 + 128) * 128); } \
        hipLaunchKernelGGL((gemm_bf16_glds_kernel<AR, BR, 4, 3>), gridw, dim3(512), (size_t)3 * (256 + 128) * 128, st, M, N, K, (const bf16*)A, lda, \
                           (const bf16*)B, ldb, (bf16*)C, ldc, e, ntn, kchunk, slab); } else { static bool once = false; if (!once) { once = true; \
            (void)hipFuncSetAttribute((const void*)gemm_bf16_glds_kernel<AR, BR, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); } \
        hipLaunchKernelGGL((gemm_bf16_glds_kernel<AR, BR, 2, 2>), grid, block, lds, st, M, N, K, (const bf16*)A, lda, \
                           (const bf16*)B, ldb, (bf16*)C, ldc, e, ntn, kchunk, slab); } } while (0)
        // wide (256 x 128) tiles when they still give every CU at least ~2 workgroups; A form R needs M % 8 (guaranteed by !safe)
        const int64_t ntm_w = cdiv(M, 256);
        const bool wide = dma && !p8 && g_wide_mode != 0 && (g_wide_mode == 2 || ntm_w * ntn * batch * S >= 512);
        dim3 gridw((unsigned)(ntm_w * ntn), (unsigned)batch, (unsigned)S);
        // epilogue kind of the 256-row kernel (see p8_tile)
        const int fmask8 = flags & (EPI_GELU | EPI_GELU_BWD | EPI_RELU_BWD | EPI_ACCUM | EPI_OUT_F32);
        // (a specialised kind only where every tile is whole: N a multiple of 256 and, unless the launch writes split-K slabs, 16-byte aligned outputs)
        const bool whole8 = N % 256 == 0 && (S > 1 ? N % 4 == 0 : e.vec != 0);
        const int ek8 = (g_ablate || !whole8) ? P8_EK_ANY : (S > 1 || fmask8 == 0) ? ((res && S == 1) ? 1 : 0) : (fmask8 == EPI_GELU && !res) ? 2 : (fmask8 == EPI_GELU_BWD && !res) ? 3 : P8_EK_ANY;
        if (ek8 == P8_EK_ANY && S == 1) mih8 = 4;        // run-time-flag epilogues exist at 256 rows only
        // a many-round product with a very short K (the local loss' S = C^ Q^^T: 1 872 tiles of 4 K tiles each) is all prologue and epilogue: the
        // persistent form (224-row tiles only) hides them behind the neighbouring tiles' K loops -- 105 -> 91.5 us -- whatever the planner's height
        if (plan8 && g_p8_short == 1 && g_p8_persist != 0 && S == 1 && (ek8 == 0 || ek8 == 2) && K / H_BK >= 4 && K / H_BK <= 8 && (K / H_BK) % 2 == 0 &&
            cdiv(M, 224) * ntn8 > 2 * ncu8) mih8 = 3;
        const int64_t rows8 = 128 + 32 * mih8;
        const int64_t ntm8h = cdiv(M, rows8);
        if (csum_dst && p8 && batch == 1 && S == 1 && e.vec && N % 256 == 0 && !(flags & EPI_OUT_F32)) {
            e.csum = dvlp_rd_reserve_push(2 * ntm8h, N, csum_dst);     // partial rows: (row tile, upper / lower wave group)
            csum_fused = e.csum != nullptr;
        }
        dim3 grid8((unsigned)(ntm8h * ntn8), (unsigned)batch, (unsigned)S);
        // persistent form: more than one round of whole tiles, an even number (>= 4) of K tiles, a specialised epilogue, no fused column sums
        const bool p8p = p8 && g_p8_persist != 0 && mih8 == 3 && M * lda < (1ll << 31) && N * ldb < (1ll << 31) && K * ldb < (1ll << 31) && !transA && S == 1 && batch == 1 && ntm8h * ntn8 > ncu8 && N % 256 == 0 && e.vec && alpha == 1.0f &&
                         (K / H_BK) % 2 == 0 && K / H_BK >= 4 && (ek8 == 0 || ek8 == 2) && !e.csum;      // (residual / aux-in epilogues: the compiler closes their loads with vmcnt(0) inside the tile loop)
#define LAUNCH_BF16(AR, BR) do { if (dma) LAUNCH_GLDS_(AR, BR); else if (safe) LAUNCH_BF16_(AR, BR, true); else LAUNCH_BF16_(AR, BR, false); } while (0)
        rec.kern = p8 ? 2 : dma ? 1 : 0;
        if (!transA && !transB) LAUNCH_BF16(false, false);
        else if (!transA && transB) LAUNCH_BF16(false, true);
        else if (transA && transB) LAUNCH_BF16(true, true);
        else LAUNCH_BF16(true, false);
        if (S > 1)
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cdiv(M * cdiv(N, 8), 256), (unsigned)batch), dim3(256), 0, st, M, N, (int)S,
                               (const float*)slab, (bf16*)C, ldc, e);
#undef LAUNCH_BF16_
#undef LAUNCH_GLDS_
#undef LAUNCH_P8_
#undef LAUNCH_P8S_
#undef LAUNCH_P8P_
#undef LAUNCH_P8K_
#undef LAUNCH_BF16
    } else {
        return DVLP_ERR_DTYPE;
    }
    if (g_prof) { (void)hipEventRecord(rec.b, st); g_recs.push_back(rec); }
    if (ext) ext->colsum_fused = csum_fused ? 1 : 0;
    if (csum_dst && !csum_fused) {            // not fused: a plain column-sum pass over the stored output (batch 1 only)
        if (int rc = dvlp_launch_status()) return rc;
        const WsEntry w2 = ws_for(stream);
        if (batch != 1 || (flags & EPI_OUT_F32) || !w2.ptr || dvlp_colsum_chunks(M) * N * 4 > w2.bytes) return DVLP_ERR_SHAPE;
        return dvlp_colsum(dtype, M, N, C, ldc, M, 0, 1, 0, csum_dst, w2.ptr, 0, stream);
    }
    return dvlp_launch_status();
}

extern "C" int dvlp_gemm_batched(int dtype, int transA, int transB, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
                                 const void* B, int64_t ldb, void* C, int64_t ldc, const float* bias, const void* res, int64_t ldres,
                                 void* aux, int64_t ldaux, int flags, float alpha, int64_t batch, int64_t strideA, int64_t strideB,
                                 int64_t strideC, int64_t strideRes, int64_t strideAux, void* stream) {
    return gemm_batched_impl(dtype, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, res, ldres, aux, ldaux, flags, alpha, batch, strideA, strideB,
                             strideC, strideRes, strideAux, nullptr, stream);
}

extern "C" int dvlp_gemm(int dtype, int transA, int transB, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
                         const void* B, int64_t ldb, void* C, int64_t ldc, const float* bias, const void* res, int64_t ldres,
                         void* aux, int64_t ldaux, int flags, float alpha, void* stream) {
    return gemm_batched_impl(dtype, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, res, ldres, aux, ldaux, flags, alpha, 1, 0, 0,
                             0, 0, 0, nullptr, stream);
}

// dvlp_gemm with extras (dvlp_gemm_ext; NULL = none)
extern "C" int dvlp_gemm_ex(int dtype, int transA, int transB, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
                            const void* B, int64_t ldb, void* C, int64_t ldc, const float* bias, const void* res, int64_t ldres,
                            void* aux, int64_t ldaux, int flags, float alpha, dvlp_gemm_ext* ext, void* stream) {
    return gemm_batched_impl(dtype, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, res, ldres, aux, ldaux, flags, alpha, 1, 0, 0,
                             0, 0, 0, ext, stream);
}

// Weight gradients of several linears in one go: dW_p[M_p, N_p] (fp32, contiguous) (+)= dY_p[K_p, M_p]^T X_p[K_p, N_p].
// bf16 operands that suit the 256 x 256 kernel run as ONE grouped launch (+ one slab reduction); anything else falls back
// to per-problem dvlp_gemm calls with identical results.
// An HF-AdamW range update (dvlp_adamw_range_dev's arguments) handed to dvlp_wgrad_grouped_ex: run on the workgroups the grouped launch leaves
// idle when there are at least 16 of them, else as a launch of its own right behind it.  `fused` (out): 1 = it rode along.
struct dvlp_wgrad_ext {
    int64_t n;
    float* p; const float* g; float* m; float* v; const float* hyper; void* bf16_shadow;
    int fused;
};
extern "C" int dvlp_adamw_range_dev(int64_t n, float* p, const float* g, float* m, float* v, const float* hyper, void* bf16_shadow, void* stream);

static int wgrad_grouped_impl(int dtype, int count, const int64_t* M, const int64_t* N, const int64_t* K, const void* const* dY,
                              const int64_t* ld_dy, const void* const* X, const int64_t* ld_x, void* const* dW, int accumulate,
                              dvlp_wgrad_ext* ext, void* stream);

extern "C" int dvlp_wgrad_grouped(int dtype, int count, const int64_t* M, const int64_t* N, const int64_t* K, const void* const* dY,
                                  const int64_t* ld_dy, const void* const* X, const int64_t* ld_x, void* const* dW, int accumulate,
                                  void* stream) {
    return wgrad_grouped_impl(dtype, count, M, N, K, dY, ld_dy, X, ld_x, dW, accumulate, nullptr, stream);
}
extern "C" int dvlp_wgrad_grouped_ex(int dtype, int count, const int64_t* M, const int64_t* N, const int64_t* K, const void* const* dY,
                                     const int64_t* ld_dy, const void* const* X, const int64_t* ld_x, void* const* dW, int accumulate,
                                     dvlp_wgrad_ext* ext, void* stream) {
    if (ext) {
        ext->fused = 0;
        if (ext->n <= 0 || !ext->hyper || (((uintptr_t)ext->p | (uintptr_t)ext->g | (uintptr_t)ext->m | (uintptr_t)ext->v) & 15) || ((uintptr_t)ext->bf16_shadow & 7))
            return DVLP_ERR_SHAPE;
    }
    const int rc = wgrad_grouped_impl(dtype, count, M, N, K, dY, ld_dy, X, ld_x, dW, accumulate, ext, stream);
    if (rc != DVLP_OK || !ext || ext->fused) return rc;
    return dvlp_adamw_range_dev(ext->n, ext->p, ext->g, ext->m, ext->v, ext->hyper, ext->bf16_shadow, stream);     // no room in the launch: its own
}

static int wgrad_grouped_impl(int dtype, int count, const int64_t* M, const int64_t* N, const int64_t* K, const void* const* dY,
                              const int64_t* ld_dy, const void* const* X, const int64_t* ld_x, void* const* dW, int accumulate,
                              dvlp_wgrad_ext* ext, void* stream) {
    if (count <= 0) return DVLP_OK;
    const int flags = EPI_OUT_F32 | (accumulate ? EPI_ACCUM : 0);
    bool group_ok = dtype == DVLP_BF16 && g_p8_mode != 0 && g_use_glds && count <= P8G_MAX && count > 1;
    int64_t T = 0;
    for (int p = 0; group_ok && p < count; ++p) {
        // (K need not be a multiple of the 64-deep K tile: the group covers Km = K - K % 64 rows, the slab reduction adds the rest)
        group_ok = M[p] >= 256 && N[p] >= 256 && M[p] % 8 == 0 && N[p] % 8 == 0 && K[p] >= 1024 && ld_dy[p] % 8 == 0 &&
                   ld_x[p] % 8 == 0 && (uintptr_t)dY[p] % 16 == 0 && (uintptr_t)X[p] % 16 == 0 && (uintptr_t)dW[p] % 16 == 0;
        T += cdiv(M[p], 256) * cdiv(N[p], 256);
    }
    int64_t Km[P8G_MAX];
    bool tails = false;
    for (int p = 0; group_ok && p < count; ++p) { Km[p] = K[p] - K[p] % H_BK; tails = tails || Km[p] != K[p]; }
    // a problem with a K tail needs the slab reduction (at least two K slices): the chip-filling split below gives that when T <= 128
    if (tails && (T > 128 || g_wgrad_split == 1)) group_ok = false;
    const WsEntry wse = ws_for(stream);
    if (!group_ok || T > 256 || !wse.ptr) {
        for (int p = 0; p < count; ++p) {
            const int rc = dvlp_gemm(dtype, 1, 1, M[p], N[p], K[p], dY[p], ld_dy[p], X[p], ld_x[p], dW[p], N[p], nullptr, nullptr, 0, nullptr, 0,
                                     flags, 1.0f, stream);
            if (rc != DVLP_OK) return rc;
        }
        return DVLP_OK;
    }
    dvlp_clear_status();
    hipStream_t st = (hipStream_t)stream;
    P8Group g{};
    g.count = count;
    int64_t S[P8G_MAX], tiles[P8G_MAX], total = 0;
    for (int p = 0; p < count; ++p) {
        tiles[p] = cdiv(M[p], 256) * cdiv(N[p], 256);
        S[p] = 256 / T > 0 ? 256 / T : 1;
        if (S[p] > Km[p] / 256) S[p] = Km[p] / 256;
        total += tiles[p] * S[p];
    }
    // The same K split for every problem of the group (they share K = batch x tokens): measured on MI355X, giving the CUs left over
    // by the uniform split to one problem as a third K slice (108 tiles: S = 3,2,2,2 -> 252 blocks instead of 216) made the whole
    // launch SLOWER -- 403 us against 309 us for a ViT layer's four products, 166 against 122 for a DistilBERT layer's -- although
    // it shortens a third of the blocks: tools/wgrad_bench.py.  g_wgrad_split > 0 forces a split (A/B measurements).
    // a co-running group (the text tower's, on its own stream beside the object tower: dvlp_stream_hint) is not split: whole-K blocks on fewer
    // CUs cost less CU-time than twice the blocks plus slabs plus a reduction launch, and the other stream uses what is left
    // (DVLP_CORUN_WGRAD_SPLIT=1 restores the split for A/B runs)
    static const bool corun_wsplit = getenv("DVLP_CORUN_WGRAD_SPLIT") && atoi(getenv("DVLP_CORUN_WGRAD_SPLIT")) != 0;
    if (!tails && !corun_wsplit && g_wgrad_split == 0 && stream_co_running(stream))
        for (int p = 0; p < count; ++p) S[p] = 1;
    if (g_wgrad_split > 0)
        for (int p = 0; p < count; ++p) { S[p] = g_wgrad_split; if (S[p] > Km[p] / 256) S[p] = Km[p] / 256 > 0 ? Km[p] / 256 : 1; }
    // slabs for every split problem must fit the split-K workspace; otherwise split less
    for (;;) {
        int64_t need = 0;
        for (int p = 0; p < count; ++p) if (S[p] > 1) need += S[p] * M[p] * N[p] * 4;
        if (need <= wse.bytes) break;
        int worst = 0;
        for (int p = 1; p < count; ++p) if (S[p] * M[p] * N[p] > S[worst] * M[worst] * N[worst]) worst = p;
        if (S[worst] == 1 || (tails && S[worst] == 2)) break;
        S[worst] -= 1;
    }
    if (tails) {
        int64_t need = 0;
        for (int p = 0; p < count; ++p) need += S[p] * M[p] * N[p] * 4;
        bool ok = need <= wse.bytes;
        for (int p = 0; p < count; ++p) ok = ok && S[p] >= 2;
        if (!ok) {                                       // (workspace too small for two slices of everything: the one-product-at-a-time path)
            for (int p = 0; p < count; ++p) {
                const int rc = dvlp_gemm(dtype, 1, 1, M[p], N[p], K[p], dY[p], ld_dy[p], X[p], ld_x[p], dW[p], N[p], nullptr, nullptr, 0, nullptr, 0,
                                         flags, 1.0f, stream);
                if (rc != DVLP_OK) return rc;
            }
            return DVLP_OK;
        }
    }
    P8GroupReduce r{};
    int blk = 0, rblk = 0, nred = 0;
    float* slab = wse.ptr;
    double flops = 0;
    for (int p = 0; p < count; ++p) {
        const int64_t kchunk = cdiv(cdiv(Km[p], S[p]), H_BK) * H_BK;
        S[p] = cdiv(Km[p], kchunk);
        g.blk0[p] = blk;
        g.ntn[p] = (int)cdiv(N[p], 256); g.tiles[p] = (int)tiles[p]; g.S[p] = (int)S[p];
        g.M[p] = M[p]; g.N[p] = N[p]; g.K[p] = Km[p]; g.lda[p] = ld_dy[p]; g.ldb[p] = ld_x[p]; g.kchunk[p] = kchunk;
        g.A[p] = (const bf16*)dY[p]; g.B[p] = (const bf16*)X[p]; g.C[p] = (float*)dW[p];
        g.slab[p] = S[p] > 1 ? slab : nullptr;
        blk += (int)((tiles[p] * S[p] + 7) / 8 * 8);
        flops += 2.0 * M[p] * N[p] * K[p];
        if (S[p] > 1) {
            r.blk0[nred] = rblk; r.S[nred] = (int)S[p]; r.MN[nred] = M[p] * N[p]; r.slab[nred] = slab; r.C[nred] = (float*)dW[p];
            r.Kt[nred] = (int)(K[p] - Km[p]); r.N[nred] = N[p]; r.lda[nred] = ld_dy[p]; r.ldb[nred] = ld_x[p];
            r.tA[nred] = (const bf16*)dY[p] + Km[p] * ld_dy[p]; r.tB[nred] = (const bf16*)X[p] + Km[p] * ld_x[p];
            rblk += (int)cdiv(M[p] * N[p], 1024);
            slab += S[p] * M[p] * N[p];
            ++nred;
        }
    }
    g.blk0[count] = blk;
    r.count = nred; r.blk0[nred] = rblk;
    // XCD patches (see P8Group): largest first, dealt round-robin, so every XCD gets the same number of blocks where the shapes allow
    g.npatch = 0;
    if (g_wgrad_patch) {
        struct Pt { int p, z, tm, tn, pm, pn; };
        std::vector<Pt> pts;
        bool fits = true;
        for (int p = 0; p < count && fits; ++p) {
            const int ntm = (int)cdiv(M[p], 256), ntn = (int)cdiv(N[p], 256);
            fits = ntm < 256 && ntn < 256 && S[p] < 256;
            for (int z = 0; z < (int)S[p]; ++z)
                for (int tm = 0; tm < ntm; tm += 3)
                    for (int tn = 0; tn < ntn; tn += 3) pts.push_back(Pt{p, z, tm, tn, ntm - tm < 3 ? ntm - tm : 3, ntn - tn < 3 ? ntn - tn : 3});
        }
        if (fits && (int)pts.size() <= P8G_MAXP) {
            std::stable_sort(pts.begin(), pts.end(), [](const Pt& a, const Pt& b) { return a.pm * a.pn > b.pm * b.pn; });
            g.npatch = (int)pts.size();
            for (int i = 0; i < g.npatch; ++i) {
                g.pp[i] = (unsigned char)pts[i].p; g.pz[i] = (unsigned char)pts[i].z; g.ptm[i] = (unsigned char)pts[i].tm; g.ptn[i] = (unsigned char)pts[i].tn;
                g.ppm[i] = (unsigned char)pts[i].pm; g.ppn[i] = (unsigned char)pts[i].pn;
            }
            blk = 72 * (int)cdiv(g.npatch, 8);
        }
    }
    ProfRec rec{};
    if (g_prof) {
        (void)hipEventCreate(&rec.a); (void)hipEventCreate(&rec.b);
        rec.flops = flops; rec.M = T; rec.N = count; rec.K = K[0]; rec.batch = 1; rec.form = 3; rec.flags = flags; rec.kern = 3;
        (void)hipEventRecord(rec.a, st);
    }
    { static bool once = false; if (!once) { once = true;
        (void)hipFuncSetAttribute((const void*)gemm_bf16_p8_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS_TOTAL); } }
    g.opt_nblk = 0;
    if (ext && g_wgrad_ride) {
        static const int ncu = [] { int d = 0, n = 256; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n > 0 ? n : 256; }();
        int spare = ncu - blk % ncu;                 // CUs the last round of GEMM blocks leaves idle (one workgroup per CU: 136 KB of LDS each)
        if (spare == ncu) spare = 0;
        if (spare > 64) spare = 64;                  // (a co-running, unsplit text-tower group leaves more than the update can use)
        // ... and only when the update does not become the launch's long pole: a CU streams ~22 GB/s through the 30 B per parameter of the
        // pass, the products take ~1.3 us per K tile (+ ~15 us): a ViT layer's group (145 K tiles: ~205 us) covers its 241-us update on 40
        // CUs, a DistilBERT layer's 2-way split group (50 K tiles: ~80 us) does not -- it ran 146 us with the update inside against 103 + 35
        // apart (profiles/r6_mid_gemm_shapes.txt).  On a co-running stream the launch's own length is not what counts: always ride.
        int64_t kt = 0;
        for (int p = 0; p < count; ++p) kt = std::max<int64_t>(kt, cdiv(Km[p] / H_BK, S[p]));
        const double gemm_us = 15.0 + 1.3 * (double)kt, upd_us = spare > 0 ? (double)ext->n * 30.0 / ((double)spare * 22e3) : 1e30;
        if (spare >= 16 && (upd_us <= 1.3 * gemm_us || stream_co_running(stream))) {
            g.opt_blk0 = blk; g.opt_nblk = spare; g.opt_n = ext->n;
            g.opt_p = ext->p; g.opt_g = ext->g; g.opt_m = ext->m; g.opt_v = ext->v; g.opt_hyper = ext->hyper; g.opt_shadow = (bf16*)ext->bf16_shadow;
            blk += spare;
            ext->fused = 1;
        }
    }
    hipLaunchKernelGGL(gemm_bf16_p8_group_kernel, dim3((unsigned)blk), dim3(512), (size_t)P_LDS_TOTAL, st, g, (accumulate ? EPI_ACCUM : 0) | (g_ablate << 24));
    if (nred) hipLaunchKernelGGL(p8_group_reduce_kernel, dim3((unsigned)rblk), dim3(256), 0, st, r, accumulate);
    if (g_prof) { (void)hipEventRecord(rec.b, st); g_recs.push_back(rec); }
    return dvlp_launch_status();
}

