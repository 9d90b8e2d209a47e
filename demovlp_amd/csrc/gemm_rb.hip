// Batched skinny products with the WHOLE B operand resident in LDS ("resident-B"): C_b[M, N] = A_b[M, K] . B_b^T with N, K <= 288 and M in the
// thousands -- the per-video / per-caption contractions of the local loss (model/loss.py:262-269 and their gradients; csrc/xattn.hip):
//
//   wc[i]  [(Bj Wp) x d] = P1[i]  [(Bj Wp) x G] . C^_i [G x d]        N = 256, K = G   (B stored [K][N]: form R)
//   dP1[i] [(Bj Wp) x G] = dwc[i] [(Bj Wp) x d] . C^_i^T               N = G,   K = 256 (B stored [N][K]: form K)
//   T[j]   [(Bi G) x Wp] = P2[j]  [(Bi G) x Wp] . Kq[j]                N = K = Wp
//
// They are HBM-streaming products (95-98 us each at 5 TB/s for B = 64, G = 288, Wp = 104); on the 128 x 128 LDS-DMA tile kernel they ran at
// 2.4-3.4 TB/s (146-204 us): a 128 x 128 tile walks K = 104..288 in 2-5 steps, so a workgroup is all prologue and epilogue, and each
// re-stages its slice of B.  Here a workgroup (8 waves, one per CU: the panel takes up to 152 KB) stages B_b once -- padded rows,
// conflict-free 16-byte fragment reads -- and then STREAMS rows of A: every wave takes 16 rows per iteration, loads them from global memory
// directly in MFMA fragment shape (buffer loads: reads beyond a row's K columns meet zero rows of the panel, reads beyond the tensor return
// zero), keeps the next iteration's fragments in flight while it multiplies, and stores 16-byte pieces.  The products are issued with the
// B fragment as the first MFMA operand (the accumulator tile is C^T: a lane holds four consecutive columns of one row), two 16-column
// blocks are paired with v_permlane16_swap so that a lane owns 8 consecutive columns, as in the 256-row kernel's epilogue (csrc/gemm.hip).
#include "common.h"

typedef unsigned u32x4_rb __attribute__((ext_vector_type(4)));

struct RbArgs {
    const bf16* A; const bf16* B; bf16* C;
    int64_t M, N, K, lda, ldb, ldc, sA, sB, sC;
    int splits;                 // workgroups per batch entry (each streams a contiguous range of 128-row groups)
};

// NB: 16-column blocks of the panel (even), KS: 32-deep k-steps; B_R: B stored [K][N] (transposed while staging)
template <int NB, int KS, bool B_R>
__global__ __launch_bounds__(512) void gemm_rb_kernel(RbArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LDK = KS * 32 + 8;                 // panel row stride in elements: 16-byte skew per row -> conflict-free ds_read_b128
    bf16* panel = (bf16*)smem;                       // [NB * 16][LDK]
    const int tid = threadIdx.x, lane = tid & 63, lc = lane & 15, lq = lane >> 4;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t b = blockIdx.y;
    const bf16* A = a.A + b * a.sA;
    const bf16* B = a.B + b * a.sB;
    bf16* C = a.C + b * a.sC;
    const int N = (int)a.N, K = (int)a.K;

    // ---- stage the panel: panel[n][k] = B(n, k), zero beyond (N, K)
    if constexpr (!B_R) {
        constexpr int CH = LDK / 8 - 1;              // 16-byte pieces per row that hold data columns (KS * 4)
        for (int p = tid; p < NB * 16 * CH; p += 512) {
            const int n = p / CH, c = p % CH;
            u32x4_rb v = {0u, 0u, 0u, 0u};
            if (n < N && c * 8 < K) {                 // K is a multiple of 8 (dispatch)
                v = *(const u32x4_rb*)(B + (int64_t)n * a.ldb + c * 8);
            }
            *(u32x4_rb*)(panel + n * LDK + c * 8) = v;
        }
    } else {
        // B stored [K][N]: a thread takes 8 consecutive n of one k (16 bytes) and scatters them down a column of the panel
        constexpr int NC = NB * 2;                   // 8-column groups
        for (int p = tid; p < KS * 32 * NC; p += 512) {
            const int k = p / NC, c = p % NC;
            bf16x8 v = {};
            if (k < K && c * 8 < N) v = *(const bf16x8*)(B + (int64_t)k * a.ldb + c * 8);
#pragma unroll
            for (int t = 0; t < 8; ++t) panel[(c * 8 + t) * LDK + k] = v[t];
        }
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(A), 0, (int)(a.M * a.lda * 2 < 0x7fffffffll ? a.M * a.lda * 2 : 0x7fffffff), 0x00020000);
    // this workgroup's range of 128-row groups
    const int64_t groups = (a.M + 127) / 128;
    const int64_t g0 = groups * blockIdx.x / a.splits, g1 = groups * (blockIdx.x + 1) / a.splits;
    const int ldab = (int)a.lda * 2;

    auto load_frags = [&](int64_t grp, bf16x8 (&af)[KS]) {
        int64_t m = grp * 128 + wid * 16 + lc;
        m = m < a.M ? m : a.M - 1;
        const int voff = (int)m * ldab + lq * 16;
#pragma unroll
        for (int s = 0; s < KS; ++s) af[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ra, voff + s * 64, 0, 0));
    };
    const bf16* prow = panel + lc * LDK + lq * 8;     // this lane's fragment origin: row (16 nb + lc), k = 32 s + 8 lq

    bf16x8 cur[KS], nxt[KS];
    if (g0 < g1) load_frags(g0, cur);
    for (int64_t grp = g0; grp < g1; ++grp) {
        if (grp + 1 < g1) load_frags(grp + 1, nxt);
        f32x4 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            // (scheduling fences per half k-step: left alone, the compiler hoists all NB x KS panel reads above the products and spills ~400 registers)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const bf16x8 bf = *(const bf16x8*)(prow + nb * 16 * LDK + s * 32);
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf, cur[s], acc[nb], 0, 0, 0);      // C^T: D[n = 16 nb + 4 lq + r][m = lc]
                if ((nb & 7) == 7 || nb == NB - 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
        // store: lane (lc, lq) holds C[m = lc][16 nb + 4 lq + r]; pair blocks (2 h, 2 h + 1) -> 8 consecutive columns per lane
        const int64_t m = grp * 128 + wid * 16 + lc;
        const int c0 = 16 * (lq & 1) + 8 * (lq >> 1);
#pragma unroll
        for (int h = 0; h < NB / 2; ++h) {
            float v[8];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[2 * h][t]), __float_as_uint(acc[2 * h + 1][t]), false, false);
                v[t] = __uint_as_float(sw[0]); v[4 + t] = __uint_as_float(sw[1]);
            }
            const int n = 32 * h + c0;
            if (m < a.M && n < N) {                   // N is a multiple of 8 (dispatch): a group of 8 columns is whole or absent
                bf16x8 o;
#pragma unroll
                for (int t = 0; t < 8; ++t) o[t] = (bf16)v[t];
                *(bf16x8*)(C + m * a.ldc + n) = o;
            }
        }
        if (grp + 1 < g1) {
#pragma unroll
            for (int s = 0; s < KS; ++s) cur[s] = nxt[s];
        }
    }
}

template <int NB, int KS, bool B_R>
static void rb_launch(const RbArgs& a, int64_t batch, hipStream_t st) {
    constexpr int LDS = NB * 16 * (KS * 32 + 8) * 2;
    static bool once = false;
    if (!once) { once = true; (void)hipFuncSetAttribute((const void*)gemm_rb_kernel<NB, KS, B_R>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); }
    hipLaunchKernelGGL((gemm_rb_kernel<NB, KS, B_R>), dim3((unsigned)a.splits, (unsigned)batch), dim3(512), (size_t)LDS, st, a);
}

// Returns true when the product was launched here (bf16, plain epilogue, shapes the panel kernel is built for); false: the caller's
// ordinary dispatch takes it.
bool dvlp_gemm_rb_try(int transA, int transB, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb, void* C,
                      int64_t ldc, int64_t batch, int64_t sA, int64_t sB, int64_t sC, hipStream_t st) {
    if (transA || batch < 8 || M < 2048 || N % 8 || K % 8 || N > 288 || K > 288 || lda % 8 || ldb % 8 || ldc % 8 || sA % 8 || sB % 8 || sC % 8) return false;
    if ((uintptr_t)A % 16 || (uintptr_t)B % 16 || (uintptr_t)C % 16 || M * lda * 2 >= 0x7fffffffll) return false;
    static const int ncu = [] { int d = 0, n = 256; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n > 0 ? n : 256; }();
    RbArgs a{(const bf16*)A, (const bf16*)B, (bf16*)C, M, N, K, lda, ldb, ldc, sA, sB, sC, 1};
    const int64_t groups = (M + 127) / 128;
    int64_t sp = (ncu + batch - 1) / batch;          // one workgroup per CU in all, at least 4 row groups each
    if (sp > groups / 4) sp = groups / 4;
    a.splits = (int)(sp < 1 ? 1 : sp);
    if (!transB) {
        if (N <= 128 && K <= 128) rb_launch<8, 4, false>(a, batch, st);
        else if (N <= 256 && K <= 256) rb_launch<16, 8, false>(a, batch, st);
        else if (N <= 288 && K <= 256) rb_launch<18, 8, false>(a, batch, st);
        else return false;
    } else {
        if (N <= 256 && K <= 256) rb_launch<16, 8, true>(a, batch, st);
        else if (N <= 256 && K <= 288) rb_launch<16, 9, true>(a, batch, st);
        else return false;
    }
    return true;
}
