// Tower prologues: region-feature split/cast + token assembly (A2, model/object_transformer.py:400-433), DistilBERT
// embedding gather + LayerNorm (A8), dtype casts.  All HBM-bound element-wise work; 8/16-byte accesses where rows allow.
#include "common.h"

constexpr int FEAT = 2048, BOX = 6, EMB = 768;

// obj [M, 2054] f32 -> feat [M, 2048] T (contiguous, 16-B aligned rows for the embed GEMM) + box [M, 6] f32
template <typename T>
__global__ void obj_split_kernel(int64_t M, const float* __restrict__ obj, T* __restrict__ feat, float* __restrict__ box) {
    const int64_t m = blockIdx.x;
    const float* src = obj + m * (FEAT + BOX);
    // rows are only 8-byte aligned (2054 * 4 B): float2 loads
    for (int i = threadIdx.x; i < FEAT / 2; i += blockDim.x) {
        const float2 v = *(const float2*)(src + 2 * i);
        feat[m * FEAT + 2 * i] = from_f<T>(v.x);
        feat[m * FEAT + 2 * i + 1] = from_f<T>(v.y);
    }
    if (threadIdx.x < BOX) box[m * BOX + threadIdx.x] = src[FEAT + threadIdx.x];
}

// x[b,0,:] = cls + pos0 ; x[b,1+t,:] = tok[b*FR+t,:] + Wp box + bp + temporal[t / R]   (tok already holds W_o feat + b_o)
// addmask[b,n] = (mask01 - 1) * 100 with CLS = 0
template <typename T>
__global__ void embed_assemble_kernel(int B, int F, int R, const T* __restrict__ tok, const float* __restrict__ box,
                                      const float* __restrict__ Wp, const float* __restrict__ bp, const float* __restrict__ temporal,
                                      const float* __restrict__ cls, const float* __restrict__ pos0, const float* __restrict__ mask01,
                                      T* __restrict__ x, float* __restrict__ addmask) {
    const int N = 1 + F * R;
    const int64_t row = blockIdx.x;                // b*N + n
    const int b = (int)(row / N), n = (int)(row % N);
    if (n == 0) {
        for (int d = threadIdx.x; d < EMB; d += blockDim.x) x[row * EMB + d] = from_f<T>(cls[d] + pos0[d]);
        if (threadIdx.x == 0) addmask[row] = 0.f;
        return;
    }
    const int64_t m = (int64_t)b * F * R + (n - 1);
    const int f = (n - 1) / R;
    float bx[BOX];
#pragma unroll
    for (int c = 0; c < BOX; ++c) bx[c] = box[m * BOX + c];
    for (int d = threadIdx.x; d < EMB; d += blockDim.x) {
        // the reference adds in this order: (W_o feat + b_o) + (W_p box + b_p), then + temporal  (:404-432)
        float pe = 0.f;
#pragma unroll
        for (int c = 0; c < BOX; ++c) pe += bx[c] * Wp[d * BOX + c];
        const float v = (to_f(tok[m * EMB + d]) + (pe + bp[d])) + temporal[f * EMB + d];
        x[row * EMB + d] = from_f<T>(v);
    }
    if (threadIdx.x == 0) addmask[row] = (mask01[m] - 1.f) * 100.f;
}

// dtok[b*FR + t, :] = dx[b, 1+t, :]   (16-byte pieces; a row is 768 elements)
template <typename T>
__global__ __launch_bounds__(256) void embed_unassemble_kernel(int B, int F, int R, const T* __restrict__ dx, T* __restrict__ dtok) {
    constexpr int PR = EMB * (int)sizeof(T) / 16;               // pieces per row
    const int FR = F * R;
    const int64_t total = (int64_t)B * FR * PR;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / PR; const int pc = (int)(i % PR);
        const int64_t src = (m / FR) * (FR + 1) + 1 + (m % FR);
        ((uint4*)dtok)[m * PR + pc] = ((const uint4*)dx)[src * PR + pc];
    }
}

// The heads' view of a tower output x [B][N][d]: row 0 of every sample (the CLS / global embedding) and rows 1.. (the local embeddings),
// each contiguous (model/model.py:70-96 slices and calls .contiguous()) -- one launch; and its backward, dx assembled from the two
// gradients in one launch (autograd's form of it is two zero fills, two strided copies and an add).  Raw 16-byte pieces: any dtype.
__global__ __launch_bounds__(256) void split_cls_kernel(int64_t B, int64_t N, int PR, const uint4* __restrict__ x, uint4* __restrict__ g, uint4* __restrict__ l) {
    const int64_t total = B * N * PR;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / PR; const int pc = (int)(i % PR);
        const int64_t b = row / N, n = row % N;
        const uint4 v = x[i];
        if (n == 0) g[b * PR + pc] = v; else l[(b * (N - 1) + n - 1) * PR + pc] = v;
    }
}
__global__ __launch_bounds__(256) void merge_cls_kernel(int64_t B, int64_t N, int PR, const uint4* __restrict__ dg, const uint4* __restrict__ dl, uint4* __restrict__ dx) {
    const int64_t total = B * N * PR;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / PR; const int pc = (int)(i % PR);
        const int64_t b = row / N, n = row % N;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (n == 0) { if (dg) v = dg[b * PR + pc]; } else if (dl) v = dl[(b * (N - 1) + n - 1) * PR + pc];
        dx[i] = v;
    }
}

__device__ __forceinline__ void ld4(const float* p, float (&o)[4]) { const float4 v = *(const float4*)p; o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
__device__ __forceinline__ void ld4(const bf16* p, float (&o)[4]) { const bf16x4 v = *(const bf16x4*)p; o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2]; o[3] = (float)v[3]; }
// stage 1 of dWp[d][c] = sum_m dtok[m][d] * box[m][c]: partial[p][c][d].  One workgroup per chunk of rows, 192 threads x 4 columns (8- /
// 16-byte row pieces), eight rows in flight per thread; the six box values of a row are workgroup-uniform (scalar loads).
template <typename T>
__global__ __launch_bounds__(192) void box_wgrad_kernel(int64_t M, const T* __restrict__ dtok, const float* __restrict__ box, int64_t rows_per, float* __restrict__ partial) {
    const int d = threadIdx.x * 4;
    const int64_t m0 = (int64_t)blockIdx.x * rows_per, m1 = m0 + rows_per < M ? m0 + rows_per : M;
    float acc[BOX][4];
#pragma unroll
    for (int c = 0; c < BOX; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[c][j] = 0.f;
    constexpr int UR = 8;
    int64_t m = m0;
    for (; m + UR <= m1; m += UR) {
        float g[UR][4];
#pragma unroll
        for (int u = 0; u < UR; ++u) ld4(dtok + (m + u) * EMB + d, g[u]);
#pragma unroll
        for (int u = 0; u < UR; ++u)
#pragma unroll
            for (int c = 0; c < BOX; ++c) {
                const float b = box[(m + u) * BOX + c];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[c][j] += g[u][j] * b;
            }
    }
    for (; m < m1; ++m) {
        float g[4];
        ld4(dtok + m * EMB + d, g);
#pragma unroll
        for (int c = 0; c < BOX; ++c) {
            const float b = box[m * BOX + c];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[c][j] += g[j] * b;
        }
    }
#pragma unroll
    for (int c = 0; c < BOX; ++c) *(float4*)(partial + ((int64_t)blockIdx.x * BOX + c) * EMB + d) = make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]);
}
// stage 2: 64 outputs x 16 slices of the P partial planes per workgroup, every load of a thread in flight at once (the one-thread-per-output
// loop over 256 planes was a chain of dependent loads: 95 us)
__global__ __launch_bounds__(1024) void box_wgrad_reduce_kernel(int64_t P, const float* __restrict__ partial, float* __restrict__ dWp, int accumulate) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;    // i = c*EMB + d  (BOX * EMB is a multiple of 64)
    float s16[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { const int64_t pp = q + 16 * u; s16[u] = pp < P ? partial[pp * BOX * EMB + i] : 0.f; }
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) s += s16[u];
    for (int64_t pp = q + 256; pp < P; pp += 16) s += partial[pp * BOX * EMB + i];
    red[q][lane] = s;
    __syncthreads();
    if (q == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][lane];
        const int c = i / EMB, d = i % EMB;
        dWp[d * BOX + c] = accumulate ? dWp[d * BOX + c] + t : t;
    }
}

// DistilBERT embeddings: e = word[id] + pos[l] (saved), y = LN(e)
template <typename T>
__global__ __launch_bounds__(256) void text_embed_kernel(int64_t M, int L, const int64_t* __restrict__ ids, const float* __restrict__ word,
                                                         const float* __restrict__ pos, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps, T* __restrict__ e_out, T* __restrict__ y,
                                                         float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int64_t id = ids[row];
    const int l = (int)(row % L);
    float v[12];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float4 w = *(const float4*)(word + id * EMB + c * 256 + lane * 4);
        const float4 p = *(const float4*)(pos + (int64_t)l * EMB + c * 256 + lane * 4);
        v[4 * c] = w.x + p.x; v[4 * c + 1] = w.y + p.y; v[4 * c + 2] = w.z + p.z; v[4 * c + 3] = w.w + p.w;
        // the saved pre-LN sum is what LayerNorm backward sees: normalise the value actually stored
#pragma unroll
        for (int j = 0; j < 4; ++j) { const T t = from_f<T>(v[4 * c + j]); e_out[row * EMB + c * 256 + lane * 4 + j] = t; v[4 * c + j] = to_f(t); s += v[4 * c + j]; }
    }
    const float mean = wave_sum(s) / (float)EMB;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 12; ++j) { const float d = v[j] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) / (float)EMB + eps);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = c * 256 + lane * 4 + j;
            y[row * EMB + col] = from_f<T>((v[4 * c + j] - mean) * rstd * gamma[col] + beta[col]);
        }
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// dword[id] += sum of de[row] over the rows holding token id (padding_idx 0 gets no gradient), WITHOUT float atomics: the first row that
// holds an id (its "leader") sums every row of that id in index order and is the only writer of dword[id] -- [CLS] / [SEP] sit in every
// caption, and an atomic sum of B addends is not reproducible from run to run (AdamW's 1e-6 epsilon turns that ulp-level noise into a few
// per cent of the learning rate for small gradients).  This was the only float atomic of the training step: steps are now bit-reproducible.
template <typename T>
__global__ __launch_bounds__(256) void text_embed_bwd_kernel(int64_t M, const int64_t* __restrict__ ids, const T* __restrict__ de, float* __restrict__ dword) {
    constexpr int LIST = 1024;                  // member rows gathered per round (a token in more than 1024 rows takes several rounds)
    constexpr int NV = EMB / 256;               // 4-column pieces per lane: a WAVE covers a whole row
    __shared__ int list[LIST];
    __shared__ int wsum[4];
    __shared__ float part[3][EMB];
    const int64_t row = blockIdx.x;
    const int64_t id = ids[row];
    if (id == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // pass 1, every candidate: is an earlier row holding this token (then its leader sums this row too), is a later one (coalesced, no order)
    int earlier = 0, later = 0;
    unsigned long long smask = 0ull;                 // bit k: position tid + 256 k holds this token (first 64 tiles; beyond: re-read in pass 2)
    for (int64_t k0 = 0; k0 * 256 < M; k0 += 8) {         // eight independent loads in flight (one by one each paid an L2 round trip)
        int64_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int64_t m = (k0 + u) * 256 + tid; v[u] = m < M ? ids[m] : -1; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t k = k0 + u, m = k * 256 + tid;
            const bool same = v[u] == id;
            earlier |= (same && m < row);
            later |= (same && m > row);
            if (same && k < 64) smask |= 1ull << k;
        }
    }
    if (__syncthreads_or(earlier)) return;
    const bool dup = __syncthreads_or(later);
    // wave w sums the members of rank = w (mod 4) in rank order (the leader row itself is wave 0's first addend); the four partial rows are
    // then added in wave order: one fixed association, whatever the launch looks like
    float acc[NV][4];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        if (wid == 0) ld4(de + row * EMB + q * 256 + lane * 4, acc[q]);
        else { acc[q][0] = 0.f; acc[q][1] = 0.f; acc[q][2] = 0.f; acc[q][3] = 0.f; }
    }
    // pass 2, the few leaders of repeated tokens: the later rows in index order, 256 candidate positions at a time (ballot per wave + wave
    // offsets), LIST of them per round
    for (int first = 0, more = dup ? 1 : 0; more; first += LIST) {
        int cnt = 0;                                 // rank of the next hit (uniform)
        // tiles aligned with pass 1 (position = tid + 256 k): the compare results are in `smask`, so a tile costs no memory round trip
        for (int64_t k = row >> 8; k * 256 < M; ++k) {
            const int64_t m = k * 256 + tid;
            const bool hit = m > row && m < M && (k < 64 ? ((smask >> k) & 1ull) != 0ull : ids[m] == id);
            const unsigned long long bal = __ballot(hit);
            if (lane == 0) wsum[wid] = __popcll(bal);
            __syncthreads();
            int off = cnt, tot = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { if (w < wid) off += wsum[w]; tot += wsum[w]; }
            const int r = off + __popcll(bal & ((1ull << lane) - 1ull));
            if (hit && r >= first && r < first + LIST) list[r - first] = (int)m;
            cnt += tot;
            __syncthreads();
        }
        const int n = cnt - first < LIST ? cnt - first : LIST;
        more = cnt > first + LIST;
#pragma unroll 4
        for (int k = wid; k < n; k += 4) {
            const int64_t m = list[k];
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                float v[4];
                ld4(de + m * EMB + q * 256 + lane * 4, v);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[q][j] += v[j];
            }
        }
        __syncthreads();
    }
    if (wid > 0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) *(float4*)&part[wid - 1][q * 256 + lane * 4] = make_float4(acc[q][0], acc[q][1], acc[q][2], acc[q][3]);
    }
    __syncthreads();
    if (wid == 0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            float4* dst = (float4*)(dword + id * EMB + q * 256 + lane * 4);
            float4 o = *dst;
            float s4[4] = {acc[q][0], acc[q][1], acc[q][2], acc[q][3]};
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                const float4 p4 = *(const float4*)&part[w][q * 256 + lane * 4];
                s4[0] += p4.x; s4[1] += p4.y; s4[2] += p4.z; s4[3] += p4.w;
            }
            o.x += s4[0]; o.y += s4[1]; o.z += s4[2]; o.w += s4[3];
            *dst = o;
        }
    }
}

template <typename S, typename D>
__global__ void cast_kernel(int64_t n, const S* __restrict__ src, D* __restrict__ dst) {
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
        if (i + 3 < n) {
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[i + j] = (D)(float)src[i + j];
        } else {
            for (int64_t j = i; j < n; ++j) dst[j] = (D)(float)src[j];
        }
    }
}

extern "C" int dvlp_obj_split(int dtype, int64_t M, const float* obj, void* feat, float* box, void* stream) {
    dvlp_clear_status();
    if (M <= 0) return DVLP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DVLP_F32) hipLaunchKernelGGL(obj_split_kernel<float>, dim3((unsigned)M), dim3(256), 0, st, M, obj, (float*)feat, box);
    else if (dtype == DVLP_BF16) hipLaunchKernelGGL(obj_split_kernel<bf16>, dim3((unsigned)M), dim3(256), 0, st, M, obj, (bf16*)feat, box);
    else return DVLP_ERR_DTYPE;
    return dvlp_launch_status();
}

extern "C" int dvlp_embed_assemble(int dtype, int64_t B, int64_t F, int64_t R, const void* tok, const float* box, const float* Wp,
                                   const float* bp, const float* temporal, const float* cls, const float* pos0, const float* mask01,
                                   void* x, float* addmask, void* stream) {
    dvlp_clear_status();
    if (B <= 0 || F <= 0 || R <= 0) return DVLP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)(B * (1 + F * R))), block(256);
    if (dtype == DVLP_F32) hipLaunchKernelGGL(embed_assemble_kernel<float>, grid, block, 0, st, (int)B, (int)F, (int)R, (const float*)tok, box, Wp, bp, temporal, cls, pos0, mask01, (float*)x, addmask);
    else if (dtype == DVLP_BF16) hipLaunchKernelGGL(embed_assemble_kernel<bf16>, grid, block, 0, st, (int)B, (int)F, (int)R, (const bf16*)tok, box, Wp, bp, temporal, cls, pos0, mask01, (bf16*)x, addmask);
    else return DVLP_ERR_DTYPE;
    return dvlp_launch_status();
}

extern "C" int dvlp_embed_unassemble(int dtype, int64_t B, int64_t F, int64_t R, const void* dx, void* dtok, void* stream) {
    dvlp_clear_status();
    if (B <= 0 || F <= 0 || R <= 0) return DVLP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t pieces = B * F * R * EMB / 4;                   // >= the 16-byte pieces of either dtype
    dim3 grid((unsigned)(pieces / 256 < 4096 ? cdiv(pieces, 256) : 4096)), block(256);
    if (dtype == DVLP_F32) hipLaunchKernelGGL(embed_unassemble_kernel<float>, grid, block, 0, st, (int)B, (int)F, (int)R, (const float*)dx, (float*)dtok);
    else if (dtype == DVLP_BF16) hipLaunchKernelGGL(embed_unassemble_kernel<bf16>, grid, block, 0, st, (int)B, (int)F, (int)R, (const bf16*)dx, (bf16*)dtok);
    else return DVLP_ERR_DTYPE;
    return dvlp_launch_status();
}

// x [B][N][row_bytes] -> g [B][row_bytes] (row 0), l [B][N-1][row_bytes] (rows 1..); row_bytes a multiple of 16, 16-byte aligned pointers
extern "C" int dvlp_split_cls(int64_t B, int64_t N, int64_t row_bytes, const void* x, void* g, void* l, void* stream) {
    dvlp_clear_status();
    if (B <= 0 || N < 2 || row_bytes <= 0 || row_bytes % 16 || ((uintptr_t)x | (uintptr_t)g | (uintptr_t)l) % 16) return DVLP_ERR_SHAPE;
    const int64_t pieces = B * N * (row_bytes / 16);
    hipLaunchKernelGGL(split_cls_kernel, dim3((unsigned)(pieces / 256 < 4096 ? cdiv(pieces, 256) : 4096)), dim3(256), 0, (hipStream_t)stream, B, N,
                       (int)(row_bytes / 16), (const uint4*)x, (uint4*)g, (uint4*)l);
    return dvlp_launch_status();
}
// backward of the above: dx [B][N][row_bytes] from dg / dl (either may be NULL: zeros)
extern "C" int dvlp_merge_cls(int64_t B, int64_t N, int64_t row_bytes, const void* dg, const void* dl, void* dx, void* stream) {
    dvlp_clear_status();
    if (B <= 0 || N < 2 || row_bytes <= 0 || row_bytes % 16 || ((uintptr_t)dx | (uintptr_t)dg | (uintptr_t)dl) % 16) return DVLP_ERR_SHAPE;
    const int64_t pieces = B * N * (row_bytes / 16);
    hipLaunchKernelGGL(merge_cls_kernel, dim3((unsigned)(pieces / 256 < 4096 ? cdiv(pieces, 256) : 4096)), dim3(256), 0, (hipStream_t)stream, B, N,
                       (int)(row_bytes / 16), (const uint4*)dg, (const uint4*)dl, (uint4*)dx);
    return dvlp_launch_status();
}

// workspace: fp32 [dvlp_box_wgrad_chunks(M) * 6 * 768]
extern "C" int64_t dvlp_box_wgrad_chunks(int64_t M) { const int64_t c = cdiv(M, 64); return c < 256 ? c : 256; }
extern "C" int dvlp_box_wgrad(int dtype, int64_t M, const void* dtok, const float* box, float* dWp, float* workspace, int accumulate, void* stream) {
    dvlp_clear_status();
    if (M <= 0) return DVLP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t P = dvlp_box_wgrad_chunks(M), rows_per = cdiv(M, P);
    dim3 grid((unsigned)P), block(192);
    if (dtype == DVLP_F32) hipLaunchKernelGGL(box_wgrad_kernel<float>, grid, block, 0, st, M, (const float*)dtok, box, rows_per, workspace);
    else if (dtype == DVLP_BF16) hipLaunchKernelGGL(box_wgrad_kernel<bf16>, grid, block, 0, st, M, (const bf16*)dtok, box, rows_per, workspace);
    else return DVLP_ERR_DTYPE;
    static_assert(BOX * EMB % 64 == 0 && EMB == 192 * 4, "box_wgrad kernels' thread maps");
    hipLaunchKernelGGL(box_wgrad_reduce_kernel, dim3(BOX * EMB / 64), dim3(1024), 0, st, P, workspace, dWp, accumulate);
    return dvlp_launch_status();
}

extern "C" int dvlp_text_embed_fwd(int dtype, int64_t B, int64_t L, const int64_t* ids, const float* word, const float* pos,
                                   const float* gamma, const float* beta, float eps, void* e_out, void* y, float* mean, float* rstd,
                                   void* stream) {
    dvlp_clear_status();
    if (B <= 0 || L <= 0 || L > 512) return DVLP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t M = B * L;
    dim3 grid((unsigned)cdiv(M, 4)), block(256);
    if (dtype == DVLP_F32) hipLaunchKernelGGL(text_embed_kernel<float>, grid, block, 0, st, M, (int)L, ids, word, pos, gamma, beta, eps, (float*)e_out, (float*)y, mean, rstd);
    else if (dtype == DVLP_BF16) hipLaunchKernelGGL(text_embed_kernel<bf16>, grid, block, 0, st, M, (int)L, ids, word, pos, gamma, beta, eps, (bf16*)e_out, (bf16*)y, mean, rstd);
    else return DVLP_ERR_DTYPE;
    return dvlp_launch_status();
}

// dword must be zeroed by the caller (dense [V,768] gradient, as nn.Embedding produces)
extern "C" int dvlp_text_embed_bwd(int dtype, int64_t M, const int64_t* ids, const void* de, float* dword, void* stream) {
    dvlp_clear_status();
    if (M <= 0) return DVLP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DVLP_F32) hipLaunchKernelGGL(text_embed_bwd_kernel<float>, dim3((unsigned)M), dim3(256), 0, st, M, ids, (const float*)de, dword);
    else if (dtype == DVLP_BF16) hipLaunchKernelGGL(text_embed_bwd_kernel<bf16>, dim3((unsigned)M), dim3(256), 0, st, M, ids, (const bf16*)de, dword);
    else return DVLP_ERR_DTYPE;
    return dvlp_launch_status();
}

// The caption-side inputs of GlobalLocalLoss from the attention mask, one launch (trainer/trainer_dist.py:152-159): text_length[b] = sum_w att[b][w]
// (int64, as torch.sum gives) and text_mask[b][w - 1] = (att[b][w] - 1) * 100 for w >= 1 (fp32) -- the stock form is a reduce, a slice copy, a
// subtraction and a multiplication: four launches inside every captured step.  One wave per caption.
__global__ __launch_bounds__(256) void text_mask_len_kernel(int64_t B, int L, const int64_t* __restrict__ att, int64_t* __restrict__ len, float* __restrict__ mask) {
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= B) return;
    float cnt = 0.f;
    for (int w = lane; w < L; w += 64) {
        const float a = (float)att[b * L + w];
        cnt += a;
        if (w >= 1) mask[b * (L - 1) + w - 1] = (a - 1.0f) * 100.0f;
    }
    cnt = wave_sum(cnt);                                         // small integers: exact in fp32
    if (lane == 0) len[b] = (int64_t)cnt;
}
extern "C" int dvlp_text_mask_len(int64_t B, int64_t L, const int64_t* att, int64_t* text_length, float* text_mask, void* stream) {
    dvlp_clear_status();
    if (B <= 0 || L < 2 || L > (1 << 20)) return DVLP_ERR_SHAPE;
    hipLaunchKernelGGL(text_mask_len_kernel, dim3((unsigned)cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, B, (int)L, att, text_length, text_mask);
    return dvlp_launch_status();
}

// DistilBERT's additive key mask from the attention mask, one launch (HF DistilBERT: scores.masked_fill(mask == 0, -inf)): key_mask[b][w] = 0 where
// att[b][w] != 0, -inf elsewhere (fp32) -- the stock form is a zero fill, a comparison and a masked fill: three launches in every captured step.
__global__ __launch_bounds__(256) void text_key_mask_kernel(int64_t n, const int64_t* __restrict__ att, float* __restrict__ mask) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) mask[i] = att[i] != 0 ? 0.f : -INFINITY;
}
extern "C" int dvlp_text_key_mask(int64_t B, int64_t L, const int64_t* att, float* key_mask, void* stream) {
    dvlp_clear_status();
    if (B <= 0 || L <= 0) return DVLP_ERR_SHAPE;
    hipLaunchKernelGGL(text_key_mask_kernel, dim3((unsigned)cdiv(B * L, 256)), dim3(256), 0, (hipStream_t)stream, B * L, att, key_mask);
    return dvlp_launch_status();
}

extern "C" int dvlp_cast(int src_dtype, int dst_dtype, int64_t n, const void* src, void* dst, void* stream) {
    dvlp_clear_status();
    if (n <= 0) return DVLP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    int64_t blocks = cdiv(n, 1024); if (blocks > 4096) blocks = 4096;
    dim3 grid((unsigned)blocks), block(256);
    if (src_dtype == DVLP_F32 && dst_dtype == DVLP_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16>), grid, block, 0, st, n, (const float*)src, (bf16*)dst);
    else if (src_dtype == DVLP_BF16 && dst_dtype == DVLP_F32) hipLaunchKernelGGL((cast_kernel<bf16, float>), grid, block, 0, st, n, (const bf16*)src, (float*)dst);
    else if (src_dtype == DVLP_F32 && dst_dtype == DVLP_F32) hipLaunchKernelGGL((cast_kernel<float, float>), grid, block, 0, st, n, (const float*)src, (float*)dst);
    else return DVLP_ERR_DTYPE;
    return dvlp_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// Token-grid transpose for the time attention of SpaceTimeBlock (model/object_transformer.py:252-258: einops
// 'b (f n) d -> (b n) f d'): dst[b][0] = src[b][0], dst[b][1 + n F + f] = src[b][1 + f R + n]  (+ res at the destination
// index when given).  Time attention over (F frames x R regions) is then exactly the space-attention kernel run on the
// transposed token order with the roles (frames, regions) = (R, F); calling this again with F and R swapped goes back.
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void token_transpose_kernel(int N, int F, int R, int D, const T* __restrict__ src, const T* __restrict__ res,
                                                              T* __restrict__ dst) {
    const int64_t row = blockIdx.x;                 // destination row b * N + t
    const int t = (int)(row % N);
    const int64_t b = row / N;
    int s = 0;
    if (t > 0) { const int n = (t - 1) / F, f = (t - 1) % F; s = 1 + f * R + n; }
    const T* sp = src + (b * N + s) * D;
    const T* rp = res ? res + row * D : nullptr;
    T* dp = dst + row * D;
    for (int d = threadIdx.x; d < D; d += blockDim.x) dp[d] = from_f<T>(to_f(sp[d]) + (rp ? to_f(rp[d]) : 0.f));
}

extern "C" int dvlp_token_transpose(int dtype, int64_t B, int64_t F, int64_t R, int64_t D, const void* src, const void* res, void* dst, void* stream) {
    dvlp_clear_status();
    if (B <= 0 || F <= 0 || R <= 0 || D <= 0) return DVLP_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int N = 1 + (int)(F * R);
    dim3 grid((unsigned)(B * N)), block(D >= 256 ? 256 : 64);
    if (dtype == DVLP_F32) hipLaunchKernelGGL(token_transpose_kernel<float>, grid, block, 0, st, N, (int)F, (int)R, (int)D, (const float*)src, (const float*)res, (float*)dst);
    else if (dtype == DVLP_BF16) hipLaunchKernelGGL(token_transpose_kernel<bf16>, grid, block, 0, st, N, (int)F, (int)R, (int)D, (const bf16*)src, (const bf16*)res, (bf16*)dst);
    else return DVLP_ERR_DTYPE;
    return dvlp_launch_status();
}
