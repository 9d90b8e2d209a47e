/* demovlp_hip.h -- C ABI of libdemovlp_hip.so: the MI355X (gfx950) kernels behind DemoVLP's cross-modal
 * forward/backward hot path.
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer borrowed from the caller (PyTorch-ROCm allocates; nothing here allocates,
 *     frees or synchronises except dvlp_prof_collect).  Workspaces are passed in; their sizes come from the
 *     *_bytes / *_blocks / *_chunks helpers below.
 *   - `stream` is a hipStream_t; every call is asynchronous on it and re-entrant per stream.
 *   - `dtype` selects the COMPUTE/STORAGE dtype of activations: DVLP_F32 (exact-fp32 MFMA, the 1e-4 parity path)
 *     or DVLP_BF16 (bf16 MFMA, fp32 accumulate, the throughput path).  Biases, LayerNorm parameters, statistics,
 *     masks, losses and optimizer state are always fp32.
 *   - Return value: 0 on success, negative DVLP_ERR_* otherwise (the Python binding raises).
 *
 * The reference (showlab/DemoVLP) has no native code: each entry point below replaces the PyTorch op sequence
 * cited next to it (paths relative to the reference repo root).
 */
#ifndef DEMOVLP_HIP_H
#define DEMOVLP_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { DVLP_F32 = 0, DVLP_BF16 = 1 };
enum { DVLP_OK = 0, DVLP_ERR_DTYPE = -1, DVLP_ERR_SHAPE = -2, DVLP_ERR_LAUNCH = -3, DVLP_ERR_UNSUPPORTED = -4 };
/* ---- Every entry point here is stateless apart from caller-registered scratch (dvlp_set_workspace*, dvlp_reduce_*); optional per-call
 *      extras travel in the `dvlp_*_ext` structs of the `*_ex` calls, never through "next call" setters.  The developer switches
 *      (`dvlp_dev_*`: A/B measurements, timing ablations, forced code paths) are NOT in this library: they exist only in the -DDVLP_DEV
 *      build (libdemovlp_hip_dev.so, declared in demovlp_hip_dev.h); here each of them is a compile-time constant at its default. ---- */
/* dvlp_gemm epilogue flags */
enum { DVLP_EPI_GELU = 1, DVLP_EPI_GELU_BWD = 2, DVLP_EPI_RELU_BWD = 4, DVLP_EPI_ACCUM = 8, DVLP_EPI_LEAKY = 16,
       DVLP_EPI_OUT_F32 = 32 /* C is fp32 whatever the compute dtype (weight gradients) */ };

/* ---- K1 region select: data_loader/WebVid_dataset.py:134-283 (read_all_object_from_disk + object_select_random) ---- */
int dvlp_region_select(int64_t BF, int64_t F, int64_t Nraw, int64_t R, const float* feats, const float* bbox, const float* conf,
                       const float* wh, const int* nvalid, float* obj, float* mask, int* order, int* lens, void* stream);

/* ---- GEMM: nn.Linear forward/backward everywhere on the path (model/object_transformer.py:116-122, 155, 194,
 *      404-408, 451; DistilBERT q/k/v/out/ffn; model/model.py:40-43) and the contractions of model/loss.py:235,267.
 *      C[M,N] = epi(alpha * op(A) op(B)^T); transX=0: X[r*ld+k], transX=1: X[k*ld+r].                              ---- */
int dvlp_gemm(int dtype, int transA, int transB, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
              int64_t ldb, void* C, int64_t ldc, const float* bias, const void* res, int64_t ldres, void* aux, int64_t ldaux,
              int flags, float alpha, void* stream);
int dvlp_gemm_batched(int dtype, int transA, int transB, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda,
                      const void* B, int64_t ldb, void* C, int64_t ldc, const float* bias, const void* res, int64_t ldres,
                      void* aux, int64_t ldaux, int flags, float alpha, int64_t batch, int64_t strideA, int64_t strideB,
                      int64_t strideC, int64_t strideRes, int64_t strideAux, void* stream);
/* The weight gradients of `count` linears at once: dW_p[M_p, N_p] (fp32, contiguous) (+)= dY_p[K_p, M_p]^T X_p[K_p, N_p], i.e.
   count calls of dvlp_gemm(transA=1, transB=1, fp32 out) -- the K = batch*tokens reductions of one transformer layer
   (reference: autograd of the nn.Linear modules in object_transformer.py:100-196 and DistilBERT's TransformerBlock).  Arguments are
   host arrays of length `count`.  bf16 problems that suit the 256 x 256 kernel run as one grouped launch. */
int dvlp_wgrad_grouped(int dtype, int count, const int64_t* M, const int64_t* N, const int64_t* K, const void* const* dY,
                       const int64_t* ld_dy, const void* const* X, const int64_t* ld_x, void* const* dW, int accumulate, void* stream);
/* The same with an optimizer update riding along (round 6): `ext` describes one dvlp_adamw_range_dev call -- the fused HF-AdamW pass over a
   contiguous range of the flat parameter / gradient / moment buffers, typically the PREVIOUS layer's weights, whose gradients are final -- that
   is executed by the workgroups this launch would leave idle (a ViT layer's group is 216 workgroups on 256 CUs) instead of as a launch of its
   own between two GEMM launches; when the group has fewer than 16 spare workgroups (or does not take the grouped path) the update is launched
   right behind it.  ext->fused (out) says which.  NULL ext = dvlp_wgrad_grouped. */
typedef struct dvlp_wgrad_ext {
    int64_t n;
    float* p; const float* g; float* m; float* v; const float* hyper; void* bf16_shadow;
    int fused;
} dvlp_wgrad_ext;
int dvlp_wgrad_grouped_ex(int dtype, int count, const int64_t* M, const int64_t* N, const int64_t* K, const void* const* dY,
                          const int64_t* ld_dy, const void* const* X, const int64_t* ld_x, void* const* dW, int accumulate,
                          dvlp_wgrad_ext* ext, void* stream);
/* Per-stream hint, caller-registered like the workspaces: co_running != 0 says that launches on `stream` share the chip with kernels of another
   stream (the text tower beside the object tower, model.ObjectRelation.parallel_towers); the bf16 GEMM dispatch then favours CU-time per FLOP
   (tall tiles) over the latency of its own grid.  0 removes the hint. */
int dvlp_stream_hint(void* stream, int co_running);
/* dvlp_gemm with optional extras, handed to the call that consumes them (NULL = none).  `colsum`: fp32 dst[N] = column sums of the stored
   output C -- e.g. the bias gradient of the Linear whose output gradient this product is (batch 1, not for fp32 outputs).  Fused into the
   256-row kernel's epilogue through the deferred-reduction queue where possible (final after dvlp_reduce_flush; colsum_fused = 1),
   otherwise a plain column-sum pass runs behind the product (colsum_fused = 0; final when the call's work is). */
typedef struct dvlp_gemm_ext {
    float* colsum;
    int colsum_fused;      /* out */
} dvlp_gemm_ext;
int dvlp_gemm_ex(int dtype, int transA, int transB, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                 int64_t ldb, void* C, int64_t ldc, const float* bias, const void* res, int64_t ldres, void* aux, int64_t ldaux,
                 int flags, float alpha, dvlp_gemm_ext* ext, void* stream);
/* scratch for split-K partial sums (weight-gradient GEMMs); caller-owned device memory, NULL disables splitting */
int dvlp_set_workspace(void* ptr, int64_t bytes);
/* same, for one stream only (GEMMs running concurrently on two streams need separate slabs) */
int dvlp_set_workspace_stream(void* stream, void* ptr, int64_t bytes);
/* per-launch HIP-event timing of the GEMM kernels (bench.py roofline figure) */
int dvlp_prof_enable(int on);
int dvlp_prof_collect(double* total_ms, double* total_flops, int64_t* count);

/* ---- LayerNorm: norm1/norm2 (object_transformer.py:261,271, eps 1e-6) and DistilBERT's post-LNs (eps 1e-12) ---- */
int dvlp_layernorm_fwd(int dtype, int64_t M, int64_t D, const void* x, const float* gamma, const float* beta, float eps, void* y,
                       void* y_relu, float* mean, float* rstd, void* stream);
int64_t dvlp_layernorm_bwd_blocks(int64_t M);
/* dx_colsum (optional): fp32 [D] <- column sums of dx, i.e. the bias gradient of the Linear that dx feeds (nn.Linear backward
   after the LayerNorm's); produced by the same kernel on the deferred path, by a dvlp_colsum pass otherwise */
int dvlp_layernorm_bwd(int dtype, int64_t M, int64_t D, const void* dy, const void* x, const float* gamma, const float* mean,
                       const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta, float* workspace, int accumulate,
                       float* dx_colsum, void* stream);
/* Deferred second stages.  dvlp_layernorm_bwd / dvlp_colsum called with (accumulate | 2) -- "nobody reads the result before
   the flush" -- park their partial sums in `workspace` and queue the final column reduction; dvlp_reduce_flush runs every
   queued reduction as ONE launch (a training step otherwise pays ~125 ten-microsecond launches for them).  The table is
   device memory for the queue (48 bytes per queued reduction).  NULL workspace disables deferral (the default): such calls
   then reduce immediately.  The trainer flushes before the optimizer step / before reducing the tail gradient bucket. */
int dvlp_reduce_defer(void* workspace, int64_t workspace_bytes, void* table, int64_t table_bytes);
int dvlp_reduce_flush(void* stream);
/* bias / table gradients: out[g][n] (+)= sum_m x_g[m][n] */
int64_t dvlp_colsum_chunks(int64_t M);
/* optional: `count` zeroed uint32 counters enabling the single-launch (last-workgroup-reduces) form of dvlp_colsum */
int dvlp_colsum_counters(void* ptr, int64_t count);
int dvlp_colsum(int dtype, int64_t M, int64_t N, const void* x, int64_t ld, int64_t inner, int64_t ostride, int64_t groups,
                int64_t gstride, float* out, float* workspace, int accumulate, void* stream);

/* ---- attention: VarAttention.forward + attn_mask (object_transformer.py:152-196, 91-97) [mode 0] and DistilBERT
 *      multi-head self-attention [mode 1]                                                                          ---- */
/* `workspace` (B*H*F*66 floats) and `cls_stats` (B*H*4 floats), both optional (mode 0, bf16): the CLS query -- the one query that
   attends to every key of the clip (object_transformer.py:162-167) -- is folded into the per-frame waves, merged flash-style by a
   64-thread launch, and its softmax statistics stay in `cls_stats` for dvlp_attention_bwd (`fwd_out` = this call's `out`), which
   then needs no statistics pass.  Backward workspace (mode 0): B*H*(F*192 + 4) floats.
   The two PLAIN calls below never fold: dvlp_attention_fwd leaves `cls_stats` untouched and dvlp_attention_bwd ignores `fwd_out` /
   `cls_stats` (it recomputes the statistics) -- pairing them is always safe; the fold is a contract of the `_ex` pair (ext->folded). */
int dvlp_attention_fwd(int dtype, int mode, int64_t B, int64_t N, int64_t H, int64_t F, int64_t R, const void* q, const void* k,
                       const void* v, int64_t ld, const float* addmask, void* out, int64_t ldo, float scale, float* workspace,
                       float* cls_stats, void* stream);
int dvlp_attention_bwd(int dtype, int mode, int64_t B, int64_t N, int64_t H, int64_t F, int64_t R, const void* q, const void* k,
                       const void* v, int64_t ld, const float* addmask, const void* dout, int64_t ldo, void* dq, void* dk, void* dv,
                       int64_t ldd, float* workspace, float scale, const void* fwd_out, int64_t ld_fwd_out, const float* cls_stats,
                       void* stream);
/* The same two calls with optional extras, handed to the call that consumes them (ext may be NULL):
     keep / keepT / keep_scale  (mode 1) dropout of the attention probabilities: keep bytes in both orientations from dvlp_dropout_attn_mask and
                                1 / (1 - p) -- HF MultiHeadSelfAttention's `weights = dropout(softmax(scores))`
     colsum                     (backward, mode 0) fp32 [3*H*64]: column sums of dq | dk | dv = the gradient of the packed qkv bias
                                (object_transformer.py:310 qkv_bias=True), queued through the deferred-reduction queue when the call can
                                (bf16 one-pass form with the forward's statistics): colsum_fused = 1; else the caller sums the columns itself
     folded                     (forward, out) 1 if the CLS query was folded and `cls_stats` filled -- only then hand them to the backward.
   The CLS fold needs `ext` (the caller must be able to learn whether `cls_stats` were written): the plain dvlp_attention_fwd never folds. */
typedef struct dvlp_attn_ext {
    const void* keep; const void* keepT; float keep_scale;
    float* colsum;
    int colsum_fused;      /* out */
    int folded;            /* out */
} dvlp_attn_ext;
int dvlp_attention_fwd_ex(int dtype, int mode, int64_t B, int64_t N, int64_t H, int64_t F, int64_t R, const void* q, const void* k,
                          const void* v, int64_t ld, const float* addmask, void* out, int64_t ldo, float scale, float* workspace,
                          float* cls_stats, dvlp_attn_ext* ext, void* stream);
int dvlp_attention_bwd_ex(int dtype, int mode, int64_t B, int64_t N, int64_t H, int64_t F, int64_t R, const void* q, const void* k,
                          const void* v, int64_t ld, const float* addmask, const void* dout, int64_t ldo, void* dq, void* dk, void* dv,
                          int64_t ldd, float* workspace, float scale, const void* fwd_out, int64_t ld_fwd_out, const float* cls_stats,
                          dvlp_attn_ext* ext, void* stream);

/* ---- tower prologues: ObjectTransformer.forward_features (object_transformer.py:400-433); DistilBERT embeddings ---- */
int dvlp_obj_split(int dtype, int64_t M, const float* obj, void* feat, float* box, void* stream);
int dvlp_embed_assemble(int dtype, int64_t B, int64_t F, int64_t R, const void* tok, const float* box, const float* Wp,
                        const float* bp, const float* temporal, const float* cls, const float* pos0, const float* mask01, void* x,
                        float* addmask, void* stream);
/* time attention of SpaceTimeBlock (object_transformer.py:252-258, 'b (f n) d -> (b n) f d'): dst[b][1 + n F + f] = src[b][1 + f R + n]
   (+ res at the destination index when non-NULL), CLS row in place; dvlp_attention_* mode 0 on the transposed order with (F, R)
   swapped IS the time attention; the same call with F and R swapped transposes back.  F = 1 is the identity (a plain add). */
int dvlp_token_transpose(int dtype, int64_t B, int64_t F, int64_t R, int64_t D, const void* src, const void* res, void* dst, void* stream);
int dvlp_embed_unassemble(int dtype, int64_t B, int64_t F, int64_t R, const void* dx, void* dtok, void* stream);

/* ---- the heads' view of a tower output (model/model.py:70-96: `[:, 0]` / `[:, 1:]` + .contiguous()): x [B][N][row_bytes] -> g [B][row_bytes]
 *      (row 0 of each sample) and l [B][N-1][row_bytes] in one launch, and its backward (dx from dg / dl, NULL = zeros) in one launch.
 *      Raw bytes, any dtype; row_bytes a multiple of 16, pointers 16-byte aligned.                                                    ---- */
int dvlp_split_cls(int64_t B, int64_t N, int64_t row_bytes, const void* x, void* g, void* l, void* stream);
int dvlp_merge_cls(int64_t B, int64_t N, int64_t row_bytes, const void* dg, const void* dl, void* dx, void* stream);
int64_t dvlp_box_wgrad_chunks(int64_t M);
int dvlp_box_wgrad(int dtype, int64_t M, const void* dtok, const float* box, float* dWp, float* workspace, int accumulate,
                   void* stream);
int dvlp_text_embed_fwd(int dtype, int64_t B, int64_t L, const int64_t* ids, const float* word, const float* pos,
                        const float* gamma, const float* beta, float eps, void* e_out, void* y, float* mean, float* rstd,
                        void* stream);
int dvlp_text_embed_bwd(int dtype, int64_t M, const int64_t* ids, const void* de, float* dword, void* stream);
int dvlp_cast(int src_dtype, int dst_dtype, int64_t n, const void* src, void* dst, void* stream);
/* the loss' caption-side inputs from the attention mask in one launch (trainer/trainer_dist.py:152-159): text_length[b] = sum_w att[b][w] (int64)
   and text_mask[b][w - 1] = (att[b][w] - 1) * 100, w = 1 .. L - 1 (fp32 [B, L - 1]); att int64 [B, L] */
int dvlp_text_mask_len(int64_t B, int64_t L, const int64_t* att, int64_t* text_length, float* text_mask, void* stream);
/* DistilBERT's additive key mask (HF: scores.masked_fill(attention_mask == 0, -inf); call site model/model.py:87): key_mask[B][L] fp32 = 0 where
   att != 0, -inf elsewhere -- one launch instead of a fill, a comparison and a masked fill */
int dvlp_text_key_mask(int64_t B, int64_t L, const int64_t* att, float* key_mask, void* stream);

/* ---- dropout of the text tower: the reference keeps DistilBERT in train mode (model/model.py:29-30), so HuggingFace's three
 *      dropouts (embeddings, attention probabilities, feed-forward output; p = 0.1) are part of every training step.  Masks are
 *      Philox4x32-10 streams keyed by (seed, step offset, site, element); `state` is DEVICE memory uint32[4] =
 *      {seed_lo, seed_hi, offset, -}; dvlp_dropout_advance bumps the offset (once per forward; replayable inside a hipGraph). ---- */
int dvlp_dropout_advance(void* state, void* stream);
/* y = x * keep / (1 - p) (+ res); keep (uint8 [n], 0/1) is stored for the backward.  n % 4 == 0. */
int dvlp_dropout_fwd(int dtype, int64_t n, const void* x, const void* res, void* y, void* keep, float p, const void* state, int site, void* stream);
int dvlp_dropout_bwd(int dtype, int64_t n, const void* dy, const void* keep, float p, void* dx, void* stream);
/* keep bytes of the attention probabilities of BH = batch * heads [N x N] maps in both orientations: keep[bh][q][key] and
   keepT[bh][key][q], row stride N rounded up to 16 (pads 0) */
int dvlp_dropout_attn_mask(int64_t BH, int64_t N, float p, const void* state, int site, void* keep, void* keepT, void* stream);
/* (the masks reach the attention kernels through dvlp_attn_ext of dvlp_attention_fwd_ex / dvlp_attention_bwd_ex) */
/* out[4] = Philox4x32-10(ctr_key[0..3], key = ctr_key[4..5]) computed on the device (known-answer test) */
int dvlp_philox_kat(const void* ctr_key, void* out, void* stream);

/* ---- local loss: xattn_score_fast / func_attention_fast / focal_equal / cosine_similarity (model/loss.py:209-330) ---- */
/* `gate`: bit 0 = focal_equal gate on (model/loss.py:274-283); bit 1 (DVLP_XATTN_ONE_STREAM) = issue everything on the caller's stream --
   by default the text->image half of the multi-kernel path (its contractions and cosine passes) runs on an internal side stream beside
   the image->text half between the softmax stages (fork / join by events, capturable into a hipGraph; its split-K slabs are a region of
   `workspace` of their own).  A per-call option, not process state: bench.py's per-launch timing pass sets it so that an event pair
   brackets one launch. */
enum { DVLP_XATTN_GATE = 1, DVLP_XATTN_ONE_STREAM = 2 };
int64_t dvlp_xattn_workspace_bytes(int dtype, int64_t Bi, int64_t Bj, int64_t G, int64_t W, int bwd);
int dvlp_xattn_fwd(int dtype, int64_t Bi, int64_t Bj, int64_t G, int64_t W, int64_t d, const void* Craw, const void* Qraw,
                   const float* mimg, const float* mcap, float lam, int gate, float* scores, void* workspace, int bwd, void* stream);
int dvlp_xattn_bwd(int dtype, int64_t Bi, int64_t Bj, int64_t G, int64_t W, int64_t d, const void* Craw, const void* Qraw,
                   const float* mimg, const float* mcap, float lam, int gate, const float* dscores, void* workspace, void* dC,
                   void* dQ, void* stream);

/* ---- loss heads: sim_matrix (model/model.py:582-590) + NormSoftmaxLoss (model/loss.py:126-138) + RWALoss tail
 *      (model/loss.py:105-116) + GlobalLocalLoss sum (:29-45); forward and analytic gradients in one launch        ---- */
int dvlp_global_local_loss(int dtype, int64_t B, int64_t d, const void* gt, const void* go, const float* xs, float temperature,
                           float lam, int use_global, int use_local, int stages, float* sim, float* dsim, void* dgt, void* dgo,
                           float* dxs, float* losses, void* stream);

/* rectangular sim_matrix (model/model.py:582-590 on [N,256] x [M,256], e.g. the whole eval set at trainer/trainer_dist.py:369):
   xn (fp32) = x / max(|x|, 1e-8) row-wise and norm = |x|; the [N,M] product and its two gradient products are dvlp_gemm calls in
   DVLP_F32; dvlp_rownorm_bwd takes d loss / d xn back through the normalisation */
int dvlp_rownorm_fwd(int dtype, int64_t M, int64_t d, const void* x, float* xn, float* norm, void* stream);
int dvlp_rownorm_bwd(int dtype, int64_t M, int64_t d, const void* x, const float* norm, const float* dxn, void* dx, void* stream);

/* ---- optimizer: transformers.AdamW as constructed at train_dist_multi.py:64 ---- */
int dvlp_adamw_step(int64_t n, float* p, const float* g, float* m, float* v, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int64_t step, float grad_scale, void* bf16_shadow, void* stream);
/* same update with the hyper-parameters in DEVICE memory: hyper[8] fp32 = {lr, beta1, beta2, eps, weight_decay, grad_scale, step,
   step_size}; each call first advances hyper[6] (the step counter) and recomputes hyper[7] on the device, so one captured hipGraph of
   a training step replays with the right bias correction, and lr / grad_scale change by writing the buffer */
int dvlp_adamw_step_dev(int64_t n, float* p, const float* g, float* m, float* v, float* hyper, void* bf16_shadow, void* stream);
/* the two halves of dvlp_adamw_step_dev, for a step whose update is spread over the backward pass: _prep advances the device step
   counter / step size once per step, _range updates n elements at the given (16-byte aligned) pointers with the hyper-parameters as
   they stand */
int dvlp_adamw_prep_dev(float* hyper, void* stream);
int dvlp_adamw_range_dev(int64_t n, float* p, const float* g, float* m, float* v, const float* hyper, void* bf16_shadow, void* stream);

#ifdef __cplusplus
}
#endif
#endif
