/* Developer switches of libdemovlp_hip_dev.so -- NOT part of the drop-in surface.
 *
 * The product library (demovlp_amd/lib/libdemovlp_hip.so, built without -DDVLP_DEV) exports NONE of these: every switch below is a
 * compile-time constant there, at the default named in its comment (`nm -D libdemovlp_hip.so | grep dvlp_dev_` is empty).  The same
 * sources built with -DDVLP_DEV give demovlp_amd/lib/libdemovlp_hip_dev.so, which exports them on top of everything in
 * demovlp_hip.h; they set process-global state and exist for A/B measurements (tools/), timing ablations and the tests that force a
 * code path (`tests/conftest.py: devlib`).  No module of the package calls one.
 */
#ifndef DEMOVLP_HIP_DEV_H
#define DEMOVLP_HIP_DEV_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* K split of dvlp_gemm's bf16 kernels: 0 (default) = automatic, > 0 = forced -- for A/B measurements (tools/gemm_sweep.py) */
int dvlp_dev_gemm_force_split(int s);
/* K split of the grouped weight-gradient launch: 0 (default) = automatic (the same split for every problem), > 0 = forced -- for
   A/B measurements (tools/wgrad_bench.py) */
int dvlp_dev_wgrad_group_split(int s);
/* 1 (default): LDS-DMA (global_load_lds) bf16 kernel; 0: register-staged bf16 kernel -- for A/B measurements */
int dvlp_dev_gemm_variant(int use_lds_dma);
/* TIMING-ONLY ablation of the LDS-DMA kernel's K loop (1 no DMA, 2 no fragment reads, 4 no MFMA); 0 in production */
int dvlp_dev_gemm_ablate(int bits);
/* 256 x 128 tile of the LDS-DMA kernel: 0 never, 1 heuristic (default), 2 always -- for A/B measurements */
int dvlp_dev_gemm_wide_mode(int mode);
/* 256 x 256 ping-pong kernel (8 waves, counted-vmcnt LDS-DMA prefetch): 0 never, 1 where the grid suits it, 2 whenever the
   operands allow -- for A/B measurements and tests */
int dvlp_dev_gemm_p8_mode(int mode);
/* tile height / K split of that kernel: 1 (default) height 160 / 192 / 224 / 256 and split chosen together by the fitted cost model
   (p8_plan); 0 always 256 rows; 2 224 rows whenever the operands allow; 3 round 5's rule (224 where rounds x rows is smaller, the old split);
   10 + MIH a forced height (128 + 32 MIH rows) -- for A/B measurements (tools/tile_sweep.py) and tests */
int dvlp_dev_gemm_p8_short_tiles(int mode);
/* persistent form of that kernel on outputs of more than one round of tiles (one workgroup per CU walks its tiles; the next tile's first
   units are staged by the previous tile's last phases, the epilogue's stores are not waited for): 0 (default) off, 1 on -- for A/B
   measurements (tools/p8p_bench.py) and tests */
int dvlp_dev_gemm_p8_persistent(int mode);
/* grouped weight gradients: 1 (default) blocks are dealt to the XCDs as 3 x 3 tile patches of one K slice, so a patch's operand panels are
   fetched into that XCD's L2 once; 0: per-problem tile order -- for A/B measurements */
int dvlp_dev_wgrad_group_patches(int on);
/* batched skinny products (N, K <= 288, M in the thousands: the local loss' per-video / per-caption contractions) on the resident-B streaming
   kernel (csrc/gemm_rb.hip): 1 (default) where its shapes fit, 0 never -- for A/B measurements and tests */
int dvlp_dev_gemm_resident_b(int on);
/* dvlp_wgrad_grouped_ex: 1 (default) the optimizer update it is handed rides on the launch's spare workgroups, 0 it is launched on its own --
   for A/B measurements */
int dvlp_dev_wgrad_group_ride(int on);
/* number of workgroups a split-K launch aims for (default 768 = 3 per CU) */
int dvlp_dev_gemm_splitk_target(int64_t n);
/* space-mode bf16 backward: 1 (default) one pass over q/k/v/dO with the CLS query folded into the frame tiles, 0 the
   three-launch form -- for A/B measurements and tests. */
int dvlp_dev_attention_bwd_variant(int merged);
/* space-mode bf16 with the CLS query folded: 1 (default) the round-5 kernels, 0 the round 3-4 ones -- for A/B measurements and tests */
int dvlp_dev_attention_lean(int on);
/* TIMING-ONLY ablation of the MFMA attention kernels: backward 1 no stores, 2 no exp, 4 stop after the softmax -- these exist in the three-launch
   backward only, which a call with any of them set is routed to (the one-pass forms are then not taken); forward, round-5 form: 8 loads and stores
   only -- what the access shape alone costs; 0 in production */
int dvlp_dev_attention_ablate(int bits);
/* bf16, D = 768: 1 (default) half a wave per row with 16-byte accesses, 0 the generic row-per-wave kernel -- for A/B measurements */
int dvlp_dev_layernorm_wide(int on);
/* 1 (default): fold the CLS query where workspaces are given; 0: separate CLS launches -- for A/B measurements and tests */
int dvlp_dev_attention_cls_fold(int on);
/* testing knob: 1 = always take the general-G (long-video) softmax path, even when the fused per-pair kernels fit LDS */
int dvlp_dev_xattn_force_general(int on);
/* 1 (default): bf16 pairs with F*R <= 288, W <= 112 run the fused per-pair kernels (everything between the embeddings and the
   score on chip); 0: always the multi-kernel path -- for A/B measurements and tests */
int dvlp_dev_xattn_fused_mode(int mode);
/* bf16 backward of the per-pair softmax stage: 1 (default) keeps both intermediate tiles on chip (LDS low halves / registers),
   0 runs the generic kernel that round-trips them through the workspace -- for A/B measurements and tests */
int dvlp_dev_xattn_bwd_variant(int packed);
/* 1 (default): bf16 pairs that fit the per-pair LDS tile use the Gram form of the text->image direction -- cos(wc2_g, C_g) from
   u_g = sum_w P2 S_raw and v_g = P2_g (Q^ Q^^T) P2_g^T, so the [Bj][Bi][G][d] weighted contexts are never formed (forward or backward);
   0: the weighted contexts are materialised as in the reference -- for A/B measurements and tests */
int dvlp_dev_xattn_gram(int on);
/* 1 (default): bf16 backward with the per-pair LDS tile: the dP1 rows are produced with regions g and g + 64 of every full block of 128
   adjacent (the product is handed a row-permuted copy of the unit regions), so the backward fetches them as 4-byte pieces; 0: natural
   order -- for A/B measurements and tests */
int dvlp_dev_xattn_pair_regions(int on);
/* 1: the text->image half of the local loss (its contractions and cosine passes) is issued on an internal side stream beside the
   image->text half between the softmax stages (fork / join by events, capturable; default since round 5); 0: everything on the caller's stream */
int dvlp_dev_xattn_parallel_halves(int on);
/* TIMING-ONLY ablation of the bf16 per-pair backward kernel: leave after stage 6 (launch + dP1 rows requested), 5 (S tile staged), 1 (+ norms),
   2 (image->text pass), 3 (text->image pass); 0 in production (tools/xbwd_stages.py) */
int dvlp_dev_xattn_bwd_stop(int stage);
/* TIMING-ONLY ablation of the fused forward kernel (stop after phase n); 0 in production */
int dvlp_dev_xfused_ablate(int stop);
/* 1 (default): bf16 embeddings with B = 32 / 64 run the three B x B x 256 products of the launch on the matrix cores; 0: the
 * one-wave-per-entry form used for every other shape -- for A/B measurements and tests */
int dvlp_dev_loss_mfma(int on);

#ifdef __cplusplus
}
#endif
#endif
