/*
 * discogan_hip.h -- C ABI of the MI355X (gfx950) DiscoGAN training-step kernels.
 *
 * The reference (fasion-image-generator-project/discogan_modernized) has no FFI of its own: its
 * hot path is `torch.nn` calls made from model.py / image_translation.py.  Each entry point below
 * replaces the ATen/cuDNN op the cited reference line dispatches; a binding is a ctypes stub
 * (see INTEGRATION.md and discogan_modernized_amd/_lib.py).
 *
 * Conventions
 *   - every function returns 0 on success, a negative DG_ERR_* code otherwise; the message is
 *     available from dg_last_error() (thread-local).  Nothing aborts or throws across the ABI.
 *   - all pointers are DEVICE pointers to fp32 unless stated; the caller owns every buffer
 *     (parameters, activations, gradients, workspace).  The library never allocates or frees device
 *     memory and keeps no pointer past a call.
 *   - `stream` is a hipStream_t passed as void*; calls only enqueue work (no synchronisation) and
 *     are capturable into a hipGraph.
 *   - ACTIVATION LAYOUT: interior feature maps are "NHWC" = [N][H][W][C] contiguous (the physical
 *     layout of a torch channels_last tensor of logical shape [N,C,H,W]).  The 3-channel image side
 *     (network input / output, model.py:8,80,142) stays NCHW exactly as the reference hands it over.
 *   - WEIGHT LAYOUT: a Conv2d weight of logical shape [K,C,4,4] (model.py:11...) is stored "KRSC" =
 *     [K][4][4][C]; a ConvTranspose2d weight of logical shape [Cin,Cout,4,4] (model.py:118...) is
 *     stored [Cin][4][4][Cout] (the same rule: dim1 moved innermost).  The 3-channel edge weights
 *     ([64,3,4,4]) stay in logical contiguous order.
 *   - spatial sizes must be powers of two; channel counts multiples of 4.
 */
#ifndef DISCOGAN_HIP_H
#define DISCOGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DG_OK 0
#define DG_ERR_INVALID (-1)     /* bad argument / unsupported shape */
#define DG_ERR_WORKSPACE (-2)   /* workspace too small            */
#define DG_ERR_HIP (-3)         /* HIP runtime error              */

#define DG_ACT_NONE 0
#define DG_ACT_LEAKY 1   /* LeakyReLU(slope)   model.py:9      */
#define DG_ACT_RELU 2    /* ReLU               model.py:116    */
#define DG_ACT_SIGMOID 3 /* Sigmoid            model.py:36,143 */

typedef void* dg_stream_t;

/* Arithmetic of the conv products, a PER-CALL argument of the *_g / *_p entry points ("dtype enum"):
 *   DG_PREC_F32   exact fp32 MFMA (v_mfma_f32_32x32x2_f32);
 *   DG_PREC_BF16  operands rounded to bf16 (RNE), bf16 MFMA, fp32 accumulation (BASELINE configs[4]);
 *   DG_PREC_F32X3 fp32-accurate products on the bf16 MFMA: every operand value as three bf16 pieces (24 significand bits), six MFMAs
 *                 per product block, fp32 accumulation.
 * Entry points WITHOUT the argument use the process default (dg_set_option("bf16", n); 0 unless set), kept for tools and tests. */
#define DG_PREC_DEFAULT (-1)  /* "whatever the process default is" (tools / tests that flip dg_set_option("bf16")) */
#define DG_PREC_F32 0
#define DG_PREC_BF16 1
#define DG_PREC_F32X3 2

/* Grouped launches (round 4).  The reference issues the passes of an iteration in independent pairs of identical shape --
 * G_B(A) | G_A(B), G_A(AB) | G_B(BA), D_A(A) | D_B(B), D_A(BA) | D_B(AB) (image_translation.py:342-361) -- and each discriminator
 * sees real and fake images with the SAME weights (:353-354,360-361).  A *_g entry point takes `groups` (1..DG_MAX_GROUPS) such
 * problems -- identical geometry, one pointer per problem in every pointer table -- and issues ONE launch per kernel of the op
 * (a block index selects the problem); each problem's result is bitwise what the one-problem call computes.  Where several problems
 * accumulate into the same tensor (`share`: a discriminator's real and fake pass into one weight / BatchNorm-parameter gradient) the
 * final reduction adds them in problem order, bitwise what consecutive accumulating calls leave. */
#define DG_MAX_GROUPS 4

int dg_version(void);
/* 0 = the product library.  bit 0: experiments build (csrc/Makefile EXPERIMENTS=1: kernels that lost their A/B, behind options "x3_mfma",
 * "dgw_persist" and the window-forward / plane-reader planner codes); bit 1: timing build (TIMING=1: option "dbg_zero"). */
int dg_build_flags(void);
const char* dg_last_error(void);

/* ---- tuning knobs (process-global, for benchmarking and tests; 0 = heuristic) ----------------
 * "kt" 16|32: K-tile of the conv kernels;  "splitk" n: force n K-splits;  "target_wgs" n: workgroups a split
 * grid aims for (512);  "split_below" n: split K only when the tile grid has fewer workgroups (256);
 * "pointer_path" 1: use the 64-bit addressing kernels that tensors >= 2 GiB fall back to;
 * "no_xcd_group" 1: plain blockIdx -> tile order (default: workgroups sharing operand rows are placed on one XCD); 3: only the
 *   row-tile blocks per XCD of single-column forward convs off (same-box A/B);
 * "bf16" 1: the interior conv GEMMs (dg_conv_fwd/_dgrad/_wgrad and their named wrappers) round both operands to
 * bf16 and multiply on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; every tensor stays fp32 (BASELINE
 * configs[4]).  The result equals the fp32 op applied to the rounded operands up to summation order.  With "bf16" 1 the
 * 3-channel edge layers also round their operands and run on the bf16 MFMA ("kt" 16 keeps their fp32-MFMA kernels);
 * "bf16" 2: fp32-accurate products from three bf16 planes per operand (f32x3);
 * "no_dma" 1: convolutions with two bf16 operands stay on the register-staged tiles instead of the LDS-DMA kernel;
 * "bn_items" 1: the fp32 BatchNorm apply passes (forward and backward) on the per-item kernels instead of the row-geometry ones
 *   (same results bit for bit; same-box A/B and the test that says so);
 * "dma_mfma" 32: the LDS-DMA kernel's 32x32x16 body instead of the default 16x16x32 one; 1: keep the input-grads with <= 128
 *   output channels on the register-staged tiles instead of the window kernels (same-box A/B);
 * (experiments library only, csrc/Makefile EXPERIMENTS=1: "x3_mfma" 16 = the paired-plane 16x16x32 body of the f32x3 plane kernel,
 *   "dgw_persist" 1 = the persistent form of the f32x3 window input-grad kernel -- both measured not faster, DESIGN.md 3.1;
 *   "understory" = probe forms of dg_act_fwd for tools/probe_corun.py);
 * (The operand-dropping timing switch of earlier rounds exists only in the separate timing build, csrc/Makefile TIMING=1; the product
 *   library has no option that changes results.) */
int dg_set_option(const char* name, int value);

/* ---- interior convolutions: implicit GEMM on v_mfma_f32_32x32x2_f32 --------------------------
 * Geometry is that of the *Conv2d*: x[N,H,W,C] --(k4, stride, pad)--> y[N,Ho,Wo,K].
 * Supported: (stride 2, pad 1)  Ho=H/2      -- nn.Conv2d(C,K,4,2,1)      model.py:11-31,83-103
 *            (stride 1, pad 0)  H=W=4,Ho=1  -- nn.Conv2d(C,K,4,1,0)      model.py:35,107
 * ConvTranspose2d(Cin,Cout,4,s,p) (model.py:114-142) is the dgrad of that Conv2d with the same
 * weight tensor: forward = dg_conv_dgrad, input-grad = dg_conv_fwd, weight-grad = dg_conv_wgrad
 * with the roles (x := grad_out, dy := input).  The named wrappers below spell this out.
 */
size_t dg_conv_workspace_bytes(int op /*0 fwd,1 dgrad,2 wgrad*/, int N, int H, int W, int C, int K,
                               int stride, int pad);
int dg_conv_fwd(const float* x, const float* w, float* y, int N, int H, int W, int C, int K,
                int stride, int pad, void* ws, size_t ws_bytes, dg_stream_t stream);
int dg_conv_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int K,
                  int stride, int pad, void* ws, size_t ws_bytes, dg_stream_t stream);
/* dw (+)= sum_pixels dy (x) im2col(x); accumulate!=0 adds into dw */
int dg_conv_wgrad(const float* dy, const float* x, float* dw, int N, int H, int W, int C, int K,
                  int stride, int pad, int accumulate, void* ws, size_t ws_bytes, dg_stream_t stream);

/* Grouped forms with the arithmetic as an argument.  stat (may be NULL): fused BatchNorm partial statistics, one buffer of
 * dg_conv_bnstats_rows_p(...) x (3 * columns + 4) floats per problem; ws: one workspace of ws_bytes >= dg_conv_workspace_bytes_p(...)
 * per problem (NULL entries allowed when that is 0).  dg_conv_wgrad_g: share > 1 = every `share` consecutive problems name the same dw.
 * plan_groups: 1 = every problem gets the split-K plan of a launch of its own (results bitwise those of the one-problem call);
 * `groups` = the plan is sized for the whole launch -- the group fills the chip, so each problem is cut into fewer K-slabs (less slab
 * traffic, shorter reduction): the same products in another, equally fixed, summation order. */
size_t dg_conv_workspace_bytes_p(int op, int N, int H, int W, int C, int K, int stride, int pad, int prec, int plan_groups);
int dg_conv_bnstats_rows_p(int op, int N, int H, int W, int C, int K, int stride, int pad, int prec);
int dg_conv_plan_splits_p(int op, int N, int H, int W, int C, int K, int stride, int pad, int prec, int plan_groups);
int dg_conv_fwd_g(int groups, const float* const* x, const float* const* w, float* const* y, int N, int H, int W, int C, int K, int stride,
                  int pad, int prec, int plan_groups, float* const* stat, size_t stat_floats, void* const* ws, size_t ws_bytes, dg_stream_t stream);
int dg_conv_dgrad_g(int groups, const float* const* dy, const float* const* w, float* const* dx, int N, int H, int W, int C, int K, int stride,
                    int pad, int prec, int plan_groups, float* const* stat, size_t stat_floats, void* const* ws, size_t ws_bytes, dg_stream_t stream);
int dg_conv_wgrad_g(int groups, int share, const float* const* dy, const float* const* x, float* const* dw, int N, int H, int W, int C, int K,
                    int stride, int pad, int prec, int plan_groups, int accumulate, void* const* ws, size_t ws_bytes, dg_stream_t stream);

/* Inference path (inference.py:149,172-187: generator.eval() forward): [Conv2d | ConvTranspose2d] -> BatchNorm2d(eval)
 * -> LeakyReLU/ReLU as ONE kernel.  The caller folds the BatchNorm scale into the weights (w * gamma*invstd per output
 * channel) and passes the shift as bias[out channels] = beta - running_mean*gamma*invstd; y = act(conv(x, w) + bias).
 * Same geometry / layouts as dg_conv_fwd / dg_conv_dgrad (dgrad = ConvTranspose2d forward); bias may be NULL. */
int dg_conv_fwd_bias_act(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int C, int K,
                         int stride, int pad, int act, float slope, void* ws, size_t ws_bytes, dg_stream_t stream);
int dg_conv_dgrad_bias_act(const float* dy, const float* w, const float* bias, float* dx, int N, int H, int W, int C, int K,
                           int stride, int pad, int act, float slope, void* ws, size_t ws_bytes, dg_stream_t stream);

/* Stride-2 conv forward / dgrad (= ConvTranspose2d forward) that ALSO emit BatchNorm partial statistics
 * of the output from the kernel epilogue (or from the split-K reduction), saving the separate read
 * pass of dg_bn_train_stats.  stat: [rows][3*cols + 4] floats with rows = dg_conv_bnstats_rows(op,...)
 * and cols = K (op 0, fwd) or C (op 1, dgrad); consume with dg_bn_stats_from_partials. */
int dg_conv_bnstats_rows(int op, int N, int H, int W, int C, int K, int stride, int pad);
/* number of K splits the plan for this shape uses (1 = no split-K reduction kernel); 0 on bad geometry */
int dg_conv_plan_splits(int op, int N, int H, int W, int C, int K, int stride, int pad);
int dg_conv_fwd_bnstats(const float* x, const float* w, float* y, int N, int H, int W, int C, int K,
                        float* stat, size_t stat_floats, void* ws, size_t ws_bytes, dg_stream_t stream);
int dg_conv_dgrad_bnstats(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int K,
                          float* stat, size_t stat_floats, void* ws, size_t ws_bytes, dg_stream_t stream);

/* named wrappers; H,W always name the LARGER spatial side of the layer */
int dg_conv4x4s2_fwd(const float* x, const float* w, float* y, int N, int H, int W, int C, int K,
                     void* ws, size_t ws_bytes, dg_stream_t s);
int dg_conv4x4s2_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int K,
                       void* ws, size_t ws_bytes, dg_stream_t s);
int dg_conv4x4s2_wgrad(const float* dy, const float* x, float* dw, int N, int H, int W, int C, int K,
                       int accumulate, void* ws, size_t ws_bytes, dg_stream_t s);
int dg_conv4x4_valid_fwd(const float* x, const float* w, float* y, int N, int C, int K,
                         void* ws, size_t ws_bytes, dg_stream_t s);
int dg_conv4x4_valid_dgrad(const float* dy, const float* w, float* dx, int N, int C, int K,
                           void* ws, size_t ws_bytes, dg_stream_t s);
int dg_conv4x4_valid_wgrad(const float* dy, const float* x, float* dw, int N, int C, int K,
                           int accumulate, void* ws, size_t ws_bytes, dg_stream_t s);
/* ConvTranspose2d(Cin,Cout,4,2,1): x[N,Hin,Win,Cin] -> y[N,2Hin,2Win,Cout]; w [Cin][4][4][Cout] */
int dg_convT4x4s2_fwd(const float* x, const float* w, float* y, int N, int Hin, int Win, int Cin, int Cout,
                      void* ws, size_t ws_bytes, dg_stream_t s);
int dg_convT4x4s2_dgrad(const float* dy, const float* w, float* dx, int N, int Hin, int Win, int Cin, int Cout,
                        void* ws, size_t ws_bytes, dg_stream_t s);
int dg_convT4x4s2_wgrad(const float* dy, const float* x, float* dw, int N, int Hin, int Win, int Cin, int Cout,
                        int accumulate, void* ws, size_t ws_bytes, dg_stream_t s);
/* ConvTranspose2d(Cin,Cout,4,1,0) on a 1x1 input: x[N,Cin] -> y[N,4,4,Cout] */
int dg_convT4x4_1to4_fwd(const float* x, const float* w, float* y, int N, int Cin, int Cout,
                         void* ws, size_t ws_bytes, dg_stream_t s);
int dg_convT4x4_1to4_dgrad(const float* dy, const float* w, float* dx, int N, int Cin, int Cout,
                           void* ws, size_t ws_bytes, dg_stream_t s);
int dg_convT4x4_1to4_wgrad(const float* dy, const float* x, float* dw, int N, int Cin, int Cout,
                           int accumulate, void* ws, size_t ws_bytes, dg_stream_t s);

/* ---- 3-channel edge layers (image side stays NCHW) ------------------------------------------
 * w is the logical contiguous [K][3][4][4] tensor in all three (Conv2d(3,K) weight, model.py:8,80;
 * ConvTranspose2d(K,3) weight, model.py:142).
 * c3_fwd  : y_nhwc[N,H/2,W/2,K] = act(conv_s2(x_nchw[N,3,H,W], w))        conv1 forward (act=LEAKY)
 *                                                                        / last-convT input-grad
 * c3_dgrad: dx_nchw[N,3,H,W] = act(conv_s2_dgrad(dy_nhwc[N,H/2,W/2,K], w)) last-convT forward
 *                                                                        (act=SIGMOID) / conv1 dgrad
 * c3_wgrad: dw[K][3][4][4] (+)= sum dy_nhwc (x) im2col(x_nchw)
 */
int dg_conv4x4s2_c3_fwd(const float* x_nchw, const float* w, float* y_nhwc, int N, int H, int W, int K,
                        int act, float slope, dg_stream_t s);
size_t dg_c3_dgrad_workspace_bytes(int K);
int dg_conv4x4s2_c3_dgrad(const float* dy_nhwc, const float* w, float* dx_nchw, int N, int H, int W, int K,
                          int act, void* ws, size_t ws_bytes, dg_stream_t s);
size_t dg_c3_wgrad_workspace_bytes(int N, int H, int W, int K);
int dg_conv4x4s2_c3_wgrad(const float* dy_nhwc, const float* x_nchw, float* dw, int N, int H, int W, int K,
                          int accumulate, void* ws, size_t ws_bytes, dg_stream_t s);
/* Same, with the backward of the layer's fused LeakyReLU/ReLU applied to dy on the fly:
 * dy_eff = dy * act'(act_out), act_out = the saved forward output (model.py:8-9, 80-81: Conv2d + in-place
 * LeakyReLU).  Saves the separate dg_act_bwd pass when only the weight gradient is needed. */
int dg_conv4x4s2_c3_wgrad_act(const float* dy_nhwc, const float* act_out_nhwc, int act, float slope,
                              const float* x_nchw, float* dw, int N, int H, int W, int K, int accumulate,
                              void* ws, size_t ws_bytes, dg_stream_t s);

/* Grouped forms of the three edge ops (K == 64: the streaming / scatter / per-wave kernels), arithmetic as an argument.
 * dg_conv4x4s2_c3_wgrad_g: act_out_nhwc NULL with act NONE; share > 1 = every `share` consecutive problems name the same dw;
 * ws: one workspace of ws_bytes >= dg_c3_wgrad_workspace_bytes(...) per problem. */
/* one-problem forms with the arithmetic as an argument (y_bf16 / dy_bf16 / io_bf16: the 64-channel NHWC side is bf16, see the *_t forms) */
int dg_conv4x4s2_c3_fwd_p(const float* x_nchw, const float* w, void* y_nhwc, int y_bf16, int N, int H, int W, int K,
                          int act, float slope, int prec, dg_stream_t s);
int dg_conv4x4s2_c3_dgrad_p(const void* dy_nhwc, int dy_bf16, const float* w, float* dx_nchw, int N, int H, int W, int K,
                            int act, int prec, void* ws, size_t ws_bytes, dg_stream_t s);
int dg_conv4x4s2_c3_wgrad_p(const void* dy_nhwc, const void* act_out_nhwc, int io_bf16, int act, float slope,
                            const float* x_nchw, float* dw, int N, int H, int W, int K, int prec, int accumulate,
                            void* ws, size_t ws_bytes, dg_stream_t s);
/* The input-gradient of nn.Conv2d(3,K,4,2,1) (model.py:8,80) with the backward of the layer's in-place LeakyReLU (model.py:9,81) applied to
 * dy on the way in -- dx = dgrad(dy * (act_out > 0 ? 1 : slope)), act_out = the layer's saved output (same layout and type as dy),
 * in_act = DG_ACT_LEAKY -- instead of a stand-alone dg_act_bwd pass in front of dg_conv4x4s2_c3_dgrad_p; bitwise the unfused result.
 * Exists where dg_c3_dgrad_act_ok(K) returns 1 (K == 64 on the scatter kernel); other arguments as dg_conv4x4s2_c3_dgrad_p. */
int dg_c3_dgrad_act_ok(int K);
int dg_conv4x4s2_c3_dgrad_act_p(const void* dy_nhwc, int dy_bf16, const void* act_out_nhwc, int in_act, float slope, const float* w,
                                float* dx_nchw, int N, int H, int W, int K, int act, int prec, void* ws, size_t ws_bytes, dg_stream_t s);
int dg_conv4x4s2_c3_fwd_g(int groups, const float* const* x_nchw, const float* const* w, float* const* y_nhwc, int N, int H, int W, int K,
                          int act, float slope, int prec, dg_stream_t s);
int dg_conv4x4s2_c3_dgrad_g(int groups, const float* const* dy_nhwc, const float* const* w, float* const* dx_nchw, int N, int H, int W, int K,
                            int act, int prec, dg_stream_t s);
int dg_conv4x4s2_c3_wgrad_g(int groups, int share, const float* const* dy_nhwc, const float* const* act_out_nhwc, int act, float slope,
                            const float* const* x_nchw, float* const* dw, int N, int H, int W, int K, int prec, int accumulate,
                            void* const* ws, size_t ws_bytes, dg_stream_t s);

/* ---- BatchNorm2d (training mode) + activation, NHWC [M][C], M = N*H*W ------------------------
 * nn.BatchNorm2d (model.py:12...; eps 1e-5, momentum 0.1, biased batch var for normalisation,
 * unbiased for running_var, num_batches_tracked int64 += 1) fused with the in-place
 * LeakyReLU(0.2)/ReLU that follows it (model.py:13,116).
 * saved: [2][C] = mean, invstd (kept for backward).
 */
size_t dg_bn_workspace_bytes(int M, int C);
int dg_bn_train_stats(const float* y, int M, int C, float eps, float momentum,
                      float* running_mean, float* running_var, int64_t* num_batches_tracked,
                      float* saved, void* ws, size_t ws_bytes, dg_stream_t s);
/* same outputs as dg_bn_train_stats, from partial rows written by dg_conv_*_bnstats (M = N*H*W rows total).
 * ws: dg_bn_partials_workspace_bytes(P, C) bytes (0 for P < 4096 rows) -- with it many rows are merged in two short launches
 * (row blocks, then channels) instead of one latency-bound one; NULL / too small: the one-launch form. */
size_t dg_bn_partials_workspace_bytes(int P, int C);
int dg_bn_stats_from_partials(const float* stat, int P, int M, int C, float eps, float momentum,
                              float* running_mean, float* running_var, int64_t* num_batches_tracked,
                              float* saved, void* ws, size_t ws_bytes, dg_stream_t s);
int dg_bn_act_fwd(const float* y, float* z, int M, int C, const float* saved, const float* gamma,
                  const float* beta, int act, float slope, dg_stream_t s);
/* dy = BN'(act'(dz)); dgamma/dbeta (+)= ; dy may alias dz */
int dg_bn_act_bwd(const float* dz, const float* y, float* dy, int M, int C, const float* saved,
                  const float* gamma, const float* beta, int act, float slope,
                  float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes, dg_stream_t s);

/* Grouped forms (fp32 tensors).  share > 1: every `share` consecutive problems are passes through the SAME BatchNorm module (a
 * discriminator's real and fake pass, image_translation.py:353-361): they name the same running_mean / running_var /
 * num_batches_tracked (dg_bn_train_stats_g) or the same dgamma / dbeta (dg_bn_act_bwd_g), and the finalize kernel applies their
 * updates one after the other in problem order.  ws: one workspace of ws_bytes >= dg_bn_workspace_bytes(M, C) per problem; the
 * backward keeps its coefficients there between its three kernels. */
int dg_bn_train_stats_g(int groups, int share, const float* const* y, int M, int C, float eps, float momentum, float* const* running_mean,
                        float* const* running_var, int64_t* const* num_batches_tracked, float* const* saved, void* const* ws, size_t ws_bytes,
                        dg_stream_t s);
int dg_bn_act_fwd_g(int groups, const float* const* y, float* const* z, int M, int C, const float* const* saved, const float* const* gamma,
                    const float* const* beta, int act, float slope, dg_stream_t s);
int dg_bn_act_bwd_g(int groups, int share, const float* const* dz, const float* const* y, float* const* dy, int M, int C,
                    const float* const* saved, const float* const* gamma, const float* const* beta, int act, float slope,
                    float* const* dgamma, float* const* dbeta, int accumulate, void* const* ws, size_t ws_bytes, dg_stream_t s);

/* ---- stand-alone activations ------------------------------------------------------------------ */
int dg_act_fwd(const float* x, float* y, size_t n, int act, float slope, dg_stream_t s);
/* out = output of the activation (in-place semantics of the reference, model.py:9,36) */
int dg_act_bwd(const float* dy, const float* out, float* dx, size_t n, int act, float slope, dg_stream_t s);
int dg_act_fwd_g(int groups, const float* const* x, float* const* y, size_t n, int act, float slope, dg_stream_t s);
int dg_act_bwd_g(int groups, const float* const* dy, const float* const* out, float* const* dx, size_t n, int act, float slope, dg_stream_t s);

/* ---- losses: scalars stay in device memory ----------------------------------------------------
 * gout points to the upstream gradient scalar (device).  ws: >= dg_loss_workspace_bytes().
 */
size_t dg_loss_workspace_bytes(void);
/* nn.MSELoss()  image_translation.py:267,349 */
int dg_mse_fwd(const float* x, const float* t, size_t n, float* loss, void* ws, size_t ws_bytes, dg_stream_t s);
int dg_mse_bwd(const float* x, const float* t, size_t n, const float* gout, float* dx, dg_stream_t s);
/* nn.BCELoss() against a constant label (image_translation.py:157-166), log clamped at -100 */
int dg_bce_fwd(const float* p, int n, float label, float* loss, void* ws, size_t ws_bytes, dg_stream_t s);
int dg_bce_bwd(const float* p, int n, float label, const float* gout, float* dp, dg_stream_t s);
/* one layer of get_fm_loss (image_translation.py:136-144): mean_j (mean_n real - mean_n fake)^2,
 * real/fake [N][J]; diff[J] kept for backward; ws >= dg_fm_workspace_bytes(N, J) */
size_t dg_fm_workspace_bytes(int N, size_t J);
int dg_fm_fwd(const float* real, const float* fake, int N, size_t J, float* diff, float* loss,
              void* ws, size_t ws_bytes, dg_stream_t s);
int dg_fm_bwd(const float* diff, int N, size_t J, const float* gout, float* dreal, float* dfake, dg_stream_t s);
/* Grouped forms: the A-side and B-side term of each loss in one launch per kernel (image_translation.py:349-350,353-365).  label: one
 * host float per problem.  dg_fm_bwd_g: dreal / dfake tables may be NULL (that side takes no gradient). */
int dg_mse_fwd_g(int groups, const float* const* x, const float* const* t, size_t n, float* const* loss, void* const* ws, size_t ws_bytes, dg_stream_t s);
int dg_mse_bwd_g(int groups, const float* const* x, const float* const* t, size_t n, const float* const* gout, float* const* dx, dg_stream_t s);
int dg_bce_fwd_g(int groups, const float* const* p, int n, const float* label, float* const* loss, dg_stream_t s);
int dg_bce_bwd_g(int groups, const float* const* p, int n, const float* label, const float* const* gout, float* const* dp, dg_stream_t s);
int dg_fm_fwd_g(int groups, const float* const* real, const float* const* fake, int N, size_t J, float* const* diff, float* const* loss,
                void* const* ws, size_t ws_bytes, dg_stream_t s);
int dg_fm_bwd_g(int groups, const float* const* diff, int N, size_t J, const float* const* gout, float* const* dreal, float* const* dfake,
                dg_stream_t s);

/* Profiling hook: the next implicit-GEMM launches write 8 int64 per workgroup into buf ({wall0, cyc0, cyc after
 * prologue, cyc after the K loop, cyc after the epilogue stores are issued, wall1, XCC<<32|HW_ID, cyc end});
 * (NULL, 0) switches it off.  wall = 100 MHz constant clock, cyc = shader clock. */
int dg_debug_igemm_stamps(void* buf, size_t bytes);

/* number of compute units of the current device (host planning aid) */
int dg_device_cu_count(void);

/* Curriculum loss mix (image_translation.py:162-166,367-382) over a vector of loss scalars, and the
 * gradient seeds of every term, each in ONE launch.  lossvec: [0,1] recon A,B; [2..4] BCE(D_A real,1),
 * BCE(D_A fake,0), BCE(D_A fake,1); [5..7] same for D_B; [8..8+nfm) FM layers of D_A; [8+nfm..8+2nfm) of D_B.
 * out8: gen_loss_A, gen_loss_B, fm_loss_A, fm_loss_B, dis_loss_A, dis_loss_B, gen_loss, dis_loss.
 * arch 0 discogan, 1 recongan, 2 gan.  which: 6 = backward of gen_loss, 7 = backward of dis_loss. */
int dg_loss_mix_fwd(const float* lossvec, float* out8, int nfm, float rate, int arch, dg_stream_t s);
int dg_loss_mix_bwd(const float* gout, float* gvec, int nfm, float rate, int arch, int which, dg_stream_t s);

/* ---- Adam over flat buffers (optim.Adam, image_translation.py:275-287) ------------------------
 * state (device, 4 x float64): [0] step count, [1] lr/(1-b1^t), [2] sqrt(1-b2^t), [3] spare.
 * dg_adam_advance increments the step and refreshes the scalars ON DEVICE (graph-capturable).
 */
int dg_adam_advance(double* state, double lr, double beta1, double beta2, dg_stream_t s);
int dg_adam_step_flat(float* p, const float* g, float* m, float* v, size_t n, const double* state,
                      float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                      dg_stream_t s);

/* nn.BCELoss against a target TENSOR (image_translation.py:157-166 materialises ones / zeros label tensors);
 * same -100 log clamp and 1e-12 backward guard as dg_bce_fwd/_bwd.  p, target: [n]. */
int dg_bce_target_fwd(const float* p, const float* target, int n, float* loss, dg_stream_t s);
int dg_bce_target_bwd(const float* p, const float* target, int n, const float* gout, float* dp, dg_stream_t s);
/* nn.HingeEmbeddingLoss(margin, mean) for targets in {+1, -1} (image_translation.py:141-142,269; with the
 * reference's all-ones targets it is x.mean()).  ws >= dg_loss_workspace_bytes(). */
int dg_hinge_fwd(const float* x, const float* y, size_t n, float margin, float* loss, void* ws, size_t ws_bytes, dg_stream_t s);
int dg_hinge_bwd(const float* x, const float* y, size_t n, float margin, const float* gout, float* dx, dg_stream_t s);

/* ---- data-parallel exchange group over RCCL / xGMI ---------------------------------------------
 * Replaces dist.init_process_group("nccl") (distributed_image_translation.py:31-38), DistributedDataParallel's
 * gradient all-reduce (:401-404,513-518), dist.barrier() (:398,573-574), destroy_process_group (:42-46).
 * One communicator per process (one process per GPU; the current HIP device at dg_dp_init is the rank's GPU).
 * Bootstrap: rank 0 calls dg_dp_get_unique_id into a HOST buffer of dg_dp_unique_id_bytes() bytes, ships it to the
 * other ranks out of band (the host layer uses the c10d TCP store), every rank calls dg_dp_init with it.
 * Collectives are in place, fp32, enqueued on the CALLER's stream (no internal stream, no synchronisation);
 * all-reduce is a SUM -- DDP's division by the world size is folded into dg_adam_step_flat's grad_scale.
 * The communicator handle is the only state the library keeps between calls.  RCCL is dlopen'ed on first use.
 * dg_dp_init is COLLECTIVE (ncclCommInitRank blocks until every rank has entered it): the host layer therefore first
 * calls dg_dp_ready on every rank -- purely local: binds RCCL, checks that a HIP device is current and that no
 * communicator exists -- and lets the ranks VOTE on the outcome out of band (dp.guarded_bootstrap: c10d store keys, no
 * collective) before any rank enters dg_dp_init, which it runs under a deadline (DG_COMM_INIT_TIMEOUT_S). */
int dg_dp_ready(int* device_out /* the current HIP device ordinal; may be NULL */);
int dg_dp_unique_id_bytes(void);
int dg_dp_get_unique_id(void* id_out_host, size_t bytes);
int dg_dp_init(int rank, int world, const void* unique_id_host, size_t bytes);
int dg_dp_world_size(void);   /* ranks in the communicator; 0 before dg_dp_init */
int dg_dp_rank(void);         /* -1 before dg_dp_init */
int dg_dp_allreduce_sum(float* buf, size_t n, dg_stream_t stream);
int dg_dp_allreduce_max(float* buf, size_t n, dg_stream_t stream);   /* in place, fp32 MAX (max-over-ranks timings) */
int dg_dp_broadcast(float* buf, size_t n, int root, dg_stream_t stream);
int dg_dp_barrier(float* scratch1 /* one device float owned by the caller */, dg_stream_t stream);
int dg_dp_destroy(void);

/* ---- image ingest (dataset.py:62-66): uint8 [N][H][W][3] -> float [N][3][H][W] = pixel / 255 -------------
 * bgr != 0 swaps the channel order (cv2.imread-style BGR sources).  H*W must be a multiple of 4. */
int dg_u8hwc_to_f32chw(const uint8_t* src, float* dst, int N, int H, int W, int bgr, dg_stream_t s);
/* ---- image preparation (dataset.py:52-66 read_images, :239-254 DiscoGANDataset._load_and_process_image) ----------------
 * uint8 [N][H][W][3] decoded rows -> float [N][3][S][S]: crop columns [x0, x0 + cw) (edges2*: the left / right half,
 * dataset.py:54,59), optional 3x3 erosion = 255 - cv2.dilate(255 - x, ones(3,3)) (domain 'A', dataset.py:53-57; neighbours
 * outside the crop do not count, like cv2's default border), cv2.resize(..., (S, S)) = INTER_LINEAR with half-pixel centres
 * and edge clamp (dataset.py:62), / 255 and CHW (dataset.py:65-66).  mode 0: float arithmetic, unrounded (what the
 * reference's float64 domain-'A' image gets); mode 1: cv2's 8-bit fixed-point path, result rounded to uint8 before / 255. */
int dg_image_prep(const uint8_t* src, float* dst, int N, int H, int W, int x0, int cw, int erode, int mode, int S, dg_stream_t s);

/* ---- bf16 shadow operands for the bf16 matrix path (option "bf16" = 1; BASELINE configs[4]) ---------------------
 * A shadow is a bf16 (RNE) copy of an fp32 tensor with the same logical layout, written by the tensor's PRODUCER so that
 * the conv kernels read half the operand bytes and convert nothing: dg_adam_step_flat_bf16 (weights, +2 B/param),
 * dg_bn_act_fwd_bf16 / dg_bn_act_bwd_bf16 (activations / gradients: +2 B/element on passes of 8 / 12 B/element),
 * dg_f32_to_bf16 (initial weight shadow; the first layer's output).  The *_mixed convolutions take either operand as fp32 (flag 0) or bf16 (flag 1)
 * and return the fp32 convolution of the RNE-rounded operands (same summation order per output element whichever operand
 * form is passed).  With BOTH operands bf16 and a GEMM of at least 192 rows and columns the work goes to the LDS-DMA kernel
 * (csrc/igemm_dma.hip: 256x256 tile, operand tiles global -> LDS by `buffer_load ... lds`; option "no_dma" 1 keeps the
 * register-staged tiles).  dg_conv_bf16_operands_ok: 0 = the shape has no bf16 kernel, 1 = register-staged bf16 tiles,
 * 2 = the LDS-DMA kernel when both operands are bf16. */
int dg_adam_step_flat_bf16(float* p, const float* g, float* m, float* v, size_t n, const double* state,
                           float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                           void* p_bf16, dg_stream_t s);
int dg_f32_to_bf16(const float* x, void* y_bf16, size_t n, dg_stream_t s);
int dg_bn_act_fwd_bf16(const float* y, float* z, void* z_bf16, int M, int C, const float* saved, const float* gamma,
                       const float* beta, int act, float slope, dg_stream_t stream);
int dg_bn_act_bwd_bf16(const float* dz, const float* y, float* dy, void* dy_bf16, int M, int C, const float* saved,
                       const float* gamma, const float* beta, int act, float slope, float* dgamma, float* dbeta,
                       int accumulate, void* ws, size_t ws_bytes, dg_stream_t stream);
int dg_conv_bf16_operands_ok(int op, int N, int H, int W, int C, int K, int stride, int pad);
/* stat (may be NULL): fused BatchNorm partial statistics of the OUTPUT, taken from the fp32 accumulators before the output is
 * rounded: dg_conv_mixed_bnstats_rows(op, ..., operand flags) rows of 3 * columns + 4 floats (0 rows: this plan emits none),
 * merged by dg_bn_stats_from_partials */
int dg_conv_mixed_bnstats_rows(int op, int N, int H, int W, int C, int K, int stride, int pad, int a_bf16, int b_bf16);
int dg_conv_fwd_mixed(const void* x, int x_bf16, const void* w, int w_bf16, void* y, int y_bf16, int N, int H, int W, int C, int K,
                      int stride, int pad, float* stat, size_t stat_floats, void* ws, size_t ws_bytes, dg_stream_t stream);
int dg_conv_dgrad_mixed(const void* dy, int dy_bf16, const void* w, int w_bf16, void* dx, int dx_bf16, int N, int H, int W, int C, int K,
                        int stride, int pad, float* stat, size_t stat_floats, void* ws, size_t ws_bytes, dg_stream_t stream);
int dg_conv_wgrad_mixed(const void* dy, int dy_bf16, const void* x, int x_bf16, float* dw, int N, int H, int W, int C, int K,
                        int stride, int pad, int accumulate, void* ws, size_t ws_bytes, dg_stream_t stream);

/* ---- fp32 operands as THREE bf16 PLANES for the f32x3 matrix path (option "bf16" = 2; DiscoGANTrainer(mfma_dtype="f32x3")) ----
 * A plane triple of an fp32 tensor v: hi = bf16(v) (RNE), mid = bf16(v - hi), lo = bf16(v - hi - mid) -- 24 significand
 * bits of v (both subtractions are exact in fp32).  Layout: plane-major, each plane in the tensor's own logical layout;
 * `*_plane` = distance between planes in BYTES (multiple of 16, >= 2 * numel; a weight inside a flat parameter group uses
 * the group's distance), dg_f32_to_bf16x3 / dg_adam_step_flat_x3 take it in ELEMENTS (multiple of 8).  Written by the
 * tensor's producer: dg_adam_step_flat_x3 (weights, +6 B/param), dg_bn_act_fwd_x3 / dg_bn_act_bwd_x3 (activations /
 * gradients), dg_f32_to_bf16x3 (everything else).  The *_x3 convolutions return the fp32 convolution with the same six
 * bf16 MFMAs per product block and the same reduction order as the fp32-pointer entry points under option "bf16" = 2
 * (bit-identical on an unsplit GEMM), without the per-element split in the conv kernel: csrc/igemm_dma_x3.hip, operand
 * planes global -> LDS by `buffer_load ... lds`, 256x256 tile.  dg_conv_x3_planes_ok: 1 = the shape has the plane kernel
 * (GEMM of at least 192 rows and columns, C % 16 == 0 forward / K % 16 == 0 input-grad; weight gradients from 96 rows; stride-2
 * input-grads with C <= 128 and a 32..128 pixel wide gradient map: csrc/igemm_dma_x3_dgw.hip), 0 = use dg_conv_fwd / _dgrad /
 * _wgrad.  Outputs are fp32; the caller keeps the fp32 tensors for BatchNorm and the element-wise kernels.
 * dg_conv_fwd_x3 with w_transposed != 0 reads the weight planes as wT[(r, s, c)][k] (k contiguous): a K-tile of the weight
 * operand is then 16 rows of 512 contiguous bytes instead of 256 pieces of 32 bytes (forward K loop 49-81 % -> ~90 % of the
 * matrix rate).  dg_x3_transpose_planes writes that copy for n conv weights of a plane buffer in one launch per 48 weights
 * ([K][J] images, J = 16 C, at element offsets w_off inside each plane; host arrays). */
int dg_f32_to_bf16x3(const float* x, void* y_planes, size_t n, size_t plane_elems, dg_stream_t s);
int dg_adam_step_flat_x3(float* p, const float* g, float* m, float* v, size_t n, const double* state,
                         float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                         void* p_planes, size_t plane_elems, dg_stream_t s);
int dg_x3_transpose_planes(const void* src_planes, void* dst_planes, size_t plane_elems, const int64_t* w_off, const int* w_K,
                           const int* w_J, int n, dg_stream_t s);
/* plane_layout / dy_layout: 0 = pixel-major planes [M][C] (the tensor's own NHWC order); 1 = QUAD-CHUNK planes
 * [M / 4][C / 16][4 pixels][16 channels] (C % 64 == 0, M % 4 == 0): the 16-channel chunk of 4 consecutive pixels is one
 * 128-byte line, a 4-pixel block of all channels stays one contiguous run.  The window input-grad kernel fetches a 16-channel
 * chunk of whole image rows per step: pixel-major that is 32 bytes of every 256- or 512-byte pixel row (each 128-byte line
 * crossed the fabric four times, PMC: 8x the algorithmic bytes); in the quad-chunk layout every fetched line is used whole.
 * Producers: the BatchNorm kernels below, for the two layer shapes per network whose input-grad dg_conv_x3_planes_ok reports
 * as 2; readers: dg_conv_dgrad_x3 (window kernel) and dg_conv_wgrad_x3 (the dy operand: its 16-pixel tile is the same block of
 * bytes, permuted).  Same products in the same order: results are bit-identical to layout 0.
 * z / dy may be NULL in the two *_x3 functions: the planes ARE the tensor (hi + mid + lo reproduces every fp32 value exactly), so
 * when all readers of the result are plane kernels the fp32 copy is not written (14 -> 10 and 26 -> 22 bytes per element). */
int dg_bn_act_fwd_x3(const float* y, float* z, void* z_planes, size_t plane_elems, int plane_layout, int M, int C, const float* saved,
                     const float* gamma, const float* beta, int act, float slope, dg_stream_t stream);
int dg_bn_act_bwd_x3(const float* dz, const float* y, float* dy, void* dy_planes, size_t plane_elems, int plane_layout, int M, int C,
                     const float* saved, const float* gamma, const float* beta, int act, float slope, float* dgamma,
                     float* dbeta, int accumulate, void* ws, size_t ws_bytes, dg_stream_t stream);
/* conv1 forward (+ fused activation) on the f32x3 path that also writes the plane triple of its output (pixel-major), so the next
 * layer's weight-gradient needs no separate split pass over the largest activation of the network (K == 64, tensors < 1 GiB) */
int dg_conv4x4s2_c3_fwd_x3(const float* x_nchw, const float* w, float* y_nhwc, void* y_planes, size_t plane_elems, int N, int H, int W,
                           int K, int act, float slope, dg_stream_t stream);
/* 0: no plane kernel for this (op, shape); 1: yes; 2: yes -- the window input-grad kernel, which prefers dy_layout 1 */
int dg_conv_x3_planes_ok(int op, int N, int H, int W, int C, int K, int stride, int pad);
/* stat (may be NULL): fused BatchNorm partial statistics of the OUTPUT, dg_conv_x3_bnstats_rows(op, ...) rows of 3 * columns + 4
 * floats in the layout of dg_conv_fwd_bnstats (merged by dg_bn_stats_from_partials): the plane kernels' epilogues (or their split-K
 * reduction) sum the fp32 accumulators per column, which removes the separate read pass of dg_bn_train_stats over the conv output. */
int dg_conv_x3_bnstats_rows(int op, int N, int H, int W, int C, int K, int stride, int pad);
int dg_conv_fwd_x3(const void* x_planes, int64_t x_plane, const void* w_planes, int64_t w_plane, int w_transposed, float* y,
                   int N, int H, int W, int C, int K, int stride, int pad, float* stat, size_t stat_floats, void* ws, size_t ws_bytes,
                   dg_stream_t stream);
int dg_conv_dgrad_x3(const void* dy_planes, int64_t dy_plane, int dy_layout, const void* w_planes, int64_t w_plane, float* dx, int N, int H, int W,
                     int C, int K, int stride, int pad, float* stat, size_t stat_floats, void* ws, size_t ws_bytes, dg_stream_t stream);
int dg_conv_wgrad_x3(const void* dy_planes, int64_t dy_plane, int dy_layout, const void* x_planes, int64_t x_plane, float* dw, int N, int H, int W,
                     int C, int K, int stride, int pad, int accumulate, void* ws, size_t ws_bytes, dg_stream_t stream);

/* ---- bf16 ACTIVATION STORAGE (option "bf16" = 1; DiscoGANTrainer(mfma_dtype="bf16", act_dtype="bf16")) --------------------
 * Feature maps and their gradients live in HBM as bf16 ONLY (no fp32 copy): the conv epilogues round the fp32
 * accumulators once (y_bf16 / dx_bf16 of the *_mixed convolutions above; never the weight gradient), BatchNorm reads
 * bf16, keeps statistics / normalisation / the backward expression in fp32 / fp64 exactly as the fp32 kernels do and
 * rounds what it stores ("bf16 MFMA + fp32 BatchNorm accum", BASELINE configs[4]).  Per element: forward 8 B instead of
 * 18 (fp32 + shadow), backward 10 B instead of 22.  `io_bf16` = every activation tensor of the call is bf16 (C % 8 == 0);
 * 0 = the fp32 kernels.  The image side (NCHW, 3 channels), weights, BatchNorm parameters / statistics, every gradient
 * of a parameter, losses and Adam stay fp32.  The K == 1 head takes a bf16 x / dx through the *_mixed entry points. */
int dg_bn_train_stats_t(const void* y, int io_bf16, int M, int C, float eps, float momentum, float* running_mean,
                        float* running_var, int64_t* num_batches_tracked, float* saved, void* ws, size_t ws_bytes,
                        dg_stream_t stream);
int dg_bn_act_fwd_t(const void* y, void* z, int io_bf16, int M, int C, const float* saved, const float* gamma,
                    const float* beta, int act, float slope, dg_stream_t stream);
int dg_bn_act_bwd_t(const void* dz, const void* y, void* dy, int io_bf16, int M, int C, const float* saved,
                    const float* gamma, const float* beta, int act, float slope, float* dgamma, float* dbeta,
                    int accumulate, void* ws, size_t ws_bytes, dg_stream_t stream);
int dg_act_bwd_t(const void* dy, const void* out, void* dx, int io_bf16, size_t n, int act, float slope, dg_stream_t stream);
/* 3-channel edge layers with the 64-channel NHWC side in bf16 (K == 64 only; the NCHW image side stays fp32) */
int dg_conv4x4s2_c3_fwd_t(const float* x_nchw, const float* w, void* y_nhwc, int y_bf16, int N, int H, int W, int K,
                          int act, float slope, dg_stream_t s);
int dg_conv4x4s2_c3_dgrad_t(const void* dy_nhwc, int dy_bf16, const float* w, float* dx_nchw, int N, int H, int W, int K,
                            int act, void* ws, size_t ws_bytes, dg_stream_t s);
int dg_conv4x4s2_c3_wgrad_t(const void* dy_nhwc, const void* act_out_nhwc /* or NULL with act NONE */, int io_bf16, int act,
                            float slope, const float* x_nchw, float* dw, int N, int H, int W, int K, int accumulate,
                            void* ws, size_t ws_bytes, dg_stream_t s);
/* feature matching on bf16 discriminator features (sums in fp32; dreal / dfake written as bf16) */
int dg_fm_fwd_t(const void* real, const void* fake, int io_bf16, int N, size_t J, float* diff, float* loss, void* ws,
                size_t ws_bytes, dg_stream_t stream);
int dg_fm_bwd_t(const float* diff, int N, size_t J, const float* gout, void* dreal, void* dfake, int io_bf16, dg_stream_t stream);

/* ---- layout helpers --------------------------------------------------------------------------- */
int dg_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, dg_stream_t s);
int dg_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, dg_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* DISCOGAN_HIP_H */
