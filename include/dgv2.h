/*
 * dgv2.h -- C ABI of libdgv2.so, the MI355X (gfx950) native kernels behind the
 * dusty_v2 G+D training hot path.
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes, no torch types; every pointer is DEVICE memory;
 *   - the CALLER allocates every output (so buffers can come from the host
 *     framework's caching allocator and launches stay hipGraph-capturable);
 *   - launches are asynchronous on `stream` (a hipStream_t passed as void*);
 *     nothing here allocates, synchronises or keeps mutable global state, so
 *     entry points are re-entrant from the autograd thread;
 *   - return value: 0 on success, a hipError_t value if the launch failed,
 *     DGV2_EINVAL (-1) for invalid arguments;
 *   - dtype: DGV2_F32 = 0, DGV2_BF16 = 1 (bf16 storage, fp32 accumulate);
 *   - activations are channels-last: [B, H, W, C] ("pixels x channels").
 *
 * Each declaration cites the reference interface it replaces (paths relative to
 * the reference tree kazuto1011/dusty-gan-v2).
 */
#ifndef DGV2_H
#define DGV2_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGV2_F32 0
#define DGV2_BF16 1
#define DGV2_FP8 2 /* OCP e4m3fn bytes; only where an entry says so (the *_fp8 / *_q8 entries) */
#define DGV2_EINVAL (-1)
#define DGV2_ENOTSUP (-3) /* valid request that this build's kernels do not cover: use the documented fallback */

/* Status words.  A few entries check a promise of the caller on the values they stage (dgv2_conv3x3_x3_fwd /
 * dgv2_conv3x3_x3_wgrad: `x_exact`; dgv2_fir_same_mfma_prep: the table contract).  A breach cannot be a return code
 * (the launch is asynchronous), and the library keeps no flag of its own: such entries take `int* status`, a device
 * int32 the CALLER owns (zero-initialised once), and OR one of the bits below into it.  The caller reads the word
 * whenever it synchronises anyway (gans.trainer.Trainer: scalar sync, validation, checkpoint; native.status_check). */
#define DGV2_STATUS_X_INEXACT 1 /* a value outside an x_exact promise was staged: that launch computed on its bf16 rounding */
#define DGV2_STATUS_FIR_TABLE 2 /* dgv2_fir_same_mfma_prep met a table entry outside its contract: the bands are unusable */

/* Library/ABI version; bumped when a signature changes. */
int dgv2_abi_version(void);

/* ---------------------------------------------------------------------------
 * fused bias + activation
 * replaces: fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale)
 *   gans/models/ops/fused_act/fused_bias_act.cpp:18-32,
 *   gans/models/ops/fused_act/fused_bias_act_kernel.cu:19-105
 * act: 1 = linear, 3 = leaky-ReLU.  grad: 0 = forward, 1 = first derivative
 * (uses `ref` = forward OUTPUT), 2 = second derivative (zero).
 * bias index = (i / step_b) % size_b; bias/ref may be NULL (= "empty tensor").
 * ------------------------------------------------------------------------- */
int dgv2_fused_bias_act(void* y, const void* x, const void* bias, const void* ref,
                        int64_t size_x, int64_t step_b, int64_t size_b,
                        int act, int grad, float alpha, float scale, int dtype, void* stream);

/* Per-channel sum over everything else: gb[c] = sum_i x[i] with c = (i/step_b)%size_b.
 * replaces: grad_input.sum(dim) in FusedLeakyReLUFunctionBackward.forward
 *   (gans/models/ops/fused_act/fused_act.py:33-45).  gb is fp32 [size_b], overwritten. */
int dgv2_bias_grad(float* gb, const void* x, int64_t size_x, int64_t step_b, int64_t size_b,
                   int dtype, void* stream);

/* Fused backward of bias + leaky-ReLU for channels-last activations: one pass produces
 * gx = (ref > 0 ? gy : alpha*gy) * scale and gb[c] = sum_rows gx[:,c] (fp32 [C]).
 * replaces: FusedLeakyReLUFunctionBackward.forward (act kernel + grad_input.sum), fused_act.py:22-45.
 * Returns DGV2_EINVAL for shapes it does not cover (C % vec != 0 or (C/vec) not dividing 256);
 * callers then use dgv2_fused_bias_act(grad=1) + dgv2_bias_grad. */
/* scratch (optional fp32 [>= 2048*C]): many-block mode with a partial-sum reduce instead of atomics. */
int dgv2_bias_act_bwd(void* gx, float* gb, const void* gy, const void* ref, int64_t rows, int C,
                      float alpha, float scale, float* scratch, int64_t scratch_elems, int dtype,
                      void* stream);
/* ... with an optional per-channel factor row_scale fp32 [C] on the STORED gradient only (gb sums the unscaled
 * one): backward of y = act(acc * row_scale[c] + b[c]) with respect to acc. */
int dgv2_bias_act_bwd_rs(void* gx, float* gb, const void* gy, const void* ref, int64_t rows, int C,
                         float alpha, float scale, const float* row_scale, float* scratch,
                         int64_t scratch_elems, int dtype, void* stream);
/* y[i] = (ydtype)(x[i] * row_scale[i % C]): the same for activation-free layers (the output heads). */
int dgv2_scale_cast(void* y, const void* x, const float* row_scale, int64_t n, int C, int xdtype, int ydtype,
                    void* stream);

/* ---------------------------------------------------------------------------
 * upfirdn2d
 * replaces: upfirdn2d_op.upfirdn2d(input[major,H,W,minor], kernel[kh,kw],
 *   up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)
 *   gans/models/ops/upfirdn2d/upfirdn2d.cpp:17-31, upfirdn2d_kernel.cu:44-425
 * out_h = (in_h*up_y + pad_y0 + pad_y1 - kh + down_y) / down_y (same for w);
 * out is [major, out_h, out_w, minor], caller-allocated.  kernel is fp32.
 * ------------------------------------------------------------------------- */
int dgv2_upfirdn2d(void* out, const void* in, const float* kernel,
                   int major, int in_h, int in_w, int minor, int kh, int kw,
                   int up_x, int up_y, int down_x, int down_y,
                   int pad_x0, int pad_x1, int pad_y0, int pad_y1, int dtype, void* stream);

/* ---------------------------------------------------------------------------
 * ring-aware separable FIR resampler (channels-last)
 * replaces: Resample.forward, gans/models/ops/common.py:105-135 (zero-insert +
 *   F.pad + depthwise F.conv2d) and its autograd transpose.
 * Per axis: out[n] = sum_i taps[i] * z[n*down + i - p0], z = zero-stuffed (x up)
 * extension of the input; W extends circularly if ring else by replication, H by
 * replication.  taps_h / taps_w: fp32 [kh] / [kw] (k <= 8; k = 1 with tap 1.0 = identity axis).
 * adjoint = 0: forward, x [B,H,W,C] -> y [B,Ho,Wo,C];
 * adjoint = 1: transpose, x is [B,Ho,Wo,C] (a gradient) -> y [B,H,W,C].
 * H, W are always the forward INPUT size, Ho, Wo the forward OUTPUT size.
 * y may be a channel slice of a wider tensor: ldy = channel stride (elements) of y
 * per pixel, ldx likewise for x (pass C for dense).
 * ------------------------------------------------------------------------- */
int dgv2_resample(void* y, const void* x, const float* taps_h, const float* taps_w,
                  int B, int H, int W, int C, int Ho, int Wo, int ldx, int ldy,
                  int kh, int up_h, int down_h, int p0_h,
                  int kw, int up_w, int down_w, int p0_w,
                  int ring, int adjoint, int dtype, void* stream);

/* Table-driven form of the same operator (the hot-path entry): per axis the sparse rows of the
 * resampling matrix, built once on the host per (spec, size, direction): idx/coef [n_out, E]
 * row-major, cnt [n_out] valid entries per row.  y[b,ho,wo,:] = sum_a sum_c coef_h[ho,a] *
 * coef_w[wo,c] * x[b, idx_h[ho,a], idx_w[wo,c], :].  Same reference lines as dgv2_resample. */
int dgv2_resample_tab(void* y, const void* x, const int* idx_h, const float* coef_h, const int* cnt_h,
                      int Eh, const int* idx_w, const float* coef_w, const int* cnt_w, int Ew,
                      int B, int C, int ldx, int ldy, int in_h, int in_w, int out_h, int out_w,
                      int dtype, void* stream);

/* y = resid + dgv2_resample_tab(x) for packed few-channel images (C in {1, 2, 4}, ld = C; y, x, resid of `dtype`):
 * `o + self.resample(skip)` of SynthesisBlock.forward (dusty_v2.py:179-180) in one launch, bit-equal to the two-launch
 * form.  DGV2_ENOTSUP for other channel counts / misaligned pointers. */
int dgv2_resample_tab_add(void* y, const void* x, const void* resid, const int* idx_h, const float* coef_h,
                          const int* cnt_h, int Eh, const int* idx_w, const float* coef_w, const int* cnt_w, int Ew,
                          int B, int C, int in_h, int in_w, int out_h, int out_w, int dtype, void* stream);
/* ... with resid entering as rscale[c] * resid + rbias[c] (fp32 [C], both or neither): `c * heads(x) + bias` of Head.forward
 * (dusty_v2.py:30-57) formed in the same store when the heads' contraction left conv2's epilogue (dgv2_modconv_pe_fwd_head). */
int dgv2_resample_tab_add_affine(void* y, const void* x, const void* resid, const float* rscale, const float* rbias,
                                 const int* idx_h, const float* coef_h, const int* cnt_h, int Eh, const int* idx_w,
                                 const float* coef_w, const int* cnt_w, int Ew, int B, int C, int in_h, int in_w, int out_h,
                                 int out_w, int dtype, void* stream);

/* dgv2_resample_tab that can also leave the sum of squares of what it wrote (the next modulated conv's input
 * statistic, ModConv2d.forward style.py:98-103: x.square().mean() for the EMA) as per-block partials, saving a
 * separate pass over the activation: sumsq fp32 [sumsq_cap] device buffer (NULL = plain dgv2_resample_tab);
 * *sumsq_used (host) = number of partials written, 0 when this launch configuration cannot provide them
 * (the caller then runs dgv2_sum_squares).  Partials feed dgv2_ema_scalar(sumsq, nsum = *sumsq_used). */
int dgv2_resample_tab_sq(void* y, const void* x, const int* idx_h, const float* coef_h, const int* cnt_h,
                         int Eh, const int* idx_w, const float* coef_w, const int* cnt_w, int Ew,
                         int B, int C, int ldx, int ldy, int in_h, int in_w, int out_h, int out_w,
                         int dtype, float* sumsq, int sumsq_cap, int* sumsq_used, void* stream);

/* Resampling (normally the ADJOINT tables of a blur/down) followed by the backward of a fused bias + leaky-ReLU in the
 * same pass:  y = R(x) * (ref > 0 ? 1 : alpha) * scale,  gb[c] = sum of the stored y over everything but the channel.
 * ref = forward OUTPUT of the activation, [B, out_h, out_w, C] contiguous like y; gb fp32 [C].
 * scratch fp32 [>= *blocks_needed * C]; a call with scratch == NULL only reports *blocks_needed.  DGV2_ENOTSUP when
 * the streaming kernel does not cover the geometry (then: dgv2_resample_tab + dgv2_bias_act_bwd).
 * replaces: Resample adjoint (common.py:105-135) + FusedLeakyReLUFunctionBackward (fused_act.py:22-45) of the
 * discriminator's conv1 -> activation -> blur/down chain (dusty_v2.py:325-345) run backwards. */
int dgv2_resample_tab_actbwd(void* y, float* gb, float* scratch, int64_t scratch_elems, int64_t* blocks_needed,
                             const void* x, const void* ref, const int* idx_h, const float* coef_h,
                             const int* cnt_h, int Eh, const int* idx_w, const float* coef_w, const int* cnt_w,
                             int Ew, int B, int C, int in_h, int in_w, int out_h, int out_w, float alpha,
                             float scale, int dtype, void* stream);

/* Same-size separable FIR (the blur in front of the discriminator's stride-2 convs, and its adjoint) on the MFMA cores,
 * bf16, x / y [B,H,W,C] contiguous:  y = R x.  Each pass is one 16x16x32 MFMA per 16 outputs x 16 channels with the
 * filter band as a constant operand (fir_mfma.hip).
 * dgv2_fir_same_mfma_prep builds the band operands ONCE from sparse-row tables exactly as dgv2_resample_tab takes them
 *   (bands: device buffer of >= *bytes_needed bytes, 16-byte aligned; bands == NULL only reports *bytes_needed).
 *   Contract on the tables (a violation raises DGV2_STATUS_FIR_TABLE in *status and the bands are unusable):
 *   |idx_h[ho][a] - ho| <= 4,  (idx_w[wo][e] - wo + 8) mod W < 24,  every coefficient -- and every sum of the
 *   coefficients of one row that name the same input -- exactly representable in bf16.  DGV2_ENOTSUP unless
 *   H % 8 == 0 and W % 32 == 0.
 * dgv2_fir_same_mfma / _actbwd: DGV2_ENOTSUP unless C % 32 == 0, H % 8 == 0, W % 32 == 0 (then: dgv2_resample_tab /
 *   dgv2_resample_tab_actbwd, whose contracts they share: same scratch protocol for the bias gradient).
 * replaces: Blur / Resample(up = down = 1) (gans/models/ops/common.py:105-135) at gans/models/dusty_v2.py:325-345, its
 *   adjoint, and FusedLeakyReLUFunctionBackward (gans/models/ops/fused_act/fused_act.py:22-45) behind it. */
int dgv2_fir_same_mfma_prep(void* bands, int64_t bands_bytes, int64_t* bytes_needed, const int* idx_h, const float* coef_h,
                            const int* cnt_h, int Eh, const int* idx_w, const float* coef_w, const int* cnt_w, int Ew,
                            int H, int W, int* status, void* stream);
int dgv2_fir_same_mfma(void* y, const void* x, const void* bands, int B, int C, int H, int W, void* stream);
int dgv2_fir_same_mfma_actbwd(void* y, float* gb, float* scratch, int64_t scratch_elems, int64_t* blocks_needed,
                              const void* x, const void* ref, const void* bands, int B, int C, int H, int W, float alpha,
                              float scale, void* stream);

/* ---------------------------------------------------------------------------
 * Fourier features (positional encoding of the laser angles)
 * replaces: FourierFeature.forward, gans/models/ops/fourier.py:77-82
 * angle fp32 [Ba, 2, H, W] (NCHW as in the module API; Ba = 1 broadcasts),
 * shift fp32 [B] or NULL (added to the azimuth channel, dusty_v2.py:267-274),
 * freqs fp32 [F, 2], phase fp32 [F];
 * out [B, H, W, ld] channels-last, writes channels [c0, c0 + 2F): sin then cos.
 * ------------------------------------------------------------------------- */
int dgv2_fourier_feature(void* out, const float* angle, const float* shift,
                         const float* freqs, const float* phase,
                         int B, int Ba, int H, int W, int F, int ld, int c0, int dtype, void* stream);

/* Angle pyramid step: sin/cos -> FIR down-2 (ring / replicate) -> atan2.
 * replaces: SynthesisBlock.downsample_angle, gans/models/dusty_v2.py:135-140.
 * in fp32 [Ba,2,H,W] (Ba = 1 broadcasts), shift fp32 [B] or NULL (added to the azimuth
 * channel first) -> out fp32 [B,2,H/2,W/2] (NCHW, both channels are angles).  taps fp32 [4]. */
int dgv2_downsample_angle(float* out, const float* in, const float* shift, const float* taps,
                          int B, int Ba, int H, int W, int ring, void* stream);

/* ---------------------------------------------------------------------------
 * batched channel GEMM = the contraction of the modulated 1x1 convolution
 * replaces: F.conv2d(x[1,B*I,H,W], w[B*O,I,1,1], groups=B) in ModConv2d.forward,
 *   gans/models/ops/style.py:105-118, and its autograd dgrad / wgrad.
 *   nn:  y[b,p,o]  = sum_i x[b,p,i] * w[b,o,i]        (forward; dgrad with w^T)
 *   tn:  gw[b,o,i] = sum_p gy[b,p,o] * x[b,p,i]       (wgrad, fp32 output)
 * x [B,P,ldx] (first I channels used), w [B,O,I], y [B,P,ldy] (first O written).
 * wstride = element stride between samples of w (0 = one weight shared by the batch).
 * ydtype = dtype of y (DGV2_F32 allowed with bf16 inputs: the heads stay fp32, dusty_v2.py:174-178).
 * Fused epilogue (nn only): y = act(y + bias[o]) with bias fp32 [O] or NULL, act 0 = none /
 * 3 = leaky-ReLU(alpha) * scale -- the FusedLeakyReLU that follows each trunk conv
 * (fused_bias_act_kernel.cu:19-65), saving one pass over the activation.
 * ------------------------------------------------------------------------- */
int dgv2_bmm_nn(void* y, const void* x, const void* w, int B, int P, int I, int O,
                int ldx, int ldy, int64_t wstride, const float* bias, int act, float alpha, float scale,
                int dtype, int ydtype, void* stream);
int dgv2_bmm_tn(float* gw, const void* gy, const void* x, int B, int P, int I, int O,
                int ldgy, int ldx, int dtype, void* stream);

/* Level-input conv of the generator with the positional encoding kept BATCH-SHARED:
 *   y[b,p,o] = act( sum_{k<Ka} xa[b,p,k] w[b,o,k] + sum_{k<Ks} xs[p,k] w[b,o,Ka+k] + bias[o] )
 * xa [B,P,Ka] = FIR-upsampled previous activation (NULL when Ka = 0), xs [P,Ks] = PE of the
 * unshifted angle grid, ONE copy for the batch (the per-sample azimuth shift of dusty_v2.py:267-274
 * is a per-frequency rotation folded into w by the host); w [B,O,Ka+Ks].  Ka, Ks multiples of the
 * 16-byte vector.  replaces: torch.cat([h, pe]) + ModConv2d contraction + FusedLeakyReLU,
 *   gans/models/dusty_v2.py:153-162, gans/models/ops/style.py:105-118.
 * tn: gw[b,o,:] = sum_p gy[b,p,o] * [xa[b,p,:] | xs[p,:]]  (fp32 [B,O,Ka+Ks]). */
int dgv2_bmm_nn_cat(void* y, const void* xa, const void* xs, const void* w, int B, int P, int Ka, int Ks,
                    int O, const float* bias, int act, float alpha, float scale, int dtype, int ydtype,
                    void* stream);
/* The same level-input conv with the block's up-sampling COMMUTED past the contraction (a 1x1 conv acts per pixel, the
 * FIR per channel): W_a . up2(h) == up2(W_a . h), so the xa columns run at a quarter of the pixels (T = W_a . h at the
 * previous level's resolution, dgv2_modconv_up_t below) and this entry evaluates
 *   y[b,p,:O] = act( row_scale * ( up2(W_a h)[b,p,:] + sum_{k<Ks} xs[p,k] w[b,:,koff+k] ) + bias )      (bf16)
 * with row_scale and gain = scale (1 + alpha) / 2 (act 3; 1 for act 0: the leaky ReLU is then the single fma
 * f' + |f'| (1 - alpha) / (1 + alpha)) already inside T and the weight image, the bias (fp32, times gain) as the C input
 * of each chain's first MFMA, and up2 as two more K-steps (of 32) of the same v_mfma_f32_16x16x32_bf16 chain (A = the wave's window of T, B = the constant interpolation
 * matrix of its 32 pixels, built in registers from the tables).
 * O in {32, 64, 128} (generator levels 4 / 3 / 2): the launch runs O / 32 slabs of 32 channels of the same kernel.
 * t [B,Hin,O/16,Win/8,16,8]: row_scale * gain * T in 8-pixel units (row, 16-channel tile, unit u, channel: pixels 8u..8u+7) and
 * wimg [B,O/32,Ks/32,2,4,16,8]: row_scale * gain * the PE columns of the prepared per-sample weights as the MFMA operand image
 * -- both written by dgv2_modconv_up_t (the caller passes it that gain), so that
 * every LDS-DMA piece of the sample walk is one contiguous 1 KB; up2 given as two-tap tables idx/coef [Hout][2],
 * [Wout][2] (low-resolution index, weight: the sparse rows of Resample(up=2), gans/models/ops/common.py:105-135, with
 * its ring / replicate extension); xs: the PE [Hout*Wout,Ks] as the MFMA fragment image [Hout*Wout/16][Ks/32][4][16][8]
 * (element [p][k] at [p/16][k/32][(k%32)/8][p%16][k%8]; a constant of the run, laid out once).  O = 32, Ks = 512
 * (generator level 4), Wout % 32 == 0,
 * Win % 8 == 0, Win >= 32; DGV2_ENOTSUP otherwise.  Contract on the (device-resident) tables, checked by the caller
 * once per table set: both W taps of output column X lie in the window [(X & ~31) / 2 - 8, +32) mod Win.
 * in_scale (device fp32 scalar or NULL): a factor c on the whole contraction, y = act(c (up2(T) + W_s PE) + bias) -- the
 * layer's input-magnitude factor when T and the image came from dgv2_modconv_up_t_lag (which cannot know it yet: the
 * statistic it emits is what updates the running mean c derives from); applied once per block to the B operands.
 * sumsq: per-block partial sums of squares of the stored y.
 * replaces: Resample(up=2) + torch.cat([h, pe]) + ModConv2d contraction + FusedLeakyReLU,
 *   gans/models/dusty_v2.py:153-162, gans/models/ops/style.py:105-118. */
int dgv2_modconv_up_fwd(void* y, const void* t, const void* xs, const void* wimg, int B, int Hout, int Wout, int Hin,
                        int Win, int Ks, int O, const int* idx_h, const float* coef_h, const int* idx_w,
                        const float* coef_w, const float* bias, const float* in_scale, int act, float alpha, float scale,
                        int dtype, float* sumsq, int sumsq_cap, int* sumsq_used, void* stream);
/* The low-resolution xa part of the commuted level-input conv and the operand images of dgv2_modconv_up_fwd:
 *   tcm [B,Hlow,O/16,Wlow/8,16,8]: T[b][o][p] = f[o] sum_{c<Ka} w[b][o][c] h[b][p][c] in 8-pixel units (per row: the units
 *   of channels 0..15, then those of channels 16..31, ...), f[o] = row_scale[o] * gain
 *   (row_scale fp32 [O] or NULL = 1: the input-magnitude factor of ModConv2d, style.py:98-103);
 *   wimg [B,O/32,Ks/32,2,4,16,8] (or NULL): f[o] w[b][32 z + 16 mt + o16][koff + 32 s + 8 kq + j] at [b][z][s][mt][kq][o16][j]
 *   (the A fragments of v_mfma_f32_16x16x32_bf16: one contiguous 1 KB per (K-step, M tile)).
 * h [B,Hlow*Wlow,Ka], w [B,O,I] (bf16); O in {32, 64, 128}, Ka in {64, 128, 256} (the fused dgv2_modconv_up_t_lag: Ka <= 128),
 * Wlow % 32 == 0, Ks % 32 == 0.
 * replaces: the xa columns of the ModConv2d contraction, gans/models/ops/style.py:105-118. */
int dgv2_modconv_up_t(void* tcm, void* wimg, const void* h, const void* w, const float* row_scale, float gain, int B,
                      int Hlow, int Wlow, int Ka, int Ks, int O, int I, int koff, int dtype, void* stream);
/* dgv2_modconv_up_t (row_scale = NULL) and dgv2_up2_lag_sumsq (when ghd != NULL; Gram vectors and sumsq contract as
 * there) in ONE pass over h: a wave walks a 32-column segment down the rows, its T fragments double as the operands of
 * the quadratic form's neighbourhood products.  The factor c then goes to dgv2_modconv_up_fwd as in_scale.
 * replaces: the xa columns of the ModConv2d contraction and its ema_var statistic, gans/models/ops/style.py:98-118. */
int dgv2_modconv_up_t_lag(void* tcm, void* wimg, const void* h, const void* w, float gain, const float* ghd,
                          const float* gho, const float* gwd, const float* gwo, int B, int Hlow, int Wlow, int Ka, int Ks,
                          int O, int I, int koff, int dtype, float* sumsq, int sumsq_cap, int* sumsq_used, void* stream);
/* Per-block partial sums of sum_{b,p,c} up2(h)[b,p,c]^2 taken from h at its own (low) resolution: with U = Uh (x) Uw
 * the up-2 operator, sum (U h)^2 = h^T (Gh (x) Gw) h with tridiagonal Gram matrices Gh = Uh^T Uh, Gw = Uw^T Uw (ring axis:
 * circulant).  ghd / gho [Hin]: diagonal and (i, i+1) entries of Gh (gho[Hin-1] = 0); gwd / gwo [Win]: diagonal and
 * (j, j+1 mod Win) entries of Gw.  h [B,Hin,Win,C] bf16, C % 8 == 0, 256 % (C/8) == 0.  sumsq / cap / used as in
 * dgv2_resample_tab_sq (DGV2_ENOTSUP when cap is too small).
 * replaces: the statistic mean(x^2) of ModConv2d's ema_var update on the up-sampled input, gans/models/ops/style.py:98-103
 *   (x = Resample(up=2)(h), gans/models/dusty_v2.py:153-155), without a pass at the up-sampled size. */
int dgv2_up2_lag_sumsq(const void* h, const float* ghd, const float* gho, const float* gwd, const float* gwo, int B,
                       int Hin, int Win, int C, int dtype, float* sumsq, int sumsq_cap, int* sumsq_used, void* stream);
/* dgv2_bmm_nn / dgv2_bmm_nn_cat that also leave per-block partial sums of squares of the stored outputs
 * (contract as in dgv2_resample_tab_sq), with an optional per-output-channel factor ahead of the bias:
 * y = act(acc * row_scale[o] + bias[o]) (row_scale fp32 [O] or NULL) -- the input-magnitude factor of
 * ModConv2d (style.py:98-103) when the weights were prepared by dgv2_mod_prep_all_fwd. */
int dgv2_bmm_nn_sq(void* y, const void* x, const void* w, int B, int P, int I, int O, int ldx, int ldy,
                   int64_t wstride, const float* row_scale, const float* bias, int act, float alpha, float scale,
                   const void* resid /* optional [B,P,ldy] in ydtype, added to the result */, int dtype, int ydtype,
                   float* sumsq, int sumsq_cap, int* sumsq_used, void* stream);
int dgv2_bmm_nn_cat_sq(void* y, const void* xa, const void* xs, const void* w, int B, int P, int Ka, int Ks,
                       int O, const float* row_scale, const float* bias, int act, float alpha, float scale, int dtype, int ydtype,
                       float* sumsq, int sumsq_cap, int* sumsq_used, void* stream);
int dgv2_bmm_tn_cat(float* gw, const void* gy, const void* xa, const void* xs, int B, int P, int Ka, int Ks,
                    int O, int dtype, void* stream);
/* The same per-sample weight gradient for O <= 4 output channels (the generator's output heads, dusty_v2.py:32-57):
 * a streaming weighted column sum instead of a 2-of-16-rows GEMM.  DGV2_ENOTSUP outside its shapes. */
int dgv2_bmm_tn_small(float* gw, const void* gy, const void* x, int B, int P, int I, int O, int dtype,
                      void* stream);
/* ... and their data gradient y[b,p,k] = sum_{o<O} x[b,p,o] w[b,k,o] (+ resid), a contraction of O <= 4 terms. */
int dgv2_bmm_nn_small(void* y, const void* x, const void* w, const void* resid, int B, int P, int O, int K,
                      int dtype, void* stream);
/* ... followed (ref != NULL) by the activation backward of the upstream bias + leaky-ReLU layer whose OUTPUT `ref`
 * [B,P,K] is this operator's forward input (the heads read the trunk's last activation):
 *   y = (x.w^T + resid) * (ref > 0 ? 1 : alpha) * ascale * row_scale[k];  gb[k] = column sums of the value before
 *   row_scale (fp32 [K]) -- FusedLeakyReLUFunctionBackward (fused_act.py:22-45) without its own pass.
 * scratch fp32 [>= *blocks_needed * K]; y == NULL with blocks_needed != NULL only reports the block count. */
int dgv2_bmm_nn_small_act(void* y, const void* x, const void* w, const void* resid, int B, int P, int O, int K,
                          const void* ref, const float* row_scale, float alpha, float ascale, float* gb,
                          float* scratch, int64_t scratch_elems, int64_t* blocks_needed, int dtype, void* stream);
/* The whole backward of a level's output heads in ONE streaming pass + one reduce launch (round 5): the heads' input
 * x = ref [B,P,K] is read once (operand of the head weight gradient AND reference of the upstream layer's activation
 * backward), the skip gradient gyf fp32 [B,P,O] once:
 *   g = T(gyf * cvec[o]);  y = ((g . w[b]^T) + resid) * (ref > 0 ? 1 : alpha) * ascale * row_scale[k]  (as _small_act);
 *   gb_up[k] = column sums of the unscaled rounded y;  gw[b,o,k] = sum_p g[p,o] ref[p,k];  gbh[o] = sum gyf[.,o].
 * replaces: the autograd of the two ModConv2d heads of a SynthesisBlock (gans/models/dusty_v2.py:30-57,171-178 on
 *   gans/models/ops/style.py:105-118) and of the FusedLeakyReLU in front of them (fused_act.py:22-45) -- five launches per
 *   level here before (column sum, scale + cast, dgv2_bmm_nn_small_act, dgv2_bmm_tn_small, reducers / zero fills).
 * scratch: fp32 [>= *blocks_needed * (K + O*K + O)]; y == NULL with blocks_needed: query.  DGV2_ENOTSUP as _small_act. */
int dgv2_head_bwd(void* y, float* gw, float* gbh, float* gb_up, float* scratch, int64_t scratch_elems,
                  int64_t* blocks_needed, const float* gyf, const float* cvec, const void* w, const void* resid,
                  const void* ref, const float* row_scale, float alpha, float ascale, int B, int P, int O, int K, int dtype,
                  void* stream);

/* The same contraction as dgv2_bmm_nn_cat, organised for the two top pyramid levels where it dominates
 * the generator ((Ka, Ks, O) = (64, 512, 32) and, as two 32-channel slabs, (128, 512, 64); bf16): a block owns a tile of pixels and walks the
 * samples, the shared PE fragments stay in registers, xa streams straight into registers and only the
 * per-sample weights go through LDS (LDS-DMA, double-buffered) -- HBM sees xa and y once, the PE once per block.
 * Returns DGV2_EINVAL for any other shape / dtype (use dgv2_bmm_nn_cat). */
int dgv2_modconv_pe_fwd(void* y, const void* xa, const void* xs, const void* w, int B, int P, int Ka, int Ks,
                        int O, const float* bias, int act, float alpha, float scale, int dtype, void* stream);
/* ... with the per-block sum-of-squares partials of the stored outputs (contract as in dgv2_resample_tab_sq). */
int dgv2_modconv_pe_fwd_sq(void* y, const void* xa, const void* xs, const void* w, int B, int P, int Ka, int Ks,
                           int O, const float* row_scale, const float* bias, int act, float alpha, float scale, int dtype, float* sumsq,
                           int sumsq_cap, int* sumsq_used, void* stream);
/* ... and with the contraction of the level's two 1-channel output heads taken in this layer's epilogue instead of by a
 * pass of their own over y: head_out[b,p,j] = sum_o y[b,p,o] head_w[b,j,o], j < 2 (on the stored bf16 values, fp32 sums;
 * the heads' input-magnitude factor and bias are applied by the caller: they depend on the statistic of y this very launch
 * produces).  head_w [B,2,O] bf16 (the heads' prepared weights), head_out fp32 [B,P,2]; both or neither.
 * replaces: the ModConv2d contractions of Head.forward (gans/models/dusty_v2.py:30-57,171-178) behind conv2 of a
 * SynthesisBlock (:161-170).  PE-free one-slab shapes (Ka, O) = (32, 32), (64, 64) (conv2 of levels 4 / 3); DGV2_ENOTSUP for
 * any other shape when head_w is given. */
int dgv2_modconv_pe_fwd_head(void* y, const void* xa, const void* xs, const void* w, int B, int P, int Ka, int Ks,
                             int O, const float* row_scale, const float* bias, int act, float alpha, float scale, int dtype,
                             float* sumsq, int sumsq_cap, int* sumsq_used, const void* head_w, float* head_out, void* stream);

/* Data gradient of a PE-free K -> K modulated 1x1 layer (conv2 of a generator level) FUSED with the activation backward
 * of the layer that produced its input (conv1 of the level): one pass over the gradient instead of three.
 * replaces: the autograd chain ModConv2d (data gradient, gans/models/ops/style.py:105-118) -> FusedLeakyReLU backward
 *   (fused_act.py:46-59, fused_bias_act_kernel.cu:19-60) between the two convs of a SynthesisBlock (dusty_v2.py:160-170)
 *   g[b,p,k]    = sum_o gy[b,p,o] wt[b,k,o]                    (rounded to bf16)
 *   t           = (yref[b,p,k] > 0 ? g : g * alpha) * scale
 *   gpre[b,p,k] = bf16(t * up_scale[k]),   gb[k] = sum_{b,p} bf16(t)
 * -- bit for bit what dgv2_modconv_pe_fwd (Ks = 0) followed by dgv2_bias_act_bwd_rs produce.
 * gy [B,P,K], wt [B,K,K], yref [B,P,K] (the upstream layer's forward output), gpre [B,P,K]: bf16; up_scale, gb fp32 [K];
 * scratch fp32 [scratch_elems] >= *rows_needed * K.  scratch == NULL: only *rows_needed is written, nothing is launched.
 * K in {32, 64} (generator levels 4 / 3), dtype DGV2_BF16; DGV2_ENOTSUP otherwise. */
int dgv2_modconv_pe_dgrad_actbwd(void* gpre, float* gb, float* scratch, int64_t scratch_elems, int64_t* rows_needed,
                                 const void* gy, const void* wt, const void* yref, const float* up_scale, float alpha,
                                 float scale, int B, int P, int K, int dtype, void* stream);

/* Per-sample weights of the modulated conv, written directly as the GEMM operand, and the exact
 * backward of that preparation (max-normalisations, modulation, demodulation, input-magnitude
 * scaling, optional rotation of the positional-encoding columns by shift_b * fw).
 * replaces: ModConv2d.forward weight path, gans/models/ops/style.py:72-103 (and its autograd).
 * W fp32 [O,I]; s fp32 [B,I] (style after the affine); ema_var fp32 [1]; wb [B,Otot,I] (this layer's
 * rows at [row_off, row_off+O)), dtype wb_dtype; stats fp32 [2+2B] and dsave fp32 [B,O] are produced
 * by fwd and consumed by bwd; shift fp32 [B] / fw fp32 [F] or NULL (F must be 256, PE columns
 * [cin, cin+2F)); I <= 1024.  bwd: G fp32 [B,Otot,I] -> gW fp32 [O,I], gs fp32 [B,I]; corr fp32 [1]. */
/* Adam (torch.optim.Adam semantics, no weight decay / amsgrad; the optimizers of gans/trainer.py:142-171) over a
 * list of L <= 72 fp32 parameter tensors in one launch -- the tensors of torch's own optimizer state, addresses
 * by value (HOST arrays of device pointers).  dgv2_adam_prep advances the device step counter and leaves the
 * bias corrections of that step in sc (fp32 [4]) for dgv2_adam_step. */
/* dst[l] <- lerp(dst[l], src[l], weight), L <= 72 fp32 tensors, one launch: ema_inplace (trainer.py:30-41). */
int dgv2_lerp_list(float* const* dst, const float* const* src, const int* n, int L, float weight, void* stream);
int dgv2_adam_prep(float* sc, float* step, float b1, float b2, void* stream);
int dgv2_adam_step(float* const* p, const float* const* g, float* const* m, float* const* v, const int* n,
                   int L, const float* sc, float lr, float b1, float b2, float eps, void* stream);

/* Pack / unpack a list of L <= 48 small fp32 matrices (HOST array of device pointers, by value in the launch)
 * into / out of one zero-padded [L, Rmax, Cmax] tensor: the 19 style affines of the generator (EqualLR Linear
 * of every ModConv2d, style.py:30,75) then run as one batched library GEMM forward and two backward. */
int dgv2_pack2d(float* packed, const float* const* src, const int* rows, const int* cols, int L, int Rmax,
                int Cmax, void* stream);
int dgv2_unpack2d(float* const* dst, const float* packed, const int* rows, const int* cols, int L, int Rmax,
                  int Cmax, void* stream);

/* Input-magnitude EMA of ModConv2d (style.py:98-103) as one scalar launch:
 * if update: ema <- lerp(ema, (sum(sumsq[0..nsum)) + add) * inv_count, weight); snapshot[0] <- ema[0].
 * sumsq (NULL allowed): the per-block partial sums dgv2_sum_squares leaves (nsum = 512).
 * cvec (NULL allowed): cvec[0..ncvec) <- 1/(sqrt(ema)+1e-8), the factor the layer applies to its output rows
 * (row_scale of the GEMM entries below); snapshot may then be NULL. */
int dgv2_ema_scalar(float* ema, float* snapshot, const float* sumsq, int nsum, float add, float inv_count,
                    float weight, int update, float* cvec, int ncvec, void* stream);
/* The same update for n <= 8 layers that share their input -- the output heads of one generator level
 * (dusty_v2.py:32-57, one ModConv2d per output with its own ema_var): emas / rows are HOST arrays (device pointers /
 * row counts, by value in the launch); layer i fills rows[i] entries of cvec behind those of the layers before it. */
int dgv2_ema_scalar_group(float* const* emas, const int* rows, int n, const float* sumsq, int nsum, float add,
                          float inv_count, float weight, int update, float* cvec, void* stream);
int dgv2_mod_prep_fwd(void* wb, float* dsave, float* stats, const float* W, const float* s,
                      const float* ema_var, const float* shift, const float* fw, int B, int O, int I,
                      int Otot, int row_off, int demod, int cin, int F, int wb_dtype, void* stream);
int dgv2_mod_prep_bwd(float* gW, float* gs, float* corr, const float* G, const float* W, const float* s,
                      const float* stats, const float* dsave, const float* ema_var, const float* shift,
                      const float* fw, int B, int O, int I, int Otot, int row_off, int demod, int cin,
                      int F, int corr_elems, void* stream);

/* All L <= 32 modulated layers of one generator pass prepared in one launch each way (the per-layer entries
 * above cost four launches per layer and pass).  Host arrays of length L: W[l] fp32 [O,I], s[l] fp32 [B,I],
 * fw[l] fp32 [256] (layers with flags & 2), wb[l] -> [B, Otot[l], I[l]] GEMM operand (bf16 if flags & 4 else
 * fp32; layers sharing a GEMM pass the same buffer and different row_off), dsave[l] fp32 [B,O]; stats fp32
 * [L, 2+2B]; rot fp32 [L, B, 512] scratch (the rotating layers' sin/cos table, filled by fwd, read by bwd).
 * flags: 1 demodulate, 2 rotate the PE columns [cin, cin+512) by shift[b] (shift NULL: none),
 * 4 bf16 operand.  These weights do NOT contain 1/(sqrt(ema_var)+1e-8): that factor is the GEMM's row_scale
 * (dgv2_*_sq), its gradient side dgv2_bias_act_bwd_rs / dgv2_scale_cast; so G below is dL/d(these weights).
 * bwd: out[l] = [gW (O*I) | gs (B*I) | corr scratch (ncorr[l] >= 1)] fp32, all inside flat[flat_elems] which
 * is cleared here once.  replaces: ModConv2d.forward weight path, gans/models/ops/style.py:72-103. */
int dgv2_mod_prep_all_fwd(void* const* wb, float* const* dsave, float* stats, float* rot, const float* const* W,
                          const float* const* s, const float* const* fw, const int* O, const int* I,
                          const int* Otot, const int* row_off, const int* cin, const int* flags,
                          const float* shift, int B, int L, void* stream);
int dgv2_mod_prep_all_bwd(float* flat, int64_t flat_elems, float* const* out, const int* ncorr,
                          const float* const* G, const float* const* W, const float* const* s,
                          const float* stats, const float* rot, float* const* dsave,
                          const float* const* fw, const int* O, const int* I, const int* Otot,
                          const int* row_off, const int* cin, const int* flags, const float* shift, int B,
                          int L, float* scratch, int64_t scratch_elems, void* stream);
/* scratch (fp32, 16-byte aligned, >= dgv2_mod_prep_all_bwd_scratch elements): the backward then keeps its sums over a
 * unit's samples in registers and leaves per-unit partials there, which its two fix-up kernels fold -- no atomics,
 * nothing cleared, bit-identical from run to run (needs I % 4 == 0 and cin % 4 == 0 on rotating layers; NULL or any
 * other shape: the atomic form on `flat`, which this entry clears).  The forward takes the matching float4 form under
 * the same shape conditions by itself. */
int dgv2_mod_prep_all_bwd_scratch(int64_t* elems, const int* O, const int* I, int B, int L);

/* Batched transposes in one launch: dst[l][b, c, r] = src[l][b, r, c] (r < rows[l], c < cols[l], source leading
 * dimension ld[l]); HOST arrays of L <= 32 device pointers, elem_size 2 or 4 bytes.  Produces the operands of all
 * modulated layers' data gradients (contraction over the output channels of the per-sample weights [B, O, I])
 * up front instead of one strided copy per layer in backward (ModConv2d autograd, style.py:105-118). */
int dgv2_transpose_list(void* const* dst, const void* const* src, const int* rows, const int* cols,
                        const int* ld, int L, int B, int elem_size, void* stream);

/* Sum of squares of the first C channels of x [N, ld] into acc[0] (fp32, ACCUMULATES).
 * replaces: x.pow(2).mean() in ModConv2d.forward (style.py:100-101). */
int dgv2_sum_squares(float* acc, const void* x, int64_t N, int C, int ld, int dtype, void* stream);

/* ---------------------------------------------------------------------------
 * minibatch standard deviation + concat (discriminator epilogue)
 * replaces: MinibatchStdDev.forward + torch.cat, gans/models/ops/common.py:226-250 (mbdis_feat = 1), used at
 *   gans/models/dusty_v2.py:376-385.  x [B, P, C] channels-last (P = H*W), out [B, P, Cp] same dtype with
 *   out[..., :C] = x, out[..., C] = the group statistic of the sample, out[..., C+1:] = 0 (channel padding for the
 *   conv engine, Cp > C, both multiples of the 16-byte vector).  B = splits * group * m: `splits` independent
 *   sub-batches (real | fake in one pass), group members strided by m as in the reference.  scratch: fp32
 *   [>= 64 * B / group].  bwd: gx [B, P, C] from the gradient of `out` and x.
 * ------------------------------------------------------------------------- */
int dgv2_mbstd_cat_fwd(void* out, float* scratch, const void* x, int B, int P, int C, int Cp, int splits,
                       int group, int dtype, void* stream);
int dgv2_mbstd_cat_bwd(void* gx, const void* gout, const void* x, int B, int P, int C, int Cp, int splits,
                       int group, int dtype, void* stream);
/* C[i, j] = scale * sum_{t < T} A(i, t) B(j, t) in fp32-equivalent arithmetic on the bf16 matrix cores: every operand
 * value is split into three bf16 planes (x = h + m + l, residual <= 2^-26 |x|) while its tile is staged into LDS and a
 * product is taken as six bf16 products with fp32 accumulation (the dropped terms are below 2^-25 of the product).
 * a_trans == 0: A [I, lda], t contiguous; != 0: A [T, lda], i contiguous (same for B).  C [I, ldo].  I % 64 == 0,
 * J % 128 == 0, T % 32 == 0, lda / ldb % 4 == 0 (DGV2_ENOTSUP otherwise).  splits > 1: split-K through
 * scratch [splits * I * J] (ldo == J) and a summing launch.
 * replaces: F.linear and its autograd GEMMs for EqualLR(nn.Linear(65536, 512)) of the discriminator's fp32 epilogue,
 *   gans/models/dusty_v2.py:381-383,394-395 (forward: both direct, split-K; dgrad: W transposed; wgrad: both
 *   transposed). */
int dgv2_gemm_x3(float* c, float* scratch, int64_t scratch_elems, const float* a, const float* b, int I, int J,
                 int64_t T, int a_trans, int b_trans, int64_t lda, int64_t ldb, int64_t ldo, int splits, float scale,
                 void* stream);
/* Weight gradient of the batch-shared positional-encoding columns of the modulated 1x1 convs:
 *   gw[b, o, col0 + k] = sum_p g[b, p, o] * pe[p, k], k < Ks      (gw fp32 [B, O, ldo]; g [B, P, O], pe [P, Ks] bf16)
 * 128 / O samples share a block's M tile, so one staged PE tile feeds several samples' accumulators (as a batched
 * library GEMM the encoding is re-read per sample); the pixel axis is split over blocks through `scratch`
 * (>= dgv2_pe_wgrad_scratch elements).  O divides 128, O % 8 == 0, B * O % 128 == 0, Ks % 128 == 0, P % 32 == 0.
 * replaces: the PE columns of ModConv2d's weight gradient (autograd of gans/models/ops/style.py:105-118 on the
 *   concatenated input of gans/models/dusty_v2.py:153-162). */
int dgv2_pe_wgrad_scratch(int64_t* elems, int B, int P, int O, int Ks);
int dgv2_pe_wgrad(float* gw, float* scratch, int64_t scratch_elems, const void* g, const void* pe, int B, int P, int O,
                  int Ks, int64_t ldo, int col0, void* stream);
/* The same with the epilogue's cast folded in: x / gx in xdtype, out (ydtype) / gout (gdtype) the same or fp32 for a
 * bf16 x -- x.float() of the reference's fp32 epilogue (gans/models/dusty_v2.py:394-395) and its adjoint without their
 * own passes over the activation. */
int dgv2_mbstd_cat_fwd_x(void* out, float* scratch, const void* x, int B, int P, int C, int Cp, int splits, int group,
                         int xdtype, int ydtype, void* stream);
int dgv2_mbstd_cat_bwd_x(void* gx, const void* gout, const void* x, int B, int P, int C, int Cp, int splits, int group,
                         int xdtype, int gdtype, void* stream);

/* ---------------------------------------------------------------------------
 * dense convolution with ring padding (discriminator)
 * replaces: ops.Conv2d = Pad(circular W / replicate H) + nn.Conv2d, via cuDNN/ATen
 *   gans/models/ops/common.py:10-24,187-210, used at gans/models/dusty_v2.py:325-385
 * x [B,H,W,C]; w [O,kh,kw,C] (channels-last filter); y [B,Ho,Wo,O];
 * pad on every side, Ho = (H + 2*pad - kh)/stride + 1.
 *   fwd  : y  = conv(pad(x), w), optionally followed by the fused bias + leaky-ReLU epilogue
 *          (bias fp32 [O] or NULL, act 0 / 3, alpha, scale) as in dgv2_bmm_nn
 *   dgrad: gx = pad^T(conv^T(gy, w))         (gx [B,H,W,C]; wt = w transposed to [C,kh,kw,O];
 *          gxp_scratch [B,H+2pad,W+2pad,C] holds the padded-domain gradient, NULL if pad == 0)
 *   wgrad: gw[o,ky,kx,c] = sum gy * pad(x)   (fp32 [O,kh,kw,C], overwritten)
 * ------------------------------------------------------------------------- */
int dgv2_conv_fwd(void* y, const void* x, const void* w, int B, int H, int W, int C, int O,
                  int kh, int kw, int stride, int pad, int ring,
                  const float* bias, int act, float alpha, float scale, int dtype, void* stream);
int dgv2_conv_dgrad(void* gx, void* gxp_scratch, const void* gy, const void* wt,
                    int B, int H, int W, int C, int O,
                    int kh, int kw, int stride, int pad, int ring, int dtype, void* stream);
int dgv2_conv_wgrad(float* gw, const void* gy, const void* x, int B, int H, int W, int C, int O,
                    int kh, int kw, int stride, int pad, int ring, int dtype, void* stream);

/* Direct (LDS halo-tile) weight gradient for the small-channel / large-image layers (same reference
 * lines as dgv2_conv_wgrad): gw fp32 [O, k*k, C] = sum over pixels of gy [B,Ho,Wo,O] x padded x
 * [B,H,W,C].  k in {1,3}, pad = (k-1)/2, stride in {1,2}, C % 32 == 0, O % vec == 0; bf16 or (fp32,
 * stride 1).  Returns DGV2_EINVAL for anything else (use dgv2_conv_wgrad). */
int dgv2_conv_wgrad_direct(float* gw, const void* gy, const void* x, int B, int H, int W, int C, int O,
                           int k, int stride, int pad, int ring, int dtype, void* stream);

/* Per-sample 1x1 weight gradient of the modulated conv (ModConv2d autograd, style.py:105-118) on the streaming
 * engine above: gw fp32 [B,O,C] = sum over the H*W pixels of sample b of gy [B,H,W,O] x [B,H,W,C]; splits stay
 * inside an image.  C, O multiples of the 16-byte vector and O*C a multiple of 4. */
int dgv2_bmm_tn_stream_scratch(int64_t* elems, int B, int H, int W, int C, int O, int dtype);
int dgv2_bmm_tn_stream(float* gw, float* scratch, int64_t scratch_elems, const void* gy, const void* x,
                       int B, int H, int W, int C, int O, int dtype, void* stream);
/* ... with x_shared != 0: x is ONE image [H, W, C] contracted against every sample's gy -- the weight gradient of the
 * batch-shared positional-encoding columns, gw[b,o,c] = sum_p gy[b,p,o] * pe[p,c]. */
int dgv2_bmm_tn_stream_x(float* gw, float* scratch, int64_t scratch_elems, const void* gy, const void* x,
                         int x_shared, int B, int H, int W, int C, int O, int dtype, void* stream);
/* ... with a row pitch: gw is [B, O, ldo] (ldo >= C, ldo % 4 == 0; 0 = contiguous), the columns [0, C) of the layer's
 * whole [B, O, Ka + Ks] weight gradient written in place next to the PE columns of dgv2_pe_wgrad -- no concatenation. */
int dgv2_bmm_tn_stream_ld(float* gw, int64_t ldo, float* scratch, int64_t scratch_elems, const void* gy, const void* x,
                          int x_shared, int B, int H, int W, int C, int O, int dtype, void* stream);

/* Compute-dtype copies of all conv weights of the discriminator in ONE launch (L <= 32): from each fp32
 * master [O,C,kh,kw] (EqualLR runtime scale folded in: common.py:158-184) the forward layout
 * wf [O, kh*kw, Cpad] and the data-gradient layout wt [Cpad, kh*kw, O].  HOST arrays of device pointers. */
int dgv2_conv_weight_bank(void* const* wf, void* const* wt, const float* const* src, const int* O,
                          const int* C, const int* Cpad, const int* kk, const float* scale, int L,
                          int dtype, void* stream);
/* ... with two more, optional outputs per layer (array or entry NULL: none).  w8t[l]: the staging image of the stride-1
 * data gradient dgv2_conv3x3_dgrad8 -- [Cpad / 64][O / 32][2304 units], row = t' * 64 + c % 64 with the gradient's tap
 * order t' = 8 - (ky * 3 + kx), plane = (o % 32) / 8 (needs Cpad % 64 == 0, O % 32 == 0).  w8[l]: the STAGING IMAGE of
 * the eight-wave forward conv dgv2_conv3x3_fwd8 -- [O / 64][Cpad / 32][2304 units of 16 bytes], unit (row = tap * 64 + o % 64,
 * plane = (c % 32) / 8) at (row >> 3) * 32 + plane * 8 + (row & 7), the order the kernel's staging slots read it: a
 * slab's K-chunk is one contiguous 36 KB run instead of 576 pieces of 64 bytes.  Same values as wf.  Needs kh*kw == 9,
 * O % 64 == 0, Cpad % 32 == 0, DGV2_BF16.  dtype DGV2_F32: the images are the three-plane images of
 * dgv2_conv3x3_x3_fwd / _dgrad instead -- every value split as h = bf16(v), m = bf16(v - h), l = bf16(v - h - m), one image
 * per plane in the layout above: w8 [3][O / 64][ceil(Cpad / 32)][2304] (O % 64 == 0, Cpad % 8 == 0; zeros past C),
 * w8t [3][floor(Cpad / 64)][O / 32][2304] (O % 32 == 0, Cpad >= 64).
 * replaces: the same EqualLR weights (common.py:158-210). */
int dgv2_conv_weight_bank_ex(void* const* wf, void* const* wt, void* const* w8, void* const* w8t,
                             const float* const* src, const int* O, const int* C, const int* Cpad, const int* kk,
                             const float* scale, int L, int dtype, void* stream);
/* Forward 3x3 ring conv, stride 1 or 2, pad 1, on that image (conv8.hip: eight waves per block = two groups of four that
 * share the halo tile (stride 2: two 64-channel slabs per 8 x 32 pixel tile) or the weight slab (stride 1: two pixel
 * tiles per slab)):  y [B, Hin/stride, Win/stride, O] (bf16) = act( conv(x [B,Hin,Win,Cin], w) + resid + bias ) * scale,
 * act 0 | 3 (leaky ReLU with `alpha`), bias [O] fp32 or NULL, resid like y or NULL.
 * replaces: ops.Conv2d forward (gans/models/ops/common.py:187-210) at ResidualBlock.conv1 / .conv2
 * (gans/models/dusty_v2.py:325-345).  DGV2_ENOTSUP where the engine does not cover the geometry (O % 64 (stride 1) /
 * O % 128 (stride 2), Cin % 32, Cin >= 64, at least 4 output rows and 32 / 64 output columns): callers then run
 * dgv2_conv_taps on wf. */
int dgv2_conv3x3_fwd8(void* y, const void* x, const void* w8, int B, int Hin, int Win, int Cin, int O, int stride,
                      const float* bias, const void* resid, int act, float alpha, float scale, int dtype, void* stream);
/* The stride-1 data gradient of the same conv on the transposed image w8t: gx [B,H,W,C] (bf16) from gy [B,H,W,O], the
 * replicate-row terms of rows 0 and H - 1 included, + resid (the gradient of a sibling branch of the same input, like gx,
 * or NULL).  replaces: the cuDNN data gradient autograd calls for ops.Conv2d (common.py:187-210) at ResidualBlock.conv1
 * (dusty_v2.py:329).  DGV2_ENOTSUP where the engine does not cover the geometry (C % 64, O % 32, O >= 64, H >= 8,
 * W >= 32 / 64): callers then run dgv2_conv_taps_ex on wt. */
int dgv2_conv3x3_dgrad8(void* gx, const void* gy, const void* w8t, int B, int H, int W, int C, int O, const void* resid,
                        int dtype, void* stream);
/* The STRIDE-2 data gradient of the 3x3 ring conv (pad 1) on the eight-wave engine (conv8_s2d.hip), from the transposed row
 * weights wt [C, 9, O] (dgv2_conv_weight_bank's wt): gx [B, 2 Hg, 2 Wg, C] (bf16) from gy [B, Hg, Wg, O], the replicate row of
 * output row 0 included -- two launches, one per output row parity (three or six taps, two column classes each) on eight-row
 * tiles where those fill the chip, one four-class launch on four-row tiles otherwise.
 * replaces: the cuDNN data gradient autograd calls for ops.Conv2d (common.py:187-210) at ResidualBlock.conv2
 * (dusty_v2.py:337-345).  DGV2_ENOTSUP where the engine does not cover the geometry (C % 128, O % 32, O >= 64, Hg % 4,
 * Wg % 32, DGV2_BF16): callers then run dgv2_conv_taps_ex on wt. */
int dgv2_conv3x3_s2_dgrad8(void* gx, const void* gy, const void* wt, int B, int Hg, int Wg, int C, int O, int dtype,
                           void* stream);

/* fp32 3x3 ring conv (stride 1, pad 1) on the bf16 matrix cores (conv_x3.hip): the fp32 operands as three bf16 planes
 * each (x = h + m + l), six bf16 products per multiply, fp32 accumulation -- fp32-equivalent (dropped terms < 2^-25 of a
 * product), at 3/8 of the cost of v_mfma_f32_16x16x4_f32.  w3 = the three plane images dgv2_conv_weight_bank_ex writes
 * for dtype DGV2_F32 (w8: [3][O / 64][ceil(Cx / 32)][2304 units of 8 bf16]); the activations are split while staged.
 *   y [B,H,W,O] (fp32) = act( conv(x [B,H,W,Cx] fp32, w) + bias ) * scale + resid,  act 0 | 3, bias [O] or NULL,
 *   resid like y or NULL.
 * replaces: ops.Conv2d(ch(4) + 1, ch(4), 3, 1, 1, ring) + FusedLeakyReLU of Discriminator.epilogue
 * (gans/models/dusty_v2.py:376-379) in the fp32 island of Discriminator.forward (:394-395).  DGV2_ENOTSUP where the
 * kernel does not cover the geometry (W % 32, Cx % 8, Cx >= 64, O % 64): callers then run dgv2_conv_taps in fp32. */
int dgv2_conv3x3_x3_fwd(void* y, const void* x, const void* w3, int B, int H, int W, int Cx, int x_exact, int O,
                        const float* bias, const void* resid, int act, float alpha, float scale, int* status,
                        void* stream);
/* x_exact (forward and weight gradient): the caller's promise that channels [0, x_exact) of x hold bf16-representable
 * values -- the activations of a bf16 trunk widened to fp32, which is what the discriminator's epilogue receives
 * (dusty_v2.py:394: h.to(torch.float32)).  Their planes m and l are zero, so three of the six products of a multiply are
 * products with zero: they are not issued (same sum, half the MFMAs).  0 = no promise.  The kernels check the promise on
 * the values they stage: a value that breaks it raises DGV2_STATUS_X_INEXACT in *status (the caller's device word, see
 * "Status words" above; required when x_exact > 0) -- that launch then computed with the bf16-rounded input. */
/* Its data gradient: gx [B,H,W,ldx] (fp32) from gy [B,H,W,O] fp32 -- channels [0, C) the gradient (+ resid), [C, ldx)
 * resid or zero.  w3t = the transposed plane images (w8t of dgv2_conv_weight_bank_ex, dtype DGV2_F32:
 * [3][ldx / 64][O / 32][2304 units]) serve the channels of whole 64-channel slabs; wt [ldx, 9, O] fp32 (the bank's
 * data-gradient layout) the channels behind them in exact fp32, one pass per channel (the minibatch-stddev channel of
 * 512 + 1; at most 16).  replaces: the cuDNN data gradient autograd calls for that conv.  DGV2_ENOTSUP: H < 2, W % 32,
 * O % 32, O < 64, C < 64, C % 64 > 16, ldx / 64 != C / 64. */
int dgv2_conv3x3_x3_dgrad(void* gx, const void* gy, const void* w3t, const void* wt, int B, int H, int W, int C, int ldx,
                          int O, const void* resid, void* stream);
/* Its weight gradient (conv_wgrad_stream.hip: conv_wgrad_x3_kernel, both operands split while staged):
 * gw fp32 [O, 9, C] (param_layout != 0: [O, C, 3, 3]) = scale * sum_{b,h,w} gy[b,h,w,o] xpad[b,h+ky-1,w+kx-1,c] from
 * gy [B,H,W,O], x [B,H,W,C] fp32.  Input channels [0, 64 floor(C/64)) on the matrix cores, [.., clive) in exact fp32
 * (at most 16), [clive, C) = 0 (x's padding channels).  scratch: fp32 [>= dgv2_conv3x3_x3_wgrad_scratch(...)].
 * replaces: the cuDNN weight gradient autograd calls for that conv.  DGV2_ENOTSUP: O % 128, C < 64, C % 8, W % 32,
 * more than 16 such channels. */
int dgv2_conv3x3_x3_wgrad(float* gw, float* scratch, int64_t scratch_elems, const void* gy, const void* x, int B, int H,
                          int W, int C, int clive, int x_exact, int O, float scale, int param_layout, int* status,
                          void* stream);
int dgv2_conv3x3_x3_wgrad_scratch(int64_t* elems, int B, int H, int W, int C, int clive, int O);
/* Both image sets from weight VALUES w [O, 9, Cp] fp32 (the operand layout of dgv2_conv_taps), for passes that do not
 * run on the weight bank (R1's double backward): w3 [3][O / 64][ceil(Cp / 32)][2304 units of 8 bf16], w3t
 * [3][Cp / 64][O / 32][2304 units]; either may be NULL.  O % 64 == 0, Cp % 8 == 0, Cp >= 64. */
int dgv2_conv_x3_images(void* w3, void* w3t, const void* w, int O, int Cp, void* stream);

/* Streaming weight gradient -- the hot-path engine for every discriminator conv (same reference lines as
 * dgv2_conv_wgrad).  A block keeps one (o, c) tile of gw for all k*k taps in registers and streams its
 * slice of the batch's pixel tiles; the per-slice partial sums go to `scratch` (plain stores) and are
 * reduced into gw fp32 [O, k*k, C] (overwritten).  k in {1,3}, pad = (k-1)/2, stride in {1,2}, C and O
 * multiples of the 16-byte vector.  dgv2_conv_wgrad_stream_scratch reports the fp32 element count the
 * scratch buffer must hold for a geometry. */
int dgv2_conv_wgrad_stream_scratch(int64_t* elems, int B, int H, int W, int C, int O, int k, int stride,
                                   int pad, int dtype);
int dgv2_conv_wgrad_stream(float* gw, float* scratch, int64_t scratch_elems, const void* gy, const void* x,
                           int B, int H, int W, int C, int O, int k, int stride, int pad, int ring,
                           int dtype, void* stream);
/* ... writing scale * gw, and with param_layout != 0 in the parameter's own layout [O, C, k, k] (nn.Conv2d weight,
 * common.py:187-210) instead of [O, k*k, C]: the EqualLR factor and the layout change of the weight gradient cost
 * no separate pass. */
int dgv2_conv_wgrad_stream_pl(float* gw, float* scratch, int64_t scratch_elems, const void* gy, const void* x,
                              int B, int H, int W, int C, int O, int k, int stride, int pad, int ring,
                              float scale, int param_layout, int dtype, void* stream);

/* Direct (LDS halo-tile) convolution with a generic tap list -- the hot-path engine for the
 * discriminator convs and their data gradients (same reference lines as dgv2_conv_*):
 *   y[b, gh*out_stride+ooff_h, gw*out_stride+ooff_w, o] (=|+=) act( sum_t sum_c
 *       x[b, H(gh*in_stride+ioff_h+dy_t), W(gw*in_stride+ioff_w+dx_t), c] * w[o, widx_t, c] + bias[o] )
 * for gh < Hg, gw < Wg.  x [B,Hin,Win,Cin], w [O,wtaps,Cin], y [B,Hy,Wy,O].
 * taps_host: HOST pointer to ntaps (<= 9) triples (dy, dx, widx).  H(): clamp (hzero = 0, replicate
 * padding) or zero outside [0,Hin) (hzero = 1, gradients); W(): wrap (ring = 1) or clamp.
 * forward conv: taps (ky-pad, kx-pad), in_stride = stride; stride-1 dgrad: taps (1-ky, 1-kx) on gy
 * with transposed weights; stride-2 dgrad: one launch per output parity class (out_stride = 2);
 * replicate-row border terms: one-row launches with accumulate = 1.
 * Cin must be a multiple of 32 (bf16) / 16 (fp32).  resid (optional, layout of y) is added to the
 * accumulator before bias / activation: the residual sum of ResidualBlock (dusty_v2.py:343-345) costs
 * no extra pass. */
int dgv2_conv_taps(void* y, const void* x, const void* w, int B, int Hin, int Win, int Cin,
                   int Hg, int Wg, int O, int Hy, int Wy, int in_stride, int ioff_h, int ioff_w,
                   int out_stride, int ooff_h, int ooff_w, int ntaps, int wtaps, const int* taps_host,
                   int hzero, int ring, int accumulate, const float* bias, const void* resid, int act,
                   float alpha, float scale, int dtype, void* stream);
/* The general form: OUTPUT CLASSES and BORDER EXTRAS, so that a whole data gradient is one launch.
 *   taps_host   ntaps quadruples (dy, dx, widx, cls) sorted by cls; class c is written at
 *               (gh*out_stride + cls_host[2c], gw*out_stride + cls_host[2c+1]); ncls in {1, 4}
 *               (4 = the parity classes of the stride-2 data gradient: the gy halo tile and all nine
 *               weight taps are staged once instead of once per class);
 *   extras_host nextra <= 6 quintuples (dy, dx, widx, cls, row): one more tap for output row gh == row
 *               only -- the replicate-padding rows of the H border (dusty_v2.py Pad / common.py:10-24)
 *               fold into the same launch instead of one-row accumulate launches.
 * Returns DGV2_ENOTSUP when the geometry needs the synchronous fallback kernel, which has neither
 * (callers then issue one dgv2_conv_taps launch per class / border row). */
int dgv2_conv_taps_ex(void* y, const void* x, const void* w, int B, int Hin, int Win, int Cin, int Hg,
                      int Wg, int O, int Hy, int Wy, int in_stride, int ioff_h, int ioff_w,
                      int out_stride, int ncls, const int* cls_host, int ntaps, int wtaps,
                      const int* taps_host, int nextra, const int* extras_host, int hzero, int ring,
                      int accumulate, const float* bias, const void* resid, int act, float alpha,
                      float scale, int dtype, void* stream);
/* dgv2_conv_taps_ex whose O output channels are written into rows of ldy >= O channels (y / resid point at the first
 * of them), so that one launch can produce a channel range of a wider tensor. */
int dgv2_conv_taps_ld(void* y, int ldy, const void* x, const void* w, int B, int Hin, int Win, int Cin, int Hg,
                      int Wg, int O, int Hy, int Wy, int in_stride, int ioff_h, int ioff_w, int out_stride, int ncls,
                      const int* cls_host, int ntaps, int wtaps, const int* taps_host, int nextra,
                      const int* extras_host, int hzero, int ring, int accumulate, const float* bias,
                      const void* resid, int act, float alpha, float scale, int dtype, void* stream);

/* ---------------------------------------------------------------------------
 * discriminator stem in one pass: BlurVH -> 1x1 conv (2 -> O) -> bias + leaky ReLU, and its backward.
 * replaces: Discriminator layers[0:3] (dusty_v2.py:364-367) = ops.BlurVH (common.py:141-155) +
 *   ops.Conv2d 1x1 (common.py:187-210) + FusedLeakyReLU (fused_act.py:20-129) and their autograd.
 * x fp32 [B,H,W] (one channel); w fp32 [O,2] (column 0 multiplies blur_v(x), column 1 blur_h(x)); O in
 * {8,16,32,64}; y, gy [B,H,W,O] fp32 or bf16.  Backward outputs gw fp32 [O,2], gb fp32 [O], gx fp32 [B,H,W]
 * (or NULL); scratch: fp32, element count from dgv2_stem_bwd_scratch.  First-order only: the R1 double
 * backward uses the composable ops. */
int dgv2_stem_fwd(void* y, const float* x, const float* w, const float* bias, int B, int H, int W, int O,
                  int ring, float alpha, float scale, int ydtype, void* stream);
int dgv2_stem_bwd_scratch(int64_t* elems, int B, int H, int W, int O);
int dgv2_stem_bwd(float* gx, float* gw, float* gb, float* scratch, int64_t scratch_elems, const void* gy,
                  const void* y, const float* x, const float* w, int B, int H, int W, int O, int ring,
                  float alpha, float scale, int dtype, void* stream);
/* ... with the gradient of the first ResidualBlock's skip branch folded in: gsk [B,Hs,Ws,O] (dtype) = dL/d(blur_down(y)),
 * the decimating blur in front of ResidualBlock.skip (gans/models/dusty_v2.py:337-345), gathered through the blur's ADJOINT
 * tables (idx_h / coef_h / cnt_h: H rows of Eh entries naming rows of gsk; idx_w / coef_w / cnt_w: W rows of Ew entries,
 * exactly the tables dgv2_resample_tab takes for the adjoint pass): dL/dy = gy + blur_down^T(gsk) is never materialised
 * (it was scattered to a [B,H,W,O] tensor by one pass, added to conv1's data gradient by another and read back here).
 * gsk NULL: dgv2_stem_bwd. */
int dgv2_stem_bwd_skip(float* gx, float* gw, float* gb, float* scratch, int64_t scratch_elems, const void* gy, const void* y,
                       const float* x, const float* w, const void* gsk, const int* idx_h, const float* coef_h,
                       const int* cnt_h, int Eh, const int* idx_w, const float* coef_w, const int* cnt_w, int Ew, int Hs,
                       int Ws, int B, int H, int W, int O, int ring, float alpha, float scale, int dtype, void* stream);

/* ---------------------------------------------------------------------------
 * generator output stage: cancel the azimuth shift (circular bilinear shift),
 * scale, tanh, Gumbel-sigmoid ray-drop mask, blend.
 * replaces: dusty_v2.py:290-306 (affine_grid + grid_sample + *0.25 + tanh),
 *   dusty_v1.py:20-25 (RayDropModel), ops/gumbel.py:23-29.
 * skip fp32 [B,H,W,2] (ch0 image, ch1 raydrop_logit); shift fp32 [B] radians or NULL;
 * u fp32 [B,H,W] uniforms in (0,1); outputs fp32 [B,H,W] each.
 * backward: g_image, g_image_orig, g_logit, g_mask (any may be NULL) -> g_skip [B,H,W,2];
 * scratch fp32 [B,H,W,2] is required when shift != NULL.
 * ------------------------------------------------------------------------- */
int dgv2_gen_tail_fwd(float* image, float* image_orig, float* logit, float* mask,
                      const float* skip, const float* shift, const float* u,
                      int B, int H, int W, float out_scale, float raydrop_const, float temperature,
                      void* stream);
int dgv2_gen_tail_bwd(float* g_skip, float* scratch, const float* g_image, const float* g_image_orig,
                      const float* g_logit, const float* g_mask,
                      const float* image_orig, const float* logit, const float* mask, const float* u,
                      const float* shift, int B, int H, int W, float out_scale, float raydrop_const,
                      float temperature, void* stream);

/* ---------------------------------------------------------------------------
 * ADA resampler, separable operator form:
 *   y[b] = a[b] * (Ay[b] @ x[b] @ Cx[b]^T) + c[b]
 * replaces: AdaptiveAugment.forward geometric + colour stages,
 *   gans/augment/adaptive_augment.py:471-545 (pad, 2x upfirdn2d, grid_sample,
 *   2x upfirdn2d, colour affine).
 * x, y fp32 [B,H,W]; Ay fp32 [B,H,H]; kx fp32 [B,K] circular taps:
 *   (x Cx^T)[j] = sum_t kx[t] * x[(sgn*j + off + t) mod W], sgn[b] in {+1,-1}, off[b] int.
 * transpose = 1 applies the adjoint (backward / used for double backward).
 * ------------------------------------------------------------------------- */
int dgv2_ada_apply(float* y, const float* x, const float* Ay, const float* kx, const int* off,
                   const int* sgn, const float* a, const float* c, int B, int H, int W, int K,
                   int transpose, void* stream);

/* ADA random parameters and operator construction (three launches instead of ~160 tiny tensor ops).
 * replaces: AdaptiveAugment.sample_affine / sample_color and the geometry set-up of forward,
 *   gans/augment/adaptive_augment.py:386-469, 488-535.
 * sample: u fp32 [B,16] uniforms, n fp32 [B,8] normals, p fp32 [1] (device); policy_host = HOST array of
 *   11 floats (lr_flip, ud_flip, int_trans, iso_scale, frac_trans, brightness, contrast, luma_flip, hue,
 *   saturation multipliers, h_trans_factor) -> gaff [B,4] = (sx, tx, sy, ty), a [B], c [B].
 * build: gaff + chain constants M1y [2(3H-2),H], M1x [2(3W-2),W], taps [12] -> Ay [B,H,H], kx [B,K],
 *   off [B], sgn [B] as consumed by dgv2_ada_apply. */
int dgv2_ada_sample(float* gaff, float* a, float* c, const float* u, const float* n, const float* p,
                    const float* policy_host, int B, int H, int W, void* stream);
int dgv2_ada_build(float* Ay, float* kx, int* off, int* sgn, const float* gaff, const float* M1y,
                   const float* M1x, const float* taps, int B, int H, int W, int K, void* stream);

/* ---------------------------------------------------------------------------
 * range-image projection
 * replaces: CoordBridge.convert / depth_to_point_map, gans/coords.py:88-185
 * mode: 0 depth -> inv_depth_norm (the fetch_reals path, trainer.py:211-217, incl.
 *         *2-1 and mask blend when mask != NULL), out [B,1,H,W];
 *       1 inv_depth_norm -> depth, out [B,1,H,W];
 *       2 inv_depth_norm -> point_map [B,3,H,W];  3 depth -> point_map.
 * angle fp32 [1,2,H,W].
 * ------------------------------------------------------------------------- */
int dgv2_coords_convert(float* out, const float* in, const float* mask, const float* angle,
                        int B, int H, int W, float min_depth, float max_depth, float raydrop_const,
                        int mode, void* stream);

/* ---------------------------------------------------------------------------
 * surface normals of a coordinated point map (logging / visualisation path)
 * replaces: estimate_surface_normal, gans/geometry.py:38-127
 * points / out fp32 [B,3,H,W]; neighbours at distance d (replicate rows, circular columns);
 * mode 0 "closest" (pair (k,k+2) with the smallest summed distance, first minimum), 1 "mean".
 * ------------------------------------------------------------------------- */
int dgv2_surface_normal(float* out, const float* points, int B, int H, int W, int d, int mode, void* stream);

/* ---------------------------------------------------------------------------
 * non-saturating GAN objective + the logged discriminator statistics in one launch
 * replaces: GANLoss("nsgan") gans/models/loss.py:37-41,66-69 (softplus + mean), the y.mean() scalars of
 *   gans/trainer.py:400-406 and AdaptiveAugment.cumulate's sign sum (adaptive_augment.py:368-370)
 * y fp32 [n_real + n_fake] logits, reals first (either count may be 0: loss_G = mean softplus(-y_fake) is the
 * "real" formula applied to the fakes); stats fp32 [4] = loss, mean y_real, mean y_fake, sum sign(y_real);
 * gy fp32 [n_real + n_fake] = gy_scale * d loss / d y  (gy_scale: the objective's weight cfg.training.loss.gan, so that
 * gy is the cotangent the step body hands to y.backward() directly -- no scalar-loss graph, gans/trainer.py:293-297).
 * sign_cum / n_cum (both or neither, fp32 [1]): AdaptiveAugment's running statistic, updated in the same launch
 * (+= sum sign(y_real), += n_real: AdaptiveAugment.cumulate, adaptive_augment.py:368-370).
 * ------------------------------------------------------------------------- */
int dgv2_nsgan_loss(float* stats, float* gy, const float* y, int n_real, int n_fake, float gy_scale, float* sign_cum,
                    float* n_cum, void* stream);

/* ---------------------------------------------------------------------------
 * tail of the discriminator's epilogue: FusedLeakyReLU(K) + EqualLR(Linear(K, 1)) on [B, K] in one launch each way
 * replaces: ops.FusedLeakyReLU(ch(4)) + ops.EqualLR(nn.Linear(ch(4), 1)), gans/models/dusty_v2.py:383-384
 *   (fused_leaky_relu, fused_act.py:20-59,113-129: lrelu(x + b) * scale, backward masked by the sign of the OUTPUT,
 *   fused_bias_act_kernel.cu:19-60; EqualLR, common.py:158-184: gain * (bias + scale * x W^T))
 * fwd: a[b,k] = lrelu_alpha(h[b,k] + b1[k]) * act_scale (written: saved for the backward), y[b] = gain2 * (b2[0] +
 *      scale2 * sum_k w2[k] a[b,k]).  h, a fp32 [B,K]; b1 fp32 [K] or NULL; w2 fp32 [K]; b2 fp32 [1] or NULL; y fp32 [B].
 * bwd: gh[b,k] = gy[b] * gain2 * scale2 * w2[k] * (a[b,k] > 0 ? act_scale : alpha * act_scale); optional (NULL = not
 *      wanted) gb1[k] = sum_b gh[b,k], gw2[k] = gain2 * scale2 * sum_b gy[b] a[b,k], gb2[0] = gain2 * sum_b gy[b].
 *      Sums run over the batch in index order: bit-identical from run to run.
 * ------------------------------------------------------------------------- */
int dgv2_d_tail_fwd(float* y, float* a, const float* h, const float* b1, const float* w2, const float* b2, int B, int K,
                    float alpha, float act_scale, float scale2, float gain2, void* stream);
int dgv2_d_tail_bwd(float* gh, float* gb1, float* gw2, float* gb2, const float* gy, const float* a, const float* w2,
                    int B, int K, float alpha, float act_scale, float scale2, float gain2, void* stream);

/* dst fp32 [K] <- lerp(dst, mean over the B rows of src (fp32 [B, ld], ld >= K), w) in one launch.
 * replaces: Generator.moving_average_w, gans/models/base.py:89-97 (w[:, 0].mean(0) and the lerp into w_avg). */
int dgv2_colmean_lerp(float* dst, const float* src, int B, int K, int64_t ld, float w, void* stream);

/* ---------------------------------------------------------------------------
 * every random number of one step body from ONE launch
 * replaces: the torch.randn / torch.rand calls of an iteration -- Trainer.sample_z (gans/trainer.py:206-208), the azimuth
 *   shift (gans/models/dusty_v2.py:267-274), the uniforms of GumbelSigmoid (gans/models/ops/gumbel.py:23-29), the draws of
 *   AdaptiveAugment.sample_affine / sample_color (gans/augment/adaptive_augment.py:386-470), the warm-up keep mask
 *   (gans/trainer.py:241-245).
 * Philox4x32-10 (the generator torch.cuda uses); the stream state is DEVICE memory the caller owns: state uint64[4] =
 * {seed, offset, 0, 0}.  The launch advances `offset` itself (its last block to finish), so a launch captured into a
 * hipGraph draws fresh numbers on every replay.  nseg <= 16 segments: out[s] fp32 [count[s]],
 *   kind 0: uniform in [a, b);  kind 1: normal, mean a, standard deviation b;  kind 2: u in [0, 1) clamped to [a, b];
 *   kind 3: Bernoulli(a) as 0.0 / 1.0 (the valid-return mask of a synthetic scan).
 * out / count / kind / a / b are HOST arrays (read during the call).  Launches on one state must be stream-ordered.
 * ------------------------------------------------------------------------- */
int dgv2_rng_fill(float* const* out, const int64_t* count, const int* kind, const float* a, const float* b, int nseg,
                  uint64_t* state, void* stream);

/* ---------------------------------------------------------------------------
 * KITTI scan -> range image (the front end of the real-data path)
 * replaces: KITTIRaw.load_pts_as_img (scan-unfolding spherical projection, numba scatter after an argsort by depth)
 *   and the nearest resize + mask of __getitem__, gans/datasets/kitti.py:264-279,317-370
 * pts fp32 [n,4] (x, y, z, reflectance); row int32 [n] = ring index per point from the scan order (the host derives
 * it from the quadrant sequence, :328-346; -1 means ring H-1 as in the reference) or NULL for rows from the pitch angle
 * (:347-351); key = uint64 scratch [H*W]; out fp32 [6,H,Wout] = x, y, z, reflectance, depth, mask of the NEAREST point
 * of pixel (h, w * W/Wout), multiplied by the mask (min_depth <= depth <= max_depth) when apply_mask (the item of
 * __getitem__; 0 = the raw projection of load_pts_as_img).  W % Wout == 0.
 * ------------------------------------------------------------------------- */
/* ring index per point from the scan order (scan unfolding, gans/datasets/kitti.py:328-346): pts fp32 [n,4] in file order ->
 * row int32 [n] as dgv2_kitti_project takes it (0 before the first ring boundary and for rings older than H + 1 from the
 * end, -1 for the (H+1)-th from the end like the reference); counts: int32 scratch of ceil(n / 4096) entries (delimiters per
 * block of 4096 points); two launches: count, then scan + write */
int dgv2_kitti_rows(int* row, int* counts, const float* pts, int n, int H, void* stream);
int dgv2_kitti_project(float* out, unsigned long long* key, const float* pts, const int* row, int n, int H, int W,
                       int Wout, float min_depth, float max_depth, int apply_mask, void* stream);

/* ---------------------------------------------------------------------------
 * point-cloud natives of the evaluation path
 *
 * furthest point sampling + gather
 * replaces: fps.furthest_point_sampling / gather_points / gather_points_grad
 *   gans/sampling/fps/furthest_point_sampling.cpp:26-112, furthest_point_sampling.cu:37-263
 * xyz fp32 [B, n, 3]; idxs int32 [B, m]: idxs[:, 0] = 0, then m - 1 rounds of "the point with the largest distance to
 * the selected set"; points with |p|^2 <= 1e-3 never take part; equal distances resolve in the order of the
 * reference's 2^floor(log2 n) (<= 512)-thread block reduction, so the indices are the reference's.  temp: fp32
 * scratch of dgv2_fps_scratch(B, n) floats (0 for n <= 32768, where the running distances live on chip; may be NULL then).
 * gather: points fp32 [B, C, n], idx int32 [B, m] -> out [B, C, m]; grad: grad_out [B, C, m] -> grad_points [B, C, n]
 * (zero-filled here, atomic adds for repeated indices).
 * ------------------------------------------------------------------------- */
int dgv2_fps_scratch(int64_t* floats, int B, int n);
int dgv2_fps(int* idxs, float* temp, const float* xyz, int B, int n, int m, void* stream);
int dgv2_gather_points(float* out, const float* points, const int* idx, int B, int C, int n, int m, void* stream);
int dgv2_gather_points_grad(float* grad_points, const float* grad_out, const int* idx, int B, int C, int n, int m,
                            void* stream);

/* chamfer distance: nearest neighbour in both directions
 * replaces: cd.forward_cuda / cd.backward_cuda (and the CPU twins cd.forward / cd.backward)
 *   gans/metrics/distance/cd/chamfer_distance.cpp:18-144, chamfer_distance.cu:6-190
 * xyz1 fp32 [B, n, 3], xyz2 fp32 [B, m, 3]; dist1 [B, n] = min_k |xyz1_j - xyz2_k|^2 with idx1 the FIRST minimiser
 * (strict <), dist2 / idx2 [B, m] the other direction; the squared distance is ((dx*dx + dy*dy) + dz*dz) in fp32
 * without contraction, i.e. bit-identical to the reference's CPU nnsearch.  bwd: gxyz1 / gxyz2 (zero-filled here)
 * receive 2 g (a - b) at the point and -2 g (a - b) at its neighbour for both directions (float atomics).
 * ------------------------------------------------------------------------- */
int dgv2_chamfer_fwd(float* dist1, int* idx1, float* dist2, int* idx2, const float* xyz1, const float* xyz2, int B,
                     int n, int m, void* stream);
/* one direction of the above: nearest neighbour of every xyz point [B, n, 3] in ref [B, m, 3], or in ONE set ref [m, 3]
 * shared by all clouds (ref_shared = 1: the occupancy-grid voting of the JSD metric, gans/metrics/jsd.py:46-62) */
int dgv2_nn_search(float* dist, int* idx, const float* xyz, const float* ref, int B, int n, int m, int ref_shared,
                   void* stream);
int dgv2_chamfer_bwd(float* gxyz1, float* gxyz2, const float* xyz1, const float* xyz2, const float* gdist1,
                     const float* gdist2, const int* idx1, const int* idx2, int B, int n, int m, void* stream);

/* earth mover's distance by approximate matching
 * replaces: emd.approxmatch_forward / matchcost_forward / matchcost_backward
 *   gans/metrics/distance/emd/earth_mover_distance.cpp:26-95, earth_mover_distance.cu:3-364
 * xyz1 fp32 [B, n, 3], xyz2 fp32 [B, m, 3]; match fp32 [B, m, n] (written, not accumulated); temp fp32
 * [B, 2 (n + m)] scratch; cost fp32 [B] = sum match[l, k] |xyz1_k - xyz2_l|; grad1 [B, n, 3], grad2 [B, m, 3] =
 * d cost / d xyz for a FIXED match (the reference does not differentiate the matching).
 * ------------------------------------------------------------------------- */
int dgv2_emd_approxmatch(float* match, float* temp, const float* xyz1, const float* xyz2, int B, int n, int m,
                         void* stream);
int dgv2_emd_matchcost(float* cost, const float* match, const float* xyz1, const float* xyz2, int B, int n, int m,
                       void* stream);
int dgv2_emd_matchcost_grad(float* grad1, float* grad2, const float* match, const float* xyz1, const float* xyz2, int B,
                            int n, int m, void* stream);

/* ---- fp8 (OCP e4m3) operands for the decimating branch convs of the discriminator (BASELINE configs[4]) --------------
 * The two tensors of a ResidualBlock (gans/models/dusty_v2.py:325-345) that exist only as MFMA operands -- the blurred
 * activation in front of the 3x3 stride-2 conv2 and the blur-decimated block input in front of the 1x1 skip conv -- are
 * written as e4m3 (unit scale) by the FIR kernels that produce them, the two convs' weights as e4m3 with a per-tensor
 * power-of-two scale; v_mfma_f32_16x16x32_fp8_fp8 contracts them with fp32 accumulation.  The residual stream and the
 * activation outputs read by a backward pass stay bf16; the weight-gradient stream reads the saved e4m3 tensor through
 * dgv2_fp8_dequant.  The reference's reduced-precision switch: gans/models/dusty_v2.py:388-394 (fp16 autocast). */
/* y8 [B,H,W,C] (e4m3 bytes) = R x for a same-size separable resampling: dgv2_fir_same_mfma with the result rounded to
 * e4m3 (saturating at +-448) at the store.  replaces: Resample(up = down = 1) / Blur, gans/models/ops/common.py:105-135. */
int dgv2_fir_same_mfma_q8(void* y8, const void* x, const void* bands, int B, int C, int H, int W, void* stream);
/* dgv2_resample_tab (x bf16, tables as there) with the result stored as e4m3 bytes y8 [B,out_h,out_w,C]; C % 8 == 0,
 * Ew <= 4, Eh <= 64, else DGV2_ENOTSUP.  replaces: Resample, gans/models/ops/common.py:105-135. */
int dgv2_resample_tab_q8(void* y8, const void* x, const int* idx_h, const float* coef_h, const int* cnt_h, int Eh,
                         const int* idx_w, const float* coef_w, const int* cnt_w, int Ew, int B, int C, int in_h,
                         int in_w, int out_h, int out_w, void* stream);
/* n <= 16 conv weights in one launch pair: w8[l] [O,kk,C] = e4m3(src[l] [O,C,kk] fp32 * 2^k_l), 2^k_l the largest power of
 * two with max|src[l]| * 2^k_l <= 448; descale[l] (device, fp32) = eq[l] / 2^k_l -- the factor the conv epilogue puts on
 * its accumulator (eq = the EqualLR factor).  amax: n words of device scratch.  w8 / src / O / C / kk / eq: HOST arrays.
 * C % 8 == 0.  replaces: the `weight * scale` operand of ops.Conv2d, gans/models/ops/common.py:158-210. */
int dgv2_fp8_quant_weights(void* const* w8, const void* const* src, const int* O, const int* C, const int* kk,
                           const float* eq, int n, float* descale, void* amax, void* stream);
/* y (bf16) = x (e4m3) * scale for n elements, n % 16 == 0: the saved e4m3 activations as the bf16 operand of the
 * weight-gradient stream (dgv2_conv_wgrad_stream). */
int dgv2_fp8_dequant(void* y, const void* x, int64_t n, float scale, void* stream);
/* dgv2_conv_taps (single class, no extras, overwrite) on e4m3 operands x8 [B,Hin,Win,Cin], w8 [O,wtaps,Cin]:
 *   y (bf16) = act( acc * acc_scale[0] + resid + bias ) * scale,  acc_scale a DEVICE scalar (dgv2_fp8_quant_weights).
 * Cin % 64 == 0, O >= 64; DGV2_ENOTSUP otherwise.  replaces: ops.Conv2d forward (gans/models/ops/common.py:187-210) of
 * ResidualBlock.conv2 / .skip (gans/models/dusty_v2.py:331-345). */
int dgv2_conv_taps_fp8(void* y, const void* x8, const void* w8, const float* acc_scale, int B, int Hin, int Win, int Cin,
                       int Hg, int Wg, int O, int Hy, int Wy, int in_stride, int ioff_h, int ioff_w, int ntaps, int wtaps,
                       const int* taps_host, int ring, const float* bias, const void* resid, int act, float alpha,
                       float scale, void* stream);

/* Grouped small Linear layers in fp32 on the matrix cores (glin.hip): L <= 24 layers that share the batch B and the
 * input width K in ONE launch, the layers' parameter addresses as HOST arrays (they travel by value in the kernel
 * arguments).  y_l [B,N_l] = act( alpha * PN(x_l) W_l^T + beta * bias_l ): x_l rows of K floats at row stride lda[l]
 * (a style vector inside ws [B,S,K]: lda = S*K), W_l [N_l,K], bias_l [N_l] or NULL; act 0 | 1 = leaky ReLU(slope);
 * prenorm: rows normalised by rsqrt(mean_k x^2 + 1e-8) first, the factors stored in rnorm [B] when given.
 * Exact fp32 (v_mfma_f32_16x16x4_f32 = fmaf chains).  K % 64 == 0.
 * replaces: PixelNorm + EqualLR(nn.Linear) + LeakyReLU of MappingNetwork (gans/models/dusty_v2.py:13-29,
 * ops/common.py:158-184,213-223) and the style affine ModConv2d.mod of every modulated conv (ops/style.py:30,75). */
int dgv2_glin_fwd(float* const* y, const float* const* x, const float* const* w, const float* const* bias, const int* N,
                  const int* lda, int L, int B, int K, float alpha, float beta, int act, float slope, int prenorm,
                  float* rnorm, void* stream);
/* Its input gradient: dx [B,K] at row stride ldx (=|+= when accumulate) alpha * sum_l (g_l . act'(y_l)) W_l over the L
 * layers that read this input; g_l, y_l [B,N_l] contiguous (yact entries NULL: no activation), N_l % 32 == 0.  The
 * contraction over all layers' features runs in chunks of 256 (one block each) whose partials a second kernel folds in
 * fixed order: scratch fp32 [>= B * K * sum_l ceil(N_l / 256)]. */
int dgv2_glin_dinput(float* dx, int ldx, float* scratch, int64_t scratch_elems, const float* const* g,
                     const float* const* yact, const float* const* w, const int* N, int L, int B, int K, float alpha,
                     float slope, int accumulate, void* stream);
/* ... and the parameter gradients of all L layers in one launch: dW_l [N_l,K] = alpha * (g_l . act'(y_l))^T (x_l * rnorm),
 * dbias_l [N_l] = beta * column sums (dbias or entries NULL: skipped); x_l rows at stride ldx[l], rnorm [B] or NULL. */
int dgv2_glin_dweight(float* const* dw, float* const* dbias, const float* const* g, const float* const* yact,
                      const float* const* x, const int* N, const int* ldx, int L, int B, int K, float alpha, float beta,
                      float slope, const float* rnorm, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DGV2_H */
