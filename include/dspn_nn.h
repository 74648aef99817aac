/*
 * dspn_nn.h -- C ABI of the MI355X (gfx950) kernels behind DSPNet's conv-heavy
 * multi-task forward/backward.
 *
 * In the reference these are MXNet built-in operators (Convolution, BatchNorm,
 * Activation, Pooling, Deconvolution, BilinearSampler + GridGenerator,
 * SoftmaxOutput, smooth_l1 + MakeLoss, the SGD updater) instantiated by
 * symbol/resnet.py, symbol/common.py and symbol/multitask_symbol_builder.py;
 * each entry cites the call site whose arithmetic it provides.  MXNet itself is
 * not vendored by the reference, so the semantics followed are the documented
 * MXNet 0.11-1.0 ones, restated in oracle/dspnet_torch.py ("parity unpinned").
 *
 * Conventions
 *   - activations are NHWC float32, dense, with a physical channel count that is
 *     a multiple of 4 (logical channels are zero padded: 3->4, 19->20, ...);
 *   - convolution weights are [Cout][R][S][Cin] (Cin physical);
 *   - all tensor pointers are DEVICE pointers; the caller owns every buffer
 *     including workspaces; nothing allocates or synchronises;
 *   - `stream` is a hipStream_t passed as void*;
 *   - return 0 or a negative dspn_status (see dspn_multibox.h), message in
 *     dspn_last_error().
 */
#ifndef DSPN_NN_H_
#define DSPN_NN_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Convolution (mx.sym.Convolution: symbol/resnet.py:32-49,95, common.py:32,
 * 393-409, multitask_symbol_builder.py:543-583).  fp32 MFMA implicit GEMM. -------- */

/* y[n,ho,wo,k] = sum x[n,ho*stride-pad_h+r*dil, wo*stride-pad_w+s*dil, c] w[k,r,s,c] (+bias[k]) (relu).
 * R x S kernels with separate pads cover the inception 1x7 / 7x1 / 1x3 / 3x1 classes (symbol/inceptionv3.py).
 * y pixel stride y_ldc (0 = Cout), batch stride y_batch_stride (0 = dense).
 * accumulate != 0: y += result (before relu). bias may be NULL.  residual (may be NULL): a tensor laid out
 * like y that is added in the epilogue -- the `conv3 + shortcut` of a residual unit (symbol/resnet.py:51)
 * without a separate pass.
 * workspace (optional, may be NULL): scratch for split-K partial tiles, used when the output grid is
 * too small to fill the chip (SSD heads / extras); any size, dspn_conv2d_split_workspace_bytes() is
 * always enough.  Partials are summed in a fixed order (deterministic). */
size_t dspn_conv2d_split_workspace_bytes(long long out_pixels, int Cout);
/* Math mode of one convolution call -- the `math` argument of the *_bn_f32 / *_slabs_f32 entry points (the plain
 * *_f32 entries are always DSPN_MATH_FP32).  A per-call argument, not library state: the library keeps no globals.
 *   DSPN_MATH_FP32        fp32 MFMA (v_mfma_f32_32x32x2_f32)
 *   DSPN_MATH_BF16        bf16 MFMA (v_mfma_f32_32x32x16_bf16), fp32 accumulate: tensors stay fp32 in HBM and are ROUNDED
 *                         to bf16 (round-to-nearest-even, 2^-9 relative) on the way into LDS -- BASELINE.json configs[3]
 *                         "bf16 MFMA convs"
 *   DSPN_MATH_F32_BF16X3  fp32 RESULTS on the bf16 MFMA: each float operand is cut into three bf16 pieces on the way into
 *                         LDS (p0 = bf16(x), p1 = bf16(x - p0), p2 = bf16(x - p0 - p1)) and a product is the sum of the six
 *                         exact partial products x_p * w_q, p + q <= 2, accumulated in fp32.  Error against float64 equal
 *                         to DSPN_MATH_FP32's (tests/test_nn_gpu.py), ~1.3x its speed; dspnet_amd's default in round 2.
 *                         WEIGHT OPERANDS in this mode: a forward / data-gradient call whose contraction runs over a multiple
 *                         of 32 channels per tap (Cin % 32 == 0, resp. ldy % 32 == 0) reads the weights as PIECE PLANES
 *                         (`w_planes` / `wt_planes`, made by dspn_conv2d_weight_planes_f32 once per weight update) instead
 *                         of cutting the float weights again in every tile of every call; the float operand may then be
 *                         NULL.  Other channel counts (3 -> 4, 20, 36 ...) read the float operand and ignore the planes.
 *                         Non-finite operands: +-inf and |x| > 3.39e38 give NaN / inf pieces (x - bf16(x) is inf - inf).
 *   DSPN_MATH_F32_F16X2   fp32 RESULTS on the fp16 MFMA (v_mfma_f32_32x32x16_f16) with HALF the matrix work of the mode above:
 *                         each float operand x, multiplied by a power of two s chosen per TENSOR, is cut into two fp16
 *                         pieces (h0 = fp16(s x), h1 = fp16(s x - h0): 11 + 11 bits and a sign, equal to s x within 2^-24)
 *                         and a product is h0 g0 + h0 g1 + h1 g0 (three exact partial products, fp32 accumulate; the
 *                         dropped h1 g1 is below 2^-24 |x w|); the accumulators are multiplied by 1 / (s_x s_w) -- exact --
 *                         in the epilogue.  Error against float64: the fp32 MFMA's (rms 1.15x, largest error lower;
 *                         tests/test_nn_gpu.py); ~1.45x its speed; what dspnet_amd passes by default (round 3).  WEIGHT
 *                         OPERANDS as in the mode above (piece planes with pieces = 2, cut relative to `w_absmax`).  What it needs that the bf16 split does not is RANGE (fp16: 2^-24 .. 65504):
 *                         the caller passes, per operand, the operand's largest magnitude in device memory
 *                         (`*_absmax` arguments: DSPN_ABSMAX_SLOTS partial maxima; dspn_absmax_f32 / dspn_absmax_batch_f32
 *                         compute them, for a folded input affine of the tensor AFTER the affine); the kernel scales that
 *                         maximum into [2^14, 2^15).
 *                         Elements more than 2^17 below the tensor's maximum keep an absolute error of 2^-39 of that
 *                         maximum instead of a relative one.  A NULL magnitude means scale 1 and needs DSPN_MATH_UNSCALED_OK
 *                         (below); one that UNDERSTATES the maximum by more than 2x overflows to inf.
 *                         Non-finite operands (round 4): NaN elements stay NaN and never enter a magnitude block; a
 *                         block with an infinite partial maximum marks a tensor that holds +-inf: the scale then comes from
 *                         the finite partial maxima, so every output no non-finite element touches keeps its accuracy.
 *                         An infinite WEIGHT (piece planes) and infinite operands of the WEIGHT GRADIENT get the pieces
 *                         (+-65504, +-inf): every output they touch is +-inf with the sign of x w, or NaN where the other
 *                         factor is 0 or two infinite terms differ in sign -- the fp32 results.  An infinite ACTIVATION of
 *                         the forward pass / infinite output gradient of the data gradient gives NaN in every output it
 *                         touches (h1 = inf - inf): non-finite exactly where fp32 is non-finite, but NaN instead of the
 *                         signed infinity (repairing it inside those kernels' k-loop cost 4 % of the training step).
 * The *_bf16 entry points (bf16 tensors in HBM) ignore the argument: their operands are bf16 already. */
#define DSPN_MATH_FP32 0
#define DSPN_MATH_BF16 1
#define DSPN_MATH_F32_BF16X3 2
#define DSPN_MATH_F32_F16X2 3
/* OR-ed into `math`: with DSPN_MATH_F32_F16X2, a NULL magnitude block means "scale 1" and is only accepted with this flag --
 * the caller vouches that every |operand| is below 65504 (round 4; without the flag a NULL block is DSPN_ERR_ARG: a silent
 * fp16 overflow otherwise).  The C default for code that keeps no magnitudes is DSPN_MATH_FP32; dspnet_amd (Python) passes
 * DSPN_MATH_F32_F16X2 with both blocks on every float-tensor call. */
#define DSPN_MATH_UNSCALED_OK 0x100
/* OR-ed into `math` of dspn_conv2d_dgrad_bn_f32 / dspn_conv2d_wgrad_bn_f32 / dspn_conv2d_wgrad_slabs_f32 (with
 * DSPN_MATH_F32_F16X2; round 4): dy is not a float tensor but the fp16 PIECE PLANES dspn_bn_backward_from_sums_f32 wrote
 * (dx_planes), cut with the scale of the block passed as dy_absmax.  Needs ldy % 32 == 0 (and ldy == Cout for the weight
 * gradient). */
#define DSPN_MATH_DY_PLANES 0x200
/* OR-ed into `math` of dspn_conv2d_forward_bn_f32 / dspn_conv2d_wgrad_bn_f32 / dspn_conv2d_wgrad_slabs_f32 (with
 * DSPN_MATH_F32_F16X2; round 4): x is the piece planes dspn_bn_apply_planes_f32 wrote, cut with the scale of the block
 * passed as x_absmax.  Needs Cin % 32 == 0 and no input affine (in_scale == NULL: the planes hold the affine's result). */
#define DSPN_MATH_X_PLANES 0x400

int dspn_conv2d_forward_f32(const float *x, const float *w, const float *bias, const float *residual, float *y,
                            int N, int H, int W, int Cin, int Cout, int R, int S,
                            int stride, int pad_h, int pad_w, int dil, int Ho, int Wo,
                            long long y_batch_stride, int y_ldc, int relu, int accumulate,
                            void *workspace, size_t workspace_bytes, void *stream);

/* Same convolution on u = x * in_scale[c] + in_shift[c] (then max(u, 0) if in_relu), with u == 0 outside
 * the image: the BatchNorm(+ReLU) that precedes every convolution of a pre-activation residual unit
 * (symbol/resnet.py:30-45: bn -> relu -> conv) folded into the convolution's tile loader, so the normalised
 * tensor is never written to or re-read from HBM.  in_scale / in_shift: Cin floats each, e.g. the scale / shift
 * outputs of dspn_bn_stats_f32; both NULL = plain convolution. */
int dspn_conv2d_forward_bn_f32(const float *x, const float *in_scale, const float *in_shift, int in_relu,
                               const float *w, const void *w_planes, const float *bias, const float *residual, float *y,
                               int N, int H, int W, int Cin, int Cout, int R, int S,
                               int stride, int pad_h, int pad_w, int dil, int Ho, int Wo,
                               long long y_batch_stride, int y_ldc, int relu, int accumulate,
                               float *out_stats, size_t out_stats_bytes, float *out_minmax, int math,
                               const float *x_absmax, const float *w_absmax,
                               void *workspace, size_t workspace_bytes, void *stream);
/* out_minmax WITHOUT out_stats (round 5, DSPN_MATH_F32_F16X2, float tensors, dense output with ldc % 4 == 0): the pointer is a
 * DSPN_ABSMAX_SLOTS magnitude block that receives, by maximum, the partial maxima of |y| as stored -- dspn_absmax_f32(y) without its
 * pass over y for the convolutions no BatchNorm reads (the wide family's epilogue takes it along; other kernels are followed by
 * that pass inside the call).
 * out_minmax (optional; written in DSPN_MATH_F32_F16X2 together with out_stats, SAME size and tiling -- the buffer must hold
 * out_stats_bytes bytes, the one size argument covers both tables): per row tile and
 * channel the smallest [(t*2 + 0)*Cout + c] and largest [(t*2 + 1)*Cout + c] stored value.  A BatchNorm(+ReLU) of y is
 * monotone per channel, so the magnitude of what the next convolution multiplies is dspn_absmax_f32 over THIS table
 * (tiles*2 rows of Cout values) with that BatchNorm's scale / shift -- a few KB instead of a pass over y.
 * x_absmax / w_absmax: DSPN_MATH_F32_F16X2 only (ignored otherwise): magnitudes (DSPN_ABSMAX_SLOTS floats each, see
 * dspn_absmax_f32) of the input AFTER its affine and of the weights.
 * out_stats (optional): BatchNorm statistics of y gathered in the convolution's epilogue, one (mean, M2) pair per
 * channel and row tile: out_stats[(t*2 + 0)*Cout + c] = mean of tile t's rows, [(t*2 + 1)*Cout + c] = their sum of
 * squared deviations.  dspn_conv2d_stats_layout() gives the number of tiles (0: not available, Cout % 4 != 0) and
 * the rows per tile for an output of out_pixels x Cout; dspn_bn_stats_from_tiles_f32 merges them.  Requires a dense
 * output (y_ldc == Cout or 0). */
int dspn_conv2d_stats_layout(long long out_pixels, int Cout, int *tile_rows);

/* Piece planes of a weight operand for the split math modes, from the float master w [Cout][taps][Cin] (Cin % 4 == 0):
 *   planes   (optional, needs Cin % 32 == 0): of w itself, [Cout][taps][Cin / 32][piece][32]
 *                                             -> `w_planes` of dspn_conv2d_forward_bn_f32,
 *   planes_t (optional, cols_t % 32 == 0, cols_t >= Cout): of w^T zero padded, [Cin][taps][cols_t / 32][piece][32]
 *                                             -> `wt_planes` of dspn_conv2d_dgrad_bn_f32 (cols_t = ldy),
 * 16-bit elements, rows * taps * cols * 2 * pieces bytes each.  Either pointer may be NULL.  One read of w serves both.
 *   pieces = 3 (DSPN_MATH_F32_BF16X3): bfloat16 p0 = bf16(x), p1 = bf16(x - p0), p2 = bf16(x - p0 - p1), round to nearest even;
 *   pieces = 2 (DSPN_MATH_F32_F16X2):  float16 p0 = f16(s x), p1 = f16(s x - p0) with s the power of two the convolution
 *              kernels derive from `w_absmax` (the weight's magnitude block, dspn_absmax_f32 of w -- filled BEFORE this call
 *              on the same stream, and the block later passed as the convolution's `w_absmax`).
 * Batch form (every weight of a training step in one launch): table of n 64-byte rows in DEVICE memory
 * { const float *w; void *planes; void *planes_t; int32 Cout, taps, Cin, cols_t; int64 begin; const float *w_absmax;
 *   int32 pieces, 0 } with begin = the sum of dspn_conv2d_weight_planes_tiles() over the preceding rows; total_tiles = that
 * sum over all rows. */
int dspn_conv2d_weight_planes_f32(const float *w, void *planes, void *planes_t, int Cout, int taps, int Cin, int cols_t,
                                  int pieces, const float *w_absmax, void *stream);
long long dspn_conv2d_weight_planes_tiles(int Cout, int taps, int Cin, int cols_t, int with_transposed);
int dspn_conv2d_weight_planes_batch_f32(const void *table, int n, long long total_tiles, void *stream);

/* wt[c][tap][k] = w[k][tap][c], k padded with zeros to Cout_pad (operand of dgrad). */
int dspn_conv2d_weight_transpose_f32(const float *w, float *wt, int Cout, int taps, int Cin,
                                     int Cout_pad, void *stream);
/* every weight transpose of a training step in one launch (one workgroup per 32 x 32 tile of a tap).  table: n rows of 48
 * bytes in DEVICE memory, { const float *w; float *wt; int32 Cout, taps, Cin, Cout_pad; int64 begin; void *reserved (NULL) }
 * where begin = the sum of dspn_conv2d_weight_transpose_tiles() over the preceding rows (rows sorted by begin);
 * total_tiles = that sum over all rows.  Cin % 4 == 0 and Cout_pad % 4 == 0. */
long long dspn_conv2d_weight_transpose_tiles(int Cout, int taps, int Cin, int Cout_pad);
int dspn_conv2d_weight_transpose_batch_f32(const void *table, int n, long long total_tiles, void *stream);

/* Data gradient of the convolution above: dx (N,H,W,dx_ldc) from dy (N,Ho,Wo,ldy) and the
 * transposed weights wt [Cin][R*S][ldy].  stride 1 (any dilation) or stride 2 (dilation 1; runs
 * as 4 output-parity classes).  accumulate != 0: dx += result.
 * The same entry is the FORWARD of mx.sym.Deconvolution (multitask_symbol_builder.py:586:
 * 4x4, stride 2, pad 1, no bias) with x := dy. */
int dspn_conv2d_dgrad_f32(const float *dy, const float *wt, float *dx,
                          int N, int H, int W, int Cin, int ldy, int R, int S,
                          int stride, int pad_h, int pad_w, int dil, int Ho, int Wo, int dx_ldc,
                          int accumulate, void *workspace, size_t workspace_bytes, void *stream);
/* The same data gradient, with the two reductions of the BatchNorm(+ReLU) backward that consumes dx gathered in the
 * epilogue (dx must be the COMPLETE gradient of the BatchNorm output after this call, i.e. this is its last writer):
 * bn_x = the BatchNorm's input (laid out like dx), bn_mean / bn_rstd (and bn_scale / bn_shift for the ReLU mask) its
 * forward statistics.  bn_sums[(t*2 + 0)*Cin + c] = sum over row tile t of dy', [(t*2 + 1)*Cin + c] = sum of dy'*xhat,
 * t < dspn_conv2d_dgrad_bn_tiles(); dspn_bn_backward_from_sums_f32 finishes the backward pass from them. */
int dspn_conv2d_dgrad_bn_tiles(int N, int H, int W, int Cin, int stride);
int dspn_conv2d_dgrad_bn_f32(const float *dy, const float *wt, const void *wt_planes, float *dx, int N, int H, int W, int Cin, int ldy,
                             int R, int S, int stride, int pad_h, int pad_w, int dil, int Ho, int Wo, int dx_ldc,
                             int accumulate, const float *bn_x, const float *bn_scale, const float *bn_shift,
                             const float *bn_mean, const float *bn_rstd, int bn_relu, float *bn_sums,
                             size_t bn_sums_bytes, float *bn_dy_absmax, int math, const float *dy_absmax, const float *w_absmax,
                             void *workspace, size_t workspace_bytes, void *stream);
/* bn_dy_absmax (optional, with bn_sums, DSPN_MATH_F32_F16X2; round 4): DSPN_ABSMAX_SLOTS floats (zeroed by the caller) that
 * receive the partial maxima of |dx| as stored -- the `dy_absmax` of the dspn_bn_backward_from_sums_f32 call that finishes
 * this BatchNorm's backward pass with dx_planes. */

/* out[c] = sum over every input pixel of the data gradient of the convolution, c < Cin <= 8, computed
 * from per-tap sums of dy without forming the gradient (the first convolution's input only feeds the
 * beta of the fix_gamma BatchNorm on the image, symbol/resnet.py:91).  w is [Cout][R][S][Cin]. */
size_t dspn_conv2d_input_sum_grad_workspace_bytes(int Ho, int Wo, int ldy, int R, int S);
int dspn_conv2d_input_sum_grad_f32(const float *dy, const float *w, float *out, int N, int H, int W,
                                   int Cin, int Cout, int ldy, int R, int S, int stride, int pad_h, int pad_w, int dil,
                                   int Ho, int Wo, void *workspace, size_t workspace_bytes, void *stream);

size_t dspn_conv2d_wgrad_workspace_bytes(int N, int Ho, int Wo, int Cin, int Cout, int R, int S);

/* dw[k][r][s][c] (+)= sum_{n,ho,wo} dy[n,ho,wo,k] x[n,ho*stride-pad+r*dil, wo*stride-pad+s*dil, c].
 * Split-K over pixels into `workspace` slabs, summed in a fixed order (deterministic). */
int dspn_conv2d_wgrad_f32(const float *x, const float *dy, float *dw,
                          int N, int H, int W, int Cin, int Cout, int ldy, int R, int S,
                          int stride, int pad_h, int pad_w, int dil, int Ho, int Wo, int accumulate,
                          void *workspace, size_t workspace_bytes, void *stream);
/* weight gradient with respect to the same u = (relu)(x * in_scale + in_shift) as dspn_conv2d_forward_bn_f32 */
int dspn_conv2d_wgrad_bn_f32(const float *x, const float *in_scale, const float *in_shift, int in_relu,
                             const float *dy, float *dw, int N, int H, int W, int Cin, int Cout, int ldy,
                             int R, int S, int stride, int pad_h, int pad_w, int dil, int Ho, int Wo,
                             int accumulate, int math, const float *x_absmax, const float *dy_absmax,
                             void *workspace, size_t workspace_bytes, void *stream);

/* The two halves of the weight gradient separately, so that a training step can run ONE slab reduction for many
 * layers (82 launches of a few microseconds each otherwise): dspn_conv2d_wgrad_slabs_f32 leaves the split-K partial
 * sums [dspn_conv2d_wgrad_splits()][Cout][R*S*Cin] in `slabs`; dspn_conv2d_slab_reduce_batch_f32 sums the slabs of
 * every row of a descriptor table (DEVICE memory, 40-byte rows { const float *slabs; float *dw; int64 n4 =
 * Cout*R*S*Cin/4; int32 splits, accumulate; int64 begin = sum of n4 over the preceding rows }) in a fixed order. */
int dspn_conv2d_wgrad_splits(int N, int Ho, int Wo, int Cin, int Cout, int R, int S, int stride);
int dspn_conv2d_wgrad_slabs_f32(const float *x, const float *in_scale, const float *in_shift, int in_relu,
                                const float *dy, float *slabs, size_t slabs_bytes, int N, int H, int W, int Cin,
                                int Cout, int ldy, int R, int S, int stride, int pad_h, int pad_w, int dil, int Ho,
                                int Wo, int math, const float *x_absmax, const float *dy_absmax, void *stream);
int dspn_conv2d_slab_reduce_batch_f32(const void *table, int n, long long total4, void *stream);

/* Operand magnitudes for DSPN_MATH_F32_F16X2.  A magnitude is DSPN_ABSMAX_SLOTS = 64 floats in device memory whose maximum is
 * the largest |u| of the operand (64 partial maxima instead of one word: thousands of atomics on one address serialise);
 * every `*_absmax` argument of the convolution entries points at such a block.  dspn_absmax_f32: out_dev[0..63] =
 * max(out_dev[..], partial maxima of |u|) over a (rows, C) float tensor, u = x or (relu)(x * scale[c] + shift[c]) when scale /
 * shift are given (the operand a convolution with a folded BatchNorm multiplies; the same fmaf as the loaders).  The caller
 * zeroes the block (once per step is enough for a whole array of them); integer atomic max on the floats' bits:
 * deterministic.  C % 4 == 0.
 * Batch form (every weight of a graph in one launch): table of n 32-byte rows in DEVICE memory { const float *x; float *out
 * (64 floats); int64 n4 (= elements / 4); int64 begin (= sum of ceil(n4 / 1024) over the preceding rows) }; total_chunks =
 * that sum. */
#define DSPN_ABSMAX_SLOTS 64
int dspn_absmax_f32(const float *x, long long rows, int C, const float *scale, const float *shift, int relu,
                    float *out_dev, void *stream);
/* Range guard of DSPN_MATH_F32_F16X2 (round 5).  dspn_absmin_rows_batch_f32: for every row of the table -- { const float *w;
 * float *out_min; int32 rows, row_len; int64 begin (= sum of `rows` over the preceding table rows) }, 32 bytes, DEVICE memory
 * -- the smallest non-zero per-output-channel magnitude of w [rows][row_len] into *out_min (preset to +inf by the caller):
 * beside the weight's magnitude block this is the span of its channels.  One launch for every weight of a graph.
 * dspn_tile_minmax_f32: minmax[(t*2 + 0)*C + c] / [(t*2 + 1)*C + c] = smallest / largest x[r][c] over the rows of tile t
 * (tile_rows rows each) -- the table a dspn_conv2d_forward_bn_f32 call writes as out_minmax, for an output that was produced
 * by a call in another math (the guard's fallback), so that the consumers' magnitudes still come from the table. */
int dspn_absmin_rows_batch_f32(const void *table, int n, long long total_rows, void *stream);
int dspn_tile_minmax_f32(const float *x, long long rows, int C, int tile_rows, float *minmax, void *stream);
int dspn_absmax_batch_f32(const void *table, int n, long long total_chunks, void *stream);
/* a BOUND instead of the magnitude (round 4): out_dev (+max)= max over c of |scale[c]| * M + |shift[c]|, M = the magnitude
 * in x_absmax_dev -- an upper bound of |(relu)(x * scale[c] + shift[c])| for a convolution that folds a BatchNorm into its
 * loader and whose raw input has a known magnitude but no per-channel extremes (a pooled tensor).  A magnitude that is too
 * large by less than 2^17 costs the two-piece math nothing (see DSPN_MATH_F32_F16X2).  One tiny launch instead of a pass. */
int dspn_absmax_affine_bound_f32(const float *scale, const float *shift, int C, const float *x_absmax_dev, float *out_dev,
                                 void *stream);

/* ---- bfloat16 TENSORS in HBM: the `*_bf16` twins (BASELINE.json configs[3] "bf16 MFMA convs" with the operands stored
 * as they are multiplied).  Same arguments and semantics as the `*_f32` entry of the same name, with every ACTIVATION
 * pointer (x, y, dy, dx, residual, bn_x) and the convolution weight operands (w, wt) pointing at bfloat16 elements
 * (dspn_bf16 = the raw 16 bits); physical channel counts are multiples of 8 for convolution operands (16-byte chunks), of
 * 4 elsewhere.  Arithmetic is fp32 throughout: elements are widened on load, results rounded to nearest even on store;
 * BatchNorm statistics gathered in a convolution epilogue are those of the ROUNDED values.  Parameters, per-channel
 * vectors (scale / shift / bias / statistics), split-K slabs, weight GRADIENTS and workspaces stay float.  The float
 * master weights are turned into the bf16 operands by dspn_conv2d_weight_prepare_bf16 (one batched launch per step).
 * `math` is accepted for symmetry and ignored (always bf16 MFMA, fp32 accumulate). */
#ifdef DSPN_BF16_ELEMENT            /* inside the library: the compiler's own bfloat16 type (same 16 bits) */
typedef DSPN_BF16_ELEMENT dspn_bf16;
#else
typedef unsigned short dspn_bf16;
#endif
int dspn_conv2d_forward_bn_bf16(const dspn_bf16 *x, const float *in_scale, const float *in_shift, int in_relu,
                                const dspn_bf16 *w, const void *w_planes_unused, const float *bias, const dspn_bf16 *residual, dspn_bf16 *y,
                                int N, int H, int W, int Cin, int Cout, int R, int S,
                                int stride, int pad_h, int pad_w, int dil, int Ho, int Wo,
                                long long y_batch_stride, int y_ldc, int relu, int accumulate,
                                float *out_stats, size_t out_stats_bytes, float *out_minmax_unused, int math,
                                const float *absmax_unused_a, const float *absmax_unused_b,
                                void *workspace, size_t workspace_bytes, void *stream);
int dspn_conv2d_dgrad_bn_bf16(const dspn_bf16 *dy, const dspn_bf16 *wt, const void *wt_planes_unused, dspn_bf16 *dx, int N, int H, int W, int Cin,
                              int ldy, int R, int S, int stride, int pad_h, int pad_w, int dil, int Ho, int Wo,
                              int dx_ldc, int accumulate, const dspn_bf16 *bn_x, const float *bn_scale,
                              const float *bn_shift, const float *bn_mean, const float *bn_rstd, int bn_relu,
                              float *bn_sums, size_t bn_sums_bytes, float *bn_dy_absmax_unused, int math,
                              const float *absmax_unused_a, const float *absmax_unused_b, void *workspace,
                              size_t workspace_bytes, void *stream);
int dspn_conv2d_wgrad_bn_bf16(const dspn_bf16 *x, const float *in_scale, const float *in_shift, int in_relu,
                              const dspn_bf16 *dy, float *dw, int N, int H, int W, int Cin, int Cout, int ldy,
                              int R, int S, int stride, int pad_h, int pad_w, int dil, int Ho, int Wo,
                              int accumulate, int math, const float *absmax_unused_a, const float *absmax_unused_b,
                              void *workspace, size_t workspace_bytes, void *stream);
int dspn_conv2d_wgrad_slabs_bf16(const dspn_bf16 *x, const float *in_scale, const float *in_shift, int in_relu,
                                 const dspn_bf16 *dy, float *slabs, size_t slabs_bytes, int N, int H, int W, int Cin,
                                 int Cout, int ldy, int R, int S, int stride, int pad_h, int pad_w, int dil, int Ho,
                                 int Wo, int math, const float *absmax_unused_a, const float *absmax_unused_b, void *stream);
int dspn_conv2d_input_sum_grad_bf16(const dspn_bf16 *dy, const float *w, float *out, int N, int H, int W,
                                    int Cin, int Cout, int ldy, int R, int S, int stride, int pad_h, int pad_w, int dil,
                                    int Ho, int Wo, void *workspace, size_t workspace_bytes, void *stream);
/* bf16 operands of a float master weight w [Cout][taps][Cin]: wt [Cin][taps][Cout_pad] (zero padded, Cout_pad % 8 == 0;
 * the data-gradient operand) and, when wh != NULL, the copy wh [Cout][taps][Cin] (the forward operand).  Batch form:
 * table of n 48-byte rows in DEVICE memory { const float *w; dspn_bf16 *wt; int32 Cout, taps, Cin, Cout_pad; int64 begin;
 * dspn_bf16 *wh } with begin = the sum of dspn_conv2d_weight_transpose_tiles() over the preceding rows, total_tiles = over
 * all rows.  (dspn_conv2d_weight_transpose_batch_f32 reads the same 48-byte rows with wh = NULL.) */
int dspn_conv2d_weight_prepare_bf16(const float *w, dspn_bf16 *wh, dspn_bf16 *wt, int Cout, int taps, int Cin,
                                    int Cout_pad, void *stream);
int dspn_conv2d_weight_prepare_batch_bf16(const void *table, int n, long long total_tiles, void *stream);

/* ---- BatchNorm with batch statistics (+ fused ReLU) (mx.sym.BatchNorm eps=2e-5:
 * symbol/resnet.py:30-41,91,96; multitask_symbol_builder.py:545-585) ------------------ */

size_t dspn_bn_workspace_bytes(long long rows, int C);

/* mean[c], rstd[c] = 1/sqrt(biased var + eps) over `rows` = N*H*W rows of x (rows, C), plus the folded
 * affine of the apply pass: scale = gamma*rstd, shift = beta - mean*scale.  gamma == NULL means
 * fix_gamma (gamma == 1). */
int dspn_bn_stats_f32(const float *x, long long rows, int C, float eps, const float *gamma,
                      const float *beta, float *mean, float *rstd, float *scale, float *shift,
                      void *workspace, size_t workspace_bytes, void *stream);

/* y = x*scale + shift, optionally max(.,0). */
/* the same outputs as dspn_bn_stats_f32 from the per-tile (mean, M2) pairs a convolution wrote (out_stats of
 * dspn_conv2d_forward_bn_f32): tiles of tile_rows rows (the last one shorter) over `rows` rows, merged pairwise in
 * double (Chan et al.), fixed order */
int dspn_bn_stats_from_tiles_f32(const float *tile_stats, int tiles, int tile_rows, long long rows, int C, float eps,
                                 const float *gamma, const float *beta, float *mean, float *rstd, float *scale,
                                 float *shift, const float *tile_minmax, int relu, float *out_absmax, float *out_absmin,
                                 float *out_chan_minmax, void *workspace, size_t workspace_bytes, void *stream);
/* tile_minmax + out_absmax (both or neither; DSPN_MATH_F32_F16X2): the (min, max) pairs the same convolution wrote beside
 * its statistics (out_minmax).  The largest |(relu)(x * scale + shift)| over the tensor -- the magnitude of what the next
 * convolution multiplies when it folds this BatchNorm into its loader -- is then max-ed INTO the magnitude block
 * out_absmax (DSPN_ABSMAX_SLOTS floats, zeroed by the caller at the start of the step), with no pass over the tensor.
 * out_absmin (optional, with tile_minmax; round 4): ONE float that receives, by atomic minimum on its bit pattern (the caller
 * sets it to +inf first), the smallest non-zero PER-CHANNEL magnitude of the same tensor -- a range monitor: a tensor whose
 * channel magnitudes span more than 2^17 leaves the two-piece math's window of relative accuracy for its small channels
 * (DSPN_MATH_F32_F16X2 above), and the ratio of the two blocks says so without a pass over the tensor.
 * out_chan_minmax (optional, with tile_minmax; round 4): 2 x C floats, the smallest [c] and largest [C + c] value of channel c
 * over the whole tensor -- what dspn_bn_backward_from_sums_f32 needs to bound the dx it writes as piece planes. */
/* optional scratch for long tile tables (>= 1024 tiles are first merged in groups of 32 by many workgroups);
 * dspn_bn_backward_from_sums_f32 uses 3*C floats + this many bytes the same way */
size_t dspn_bn_tiles_workspace_bytes(int tiles, int C);
/* out_absmax (optional, float tensors): DSPN_ABSMAX_SLOTS floats that receive the partial maxima of |y| as stored --
 * dspn_absmax_f32(y) without its pass over y (round 4: the magnitude of a MATERIALISED BatchNorm output for the
 * convolutions that multiply it in DSPN_MATH_F32_F16X2).  The caller zeroes it, as for dspn_absmax_f32. */
int dspn_bn_apply_f32(const float *x, const float *scale, const float *shift, float *y, long long rows,
                      int C, int relu, float *out_absmax, void *stream);
/* (round 4) the same values written as fp16 PIECE PLANES for DSPN_MATH_F32_F16X2 -- [pixel][C / 32][piece][32] halves, the
 * bytes and byte offsets of the float tensor (rows * C * 4 bytes) -- cut with the scale of the block y_absmax, which the
 * caller has BEFORE this pass: dspn_bn_stats_from_tiles_f32 forms it from the producer's per-channel extremes (out_absmax).
 * A multi-tap convolution behind the BatchNorm then reads the planes with DSPN_MATH_X_PLANES and copies them into LDS
 * instead of applying the affine and cutting each element once per (tap, column tile).  C % 32 == 0; not in place. */
int dspn_bn_apply_planes_f32(const float *x, const float *scale, const float *shift, void *y_planes, long long rows, int C,
                             int relu, const float *y_absmax, void *stream);

/* Backward of the fused op.  If relu != 0, dy is first masked with (x*scale + shift > 0), i.e. the
 * forward output's sign recomputed from x (the forward output itself is not read).
 * dx (+)= gamma*rstd*(dy - mean(dy) - xhat*mean(dy*xhat)); dgamma = sum dy*xhat; dbeta = sum dy.
 * dgamma may be NULL (fix_gamma).  accumulate != 0: dx += .
 * dx_absmax (optional, float tensors): DSPN_ABSMAX_SLOTS floats that receive the partial maxima of |dx| AS STORED (after
 * the accumulation), i.e. dspn_absmax_f32(dx) without its pass over dx: the magnitude the convolution that produced x
 * needs for its data / weight gradient in DSPN_MATH_F32_F16X2.  The caller zeroes it, as for dspn_absmax_f32. */
int dspn_bn_backward_f32(const float *x, const float *scale, const float *shift, const float *dy,
                         const float *mean, const float *rstd, const float *gamma, float *dx,
                         float *dgamma, float *dbeta, long long rows, int C, int relu, int accumulate,
                         float *dx_absmax, void *workspace, size_t workspace_bytes, void *stream);
int dspn_bn_backward_from_sums_f32(const float *x, const float *scale, const float *shift, const float *dy,
                                   const float *mean, const float *rstd, const float *gamma, const float *tile_sums,
                                   int tiles, float *dx, float *dgamma, float *dbeta, long long rows, int C, int relu,
                                   int accumulate, float *dx_absmax, float *dx_absmin, const float *dy_absmax, const float *x_chan_minmax,
                                   int dx_planes, void *workspace, size_t workspace_bytes, void *stream);
/* dx_planes is a flag word (round 6): bit 0 = dx as piece planes (below); | 2 = the FINALIZE alone (per-channel coefficients
 * into the first 3 * C floats of the workspace, dgamma / dbeta, the bound of dx); | 4 = the APPLY pass alone, from the
 * coefficients an earlier call with | 2 (same arguments, same workspace) left there.  The finalize is a chain of one or two
 * latency-bound launches on 1 - 64 workgroups: a caller runs it on a second stream beside the layer's weight gradient
 * (dspnet_amd/engine.py, Conv.backward) and the apply behind both.  0 / 1: both, as before. */
/* dx_absmin (optional, with dx_planes; round 5): ONE float, preset to +inf by the caller, that receives the smallest non-zero
 * per-channel bound of |dx| -- with dx_absmax, the span of channel magnitudes the planes are cut over (the range guard of
 * DSPN_MATH_F32_F16X2 reads it; elements more than 2^17 below the tensor's largest magnitude lose relative accuracy). */
/* dx_planes != 0 (round 4; float tensors, C % 32 == 0, accumulate == 0): dx is written as fp16 PIECE PLANES for
 * DSPN_MATH_F32_F16X2 instead of floats -- [row][C / 32][piece][32] 16-bit elements, the same 4 bytes per element and the
 * same byte offset for every group of four channels as the float tensor, (p0, p1) = (f16(s dx), f16(s dx - p0)) -- so that
 * the data gradient and the weight gradient of the convolution that produced x (DSPN_MATH_DY_PLANES) copy it into LDS
 * without cutting every element once per tap and column tile.  The scale s has to be known BEFORE the pass that forms dx:
 * it is the power of two of a BOUND, max over channels of |a| D + max(|c1 lo + c0|, |c1 hi + c0|) with dx = a dy' + c1 x + c0
 * per channel, D = the magnitude in dy_absmax (of dy: e.g. `bn_dy_absmax` of the data gradient that produced it) and
 * [lo, hi] = the channel's extremes of x in x_chan_minmax (2 x C floats: dspn_bn_stats_from_tiles_f32's out_chan_minmax).
 * dx_absmax (zeroed by the caller) RECEIVES that bound and is the block the consuming convolutions are given as their dy
 * magnitude -- a few times the true maximum at most, which the two-piece math does not feel (2^17). */

/* ---- element-wise / layout --------------------------------------------------------------- */
int dspn_add_f32(const float *a, const float *b, float *out, long long n, void *stream);      /* out = a + b */
int dspn_relu_backward_f32(const float *y, const float *dy, float *dx, long long n, int accumulate, void *stream);
/* ReLU backward and the bias gradient of the convolution whose epilogue applied the ReLU, in one pass:
 * dx = (y > 0) ? dy : 0 (dx may alias dy), out[c] = sum_rows dx[:, c] for c < C; rows of ld floats (ld % 4 == 0);
 * workspace: dspn_colsum_workspace_bytes(rows, C) */
int dspn_relu_backward_colsum_f32(const float *y, const float *dy, float *dx, long long rows, int C, int ld,
                                  float *out, float *dx_absmax /* optional: the magnitude block of dx as stored (round 5) */,
                                  void *workspace, size_t workspace_bytes, void *stream);
int dspn_fill_f32(float *p, float v, long long n, void *stream);
/* per-column sum of a (rows, ld) matrix over its first C columns: out[c] = sum_r a[r, c] (bias grads) */
size_t dspn_colsum_workspace_bytes(long long rows, int C);
int dspn_colsum_f32(const float *a, long long rows, int C, int ld, float *out,
                    void *workspace, size_t workspace_bytes, void *stream);
/* (N,C,H,W) -> (N,H,W,Cp) with zero padded channels, and back (Cp -> first C channels) */
int dspn_nchw_to_nhwc_f32(const float *src, float *dst, int N, int C, int H, int W, int Cp, void *stream);
int dspn_nhwc_to_nchw_f32(const float *src, float *dst, int N, int C, int H, int W, int Cp, void *stream);
/* copy a (rows, C) block between matrices with different row strides / column offsets:
 * dst[r*ldd + doff + c] (+)= src[r*lds + soff + c]; rows are grouped per sample:
 * row r belongs to sample r / rows_per_sample and the sample strides are given explicitly. */
int dspn_copy_block_f32(const float *src, float *dst, int samples, long long rows_per_sample, int C,
                        long long src_sample_stride, int lds, int soff,
                        long long dst_sample_stride, int ldd, int doff, int accumulate, void *stream);
/* (round 4) n block copies in one launch -- the SSD head packing moves six small maps per pass.  `table`: n rows of 72 bytes in
 * device memory, { const float *src; float *dst; int64 rows_per_sample, src_sample_stride, dst_sample_stride;
 * int32 C, lds, soff, ldd, doff, accumulate; int64 begin } with the meaning of dspn_copy_block_f32's arguments; begin = the
 * number of elements (samples * rows_per_sample * C) of all earlier rows, total = that of all rows. */
int dspn_copy_block_batch_f32(const void *table, int n, long long total, void *stream);
/* (B, N, C) -> (B, C, N) */
int dspn_transpose_bnc_f32(const float *src, float *dst, int B, int N, int C, void *stream);

/* ---- Tap-expanded form of a stride-1 RxS convolution with few output channels (score3_conv,
 * multitask_symbol_builder.py:583: 3328 -> 19, 3x3): run dspn_conv2d_forward_f32 as a 1x1 convolution
 * with Cout*R*S output channels on the unchanged weight buffer ([Cout][R][S][Cin] == [(Cout*R*S)][Cin]),
 * then tap_sum; in backward tap_spread builds the (co, tap) gradient for the 1x1 weight-gradient.
 *   y[n,h,w,co]              = bias[co] + sum_{r,s} z[n, h+r-pad_h, w+s-pad_w, co*R*S + r*S + s]
 *   dz[n,h,w,co*R*S + r*S+s] = dy[n, h-(r-pad_h), w-(s-pad_w), co]                                   */
int dspn_tap_sum_f32(const float *z, const float *bias, float *y, int N, int H, int W, int Cout, int ldy,
                     int ldz, int R, int S, int pad_h, int pad_w, void *stream);
int dspn_tap_spread_f32(const float *dy, float *dz, int N, int H, int W, int Cout, int ldy, int ldz, int R,
                        int S, int pad_h, int pad_w, void *stream);

/* ---- Pooling (mx.sym.Pooling: symbol/resnet.py:98 max 3x3/2 pad 1;
 * multitask_symbol_builder.py:560-562 avg k x k / k) ----------------------------------------- */
/* argmax (optional, N*Ho*Wo*C bytes): position r*k+s of the first maximum of each window in (h, w)
 * scan order, 255 if none */
int dspn_maxpool_forward_f32(const float *x, float *y, unsigned char *argmax, int N, int H, int W, int C, int k,
                             int stride, int pad, int Ho, int Wo, void *stream);
/* the same pooling of (relu)(x * in_scale[c] + in_shift[c]) (round 4): the BatchNorm(+ReLU) in front of a pooling layer
 * (symbol/resnet.py:96-98, bn0 -> relu0 -> pooling0) folded into the pooling pass -- the normalised tensor is never written.
 * in_scale / in_shift: C floats each (dspn_bn_stats*'s scale / shift).  out_absmax (optional, float tensors): the magnitude
 * block of y (DSPN_ABSMAX_SLOTS floats, zeroed by the caller). */
int dspn_maxpool_forward_bn_f32(const float *x, const float *in_scale, const float *in_shift, int in_relu, float *y,
                                unsigned char *argmax, int N, int H, int W, int C, int k, int stride, int pad, int Ho, int Wo,
                                float *out_absmax, void *stream);
/* ... and its backward (round 4): the backward of that BatchNorm(+ReLU) with its output gradient formed ON THE FLY from the
 * pooled gradient dy_pool (N, Ho, Wo, C) and the argmax record (the gather of dspn_maxpool_backward_argmax_f32) -- the dense
 * (N, H, W, C) gradient of the pooling input is neither written nor read back twice.  x: the BatchNorm input (N, H, W, C);
 * dx, dgamma (may be NULL), dbeta as dspn_bn_backward_f32; dx_absmax (optional): magnitude block of dx, zeroed by the caller;
 * workspace: dspn_bn_workspace_bytes(N * H * W, C). */
int dspn_bn_backward_maxpool_f32(const float *x, const float *scale, const float *shift, const float *dy_pool,
                                 const unsigned char *argmax, int N, int H, int W, int C, int k, int stride, int pad, int Ho,
                                 int Wo, const float *mean, const float *rstd, const float *gamma, float *dx, float *dgamma,
                                 float *dbeta, int relu, float *dx_absmax, void *workspace, size_t workspace_bytes, void *stream);
/* gradient goes to the first maximum of each window in (h, w) scan order; from the argmax record ... */
int dspn_maxpool_backward_argmax_f32(const unsigned char *argmax, const float *dy, float *dx, int N, int H,
                                     int W, int C, int k, int stride, int pad, int Ho, int Wo, void *stream);
/* ... or recomputed from x and y */
int dspn_maxpool_backward_f32(const float *x, const float *y, const float *dy, float *dx, int N, int H,
                              int W, int C, int k, int stride, int pad, int Ho, int Wo, void *stream);
int dspn_avgpool_forward_f32(const float *x, float *y, int N, int H, int W, int C, int k, int Ho, int Wo, void *stream);
int dspn_avgpool_backward_f32(const float *dy, float *dx, int N, int H, int W, int C, int k, int Ho, int Wo,
                              int accumulate, void *stream);

/* overlapping average pooling, floor ("valid") output size, padding counted in the divisor k*k
 * (symbol/inceptionv3.py:31 `Pooling(kernel=(3,3), stride=(1,1), pad=(1,1), pool_type='avg')`) */
int dspn_avgpool2d_forward_f32(const float *x, float *y, int N, int H, int W, int C, int k, int stride, int pad,
                               int Ho, int Wo, void *stream);
int dspn_avgpool2d_backward_f32(const float *dy, float *dx, int N, int H, int W, int C, int k, int stride, int pad,
                                int Ho, int Wo, int accumulate, void *stream);

/* ---- BilinearSampler over GridGenerator(affine = identity) (multitask_symbol_builder.py:
 * 574-581): align-corners bilinear resize (Hin,Win) -> (Ho,Wo), written into / read from a
 * channel slice [coff, coff+C) of a (N,Ho,Wo,ldo) concat buffer. ------------------------------ */
int dspn_bilinear_forward_f32(const float *x, float *y, int N, int Hin, int Win, int C, int Ho, int Wo,
                              int ldo, int coff, void *stream);
/* the same resize ADDED to the destination channels (a sum of several resized maps, accumulated in call order) */
int dspn_bilinear_forward_acc_f32(const float *x, float *y, int N, int Hin, int Win, int C, int Ho, int Wo,
                                  int ldo, int coff, void *stream);
int dspn_bilinear_backward_f32(const float *dy, float *dx, int N, int Hin, int Win, int C, int Ho, int Wo,
                               int ldo, int coff, void *stream);
/* the same gradient computed separably (along W into a scratch tensor, then along H): ~2s loads per source pixel
 * instead of ~(2s)^2 for an s-fold upsampling; the summation order differs from the one-pass kernel */
size_t dspn_bilinear_backward_workspace_bytes(int N, int Win, int C, int Ho);
int dspn_bilinear_backward_ws_f32(const float *dy, float *dx, int N, int Hin, int Win, int C, int Ho, int Wo,
                                  int ldo, int coff, void *workspace, size_t workspace_bytes, void *stream);

/* ---- GridGenerator(transform_type='affine', target_shape=(Ho,Wo)) + BilinearSampler with the LEARNABLE
 * `affine_matrix` argument (symbol/multitask_symbol_builder.py:574-581; initialised to (1,0,0,0,1,0) by
 * multi_init.py:72 and updated by the optimizer like every other argument, multi_solver.py:291-293).
 * theta: 6 floats in DEVICE memory (row-major 2x3), one grid shared by the whole batch.  With the identity theta
 * the result equals dspn_bilinear_forward_f32 bit for bit.  Up to DSPN_SAMPLER_MAX_SOURCES source maps per call:
 * source i (N,Hin[i],Win[i],C[i]) is sampled into channels [coff[i], coff[i]+C[i]) of y (N,Ho,Wo,ldo); sources
 * whose channel ranges coincide are summed (in table order), channels no source covers are written as 0.
 * x / Hin / Win / C / coff are HOST arrays of nsrc entries (x[i] are device pointers). */
#define DSPN_SAMPLER_MAX_SOURCES 8
int dspn_affine_sampler_forward_f32(const float *const *x, const int *Hin, const int *Win, const int *C, const int *coff,
                                    int nsrc, const float *theta, float *y, int N, int Ho, int Wo, int ldo, void *stream);
/* gradient with respect to ONE source map: dx (N,Hin,Win,C) (+)= sum over the target pixels that sample it of
 * weight * dy[..., coff:coff+C]; gather form, fixed summation order (no atomics) */
int dspn_affine_sampler_backward_data_f32(const float *dy, const float *theta, float *dx, int N, int Hin, int Win, int C,
                                          int Ho, int Wo, int ldo, int coff, int accumulate, void *stream);
/* (dspn_affine_sampler_backward_data_theta_f32 below also takes dx_absmax: the magnitude block of the dx it stores) */
/* gradient with respect to theta, all sources of a forward call at once: dtheta[6] (+)= sum over batch, target
 * pixels, sources and channels (BilinearSampler's grid gradient contracted with GridGenerator's backward);
 * two-stage fixed-order reduction in double */
size_t dspn_affine_sampler_theta_workspace_bytes(int N, int Ho, int Wo);
int dspn_affine_sampler_backward_theta_f32(const float *const *x, const int *Hin, const int *Win, const int *C,
                                           const int *coff, int nsrc, const float *theta, const float *dy, int N, int Ho,
                                           int Wo, int ldo, float *dtheta, int accumulate, void *workspace,
                                           size_t workspace_bytes, void *stream);
/* the data gradient of ONE source map and that source's share of the theta gradient from the same pass (round 4): as
 * dspn_affine_sampler_backward_data_f32, plus theta_partial[dspn_affine_sampler_theta_rows()][6] (doubles, DEVICE, overwritten) -- per source
 * pixel the sum over the target pixels that sample it of (d weight / d grid) . <dy[target], x[source pixel]> x (x_t, y_t, 1).
 * x: the source map itself (N,Hin,Win,C), as the forward call sampled it; it MAY be the buffer dx when accumulate == 0
 * (every workgroup reads its pixel before it writes it).  The sum over the rows of every source of one forward call is
 * d L / d theta (GridGenerator's backward): dspn_affine_sampler_theta_reduce adds them in a fixed order, two levels, in
 * double.  Equals dspn_affine_sampler_backward_theta_f32 up to the summation order.
 * dx_absmax (optional, float tensors): DSPN_ABSMAX_SLOTS floats that receive the partial maxima of |dx| as stored. */
int dspn_affine_sampler_backward_data_theta_f32(const float *dy, const float *theta, const float *x, float *dx, int N, int Hin,
                                                int Win, int C, int Ho, int Wo, int ldo, int coff, int accumulate,
                                                double *theta_partial, size_t theta_partial_bytes, float *dx_absmax,
                                                void *workspace, size_t workspace_bytes, void *stream);
/* rows of theta_partial one call writes (N * Hin * Win, times the number of workgroups a SMALL source map's pixels are split
 * over: a 2 x 2 or 4 x 4 map would otherwise leave a few hundred workgroups walking thousands of target pixels each), and the
 * scratch such a split call needs for the partial data gradients (0 for the others; `workspace` may then be NULL) */
long long dspn_affine_sampler_theta_rows(int N, int Hin, int Win, int Ho);
size_t dspn_affine_sampler_backward_workspace_bytes(int N, int Hin, int Win, int C, int Ho);
size_t dspn_affine_sampler_theta_reduce_workspace_bytes(long long rows);
int dspn_affine_sampler_theta_reduce(const double *theta_partial, long long rows, float *dtheta, int accumulate, void *workspace,
                                     size_t workspace_bytes, void *stream);

/* ---- losses ---------------------------------------------------------------------------------- */
/* ---- Segmentation readouts (train/metric.py:100-133 CustomAccuracyMetric, evaluate/eval_metric.py:278-388
 * IoUMetric): pred = argmax over the C classes of each row (first maximum, like mx.nd.argmax);
 * counts[c] = #(label == c and pred == c), counts[C + c] = #(pred == c), counts[2C + c] = #(label == c),
 * counts[3C] = #(pred == label); counts: 3C+1 unsigned 64-bit DEVICE words, overwritten.  C <= 64. ---- */
int dspn_seg_counts_f32(const float *scores, const float *label, long long rows, int C, int ld,
                        unsigned long long *counts, void *stream);

/* Full-resolution segmentation read-out (multi_eval.py:28-34 prob_upsampling: mx.nd.BilinearSampler of the class
 * probabilities on the identity-affine grid GridGenerator(target_shape=(Ho, Wo)) then mx.nd.argmax(axis=1) -> uint8),
 * fused: prob is the NHWC probability tensor (N, Hin, Win, ld >= C), out (N, Ho, Wo) unsigned bytes, fully
 * overwritten.  Source coordinate (g + 1) * (I - 1) / 2 with g = -1 + o * 2 / (O - 1); corners outside the map
 * contribute 0; per class tl*wy*wx + tr*wy*(1-wx) + bl*(1-wy)*wx + br*(1-wy)*(1-wx) in that order; ties keep the
 * lowest class.  0 < C <= 256. */
int dspn_seg_upsample_argmax_f32(const float *prob, unsigned char *out, int N, int Hin, int Win, int C, int ld,
                                 int Ho, int Wo, void *stream);

/* SoftmaxOutput(multi_output, use_ignore) over the last (channel) axis of logits (rows, ld):
 * prob (rows, ld) = softmax over the first C channels (pad channels -> 0);
 * grad (rows, ld) = (prob - onehot(label)) * scale, 0 for rows whose label == ignore_label.
 * scale = grad_scale / (*valid_count) if valid_count != NULL (normalization='valid': number of
 * labels != ignore_label, at least 1, computed on device by dspn_count_f32) else grad_scale.
 * (multitask_symbol_builder.py:526-528 and :588).  grad may be NULL (forward only). */
int dspn_softmax_output_f32(const float *logits, const float *label, float *prob, float *grad,
                            long long rows, int C, int ld, float ignore_label, float grad_scale,
                            const float *valid_count, void *stream);
/* out[0] = max(1, #{i : mode 0: a[i] != ref ; mode 1: a[i] > ref}) as float */
int dspn_count_f32(const float *a, long long n, int mode, float ref, float *out, void *stream);
/* loc_loss = smooth_l1(mask*(pred-target), sigma=1) (multitask_symbol_builder.py:529-530) */
int dspn_smooth_l1_forward_f32(const float *pred, const float *target, const float *mask, float *loss,
                               long long n, void *stream);
/* MakeLoss(normalization='valid') backward: grad = grad_scale * mask * smooth_l1'(mask*(pred-target))
 * / *valid_count, valid_count = max(1, #{loss > 0}) (multitask_symbol_builder.py:531-532) */
int dspn_smooth_l1_backward_f32(const float *pred, const float *target, const float *mask, float *grad,
                                long long n, float grad_scale, const float *valid_count, void *stream);
/* MultiBoxMetric / seg cross-entropy readout (train/metric.py:27-46): out[0] = sum_i -log(prob[i,label_i]+eps)
 * over rows with label != ignore, out[1] = number of such rows.  prob is (rows, ld). */
int dspn_cross_entropy_sum_f32(const float *prob, const float *label, long long rows, int C, int ld,
                               float ignore_label, float eps, float *out2, void *stream);
/* out[0] = sum a[i] */
int dspn_sum_f32(const float *a, long long n, float *out, void *stream);

/* ---- optimizer (multi_solver.py:221,291-293: mx 'sgd' with momentum) --------------------------
 * g = rescale*grad + wd*w ; mom = momentum*mom - lr*g ; w += mom, over a flat parameter arena. */
int dspn_sgd_momentum_f32(float *w, const float *grad, float *mom, long long n, float lr, float momentum,
                          float wd, float rescale, void *stream);

/* ---- bfloat16 twins of the HBM-bound kernels (same conventions as the convolution twins above: activation pointers
 * are bfloat16, everything per-channel / reduced / float-only keeps its type; dspn_nchw_to_nhwc_bf16 takes the float
 * NCHW image and writes the bf16 NHWC tensor; dspn_softmax_output_bf16 reads bf16 logits, writes FLOAT probabilities
 * and the bf16 gradient). ------------------------------------------------------------------------------------- */
int dspn_bn_stats_bf16(const dspn_bf16 *x, long long rows, int C, float eps, const float *gamma,
                      const float *beta, float *mean, float *rstd, float *scale, float *shift,
                      void *workspace, size_t workspace_bytes, void *stream);
int dspn_bn_apply_bf16(const dspn_bf16 *x, const float *scale, const float *shift, dspn_bf16 *y, long long rows,
                      int C, int relu, float *out_absmax /* ignored */, void *stream);
int dspn_bn_backward_bf16(const dspn_bf16 *x, const float *scale, const float *shift, const dspn_bf16 *dy,
                         const float *mean, const float *rstd, const float *gamma, dspn_bf16 *dx,
                         float *dgamma, float *dbeta, long long rows, int C, int relu, int accumulate,
                         float *dx_absmax_unused, void *workspace, size_t workspace_bytes, void *stream);
int dspn_bn_backward_from_sums_bf16(const dspn_bf16 *x, const float *scale, const float *shift, const dspn_bf16 *dy,
                                   const float *mean, const float *rstd, const float *gamma, const float *tile_sums,
                                   int tiles, dspn_bf16 *dx, float *dgamma, float *dbeta, long long rows, int C, int relu,
                                   int accumulate, float *dx_absmax_unused, float *dx_absmin_unused, const float *dy_absmax_unused,
                                   const float *x_chan_minmax_unused, int dx_planes /* must be 0 */, void *workspace,
                                   size_t workspace_bytes, void *stream);
int dspn_add_bf16(const dspn_bf16 *a, const dspn_bf16 *b, dspn_bf16 *out, long long n, void *stream);
int dspn_relu_backward_bf16(const dspn_bf16 *y, const dspn_bf16 *dy, dspn_bf16 *dx, long long n, int accumulate, void *stream);
int dspn_relu_backward_colsum_bf16(const dspn_bf16 *y, const dspn_bf16 *dy, dspn_bf16 *dx, long long rows, int C, int ld,
                                  float *out, float *dx_absmax_unused /* must be NULL */, void *workspace, size_t workspace_bytes,
                                  void *stream);
int dspn_colsum_bf16(const dspn_bf16 *a, long long rows, int C, int ld, float *out,
                    void *workspace, size_t workspace_bytes, void *stream);
int dspn_nchw_to_nhwc_bf16(const float *src, dspn_bf16 *dst, int N, int C, int H, int W, int Cp, void *stream);
int dspn_copy_block_bf16(const dspn_bf16 *src, dspn_bf16 *dst, int samples, long long rows_per_sample, int C,
                        long long src_sample_stride, int lds, int soff,
                        long long dst_sample_stride, int ldd, int doff, int accumulate, void *stream);
int dspn_tap_sum_bf16(const dspn_bf16 *z, const float *bias, dspn_bf16 *y, int N, int H, int W, int Cout, int ldy,
                     int ldz, int R, int S, int pad_h, int pad_w, void *stream);
int dspn_tap_spread_bf16(const dspn_bf16 *dy, dspn_bf16 *dz, int N, int H, int W, int Cout, int ldy, int ldz, int R,
                        int S, int pad_h, int pad_w, void *stream);
int dspn_maxpool_forward_bf16(const dspn_bf16 *x, dspn_bf16 *y, unsigned char *argmax, int N, int H, int W, int C, int k,
                             int stride, int pad, int Ho, int Wo, void *stream);
int dspn_maxpool_forward_bn_bf16(const dspn_bf16 *x, const float *in_scale, const float *in_shift, int in_relu, dspn_bf16 *y,
                                 unsigned char *argmax, int N, int H, int W, int C, int k, int stride, int pad, int Ho, int Wo,
                                 float *out_absmax_unused, void *stream);
int dspn_maxpool_backward_argmax_bf16(const unsigned char *argmax, const dspn_bf16 *dy, dspn_bf16 *dx, int N, int H,
                                     int W, int C, int k, int stride, int pad, int Ho, int Wo, void *stream);
int dspn_maxpool_backward_bf16(const dspn_bf16 *x, const dspn_bf16 *y, const dspn_bf16 *dy, dspn_bf16 *dx, int N, int H,
                              int W, int C, int k, int stride, int pad, int Ho, int Wo, void *stream);
int dspn_avgpool_forward_bf16(const dspn_bf16 *x, dspn_bf16 *y, int N, int H, int W, int C, int k, int Ho, int Wo, void *stream);
int dspn_avgpool_backward_bf16(const dspn_bf16 *dy, dspn_bf16 *dx, int N, int H, int W, int C, int k, int Ho, int Wo,
                              int accumulate, void *stream);
int dspn_avgpool2d_forward_bf16(const dspn_bf16 *x, dspn_bf16 *y, int N, int H, int W, int C, int k, int stride, int pad,
                               int Ho, int Wo, void *stream);
int dspn_avgpool2d_backward_bf16(const dspn_bf16 *dy, dspn_bf16 *dx, int N, int H, int W, int C, int k, int stride, int pad,
                                int Ho, int Wo, int accumulate, void *stream);
int dspn_softmax_output_bf16(const dspn_bf16 *logits, const float *label, float *prob, dspn_bf16 *grad,
                            long long rows, int C, int ld, float ignore_label, float grad_scale,
                            const float *valid_count, void *stream);
int dspn_affine_sampler_backward_data_bf16(const dspn_bf16 *dy, const float *theta, dspn_bf16 *dx, int N, int Hin, int Win, int C,
                                          int Ho, int Wo, int ldo, int coff, int accumulate, void *stream);
int dspn_affine_sampler_backward_data_theta_bf16(const dspn_bf16 *dy, const float *theta, const dspn_bf16 *x, dspn_bf16 *dx, int N,
                                                 int Hin, int Win, int C, int Ho, int Wo, int ldo, int coff, int accumulate,
                                                 double *theta_partial, size_t theta_partial_bytes, float *dx_absmax /* ignored */,
                                                 void *workspace /* unused */, size_t workspace_bytes, void *stream);
int dspn_affine_sampler_forward_bf16(const dspn_bf16 *const *x, const int *Hin, const int *Win, const int *C, const int *coff,
                                    int nsrc, const float *theta, dspn_bf16 *y, int N, int Ho, int Wo, int ldo, void *stream);
int dspn_affine_sampler_backward_theta_bf16(const dspn_bf16 *const *x, const int *Hin, const int *Win, const int *C,
                                           const int *coff, int nsrc, const float *theta, const dspn_bf16 *dy, int N, int Ho,
                                           int Wo, int ldo, float *dtheta, int accumulate, void *workspace,
                                           size_t workspace_bytes, void *stream);
/* dspn_copy_block between storage types: bf16 -> float (SSD head maps into the float loss inputs) and float -> bf16
 * (their gradients back into the per-map gradient tensors) */
int dspn_copy_block_bf16_f32(const dspn_bf16 *src, float *dst, int samples, long long rows_per_sample, int C,
                             long long src_sample_stride, int lds, int soff, long long dst_sample_stride,
                             int ldd, int doff, int accumulate, void *stream);
int dspn_copy_block_f32_bf16(const float *src, dspn_bf16 *dst, int samples, long long rows_per_sample, int C,
                             long long src_sample_stride, int lds, int soff, long long dst_sample_stride,
                             int ldd, int doff, int accumulate, void *stream);

#ifdef __cplusplus
}
#endif
#endif  /* DSPN_NN_H_ */
