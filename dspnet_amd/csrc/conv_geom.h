// Geometry record and helpers shared by the convolution translation units (conv.hip: every math mode and storage type;
// conv_wide.hip: the wide tile family of the two-piece math).  PRIVATE to csrc/.
#pragma once
#include <hip/hip_runtime.h>

namespace dspn {
namespace conv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <typename st_t>
struct ConvGeomT {
  int N, Hin, Win, Cin;            // gathered tensor (Cin % kEPC == 0)
  int Hg, Wg;                      // grid of output points per image
  int ish, isw, ioh, iow, idh, idw;  // ih = i*ish + ioh + tr*idh
  int TR, TS;                      // taps enumerated
  int WTAPS, WS, wr0, wrs, ws0, wss;  // weight tap = (wr0+tr*wrs)*WS + ws0+ts*wss
  int Cout;
  long long obs;                   // output batch stride (floats)
  int OW, osh, osw, ooh, oow, ldc; // out pixel = ((i*osh+ooh)*OW + j*osw+oow)*ldc
  int flags;                       // 1 bias, 2 relu, 4 accumulate, 8 add residual (same layout as out), 16 float4 rows legal, 32 ReLU after the input affine
  int dense;                       // output address = m*ldc (no decomposition needed)
  int dbg;                         // timing-only ablation bits (DSPN_ABLATE builds), 0 in production
  int bf16;                        // host side only: the call's math mode (DSPN_MATH_*): 0 fp32 MFMA, 1 bf16 MFMA, 2 three-piece bf16
  unsigned in_bytes, w_bytes;      // sizes of the gathered tensor / weight tensor (buffer bounds)
  // optional per-input-channel affine (+ReLU when flags & 32) applied to the gathered tensor on its way into
  // LDS: the BatchNorm(+ReLU) in front of a convolution (symbol/resnet.py:30-45) without materialising its output
  const float *in_scale, *in_shift;
  // optional BatchNorm statistics of the OUTPUT, per row tile: stats[(mt*2 + 0)*Cout + c] = mean over the tile's
  // rows, stats[(mt*2 + 1)*Cout + c] = sum of squared deviations from that mean (merged by dspn_bn_stats_from_tiles_f32)
  float *stats;
  // optional (two-piece math, with stats): minmax[(mt*2 + 0)*Cout + c] = smallest, [(mt*2 + 1)*Cout + c] = largest stored value
  // of the tile's rows in column c.  A BatchNorm(+ReLU) of the output is monotone per channel, so the magnitude of what the
  // NEXT convolution's loader forms from this tensor is the largest |f_c(extreme)| over this small table
  // (dspn_absmax_f32 on it, with the affine) -- instead of a pass over the whole tensor
  float *minmax;
  // optional BatchNorm-backward sums of the OUTPUT (a data gradient dy of a BatchNorm(+ReLU) output whose input was
  // bn_x, same layout as out): per row tile t, bn_sums[((tile_base + t)*2 + 0)*Cout + c] = sum of dy' and
  // [... + 1 ...] = sum of dy' * xhat, with dy' = dy where (bn_x*bn_scale + bn_shift > 0 or no ReLU) else 0 and
  // xhat = (bn_x - bn_mean) * bn_rstd: the layout dspn_bn_backward_from_sums_f32 reads
  const st_t *bn_x;
  const float *bn_scale, *bn_shift, *bn_mean, *bn_rstd;
  float *bn_sums;
  int bn_relu, bn_tile_base;
  // optional, with bn_sums (round 4): 64 partial maxima of |dx| as stored -- the largest output gradient the following
  // BatchNorm backward meets, from which it bounds the dx IT stores before it writes it as fp16 piece planes (dspn_nn.h)
  unsigned *bn_dy_absmax;
  // DSPN_MATH_F32_F16X2: device scalars holding the largest magnitude of the A operand (the gathered tensor AFTER its input
  // affine) and of the B operand (the weights); the kernel derives the power-of-two scales that put them just below 2^15
  // (operand_scale) and undoes both in the epilogue.  NULL = scale 1 (the caller vouches for |operand| < 65504).
  const float *a_absmax, *b_absmax;
  int a_planes;                    // host side only: the gathered tensor is fp16 piece planes (conv_nt_kernel, EPIX & 4)
  // host side only: the weight operand as three bf16 piece planes [Cout][WTAPS][Cin / 32][3][32] (split mode, Cin % 32 == 0:
  // dspn_conv2d_weight_planes_f32); the kernel then receives this pointer in place of the float weights
  const void *w_planes;
};

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  // blocks b, b+8, ... share an XCD (round-robin dispatch): give each XCD a
  // contiguous run of logical tiles so neighbouring tiles share its L2.
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
  const int start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return start + (bid >> 3);
}


// conv_wide.hip: which wide tile a plane-fed layer runs on: 0 none (conv_nt_kernel), 1 256 x 128, 2 128 x 256 (8 waves, one
// workgroup per CU), 3 128 x 128 on four waves (two workgroups per CU), 4 256 x 64 on four waves (Cout <= 64, 64-row
// BatchNorm tables); M output points, nk k-steps of 32 channels
int wide_tile_choice(long long M, int Cout, int nk, int fused_epilogue);
// conv_wide.hip: one launch of the wide family (shape 1: 256 x 128, 2: 128 x 256, 3: 128 x 128 on four waves) -- both operands
// piece planes, vector epilogue, at least one tap (dispatch_nt checks)
int launch_wide(int shape, const float *in, const float *w_planes, const float *bias, float *out, const ConvGeomT<float> &g,
                hipStream_t s, const float *residual);
// ... and on bfloat16 tensors (conv_wide_h.hip): the activations / weight copies themselves are the operands; whole 64-channel
// blocks, no input affine
int launch_wide(int shape, const __bf16 *in, const __bf16 *w, const float *bias, __bf16 *out, const ConvGeomT<__bf16> &g,
                hipStream_t s, const __bf16 *residual);

// conv_wide.hip (conv_stem.h): the 7x7 / 2, pad 3, 4 -> 64 channel stem convolution in the two-piece math with BatchNorm
// statistics (64-row tiles) and extremes: 0 = launched, 1 = not this shape (the generic kernel runs), < 0 = launch error
int launch_stem(const float *x, const float *w, float *y, int N, int H, int W, int Cin, int Cout, int Ho, int Wo,
                const float *x_absmax, const float *w_absmax, float *stats, float *minmax, hipStream_t s);

}  // namespace conv
}  // namespace dspn
