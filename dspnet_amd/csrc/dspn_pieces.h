// The two-piece fp16 operand format of DSPN_MATH_F32_F16X2 (include/dspn_nn.h): scale from a magnitude block, the cut of a
// float4 into its two fp16 pieces, the repair of infinite elements.  Shared by the convolution kernels (conv.hip) and by the
// BatchNorm backward that writes an output gradient as piece planes (nn.hip).  PRIVATE to csrc/.
#pragma once
#include <hip/hip_runtime.h>

namespace dspn {
namespace pieces {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));   // 4 x 16 bits: the piece registers / LDS images are typed bf16x4 in both split modes

constexpr int kAbsmaxSlots = 64;    // a magnitude "scalar" is 64 partial maxima (see absmax_kernel): one per lane here
// Non-finite elements (round 4): a partial maximum that is infinite only says "this tensor holds an inf" -- the scale comes
// from the FINITE partial maxima (64 independent slots: the finite values still bound the rest of the tensor unless every
// slot met an inf), so the finite elements keep their fp32 accuracy.  NaNs never enter a magnitude block (fmaxf skips them).
// What an infinite ELEMENT becomes: in the weight-plane kernel and in the weight-gradient kernel its pieces are repaired
// (repair_inf below; operand_nonfinite() says when) and every product is what fp32 gives.  In conv_nt_kernel's own loaders
// (activations of the forward pass / output gradients of the data gradient) they are NOT: a wave-uniform branch around the
// repair inside the k-loop cost the 128-register kernels 12 - 19 registers (scratch, -4 % on the training step), so an
// infinite activation gives NaN (h1 = inf - inf) in every output it touches -- non-finite where fp32 is non-finite, but not
// the signed infinity (include/dspn_nn.h; tests/test_nn_gpu.py::test_two_piece_math_propagates_non_finite_operands...).
__device__ __forceinline__ float operand_scale(const float *absmax) {
  if (!absmax) return 1.f;
  float m = absmax[threadIdx.x & 63];
  m = m < 3.0e38f ? m : 0.f;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (!(m > 0.f)) return 1.f;
  int e;
  (void)frexpf(m, &e);                       // m = f * 2^e, 0.5 <= f < 1  ->  m * 2^(15 - e) < 2^15
  e = 15 - e;
  e = e < -100 ? -100 : (e > 100 ? 100 : e);
  return __uint_as_float((unsigned)(127 + e) << 23);
}
// wave-uniform: does the tensor behind this magnitude block hold an infinite element?
__device__ __forceinline__ bool operand_nonfinite(const float *absmax) {
  if (!absmax) return false;
  const float m = absmax[threadIdx.x & 63];
  return __ballot(!(m < 3.0e38f)) != 0ull;
}
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// The two pieces of an infinite element come out of split2h as h0 = +-inf, h1 = NaN (inf - inf), and h0 g1 alone would be
// NaN wherever the other operand's residual piece is 0.  Repaired (only in tensors whose magnitude block says so, in the
// store path, outside the MFMA-interleaved piece arithmetic): h0 = +-65504, h1 = +-inf, so that x w = h0 g0 + h0 g1 + h1 g0
// is +-inf with the sign of x w, or NaN where w is 0 -- what fp32 arithmetic gives.
__device__ __forceinline__ void repair_inf(bf16x4 &p0, bf16x4 &p1) {
  f16x4 h0 = __builtin_bit_cast(f16x4, p0), h1 = __builtin_bit_cast(f16x4, p1);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float f = (float)h0[e];
    if (f == __builtin_huge_valf() || f == -__builtin_huge_valf()) {
      h1[e] = h0[e];
      h0[e] = f > 0.f ? (_Float16)65504.f : (_Float16)-65504.f;
    }
  }
  p0 = __builtin_bit_cast(bf16x4, h0);
  p1 = __builtin_bit_cast(bf16x4, h1);
}
__device__ __forceinline__ void split2h(const float4 v, const float s, bf16x4 &p0, bf16x4 &p1) {
  const float4 u = make_float4(v.x * s, v.y * s, v.z * s, v.w * s);
  const f16x4 h0 = {(_Float16)u.x, (_Float16)u.y, (_Float16)u.z, (_Float16)u.w};
  const f16x4 h1 = {(_Float16)(u.x - (float)h0[0]), (_Float16)(u.y - (float)h0[1]), (_Float16)(u.z - (float)h0[2]),
                    (_Float16)(u.w - (float)h0[3])};
  p0 = __builtin_bit_cast(bf16x4, h0);     // (the piece registers / LDS images are typed bf16x4: 4 x 16 bits either way)
  p1 = __builtin_bit_cast(bf16x4, h1);
}


}  // namespace pieces
}  // namespace dspn
