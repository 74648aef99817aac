// Batch image pipeline of the record iterator for MI355X (gfx950); C ABI and the arithmetic contract in
// include/dspn_augment.h.  Two kernels on the caller's stream, each covering the whole batch:
//   augment_image_kernel : affine warp (OpenCV's fixed-point bilinear, constant border) + horizontal flip + channel
//                          reorder + mean subtraction, uint8 HWC source -> float32 planes.  One thread per output
//                          pixel: consecutive lanes write consecutive floats of each plane (coalesced 256-B stores),
//                          the 4 x 3 source bytes it reads are shared with its neighbours through L1/L2.
//   augment_seg_kernel   : nearest warp + flip + quarter-size nearest resize + LUT, evaluated only at the pixels the
//                          resize keeps (1/16 of the warped map, which is never materialised).
// HBM-bound byte work; nothing here is shaped for the matrix cores.  Compiled with -ffp-contract=off: the double
// products below must round exactly like the separate multiplies of the reference implementation.
#include "dspn_common.h"
#include "../../include/dspn_augment.h"

namespace {

constexpr int kT = 256;

__device__ __forceinline__ int cv_round(double v) { return __double2int_rn(v); }   // cvRound: round half to even
__device__ __forceinline__ int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

__global__ __launch_bounds__(kT) void augment_image_kernel(const unsigned char *__restrict__ images,
                                                           const dspn_warp_sample *__restrict__ samples, int H, int W,
                                                           int c0, int c1, int c2, double mean0, double mean1,
                                                           double mean2, float *__restrict__ out) {
  const int b = blockIdx.z, y = blockIdx.y;
  const int x = blockIdx.x * kT + threadIdx.x;
  if (x >= W) return;
  const dspn_warp_sample s = samples[b];
  const unsigned char *src = images + s.img_offset;
  const int xd = s.flip ? W - 1 - x : x;                       // column of the (unflipped) warped image
  const int X0 = cv_round((s.minv[1] * (double)y + s.minv[2]) * 1024.0) + 16;
  const int Y0 = cv_round((s.minv[4] * (double)y + s.minv[5]) * 1024.0) + 16;
  const int X = (X0 + cv_round(s.minv[0] * (double)xd * 1024.0)) >> 5;
  const int Y = (Y0 + cv_round(s.minv[3] * (double)xd * 1024.0)) >> 5;
  const int sx = sat_short(X >> 5), sy = sat_short(Y >> 5);
  const int fx = X & 31, fy = Y & 31;
  const int w00 = 32 * (32 - fy) * (32 - fx), w01 = 32 * (32 - fy) * fx, w10 = 32 * fy * (32 - fx), w11 = 32 * fy * fx;
  const bool vx0 = (unsigned)sx < (unsigned)s.src_w, vx1 = (unsigned)(sx + 1) < (unsigned)s.src_w;
  const bool vy0 = (unsigned)sy < (unsigned)s.src_h, vy1 = (unsigned)(sy + 1) < (unsigned)s.src_h;
  const long long r0 = (long long)sy * s.src_w, r1 = (long long)(sy + 1) * s.src_w;
  const int cmap[3] = {c0, c1, c2};
  const double mean[3] = {mean0, mean1, mean2};
  const long long plane = (long long)H * W;
  float *o = out + (long long)b * 3 * plane + (long long)y * W + x;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int ch = cmap[c];
    const int t00 = (vy0 && vx0) ? src[(r0 + sx) * 3 + ch] : s.img_border;
    const int t01 = (vy0 && vx1) ? src[(r0 + sx + 1) * 3 + ch] : s.img_border;
    const int t10 = (vy1 && vx0) ? src[(r1 + sx) * 3 + ch] : s.img_border;
    const int t11 = (vy1 && vx1) ? src[(r1 + sx + 1) * 3 + ch] : s.img_border;
    const int v = (t00 * w00 + t01 * w01 + t10 * w10 + t11 * w11 + (1 << 14)) >> 15;
    o[c * plane] = (float)((double)v - mean[c]);
  }
}

__global__ __launch_bounds__(kT) void augment_seg_kernel(const unsigned char *__restrict__ segs,
                                                         const dspn_warp_sample *__restrict__ samples, int H, int W,
                                                         const unsigned char *__restrict__ lut, float *__restrict__ out) {
  const int Hq = H / 4, Wq = W / 4;
  const int b = blockIdx.z, yq = blockIdx.y;
  const int xq = blockIdx.x * kT + threadIdx.x;
  if (xq >= Wq) return;
  const dspn_warp_sample s = samples[b];
  float *o = out + ((long long)b * Hq + yq) * Wq + xq;
  if (s.seg_offset < 0) { *o = 0.f; return; }
  const unsigned char *src = segs + s.seg_offset;
  const int y = 4 * yq;                                        // the resize keeps pixel (4 yq, 4 xq) of the flipped map
  const int xf = 4 * xq;
  const int xd = s.flip ? W - 1 - xf : xf;
  const int X0 = cv_round((s.minv[1] * (double)y + s.minv[2]) * 1024.0) + 512;
  const int Y0 = cv_round((s.minv[4] * (double)y + s.minv[5]) * 1024.0) + 512;
  const int sx = sat_short((X0 + cv_round(s.minv[0] * (double)xd * 1024.0)) >> 10);
  const int sy = sat_short((Y0 + cv_round(s.minv[3] * (double)xd * 1024.0)) >> 10);
  int v = s.seg_border;
  if ((unsigned)sx < (unsigned)s.src_w && (unsigned)sy < (unsigned)s.src_h) v = src[(long long)sy * s.src_w + sx];
  if (lut) v = lut[v & 255];
  *o = (float)v;
}

}  // namespace

extern "C" int dspn_augment_batch_u8(const unsigned char *images, const unsigned char *segs,
                                     const dspn_warp_sample *samples, int B, int H, int W, const int channel_map[3],
                                     const double mean[3], const unsigned char *lut, float *data_out, float *seg_out,
                                     void *stream) {
  DSPN_REQUIRE(images && samples && data_out && channel_map && mean, "augment_batch: null argument");
  DSPN_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0 && H <= 65535 && H % 4 == 0 && W % 4 == 0,
               "augment_batch: bad shape (0 < B, H <= 65535; H, W multiples of 4)");
  for (int c = 0; c < 3; ++c) DSPN_REQUIRE((unsigned)channel_map[c] < 3u, "augment_batch: channel_map entries must be 0..2");
  DSPN_REQUIRE(!seg_out || segs, "augment_batch: seg_out needs the label pool");
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(augment_image_kernel, dim3(dspn::cdiv(W, kT), H, B), dim3(kT), 0, s, images, samples, H, W,
                     channel_map[0], channel_map[1], channel_map[2], mean[0], mean[1], mean[2], data_out);
  if (seg_out)
    hipLaunchKernelGGL(augment_seg_kernel, dim3(dspn::cdiv(W / 4, kT), H / 4, B), dim3(kT), 0, s, segs, samples, H, W, lut,
                       seg_out);
  return dspn::check_launch("augment_batch");
}
