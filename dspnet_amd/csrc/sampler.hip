// GridGenerator(transform_type='affine') + BilinearSampler with a LEARNABLE affine_matrix
// (symbol/multitask_symbol_builder.py:574-581, multi_init.py:72; the optimizer updates it like any other
// argument, multi_solver.py:291-293).  C ABI in include/dspn_nn.h.
//
// Semantics restated from MXNet's operators (not vendored by the reference):
//   grid_dst(ho, wo) = (x_t, y_t, 1),  x_t = -1 + wo * 2/(Wo-1),  y_t = -1 + ho * 2/(Ho-1)
//   grid_src = theta(2x3) . grid_dst                        (one grid, shared by the whole batch)
//   x_real = (x_src + 1) * (Win-1) / 2,  y_real likewise;  the four neighbours are weighted
//   bilinearly and a neighbour outside the image contributes 0.
//   d out / d x_real = (1-fy) (v01 - v00) + fy (v11 - v10)   (out-of-image neighbours are 0),
//   d L / d theta = sum over target pixels of [dL/dx_src * (x_t, y_t, 1), dL/dy_src * (x_t, y_t, 1)].
// With theta = (1,0,0,0,1,0) every product above is exact, so the identity case reproduces the plain
// align-corners resize of nn.hip bit for bit.
//
// Forward: one thread per (pixel, float4 column) of the destination sums every source that covers the
// column (a concat writes disjoint slices, the per-level evaluation of score3_conv sums six maps in one
// pass).  Data gradient: gather form -- a workgroup per source pixel walks the pre-image of its 2x2
// footprint under the affine map (bounding box from the inverse map, exact membership test with the
// forward's own coordinate function), rows of the box split over the workgroup's slices and summed in a
// fixed order: no atomics, bitwise reproducible.  theta gradient: one wave per target pixel (lanes over
// channels, butterfly reduction), per-wave double accumulators, fixed-order final sum.
#include "dspn_common.h"
#include "dspn_store.h"
#include "../../include/dspn_nn.h"

// compiled twice (dspn_store.h): float maps -> `*_f32`; through sampler_h.hip -> bfloat16 maps -> `*_bf16`
// (theta, its gradient and the reduction workspace are float / double in both)
using dspn::st_t;
using dspn::A4Ptr;
using dspn::CA4Ptr;

#pragma clang fp contract(fast)   // as nn.hip: the identity grid then reproduces its plain resize kernel bit for bit

namespace {

constexpr int kMaxSrc = DSPN_SAMPLER_MAX_SOURCES;

struct Src { const st_t *x; int Hin, Win, C4, coff4; };
__device__ __forceinline__ float4 ld4(const st_t *p) { return dspn::CA1Ptr(p).vec4()[0]; }
struct SrcTable { Src s[kMaxSrc]; int n; };

struct Theta { float t[6]; };
__device__ __forceinline__ Theta load_theta(const float *theta) {
  Theta r;
#pragma unroll
  for (int i = 0; i < 6; ++i) r.t[i] = theta[i];
  return r;
}
__device__ __forceinline__ float tgt_coord(int o, int O) {
  return O > 1 ? -1.f + (float)o * (2.f / (float)(O - 1)) : 0.f;
}
// source coordinates of target pixel (ho, wo); clamped to a band around the image so that the int conversion is
// defined for any theta (inside the band nothing changes; outside every neighbour is out of the image anyway)
__device__ __forceinline__ void src_xy(const Theta &th, float xt, float yt, int Hin, int Win, float &xs, float &ys) {
  const float gx = th.t[0] * xt + th.t[1] * yt + th.t[2];
  const float gy = th.t[3] * xt + th.t[4] * yt + th.t[5];
  xs = (gx + 1.f) * (float)(Win - 1) / 2.f;
  ys = (gy + 1.f) * (float)(Hin - 1) / 2.f;
  xs = fminf(fmaxf(xs, -2.f), (float)Win + 1.f);
  ys = fminf(fmaxf(ys, -2.f), (float)Hin + 1.f);
}

__global__ void sampler_fwd_kernel(const SrcTable tab, const float *__restrict__ theta, const A4Ptr y,
                                   int Ho, int Wo, int ld4, long long total) {
  const Theta th = load_theta(theta);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % ld4);
    long long t = i / ld4;
    const int wo = (int)(t % Wo); t /= Wo;
    const int ho = (int)(t % Ho);
    const long long n = t / Ho;
    const float xt = tgt_coord(wo, Wo), yt = tgt_coord(ho, Ho);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int si = 0; si < tab.n; ++si) {
      const Src s = tab.s[si];
      const int c = c4 - s.coff4;
      if (c < 0 || c >= s.C4) continue;
      float xs, ys;
      src_xy(th, xt, yt, s.Hin, s.Win, xs, ys);
      const int y0 = (int)floorf(ys), x0 = (int)floorf(xs);
      const float wy0 = 1.f - (ys - (float)y0), wx0 = 1.f - (xs - (float)x0);
      const CA4Ptr x(s.x);
#pragma unroll
      for (int dyy = 0; dyy < 2; ++dyy)
#pragma unroll
        for (int dxx = 0; dxx < 2; ++dxx) {
          const int yy = y0 + dyy, xx = x0 + dxx;
          if ((unsigned)yy >= (unsigned)s.Hin || (unsigned)xx >= (unsigned)s.Win) continue;
          const float wgt = (dyy ? 1.f - wy0 : wy0) * (dxx ? 1.f - wx0 : wx0);
          const float4 v = x[((n * s.Hin + yy) * s.Win + xx) * s.C4 + c];
          acc.x += wgt * v.x; acc.y += wgt * v.y; acc.z += wgt * v.z; acc.w += wgt * v.w;
        }
    }
    y[i] = acc;
  }
}

// one workgroup (64 channel lanes x SL slices) per source pixel
// TH (round 4): the gradient with respect to theta from the SAME pass.  The theta gradient is
//   sum over target pixels t and their four neighbours s of  (d weight(t, s) / d xs, d weight(t, s) / d ys) . <dy[t], x[s]>
// times (x_t, y_t, 1), and this kernel already visits every (target, neighbour) pair from the neighbour's side: with the
// forward value x[s] of its own source pixel in registers (read before dx, which may alias it, is written) every pair costs
// one dot product and six multiply-adds more.  d weight / d xs = -wy for the left neighbour (x0 == w), +wy for the right
// one; rows likewise.  Per workgroup the six sums leave as doubles, lanes and slices combined in a fixed order
// (tpart[pixel][6]; dspn_affine_sampler_theta_reduce adds the rows in a fixed order).  The stand-alone kernel it replaces
// (one wave per TARGET pixel, every source's four corner rows fetched per pixel) took 0.54 ms per training step to
// produce six numbers.
template <int SL, bool TH>
__global__ __launch_bounds__(64 * SL) void sampler_bwd_data_kernel(const st_t *__restrict__ dy,
                                                                   const float *__restrict__ theta,
                                                                   const A4Ptr dx, int Hin, int Win, int C4,
                                                                   int Ho, int Wo, int ldo, int coff, int accumulate,
                                                                   const st_t *__restrict__ xfwd, double *__restrict__ tpart,
                                                                   unsigned *__restrict__ absmax, float4 *__restrict__ part) {
  // part != NULL (small source maps, round 4): the rows of the pre-image box are cut into gridDim.y chunks, one workgroup
  // each -- a 2 x 2 or 4 x 4 source otherwise leaves 128 / 512 workgroups walking ~4000 target pixels each (0.15 ms per map
  // for a few KB of output) -- whose partial sums go to part[chunk][pixel][C4] and are added in chunk order by
  // sampler_bwd_reduce_kernel; the theta rows become tpart[pixel * chunks + chunk]
  __shared__ float4 red[SL > 1 ? SL : 1][64];
  float mx = 0.f;       // largest |dx| this thread stores (absmax: the magnitude block of dx, two-piece math)
  __shared__ double tred[TH ? SL : 1][6];
  float ta[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float khw = (float)(Win - 1) * 0.5f, khh = (float)(Hin - 1) * 0.5f;
  const Theta th = load_theta(theta);
  const int lane = threadIdx.x, sl = threadIdx.y;
  const long long pix = blockIdx.x;
  const int w = (int)(pix % Win);
  const int h = (int)((pix / Win) % Hin);
  const long long n = pix / ((long long)Win * Hin);
  // pixel-space affine map: xs = axw*wo + axh*ho + ax0, ys = ayw*wo + ayh*ho + ay0 (for the bounding box only)
  const float kx = Wo > 1 ? 2.f / (float)(Wo - 1) : 0.f, ky = Ho > 1 ? 2.f / (float)(Ho - 1) : 0.f;
  const float hw = (float)(Win - 1) / 2.f, hh = (float)(Hin - 1) / 2.f;
  const float x0t = Wo > 1 ? -1.f : 0.f, y0t = Ho > 1 ? -1.f : 0.f;
  const float axw = th.t[0] * kx * hw, axh = th.t[1] * ky * hw, ax0 = (th.t[0] * x0t + th.t[1] * y0t + th.t[2] + 1.f) * hw;
  const float ayw = th.t[3] * kx * hh, ayh = th.t[4] * ky * hh, ay0 = (th.t[3] * x0t + th.t[4] * y0t + th.t[5] + 1.f) * hh;
  const float det = axw * ayh - axh * ayw;
  int wo_lo = 0, wo_hi = Wo - 1, ho_lo = 0, ho_hi = Ho - 1;
  if (fabsf(det) > 1e-20f && isfinite(det)) {
    float fw_lo = 3e38f, fw_hi = -3e38f, fh_lo = 3e38f, fh_hi = -3e38f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float sx = (float)w + ((k & 1) ? 1.f : -1.f) - ax0, sy = (float)h + ((k & 2) ? 1.f : -1.f) - ay0;
      const float fw = (ayh * sx - axh * sy) / det, fh = (axw * sy - ayw * sx) / det;
      fw_lo = fminf(fw_lo, fw); fw_hi = fmaxf(fw_hi, fw); fh_lo = fminf(fh_lo, fh); fh_hi = fmaxf(fh_hi, fh);
    }
    if (isfinite(fw_lo) && isfinite(fw_hi) && isfinite(fh_lo) && isfinite(fh_hi)) {
      // one pixel of slack for the rounding of the inverse (a target pixel ON the edge of the pre-image has weight 0);
      // membership is tested exactly below
      wo_lo = (int)fmaxf(floorf(fw_lo) - 1.f, 0.f); wo_hi = (int)fminf(ceilf(fw_hi) + 1.f, (float)(Wo - 1));
      ho_lo = (int)fmaxf(floorf(fh_lo) - 1.f, 0.f); ho_hi = (int)fminf(ceilf(fh_hi) + 1.f, (float)(Ho - 1));
    }
  }
  if (part) {          // this workgroup's share of the box's rows
    const int nch = (int)gridDim.y, per = (ho_hi - ho_lo + nch) / nch;
    ho_lo += (int)blockIdx.y * per;
    ho_hi = min(ho_hi, ho_lo + per - 1);
  }
  for (int cb = 0; cb < C4; cb += 64) {
    const int c4 = cb + lane;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (TH && c4 < C4) xv = ld4(xfwd + (pix * C4 + c4) * 4);
    for (int ho = ho_lo + sl; ho <= ho_hi; ho += SL) {
      const float yt = tgt_coord(ho, Ho);
      for (int wo = wo_lo; wo <= wo_hi; ++wo) {
        float xs, ys;
        const float xt = tgt_coord(wo, Wo);
        src_xy(th, xt, yt, Hin, Win, xs, ys);
        const int y0 = (int)floorf(ys), x0 = (int)floorf(xs);
        float wy, wx, sy, sx;
        if (y0 == h) { wy = 1.f - (ys - (float)y0); sy = -1.f; }
        else if (y0 + 1 == h) { wy = 1.f - (1.f - (ys - (float)y0)); sy = 1.f; }
        else continue;
        if (x0 == w) { wx = 1.f - (xs - (float)x0); sx = -1.f; }
        else if (x0 + 1 == w) { wx = 1.f - (1.f - (xs - (float)x0)); sx = 1.f; }
        else continue;
        if (c4 >= C4) continue;
        const float wgt = wy * wx;
        const float4 v = ld4(dy + ((n * Ho + ho) * Wo + wo) * (long long)ldo + coff + c4 * 4);
        acc.x += wgt * v.x; acc.y += wgt * v.y; acc.z += wgt * v.z; acc.w += wgt * v.w;
        if (TH) {
          const float dot = v.x * xv.x + v.y * xv.y + v.z * xv.z + v.w * xv.w;
          const float cx = dot * (wy * sx) * khw, cy = dot * (wx * sy) * khh;
          ta[0] += cx * xt; ta[1] += cx * yt; ta[2] += cx;
          ta[3] += cy * xt; ta[4] += cy * yt; ta[5] += cy;
        }
      }
    }
    if (SL > 1) {
      red[sl][lane] = acc;
      __syncthreads();
      if (sl == 0) {
#pragma unroll
        for (int k = 1; k < SL; ++k) {
          const float4 v = red[k][lane];
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
      }
    }
    if (sl == 0 && c4 < C4) {
      if (part) {
        part[((long long)blockIdx.y * gridDim.x + pix) * C4 + c4] = acc;
      } else {
        if (accumulate) { const float4 p = dx[pix * C4 + c4]; acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w; }
        dx[pix * C4 + c4] = acc;
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(acc.x), fabsf(acc.y)), fmaxf(fabsf(acc.z), fabsf(acc.w))));
      }
    }
    if (SL > 1) __syncthreads();
  }
  if (absmax && !part && sl == 0) {       // (kernel-uniform pointers; slice 0 = one wave holds every stored value)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    unsigned *o = absmax + ((unsigned)pix & 63u);
    if (lane == 0 && mx > 0.f && __float_as_uint(mx) > __builtin_nontemporal_load(o)) atomicMax(o, __float_as_uint(mx));
  }
  if (TH) {       // lanes by a fixed butterfly in double, then the slices in order
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      double v = (double)ta[k];
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
      if (lane == 0) tred[sl][k] = v;
    }
    __syncthreads();
    if (sl == 0 && lane < 6) {
      double v = tred[0][lane];
      for (int q = 1; q < SL; ++q) v += tred[q][lane];
      tpart[(part ? pix * gridDim.y + blockIdx.y : pix) * 6 + lane] = v;
    }
  }
}

// Round 6: the same sums with the GEOMETRY taken once per source position instead of once per source pixel of every image.
// The pre-image of a source pixel under the (batch-shared) grid does not depend on the image: one workgroup per position (h, w)
// lists the target pixels that touch it -- wave 0, one lane per candidate of the bounding box, compacted in the order the
// kernel above visits them: slice by slice (rows ho_lo + s, + SLE, ...), rows ascending, columns ascending -- with their
// bilinear weights and theta factors, in LDS; then every wave walks its share of the batch with lanes over channels: the
// rows of a group of four matches are requested together (the kernel above waits for each match's row before it asks for
// the next: a 64 x 64 map took 0.32 ms for 90 MB), the sums run in the listed order -- per slice, then the slices in order --
// i.e. the same sums in the same order as sampler_bwd_data_kernel<SLE, TH>.  NOT guaranteed the same bits: this file compiles
// with fp contract(fast), and which product of `t0 xt + t1 yt + t2` hipcc fuses depends on the loop around it (measured: the
// coordinates differ in the last place for a generic theta, dx by <= 5e-6 of its largest entry; the identity grid gives the
// same bits).  Both entry points of the data gradient run THIS kernel, so they agree with each other bit for bit.  A box with
// more matches than the list holds (a strongly minifying theta) is walked as before.  At most 64 float4 columns (the host
// routes).
constexpr int kSbCap = 768;        // matches per source position kept in LDS
template <int SLE, bool TH>
__global__ __launch_bounds__(256) void sampler_bwd_data_batched_kernel(const st_t *__restrict__ dy, const float *__restrict__ theta,
                                                                       const A4Ptr dx, int N, int Hin, int Win, int C4, int Ho, int Wo,
                                                                       int ldo, int coff, int accumulate, const st_t *__restrict__ xfwd,
                                                                       double *__restrict__ tpart, unsigned *__restrict__ absmax) {
  __shared__ int s_off[kSbCap];                       // (ho * Wo + wo) * ldo: element offset of the target pixel's row inside an image
  __shared__ float s_w[kSbCap], s_cx[kSbCap], s_cy[kSbCap], s_xt[kSbCap], s_yt[kSbCap];
  __shared__ int s_cnt[SLE + 1];                      // [s]: matches of slices 0 .. s - 1; [SLE]: all, or -1 when the list overflowed
  const float khw = (float)(Win - 1) * 0.5f, khh = (float)(Hin - 1) * 0.5f;
  const Theta th = load_theta(theta);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int w = (int)(blockIdx.x % Win), h = (int)(blockIdx.x / Win);
  // (the bounding box: the statements of sampler_bwd_data_kernel)
  const float kx = Wo > 1 ? 2.f / (float)(Wo - 1) : 0.f, ky = Ho > 1 ? 2.f / (float)(Ho - 1) : 0.f;
  const float hw = (float)(Win - 1) / 2.f, hh = (float)(Hin - 1) / 2.f;
  const float x0t = Wo > 1 ? -1.f : 0.f, y0t = Ho > 1 ? -1.f : 0.f;
  const float axw = th.t[0] * kx * hw, axh = th.t[1] * ky * hw, ax0 = (th.t[0] * x0t + th.t[1] * y0t + th.t[2] + 1.f) * hw;
  const float ayw = th.t[3] * kx * hh, ayh = th.t[4] * ky * hh, ay0 = (th.t[3] * x0t + th.t[4] * y0t + th.t[5] + 1.f) * hh;
  const float det = axw * ayh - axh * ayw;
  int wo_lo = 0, wo_hi = Wo - 1, ho_lo = 0, ho_hi = Ho - 1;
  if (fabsf(det) > 1e-20f && isfinite(det)) {
    float fw_lo = 3e38f, fw_hi = -3e38f, fh_lo = 3e38f, fh_hi = -3e38f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float sx = (float)w + ((k & 1) ? 1.f : -1.f) - ax0, sy = (float)h + ((k & 2) ? 1.f : -1.f) - ay0;
      const float fw = (ayh * sx - axh * sy) / det, fh = (axw * sy - ayw * sx) / det;
      fw_lo = fminf(fw_lo, fw); fw_hi = fmaxf(fw_hi, fw); fh_lo = fminf(fh_lo, fh); fh_hi = fmaxf(fh_hi, fh);
    }
    if (isfinite(fw_lo) && isfinite(fw_hi) && isfinite(fh_lo) && isfinite(fh_hi)) {
      wo_lo = (int)fmaxf(floorf(fw_lo) - 1.f, 0.f); wo_hi = (int)fminf(ceilf(fw_hi) + 1.f, (float)(Wo - 1));
      ho_lo = (int)fmaxf(floorf(fh_lo) - 1.f, 0.f); ho_hi = (int)fminf(ceilf(fh_hi) + 1.f, (float)(Ho - 1));
    }
  }
  // one candidate of the box: does it touch (h, w), and with which weights (the statements of sampler_bwd_data_kernel)
  auto probe = [&](const int ho, const int wo, float &wgt, float &cxk, float &cyk, float &xt, float &yt) __attribute__((always_inline)) {
    yt = tgt_coord(ho, Ho);
    xt = tgt_coord(wo, Wo);
    float xs, ys;
    src_xy(th, xt, yt, Hin, Win, xs, ys);
    const int y0 = (int)floorf(ys), x0 = (int)floorf(xs);
    float wy, wx, sy, sx;
    if (y0 == h) { wy = 1.f - (ys - (float)y0); sy = -1.f; }
    else if (y0 + 1 == h) { wy = 1.f - (1.f - (ys - (float)y0)); sy = 1.f; }
    else return false;
    if (x0 == w) { wx = 1.f - (xs - (float)x0); sx = -1.f; }
    else if (x0 + 1 == w) { wx = 1.f - (1.f - (xs - (float)x0)); sx = 1.f; }
    else return false;
    wgt = wy * wx;
    cxk = wy * sx; cyk = wx * sy;
    return true;
  };
  const int bw = wo_hi - wo_lo + 1;
  if (wv == 0) {
    int cnt = 0;
    bool over = false;
#pragma unroll 1
    for (int sidx = 0; sidx < SLE; ++sidx) {
      if (lane == 0) s_cnt[sidx] = cnt;
      const int nrows = ho_hi >= ho_lo + sidx ? (ho_hi - (ho_lo + sidx)) / SLE + 1 : 0;
      const int ncand = nrows * bw;
#pragma unroll 1
      for (int c0 = 0; c0 < ncand && !over; c0 += 64) {
        const int c = c0 + lane;
        bool hit = false;
        float wgt = 0.f, cxk = 0.f, cyk = 0.f, xt = 0.f, yt = 0.f;
        int ho = 0, wo = 0;
        if (c < ncand) {
          ho = ho_lo + sidx + (c / bw) * SLE; wo = wo_lo + c % bw;
          hit = probe(ho, wo, wgt, cxk, cyk, xt, yt);
        }
        const unsigned long long m = __ballot(hit);
        const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
        const int tot = cnt + __popcll(m);
        if (tot > kSbCap) { over = true; }
        else {
          if (hit) { s_off[pos] = (ho * Wo + wo) * ldo; s_w[pos] = wgt; s_cx[pos] = cxk; s_cy[pos] = cyk; s_xt[pos] = xt; s_yt[pos] = yt; }
          cnt = tot;
        }
      }
    }
    if (lane == 0) s_cnt[SLE] = over ? -1 : cnt;
  }
  __syncthreads();
  const bool listed = s_cnt[SLE] >= 0;
  float mx = 0.f;
  const long long img = (long long)Ho * Wo * ldo;
  const int c4 = lane;
  const bool cval = c4 < C4;
  for (int n = (int)blockIdx.y * 4 + wv; n < N; n += (int)gridDim.y * 4) {
    const long long pix = ((long long)n * Hin + h) * Win + w;
    float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (TH && cval) xv = ld4(xfwd + (pix * C4 + c4) * 4);
    const st_t *base = dy + n * img + coff + (cval ? c4 : 0) * 4;
    float4 total = make_float4(0.f, 0.f, 0.f, 0.f);
    double tv[6] = {0., 0., 0., 0., 0., 0.};
#pragma unroll 1
    for (int sidx = 0; sidx < SLE; ++sidx) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      float ta[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (listed) {
        const int m0 = s_cnt[sidx], m1 = sidx + 1 < SLE ? s_cnt[sidx + 1] : s_cnt[SLE];
#pragma unroll 1
        for (int m = m0; m < m1; m += 4) {
          float4 v[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = (m + q < m1 && cval) ? ld4(base + s_off[m + q]) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (m + q < m1 && cval) {
              const float wgt = s_w[m + q];
              acc.x += wgt * v[q].x; acc.y += wgt * v[q].y; acc.z += wgt * v[q].z; acc.w += wgt * v[q].w;
              if (TH) {
                const float dot = v[q].x * xv.x + v[q].y * xv.y + v[q].z * xv.z + v[q].w * xv.w;
                const float cx = dot * s_cx[m + q] * khw, cy = dot * s_cy[m + q] * khh;
                const float xt = s_xt[m + q], yt = s_yt[m + q];
                ta[0] += cx * xt; ta[1] += cx * yt; ta[2] += cx;
                ta[3] += cy * xt; ta[4] += cy * yt; ta[5] += cy;
              }
            }
          }
        }
      } else {
        for (int ho = ho_lo + sidx; ho <= ho_hi; ho += SLE)
          for (int wo = wo_lo; wo <= wo_hi; ++wo) {
            float wgt, cxk, cyk, xt, yt;
            if (!probe(ho, wo, wgt, cxk, cyk, xt, yt) || !cval) continue;
            const float4 v = ld4(base + (ho * Wo + wo) * ldo);
            acc.x += wgt * v.x; acc.y += wgt * v.y; acc.z += wgt * v.z; acc.w += wgt * v.w;
            if (TH) {
              const float dot = v.x * xv.x + v.y * xv.y + v.z * xv.z + v.w * xv.w;
              const float cx = dot * cxk * khw, cy = dot * cyk * khh;
              ta[0] += cx * xt; ta[1] += cx * yt; ta[2] += cx;
              ta[3] += cy * xt; ta[4] += cy * yt; ta[5] += cy;
            }
          }
      }
      if (sidx == 0) total = acc;
      else { total.x += acc.x; total.y += acc.y; total.z += acc.z; total.w += acc.w; }
      if (TH) {       // the slice's theta sums: lanes by the fixed butterfly in double, then the slices in order
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          double v = (double)ta[k];
#pragma unroll
          for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
          tv[k] = sidx == 0 ? v : tv[k] + v;
        }
      }
    }
    if (cval) {
      if (accumulate) { const float4 p = dx[pix * C4 + c4]; total.x += p.x; total.y += p.y; total.z += p.z; total.w += p.w; }
      dx[pix * C4 + c4] = total;
      mx = fmaxf(mx, fmaxf(fmaxf(fabsf(total.x), fabsf(total.y)), fmaxf(fabsf(total.z), fabsf(total.w))));
    }
    if (TH && lane < 6) {
      double v = tv[0];
#pragma unroll
      for (int k = 1; k < 6; ++k) v = lane == k ? tv[k] : v;
      tpart[pix * 6 + lane] = v;
    }
  }
  if (absmax) {       // (kernel-uniform pointer) one atomic per wave: the magnitude is a maximum over 64 slots, any slot serves
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    unsigned *o = absmax + ((blockIdx.x + (unsigned)wv) & 63u);
    if (lane == 0 && mx > 0.f) atomicMax(o, __float_as_uint(mx));
  }
}

// dx[pixel][C4] (+)= sum over the chunks of part[chunk][pixel][C4], in chunk order; optional magnitude block of the stored dx
__global__ __launch_bounds__(256) void sampler_bwd_reduce_kernel(const float4 *__restrict__ part, const A4Ptr dx, long long n4,
                                                                 int chunks, int accumulate, unsigned *__restrict__ absmax) {
  float mx = 0.f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 a = part[i];
    for (int c = 1; c < chunks; ++c) { const float4 v = part[(long long)c * n4 + i]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    if (accumulate) { const float4 p = dx[i]; a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w; }
    dx[i] = a;
    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
  }
  if (absmax) {
    __shared__ float sm[4];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
      mx = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
      if (mx > 0.f) atomicMax(absmax + (blockIdx.x & 63), __float_as_uint(mx));
    }
  }
}

// first level of the fixed-order sum of tpart[rows][6]: block b adds its contiguous share of the rows (64 row lanes x 6)
__global__ __launch_bounds__(384) void sampler_theta_group_kernel(const double *__restrict__ partial, long long rows,
                                                                  double *__restrict__ out) {
  __shared__ double sm[64][6];
  const int k = threadIdx.x % 6, j = threadIdx.x / 6;
  const long long per = (rows + gridDim.x - 1) / gridDim.x;
  const long long r0 = blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
  double s = 0;
  for (long long r = r0 + j; r < r1; r += 64) s += partial[r * 6 + k];
  sm[j][k] = s;
  __syncthreads();
  if (j == 0) {
    for (int q = 1; q < 64; ++q) s += sm[q][k];
    out[(long long)blockIdx.x * 6 + k] = s;
  }
}

// partial[wave][6] (double): this wave's target pixels, all sources.  Lanes run over channels; every lane keeps its own
// six sums over all the pixels and sources its wave visits (the sum over lanes commutes with the sum over pixels), and
// the lanes are combined ONCE, by a fixed butterfly in double, at the end.
constexpr int kThetaWavesPerBlock = 4;
__global__ __launch_bounds__(64 * kThetaWavesPerBlock) void sampler_bwd_theta_kernel(
    const SrcTable tab, const float *__restrict__ theta, const st_t *__restrict__ dy, double *__restrict__ partial,
    int Ho, int Wo, int ldo, long long pixels) {
  const Theta th = load_theta(theta);
  const int lane = threadIdx.x & 63;
  const long long gw = (long long)blockIdx.x * kThetaWavesPerBlock + (threadIdx.x >> 6);
  const long long nw = (long long)gridDim.x * kThetaWavesPerBlock;
  float a[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (long long p = gw; p < pixels; p += nw) {
    const int wo = (int)(p % Wo);
    const int ho = (int)((p / Wo) % Ho);
    const long long n = p / ((long long)Wo * Ho);
    const float xt = tgt_coord(wo, Wo), yt = tgt_coord(ho, Ho);
    const st_t *g = dy + p * (long long)ldo;
    for (int si = 0; si < tab.n; ++si) {
      const Src s = tab.s[si];
      float xs, ys;
      src_xy(th, xt, yt, s.Hin, s.Win, xs, ys);
      const int y0 = (int)floorf(ys), x0 = (int)floorf(xs);
      const float fy = ys - (float)y0, fx = xs - (float)x0;
      const bool vy0 = (unsigned)y0 < (unsigned)s.Hin, vy1 = (unsigned)(y0 + 1) < (unsigned)s.Hin;
      const bool vx0 = (unsigned)x0 < (unsigned)s.Win, vx1 = (unsigned)(x0 + 1) < (unsigned)s.Win;
      if (!((vy0 || vy1) && (vx0 || vx1))) continue;
      const CA4Ptr x(s.x);
      const long long base = ((n * s.Hin + y0) * s.Win + x0) * s.C4;
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      float gx = 0.f, gy = 0.f;
      for (int c = lane; c < s.C4; c += 64) {
        const float4 v00 = (vy0 && vx0) ? x[base + c] : z;
        const float4 v01 = (vy0 && vx1) ? x[base + s.C4 + c] : z;
        const float4 v10 = (vy1 && vx0) ? x[base + (long long)s.Win * s.C4 + c] : z;
        const float4 v11 = (vy1 && vx1) ? x[base + (long long)(s.Win + 1) * s.C4 + c] : z;
        const float4 d = ld4(g + (s.coff4 + c) * 4);
        const float ax_[4] = {v01.x - v00.x, v01.y - v00.y, v01.z - v00.z, v01.w - v00.w};
        const float bx_[4] = {v11.x - v10.x, v11.y - v10.y, v11.z - v10.z, v11.w - v10.w};
        const float ay_[4] = {v10.x - v00.x, v10.y - v00.y, v10.z - v00.z, v10.w - v00.w};
        const float by_[4] = {v11.x - v01.x, v11.y - v01.y, v11.z - v01.z, v11.w - v01.w};
        const float dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          gx += dd[e] * ((1.f - fy) * ax_[e] + fy * bx_[e]);
          gy += dd[e] * ((1.f - fx) * ay_[e] + fx * by_[e]);
        }
      }
      const float cx = gx * ((float)(s.Win - 1) * 0.5f), cy = gy * ((float)(s.Hin - 1) * 0.5f);
      a[0] += cx * xt; a[1] += cx * yt; a[2] += cx;
      a[3] += cy * xt; a[4] += cy * yt; a[5] += cy;
    }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    double v = (double)a[k];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);     // fixed butterfly over the 64 lanes
    if (lane == 0) partial[gw * 6 + k] = v;
  }
}

__global__ __launch_bounds__(384) void sampler_theta_final_kernel(const double *__restrict__ partial, long long rows,
                                                                  float *__restrict__ dtheta, int accumulate) {
  __shared__ double sm[64][6];
  const int k = threadIdx.x % 6, j = threadIdx.x / 6;      // 64 row groups x 6 components, fixed assignment
  double s = 0;
  for (long long r = j; r < rows; r += 64) s += partial[r * 6 + k];
  sm[j][k] = s;
  __syncthreads();
  if (j == 0) {
    for (int q = 1; q < 64; ++q) s += sm[q][k];
    dtheta[k] = (float)(accumulate ? (double)dtheta[k] + s : s);
  }
}

int theta_blocks(long long pixels) {
  return (int)std::max<long long>(1, std::min<long long>((pixels + 4 * kThetaWavesPerBlock - 1) / (4 * kThetaWavesPerBlock), 1024));
}

int make_table(SrcTable &tab, const st_t *const *x, const int *Hin, const int *Win, const int *C, const int *coff,
               int nsrc, int ldo, const char *what) {
  DSPN_REQUIRE(x && Hin && Win && C && coff && nsrc >= 1 && nsrc <= kMaxSrc, "%s: 1..%d sources", what, kMaxSrc);
  tab.n = nsrc;
  for (int i = 0; i < nsrc; ++i) {
    DSPN_REQUIRE(x[i] && Hin[i] > 0 && Win[i] > 0 && C[i] > 0 && C[i] % 4 == 0 && coff[i] >= 0 && coff[i] % 4 == 0 &&
                     coff[i] + C[i] <= ldo, "%s: bad source %d", what, i);
    tab.s[i] = Src{x[i], Hin[i], Win[i], C[i] / 4, coff[i] / 4};
  }
  return 0;
}

}  // namespace

extern "C" {

int DSPN_FN(dspn_affine_sampler_forward)(const st_t *const *x, const int *Hin, const int *Win, const int *C, const int *coff,
                                    int nsrc, const float *theta, st_t *y, int N, int Ho, int Wo, int ldo, void *stream) {
  DSPN_REQUIRE(theta && y && N > 0 && Ho > 0 && Wo > 0 && ldo > 0 && ldo % 4 == 0, "affine_sampler_forward: bad argument");
  SrcTable tab;
  if (int rc = make_table(tab, x, Hin, Win, C, coff, nsrc, ldo, "affine_sampler_forward")) return rc;
  const long long total = (long long)N * Ho * Wo * (ldo / 4);
  const int blocks = (int)std::max<long long>(1, std::min<long long>((total + 255) / 256, 16384));
  hipLaunchKernelGGL(sampler_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, tab, theta,
                     A4Ptr(y), Ho, Wo, ldo / 4, total);
  return dspn::check_launch("affine_sampler_forward");
}

int DSPN_FN(dspn_affine_sampler_backward_data)(const st_t *dy, const float *theta, st_t *dx, int N, int Hin, int Win, int C,
                                          int Ho, int Wo, int ldo, int coff, int accumulate, void *stream) {
  DSPN_REQUIRE(dy && theta && dx && N > 0 && Hin > 0 && Win > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 4 == 0 &&
                   ldo % 4 == 0 && coff >= 0 && coff % 4 == 0 && coff + C <= ldo, "affine_sampler_backward_data: bad argument");
  const long long pix = (long long)N * Hin * Win;
  DSPN_REQUIRE(pix < (1ll << 31), "affine_sampler_backward_data: too many source pixels");
  // slices by the nominal footprint (the grid is near the identity map): rows of the pre-image box per source pixel
  const int rows = 2 * ((Ho + Hin - 1) / Hin) + 2;
  hipStream_t s = (hipStream_t)stream;
  if (dspn::sampler_batched() && rows < 32 && C / 4 <= 64) {      // (as dspn_affine_sampler_backward_data_theta below: the same kernel)
    const long long pos = (long long)Hin * Win;
    const int gy = (int)std::max<long long>(1, std::min<long long>((N + 3) / 4, (2048 + pos - 1) / pos));
#define DSPN_SBB_(SL) hipLaunchKernelGGL((sampler_bwd_data_batched_kernel<SL, false>), dim3((unsigned)pos, gy), dim3(256), 0, s, dy, theta, \
                                         A4Ptr(dx), N, Hin, Win, C / 4, Ho, Wo, ldo, coff, accumulate, nullptr, nullptr, nullptr)
    if (rows >= 10) DSPN_SBB_(4);
    else DSPN_SBB_(1);
#undef DSPN_SBB_
    return dspn::check_launch("affine_sampler_backward_data");
  }
#define DSPN_SBD_(SL) hipLaunchKernelGGL((sampler_bwd_data_kernel<SL, false>), dim3((unsigned)pix), dim3(64, SL), 0, s, dy, theta, \
                                         A4Ptr(dx), Hin, Win, C / 4, Ho, Wo, ldo, coff, accumulate, nullptr, nullptr, nullptr, nullptr)
  if (rows >= 32) DSPN_SBD_(16);
  else if (rows >= 10) DSPN_SBD_(4);
  else DSPN_SBD_(1);
#undef DSPN_SBD_
  return dspn::check_launch("affine_sampler_backward_data");
}

// chunks of the pre-image rows per source pixel (1: one workgroup per pixel does it all)
static int sampler_batched() { return dspn::sampler_batched(); }
static int sampler_chunks(int N, int Hin, int Win, int Ho) {
  const int rows = 2 * ((Ho + Hin - 1) / Hin) + 2;
  return (rows >= 32 && (long long)N * Hin * Win <= 4096) ? 8 : 1;
}
#ifndef DSPN_HALF
long long dspn_affine_sampler_theta_rows(int N, int Hin, int Win, int Ho) {
  if (N <= 0 || Hin <= 0 || Win <= 0 || Ho <= 0) return 0;
  return (long long)N * Hin * Win * sampler_chunks(N, Hin, Win, Ho);
}
size_t dspn_affine_sampler_backward_workspace_bytes(int N, int Hin, int Win, int C, int Ho) {
  if (N <= 0 || Hin <= 0 || Win <= 0 || Ho <= 0 || C <= 0) return 0;
  const int ch = sampler_chunks(N, Hin, Win, Ho);
  return ch > 1 ? sizeof(float) * (size_t)ch * N * Hin * Win * C : 0;
}
#endif

int DSPN_FN(dspn_affine_sampler_backward_data_theta)(const st_t *dy, const float *theta, const st_t *x, st_t *dx, int N, int Hin,
                                                int Win, int C, int Ho, int Wo, int ldo, int coff, int accumulate,
                                                double *theta_partial, size_t theta_partial_bytes, float *dx_absmax,
                                                void *workspace, size_t workspace_bytes, void *stream) {
  DSPN_REQUIRE(dy && theta && x && dx && theta_partial && N > 0 && Hin > 0 && Win > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 4 == 0 &&
                   ldo % 4 == 0 && coff >= 0 && coff % 4 == 0 && coff + C <= ldo, "affine_sampler_backward_data_theta: bad argument");
  DSPN_REQUIRE(!accumulate || x != dx, "affine_sampler_backward_data_theta: x may alias dx only when dx is overwritten");
  const long long pix = (long long)N * Hin * Win;
  DSPN_REQUIRE(pix < (1ll << 31), "affine_sampler_backward_data_theta: too many source pixels");
  const int chunks = dspn::kHalf ? 1 : sampler_chunks(N, Hin, Win, Ho);
  if (theta_partial_bytes < sizeof(double) * 6 * (size_t)pix * chunks)
    return dspn::fail(DSPN_ERR_WORKSPACE_, "affine_sampler_backward_data_theta: theta_partial holds dspn_affine_sampler_theta_rows() rows of 6 doubles");
  const int rows = 2 * ((Ho + Hin - 1) / Hin) + 2;
  hipStream_t s = (hipStream_t)stream;
  if (chunks > 1) {       // small source map: the pre-image rows of every pixel over `chunks` workgroups, then one reduce
    if (!workspace || workspace_bytes < sizeof(float) * (size_t)chunks * pix * C)
      return dspn::fail(DSPN_ERR_WORKSPACE_, "affine_sampler_backward_data_theta: workspace < dspn_affine_sampler_backward_workspace_bytes()");
    float4 *part = static_cast<float4 *>(workspace);
    hipLaunchKernelGGL((sampler_bwd_data_kernel<4, true>), dim3((unsigned)pix, chunks), dim3(64, 4), 0, s, dy, theta, A4Ptr(dx), Hin, Win,
                       C / 4, Ho, Wo, ldo, coff, accumulate, x, theta_partial, nullptr, part);
    const long long n4 = pix * (C / 4);
    hipLaunchKernelGGL(sampler_bwd_reduce_kernel, dim3((unsigned)std::min<long long>((n4 + 255) / 256, 2048)), dim3(256), 0, s, part,
                       A4Ptr(dx), n4, chunks, accumulate, reinterpret_cast<unsigned *>(dx_absmax));
    return dspn::check_launch("affine_sampler_backward_data_theta");
  }
#define DSPN_SBD_(SL) hipLaunchKernelGGL((sampler_bwd_data_kernel<SL, true>), dim3((unsigned)pix), dim3(64, SL), 0, s, dy, theta, \
                                         A4Ptr(dx), Hin, Win, C / 4, Ho, Wo, ldo, coff, accumulate, x, theta_partial, \
                                         dspn::kHalf ? nullptr : reinterpret_cast<unsigned *>(dx_absmax), nullptr)
  // round 6: one workgroup per source POSITION for the whole batch (the geometry once, the rows of four matches in flight):
  // same bits; dspn_affine_sampler_set_batched(0) keeps the kernel above (tests)
  if (sampler_batched() && rows < 32 && C / 4 <= 64) {
    const long long pos = (long long)Hin * Win;
    const int gy = (int)std::max<long long>(1, std::min<long long>((N + 3) / 4, (2048 + pos - 1) / pos));
#define DSPN_SBB_(SL) hipLaunchKernelGGL((sampler_bwd_data_batched_kernel<SL, true>), dim3((unsigned)pos, gy), dim3(256), 0, s, dy, theta, \
                                         A4Ptr(dx), N, Hin, Win, C / 4, Ho, Wo, ldo, coff, accumulate, x, theta_partial, \
                                         dspn::kHalf ? nullptr : reinterpret_cast<unsigned *>(dx_absmax))
    if (rows >= 10) DSPN_SBB_(4);
    else DSPN_SBB_(1);
#undef DSPN_SBB_
    return dspn::check_launch("affine_sampler_backward_data_theta");
  }
  if (rows >= 32) DSPN_SBD_(16);
  else if (rows >= 10) DSPN_SBD_(4);
  else DSPN_SBD_(1);
#undef DSPN_SBD_
  return dspn::check_launch("affine_sampler_backward_data_theta");
}

#ifndef DSPN_HALF
/* fixed-order sum of theta_partial[rows][6] (the rows of every source of one sampler, concatenated) -> dtheta[6] */
size_t dspn_affine_sampler_theta_reduce_workspace_bytes(long long rows) { return rows > 0 ? sizeof(double) * 6 * 256 : 0; }
int dspn_affine_sampler_theta_reduce(const double *theta_partial, long long rows, float *dtheta, int accumulate, void *workspace,
                                     size_t workspace_bytes, void *stream) {
  DSPN_REQUIRE(theta_partial && dtheta && workspace && rows > 0, "affine_sampler_theta_reduce: bad argument");
  if (workspace_bytes < dspn_affine_sampler_theta_reduce_workspace_bytes(rows))
    return dspn::fail(DSPN_ERR_WORKSPACE_, "affine_sampler_theta_reduce: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int groups = (int)std::min<long long>(256, (rows + 255) / 256);
  hipLaunchKernelGGL(sampler_theta_group_kernel, dim3(groups), dim3(384), 0, s, theta_partial, rows, static_cast<double *>(workspace));
  hipLaunchKernelGGL(sampler_theta_final_kernel, dim3(1), dim3(384), 0, s, static_cast<const double *>(workspace), (long long)groups,
                     dtheta, accumulate);
  return dspn::check_launch("affine_sampler_theta_reduce");
}
#endif

static size_t theta_workspace_bytes(int N, int Ho, int Wo) {
  if (N <= 0 || Ho <= 0 || Wo <= 0) return 0;
  return sizeof(double) * 6 * (size_t)theta_blocks((long long)N * Ho * Wo) * kThetaWavesPerBlock;
}
#ifndef DSPN_HALF
size_t dspn_affine_sampler_theta_workspace_bytes(int N, int Ho, int Wo) { return theta_workspace_bytes(N, Ho, Wo); }
#endif

int DSPN_FN(dspn_affine_sampler_backward_theta)(const st_t *const *x, const int *Hin, const int *Win, const int *C,
                                           const int *coff, int nsrc, const float *theta, const st_t *dy, int N, int Ho,
                                           int Wo, int ldo, float *dtheta, int accumulate, void *workspace,
                                           size_t workspace_bytes, void *stream) {
  DSPN_REQUIRE(theta && dy && dtheta && workspace && N > 0 && Ho > 0 && Wo > 0 && ldo > 0 && ldo % 4 == 0,
               "affine_sampler_backward_theta: bad argument");
  SrcTable tab;
  if (int rc = make_table(tab, x, Hin, Win, C, coff, nsrc, ldo, "affine_sampler_backward_theta")) return rc;
  if (workspace_bytes < theta_workspace_bytes(N, Ho, Wo))
    return dspn::fail(DSPN_ERR_WORKSPACE_, "affine_sampler_backward_theta: workspace too small");
  const long long pixels = (long long)N * Ho * Wo;
  const int blocks = theta_blocks(pixels);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sampler_bwd_theta_kernel, dim3(blocks), dim3(64 * kThetaWavesPerBlock), 0, s, tab, theta, dy,
                     static_cast<double *>(workspace), Ho, Wo, ldo, pixels);
  hipLaunchKernelGGL(sampler_theta_final_kernel, dim3(1), dim3(384), 0, s, static_cast<const double *>(workspace),
                     (long long)blocks * kThetaWavesPerBlock, dtheta, accumulate);
  return dspn::check_launch("affine_sampler_backward_theta");
}

}  // extern "C"
