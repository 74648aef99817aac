// HBM-bound kernels of the DSPNet multi-task step on MI355X (gfx950): batch-stat
// BatchNorm (+ReLU) forward/backward, pooling, bilinear sampler, losses, SGD and
// layout helpers.  C ABI in include/dspn_nn.h.
//
// All tensors are NHWC fp32 with a physical channel count that is a multiple of
// 4, so every streaming access is a 16-byte-per-lane float4 with consecutive
// lanes on consecutive addresses.  Reductions are two-stage (per-slab partials
// in a caller workspace, then a fixed-order finalize in double), never float
// atomics, so every result is bitwise reproducible run to run.
#include "dspn_common.h"
#include "bn_final_job.h"
#include "dspn_store.h"
#include "dspn_pieces.h"
#include "../../include/dspn_nn.h"

// Compiled twice (dspn_store.h): float activations -> `*_f32`, and through nn_h.hip with DSPN_HALF -> bfloat16
// activations -> `*_bf16` (element proxies CA4Ptr / A4Ptr / CA1Ptr / A1Ptr widen on load and round on store; all
// arithmetic and every reduction stays fp32 / double).  Entry points that never touch an activation tensor exist once.
using dspn::st_t;
using dspn::A1Ptr;
using dspn::A4Ptr;
using dspn::CA1Ptr;
using dspn::CA4Ptr;

#pragma clang fp contract(fast)

namespace {

constexpr int kT = 256;
// rows per slab of the two-stage reductions: 512 on the large tensors; small ones (inceptionv3 at batch 8: 15 k rows)
// are cut finer so that a reduction still spreads over ~1000 workgroups instead of 29 (2.5 ms per step went there)
inline int slab_rows_for(long long rows) {
  const long long want = (rows + 1023) / 1024;
  return (int)std::min<long long>(512, std::max<long long>(32, (want + 7) / 8 * 8));
}

inline int grid_for(long long n, int per_block = kT, int cap = 8192) {
  long long b = (n + per_block - 1) / per_block;
  return (int)std::max<long long>(1, std::min<long long>(b, cap));
}

// grid of a chunked streaming kernel (U x kT consecutive elements per workgroup and step) over n elements of `groups`
// channel groups per row; *fixed: kT % groups == 0, i.e. every element of a thread belongs to one channel group; *u4: the
// tensor is large enough for four elements per thread (>= 2048 workgroups either way)
inline int grid_fixed_channel(long long n, int groups, int *fixed, int *u4) {
  *fixed = groups > 0 && kT % groups == 0;
  *u4 = n >= 4ll * kT * 2048;
  return grid_for(n, (*u4 ? 4 : 1) * kT, 4096);
}

// bias gradients are column sums of small, narrow tensors (20..54 channels): slabs of 64 rows keep a few
// hundred workgroups busy instead of rows/512
inline int colsum_slab_rows(long long rows) { return (int)std::max<long long>(64, (rows + 4095) / 4096); }

// ------------------------------------------------------------------ BN statistics
// partial[slab][0][c] = sum (x - K[c]), partial[slab][1][c] = sum (x - K[c])^2, K = row 0
__global__ __launch_bounds__(kT) void bn_stats_partial_kernel(const CA4Ptr x,
                                                              long long rows, int C4, int CL,
                                                              float *__restrict__ partial, int kSlabRows) {
  extern __shared__ __attribute__((aligned(16))) float4 sm4[];
  const int RL = kT / CL;
  const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
  const int c4 = blockIdx.y * CL + cl;
  const long long r0 = (long long)blockIdx.x * kSlabRows;
  const long long r1 = min(rows, r0 + kSlabRows);
  float4 s = make_float4(0, 0, 0, 0), ss = make_float4(0, 0, 0, 0);
  const bool act = rl < RL && c4 < C4;
  if (act) {
    const float4 K = x[c4];
#pragma unroll 4
    for (long long r = r0 + rl; r < r1; r += RL) {
      const float4 v = x[r * C4 + c4];
      const float a = v.x - K.x, b = v.y - K.y, c = v.z - K.z, d = v.w - K.w;
      s.x += a; s.y += b; s.z += c; s.w += d;
      ss.x += a * a; ss.y += b * b; ss.z += c * c; ss.w += d * d;
    }
  }
  float4 *s_s = sm4, *s_ss = sm4 + kT;
  s_s[threadIdx.x] = s; s_ss[threadIdx.x] = ss;
  __syncthreads();
  if (rl == 0 && c4 < C4) {
    for (int k = 1; k < RL; ++k) {
      const float4 a = s_s[k * CL + cl], b = s_ss[k * CL + cl];
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
      ss.x += b.x; ss.y += b.y; ss.z += b.z; ss.w += b.w;
    }
    float4 *p = reinterpret_cast<float4 *>(partial) + (long long)blockIdx.x * 2 * C4;
    p[c4] = s; p[C4 + c4] = ss;
  }
}

// 64 channels x 16 slab-lanes per block; also folds gamma/beta into scale/shift for the apply pass
__global__ __launch_bounds__(1024) void bn_stats_final_kernel(
    const CA1Ptr x, const float *__restrict__ partial, int nslabs, long long rows, int C,
    float eps, const float *__restrict__ gamma, const float *__restrict__ beta,
    float *__restrict__ mean, float *__restrict__ rstd, float *__restrict__ scale,
    float *__restrict__ shift) {
  __shared__ double s_S[16][64];
  __shared__ double s_SS[16][64];
  const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double S = 0, SS = 0;
  if (c < C) {
#pragma unroll 8
    for (int k = sl; k < nslabs; k += 16) {   // unrolled: the kernel is pure load latency
      S += partial[(long long)k * 2 * C + c];
      SS += partial[(long long)k * 2 * C + C + c];
    }
  }
  s_S[sl][cl] = S; s_SS[sl][cl] = SS;
  __syncthreads();
  if (sl == 0 && c < C) {
    for (int k = 1; k < 16; ++k) { S += s_S[k][cl]; SS += s_SS[k][cl]; }
    const double n = (double)rows;
    const double m = S / n;
    double var = SS / n - m * m;
    if (var < 0) var = 0;
    const float mu = (float)((double)x[c] + m);
    const float rs = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = mu; rstd[c] = rs;
    const float sc = (gamma ? gamma[c] : 1.f) * rs;
    scale[c] = sc;
    shift[c] = beta[c] - mu * sc;
  }
}

// merge of per-tile (mean, M2) pairs, 16 channels x 64 tile-lanes per block, ONE sweep in double, fixed order:
//   mean = A / rows, var = B / rows - mean^2  with  A = sum_t n_t mean_t,  B = sum_t (M2_t + n_t mean_t^2)
// (the cancellation in B/rows - mean^2 is harmless in double: the per-tile quantities carry 24 bits)
// The loop is unrolled so that 8 loads per accumulator are in flight: the kernel sits alone between two
// convolutions and is pure load latency.
__global__ __launch_bounds__(1024) void bn_stats_tiles_final_kernel(
    const float *__restrict__ ts, int tiles, int tile_rows, long long rows, int C, float eps,
    const float *__restrict__ gamma, const float *__restrict__ beta, float *__restrict__ mean,
    float *__restrict__ rstd, float *__restrict__ scale, float *__restrict__ shift,
    const float *__restrict__ mm, int mm_tiles, int relu, unsigned *__restrict__ absmax, unsigned *__restrict__ absmin,
    float *__restrict__ chan_minmax) {
  __shared__ double sA[64][17], sB[64][17];
  __shared__ float sLo[64][17], sHi[64][17];
  const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const long long last_n = rows - (long long)(tiles - 1) * tile_rows;
  double A = 0, B = 0;
  if (c < C) {
#pragma unroll 8
    for (int t = sl; t < tiles; t += 64) {
      const double n = (double)(t == tiles - 1 ? last_n : (long long)tile_rows);
      const double mt = ts[((long long)t * 2 + 0) * C + c], qt = ts[((long long)t * 2 + 1) * C + c];
      A += n * mt;
      B += qt + n * mt * mt;
    }
  }
  // (round 4: the four tile lanes a wave holds per channel -- sl & 3, lanes 16 apart -- are added by shuffles first, in a fixed
  // order, so the serial tail below walks 16 LDS entries per channel instead of 64: ~2 us off a 6.5-us kernel, 56 launches)
  A += __shfl_xor(A, 16, 64); B += __shfl_xor(B, 16, 64);
  A += __shfl_xor(A, 32, 64); B += __shfl_xor(B, 32, 64);
  sA[sl][cl] = A; sB[sl][cl] = B;
  if (mm) {      // the extremes of the tensor, per channel, from the (min, max) pairs its producer wrote beside the statistics
    float lo = INFINITY, hi = -INFINITY;
    if (c < C) {
#pragma unroll 8
      for (int t = sl; t < mm_tiles; t += 64) {
        lo = fminf(lo, mm[((long long)t * 2 + 0) * C + c]);
        hi = fmaxf(hi, mm[((long long)t * 2 + 1) * C + c]);
      }
    }
    lo = fminf(lo, __shfl_xor(lo, 16, 64)); hi = fmaxf(hi, __shfl_xor(hi, 16, 64));
    lo = fminf(lo, __shfl_xor(lo, 32, 64)); hi = fmaxf(hi, __shfl_xor(hi, 32, 64));
    sLo[sl][cl] = lo; sHi[sl][cl] = hi;
  }
  __syncthreads();
  if (sl == 0 && c < C) {
    for (int k = 4; k < 64; k += 4) { A += sA[k][cl]; B += sB[k][cl]; }      // (entries k .. k + 3 hold the same wave sum)
    const double m = A / (double)rows;
    double var = B / (double)rows - m * m;
    if (var < 0) var = 0;
    const float mu = (float)m;
    const float rs = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = mu; rstd[c] = rs;
    const float sc = (gamma ? gamma[c] : 1.f) * rs;
    const float sh = beta[c] - mu * sc;
    scale[c] = sc;
    shift[c] = sh;
    if (mm) {    // largest |(relu)(x * scale + shift)|: the affine is monotone per channel, so it sits at an extreme of x
      float lo = sLo[0][cl], hi = sHi[0][cl];
      for (int k = 4; k < 64; k += 4) { lo = fminf(lo, sLo[k][cl]); hi = fmaxf(hi, sHi[k][cl]); }
      if (chan_minmax) { chan_minmax[c] = lo; chan_minmax[C + c] = hi; }     // the tensor's extremes per channel (BatchNorm backward's bound)
      float a = fmaf(lo, sc, sh), b = fmaf(hi, sc, sh);        // the same fmaf as the loaders that apply this affine
      if (relu) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); }
      const float v = fmaxf(fabsf(a), fabsf(b));
      if (v == v) atomicMax(absmax + (c & 63), __float_as_uint(v));    // non-negative floats order as their bit patterns
      // range monitor (optional): the SMALLEST non-zero per-channel magnitude of the same tensor.  A tensor whose channels
      // span more than 2^17 leaves the two-piece math's relative-accuracy window for its small channels (include/dspn_nn.h);
      // the caller compares the two blocks after the step (Graph.range_report)
      if (absmin && v > 0.f && v < INFINITY) atomicMin(absmin, __float_as_uint(v));
    }
  }
}

// Pre-reduction of a long tile table (thousands of row tiles on the 128x128-pixel stages): groups of kTileGroup
// consecutive tiles are merged into one entry of the SAME format, by hundreds of workgroups, so that the
// single-workgroup-per-16-channels finalize kernels below read 32x fewer entries.
//   MODE 0: (mean, M2) pairs  -> (mean_g, M2_g) by the one-sweep double formula;  MODE 1: plain sums
constexpr int kTileGroup = 32;
// tables at least this long are pre-reduced (DSPN_TILE_GROUP_MIN: timing experiments)
static int tile_group_min() {
  static const int v = [] { const char *e = getenv("DSPN_TILE_GROUP_MIN"); return e ? atoi(e) : 1024; }();
  return v;
}
// mm / mm_out (MODE 0, optional): the per-tile (min, max) table of the same tiles, merged into one pair per group
template <int MODE>
__global__ __launch_bounds__(256) void tile_group_kernel(const float *__restrict__ ts, int tiles, int tile_rows,
                                                         long long rows, int C, float *__restrict__ out,
                                                         const float *__restrict__ mm, float *__restrict__ mm_out) {
  __shared__ double sA[4][64], sB[4][64];
  __shared__ float sLo[4][64], sHi[4][64];
  const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + cl;
  const int t0 = blockIdx.x * kTileGroup, t1 = min(tiles, t0 + kTileGroup);
  const long long last_n = rows - (long long)(tiles - 1) * tile_rows;
  double A = 0, B = 0;
  if (c < C) {
#pragma unroll 8
    for (int t = t0 + sl; t < t1; t += 4) {
      const double a = ts[((long long)t * 2 + 0) * C + c], b = ts[((long long)t * 2 + 1) * C + c];
      if (MODE == 0) {
        const double nt = (double)(t == tiles - 1 ? last_n : (long long)tile_rows);
        A += nt * a; B += b + nt * a * a;
      } else {
        A += a; B += b;
      }
    }
  }
  sA[sl][cl] = A; sB[sl][cl] = B;
  if (MODE == 0 && mm) {
    float lo = INFINITY, hi = -INFINITY;
    if (c < C) {
#pragma unroll 8
      for (int t = t0 + sl; t < t1; t += 4) {
        lo = fminf(lo, mm[((long long)t * 2 + 0) * C + c]);
        hi = fmaxf(hi, mm[((long long)t * 2 + 1) * C + c]);
      }
    }
    sLo[sl][cl] = lo; sHi[sl][cl] = hi;
  }
  __syncthreads();
  if (MODE == 0 && mm && sl == 0 && c < C) {
    float lo = sLo[0][cl], hi = sHi[0][cl];
    for (int k = 1; k < 4; ++k) { lo = fminf(lo, sLo[k][cl]); hi = fmaxf(hi, sHi[k][cl]); }
    mm_out[((long long)blockIdx.x * 2 + 0) * C + c] = lo;
    mm_out[((long long)blockIdx.x * 2 + 1) * C + c] = hi;
  }
  if (sl == 0 && c < C) {
    for (int k = 1; k < 4; ++k) { A += sA[k][cl]; B += sB[k][cl]; }
    if (MODE == 0) {
      // rows of the whole group (every lane saw a different subset: recompute)
      double ng = 0;
      for (int t = t0; t < t1; ++t) ng += (double)(t == tiles - 1 ? last_n : (long long)tile_rows);
      const double m = A / ng;
      out[((long long)blockIdx.x * 2 + 0) * C + c] = (float)m;
      out[((long long)blockIdx.x * 2 + 1) * C + c] = (float)fmax(B - ng * m * m, 0.0);
    } else {
      out[((long long)blockIdx.x * 2 + 0) * C + c] = (float)A;
      out[((long long)blockIdx.x * 2 + 1) * C + c] = (float)B;
    }
  }
}

// absmax (optional, float tensors): 64 partial maxima of |y| as stored -- the magnitude block of the tensor for the
// convolutions that multiply it in the two-piece fp16 math (dspn_absmax_f32), taken here instead of by a pass of its own
__global__ __launch_bounds__(kT) void bn_apply_kernel(const CA4Ptr x, const float4 *__restrict__ scale,
                                const float4 *__restrict__ shift, const A4Ptr y,
                                long long n4, int C4, int relu, unsigned *__restrict__ absmax) {
  float mx = 0.f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4;
       i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    const float4 v = x[i], a = scale[c4], b = shift[c4];
    float4 o = make_float4(fmaf(v.x, a.x, b.x), fmaf(v.y, a.y, b.y), fmaf(v.z, a.z, b.z), fmaf(v.w, a.w, b.w));
    if (relu) {
      o.x = o.x > 0.f ? o.x : 0.f; o.y = o.y > 0.f ? o.y : 0.f;
      o.z = o.z > 0.f ? o.z : 0.f; o.w = o.w > 0.f ? o.w : 0.f;
    }
    y[i] = o;
    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
  }
  if (absmax) {          // (kernel-uniform)
    __shared__ float sm[kT / 64];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < kT / 64; ++w) mx = fmaxf(mx, sm[w]);
      unsigned *o = absmax + (blockIdx.x & 63);
      if (mx > 0.f && __float_as_uint(mx) > __builtin_nontemporal_load(o)) atomicMax(o, __float_as_uint(mx));
    }
  }
}

// (relu)(x * scale[c] + shift[c]) written as fp16 PIECE PLANES (the layout of bn_bwd_apply_planes_kernel below), cut by
// the scale of `block` -- the magnitude the statistics finalize formed from the producer's per-channel extremes BEFORE this
// pass (dspn_bn_stats_from_tiles_f32 out_absmax).  For a BatchNorm in front of a multi-tap convolution: that convolution's
// forward and weight gradient copy the records (conv_nt_kernel EPIX & 4, conv_wgrad_kernel MATHX = 5 / 6) instead of
// applying the affine and cutting every element once per (tap, column tile).
// A thread OWNS one 16-byte chunk of the planes: index i = float4 index of the tensor = 16-byte chunk index of the planes;
// chunk l = i & 7 of its 128-byte record holds piece l >> 2 of channels 8 (l & 3) .. + 7, so the thread reads the two float4
// of those eight channels (each is read by two threads -- the second read is a cache hit), forms both pieces and stores
// the one it owns: eight lanes write one whole 128-byte line per store instruction (the first form of this kernel wrote two
// 8-byte halves per float4, 64-byte runs from two instructions per line: 3.7 TB/s).  Streaming form of bn_bwd_apply_kernel
// below: a grid whose stride is a multiple of C4 (`fixed_c`) gives a thread ONE chunk position, and U chunks' inputs are
// requested before the first is used.
struct PlaneChunk { int ca; bool hi; };        // float4 index (within the pixel) of the chunk's first four channels; which piece
__device__ __forceinline__ PlaneChunk plane_chunk(const int c) {   // c = i % C4
  return PlaneChunk{(c & ~7) + 2 * (c & 3), (c & 4) != 0};
}
__device__ __forceinline__ uint4 plane_pack(const dspn::pieces::bf16x4 a, const dspn::pieces::bf16x4 b) {
  const uint2 x = __builtin_bit_cast(uint2, a), y = __builtin_bit_cast(uint2, b);
  return make_uint4(x.x, x.y, y.x, y.y);
}
template <int U>
__global__ __launch_bounds__(256) void bn_apply_planes_kernel(const float4 *__restrict__ x, const float4 *__restrict__ scale,
                                const float4 *__restrict__ shift, uint4 *__restrict__ planes,
                                long long n4, int C4, int relu, const float *__restrict__ block, int fixed_c) {
  const float s = dspn::pieces::operand_scale(block);
  const bool nf = dspn::pieces::operand_nonfinite(block);
  auto affine = [&](const float4 v, const float4 a, const float4 b) __attribute__((always_inline)) {
    float4 o = make_float4(fmaf(v.x, a.x, b.x), fmaf(v.y, a.y, b.y), fmaf(v.z, a.z, b.z), fmaf(v.w, a.w, b.w));
    if (relu) {
      o.x = o.x > 0.f ? o.x : 0.f; o.y = o.y > 0.f ? o.y : 0.f;
      o.z = o.z > 0.f ? o.z : 0.f; o.w = o.w > 0.f ? o.w : 0.f;
    }
    return o;
  };
  auto one = [&](const long long i, const bool hi, const float4 v0, const float4 v1, const float4 a0, const float4 b0,
                 const float4 a1, const float4 b1) __attribute__((always_inline)) {
    dspn::pieces::bf16x4 p0, p1, q0, q1;
    dspn::pieces::split2h(affine(v0, a0, b0), s, p0, p1);
    dspn::pieces::split2h(affine(v1, a1, b1), s, q0, q1);
    if (__builtin_expect(nf, 0)) { dspn::pieces::repair_inf(p0, p1); dspn::pieces::repair_inf(q0, q1); }
    planes[i] = hi ? plane_pack(p1, q1) : plane_pack(p0, q0);
  };
  const long long cstride = (long long)gridDim.x * (U * 256);
  long long base = blockIdx.x * (long long)(U * 256) + threadIdx.x;
  if (fixed_c) {
    const int c = (int)(base % C4);
    const PlaneChunk pc = plane_chunk(c);
    const float4 a0 = scale[pc.ca], b0 = shift[pc.ca], a1 = scale[pc.ca + 1], b1 = shift[pc.ca + 1];
    const int d = pc.ca - c;
    for (; base + (U - 1) * 256 < n4; base += cstride) {
      float4 v0[U], v1[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { v0[u] = x[base + u * 256 + d]; v1[u] = x[base + u * 256 + d + 1]; }
#pragma unroll
      for (int u = 0; u < U; ++u) one(base + u * 256, pc.hi, v0[u], v1[u], a0, b0, a1, b1);
    }
    for (int u = 0; u < U; ++u) {
      const long long i = base + u * 256;
      if (i < n4) one(i, pc.hi, x[i + d], x[i + d + 1], a0, b0, a1, b1);
    }
  } else {
    for (; base < n4; base += cstride) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long i = base + u * 256;
        if (i >= n4) break;
        const int c = (int)(i % C4);
        const PlaneChunk pc = plane_chunk(c);
        const long long xa = i - c + pc.ca;
        one(i, pc.hi, x[xa], x[xa + 1], scale[pc.ca], shift[pc.ca], scale[pc.ca + 1], shift[pc.ca + 1]);
      }
    }
  }
}

// a BOUND of the magnitude of (relu)(x * scale[c] + shift[c]) from the magnitude M of x alone: max over c of
// |scale[c]| * M + |shift[c]| -- for a convolution that folds a BatchNorm into its loader and whose raw input has a known
// magnitude but no per-channel extremes (a pooled tensor).  A bound that is too large by less than 2^17 costs the two-piece
// math nothing (include/dspn_nn.h).  One workgroup.
__global__ __launch_bounds__(256) void absmax_affine_bound_kernel(const float *__restrict__ scale, const float *__restrict__ shift,
                                                                  int C, const float *__restrict__ x_absmax,
                                                                  unsigned *__restrict__ out) {
  __shared__ float sm[4];
  float m = x_absmax[threadIdx.x & 63];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  float b = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) b = fmaxf(b, fabsf(scale[c]) * m + fabsf(shift[c]));
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) b = fmaxf(b, __shfl_xor(b, o, 64));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = b;
  __syncthreads();
  if (threadIdx.x == 0) {
    b = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    if (!(b == b)) b = INFINITY;                        // a NaN coefficient: no finite bound exists
    if (b > 0.f) atomicMax(out, __float_as_uint(b));
  }
}

// The output gradient of a BatchNorm whose (ReLU) output feeds a max pooling ALONE (the resnet stem), formed on the fly from
// the pooled gradient and the argmax record instead of read from a dense tensor the pooling backward would first write
// (round 4): the gather of maxpool_bwd_idx_kernel for 4 channels of one input pixel.  argmax == NULL: no pooling.
struct PoolGrad { const uchar4 *argmax; const float4 *dy; int H, W, k, stride, pad, Ho, Wo; };
__device__ __forceinline__ float4 pool_grad_at(const PoolGrad &p, long long row, int c4, int C4) {
  // (rows < 2^31 -- the entry point checks it: 32-bit divisions, a third of the instructions of the 64-bit ones)
  const unsigned r = (unsigned)row;
  const unsigned t = r / (unsigned)p.W;
  const int w = (int)(r - t * (unsigned)p.W);
  const unsigned nn = t / (unsigned)p.H;
  const int h = (int)(t - nn * (unsigned)p.H);
  const long long n = nn;
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  int ho_lo = (h + p.pad - p.k + p.stride) / p.stride; if (h + p.pad - p.k + 1 <= 0) ho_lo = 0;
  int wo_lo = (w + p.pad - p.k + p.stride) / p.stride; if (w + p.pad - p.k + 1 <= 0) wo_lo = 0;
  const int ho_hi = min(p.Ho - 1, (h + p.pad) / p.stride), wo_hi = min(p.Wo - 1, (w + p.pad) / p.stride);
  for (int ho = ho_lo; ho <= ho_hi; ++ho)
    for (int wo = wo_lo; wo <= wo_hi; ++wo) {
      const long long oi = ((n * p.Ho + ho) * p.Wo + wo) * C4 + c4;
      const int pos = (h - (ho * p.stride - p.pad)) * p.k + (w - (wo * p.stride - p.pad));
      const uchar4 a = p.argmax[oi];
      const float4 d = p.dy[oi];
      g.x += a.x == pos ? d.x : 0.f; g.y += a.y == pos ? d.y : 0.f;
      g.z += a.z == pos ? d.z : 0.f; g.w += a.w == pos ? d.w : 0.f;
    }
  return g;
}

// partial[slab][0][c] = sum dy', partial[slab][1][c] = sum dy' * xhat ; dy' = relu ? dy*(y>0) : dy
// The ReLU mask is recomputed as (x*scale + shift > 0) -- the very fma of the forward apply pass --
// so the forward output is never re-read.
template <bool POOL>
__global__ __launch_bounds__(kT) void bn_bwd_partial_kernel(
    const CA4Ptr x, const float4 *__restrict__ scale, const float4 *__restrict__ shift,
    const CA4Ptr dy, const float *__restrict__ mean, const float *__restrict__ rstd,
    long long rows, int C4, int CL, int relu, float *__restrict__ partial, int kSlabRows, const PoolGrad pool) {
  extern __shared__ __attribute__((aligned(16))) float4 sm4[];
  const int RL = kT / CL;
  const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
  const int c4 = blockIdx.y * CL + cl;
  const long long r0 = (long long)blockIdx.x * kSlabRows;
  const long long r1 = min(rows, r0 + kSlabRows);
  float4 s = make_float4(0, 0, 0, 0), ss = make_float4(0, 0, 0, 0);
  if (rl < RL && c4 < C4) {
    const float4 m = reinterpret_cast<const float4 *>(mean)[c4];
    const float4 rs = reinterpret_cast<const float4 *>(rstd)[c4];
    const float4 sa = scale[c4], sb = shift[c4];
#pragma unroll 4
    for (long long r = r0 + rl; r < r1; r += RL) {
      const float4 xv = x[r * C4 + c4];
      float4 g;
      if constexpr (POOL) g = pool_grad_at(pool, r, c4, C4);
      else g = dy[r * C4 + c4];
      if (relu) {
        g.x = fmaf(xv.x, sa.x, sb.x) > 0.f ? g.x : 0.f; g.y = fmaf(xv.y, sa.y, sb.y) > 0.f ? g.y : 0.f;
        g.z = fmaf(xv.z, sa.z, sb.z) > 0.f ? g.z : 0.f; g.w = fmaf(xv.w, sa.w, sb.w) > 0.f ? g.w : 0.f;
      }
      s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
      ss.x += g.x * ((xv.x - m.x) * rs.x); ss.y += g.y * ((xv.y - m.y) * rs.y);
      ss.z += g.z * ((xv.z - m.z) * rs.z); ss.w += g.w * ((xv.w - m.w) * rs.w);
    }
  }
  float4 *s_s = sm4, *s_ss = sm4 + kT;
  s_s[threadIdx.x] = s; s_ss[threadIdx.x] = ss;
  __syncthreads();
  if (rl == 0 && c4 < C4) {
    for (int k = 1; k < RL; ++k) {
      const float4 a = s_s[k * CL + cl], b = s_ss[k * CL + cl];
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
      ss.x += b.x; ss.y += b.y; ss.z += b.z; ss.w += b.w;
    }
    float4 *p = reinterpret_cast<float4 *>(partial) + (long long)blockIdx.x * 2 * C4;
    p[c4] = s; p[C4 + c4] = ss;
  }
}

// finalize: dgamma / dbeta and the three per-channel coefficients of
//   dx = a*dy' + c1*x + c0,  a = gamma*rstd, c1 = -a*rstd*mean(dy' xhat), c0 = -a*mean(dy') - c1*mean
__global__ __launch_bounds__(1024) void bn_bwd_final_kernel(
    const float *__restrict__ partial, int nslabs, int C, double inv_rows,
    const float *__restrict__ mean, const float *__restrict__ rstd, const float *__restrict__ gamma,
    float *__restrict__ coef, float *__restrict__ dgamma, float *__restrict__ dbeta,
    const float *__restrict__ dy_absmax, const float *__restrict__ x_minmax, unsigned *__restrict__ dx_bound,
    unsigned *__restrict__ dx_bound_min) {
  __shared__ double s_S[16][64];
  __shared__ double s_SS[16][64];
  __shared__ float s_D;
  if (dx_bound && threadIdx.x < 64) {      // D = the largest |dy'| the producing data gradient stored (finite partial maxima)
    float m = dy_absmax[threadIdx.x];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (threadIdx.x == 0) s_D = m;
  }
  const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double S = 0, SS = 0;
  if (c < C) {
#pragma unroll 8
    for (int k = sl; k < nslabs; k += 16) {   // unrolled: the kernel is pure load latency
      S += partial[(long long)k * 2 * C + c];
      SS += partial[(long long)k * 2 * C + C + c];
    }
  }
  s_S[sl][cl] = S; s_SS[sl][cl] = SS;
  __syncthreads();
  if (sl == 0 && c < C) {
    for (int k = 1; k < 16; ++k) { S += s_S[k][cl]; SS += s_SS[k][cl]; }
    dspn::bn_final_channel(c, C, S, SS, inv_rows, mean, rstd, gamma, coef, dgamma, dbeta, x_minmax, dx_bound, dx_bound_min,
                           dx_bound ? s_D : 0.f);      // (bn_final_job.h: the same arithmetic as the job form)
  }
}

// a parked finalize job (bn_final_job.h) that no weight-gradient launch took: on its own
__global__ __launch_bounds__(256) void bn_final_job_kernel(const dspn::BnFinalJob j) {
  __shared__ double sm[2 * 16 * dspn::kBnJobChannels + 1];
  dspn::bn_final_job_run(j, blockIdx.x, sm);
}

// the apply pass of the BatchNorm backward with dx written as fp16 PIECE PLANES for the two-piece math (round 4):
// [pixel][C / 32][piece][32], the same 4 bytes per element and the same byte offset for a float4's four channels as the
// float tensor -- (h0, h1) = split2h(dx, s) with s = operand_scale(bound), the bound bn_bwd_final_kernel has just left in
// the block the convolution in front of this BatchNorm reads as its dy magnitude.  That convolution's data gradient and
// weight gradient then copy the records into LDS without any arithmetic (conv_nt_kernel EPIX & 4, conv_wgrad_kernel
// MATHX = 4) instead of cutting every element once per (tap, column tile) that reads it.
// (a thread owns one 16-byte chunk of the planes and forms it from the two float4 of its eight channels: bn_apply_planes_kernel)
template <int U>
__global__ __launch_bounds__(256) void bn_bwd_apply_planes_kernel(const CA4Ptr x, const float4 *__restrict__ scale,
                                    const float4 *__restrict__ shift, const CA4Ptr dy,
                                    const float4 *__restrict__ coef, uint4 *__restrict__ planes,
                                    long long n4, int C4, int relu, const float *__restrict__ bound, int fixed_c) {
  const float s = dspn::pieces::operand_scale(bound);
  const bool nf = dspn::pieces::operand_nonfinite(bound);       // (round 5: an infinite gradient leaves as (+-65504, +-inf), as bn_apply_planes_kernel's)
  struct Co { float4 sa, sb, a, c1, c0; };
  auto coefs = [&](const int c4) __attribute__((always_inline)) {
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    return Co{relu ? scale[c4] : zero, relu ? shift[c4] : zero, coef[c4], coef[C4 + c4], coef[2 * C4 + c4]};
  };
  auto grad = [&](const float4 xv, float4 g, const Co &k) __attribute__((always_inline)) {
    if (relu) {
      g.x = fmaf(xv.x, k.sa.x, k.sb.x) > 0.f ? g.x : 0.f; g.y = fmaf(xv.y, k.sa.y, k.sb.y) > 0.f ? g.y : 0.f;
      g.z = fmaf(xv.z, k.sa.z, k.sb.z) > 0.f ? g.z : 0.f; g.w = fmaf(xv.w, k.sa.w, k.sb.w) > 0.f ? g.w : 0.f;
    }
    return make_float4(k.a.x * g.x + k.c1.x * xv.x + k.c0.x, k.a.y * g.y + k.c1.y * xv.y + k.c0.y,
                       k.a.z * g.z + k.c1.z * xv.z + k.c0.z, k.a.w * g.w + k.c1.w * xv.w + k.c0.w);     // (as bn_bwd_apply_kernel)
  };
  auto one = [&](const long long i, const bool hi, const float4 x0, const float4 g0, const float4 x1, const float4 g1,
                 const Co &k0, const Co &k1) __attribute__((always_inline)) {
    dspn::pieces::bf16x4 p0, p1, q0, q1;
    dspn::pieces::split2h(grad(x0, g0, k0), s, p0, p1);
    dspn::pieces::split2h(grad(x1, g1, k1), s, q0, q1);
    if (__builtin_expect(nf, 0)) { dspn::pieces::repair_inf(p0, p1); dspn::pieces::repair_inf(q0, q1); }
    planes[i] = hi ? plane_pack(p1, q1) : plane_pack(p0, q0);
  };
  const long long cstride = (long long)gridDim.x * (U * 256);
  long long base = blockIdx.x * (long long)(U * 256) + threadIdx.x;
  if (fixed_c) {
    const int c = (int)(base % C4);
    const PlaneChunk pc = plane_chunk(c);
    const Co k0 = coefs(pc.ca), k1 = coefs(pc.ca + 1);
    const int d = pc.ca - c;
    for (; base + (U - 1) * 256 < n4; base += cstride) {
      float4 x0[U], g0[U], x1[U], g1[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long a = base + u * 256 + d;
        x0[u] = x[a]; g0[u] = dy[a]; x1[u] = x[a + 1]; g1[u] = dy[a + 1];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) one(base + u * 256, pc.hi, x0[u], g0[u], x1[u], g1[u], k0, k1);
    }
    for (int u = 0; u < U; ++u) {
      const long long i = base + u * 256;
      if (i < n4) one(i, pc.hi, x[i + d], dy[i + d], x[i + d + 1], dy[i + d + 1], k0, k1);
    }
  } else {
    for (; base < n4; base += cstride) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long i = base + u * 256;
        if (i >= n4) break;
        const int c = (int)(i % C4);
        const PlaneChunk pc = plane_chunk(c);
        const long long a = i - c + pc.ca;
        one(i, pc.hi, x[a], dy[a], x[a + 1], dy[a + 1], coefs(pc.ca), coefs(pc.ca + 1));
      }
    }
  }
}

// the apply pass with the pooled output gradient (pool_grad_at): dx = a dy' + c1 x + c0, float tensors, no accumulation
__global__ __launch_bounds__(256) void bn_bwd_apply_pool_kernel(const float4 *__restrict__ x, const float4 *__restrict__ scale,
                                    const float4 *__restrict__ shift, const float4 *__restrict__ coef, float4 *__restrict__ dx,
                                    long long n4, int C4, int relu, unsigned *__restrict__ absmax, const PoolGrad pool) {
  float mx = 0.f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const unsigned row = (unsigned)i / (unsigned)C4;          // (n4 < 2^31, checked by the entry point)
    const int c4 = (int)((unsigned)i - row * (unsigned)C4);
    const float4 xv = x[i];
    float4 g = pool_grad_at(pool, row, c4, C4);
    if (relu) {
      const float4 sa = scale[c4], sb = shift[c4];
      g.x = fmaf(xv.x, sa.x, sb.x) > 0.f ? g.x : 0.f; g.y = fmaf(xv.y, sa.y, sb.y) > 0.f ? g.y : 0.f;
      g.z = fmaf(xv.z, sa.z, sb.z) > 0.f ? g.z : 0.f; g.w = fmaf(xv.w, sa.w, sb.w) > 0.f ? g.w : 0.f;
    }
    const float4 a = coef[c4], c1 = coef[C4 + c4], c0 = coef[2 * C4 + c4];
    const float4 o = make_float4(a.x * g.x + c1.x * xv.x + c0.x, a.y * g.y + c1.y * xv.y + c0.y,
                                 a.z * g.z + c1.z * xv.z + c0.z, a.w * g.w + c1.w * xv.w + c0.w);     // (as bn_bwd_apply_kernel)
    dx[i] = o;
    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
  }
  if (absmax) {          // (kernel-uniform)
    __shared__ float sm[4];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < 4; ++w) mx = fmaxf(mx, sm[w]);
      if (mx > 0.f) atomicMax(absmax + (blockIdx.x & 63), __float_as_uint(mx));
    }
  }
}

// absmax (optional, float tensors): 64 partial maxima of |dx| as stored -- the magnitude of the gradient the producing
// convolution's data / weight gradient multiply next in the two-piece fp16 math (include/dspn_nn.h dspn_absmax_f32); this
// kernel is HBM-bound and has the vector slots to spare, a separate pass would read dx again
// Streaming form: the launcher picks a grid whose stride (gridDim.x * 256 float4) is a multiple of C4 whenever it can
// (`fixed_c`), so a thread meets ONE channel group -- its five coefficient vectors are loaded once, no 64-bit modulo per
// element -- and four elements per tensor are requested before the first is used (this kernel is 3 - 4 passes over HBM and
// nothing else; one element at a time held it to 4.85 TB/s where a plain add reaches 6.1 on the same box).
template <int U>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const CA4Ptr x, const float4 *__restrict__ scale,
                                    const float4 *__restrict__ shift, const CA4Ptr dy,
                                    const float4 *__restrict__ coef, const A4Ptr dx,
                                    long long n4, int C4, int relu, int accumulate, unsigned *__restrict__ absmax,
                                    int fixed_c) {
  float mx = 0.f;
  auto one = [&](const float4 xv, float4 g, const float4 d, const float4 sa, const float4 sb, const float4 a,
                 const float4 c1, const float4 c0) __attribute__((always_inline)) {
    if (relu) {
      g.x = fmaf(xv.x, sa.x, sb.x) > 0.f ? g.x : 0.f; g.y = fmaf(xv.y, sa.y, sb.y) > 0.f ? g.y : 0.f;
      g.z = fmaf(xv.z, sa.z, sb.z) > 0.f ? g.z : 0.f; g.w = fmaf(xv.w, sa.w, sb.w) > 0.f ? g.w : 0.f;
    }
    float4 o = make_float4(a.x * g.x + c1.x * xv.x + c0.x, a.y * g.y + c1.y * xv.y + c0.y,
                           a.z * g.z + c1.z * xv.z + c0.z, a.w * g.w + c1.w * xv.w + c0.w);
    if (accumulate) { o.x += d.x; o.y += d.y; o.z += d.z; o.w += d.w; }
    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
    return o;
  };
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  // a workgroup walks chunks of U x 256 consecutive elements (U = 4: 16 KB per tensor; thread t takes t, t + 256, t + 512,
  // t + 768); small tensors run U = 1 so that they still spread over a few thousand workgroups
  const long long cstride = (long long)gridDim.x * (U * 256);
  long long base = blockIdx.x * (long long)(U * 256) + threadIdx.x;
  if (fixed_c) {       // 256 % C4 == 0: all of a thread's elements belong to one channel group
    const int c4 = (int)(base % C4);
    const float4 sa = relu ? scale[c4] : zero, sb = relu ? shift[c4] : zero;
    const float4 a = coef[c4], c1 = coef[C4 + c4], c0 = coef[2 * C4 + c4];
    for (; base + (U - 1) * 256 < n4; base += cstride) {
      float4 xv[U], g[U], d[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { xv[u] = x[base + u * 256]; g[u] = dy[base + u * 256]; }
#pragma unroll
      for (int u = 0; u < U; ++u) d[u] = accumulate ? (float4)dx[base + u * 256] : zero;
#pragma unroll
      for (int u = 0; u < U; ++u) dx[base + u * 256] = one(xv[u], g[u], d[u], sa, sb, a, c1, c0);
    }
    for (int u = 0; u < U; ++u) {       // the last, partial chunk (at most one workgroup's)
      const long long j = base + u * 256;
      if (j < n4) dx[j] = one(x[j], dy[j], accumulate ? (float4)dx[j] : zero, sa, sb, a, c1, c0);
    }
  } else {
    for (; base < n4; base += cstride) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long j = base + u * 256;
        if (j >= n4) break;
        const int c4 = (int)(j % C4);
        dx[j] = one(x[j], dy[j], accumulate ? (float4)dx[j] : zero, relu ? scale[c4] : zero, relu ? shift[c4] : zero, coef[c4],
                    coef[C4 + c4], coef[2 * C4 + c4]);
      }
    }
  }
  if (absmax) {          // (kernel-uniform)
    __shared__ float sm[kT / 64];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < kT / 64; ++w) mx = fmaxf(mx, sm[w]);
      unsigned *o = absmax + (blockIdx.x & 63);
      if (mx > 0.f && __float_as_uint(mx) > __builtin_nontemporal_load(o)) atomicMax(o, __float_as_uint(mx));   // (look before the atomic)
    }
  }
}

#ifdef DSPN_HALF
// BatchNorm-apply(+ReLU) on bf16 tensors with 16-byte accesses (8 channels per lane)
__global__ void bn_apply8_kernel(const dspn::u32x4_t *__restrict__ x, const float4 *__restrict__ scale,
                                 const float4 *__restrict__ shift, dspn::u32x4_t *__restrict__ y, long long n8, int C8,
                                 int relu) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8;
       i += (long long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % C8);
    const dspn::u32x4_t xw = x[i];
    dspn::u32x4_t ow;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const dspn::u32x2_t xh = {xw[2 * h], xw[2 * h + 1]};
      const float4 v = dspn::widen4(xh), a = scale[c8 * 2 + h], b = shift[c8 * 2 + h];
      float4 o = make_float4(fmaf(v.x, a.x, b.x), fmaf(v.y, a.y, b.y), fmaf(v.z, a.z, b.z), fmaf(v.w, a.w, b.w));
      if (relu) {
        o.x = o.x > 0.f ? o.x : 0.f; o.y = o.y > 0.f ? o.y : 0.f;
        o.z = o.z > 0.f ? o.z : 0.f; o.w = o.w > 0.f ? o.w : 0.f;
      }
      const dspn::u32x2_t on = dspn::narrow4(o);
      ow[2 * h] = on[0]; ow[2 * h + 1] = on[1];
    }
    y[i] = ow;
  }
}
// the backward apply pass likewise: this kernel moves ~12 GB per resnet-50 step
// and 8-byte lanes reach only ~4.4 TB/s
template <int U>
__global__ __launch_bounds__(256) void bn_bwd_apply8_kernel(const dspn::u32x4_t *__restrict__ x, const float4 *__restrict__ scale,
                                     const float4 *__restrict__ shift, const dspn::u32x4_t *__restrict__ dy,
                                     const float4 *__restrict__ coef, dspn::u32x4_t *__restrict__ dx,
                                     long long n8, int C8, int relu, int accumulate, int fixed_c) {
  const int C4 = C8 * 2;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  // one 16-byte element = 8 channels = two coefficient groups h
  auto one = [&](const dspn::u32x4_t xw, const dspn::u32x4_t gw, const dspn::u32x4_t dw, const float4 (&sa)[2],
                 const float4 (&sb)[2], const float4 (&a)[2], const float4 (&c1)[2], const float4 (&c0)[2])
                 __attribute__((always_inline)) {
    dspn::u32x4_t ow;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const dspn::u32x2_t xh = {xw[2 * h], xw[2 * h + 1]}, gh = {gw[2 * h], gw[2 * h + 1]}, dh = {dw[2 * h], dw[2 * h + 1]};
      const float4 xv = dspn::widen4(xh);
      float4 g = dspn::widen4(gh);
      if (relu) {
        g.x = fmaf(xv.x, sa[h].x, sb[h].x) > 0.f ? g.x : 0.f; g.y = fmaf(xv.y, sa[h].y, sb[h].y) > 0.f ? g.y : 0.f;
        g.z = fmaf(xv.z, sa[h].z, sb[h].z) > 0.f ? g.z : 0.f; g.w = fmaf(xv.w, sa[h].w, sb[h].w) > 0.f ? g.w : 0.f;
      }
      float4 o = make_float4(a[h].x * g.x + c1[h].x * xv.x + c0[h].x, a[h].y * g.y + c1[h].y * xv.y + c0[h].y,
                             a[h].z * g.z + c1[h].z * xv.z + c0[h].z, a[h].w * g.w + c1[h].w * xv.w + c0[h].w);
      if (accumulate) { const float4 d = dspn::widen4(dh); o.x += d.x; o.y += d.y; o.z += d.z; o.w += d.w; }
      const dspn::u32x2_t on = dspn::narrow4(o);
      ow[2 * h] = on[0]; ow[2 * h + 1] = on[1];
    }
    return ow;
  };
  auto coefs = [&](const int c8, float4 (&sa)[2], float4 (&sb)[2], float4 (&a)[2], float4 (&c1)[2], float4 (&c0)[2])
                   __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c4 = c8 * 2 + h;
      sa[h] = relu ? scale[c4] : zero; sb[h] = relu ? shift[c4] : zero;
      a[h] = coef[c4]; c1[h] = coef[C4 + c4]; c0[h] = coef[2 * C4 + c4];
    }
  };
  const dspn::u32x4_t z4 = {0u, 0u, 0u, 0u};
  float4 sa[2], sb[2], a[2], c1[2], c0[2];
  // (chunks of U x 256 elements per workgroup, as in bn_bwd_apply_kernel)
  const long long cstride = (long long)gridDim.x * (U * 256);
  long long base = blockIdx.x * (long long)(U * 256) + threadIdx.x;
  if (fixed_c) {
    coefs((int)(base % C8), sa, sb, a, c1, c0);
    for (; base + (U - 1) * 256 < n8; base += cstride) {
      dspn::u32x4_t xw[U], gw[U], dw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { xw[u] = x[base + u * 256]; gw[u] = dy[base + u * 256]; }
#pragma unroll
      for (int u = 0; u < U; ++u) dw[u] = accumulate ? dx[base + u * 256] : z4;
#pragma unroll
      for (int u = 0; u < U; ++u) dx[base + u * 256] = one(xw[u], gw[u], dw[u], sa, sb, a, c1, c0);
    }
    for (int u = 0; u < U; ++u) {
      const long long j = base + u * 256;
      if (j < n8) dx[j] = one(x[j], dy[j], accumulate ? dx[j] : z4, sa, sb, a, c1, c0);
    }
  } else {
    for (; base < n8; base += cstride) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long j = base + u * 256;
        if (j >= n8) break;
        coefs((int)(j % C8), sa, sb, a, c1, c0);
        dx[j] = one(x[j], dy[j], accumulate ? dx[j] : z4, sa, sb, a, c1, c0);
      }
    }
  }
}
#endif

// ------------------------------------------------------------------ element-wise
__global__ void add_kernel(const CA4Ptr a, const CA4Ptr b, const A4Ptr o, long long n4) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4;
       i += (long long)gridDim.x * blockDim.x) {
    const float4 x = a[i], y = b[i];
    o[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  }
}
__global__ void add_tail_kernel(const CA1Ptr a, const CA1Ptr b, const A1Ptr o, long long start, long long n) {
  const long long i = start + blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i < n) o[i] = a[i] + b[i];
}
__global__ void relu_bwd_kernel(const CA1Ptr y, const CA1Ptr dy, const A1Ptr dx, long long n, int accumulate) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    float v = y[i] > 0.f ? dy[i] : 0.f;
    if (accumulate) v += dx[i];
    dx[i] = v;
  }
}
#ifndef DSPN_HALF
__global__ void fill_kernel(float *p, float v, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    p[i] = v;
}
#endif

// column sums, two stage
__global__ __launch_bounds__(kT) void colsum_partial_kernel(const CA1Ptr a,
                                                            long long rows, int C, int ld,
                                                            float *__restrict__ partial, int slab_rows) {
  __shared__ float sm[kT];
  const int CL = min(C, kT), RL = kT / CL;
  const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
  const long long r0 = (long long)blockIdx.x * slab_rows, r1 = min(rows, r0 + slab_rows);
  for (int cb = 0; cb < C; cb += CL) {
    const int c = cb + cl;
    float s = 0.f;
    if (rl < RL && c < C)
      for (long long r = r0 + rl; r < r1; r += RL) s += a[r * ld + c];
    sm[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
      for (int k = 1; k < RL; ++k) s += sm[k * CL + cl];
      partial[(long long)blockIdx.x * C + c] = s;
    }
    __syncthreads();
  }
}
// dx = (y > 0) ? dy : 0 in place of a separate ReLU-backward pass, AND the per-slab column sums of dx (the bias
// gradient of a convolution with a fused ReLU epilogue): one read of y and dy instead of three passes.
// Layout: rows x ld floats, channels [0, C) summed; ld % 4 == 0; blocks of 256 threads = CL channel quads x RL rows.
__global__ __launch_bounds__(kT) void relu_bwd_colsum_partial_kernel(const CA4Ptr y,
                                                                   const CA4Ptr dy,
                                                                   const A4Ptr dx, long long rows, int ld4,
                                                                   int CL, float *__restrict__ partial, int C,
                                                                   int slab_rows, unsigned *__restrict__ dx_absmax) {
  extern __shared__ __attribute__((aligned(16))) float4 sm4[];
  const int RL = kT / CL;
  const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
  const int c4 = blockIdx.y * CL + cl;
  const long long r0 = (long long)blockIdx.x * slab_rows, r1 = min(rows, r0 + slab_rows);
  float4 s = make_float4(0, 0, 0, 0);
  float mx = 0.f;
  if (rl < RL && c4 < ld4) {
#pragma unroll 4
    for (long long r = r0 + rl; r < r1; r += RL) {
      const float4 yv = y[r * ld4 + c4];
      float4 g = dy[r * ld4 + c4];
      g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;
      g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
      dx[r * ld4 + c4] = g;
      s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
      mx = fmaxf(mx, fmaxf(fmaxf(fabsf(g.x), fabsf(g.y)), fmaxf(fabsf(g.z), fabsf(g.w))));
    }
  }
  if (dx_absmax) {     // (kernel-uniform) the magnitude block of dx as stored: one atomic per wave
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax(dx_absmax + ((blockIdx.x * 4u + (threadIdx.x >> 6)) & 63u), __float_as_uint(mx));
  }
  sm4[threadIdx.x] = s;
  __syncthreads();
  if (rl == 0 && c4 < ld4) {
    for (int k = 1; k < RL; ++k) {
      const float4 a = sm4[k * CL + cl];
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    float *p = partial + (long long)blockIdx.x * C;
    const float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c4 * 4 + e < C) p[c4 * 4 + e] = v[e];
  }
}
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float *partial, int nslabs, int C, float *out) {
  __shared__ double sm[64][17];
  const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double s = 0;
  if (c < C) {
#pragma unroll 8
    for (int k = sl; k < nslabs; k += 64) s += partial[(long long)k * C + c];
  }
  sm[sl][cl] = s;
  __syncthreads();
  if (sl == 0 && c < C) {
    for (int k = 1; k < 64; ++k) s += sm[k][cl];
    out[c] = (float)s;
  }
}

__global__ void nchw_to_nhwc_kernel(const float *__restrict__ src, const A1Ptr dst,
                                    int C, long long HW, long long total_pix, int Cp) {
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < total_pix;
       p += (long long)gridDim.x * blockDim.x) {
    const long long n = p / HW, hw = p - n * HW;
    for (int c = 0; c < Cp; ++c)
      dst[p * Cp + c] = c < C ? src[(n * C + c) * HW + hw] : 0.f;
  }
}
#ifndef DSPN_HALF
__global__ void nhwc_to_nchw_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                    int C, long long HW, long long total, int Cp) {
  // thread per dst element, hw fastest
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long hw = i % HW, nc = i / HW;
    const long long n = nc / C; const int c = (int)(nc - n * C);
    dst[i] = src[(n * HW + hw) * Cp + c];
  }
}
#endif
// SRC / DST: element pointers of either storage type (the SSD head packing converts bf16 maps into the float
// (B, N*5) / (B, N*(C+1)) tensors the losses and the multibox operators read, and their gradients back)
template <typename SRC, typename DST>
__global__ void copy_block_kernel(const SRC src, const DST dst,
                                  long long rows_per_sample, int C, long long sss, int lds, int soff,
                                  long long dss, int ldd, int doff, long long total, int accumulate) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long r = i / C;
    const long long s = r / rows_per_sample, rr = r - s * rows_per_sample;
    const float v = src[s * sss + rr * lds + soff + c];
    const long long di = s * dss + rr * ldd + doff + c;
    const float old = accumulate ? (float)dst[di] : 0.f;
    dst[di] = old + v;
  }
}
// the same copy in 16-byte units (same storage type on both sides, no accumulation, every extent / offset / stride a whole
// number of units): the channel-concat copies of inceptionv3 move 16 bytes per lane instead of one element after three
// 64-bit divisions
__global__ void copy_block_vec_kernel(const dspn::u32x4_t *__restrict__ src, dspn::u32x4_t *__restrict__ dst,
                                      long long rows_per_sample, int C, long long sss, int lds, int soff,
                                      long long dss, int ldd, int doff, long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long r = i / C;
    const long long s = r / rows_per_sample, rr = r - s * rows_per_sample;
    dst[s * dss + rr * ldd + doff + c] = src[s * sss + rr * lds + soff + c];
  }
}
#ifndef DSPN_HALF
// many block copies in ONE launch (the SSD head packing: six small maps per pass): rows sorted by `begin`, the first
// element of the row's range in the launch's flat index space, found by binary search (as slab_reduce_batch_kernel)
struct CopyDesc { const float *src; float *dst; long long rows_per_sample, sss, dss; int C, lds, soff, ldd, doff, accumulate; long long begin; };
static_assert(sizeof(CopyDesc) == 72, "copy_block_batch table row (dspnet_amd/functional.py copy_block_table)");
__global__ void copy_block_batch_kernel(const CopyDesc *__restrict__ d, int n, long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (d[mid].begin <= i) lo = mid; else hi = mid - 1;
    }
    const CopyDesc e = d[lo];
    const long long j = i - e.begin;
    const int c = (int)(j % e.C);
    const long long r = j / e.C;
    const long long sm = r / e.rows_per_sample, rr = r - sm * e.rows_per_sample;
    const float v = e.src[sm * e.sss + rr * e.lds + e.soff + c];
    const long long di = sm * e.dss + rr * e.ldd + e.doff + c;
    e.dst[di] = e.accumulate ? e.dst[di] + v : v;
  }
}
__global__ void transpose_bnc_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                     int N, int C, long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i % N);
    const long long bc = i / N;
    const int c = (int)(bc % C);
    const long long b = bc / C;
    dst[i] = src[(b * N + n) * C + c];
  }
}
#endif


// ------------------------------------------------------------------ tap-expanded convolution
// A stride-1 RxS convolution with very few output channels (score3_conv: 3328 -> 19) starves the
// 32-wide MFMA column dimension.  It is computed instead as a 1x1 convolution with Cout*R*S output
// channels (the weight [Cout][R][S][Cin] read as [(Cout*R*S)][Cin], no re-layout) followed by a
// shifted sum over taps; the backward pass spreads dy into the same (co, tap) channel layout and runs
// the 1x1 weight-gradient on it.
//   y[n,h,w,co] = bias[co] + sum_{r,s} z[n, h + r - ph, w + s - pw, co*R*S + r*S + s]
__global__ void tap_sum_kernel_(const CA1Ptr z, const float *__restrict__ bias, const A1Ptr y,
                                int H, int W, int Cout, int ldy, int ldz, int R, int S, int ph, int pw,
                                long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int co = (int)(i % ldy);
    long long t = i / ldy;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const long long n = t / H;
    float acc = 0.f;
    if (co < Cout) {
      for (int r = 0; r < R; ++r) {
        const int hh = h + r - ph;
        if ((unsigned)hh >= (unsigned)H) continue;
        for (int q = 0; q < S; ++q) {
          const int ww = w + q - pw;
          if ((unsigned)ww >= (unsigned)W) continue;
          acc += z[((n * H + hh) * W + ww) * ldz + co * R * S + r * S + q];
        }
      }
      if (bias) acc += bias[co];
    }
    y[i] = acc;
  }
}
//   dz[n,h,w,co*R*S + r*S + s] = dy[n, h - (r - ph), w - (s - pw), co]   (0 outside / in pad channels)
__global__ void tap_spread_kernel_(const CA1Ptr dy, const A1Ptr dz, int H, int W, int Cout,
                                   int ldy, int ldz, int R, int S, int ph, int pw, long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % ldz);
    long long t = i / ldz;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const long long n = t / H;
    float v = 0.f;
    if (ch < Cout * R * S) {
      const int co = ch / (R * S), tap = ch - co * R * S;
      const int r = tap / S, q = tap - r * S;
      const int hh = h - (r - ph), ww = w - (q - pw);
      if ((unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) v = dy[((n * H + hh) * W + ww) * ldy + co];
    }
    dz[i] = v;
  }
}

// ------------------------------------------------------------------ pooling
// optional argmax record: position r*k+s of the FIRST maximum of each window in (h, w) scan order, one
// byte per element (255 = no finite maximum); the backward pass then needs neither x nor y
// AFF (round 4): the pooled tensor is (relu)(x * scale[c] + shift[c]) -- a BatchNorm(+ReLU) folded into the pooling pass
// (symbol/resnet.py:96-98: bn0 -> relu0 -> pooling0): the normalised 32 x 256 x 256 x 64 tensor (537 MB at the bench shape) is
// neither written nor read back; same fmaf as every other evaluation of the affine, so the ReLU mask the BatchNorm backward
// recomputes from x is the one applied here.  absmax (optional): the magnitude block of the pooled output.
template <bool IDX, bool AFF>
__global__ __launch_bounds__(kT) void maxpool_fwd_kernel(const CA4Ptr x, const A4Ptr y,
                                   uchar4 *__restrict__ argmax, int H, int W,
                                   int C4, int k, int stride, int pad, int Ho, int Wo, long long total,
                                   const float4 *__restrict__ scale, const float4 *__restrict__ shift, int relu,
                                   unsigned *__restrict__ absmax) {
  float mx = 0.f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    float4 sa = make_float4(1.f, 1.f, 1.f, 1.f), sb = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (AFF) { sa = scale[c4]; sb = shift[c4]; }
    long long t = i / C4;
    const int wo = (int)(t % Wo); t /= Wo;
    const int ho = (int)(t % Ho);
    const long long n = t / Ho;
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    int ix = 255, iy = 255, iz = 255, iw = 255;
    for (int r = 0; r < k; ++r) {
      const int h = ho * stride - pad + r;
      if ((unsigned)h >= (unsigned)H) continue;
      for (int s = 0; s < k; ++s) {
        const int w = wo * stride - pad + s;
        if ((unsigned)w >= (unsigned)W) continue;
        float4 v = x[((n * H + h) * W + w) * C4 + c4];
        if constexpr (AFF) {
          v = make_float4(fmaf(v.x, sa.x, sb.x), fmaf(v.y, sa.y, sb.y), fmaf(v.z, sa.z, sb.z), fmaf(v.w, sa.w, sb.w));
          if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        }
        const int pos = r * k + s;
        if (v.x > m.x) { m.x = v.x; ix = pos; }
        if (v.y > m.y) { m.y = v.y; iy = pos; }
        if (v.z > m.z) { m.z = v.z; iz = pos; }
        if (v.w > m.w) { m.w = v.w; iw = pos; }
      }
    }
    y[i] = m;
    if constexpr (IDX) argmax[i] = make_uchar4((unsigned char)ix, (unsigned char)iy, (unsigned char)iz, (unsigned char)iw);
    if constexpr (AFF) mx = fmaxf(mx, fmaxf(fmaxf(fabsf(m.x), fabsf(m.y)), fmaxf(fabsf(m.z), fabsf(m.w))));
  }
  if constexpr (AFF) {
    if (absmax) {          // (kernel-uniform)
      __shared__ float sm[kT / 64];
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mx;
      __syncthreads();
      if (threadIdx.x == 0) {
        for (int w = 1; w < kT / 64; ++w) mx = fmaxf(mx, sm[w]);
        if (mx > 0.f && mx < INFINITY) atomicMax(absmax + (blockIdx.x & 63), __float_as_uint(mx));
      }
    }
  }
}

// gather form over the argmax record: every input pixel visits the (at most ceil(k/stride)^2) windows
// covering it in a fixed order and takes dy where the record names its position.  No atomics.
__global__ void maxpool_bwd_idx_kernel(const uchar4 *__restrict__ argmax, const CA4Ptr dy,
                                       const A4Ptr dx, int H, int W, int C4, int k, int stride,
                                       int pad, int Ho, int Wo, long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long long t = i / C4;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const long long n = t / H;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    int ho_lo = (h + pad - k + stride) / stride; if (h + pad - k + 1 <= 0) ho_lo = 0;
    int wo_lo = (w + pad - k + stride) / stride; if (w + pad - k + 1 <= 0) wo_lo = 0;
    const int ho_hi = min(Ho - 1, (h + pad) / stride), wo_hi = min(Wo - 1, (w + pad) / stride);
    for (int ho = ho_lo; ho <= ho_hi; ++ho)
      for (int wo = wo_lo; wo <= wo_hi; ++wo) {
        const long long oi = ((n * Ho + ho) * Wo + wo) * C4 + c4;
        const int pos = (h - (ho * stride - pad)) * k + (w - (wo * stride - pad));
        const uchar4 a = argmax[oi];
        const float4 d = dy[oi];
        g.x += a.x == pos ? d.x : 0.f; g.y += a.y == pos ? d.y : 0.f;
        g.z += a.z == pos ? d.z : 0.f; g.w += a.w == pos ? d.w : 0.f;
      }
    dx[i] = g;
  }
}

// gather form: for every input pixel, visit the windows covering it in a fixed order and take
// dy where this pixel is the FIRST maximum of the window in (h, w) scan order.  4 channels per thread.
__global__ void maxpool_bwd_kernel(const CA4Ptr x, const CA4Ptr y,
                                   const CA4Ptr dy, const A4Ptr dx, int H, int W,
                                   int C4, int k, int stride, int pad, int Ho, int Wo, long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long long t = i / C4;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const long long n = t / H;
    const float4 xv = x[i];
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    int ho_lo = (h + pad - k + stride) / stride; if (h + pad - k + 1 <= 0) ho_lo = 0;
    int wo_lo = (w + pad - k + stride) / stride; if (w + pad - k + 1 <= 0) wo_lo = 0;
    const int ho_hi = min(Ho - 1, (h + pad) / stride), wo_hi = min(Wo - 1, (w + pad) / stride);
    for (int ho = ho_lo; ho <= ho_hi; ++ho)
      for (int wo = wo_lo; wo <= wo_hi; ++wo) {
        const long long oi = ((n * Ho + ho) * Wo + wo) * C4 + c4;
        const float4 yv = y[oi];
        bool cand[4] = {yv.x == xv.x, yv.y == xv.y, yv.z == xv.z, yv.w == xv.w};
        if (!(cand[0] | cand[1] | cand[2] | cand[3])) continue;
        // an earlier position of this window holding the same (max) value takes the gradient instead
        const int h0 = ho * stride - pad, w0 = wo * stride - pad;
        bool done = false;
        for (int r = 0; r < k && !done; ++r) {
          const int hh = h0 + r;
          if ((unsigned)hh >= (unsigned)H) continue;
          for (int q = 0; q < k; ++q) {
            const int ww = w0 + q;
            if ((unsigned)ww >= (unsigned)W) continue;
            if (hh == h && ww == w) { done = true; break; }
            const float4 e = x[((n * H + hh) * W + ww) * C4 + c4];
            cand[0] &= e.x != xv.x; cand[1] &= e.y != xv.y; cand[2] &= e.z != xv.z; cand[3] &= e.w != xv.w;
          }
        }
        const float4 d = dy[oi];
        if (cand[0]) g[0] += d.x;
        if (cand[1]) g[1] += d.y;
        if (cand[2]) g[2] += d.z;
        if (cand[3]) g[3] += d.w;
      }
    dx[i] = make_float4(g[0], g[1], g[2], g[3]);
  }
}

__global__ void avgpool_fwd_kernel(const CA4Ptr x, const A4Ptr y, int H, int W,
                                   int C4, int k, int Ho, int Wo, long long total) {
  const float inv = 1.f / (float)(k * k);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long long t = i / C4;
    const int wo = (int)(t % Wo); t /= Wo;
    const int ho = (int)(t % Ho);
    const long long n = t / Ho;
    float4 s = make_float4(0, 0, 0, 0);
    for (int r = 0; r < k; ++r)
      for (int q = 0; q < k; ++q) {
        const float4 v = x[((n * H + ho * k + r) * W + wo * k + q) * C4 + c4];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    y[i] = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
  }
}
__global__ void avgpool_bwd_kernel(const CA4Ptr dy, const A4Ptr dx, int H, int W,
                                   int C4, int k, int Ho, int Wo, long long total, int accumulate) {
  const float inv = 1.f / (float)(k * k);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long long t = i / C4;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const long long n = t / H;
    const int ho = h / k, wo = w / k;
    float4 g = make_float4(0, 0, 0, 0);
    if (ho < Ho && wo < Wo) {
      const float4 v = dy[((n * Ho + ho) * Wo + wo) * C4 + c4];
      g = make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
    }
    if (accumulate) { const float4 d = dx[i]; g.x += d.x; g.y += d.y; g.z += d.z; g.w += d.w; }
    dx[i] = g;
  }
}

// overlapping average pooling (symbol/inceptionv3.py:31,74,115: 3x3 stride 1 pad 1): the sum over the
// in-image part of the window divided by k*k -- MXNet counts the padding (mshadow pool over a padded
// tensor times 1/(ky*kx))
__global__ void avgpool2d_fwd_kernel(const CA4Ptr x, const A4Ptr y, int H, int W, int C4,
                                     int k, int stride, int pad, int Ho, int Wo, long long total) {
  const float inv = 1.f / (float)(k * k);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long long t = i / C4;
    const int wo = (int)(t % Wo); t /= Wo;
    const int ho = (int)(t % Ho);
    const long long n = t / Ho;
    float4 s = make_float4(0, 0, 0, 0);
    for (int r = 0; r < k; ++r) {
      const int h = ho * stride - pad + r;
      if ((unsigned)h >= (unsigned)H) continue;
      for (int q = 0; q < k; ++q) {
        const int w = wo * stride - pad + q;
        if ((unsigned)w >= (unsigned)W) continue;
        const float4 v = x[((n * H + h) * W + w) * C4 + c4];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    }
    y[i] = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
  }
}
// gather form: every input pixel sums dy over the windows that contain it, in a fixed order
__global__ void avgpool2d_bwd_kernel(const CA4Ptr dy, const A4Ptr dx, int H, int W, int C4,
                                     int k, int stride, int pad, int Ho, int Wo, long long total, int accumulate) {
  const float inv = 1.f / (float)(k * k);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long long t = i / C4;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const long long n = t / H;
    int ho_lo = (h + pad - k + stride) / stride; if (h + pad - k + 1 <= 0) ho_lo = 0;
    int wo_lo = (w + pad - k + stride) / stride; if (w + pad - k + 1 <= 0) wo_lo = 0;
    const int ho_hi = min(Ho - 1, (h + pad) / stride), wo_hi = min(Wo - 1, (w + pad) / stride);
    float4 g = make_float4(0, 0, 0, 0);
    for (int ho = ho_lo; ho <= ho_hi; ++ho)
      for (int wo = wo_lo; wo <= wo_hi; ++wo) {
        const float4 v = dy[((n * Ho + ho) * Wo + wo) * C4 + c4];
        g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w;
      }
    g.x *= inv; g.y *= inv; g.z *= inv; g.w *= inv;
    if (accumulate) { const float4 d = dx[i]; g.x += d.x; g.y += d.y; g.z += d.z; g.w += d.w; }
    dx[i] = g;
  }
}

#ifndef DSPN_HALF   // plain resize (identity grid) and the evaluation read-outs: float tensors only
// ------------------------------------------------------------------ bilinear sampler
// GridGenerator(affine identity, target (Ho,Wo)) + BilinearSampler: normalised target coordinate
// g = -1 + o*2/(O-1); source coordinate s = (g+1)*(I-1)/2; corners outside [0,I-1] contribute 0.
__device__ __forceinline__ float src_coord(int o, int O, int I) {
  const float g = O > 1 ? -1.f + (float)o * (2.f / (float)(O - 1)) : 0.f;
  return (g + 1.f) * (float)(I - 1) / 2.f;
}
template <bool ACC>   // ACC: y += resize(x) (sum of several resized maps in a fixed order)
__global__ void bilinear_fwd_kernel(const float4 *__restrict__ x, float *__restrict__ y, int Hin,
                                    int Win, int C4, int Ho, int Wo, int ldo, int coff, long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long long t = i / C4;
    const int wo = (int)(t % Wo); t /= Wo;
    const int ho = (int)(t % Ho);
    const long long n = t / Ho;
    const float ys = src_coord(ho, Ho, Hin), xs = src_coord(wo, Wo, Win);
    const int y0 = (int)floorf(ys), x0 = (int)floorf(xs);
    const float wy0 = 1.f - (ys - (float)y0), wx0 = 1.f - (xs - (float)x0);
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int dyy = 0; dyy < 2; ++dyy)
#pragma unroll
      for (int dxx = 0; dxx < 2; ++dxx) {
        const int yy = y0 + dyy, xx = x0 + dxx;
        if ((unsigned)yy >= (unsigned)Hin || (unsigned)xx >= (unsigned)Win) continue;
        const float wgt = (dyy ? 1.f - wy0 : wy0) * (dxx ? 1.f - wx0 : wx0);
        const float4 v = x[((n * Hin + yy) * Win + xx) * C4 + c4];
        acc.x += wgt * v.x; acc.y += wgt * v.y; acc.z += wgt * v.z; acc.w += wgt * v.w;
      }
    float4 *o = reinterpret_cast<float4 *>(y + ((n * Ho + ho) * Wo + wo) * (long long)ldo + coff + c4 * 4);
    if (ACC) { const float4 p = *o; acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w; }
    *o = acc;
  }
}
// gather form of the backward: every source pixel collects from the target pixels that touch it
__global__ void bilinear_bwd_kernel(const float *__restrict__ dy, float4 *__restrict__ dx, int Hin,
                                    int Win, int C4, int Ho, int Wo, int ldo, int coff, long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long long t = i / C4;
    const int w = (int)(t % Win); t /= Win;
    const int h = (int)(t % Hin);
    const long long n = t / Hin;
    // candidate target range: |s(o) - h| < 1  (s is monotone in o); widen by one and test exactly
    const float sy = Hin > 1 ? (float)(Ho - 1) / (float)(Hin - 1) : 0.f;
    const float sx = Win > 1 ? (float)(Wo - 1) / (float)(Win - 1) : 0.f;
    int ho_lo = Hin > 1 ? (int)floorf((float)(h - 1) * sy) - 1 : 0;
    int ho_hi = Hin > 1 ? (int)ceilf((float)(h + 1) * sy) + 1 : Ho - 1;
    int wo_lo = Win > 1 ? (int)floorf((float)(w - 1) * sx) - 1 : 0;
    int wo_hi = Win > 1 ? (int)ceilf((float)(w + 1) * sx) + 1 : Wo - 1;
    ho_lo = max(ho_lo, 0); wo_lo = max(wo_lo, 0); ho_hi = min(ho_hi, Ho - 1); wo_hi = min(wo_hi, Wo - 1);
    float4 acc = make_float4(0, 0, 0, 0);
    for (int ho = ho_lo; ho <= ho_hi; ++ho) {
      const float ys = src_coord(ho, Ho, Hin);
      const int y0 = (int)floorf(ys);
      float wy;
      if (y0 == h) wy = 1.f - (ys - (float)y0);
      else if (y0 + 1 == h) wy = ys - (float)y0;
      else continue;
      for (int wo = wo_lo; wo <= wo_hi; ++wo) {
        const float xs = src_coord(wo, Wo, Win);
        const int x0 = (int)floorf(xs);
        float wx;
        if (x0 == w) wx = 1.f - (xs - (float)x0);
        else if (x0 + 1 == w) wx = xs - (float)x0;
        else continue;
        const float wgt = wy * wx;
        const float4 v = *reinterpret_cast<const float4 *>(
            dy + ((n * Ho + ho) * Wo + wo) * (long long)ldo + coff + c4 * 4);
        acc.x += wgt * v.x; acc.y += wgt * v.y; acc.z += wgt * v.z; acc.w += wgt * v.w;
      }
    }
    dx[i] = acc;
  }
}

// Separable form of the same backward pass (the bilinear weights factor into wy * wx): first along W into a
// (N, Ho, Win, C) scratch tensor, then along H.  A source pixel of a map upsampled by s collects from ~2s targets per
// pass instead of ~(2s)^2 (the 4x4 pyramid level upsampled to 64x64: 84 loads instead of 1764), and the first pass
// has N*Ho*Win*C/4 threads however small the source map is.
__device__ __forceinline__ void bl_range(int i, int I, int O, int &lo, int &hi) {
  const float s = I > 1 ? (float)(O - 1) / (float)(I - 1) : 0.f;
  lo = I > 1 ? (int)floorf((float)(i - 1) * s) - 1 : 0;
  hi = I > 1 ? (int)ceilf((float)(i + 1) * s) + 1 : O - 1;
  lo = max(lo, 0); hi = min(hi, O - 1);
}
__device__ __forceinline__ bool bl_weight(int o, int O, int I, int i, float &wgt) {
  const float ss = src_coord(o, O, I);
  const int i0 = (int)floorf(ss);
  if (i0 == i) { wgt = 1.f - (ss - (float)i0); return true; }
  if (i0 + 1 == i) { wgt = ss - (float)i0; return true; }
  return false;
}
__global__ void bilinear_bwd_w_kernel(const float *__restrict__ dy, float4 *__restrict__ tmp, int Win, int C4,
                                      int Ho, int Wo, int ldo, int coff, long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long long t = i / C4;
    const int w = (int)(t % Win); t /= Win;          // t = n * Ho + ho
    int lo, hi;
    bl_range(w, Win, Wo, lo, hi);
    float4 acc = make_float4(0, 0, 0, 0);
    for (int wo = lo; wo <= hi; ++wo) {
      float wx;
      if (!bl_weight(wo, Wo, Win, w, wx)) continue;
      const float4 v = *reinterpret_cast<const float4 *>(dy + (t * Wo + wo) * (long long)ldo + coff + c4 * 4);
      acc.x += wx * v.x; acc.y += wx * v.y; acc.z += wx * v.z; acc.w += wx * v.w;
    }
    tmp[i] = acc;
  }
}
__global__ void bilinear_bwd_h_kernel(const float4 *__restrict__ tmp, float4 *__restrict__ dx, int Hin, int Win,
                                      int C4, int Ho, long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long long t = i / C4;
    const int w = (int)(t % Win); t /= Win;
    const int h = (int)(t % Hin);
    const long long n = t / Hin;
    int lo, hi;
    bl_range(h, Hin, Ho, lo, hi);
    float4 acc = make_float4(0, 0, 0, 0);
    for (int ho = lo; ho <= hi; ++ho) {
      float wy;
      if (!bl_weight(ho, Ho, Hin, h, wy)) continue;
      const float4 v = tmp[((n * Ho + ho) * Win + w) * C4 + c4];
      acc.x += wy * v.x; acc.y += wy * v.y; acc.z += wy * v.z; acc.w += wy * v.w;
    }
    dx[i] = acc;
  }
}

// ------------------------------------------------------------------ segmentation readouts
// Counts behind CustomAccuracyMetric (train/metric.py:100-133) and IoUMetric (evaluate/eval_metric.py:359-384):
// pred = argmax over the class axis (first maximum), counts[0*C + c] = #(label == c & pred == c),
// counts[1*C + c] = #(pred == c), counts[2*C + c] = #(label == c), counts[3*C] = #(pred == label) -- integer
// atomics (order independent).  scores: rows x ld floats, classes [0, C); label: rows floats (any value, e.g. 255).
constexpr int kMaxSegC = 64;
__global__ __launch_bounds__(kT) void seg_counts_kernel(const float *__restrict__ scores, const float *__restrict__ label,
                                                       long long rows, int C, int ld,
                                                       unsigned long long *__restrict__ counts) {
  __shared__ unsigned int h[3 * kMaxSegC + 1];
  for (int i = threadIdx.x; i < 3 * C + 1; i += kT) h[i] = 0;
  __syncthreads();
  for (long long r = blockIdx.x * (long long)kT + threadIdx.x; r < rows; r += (long long)gridDim.x * kT) {
    const float *p = scores + r * ld;
    int best = 0;
    float bv = p[0];
    for (int c = 1; c < C; ++c) {
      const float v = p[c];
      if (v > bv) { bv = v; best = c; }
    }
    const int lab = (int)label[r];
    atomicAdd(&h[C + best], 1u);
    if (lab >= 0 && lab < C) atomicAdd(&h[2 * C + lab], 1u);
    if (lab == best) { atomicAdd(&h[best], 1u); atomicAdd(&h[3 * C], 1u); }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * C + 1; i += kT)
    if (h[i]) atomicAdd(&counts[i < 3 * C ? i : 3 * C], (unsigned long long)h[i]);
}

// Segmentation read-out at full resolution (multi_eval.py:28-34, prob_upsampling): BilinearSampler of the class
// probabilities onto an identity-affine grid of Ho x Wo followed by argmax over classes -- fused, the (C, Ho, Wo)
// probability volume (19 x 1024 x 2048 floats per image) is never written.  Each thread produces 4 neighbouring
// output pixels (one 4-byte store); the <= 4 x 2 source pixels they touch are L1/L2 hits for the whole 8x8 patch.
// Arithmetic order per class follows the sampler: tl*wy*wx + tr*wy*(1-wx) + bl*(1-wy)*wx + br*(1-wy)*(1-wx),
// corners outside the map contribute 0; ties keep the lowest class (argmax).
// (separate, uncontracted multiplies and adds: the class map is compared bit for bit with the CPU restatement)
__device__ __forceinline__ float src_coord_exact(int o, int O, int I) {
#pragma clang fp contract(off)
  const float step = 2.f / (float)(O - 1);
  const float prod = (float)o * step;
  const float g = O > 1 ? -1.f + prod : 0.f;
  return (g + 1.f) * (float)(I - 1) / 2.f;
}
template <bool VEC>   // VEC: ld % 4 == 0, the class vectors of the four corners are read as float4
__global__ __launch_bounds__(kT) void seg_upsample_argmax_kernel(const float *__restrict__ prob, unsigned char *__restrict__ out,
                                                                int Hin, int Win, int C, int ld, int Ho, int Wo,
                                                                long long total) {
#pragma clang fp contract(off)
  const int Wq = (Wo + 3) / 4;
  for (long long i = blockIdx.x * (long long)kT + threadIdx.x; i < total; i += (long long)gridDim.x * kT) {
    const int wq = (int)(i % Wq);
    long long t = i / Wq;
    const int ho = (int)(t % Ho);
    const long long n = t / Ho;
    const float ys = src_coord_exact(ho, Ho, Hin);
    const int y0 = (int)floorf(ys);
    const float wy = 1.f - (ys - (float)y0);
    const bool vy0 = (unsigned)y0 < (unsigned)Hin, vy1 = (unsigned)(y0 + 1) < (unsigned)Hin;
    const float *r0 = prob + ((n * Hin + (vy0 ? y0 : 0)) * Win) * (long long)ld;
    const float *r1 = prob + ((n * Hin + (vy1 ? y0 + 1 : 0)) * Win) * (long long)ld;
    unsigned char res[4] = {0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int wo = wq * 4 + e;
      if (wo >= Wo) break;
      const float xs = src_coord_exact(wo, Wo, Win);
      const int x0 = (int)floorf(xs);
      const float wx = 1.f - (xs - (float)x0);
      const bool vx0 = (unsigned)x0 < (unsigned)Win, vx1 = (unsigned)(x0 + 1) < (unsigned)Win;
      const float *tl = r0 + (long long)(vx0 ? x0 : 0) * ld, *tr = r0 + (long long)(vx1 ? x0 + 1 : 0) * ld;
      const float *bl = r1 + (long long)(vx0 ? x0 : 0) * ld, *br = r1 + (long long)(vx1 ? x0 + 1 : 0) * ld;
      // a corner outside the map contributes 0: its weight product is applied to a zero value
      const float mtl = (vy0 && vx0) ? 1.f : 0.f, mtr = (vy0 && vx1) ? 1.f : 0.f;
      const float mbl = (vy1 && vx0) ? 1.f : 0.f, mbr = (vy1 && vx1) ? 1.f : 0.f;
      const float wx1 = 1.f - wx, wy1 = 1.f - wy;
      float bv = -INFINITY;
      int best = 0;
      auto consider = [&](const int c, float a, float b, float d, float f) __attribute__((always_inline)) {
        a = mtl != 0.f ? a : 0.f; b = mtr != 0.f ? b : 0.f; d = mbl != 0.f ? d : 0.f; f = mbr != 0.f ? f : 0.f;
        const float v = a * wy * wx + b * wy * wx1 + d * wy1 * wx + f * wy1 * wx1;
        if (c < C && (v > bv || c == 0)) { bv = v; best = c; }
      };
      if constexpr (VEC) {
        for (int c = 0; c < C; c += 4) {
          const float4 a = *reinterpret_cast<const float4 *>(tl + c), b = *reinterpret_cast<const float4 *>(tr + c);
          const float4 d = *reinterpret_cast<const float4 *>(bl + c), f = *reinterpret_cast<const float4 *>(br + c);
          consider(c, a.x, b.x, d.x, f.x);
          consider(c + 1, a.y, b.y, d.y, f.y);
          consider(c + 2, a.z, b.z, d.z, f.z);
          consider(c + 3, a.w, b.w, d.w, f.w);
        }
      } else {
        for (int c = 0; c < C; ++c) consider(c, tl[c], tr[c], bl[c], br[c]);
      }
      res[e] = (unsigned char)best;
    }
    unsigned char *o = out + (n * Ho + ho) * (long long)Wo + wq * 4;
    if (wq * 4 + 3 < Wo && (Wo & 3) == 0) {
      *reinterpret_cast<uchar4 *>(o) = make_uchar4(res[0], res[1], res[2], res[3]);
    } else {
      for (int e = 0; e < 4 && wq * 4 + e < Wo; ++e) o[e] = res[e];
    }
  }
}

#endif   // !DSPN_HALF

// ------------------------------------------------------------------ losses
// logits and their gradient are activations (storage type); the probabilities are a float output
constexpr int kMaxSoftmaxC = 64;
// CMAX: compile-time bound of ld (the row length) -- the row lives in CMAX registers, every loop over it is unrolled with a
// `c < C` / `c < ld` guard (a `float v[64]` indexed by a run-time C lived in scratch memory: 0.14 ms for the 524 288 x 19
// segmentation rows of the bench step).  Same operations in the same order for every CMAX.
template <int CMAX>
__global__ void softmax_output_kernel(const CA1Ptr logits, const float *__restrict__ label,
                                      float *__restrict__ prob, const A1Ptr grad, long long rows,
                                      int C, int ld, float ignore_label, float grad_scale,
                                      const float *__restrict__ valid_count) {
  float scale = grad_scale;
  if (valid_count) scale = grad_scale / fmaxf(1.f, *valid_count);
  for (long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x; r < rows;
       r += (long long)gridDim.x * blockDim.x) {
    const CA1Ptr p = logits + r * ld;
    float v[CMAX];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      v[c] = c < C ? (float)p[c] : -INFINITY;
      mx = v[c] > mx ? v[c] : mx;
    }
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      if (c < C) { v[c] = expf(v[c] - mx); sum += v[c]; }
    }
    const float inv = 1.f / sum;
    const float lab = label ? label[r] : 0.f;
    const bool ign = label ? (lab == ignore_label) : true;
    const int li = (int)lab;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      if (c < ld) {
        const float pr = c < C ? v[c] * inv : 0.f;
        prob[r * ld + c] = pr;
        if (grad) grad[r * ld + c] = (ign || c >= C) ? 0.f : (pr - (c == li ? 1.f : 0.f)) * scale;
      }
    }
  }
}
#ifndef DSPN_HALF
__global__ void count_kernel(const float *__restrict__ a, long long n, int mode, float ref, float *out) {
  float c = 0.f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    c += (mode == 0 ? (a[i] != ref) : (a[i] > ref)) ? 1.f : 0.f;
  for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
  // integer-valued float adds below 2^24 are exact, hence order independent
  if ((threadIdx.x & 63) == 0 && c != 0.f) atomicAdd(out, c);
}
__device__ __forceinline__ float smooth_l1(float x) {
  const float ax = fabsf(x);
  return ax < 1.f ? 0.5f * x * x : ax - 0.5f;
}
__global__ void smooth_l1_fwd_kernel(const float *pred, const float *target, const float *mask,
                                     float *loss, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    loss[i] = smooth_l1(mask[i] * (pred[i] - target[i]));
}
__global__ void smooth_l1_bwd_kernel(const float *pred, const float *target, const float *mask,
                                     float *grad, long long n, float grad_scale, const float *valid_count) {
  const float sc = grad_scale / fmaxf(1.f, *valid_count);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float m = mask[i];
    const float x = m * (pred[i] - target[i]);
    const float d = fabsf(x) < 1.f ? x : (x > 0.f ? 1.f : -1.f);
    grad[i] = sc * m * d;
  }
}
// single-block deterministic reductions for the (tiny) metric readouts
__global__ __launch_bounds__(1024) void ce_sum_kernel(const float *prob, const float *label, long long rows,
                                                      int C, int ld, float ignore_label, float eps,
                                                      float *out2) {
  __shared__ double s_a[1024];
  __shared__ double s_b[1024];
  double a = 0, b = 0;
  for (long long r = threadIdx.x; r < rows; r += 1024) {
    const float lab = label[r];
    if (lab == ignore_label) continue;
    const int li = (int)lab;
    if (li < 0 || li >= C) continue;
    a += -log((double)(prob[r * ld + li] + eps));
    b += 1.0;
  }
  s_a[threadIdx.x] = a; s_b[threadIdx.x] = b;
  __syncthreads();
  for (int s = 512; s >= 1; s >>= 1) {
    if ((int)threadIdx.x < s) { s_a[threadIdx.x] += s_a[threadIdx.x + s]; s_b[threadIdx.x] += s_b[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out2[0] = (float)s_a[0]; out2[1] = (float)s_b[0]; }
}
__global__ __launch_bounds__(1024) void sum_kernel(const float *a, long long n, float *out) {
  __shared__ double s_a[1024];
  double v = 0;
  for (long long i = threadIdx.x; i < n; i += 1024) v += a[i];
  s_a[threadIdx.x] = v;
  __syncthreads();
  for (int s = 512; s >= 1; s >>= 1) {
    if ((int)threadIdx.x < s) s_a[threadIdx.x] += s_a[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)s_a[0];
}

__global__ void sgd_kernel(float4 *__restrict__ w, const float4 *__restrict__ g, float4 *__restrict__ m,
                           long long n4, float lr, float mu, float wd, float rescale) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4;
       i += (long long)gridDim.x * blockDim.x) {
    float4 wv = w[i]; const float4 gv = g[i]; float4 mv = m[i];
    mv.x = mu * mv.x - lr * (rescale * gv.x + wd * wv.x); wv.x += mv.x;
    mv.y = mu * mv.y - lr * (rescale * gv.y + wd * wv.y); wv.y += mv.y;
    mv.z = mu * mv.z - lr * (rescale * gv.z + wd * wv.z); wv.z += mv.z;
    mv.w = mu * mv.w - lr * (rescale * gv.w + wd * wv.w); wv.w += mv.w;
    w[i] = wv; m[i] = mv;
  }
}

#endif   // !DSPN_HALF

int bn_slabs(long long rows) { const int sr = slab_rows_for(rows); return (int)((rows + sr - 1) / sr); }

}  // namespace

#define S_(x) ((hipStream_t)(x))

extern "C" {

static size_t bn_workspace_bytes(long long rows, int C) {
  if (rows <= 0 || C <= 0) return 0;
  return sizeof(float) * ((size_t)bn_slabs(rows) * 2 * C + 4 * (size_t)C);
}
static size_t bn_tiles_workspace_bytes(int tiles, int C) {
  if (tiles <= 0 || C <= 0) return 0;
  return sizeof(float) * 4 * (size_t)((tiles + kTileGroup - 1) / kTileGroup) * C;     // grouped statistics + grouped (min, max)
}
#ifndef DSPN_HALF
size_t dspn_bn_workspace_bytes(long long rows, int C) { return bn_workspace_bytes(rows, C); }
#endif

int DSPN_FN(dspn_bn_stats)(const st_t *x, long long rows, int C, float eps, const float *gamma,
                      const float *beta, float *mean, float *rstd, float *scale, float *shift,
                      void *workspace, size_t workspace_bytes, void *stream) {
  DSPN_REQUIRE(x && beta && mean && rstd && scale && shift && workspace, "bn_stats: null pointer");
  DSPN_REQUIRE(rows > 0 && C > 0 && C % 4 == 0, "bn_stats: C must be a positive multiple of 4");
  if (workspace_bytes < bn_workspace_bytes(rows, C))
    return dspn::fail(DSPN_ERR_WORKSPACE_, "bn_stats: workspace too small");
  const int C4 = C / 4, CL = std::min(C4, 64), ns = bn_slabs(rows);
  float *partial = static_cast<float *>(workspace);
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(ns, (C4 + CL - 1) / CL), dim3(kT),
                     sizeof(float4) * 2 * kT, S_(stream), CA4Ptr(x), rows,
                     C4, CL, partial, slab_rows_for(rows));
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3((C + 63) / 64), dim3(1024), 0, S_(stream), CA1Ptr(x), partial,
                     ns, rows, C, eps, gamma, beta, mean, rstd, scale, shift);
  return dspn::check_launch("bn_stats");
}

#ifndef DSPN_HALF
size_t dspn_bn_tiles_workspace_bytes(int tiles, int C) { return bn_tiles_workspace_bytes(tiles, C); }
#endif
#ifndef DSPN_HALF
int dspn_bn_stats_from_tiles_f32(const float *tile_stats, int tiles, int tile_rows, long long rows, int C, float eps,
                                 const float *gamma, const float *beta, float *mean, float *rstd, float *scale,
                                 float *shift, const float *tile_minmax, int relu, float *out_absmax, float *out_absmin,
                                 float *out_chan_minmax, void *workspace, size_t workspace_bytes, void *stream) {
  DSPN_REQUIRE(tile_stats && beta && mean && rstd && scale && shift && tiles > 0 && tile_rows > 0 && C > 0 &&
                   rows > (long long)(tiles - 1) * tile_rows && rows <= (long long)tiles * tile_rows,
               "bn_stats_from_tiles: bad argument");
  DSPN_REQUIRE((tile_minmax != nullptr) == (out_absmax != nullptr), "bn_stats_from_tiles: tile_minmax and out_absmax go together");
  int mm_tiles = tiles;
  if (tiles >= tile_group_min() && workspace && workspace_bytes >= bn_tiles_workspace_bytes(tiles, C)) {
    const int groups = (tiles + kTileGroup - 1) / kTileGroup;
    float *grouped = static_cast<float *>(workspace);
    float *grouped_mm = tile_minmax ? grouped + 2 * (size_t)groups * C : nullptr;
    hipLaunchKernelGGL(tile_group_kernel<0>, dim3(groups, (C + 63) / 64), dim3(256), 0, S_(stream), tile_stats, tiles,
                       tile_rows, rows, C, grouped, tile_minmax, grouped_mm);
    tile_stats = grouped; tiles = groups; tile_rows *= kTileGroup;
    if (tile_minmax) { tile_minmax = grouped_mm; mm_tiles = groups; }
  }
  hipLaunchKernelGGL(bn_stats_tiles_final_kernel, dim3((C + 15) / 16), dim3(1024), 0, S_(stream), tile_stats, tiles,
                     tile_rows, rows, C, eps, gamma, beta, mean, rstd, scale, shift, tile_minmax, mm_tiles, relu,
                     reinterpret_cast<unsigned *>(out_absmax), reinterpret_cast<unsigned *>(tile_minmax ? out_absmin : nullptr),
                     tile_minmax ? out_chan_minmax : nullptr);
  return dspn::check_launch("bn_stats_from_tiles");
}
#endif
int DSPN_FN(dspn_bn_apply)(const st_t *x, const float *scale, const float *shift, st_t *y, long long rows,
                      int C, int relu, float *out_absmax, void *stream) {
  DSPN_REQUIRE(x && scale && shift && y, "bn_apply: null pointer");
  DSPN_REQUIRE(rows > 0 && C > 0 && C % 4 == 0, "bn_apply: C must be a positive multiple of 4");
  const long long n4 = rows * (C / 4);
#ifdef DSPN_HALF
  if (C % 8 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0) {
    hipLaunchKernelGGL(bn_apply8_kernel, dim3(grid_for(n4 / 2)), dim3(kT), 0, S_(stream),
                       reinterpret_cast<const dspn::u32x4_t *>(x), reinterpret_cast<const float4 *>(scale),
                       reinterpret_cast<const float4 *>(shift), reinterpret_cast<dspn::u32x4_t *>(y), n4 / 2, C / 8, relu);
    return dspn::check_launch("bn_apply");
  }
#endif
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(n4)), dim3(kT), 0, S_(stream),
                     CA4Ptr(x), reinterpret_cast<const float4 *>(scale),
                     reinterpret_cast<const float4 *>(shift), A4Ptr(y), n4, C / 4,
                     relu, dspn::kHalf ? nullptr : reinterpret_cast<unsigned *>(out_absmax));
  return dspn::check_launch("bn_apply");
}

#ifndef DSPN_HALF
int dspn_absmax_affine_bound_f32(const float *scale, const float *shift, int C, const float *x_absmax, float *out,
                                 void *stream) {
  DSPN_REQUIRE(scale && shift && x_absmax && out && C > 0, "absmax_affine_bound: bad argument");
  hipLaunchKernelGGL(absmax_affine_bound_kernel, dim3(1), dim3(256), 0, S_(stream), scale, shift, C, x_absmax,
                     reinterpret_cast<unsigned *>(out));
  return dspn::check_launch("absmax_affine_bound");
}
#endif

int DSPN_FN(dspn_bn_backward)(const st_t *x, const float *scale, const float *shift, const st_t *dy,
                         const float *mean, const float *rstd, const float *gamma, st_t *dx,
                         float *dgamma, float *dbeta, long long rows, int C, int relu, int accumulate,
                         float *dx_absmax, void *workspace, size_t workspace_bytes, void *stream) {
  DSPN_REQUIRE(x && dy && mean && rstd && dx && workspace, "bn_backward: null pointer");
  DSPN_REQUIRE(!relu || (scale && shift), "bn_backward: relu needs the forward scale/shift");
  DSPN_REQUIRE(rows > 0 && C > 0 && C % 4 == 0, "bn_backward: C must be a positive multiple of 4");
  if (workspace_bytes < bn_workspace_bytes(rows, C))
    return dspn::fail(DSPN_ERR_WORKSPACE_, "bn_backward: workspace too small");
  const int C4 = C / 4, CL = std::min(C4, 64), ns = bn_slabs(rows);
  float *partial = static_cast<float *>(workspace);
  float *coef = partial + (size_t)ns * 2 * C;
  hipLaunchKernelGGL(bn_bwd_partial_kernel<false>, dim3(ns, (C4 + CL - 1) / CL), dim3(kT),
                     sizeof(float4) * 2 * kT, S_(stream), CA4Ptr(x),
                     reinterpret_cast<const float4 *>(scale), reinterpret_cast<const float4 *>(shift),
                     CA4Ptr(dy), mean, rstd, rows, C4, CL, relu, partial, slab_rows_for(rows), PoolGrad{});
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((C + 63) / 64), dim3(1024), 0, S_(stream), partial, ns, C,
                     1.0 / (double)rows, mean, rstd, gamma, coef, dgamma, dbeta, nullptr, nullptr, nullptr,
                     static_cast<unsigned *>(nullptr));
  const long long n4 = rows * C4;
#ifdef DSPN_HALF
  if (C % 8 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0) {
    int fixed8 = 0, u4 = 0;
    const int grid8 = grid_fixed_channel(n4 / 2, C / 8, &fixed8, &u4);
    hipLaunchKernelGGL(u4 ? bn_bwd_apply8_kernel<4> : bn_bwd_apply8_kernel<1>, dim3(grid8), dim3(kT), 0, S_(stream),
                       reinterpret_cast<const dspn::u32x4_t *>(x), reinterpret_cast<const float4 *>(scale),
                       reinterpret_cast<const float4 *>(shift), reinterpret_cast<const dspn::u32x4_t *>(dy),
                       reinterpret_cast<const float4 *>(coef), reinterpret_cast<dspn::u32x4_t *>(dx), n4 / 2, C / 8, relu,
                       accumulate, fixed8);
    return dspn::check_launch("bn_backward");
  }
#endif
  int fixed4 = 0, u4 = 0;
  const int grid4 = grid_fixed_channel(n4, C4, &fixed4, &u4);
  hipLaunchKernelGGL(u4 ? bn_bwd_apply_kernel<4> : bn_bwd_apply_kernel<1>, dim3(grid4), dim3(kT), 0, S_(stream),
                     CA4Ptr(x), reinterpret_cast<const float4 *>(scale),
                     reinterpret_cast<const float4 *>(shift), CA4Ptr(dy),
                     reinterpret_cast<const float4 *>(coef), A4Ptr(dx), n4, C4, relu,
                     accumulate, dspn::kHalf ? nullptr : reinterpret_cast<unsigned *>(dx_absmax), fixed4);
  return dspn::check_launch("bn_backward");
}

#ifndef DSPN_HALF
int dspn_bn_backward_maxpool_f32(const float *x, const float *scale, const float *shift, const float *dy_pool,
                                 const unsigned char *argmax, int N, int H, int W, int C, int k, int stride, int pad, int Ho,
                                 int Wo, const float *mean, const float *rstd, const float *gamma, float *dx, float *dgamma,
                                 float *dbeta, int relu, float *dx_absmax, void *workspace, size_t workspace_bytes, void *stream) {
  DSPN_REQUIRE(x && dy_pool && argmax && mean && rstd && dx && dbeta && workspace, "bn_backward_maxpool: null pointer");
  DSPN_REQUIRE(!relu || (scale && shift), "bn_backward_maxpool: relu needs the forward scale/shift");
  DSPN_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && k > 0 && k * k <= 255 && stride > 0 && pad >= 0 && Ho > 0 && Wo > 0,
               "bn_backward_maxpool: bad geometry (C %% 4 == 0, k * k <= 255)");
  const long long rows = (long long)N * H * W;
  DSPN_REQUIRE(rows * (C / 4) < (1ll << 31), "bn_backward_maxpool: N * H * W * C / 4 must stay below 2^31");
  if (workspace_bytes < bn_workspace_bytes(rows, C))
    return dspn::fail(DSPN_ERR_WORKSPACE_, "bn_backward_maxpool: workspace too small (dspn_bn_workspace_bytes)");
  const int C4 = C / 4, CL = std::min(C4, 64), ns = bn_slabs(rows);
  float *partial = static_cast<float *>(workspace);
  float *coef = partial + (size_t)ns * 2 * C;
  const PoolGrad pool{reinterpret_cast<const uchar4 *>(argmax), reinterpret_cast<const float4 *>(dy_pool), H, W, k, stride, pad, Ho, Wo};
  hipLaunchKernelGGL(bn_bwd_partial_kernel<true>, dim3(ns, (C4 + CL - 1) / CL), dim3(kT), sizeof(float4) * 2 * kT, S_(stream),
                     CA4Ptr(x), reinterpret_cast<const float4 *>(scale), reinterpret_cast<const float4 *>(shift),
                     CA4Ptr(x), mean, rstd, rows, C4, CL, relu, partial, slab_rows_for(rows), pool);
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((C + 63) / 64), dim3(1024), 0, S_(stream), partial, ns, C,
                     1.0 / (double)rows, mean, rstd, gamma, coef, dgamma, dbeta, nullptr, nullptr, nullptr,
                     static_cast<unsigned *>(nullptr));
  const long long n4 = rows * C4;
  hipLaunchKernelGGL(bn_bwd_apply_pool_kernel, dim3(grid_for(n4)), dim3(256), 0, S_(stream), reinterpret_cast<const float4 *>(x),
                     reinterpret_cast<const float4 *>(scale), reinterpret_cast<const float4 *>(shift),
                     reinterpret_cast<const float4 *>(coef), reinterpret_cast<float4 *>(dx), n4, C4, relu,
                     reinterpret_cast<unsigned *>(dx_absmax), pool);
  return dspn::check_launch("bn_backward_maxpool");
}
#endif

// BatchNorm backward whose two reductions (sum dy', sum dy' xhat) were gathered per row tile by the data-gradient
// kernel that produced dy (bn_sums of dspn_conv2d_dgrad_bn_f32): finalize + apply only
#ifndef DSPN_HALF
int dspn_bn_apply_planes_f32(const float *x, const float *scale, const float *shift, void *y_planes, long long rows, int C,
                             int relu, const float *y_absmax, void *stream) {
  DSPN_REQUIRE(x && scale && shift && y_planes && y_absmax, "bn_apply_planes: null pointer (y_absmax is the block the planes are cut by)");
  DSPN_REQUIRE(rows > 0 && C > 0 && C % 32 == 0, "bn_apply_planes: C must be a positive multiple of 32");
  DSPN_REQUIRE(static_cast<const void *>(x) != y_planes, "bn_apply_planes: in place is not supported");
  const long long n4 = rows * (C / 4);
  int fixed4 = 0, u4 = 0;
  const int grid4 = grid_fixed_channel(n4, C / 4, &fixed4, &u4);
  hipLaunchKernelGGL(u4 ? bn_apply_planes_kernel<4> : bn_apply_planes_kernel<1>, dim3(grid4), dim3(kT), 0, S_(stream),
                     reinterpret_cast<const float4 *>(x), reinterpret_cast<const float4 *>(scale),
                     reinterpret_cast<const float4 *>(shift), static_cast<uint4 *>(y_planes), n4, C / 4, relu, y_absmax, fixed4);
  return dspn::check_launch("bn_apply_planes");
}
#endif

int DSPN_FN(dspn_bn_backward_from_sums)(const st_t *x, const float *scale, const float *shift, const st_t *dy,
                                   const float *mean, const float *rstd, const float *gamma, const float *tile_sums,
                                   int tiles, st_t *dx, float *dgamma, float *dbeta, long long rows, int C, int relu,
                                   int accumulate, float *dx_absmax, float *dx_absmin, const float *dy_absmax, const float *x_chan_minmax,
                                   int dx_planes_phase, void *workspace, size_t workspace_bytes, void *stream) {
  // round 6: bits 1 / 2 of the flag word split the call in two -- 2 = the finalize alone (the per-channel coefficients into the
  // workspace, dgamma / dbeta, the bound of dx), 4 = the apply pass alone from the coefficients an earlier call with bit 2 left
  // in the SAME workspace -- so that a caller can run the latency-bound finalize on a second stream beside a weight gradient
  // | 8 (with | 2): the finalize is PARKED as a job for the next weight-gradient launch on this stream (bn_final_job.h); the
  // apply-only call runs it stand-alone if no weight gradient took it
  const int dx_planes = dx_planes_phase & 1, phase = (dx_planes_phase >> 1) & 3, defer = (dx_planes_phase >> 3) & 1;
  DSPN_REQUIRE(dx_planes_phase >= 0 && dx_planes_phase < 16 && phase <= 2 && (!defer || phase == 1),
               "bn_backward_from_sums: flag word = dx_planes | 2 (finalize only, | 8: parked for the next weight gradient) | 4 (apply only)");
  DSPN_REQUIRE(x && dy && mean && rstd && dx && workspace && tile_sums && tiles > 0, "bn_backward_from_sums: null pointer");
  DSPN_REQUIRE(!dx_planes || (!dspn::kHalf && !accumulate && C % 32 == 0 && dx_absmax && dy_absmax && x_chan_minmax &&
                              static_cast<const void *>(dx) != static_cast<const void *>(dy) && static_cast<const void *>(dx) != static_cast<const void *>(x)),
               "bn_backward_from_sums: dx as piece planes needs float tensors, C %% 32 == 0, no accumulation, dx apart from x and dy, "
               "and dx_absmax / dy_absmax / x_chan_minmax (the bound of dx is formed from the last two)");
  DSPN_REQUIRE(!relu || (scale && shift), "bn_backward_from_sums: relu needs the forward scale/shift");
  DSPN_REQUIRE(rows > 0 && C > 0 && C % 4 == 0, "bn_backward_from_sums: C must be a positive multiple of 4");
  if (workspace_bytes < sizeof(float) * 3 * (size_t)C)
    return dspn::fail(DSPN_ERR_WORKSPACE_, "bn_backward_from_sums: workspace too small (3*C floats)");
  const int C4 = C / 4;
  float *coef = static_cast<float *>(workspace);
  const bool group_first = tiles >= tile_group_min() && workspace_bytes >= sizeof(float) * 3 * (size_t)C + bn_tiles_workspace_bytes(tiles, C);
  if (defer) {
    if (group_first) {      // (the group level needs hundreds of workgroups: a launch of its own, now)
      const int groups = (tiles + kTileGroup - 1) / kTileGroup;
      float *grouped = coef + 3 * (size_t)C;
      hipLaunchKernelGGL(tile_group_kernel<1>, dim3(groups, (C + 63) / 64), dim3(256), 0, S_(stream), tile_sums, tiles, 1,
                         (long long)tiles, C, grouped, static_cast<const float *>(nullptr), static_cast<float *>(nullptr));
      tile_sums = grouped; tiles = groups;
    }
    dspn::BnFinalJob j;
    j.tile_sums = tile_sums; j.tiles = tiles; j.C = C;
    j.blocks = (C + dspn::kBnJobChannels - 1) / dspn::kBnJobChannels;
    j.inv_rows = 1.0 / (double)rows; j.mean = mean; j.rstd = rstd; j.gamma = gamma; j.coef = coef; j.dgamma = dgamma; j.dbeta = dbeta;
    j.dy_absmax = dx_planes ? dy_absmax : nullptr; j.x_minmax = dx_planes ? x_chan_minmax : nullptr;
    j.dx_bound = dx_planes ? reinterpret_cast<unsigned *>(dx_absmax) : nullptr;
    j.dx_bound_min = dx_planes ? reinterpret_cast<unsigned *>(dx_absmin) : nullptr;
    dspn::bn_job_defer(S_(stream), j);
    return dspn::check_launch("bn_backward_from_sums (parked)");
  }
  if (phase == 2)      // (jobs of this stream no weight gradient took: here, before the apply pass)
    for (dspn::BnFinalJob parked; dspn::bn_job_take(S_(stream), &parked);)
      hipLaunchKernelGGL(bn_final_job_kernel, dim3(parked.blocks), dim3(256), 0, S_(stream), parked);
  if (phase != 2) {
  if (group_first) {
    const int groups = (tiles + kTileGroup - 1) / kTileGroup;
    float *grouped = coef + 3 * (size_t)C;
    hipLaunchKernelGGL(tile_group_kernel<1>, dim3(groups, (C + 63) / 64), dim3(256), 0, S_(stream), tile_sums, tiles, 1,
                       (long long)tiles, C, grouped, static_cast<const float *>(nullptr), static_cast<float *>(nullptr));
    tile_sums = grouped; tiles = groups;
  }
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((C + 63) / 64), dim3(1024), 0, S_(stream), tile_sums, tiles, C,
                     1.0 / (double)rows, mean, rstd, gamma, coef, dgamma, dbeta, dx_planes ? dy_absmax : nullptr,
                     dx_planes ? x_chan_minmax : nullptr, dx_planes ? reinterpret_cast<unsigned *>(dx_absmax) : nullptr,
                     dx_planes ? reinterpret_cast<unsigned *>(dx_absmin) : nullptr);
  }
  if (phase == 1) return dspn::check_launch("bn_backward_from_sums (finalize)");
  const long long n4 = rows * C4;
#ifndef DSPN_HALF
  if (dx_planes) {
    int fixed4 = 0, u4 = 0;
    const int grid4 = grid_fixed_channel(n4, C4, &fixed4, &u4);
    hipLaunchKernelGGL(u4 ? bn_bwd_apply_planes_kernel<4> : bn_bwd_apply_planes_kernel<1>, dim3(grid4), dim3(kT), 0, S_(stream),
                       CA4Ptr(x), reinterpret_cast<const float4 *>(scale), reinterpret_cast<const float4 *>(shift), CA4Ptr(dy),
                       reinterpret_cast<const float4 *>(coef), reinterpret_cast<uint4 *>(dx), n4, C4, relu, dx_absmax, fixed4);
    return dspn::check_launch("bn_backward_from_sums");
  }
#endif
#ifdef DSPN_HALF
  if (C % 8 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0) {
    int fixed8 = 0, u4 = 0;
    const int grid8 = grid_fixed_channel(n4 / 2, C / 8, &fixed8, &u4);
    hipLaunchKernelGGL(u4 ? bn_bwd_apply8_kernel<4> : bn_bwd_apply8_kernel<1>, dim3(grid8), dim3(kT), 0, S_(stream),
                       reinterpret_cast<const dspn::u32x4_t *>(x), reinterpret_cast<const float4 *>(scale),
                       reinterpret_cast<const float4 *>(shift), reinterpret_cast<const dspn::u32x4_t *>(dy),
                       reinterpret_cast<const float4 *>(coef), reinterpret_cast<dspn::u32x4_t *>(dx), n4 / 2, C / 8, relu,
                       accumulate, fixed8);
    return dspn::check_launch("bn_backward");
  }
#endif
  int fixed4 = 0, u4 = 0;
  const int grid4 = grid_fixed_channel(n4, C4, &fixed4, &u4);
  hipLaunchKernelGGL(u4 ? bn_bwd_apply_kernel<4> : bn_bwd_apply_kernel<1>, dim3(grid4), dim3(kT), 0, S_(stream),
                     CA4Ptr(x), reinterpret_cast<const float4 *>(scale),
                     reinterpret_cast<const float4 *>(shift), CA4Ptr(dy),
                     reinterpret_cast<const float4 *>(coef), A4Ptr(dx), n4, C4, relu,
                     accumulate, dspn::kHalf ? nullptr : reinterpret_cast<unsigned *>(dx_absmax), fixed4);
  return dspn::check_launch("bn_backward_from_sums");
}

int DSPN_FN(dspn_add)(const st_t *a, const st_t *b, st_t *out, long long n, void *stream) {
  DSPN_REQUIRE(a && b && out && n >= 0, "add: bad argument");
  const long long n4 = n / 4;
  if (n4 > 0)
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n4)), dim3(kT), 0, S_(stream),
                       CA4Ptr(a), CA4Ptr(b),
                       A4Ptr(out), n4);
  if (n4 * 4 < n)
    hipLaunchKernelGGL(add_tail_kernel, dim3(1), dim3(64), 0, S_(stream), CA1Ptr(a), CA1Ptr(b), A1Ptr(out), n4 * 4, n);
  return dspn::check_launch("add");
}

int DSPN_FN(dspn_relu_backward)(const st_t *y, const st_t *dy, st_t *dx, long long n, int accumulate,
                           void *stream) {
  DSPN_REQUIRE(y && dy && dx && n >= 0, "relu_backward: bad argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n)), dim3(kT), 0, S_(stream), CA1Ptr(y), CA1Ptr(dy), A1Ptr(dx), n, accumulate);
  return dspn::check_launch("relu_backward");
}

/* dx = (y > 0) ? dy : 0 (dx may alias dy) and out[c] = sum over rows of dx[:, c], c < C, in one pass; dx_absmax (float tensors,
 * optional): the magnitude block of dx as stored (max into a block the caller zeroed) */
int DSPN_FN(dspn_relu_backward_colsum)(const st_t *y, const st_t *dy, st_t *dx, long long rows, int C, int ld, float *out,
                                  float *dx_absmax, void *workspace, size_t workspace_bytes, void *stream) {
  DSPN_REQUIRE(!dx_absmax || !dspn::kHalf, "relu_backward_colsum: dx_absmax is for float tensors");
  DSPN_REQUIRE(y && dy && dx && out && workspace && rows > 0 && C > 0 && ld >= C && ld % 4 == 0,
               "relu_backward_colsum: bad argument");
  const int sr = colsum_slab_rows(rows);
  const int ns = (int)((rows + sr - 1) / sr);
  if (workspace_bytes < sizeof(float) * (size_t)ns * C)
    return dspn::fail(DSPN_ERR_WORKSPACE_, "relu_backward_colsum: workspace too small (dspn_colsum_workspace_bytes)");
  const int ld4 = ld / 4, CL = std::min(ld4, 64);
  float *partial = static_cast<float *>(workspace);
  hipLaunchKernelGGL(relu_bwd_colsum_partial_kernel, dim3(ns, (ld4 + CL - 1) / CL), dim3(kT), sizeof(float4) * kT,
                     S_(stream), CA4Ptr(y), CA4Ptr(dy),
                     A4Ptr(dx), rows, ld4, CL, partial, C, sr, reinterpret_cast<unsigned *>(dx_absmax));
  hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 15) / 16), dim3(1024), 0, S_(stream), partial, ns, C, out);
  return dspn::check_launch("relu_backward_colsum");
}

#ifndef DSPN_HALF
int dspn_fill_f32(float *p, float v, long long n, void *stream) {
  DSPN_REQUIRE(p && n >= 0, "fill: bad argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(kT), 0, S_(stream), p, v, n);
  return dspn::check_launch("fill");
}
#endif

#ifndef DSPN_HALF
size_t dspn_colsum_workspace_bytes(long long rows, int C) {
  if (rows <= 0 || C <= 0) return 0;
  const int sr = colsum_slab_rows(rows);
  return sizeof(float) * (size_t)((rows + sr - 1) / sr) * C;
}
#endif
int DSPN_FN(dspn_colsum)(const st_t *a, long long rows, int C, int ld, float *out, void *workspace,
                    size_t workspace_bytes, void *stream) {
  DSPN_REQUIRE(a && out && workspace && rows > 0 && C > 0 && ld >= C, "colsum: bad argument");
  const int sr = colsum_slab_rows(rows);
  const int ns = (int)((rows + sr - 1) / sr);
  if (workspace_bytes < sizeof(float) * (size_t)ns * C)
    return dspn::fail(DSPN_ERR_WORKSPACE_, "colsum: workspace too small");
  float *partial = static_cast<float *>(workspace);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(ns), dim3(kT), 0, S_(stream), CA1Ptr(a), rows, C, ld, partial, sr);
  hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 15) / 16), dim3(1024), 0, S_(stream), partial, ns, C, out);
  return dspn::check_launch("colsum");
}

int DSPN_FN(dspn_nchw_to_nhwc)(const float *src, st_t *dst, int N, int C, int H, int W, int Cp, void *stream) {
  DSPN_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && Cp >= C, "nchw_to_nhwc: bad argument");
  const long long HW = (long long)H * W, total = HW * N;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for(total)), dim3(kT), 0, S_(stream), src, A1Ptr(dst), C, HW, total, Cp);
  return dspn::check_launch("nchw_to_nhwc");
}
#ifndef DSPN_HALF
int dspn_nhwc_to_nchw_f32(const float *src, float *dst, int N, int C, int H, int W, int Cp, void *stream) {
  DSPN_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && Cp >= C, "nhwc_to_nchw: bad argument");
  const long long HW = (long long)H * W, total = HW * N * C;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for(total)), dim3(kT), 0, S_(stream), src, dst, C, HW, total, Cp);
  return dspn::check_launch("nhwc_to_nchw");
}
#endif

int DSPN_FN(dspn_copy_block)(const st_t *src, st_t *dst, int samples, long long rows_per_sample, int C,
                        long long src_sample_stride, int lds, int soff, long long dst_sample_stride,
                        int ldd, int doff, int accumulate, void *stream) {
  DSPN_REQUIRE(src && dst && samples > 0 && rows_per_sample > 0 && C > 0, "copy_block: bad argument");
  const long long total = (long long)samples * rows_per_sample * C;
  constexpr int E = 16 / (int)sizeof(st_t);      // elements per 16-byte unit
  if (!accumulate && C % E == 0 && lds % E == 0 && ldd % E == 0 && soff % E == 0 && doff % E == 0 &&
      src_sample_stride % E == 0 && dst_sample_stride % E == 0 &&
      ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
    hipLaunchKernelGGL(copy_block_vec_kernel, dim3(grid_for(total / E)), dim3(kT), 0, S_(stream),
                       reinterpret_cast<const dspn::u32x4_t *>(src), reinterpret_cast<dspn::u32x4_t *>(dst), rows_per_sample,
                       C / E, src_sample_stride / E, lds / E, soff / E, dst_sample_stride / E, ldd / E, doff / E, total / E);
    return dspn::check_launch("copy_block");
  }
  hipLaunchKernelGGL((copy_block_kernel<CA1Ptr, A1Ptr>), dim3(grid_for(total)), dim3(kT), 0, S_(stream), CA1Ptr(src),
                     A1Ptr(dst), rows_per_sample, C, src_sample_stride, lds, soff, dst_sample_stride, ldd, doff,
                     total, accumulate);
  return dspn::check_launch("copy_block");
}
#ifndef DSPN_HALF
int dspn_copy_block_batch_f32(const void *table, int n, long long total, void *stream) {
  DSPN_REQUIRE(table && n > 0 && total > 0, "copy_block_batch: bad argument");
  hipLaunchKernelGGL(copy_block_batch_kernel, dim3(grid_for(total)), dim3(kT), 0, S_(stream),
                     static_cast<const CopyDesc *>(table), n, total);
  return dspn::check_launch("copy_block_batch");
}
#endif
#ifdef DSPN_HALF
/* the same strided block copy between storage types: bf16 -> float (SSD head maps into the float loss inputs) and
 * float -> bf16 (their gradients back into the per-map gradient tensors) */
int dspn_copy_block_bf16_f32(const st_t *src, float *dst, int samples, long long rows_per_sample, int C,
                             long long src_sample_stride, int lds, int soff, long long dst_sample_stride,
                             int ldd, int doff, int accumulate, void *stream) {
  DSPN_REQUIRE(src && dst && samples > 0 && rows_per_sample > 0 && C > 0, "copy_block: bad argument");
  const long long total = (long long)samples * rows_per_sample * C;
  hipLaunchKernelGGL((copy_block_kernel<CA1Ptr, float *>), dim3(grid_for(total)), dim3(kT), 0, S_(stream), CA1Ptr(src),
                     dst, rows_per_sample, C, src_sample_stride, lds, soff, dst_sample_stride, ldd, doff, total,
                     accumulate);
  return dspn::check_launch("copy_block");
}
int dspn_copy_block_f32_bf16(const float *src, st_t *dst, int samples, long long rows_per_sample, int C,
                             long long src_sample_stride, int lds, int soff, long long dst_sample_stride,
                             int ldd, int doff, int accumulate, void *stream) {
  DSPN_REQUIRE(src && dst && samples > 0 && rows_per_sample > 0 && C > 0, "copy_block: bad argument");
  const long long total = (long long)samples * rows_per_sample * C;
  hipLaunchKernelGGL((copy_block_kernel<const float *, A1Ptr>), dim3(grid_for(total)), dim3(kT), 0, S_(stream), src,
                     A1Ptr(dst), rows_per_sample, C, src_sample_stride, lds, soff, dst_sample_stride, ldd, doff, total,
                     accumulate);
  return dspn::check_launch("copy_block");
}
#endif

#ifndef DSPN_HALF
int dspn_transpose_bnc_f32(const float *src, float *dst, int B, int N, int C, void *stream) {
  DSPN_REQUIRE(src && dst && B > 0 && N > 0 && C > 0, "transpose_bnc: bad argument");
  const long long total = (long long)B * N * C;
  hipLaunchKernelGGL(transpose_bnc_kernel, dim3(grid_for(total)), dim3(kT), 0, S_(stream), src, dst, N, C, total);
  return dspn::check_launch("transpose_bnc");
}
#endif

int DSPN_FN(dspn_tap_sum)(const st_t *z, const float *bias, st_t *y, int N, int H, int W, int Cout, int ldy,
                     int ldz, int R, int S, int pad_h, int pad_w, void *stream) {
  DSPN_REQUIRE(z && y && N > 0 && H > 0 && W > 0 && Cout > 0 && ldy >= Cout && ldz >= Cout * R * S,
               "tap_sum: bad argument");
  const long long total = (long long)N * H * W * ldy;
  hipLaunchKernelGGL(tap_sum_kernel_, dim3(grid_for(total, kT, 65535)), dim3(kT), 0, S_(stream), CA1Ptr(z), bias, A1Ptr(y), H, W,
                     Cout, ldy, ldz, R, S, pad_h, pad_w, total);
  return dspn::check_launch("tap_sum");
}
int DSPN_FN(dspn_tap_spread)(const st_t *dy, st_t *dz, int N, int H, int W, int Cout, int ldy, int ldz, int R,
                        int S, int pad_h, int pad_w, void *stream) {
  DSPN_REQUIRE(dy && dz && N > 0 && H > 0 && W > 0 && Cout > 0 && ldy >= Cout && ldz >= Cout * R * S,
               "tap_spread: bad argument");
  const long long total = (long long)N * H * W * ldz;
  hipLaunchKernelGGL(tap_spread_kernel_, dim3(grid_for(total, kT, 65535)), dim3(kT), 0, S_(stream), CA1Ptr(dy), A1Ptr(dz), H, W,
                     Cout, ldy, ldz, R, S, pad_h, pad_w, total);
  return dspn::check_launch("tap_spread");
}
int DSPN_FN(dspn_maxpool_forward)(const st_t *x, st_t *y, unsigned char *argmax, int N, int H, int W, int C, int k,
                             int stride, int pad, int Ho, int Wo, void *stream) {
  DSPN_REQUIRE(x && y && C % 4 == 0 && N > 0, "maxpool_forward: bad argument");
  DSPN_REQUIRE(!argmax || k * k < 255, "maxpool_forward: argmax record needs k*k < 255");
  const long long total = (long long)N * Ho * Wo * (C / 4);
  if (argmax)
    hipLaunchKernelGGL((maxpool_fwd_kernel<true, false>), dim3(grid_for(total)), dim3(kT), 0, S_(stream),
                       CA4Ptr(x), A4Ptr(y),
                       reinterpret_cast<uchar4 *>(argmax), H, W, C / 4, k, stride, pad, Ho, Wo, total, nullptr, nullptr, 0, nullptr);
  else
    hipLaunchKernelGGL((maxpool_fwd_kernel<false, false>), dim3(grid_for(total)), dim3(kT), 0, S_(stream),
                       CA4Ptr(x), A4Ptr(y),
                       static_cast<uchar4 *>(nullptr), H, W, C / 4, k, stride, pad, Ho, Wo, total, nullptr, nullptr, 0, nullptr);
  return dspn::check_launch("maxpool_forward");
}

/* max pooling of (relu)(x * in_scale[c] + in_shift[c]): the BatchNorm(+ReLU) in front of the pooling layer folded in */
int DSPN_FN(dspn_maxpool_forward_bn)(const st_t *x, const float *in_scale, const float *in_shift, int in_relu, st_t *y,
                                unsigned char *argmax, int N, int H, int W, int C, int k, int stride, int pad, int Ho, int Wo,
                                float *out_absmax, void *stream) {
  DSPN_REQUIRE(x && y && in_scale && in_shift && C % 4 == 0 && N > 0, "maxpool_forward_bn: bad argument");
  DSPN_REQUIRE(!argmax || k * k < 255, "maxpool_forward_bn: argmax record needs k*k < 255");
  const long long total = (long long)N * Ho * Wo * (C / 4);
  unsigned *am = dspn::kHalf ? nullptr : reinterpret_cast<unsigned *>(out_absmax);
  if (argmax)
    hipLaunchKernelGGL((maxpool_fwd_kernel<true, true>), dim3(grid_for(total)), dim3(kT), 0, S_(stream), CA4Ptr(x), A4Ptr(y),
                       reinterpret_cast<uchar4 *>(argmax), H, W, C / 4, k, stride, pad, Ho, Wo, total,
                       reinterpret_cast<const float4 *>(in_scale), reinterpret_cast<const float4 *>(in_shift), in_relu, am);
  else
    hipLaunchKernelGGL((maxpool_fwd_kernel<false, true>), dim3(grid_for(total)), dim3(kT), 0, S_(stream), CA4Ptr(x), A4Ptr(y),
                       static_cast<uchar4 *>(nullptr), H, W, C / 4, k, stride, pad, Ho, Wo, total,
                       reinterpret_cast<const float4 *>(in_scale), reinterpret_cast<const float4 *>(in_shift), in_relu, am);
  return dspn::check_launch("maxpool_forward_bn");
}
int DSPN_FN(dspn_maxpool_backward_argmax)(const unsigned char *argmax, const st_t *dy, st_t *dx, int N, int H,
                                     int W, int C, int k, int stride, int pad, int Ho, int Wo, void *stream) {
  DSPN_REQUIRE(argmax && dy && dx && N > 0 && C % 4 == 0, "maxpool_backward_argmax: bad argument");
  const long long total = (long long)N * H * W * (C / 4);
  hipLaunchKernelGGL(maxpool_bwd_idx_kernel, dim3(grid_for(total, kT, 65535)), dim3(kT), 0, S_(stream),
                     reinterpret_cast<const uchar4 *>(argmax), CA4Ptr(dy),
                     A4Ptr(dx), H, W, C / 4, k, stride, pad, Ho, Wo, total);
  return dspn::check_launch("maxpool_backward_argmax");
}
int DSPN_FN(dspn_maxpool_backward)(const st_t *x, const st_t *y, const st_t *dy, st_t *dx, int N, int H,
                              int W, int C, int k, int stride, int pad, int Ho, int Wo, void *stream) {
  DSPN_REQUIRE(x && y && dy && dx && N > 0 && C % 4 == 0, "maxpool_backward: bad argument");
  const long long total = (long long)N * H * W * (C / 4);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total, kT, 65535)), dim3(kT), 0, S_(stream),
                     CA4Ptr(x), CA4Ptr(y),
                     CA4Ptr(dy), A4Ptr(dx), H, W, C / 4, k,
                     stride, pad, Ho, Wo, total);
  return dspn::check_launch("maxpool_backward");
}
int DSPN_FN(dspn_avgpool_forward)(const st_t *x, st_t *y, int N, int H, int W, int C, int k, int Ho, int Wo,
                             void *stream) {
  DSPN_REQUIRE(x && y && C % 4 == 0 && k > 0 && Ho * k <= H && Wo * k <= W, "avgpool_forward: bad argument");
  const long long total = (long long)N * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(grid_for(total)), dim3(kT), 0, S_(stream),
                     CA4Ptr(x), A4Ptr(y), H, W, C / 4, k, Ho,
                     Wo, total);
  return dspn::check_launch("avgpool_forward");
}
int DSPN_FN(dspn_avgpool_backward)(const st_t *dy, st_t *dx, int N, int H, int W, int C, int k, int Ho, int Wo,
                              int accumulate, void *stream) {
  DSPN_REQUIRE(dy && dx && C % 4 == 0 && k > 0, "avgpool_backward: bad argument");
  const long long total = (long long)N * H * W * (C / 4);
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid_for(total)), dim3(kT), 0, S_(stream),
                     CA4Ptr(dy), A4Ptr(dx), H, W, C / 4, k,
                     Ho, Wo, total, accumulate);
  return dspn::check_launch("avgpool_backward");
}

int DSPN_FN(dspn_avgpool2d_forward)(const st_t *x, st_t *y, int N, int H, int W, int C, int k, int stride, int pad,
                               int Ho, int Wo, void *stream) {
  DSPN_REQUIRE(x && y && C % 4 == 0 && k > 0 && stride > 0 && pad >= 0 && pad < k, "avgpool2d_forward: bad argument");
  DSPN_REQUIRE(Ho == (H + 2 * pad - k) / stride + 1 && Wo == (W + 2 * pad - k) / stride + 1,
               "avgpool2d_forward: output size mismatch");
  const long long total = (long long)N * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(avgpool2d_fwd_kernel, dim3(grid_for(total)), dim3(kT), 0, S_(stream),
                     CA4Ptr(x), A4Ptr(y), H, W, C / 4, k, stride,
                     pad, Ho, Wo, total);
  return dspn::check_launch("avgpool2d_forward");
}
int DSPN_FN(dspn_avgpool2d_backward)(const st_t *dy, st_t *dx, int N, int H, int W, int C, int k, int stride, int pad,
                                int Ho, int Wo, int accumulate, void *stream) {
  DSPN_REQUIRE(dy && dx && C % 4 == 0 && k > 0 && stride > 0 && pad >= 0 && pad < k, "avgpool2d_backward: bad argument");
  const long long total = (long long)N * H * W * (C / 4);
  hipLaunchKernelGGL(avgpool2d_bwd_kernel, dim3(grid_for(total)), dim3(kT), 0, S_(stream),
                     CA4Ptr(dy), A4Ptr(dx), H, W, C / 4, k, stride,
                     pad, Ho, Wo, total, accumulate);
  return dspn::check_launch("avgpool2d_backward");
}

#ifndef DSPN_HALF
int dspn_bilinear_forward_f32(const float *x, float *y, int N, int Hin, int Win, int C, int Ho, int Wo,
                              int ldo, int coff, void *stream) {
  DSPN_REQUIRE(x && y && C % 4 == 0 && ldo % 4 == 0 && coff % 4 == 0 && coff + C <= ldo, "bilinear_forward: bad argument");
  const long long total = (long long)N * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(bilinear_fwd_kernel<false>, dim3(grid_for(total)), dim3(kT), 0, S_(stream),
                     reinterpret_cast<const float4 *>(x), y, Hin, Win, C / 4, Ho, Wo, ldo, coff, total);
  return dspn::check_launch("bilinear_forward");
}
#endif
#ifndef DSPN_HALF
int dspn_bilinear_forward_acc_f32(const float *x, float *y, int N, int Hin, int Win, int C, int Ho, int Wo,
                                  int ldo, int coff, void *stream) {
  DSPN_REQUIRE(x && y && C % 4 == 0 && ldo % 4 == 0 && coff % 4 == 0 && coff + C <= ldo, "bilinear_forward_acc: bad argument");
  const long long total = (long long)N * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(bilinear_fwd_kernel<true>, dim3(grid_for(total)), dim3(kT), 0, S_(stream),
                     reinterpret_cast<const float4 *>(x), y, Hin, Win, C / 4, Ho, Wo, ldo, coff, total);
  return dspn::check_launch("bilinear_forward_acc");
}
#endif
#ifndef DSPN_HALF
int dspn_bilinear_backward_f32(const float *dy, float *dx, int N, int Hin, int Win, int C, int Ho, int Wo,
                               int ldo, int coff, void *stream) {
  DSPN_REQUIRE(dy && dx && C % 4 == 0 && ldo % 4 == 0 && coff % 4 == 0 && coff + C <= ldo, "bilinear_backward: bad argument");
  const long long total = (long long)N * Hin * Win * (C / 4);
  hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(grid_for(total)), dim3(kT), 0, S_(stream), dy,
                     reinterpret_cast<float4 *>(dx), Hin, Win, C / 4, Ho, Wo, ldo, coff, total);
  return dspn::check_launch("bilinear_backward");
}
#endif

#ifndef DSPN_HALF
size_t dspn_bilinear_backward_workspace_bytes(int N, int Win, int C, int Ho) {
  if (N <= 0 || Win <= 0 || C <= 0 || Ho <= 0) return 0;
  return sizeof(float) * (size_t)N * Ho * Win * C;
}
#endif
#ifndef DSPN_HALF
/* separable two-pass backward (needs dspn_bilinear_backward_workspace_bytes of scratch) */
int dspn_bilinear_backward_ws_f32(const float *dy, float *dx, int N, int Hin, int Win, int C, int Ho, int Wo,
                                  int ldo, int coff, void *workspace, size_t workspace_bytes, void *stream) {
  DSPN_REQUIRE(dy && dx && workspace && C % 4 == 0 && ldo % 4 == 0 && coff % 4 == 0 && coff + C <= ldo,
               "bilinear_backward: bad argument");
  if (workspace_bytes < dspn_bilinear_backward_workspace_bytes(N, Win, C, Ho))
    return dspn::fail(DSPN_ERR_WORKSPACE_, "bilinear_backward: workspace too small");
  const long long t1 = (long long)N * Ho * Win * (C / 4), t2 = (long long)N * Hin * Win * (C / 4);
  hipLaunchKernelGGL(bilinear_bwd_w_kernel, dim3(grid_for(t1, kT, 65535)), dim3(kT), 0, S_(stream), dy,
                     static_cast<float4 *>(workspace), Win, C / 4, Ho, Wo, ldo, coff, t1);
  hipLaunchKernelGGL(bilinear_bwd_h_kernel, dim3(grid_for(t2, kT, 65535)), dim3(kT), 0, S_(stream),
                     static_cast<const float4 *>(workspace), reinterpret_cast<float4 *>(dx), Hin, Win, C / 4, Ho, t2);
  return dspn::check_launch("bilinear_backward_ws");
}
#endif

#ifndef DSPN_HALF
int dspn_seg_counts_f32(const float *scores, const float *label, long long rows, int C, int ld,
                        unsigned long long *counts, void *stream) {
  DSPN_REQUIRE(scores && label && counts && rows > 0 && C > 0 && C <= kMaxSegC && ld >= C, "seg_counts: bad argument (C <= 64)");
  (void)hipMemsetAsync(counts, 0, sizeof(unsigned long long) * (3 * C + 1), S_(stream));
  hipLaunchKernelGGL(seg_counts_kernel, dim3(grid_for(rows, kT, 2048)), dim3(kT), 0, S_(stream), scores, label, rows, C, ld,
                     counts);
  return dspn::check_launch("seg_counts");
}
#endif

#ifndef DSPN_HALF
int dspn_seg_upsample_argmax_f32(const float *prob, unsigned char *out, int N, int Hin, int Win, int C, int ld,
                                 int Ho, int Wo, void *stream) {
  DSPN_REQUIRE(prob && out && N > 0 && Hin > 0 && Win > 0 && Ho > 0 && Wo > 0 && C > 0 && C <= 256 && ld >= C,
               "seg_upsample_argmax: bad argument (0 < C <= 256, ld >= C)");
  const long long total = (long long)N * Ho * ((Wo + 3) / 4);
  // float4 reads of the class vectors need 16-byte aligned rows that hold the rounded-up channel count
  const bool vec = ld % 4 == 0 && (C + 3) / 4 * 4 <= ld && reinterpret_cast<uintptr_t>(prob) % 16 == 0;
  if (vec)
    hipLaunchKernelGGL(seg_upsample_argmax_kernel<true>, dim3(grid_for(total, kT, 65535)), dim3(kT), 0, S_(stream), prob, out,
                       Hin, Win, C, ld, Ho, Wo, total);
  else
    hipLaunchKernelGGL(seg_upsample_argmax_kernel<false>, dim3(grid_for(total, kT, 65535)), dim3(kT), 0, S_(stream), prob, out,
                       Hin, Win, C, ld, Ho, Wo, total);
  return dspn::check_launch("seg_upsample_argmax");
}
#endif

int DSPN_FN(dspn_softmax_output)(const st_t *logits, const float *label, float *prob, st_t *grad,
                            long long rows, int C, int ld, float ignore_label, float grad_scale,
                            const float *valid_count, void *stream) {
  DSPN_REQUIRE(logits && prob && rows > 0 && C > 0 && C <= kMaxSoftmaxC && ld >= C && ld <= kMaxSoftmaxC,
               "softmax_output: bad argument (C <= ld <= 64)");
  DSPN_REQUIRE(!grad || label, "softmax_output: gradient needs labels");
  auto kern = ld <= 12 ? softmax_output_kernel<12> : ld <= 24 ? softmax_output_kernel<24> : softmax_output_kernel<kMaxSoftmaxC>;
  hipLaunchKernelGGL(kern, dim3(grid_for(rows, 128)), dim3(128), 0, S_(stream), CA1Ptr(logits), label,
                     prob, A1Ptr(grad), rows, C, ld, ignore_label, grad_scale, valid_count);
  return dspn::check_launch("softmax_output");
}
#ifndef DSPN_HALF
int dspn_count_f32(const float *a, long long n, int mode, float ref, float *out, void *stream) {
  DSPN_REQUIRE(a && out && n > 0 && n < (1ll << 24), "count: n must be in (0, 2^24)");
  hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(64), 0, S_(stream), out, 0.f, 1ll);
  hipLaunchKernelGGL(count_kernel, dim3(grid_for(n, kT, 1024)), dim3(kT), 0, S_(stream), a, n, mode, ref, out);
  return dspn::check_launch("count");
}
#endif
#ifndef DSPN_HALF
int dspn_smooth_l1_forward_f32(const float *pred, const float *target, const float *mask, float *loss,
                               long long n, void *stream) {
  DSPN_REQUIRE(pred && target && mask && loss && n > 0, "smooth_l1_forward: bad argument");
  hipLaunchKernelGGL(smooth_l1_fwd_kernel, dim3(grid_for(n)), dim3(kT), 0, S_(stream), pred, target, mask, loss, n);
  return dspn::check_launch("smooth_l1_forward");
}
#endif
#ifndef DSPN_HALF
int dspn_smooth_l1_backward_f32(const float *pred, const float *target, const float *mask, float *grad,
                                long long n, float grad_scale, const float *valid_count, void *stream) {
  DSPN_REQUIRE(pred && target && mask && grad && valid_count && n > 0, "smooth_l1_backward: bad argument");
  hipLaunchKernelGGL(smooth_l1_bwd_kernel, dim3(grid_for(n)), dim3(kT), 0, S_(stream), pred, target, mask, grad,
                     n, grad_scale, valid_count);
  return dspn::check_launch("smooth_l1_backward");
}
#endif
#ifndef DSPN_HALF
int dspn_cross_entropy_sum_f32(const float *prob, const float *label, long long rows, int C, int ld,
                               float ignore_label, float eps, float *out2, void *stream) {
  DSPN_REQUIRE(prob && label && out2 && rows > 0 && C > 0 && ld >= C, "cross_entropy_sum: bad argument");
  hipLaunchKernelGGL(ce_sum_kernel, dim3(1), dim3(1024), 0, S_(stream), prob, label, rows, C, ld, ignore_label, eps, out2);
  return dspn::check_launch("cross_entropy_sum");
}
#endif
#ifndef DSPN_HALF
int dspn_sum_f32(const float *a, long long n, float *out, void *stream) {
  DSPN_REQUIRE(a && out && n > 0, "sum: bad argument");
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, S_(stream), a, n, out);
  return dspn::check_launch("sum");
}
#endif

#ifndef DSPN_HALF
int dspn_sgd_momentum_f32(float *w, const float *grad, float *mom, long long n, float lr, float momentum,
                          float wd, float rescale, void *stream) {
  DSPN_REQUIRE(w && grad && mom && n > 0 && n % 4 == 0, "sgd_momentum: n must be a positive multiple of 4");
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n / 4)), dim3(kT), 0, S_(stream), reinterpret_cast<float4 *>(w),
                     reinterpret_cast<const float4 *>(grad), reinterpret_cast<float4 *>(mom), n / 4, lr,
                     momentum, wd, rescale);
  return dspn::check_launch("sgd_momentum");
}
#endif

}  // extern "C"
