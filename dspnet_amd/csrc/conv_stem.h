// The stem convolution of the ResNet backbones (symbol/resnet.py:94: Convolution(kernel 7x7, stride 2, pad 3) on the
// BatchNorm'ed image, 4 physical input channels -> 64) in the two-piece math, as a kernel of its own (round 5).
// Included by conv_wide.hip inside its anonymous namespace.
//
// In conv_nt_kernel this layer is the one non-uniform-tap shape of the headline graph (a k-step of 32 values straddles 8 taps:
// every lane decodes its own tap) on 64 x 64 tiles: 32 768 workgroups with a statistics epilogue each, 0.56 - 0.63 ms for
// 671 MB of HBM traffic (0.08 of the matrix peak, 1.1 TB/s).  Here the layout of the data does the im2col:
//   * with 4 channels per pixel, the 7 taps of ONE kernel row over output pixel (oy, ox) are 28 CONTIGUOUS values of input
//     row 2 oy + r - 3, starting at column 2 ox - 3 -- so a kernel row is one 32-deep k-step (an 8th tap with zero weights
//     pads it) whose A fragment is a 16-byte read of the row image at pixel 2 ox + 4 kk + 2 half: no index arithmetic at all;
//   * a workgroup walks DOWN a strip of output rows, 256 pixels wide: its LDS holds a ring of 9 input rows as fp16 piece
//     planes ([row][piece][pixel][4 channels]: cut once per element) and all 64 x 7 x 8 x 4 weights as pieces (57 KiB, cut in
//     the prologue); each output row needs two new input rows, requested before the row is multiplied and stored behind it;
//   * a wave owns 32 pixels x 64 output channels: 84 MFMAs per output row, accumulators straight to global memory (one
//     128-byte run per pixel and 32 channels), BatchNorm statistics and extremes per 64-pixel tile merged from two waves
//     with Chan's update -- the (mean, M2) / (min, max) tables have the layout dspn_conv2d_stats_layout gives for this shape.
// HBM floor at 5.5 TB/s: 0.12 ms (537 MB out, 134 MB in).

constexpr int kStemSeg = 256;              // output pixels of a row per workgroup
constexpr int kStemRing = 9;               // input rows resident (7 in use + the 2 being refilled)
constexpr int kStemRowPx = 2 * kStemSeg + 8;     // input pixels of a ring row: 2 x 256 + 5 halo, padded to a multiple of 4
constexpr int kStemRowBytes = kStemRowPx * 8;    // one piece of one row: 4 channels x 2 bytes per pixel
constexpr int kStemWRow = 7 * 2 * 64 + 16;       // bytes of one output channel's weights: [kernel row][piece][32 k] halves + 16: 32
                                                 // consecutive channels then sit on 16 different 16-byte slots (conflict-free ds_read_b128)
constexpr int kStemWBytes = 64 * kStemWRow;

__global__ __launch_bounds__(512, 2) void conv_stem_f16x2_kernel(
    const float *__restrict__ x, const float *__restrict__ w, float *__restrict__ y, int N, int H, int W, int Ho, int Wo,
    int rows_per_strip, const float *__restrict__ x_absmax, const float *__restrict__ w_absmax,
    float *__restrict__ stats, float *__restrict__ minmax) {
  extern __shared__ __attribute__((aligned(1024))) char ssm[];
  char *sW = ssm;                                        // weights as pieces
  char *sX = ssm + kStemWBytes;                          // ring: [slot][piece][pixel][4]
  float *sRed = reinterpret_cast<float *>(sX + kStemRing * 2 * kStemRowBytes);   // [8 waves][64][4]: mean, M2, min, max
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float sc_a = operand_scale(x_absmax), sc_b = operand_scale(w_absmax);
  const float inv_a = 1.f / sc_a, inv_b = 1.f / sc_b;
  const int segs = Wo / kStemSeg, strips = (Ho + rows_per_strip - 1) / rows_per_strip;
  int b = blockIdx.x;
  const int seg = b % segs; b /= segs;
  const int strip = b % strips;
  const int n = b / strips;
  const int oy0 = strip * rows_per_strip, oy1 = min(Ho, oy0 + rows_per_strip);
  const int ix0 = 2 * seg * kStemSeg - 3;                // input column of ring pixel 0

  // ---- weights: [cout][r][s][c] floats -> sW[(cout * 7 + r) * 2 + piece][s * 4 + c], s = 7 zero
  for (int i = tid; i < 64 * 7 * 8; i += 512) {
    const int s = i & 7, r = (i >> 3) % 7, k = i / 56;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s < 7) v = *reinterpret_cast<const float4 *>(w + ((k * 7 + r) * 7 + s) * 4);
    bf16x4 p0, p1;
    split2h(v, sc_b, p0, p1);
    repair_inf(p0, p1);
    *reinterpret_cast<bf16x4 *>(sW + k * kStemWRow + (r * 2 + 0) * 64 + s * 8) = p0;
    *reinterpret_cast<bf16x4 *>(sW + k * kStemWRow + (r * 2 + 1) * 64 + s * 8) = p1;
  }
  // ---- input rows: thread t moves pixel t (and t + 512: the ring row has 520) of a row; rows outside the image are zeros
  const float4 *xin = reinterpret_cast<const float4 *>(x) + (long long)n * H * W;
  auto load_px = [&](const int iy, const int p) __attribute__((always_inline)) {
    const int ix = ix0 + p;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p < kStemRowPx && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = xin[(long long)iy * W + ix];
    return v;
  };
  auto store_px = [&](const int slot, const int p, const float4 v) __attribute__((always_inline)) {
    if (p >= kStemRowPx) return;
    bf16x4 p0, p1;
    split2h(v, sc_a, p0, p1);
    *reinterpret_cast<bf16x4 *>(sX + (slot * 2 + 0) * kStemRowBytes + p * 8) = p0;
    *reinterpret_cast<bf16x4 *>(sX + (slot * 2 + 1) * kStemRowBytes + p * 8) = p1;
  };
  // ring slot of input row iy: (iy + 3) mod 9 relative to the strip's first row (iy = 2 oy0 - 3 sits in slot 0)
  const int iy_base = 2 * oy0 - 3;
  for (int j = 0; j < 7; ++j) {                          // the first output row's seven input rows
    store_px(j, tid, load_px(iy_base + j, tid));
    store_px(j, tid + 512, load_px(iy_base + j, tid + 512));
  }
  __syncthreads();

  const int frow = lane & 31, half = lane >> 5;
  const int px = wave * 32 + frow;                       // this lane's output pixel within the segment (A fragment row)
  f32x16 acc[2];
  for (int oy = oy0; oy < oy1; ++oy) {
    const int j0 = ((oy - oy0) * 2) % kStemRing;         // ring slot of this output row's first input row
    // the two input rows the NEXT output row adds, requested now, stored behind the MFMAs
    const int iy_new = 2 * oy + 4;
    float4 n0 = load_px(iy_new, tid), n1 = load_px(iy_new, tid + 512), n2 = load_px(iy_new + 1, tid), n3 = load_px(iy_new + 1, tid + 512);
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      int slot = j0 + r; slot = slot >= kStemRing ? slot - kStemRing : slot;
      const char *xa = sX + slot * 2 * kStemRowBytes + (2 * px + 2 * half) * 8;
      const char *wb = sW + frow * kStemWRow + r * 128 + half * 16;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const bf16x8 a0 = *reinterpret_cast<const bf16x8 *>(xa + kk * 32), a1 = *reinterpret_cast<const bf16x8 *>(xa + kStemRowBytes + kk * 32);
        bf16x8 b0[2], b1[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          b0[j] = *reinterpret_cast<const bf16x8 *>(wb + j * (32 * kStemWRow) + kk * 32);
          b1[j] = *reinterpret_cast<const bf16x8 *>(wb + j * (32 * kStemWRow) + 64 + kk * 32);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {      // x w = h0 g0 + h0 g1 + h1 g0, smallest terms first
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, b0[j]), acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, b1[j]), acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a0), __builtin_bit_cast(f16x8, b0[j]), acc[j], 0, 0, 0);
        }
      }
    }
    // ---- epilogue of the row.  C/D layout: col = lane & 31 (cout), row = (q & 3) + 8 (q >> 2) + 4 half (pixel of the wave's 32)
    const long long m_row = ((long long)n * Ho + oy) * Wo + seg * kStemSeg + wave * 32;
    float s1[2], mn[2], mx[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int co = j * 32 + frow;
      s1[j] = 0.f; mn[j] = __builtin_huge_valf(); mx[j] = -__builtin_huge_valf();
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float v = acc[j][q] * inv_a * inv_b;
        acc[j][q] = v;
        y[(m_row + (q & 3) + 8 * (q >> 2) + 4 * half) * 64 + co] = v;
        s1[j] += v; mn[j] = fminf(mn[j], v); mx[j] = fmaxf(mx[j], v);
      }
    }
    if (stats) {
      // per wave and channel: mean and M2 of its 32 pixels (two halves of 16, merged), then the two waves of a 64-pixel tile
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float mean16 = s1[j] * (1.f / 16.f);
        float m2 = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) { const float d = acc[j][q] - mean16; m2 += d * d; }
        const float mean_o = __shfl_xor(mean16, 32, 64), m2_o = __shfl_xor(m2, 32, 64);
        const float d = mean_o - mean16;
        const float mean32 = mean16 + 0.5f * d;                    // Chan: n_a = n_b = 16
        const float m2_32 = m2 + m2_o + d * d * 8.f;               // + d^2 n_a n_b / (n_a + n_b)
        const float mn32 = fminf(mn[j], __shfl_xor(mn[j], 32, 64)), mx32 = fmaxf(mx[j], __shfl_xor(mx[j], 32, 64));
        if (half == 0) *reinterpret_cast<float4 *>(sRed + (wave * 64 + j * 32 + frow) * 4) = make_float4(mean32, m2_32, mn32, mx32);
      }
    }
    __syncthreads();                                     // every wave has read this row's fragments (and left its statistics)
    if (stats && tid < 256) {
      const int tl = tid >> 6, co = tid & 63;            // 64-pixel tile of the segment, channel
      const float4 a = *reinterpret_cast<const float4 *>(sRed + ((2 * tl) * 64 + co) * 4);
      const float4 c = *reinterpret_cast<const float4 *>(sRed + ((2 * tl + 1) * 64 + co) * 4);
      const float d = c.x - a.x;
      const long long t = (((long long)n * Ho + oy) * Wo + seg * kStemSeg) / 64 + tl;
      stats[(t * 2 + 0) * 64 + co] = a.x + 0.5f * d;
      stats[(t * 2 + 1) * 64 + co] = a.y + c.y + d * d * 16.f;     // n_a = n_b = 32
      if (minmax) {
        minmax[(t * 2 + 0) * 64 + co] = fminf(a.z, c.z);
        minmax[(t * 2 + 1) * 64 + co] = fmaxf(a.w, c.w);
      }
    }
    // the rows 2 oy - 3 and 2 oy - 2 are done with: their slots take the two new rows
    int sl0 = j0 + 7; sl0 = sl0 >= kStemRing ? sl0 - kStemRing : sl0;
    int sl1 = j0 + 8; sl1 = sl1 >= kStemRing ? sl1 - kStemRing : sl1;
    store_px(sl0, tid, n0); store_px(sl0, tid + 512, n1);
    store_px(sl1, tid, n2); store_px(sl1, tid + 512, n3);
    __syncthreads();
  }
}

// dispatch_nt's hook (conv_geom.h): 0 = launched, < 0 error, 1 = not this shape (the generic kernel runs)
int launch_conv_stem(const float *x, const float *w, float *y, int N, int H, int W, int Cin, int Cout, int Ho, int Wo,
                     const float *x_absmax, const float *w_absmax, float *stats, float *minmax, hipStream_t s) {
  if (dspn::wide_tiles_mode() == 1) return 1;
  if (Cin != 4 || Cout != 64 || Wo % kStemSeg != 0 || Ho != (H + 6 - 7) / 2 + 1 || Wo != (W + 6 - 7) / 2 + 1) return 1;
  const int lds = kStemWBytes + kStemRing * 2 * kStemRowBytes + 8 * 64 * 4 * (int)sizeof(float);
  static dspn::KernelDeviceState st;
  if (const int dev = dspn::ensure_dynamic_lds(reinterpret_cast<const void *>(conv_stem_f16x2_kernel), (size_t)lds, st, "conv_stem"); dev < 0) return dev;
  // strips: enough workgroups for two rounds of the chip, at least 8 output rows each (the first row of a strip loads 7 input rows)
  int rows = 16;
  while (rows > 8 && (long long)N * ((Ho + rows - 1) / rows) * (Wo / kStemSeg) < 512) rows -= 4;
  const long long grid = (long long)N * ((Ho + rows - 1) / rows) * (Wo / kStemSeg);
  {
    dspn::ProfScope prof(0, s);
    hipLaunchKernelGGL(conv_stem_f16x2_kernel, dim3((unsigned)grid), dim3(512), lds, s, x, w, y, N, H, W, Ho, Wo, rows, x_absmax,
                       w_absmax, stats, minmax);
  }
  return dspn::check_launch("conv_stem");
}
