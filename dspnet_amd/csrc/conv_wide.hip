// The wide tile family of the two-piece convolution math as a translation unit of its own (conv_wide.h holds the kernel):
// compiled in under a minute, where conv.hip with its four math modes and two storage types takes five.
// Compiled twice, like conv.hip (dspn_store.h): as is -- float tensors whose operands are fp16 piece planes -- and through
// conv_wide_h.hip with DSPN_HALF -- bfloat16 tensors, whose activations and weight copies are the operands as they stand.
#include "dspn_common.h"
#include "dspn_store.h"
#include "dspn_pieces.h"
#include "conv_geom.h"
#include "../../include/dspn_nn.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {
using namespace dspn::pieces;
using dspn::st_t;
using dspn::kHalf;
using dspn::u32x4_t;
using dspn::conv::f32x16;
using dspn::conv::bf16x8;
using dspn::conv::xcd_remap;
using ConvGeom = dspn::conv::ConvGeomT<st_t>;
#include "conv_wide.h"
#ifndef DSPN_HALF
#include "conv_stem.h"
#endif
}  // namespace

#ifndef DSPN_HALF
// ---- range guard of the two-piece math (round 5): two small bookkeeping kernels (include/dspn_nn.h)
namespace {
struct AbsminRowsDesc { const float *w; unsigned *out; int rows, row_len; long long begin; };
__global__ __launch_bounds__(256) void absmin_rows_batch_kernel(const AbsminRowsDesc *__restrict__ d, int n) {
  __shared__ float sm[4];
  const long long row = blockIdx.x;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (d[mid].begin <= row) lo = mid; else hi = mid - 1;
  }
  const AbsminRowsDesc e = d[lo];
  const float *p = e.w + (row - e.begin) * (long long)e.row_len;
  float m = 0.f;
  for (int i = threadIdx.x; i < e.row_len; i += 256) m = fmaxf(m, fabsf(p[i]));
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    if (m > 0.f && m < __builtin_huge_valf()) atomicMin(e.out, __float_as_uint(m));     // positive floats order like their bits
  }
}
__global__ __launch_bounds__(256) void tile_minmax_kernel(const float4 *__restrict__ x, long long rows, int C4, int tile_rows,
                                                          float4 *__restrict__ minmax) {
  const long long t = blockIdx.x;
  const long long r0 = t * tile_rows, r1 = r0 + tile_rows < rows ? r0 + tile_rows : rows;
  for (int c = blockIdx.y * 256 + threadIdx.x; c < C4; c += gridDim.y * 256) {
    constexpr float kInf = __builtin_huge_valf();
    float4 mn = make_float4(kInf, kInf, kInf, kInf), mx = make_float4(-kInf, -kInf, -kInf, -kInf);
    for (long long r = r0; r < r1; ++r) {
      const float4 v = x[r * C4 + c];
      mn.x = fminf(mn.x, v.x); mn.y = fminf(mn.y, v.y); mn.z = fminf(mn.z, v.z); mn.w = fminf(mn.w, v.w);
      mx.x = fmaxf(mx.x, v.x); mx.y = fmaxf(mx.y, v.y); mx.z = fmaxf(mx.z, v.z); mx.w = fmaxf(mx.w, v.w);
    }
    minmax[(t * 2 + 0) * C4 + c] = mn;
    minmax[(t * 2 + 1) * C4 + c] = mx;
  }
}
}  // namespace

extern "C" {
int dspn_absmin_rows_batch_f32(const void *table, int n, long long total_rows, void *stream) {
  DSPN_REQUIRE(table && n > 0 && total_rows > 0 && total_rows < (1ll << 31), "absmin_rows_batch: bad argument");
  static_assert(sizeof(AbsminRowsDesc) == 32, "table row layout: 2 pointers, 2 ints, 1 int64");
  hipLaunchKernelGGL(absmin_rows_batch_kernel, dim3((unsigned)total_rows), dim3(256), 0, (hipStream_t)stream,
                     static_cast<const AbsminRowsDesc *>(table), n);
  return dspn::check_launch("absmin_rows_batch");
}
int dspn_tile_minmax_f32(const float *x, long long rows, int C, int tile_rows, float *minmax, void *stream) {
  DSPN_REQUIRE(x && minmax && rows > 0 && C > 0 && C % 4 == 0 && tile_rows > 0, "tile_minmax: bad argument (C must be a multiple of 4)");
  const long long tiles = (rows + tile_rows - 1) / tile_rows;
  DSPN_REQUIRE(tiles < (1ll << 31), "tile_minmax: too many tiles");
  hipLaunchKernelGGL(tile_minmax_kernel, dim3((unsigned)tiles, (C / 4 + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4 *>(x), rows, C / 4, tile_rows, reinterpret_cast<float4 *>(minmax));
  return dspn::check_launch("tile_minmax");
}
}  // extern "C"

#endif   // !DSPN_HALF

namespace dspn {
namespace conv {
#ifndef DSPN_HALF
// dspn_conv_set_wide_tiles(mode): 0 automatic, 1 never, 2 / 3 / 4 always that shape where it is legal (tests, experiments) -- a
// launch setting: the K order and the accumulation order per output are the same on every tile.
// Automatic choice, measured on MI355X (scratch/r05/ntw_check.py, profiles/r05_ntw_check_*.txt; plain / fused-epilogue time of
// the stage-3 3x3 layer: conv_nt_kernel 125 / 148 us, 256 x 128 119 / 118, 128 x 256 110 / 110, 128 x 128 on four waves 111 / 112):
// the four-wave tile wins or ties on every layer of the headline graph -- two workgroups per CU cover each other's epilogue,
// which the one-workgroup-per-CU tiles expose -- and is the only one that also pays on the short-K 1x1 layers.
int wide_tile_choice(long long M, int Cout, int nk, int fused_epilogue) {
  const int mode = dspn::wide_tiles_mode();
  if (mode == 1) return 0;
  if (mode >= 2) return Cout <= 64 ? 4 : mode - 1;
  // one or two k-steps (K = 32 / 64): the kernels are all epilogue -- the wide family's costs less with BatchNorm statistics
  // or BatchNorm-backward sums in it (K = 64: 250 -> 229 us forward, 323 -> 259 us data gradient at 128 x 128 x 32 images),
  // the plain one is a tie
  if (Cout <= 64) return nk >= 4 ? 4 : 0;      // 64 output columns (stage 1): 256 x 64 on four waves
  if (nk < 4 && !(fused_epilogue && nk >= 2)) return 0;
  // 128 x 256 where it divides the columns, fills the chip and the k-loop is long enough to matter (K >= 256, N = 512: 155 / 172
  // against 179 / 190 us on four waves, plain / fused); the four-wave tile everywhere else
  if (Cout % 256 == 0 && nk >= 8 && ((M + 127) / 128) * (Cout / 256) >= 256) return 2;
  return 3;
}
int launch_stem(const float *x, const float *w, float *y, int N, int H, int W, int Cin, int Cout, int Ho, int Wo,
                const float *x_absmax, const float *w_absmax, float *stats, float *minmax, hipStream_t s) {
  return launch_conv_stem(x, w, y, N, H, W, Cin, Cout, Ho, Wo, x_absmax, w_absmax, stats, minmax, s);
}
#endif   // !DSPN_HALF
int launch_wide(int shape, const st_t *in, const st_t *w_planes, const float *bias, st_t *out, const ConvGeomT<st_t> &g,
                hipStream_t s, const st_t *residual) {
  if (shape == 1) return launch_ntw<4, 2, 3>(in, w_planes, bias, out, g, s, residual);
  if (shape == 2) return launch_ntw<2, 4, 3>(in, w_planes, bias, out, g, s, residual);
  if (shape == 3) return launch_ntw<2, 2, 2>(in, w_planes, bias, out, g, s, residual);
  if (shape == 4) return launch_ntw<4, 1, 2, 64>(in, w_planes, bias, out, g, s, residual);      // 256 x 64, 64-row BatchNorm tables
#ifndef DSPN_HALF
  // 10 + shape: the A operand is a float tensor (cut, and optionally affine-transformed, in the loader): conv_ntv_kernel
  if (shape == 12) return launch_ntv<2, 4>(in, w_planes, bias, out, g, s, residual);
  if (shape == 13) return launch_ntv<2, 2>(in, w_planes, bias, out, g, s, residual);
  if (shape == 14) return launch_ntv<4, 1, 64>(in, w_planes, bias, out, g, s, residual);
#endif
  return dspn::fail(DSPN_ERR_ARG_, "conv: no wide tile shape %d", shape);
}
}  // namespace conv
}  // namespace dspn
