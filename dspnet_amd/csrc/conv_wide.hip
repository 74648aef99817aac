// The wide tile family of the two-piece convolution math as a translation unit of its own (conv_wide.h holds the kernel):
// compiled in under a minute, where conv.hip with its four math modes and two storage types takes five.
#include "dspn_common.h"
#include "dspn_pieces.h"
#include "conv_geom.h"
#include <algorithm>
#include <cstdlib>

namespace {
using namespace dspn::pieces;
using dspn::conv::f32x16;
using dspn::conv::bf16x8;
using dspn::conv::xcd_remap;
using ConvGeom = dspn::conv::ConvGeomT<float>;
#include "conv_wide.h"
}  // namespace

namespace dspn {
namespace conv {
// dspn_conv_set_wide_tiles(mode): 0 automatic, 1 never, 2 / 3 / 4 always that shape where it is legal (tests, experiments) -- a
// launch setting: the K order and the accumulation order per output are the same on every tile.
// Automatic choice, measured on MI355X (scratch/r05/ntw_check.py, profiles/r05_ntw_check_*.txt; plain / fused-epilogue time of
// the stage-3 3x3 layer: conv_nt_kernel 125 / 148 us, 256 x 128 119 / 118, 128 x 256 110 / 110, 128 x 128 on four waves 111 / 112):
// the four-wave tile wins or ties on every layer of the headline graph -- two workgroups per CU cover each other's epilogue,
// which the one-workgroup-per-CU tiles expose -- and is the only one that also pays on the short-K 1x1 layers.
int wide_tile_choice(long long M, int Cout, int nk) {
  const int mode = dspn::wide_tiles_mode();
  if (mode == 1) return 0;
  if (mode >= 2) return mode - 1;
  if (nk < 4) return 0;
  return 3;
}
int launch_wide(int shape, const float *in, const float *w_planes, const float *bias, float *out, const ConvGeomT<float> &g,
                hipStream_t s, const float *residual) {
  if (shape == 1) return launch_ntw<4, 2, 3>(in, w_planes, bias, out, g, s, residual);
  if (shape == 2) return launch_ntw<2, 4, 3>(in, w_planes, bias, out, g, s, residual);
  if (shape == 3) return launch_ntw<2, 2, 2>(in, w_planes, bias, out, g, s, residual);
  return dspn::fail(DSPN_ERR_ARG_, "conv: no wide tile shape %d", shape);
}
}  // namespace conv
}  // namespace dspn
