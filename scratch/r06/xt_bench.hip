// Round 6: standalone timing + phase-stamp harness of the short-K members of the wide family (conv_wide.h: conv_ntv_kernel /
// conv_ntw_kernel on 128 x 128 four-wave tiles), with and without the tile-spanning loop.  Timing only (random operands).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDSPN_STAMPS -I dspnet_amd/csrc -o scratch/r06/xt_bench scratch/r06/xt_bench.hip
#include "dspn_common.h"
#include "dspn_store.h"
#include "dspn_pieces.h"
#include "conv_geom.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>
static int g_span = 1;
namespace dspn {
char *last_error_buf() { static char b[512]; return b; }
bool prof_enabled() { return false; }
void prof_begin(int, hipStream_t) {}
void prof_end(hipStream_t) {}
int reserved_cus() { return 0; }
int wide_tiles_mode() { return 0; }
int tile_spanning() { return g_span; }
}
namespace {
using namespace dspn::pieces;
using dspn::st_t;
using dspn::kHalf;
using dspn::u32x4_t;
using dspn::conv::f32x16;
using dspn::conv::bf16x8;
using dspn::conv::xcd_remap;
using ConvGeom = dspn::conv::ConvGeomT<float>;
#include "conv_wide.h"
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static ConvGeom geom1x1(int N, int H, int W, int Cin, int Cout) {
  ConvGeom g; memset(&g, 0, sizeof(g));
  g.N = N; g.Hin = H; g.Win = W; g.Cin = Cin; g.Hg = H; g.Wg = W;
  g.ish = 1; g.isw = 1; g.idh = 1; g.idw = 1;
  g.TR = 1; g.TS = 1; g.WTAPS = 1; g.WS = 1; g.wrs = 1; g.wss = 1;
  g.Cout = Cout; g.ldc = Cout; g.obs = (long long)H * W * Cout; g.OW = W; g.osh = 1; g.osw = 1;
  g.dense = 1; g.flags = 16; g.bf16 = 3;
  g.in_bytes = (unsigned)(4ll * N * H * W * Cin); g.w_bytes = (unsigned)(4ll * Cout * Cin);
  return g;
}
template <typename F> static float timeit(F f, int reps = 5) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 2; ++i) f();
  CK(hipDeviceSynchronize()); CK(hipEventRecord(a, 0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b, 0)); CK(hipDeviceSynchronize());
  float ms = 0; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps * 1e3f;
}
static void stamps(int ntiles, float us) {
  static unsigned long long h[1024 * 16];
  CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wide_stamps), sizeof(h)));
  double sum[16] = {0};
  const int wgs = std::min(512, ntiles);
  for (int b = 0; b < wgs; ++b) for (int i = 0; i < 16; ++i) sum[i] += (double)h[b * 16 + i];
  double tot = 0; for (int i = 0; i < 16; ++i) tot += sum[i];
  printf("    cycles per tile (wave 0, mean): head %6.0f  k-loop %6.0f  requests+staging %6.0f  publish %6.0f  rows+stores %6.0f  tables %6.0f | sum %6.0f = %5.2f us per tile and workgroup at %4.0f MHz\n",
         sum[1] / ntiles, sum[2] / ntiles, sum[3] / ntiles, sum[4] / ntiles, sum[5] / ntiles, sum[6] / ntiles, tot / ntiles,
         us * wgs / ntiles, tot / ntiles / (us * wgs / ntiles));
}
int main(int argc, char **argv) {
  struct Case { int N, H, W, Cin, Cout; };
  const Case cases[] = {{32, 128, 128, 64, 256}, {32, 128, 128, 256, 128}, {32, 64, 64, 512, 128}};
  for (const Case &cs : cases) {
    ConvGeom g = geom1x1(cs.N, cs.H, cs.W, cs.Cin, cs.Cout);
    const size_t M = (size_t)cs.N * cs.H * cs.W, xe = M * cs.Cin, we = (size_t)cs.Cout * cs.Cin, ye = M * cs.Cout;
    std::vector<float> hx(xe), hc(cs.Cin, 1.f);
    std::vector<unsigned short> hw(2 * we), hp(2 * xe);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (unsigned short)(((s >> 16) & 0x83ffu) | 0x3000u); };   // +-(0.125 .. 0.25)
    for (auto &v : hw) v = rnd();
    for (auto &v : hp) v = rnd();
    for (auto &v : hx) { s = s * 1664525u + 1013904223u; v = (float)((int)(s >> 9) - (1 << 22)) / (float)(1 << 22); }
    float *x, *xp, *w, *y, *sc, *sh, *st, *mm, *res;
    const int mt = (int)((M + 127) / 128), ntiles = mt * ((cs.Cout + 127) / 128);
    CK(hipMalloc(&x, 4 * xe)); CK(hipMalloc(&xp, 4 * xe)); CK(hipMalloc(&w, 4 * we)); CK(hipMalloc(&y, 4 * ye)); CK(hipMalloc(&res, 4 * ye));
    CK(hipMalloc(&sc, 4 * cs.Cin)); CK(hipMalloc(&sh, 4 * cs.Cin)); CK(hipMalloc(&st, 8ull * mt * cs.Cout)); CK(hipMalloc(&mm, 8ull * mt * cs.Cout));
    CK(hipMemcpy(x, hx.data(), 4 * xe, hipMemcpyHostToDevice)); CK(hipMemcpy(xp, hp.data(), 4 * xe, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, hw.data(), 4 * we, hipMemcpyHostToDevice));
    CK(hipMemcpy(sc, hc.data(), 4 * cs.Cin, hipMemcpyHostToDevice)); CK(hipMemset(sh, 0, 4 * cs.Cin)); CK(hipMemset(res, 0, 4 * ye));
    printf("case M %zu K %d N %d (%d tiles; the bytes at 5.6 TB/s: %.0f us)\n", M, cs.Cin, cs.Cout, ntiles, (xe + ye) * 4 / 5.6e6);
    for (int span = 0; span < 2; ++span) {
      g_span = span;
      for (int variant = 0; variant < 6; ++variant) {
        ConvGeom c = g;
        const char *name = "";
        const float *in = x, *rs = nullptr;
        bool planes = false;
        switch (variant) {
          case 0: name = "float A, plain"; break;
          case 1: name = "float A, statistics"; c.stats = st; c.minmax = mm; break;
          case 2: name = "float A + affine, statistics"; c.stats = st; c.minmax = mm; c.in_scale = sc; c.in_shift = sh; c.flags |= 32; break;
          case 3: name = "float A + affine, residual"; c.in_scale = sc; c.in_shift = sh; c.flags |= 32 | 8; rs = res; break;
          case 4: name = "planes A, plain"; planes = true; in = xp; c.a_planes = 1; break;
          case 5: name = "planes A, statistics"; planes = true; in = xp; c.a_planes = 1; c.stats = st; c.minmax = mm; break;
        }
        const float us = timeit([&] {
          if (planes) launch_ntw<2, 2, 2>(in, w, nullptr, y, c, 0, rs); else launch_ntv<2, 2>(in, w, nullptr, y, c, 0, rs);
        });
        printf("  span %d  %-30s %7.1f us\n", span, name, us);
        stamps(ntiles, us);
      }
    }
    CK(hipFree(x)); CK(hipFree(xp)); CK(hipFree(w)); CK(hipFree(y)); CK(hipFree(res)); CK(hipFree(sc)); CK(hipFree(sh)); CK(hipFree(st)); CK(hipFree(mm));
  }
  return 0;
}
