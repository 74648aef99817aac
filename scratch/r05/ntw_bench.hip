// Round 5: standalone timing harness of the wide tile family (dspnet_amd/csrc/conv_wide.h) -- compiles in seconds, runs
// without Python.  Operands are random finite fp16 bit patterns (timing only; correctness is scratch/r05/ntw_check.py).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDSPN_ABLATE -I dspnet_amd/csrc -o scratch/r05/ntw_bench scratch/r05/ntw_bench.hip
#include "dspn_common.h"
#include "dspn_pieces.h"
#include "conv_geom.h"
#include <algorithm>
#include <cstdlib>
#include <vector>
namespace dspn {
char *last_error_buf() { static char b[512]; return b; }
bool prof_enabled() { return false; }
void prof_begin(int, hipStream_t) {}
void prof_end(hipStream_t) {}
int reserved_cus() { return 0; }
int wide_tiles_mode() { return 0; }
}
namespace {
using namespace dspn::pieces;
using dspn::conv::f32x16;
using dspn::conv::bf16x8;
using dspn::conv::xcd_remap;
using ConvGeom = dspn::conv::ConvGeomT<float>;
#include "conv_wide.h"
#include "../../scratch/r05/conv_ntp_specialised_kernel.h"
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static ConvGeom fwd_geom(int N, int H, int W, int Cin, int Cout, int k, int stride) {
  ConvGeom g; memset(&g, 0, sizeof(g));
  const int pad = k / 2, Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  g.N = N; g.Hin = H; g.Win = W; g.Cin = Cin; g.Hg = Ho; g.Wg = Wo;
  g.ish = stride; g.isw = stride; g.ioh = -pad; g.iow = -pad; g.idh = 1; g.idw = 1;
  g.TR = k; g.TS = k; g.WTAPS = k * k; g.WS = k; g.wr0 = 0; g.wrs = 1; g.ws0 = 0; g.wss = 1;
  g.Cout = Cout; g.ldc = Cout; g.obs = (long long)Ho * Wo * Cout; g.OW = Wo; g.osh = 1; g.osw = 1;
  g.dense = 1; g.flags = 16; g.bf16 = 3; g.a_planes = 1;
  g.in_bytes = (unsigned)(4ll * N * H * W * Cin); g.w_bytes = (unsigned)(4ll * Cout * k * k * Cin);
  return g;
}
template <int WM, int WN, int ST>
static float run(const ConvGeom &g, const float *x, const float *w, float *y, int dbg, int reps = 10) {
  ConvGeom c = g; c.dbg = dbg;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) launch_ntw<WM, WN, ST>(x, w, nullptr, y, c, 0, nullptr);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < reps; ++i) launch_ntw<WM, WN, ST>(x, w, nullptr, y, c, 0, nullptr);
  CK(hipEventRecord(b, 0)); CK(hipDeviceSynchronize());
  float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps * 1e3f;
}
template <int TM, int TN>
static float runp(const ConvGeom &g, const float *x, const float *w, float *y, int dbg, int reps = 10) {
  ConvGeom c = g; c.dbg = dbg;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) launch_ntp<TM, TN>(x, w, nullptr, y, c, 0, nullptr);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < reps; ++i) launch_ntp<TM, TN>(x, w, nullptr, y, c, 0, nullptr);
  CK(hipEventRecord(b, 0)); CK(hipDeviceSynchronize());
  float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps * 1e3f;
}
int main(int argc, char **argv) {
  struct Case { int N, H, W, Cin, Cout, k, s; };
  const Case cases[] = {{32, 32, 32, 256, 256, 3, 1}, {32, 64, 64, 128, 128, 3, 1}, {32, 16, 16, 512, 512, 3, 1}, {32, 32, 32, 1024, 256, 1, 1}};
  // ntw_bench [case index, -1 = all] [dbg value, -1 = the ablation list]
  const int only_case = argc > 1 ? atoi(argv[1]) : -1, only_dbg = argc > 2 ? atoi(argv[2]) : -1;
  std::vector<int> dbgs = {0, 16, 1 | 16, 2 | 16, 4 | 16, 8 | 16, 1 | 4 | 16, 1 | 4 | 8 | 16, 2 | 4 | 16};
  if (only_dbg >= 0) dbgs = {only_dbg};
  int ci = -1;
  for (const Case &cs : cases) {
    ++ci;
    if (only_case >= 0 && ci != only_case) continue;
    const ConvGeom g = fwd_geom(cs.N, cs.H, cs.W, cs.Cin, cs.Cout, cs.k, cs.s);
    const size_t xe = (size_t)cs.N * cs.H * cs.W * cs.Cin, we = (size_t)cs.Cout * cs.k * cs.k * cs.Cin, ye = (size_t)g.N * g.Hg * g.Wg * cs.Cout;
    std::vector<unsigned short> hx(2 * xe), hw(2 * we);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (unsigned short)(((s >> 16) & 0x83ffu) | 0x3000u); };   // +-(0.125 .. 0.25)
    for (auto &v : hx) v = rnd();
    for (auto &v : hw) v = rnd();
    float *x, *w, *y;
    CK(hipMalloc(&x, 4 * xe)); CK(hipMalloc(&w, 4 * we)); CK(hipMalloc(&y, 4 * ye));
    CK(hipMemcpy(x, hx.data(), 4 * xe, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), 4 * we, hipMemcpyHostToDevice));
    const double flop = 2.0 * g.N * g.Hg * g.Wg * cs.Cout * cs.Cin * cs.k * cs.k;
    printf("case N%d %dx%d Cin%d Cout%d k%d s%d   (MFMA floor at 2.5 PF / 3: %.0f us)\n", cs.N, cs.H, cs.W, cs.Cin, cs.Cout, cs.k, cs.s, flop / 833.3e12 * 1e6);
    for (int d : dbgs) {
      printf("  dbg %2d:", d);
      printf("  256x128 %6.1f us", run<4, 2, 3>(g, x, w, y, d));
      if (cs.Cout % 256 == 0) printf("  128x256 %6.1f us", run<2, 4, 3>(g, x, w, y, d));
      printf("  128x128w %6.1f us", run<2, 2, 2>(g, x, w, y, d));
      printf("  || spec 256x128 %6.1f us", runp<4, 2>(g, x, w, y, d));
      if (cs.Cout % 256 == 0) printf("  spec 128x256 %6.1f us", runp<2, 4>(g, x, w, y, d));
      printf("\n");
    }
    CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(y));
  }
  return 0;
}
