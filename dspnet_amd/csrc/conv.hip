// NHWC fp32 convolution family for MI355X (gfx950) on v_mfma_f32_32x32x2_f32.
//
// One implicit-GEMM "NT" kernel serves forward, data-gradient and transposed
// convolution: the GEMM-M dimension enumerates a grid of output pixels, the
// K dimension enumerates (tap, channel) of a GATHERED tensor, and a small
// geometry record says how a grid point + tap maps to a gathered pixel, to a
// weight tap and to an output address.  Stride-2 data gradients / 4x4 s2
// transposed convs run as 4 parity classes (no MFMA work on structurally-zero
// taps).  A second "TN" kernel computes weight gradients with split-K over
// pixels into slabs that a reduce kernel sums in a fixed order (bitwise
// reproducible; no float atomics).
//
// Tiling (wave64, 4 waves per workgroup -- 8 for the 128x128 weight gradient and the plain-epilogue 128x128
// forward, where two workgroups then keep 4 waves on every SIMD): block tile
// (WAVES_M*TM*32) x (WAVES_N*TN*32), BK = 32 floats.  Tiles are staged
// global -> registers -> LDS (16 B per lane, coalesced along C of NHWC) and
// double buffered: the loads of k-step t+1 are issued before the MFMAs of step
// t and written to the other LDS buffer after them (one barrier per k-step).
// LDS rows are padded to 36 floats: 16 B aligned for ds_write_b128 and at most
// 2-way conflicts for the ds_read_b64 fragment reads (a half-swap swizzle that
// removes them was measured perf-neutral -- LDS is ~25 % busy -- and dropped).  Each ds_read_b64 feeds
// two MFMAs: within a group of 4 k values lanes 0-31 take k={0,1} and lanes
// 32-63 take k={2,3} for BOTH operands (the k order inside a sum is free).
//
// fp32 MFMA is an exact fmaf chain (no reduced precision); peak 157 TFLOP/s.
#include "dspn_common.h"
#include "bn_final_job.h"
#include "dspn_store.h"
#include "dspn_pieces.h"
#include "conv_geom.h"
#include <cstdlib>
#include "../../include/dspn_nn.h"

// This file is compiled twice (dspn_store.h): float tensors -> the `*_f32` entry points, and through conv_h.hip with
// DSPN_HALF -> bfloat16 activations, bfloat16 weight copies and the `*_bf16` entry points.  In the bf16 build a 16-byte
// chunk holds 8 channels, a k-step 64 of them, tiles go global -> registers -> LDS without conversion (the folded
// BatchNorm-apply loader widens, applies and rounds), the MFMA is always v_mfma_f32_32x32x16_bf16 and the epilogue
// rounds the fp32 accumulators to bf16 on the way out (BatchNorm statistics are taken of the ROUNDED values, i.e. of the
// tensor the consumers read).
using dspn::st_t;
using dspn::kHalf;
using dspn::u32x4_t;

namespace {

using dspn::conv::f32x16;
using dspn::conv::bf16x8;
using dspn::pieces::bf16x4;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

// bf16-MFMA math mode (per call: the `math` argument of the *_bn_f32 entry points): tensors stay fp32 in HBM, the loaders round to bf16 (RNE,
// v_cvt_pk_bf16_f32) on the way into LDS, v_mfma_f32_32x32x16_bf16 accumulates in fp32.
// padded LDS row of the bf16 NT tiles, in bf16: 32 + 8 (80 B) for float tensors rounded on the way in, 64 + 8 (144 B) for
// bf16 tensors; both strides are conflict-free for ds_read_b128
constexpr int kLdsRowH = kHalf ? 72 : 40;
__device__ __forceinline__ bf16x4 to_bf16x4(const float4 v) {
  bf16x4 r = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
  return r;
}

// DSPN_MATH_F32_BF16X3 ("split" mode, float tensors): every float operand x is cut into three bf16 pieces on its way into
// LDS, p0 = bf16(x), p1 = bf16(x - p0), p2 = bf16(x - p0 - p1) (both differences exact in fp32, p0 + p1 + p2 = x to within
// 2^-24 |x|), and a product x * w is formed as the six partial products x_p * w_q with p + q <= 2 -- each exact, bf16 x bf16
// fits the fp32 accumulator's 24 bits -- summed by the same fp32 accumulation as the fp32 MFMA.  What is dropped (x1 w2,
// x2 w1, x2 w2) is below 2^-22 |x w| in the worst case and 2^-25 |x w| rms (tests/test_split_math.py): the order of the
// rounding of one fp32 product.  Six v_mfma_f32_32x32x16_bf16 replace sixteen v_mfma_f32_32x32x2_f32 of twice the cycles
// each.  Operands beyond the largest bf16 (|x| > 3.39e38) become inf on the way in.
constexpr int kSplitValuPerMfma = 8;   // vector instructions of the piece arithmetic scheduled behind each MFMA
constexpr int kLdsRowS = 3 * 32 + 8;   // LDS row of the three-piece image (bf16): 208 B, conflict-free for ds_read_b128
__device__ __forceinline__ void split3(const float4 v, bf16x4 &p0, bf16x4 &p1, bf16x4 &p2) {
  p0 = to_bf16x4(v);
  const float4 r1 = make_float4(v.x - (float)p0[0], v.y - (float)p0[1], v.z - (float)p0[2], v.w - (float)p0[3]);
  p1 = to_bf16x4(r1);
  const float4 r2 = make_float4(r1.x - (float)p1[0], r1.y - (float)p1[1], r1.z - (float)p1[2], r1.w - (float)p1[3]);
  p2 = to_bf16x4(r2);
}

// DSPN_MATH_F32_F16X2 ("two-piece" mode, float tensors): every float operand x, scaled by a power of two s chosen per tensor
// (include/dspn_nn.h), is cut into TWO fp16 pieces, h0 = fp16(s x), h1 = fp16(s x - h0) (round to nearest even; 11 + 11 bits
// and a sign: h0 + h1 = s x to within 2^-24 |s x|, the rounding of the float itself), and a product is the sum of THREE
// exact partial products h0 g0 + h0 g1 + h1 g0 (11 x 11 bits fit the fp32 accumulator input); the dropped h1 g1 is below
// 2^-24 |x w|.  Half the MFMAs of the three-piece bf16 split for the same fp32-level result -- what it costs is RANGE: fp16
// holds 2^-24 .. 65504, so the scale must put the tensor's largest magnitude below 2^15; elements more than 2^17 below
// that maximum keep an ABSOLUTE error of 2^-25 / s (2^-40 of the maximum) instead of a relative one.
// the power of two that maps a largest magnitude m into [2^14, 2^15) (1 for m = 0 / non-finite m).  Exponent kept within
// +-100, i.e. magnitudes from 2^-85 to 2^115 are scaled exactly into place (beyond that the pieces underflow / overflow);
// the epilogues undo the two operand scales one after the other, so no product of scales is ever formed
// (operand_scale, operand_nonfinite, split2h, repair_inf: csrc/dspn_pieces.h, shared with the BatchNorm backward that writes
// gradients as piece planes)
using namespace dspn::pieces;

constexpr int kEPC = 16 / (int)sizeof(st_t);   // elements per 16-byte chunk: 4 floats or 8 bf16
constexpr int kBK = 8 * kEPC;                  // K elements per k-step of the NT kernel (8 chunks per tile row): 32 or 64
constexpr int kBKF = 32;                       // ... of its fp32-MFMA path
constexpr int kPK = 32;                              // pixels per k-step of the weight-gradient kernel
__device__ __forceinline__ float4 ld4(const st_t *p) { return dspn::CA1Ptr(p).vec4()[0]; }
__device__ __forceinline__ void st4(st_t *p, const float4 v) { dspn::A1Ptr(p).vec4()[0] = v; }
// 8 bf16 of one 16-byte chunk <-> 8 floats
__device__ __forceinline__ void widen8(const u32x4_t w, float (&f)[8]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) { f[2 * e] = dspn::bf16_lo(w[e]); f[2 * e + 1] = dspn::bf16_hi(w[e]); }
}
__device__ __forceinline__ u32x4_t narrow8(const float (&f)[8]) {
  u32x4_t r = {dspn::pack_bf16x2(f[0], f[1]), dspn::pack_bf16x2(f[2], f[3]), dspn::pack_bf16x2(f[4], f[5]),
               dspn::pack_bf16x2(f[6], f[7])};
  return r;
}
constexpr int kLdsRow = 36;    // padded LDS row (floats)

using ConvGeom = dspn::conv::ConvGeomT<st_t>;
using dspn::conv::xcd_remap;

// Split modes, round 3 -- PRE (every split-mode kernel whose k-steps never straddle a tap, Cin % 32 == 0): the weight operand
// arrives as piece planes, [row][tap][Cin / 32][piece][32] with three bf16 or two fp16 pieces (dspn_conv2d_weight_planes_*: cut
// once per step for the whole graph), so the B tile goes global -> registers -> LDS as 16-byte chunks with no arithmetic
// (three-piece math: 9.5 -> 3.5 non-MFMA vector instructions per MFMA on the 3x3 layers).
// (A halo-resident A tile for the 3x3 stride-1 layers -- the (8 + 2) x (16 + 2) input patch under an 8 x 16 output tile loaded,
// affine-transformed and cut once per 32 channels for all nine taps -- was built on top of this in the three-piece math and
// removed again: 2.2 instructions per MFMA and a 4 % higher clock, but 14 % more wave cycles (the image load is exposed once
// per nine k-steps and its 30 registers do not fit beside the fused epilogues of the 128-register kernels): 630 against 675
// images/s.  Its one-dimensional form in the two-piece math -- scratch/row_resident_a_tile_experiment.patch -- lost as well;
// DESIGN.md section 4.)

#ifdef DSPN_ABLATE
__device__ unsigned g_phase_stamps[8192 * 8];     // dspn_debug_set bit 16384: 8 words per wave (conv_nt_kernel), read by dspn_debug_read_stamps
#endif
// EPI: 0 plain epilogue, 1 + BatchNorm statistics of the output (g.stats), 2 + BatchNorm-backward sums (g.bn_sums)
// EPIX = EPI + 4: the A operand arrives as fp16 PIECE PLANES (two-piece math, round 4): [pixel][channels / 32][piece][32],
// the same 4 bytes per element and the same byte address for chunk c of a (pixel, 32-channel block) as the float tensor has
// -- only that the 128-byte record now IS the LDS row image (piece 0: 64 bytes, piece 1: 64 bytes), cut with the power of
// two that operand_scale(g.a_absmax) gives: the loader copies 16-byte chunks, no arithmetic, one ds_write_b128 per chunk
// instead of two ds_write_b64.  Written by dspn_bn_backward_from_sums_f32 (dx_planes): the output gradient a BatchNorm
// backward hands to the convolution in front of it.
template <int WAVES_M, int WAVES_N, int TM, int TN, bool UNIFORM_TAP, int MATH, bool INTF, int EPIX>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, (WAVES_M * WAVES_N == 8 && TM * TN <= 2) ? 4 : 2) void conv_nt_kernel(
    const st_t *__restrict__ in, const st_t *__restrict__ wgt, const float *__restrict__ bias,
    st_t *__restrict__ out, const ConvGeom g, const int m_tiles, const int n_tiles,
    const int ksteps_per_split, float *__restrict__ slab, const st_t *__restrict__ residual) {
  constexpr int EPI = EPIX & 3;
  constexpr bool A_PL = (EPIX & 4) != 0;
  static_assert(!A_PL || (MATH == 3 && UNIFORM_TAP && !INTF && !kHalf), "piece-plane A operands: two-piece math, whole 32-channel blocks, no input affine");
  constexpr bool BF16 = MATH != 0;      // the bf16 matrix instruction
  constexpr bool SPLIT = MATH >= 2;     // ... fed with the pieces of every float operand: 2 = three bf16, 3 = two fp16 pieces
  constexpr int NPC = MATH == 3 ? 2 : 3;   // pieces per operand
  static_assert(!kHalf || MATH == 1, "bf16 tensors always run on the bf16 MFMA");
  constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
  constexpr int NTHR = WAVES_M * WAVES_N * 64;          // 4 waves, or 8 (two workgroups then put 4 waves on every SIMD)
  constexpr int RSTEP = NTHR / 8;                       // tile rows covered by one pass of 16-byte loads (8 chunks per row)
  constexpr bool PRE = SPLIT && UNIFORM_TAP;            // the weight operand arrives as NPC piece planes (see above)
  constexpr int PB = NPC * 32;                          // 16-bit elements of one (row, tap, 32-channel block) of the planes
  // 16-B loads per thread per k-step
  constexpr int A_LD = BM / RSTEP;
  // (Round 4 had the weight planes go global -> LDS directly inside this kernel as a build option -- 782 against 785 - 787
  // images/s with the register path below; round 5 made that, for BOTH operands, the wide family of conv_wide.h.)
  constexpr int B_LD = PRE ? NPC * ((4 * BN + NTHR - 1) / NTHR) : BN / RSTEP;
  constexpr bool B_EXACT = (4 * BN) % NTHR == 0;
  extern __shared__ __attribute__((aligned(1024))) float smem[];
  float *sA = smem;                          // [2][BM][kLdsRow]
  float *sB = smem + 2 * BM * kLdsRow;       // [2][BN][kLdsRow]
  constexpr int ROWH = SPLIT ? NPC * 32 + 8 : kLdsRowH;   // LDS row (16-bit elements) of the 16-bit images: 208 / 144 / 80 B
  // the three-piece image is single-buffered (two barriers per k-step: two stages of 208-B rows do not leave two workgroups
  // per CU); the two-piece image (144-B rows) fits twice: one barrier per k-step, as in the unsplit modes
  constexpr int STAGES = MATH == 2 ? 1 : 2;
  __bf16 *hA = reinterpret_cast<__bf16 *>(smem);   // bf16 mode: [STAGES][BM][ROWH], then [STAGES][BN][ROWH] 
  __bf16 *hB = hA + STAGES * BM * ROWH;

#ifdef DSPN_ABLATE
  const int dbg = g.dbg;   // timing-only ablation build (make ABLATE=1): results are WRONG when non-zero
#else
  constexpr int dbg = 0;   // production build: every ablation branch below is compiled out
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntiles = m_tiles * n_tiles;
  const int M = g.N * g.Hg * g.Wg;
  const int CQ = g.Cin / kEPC;
  const int total_q = g.TR * g.TS * CQ;
  const int nk_all = (total_q + 7) >> 3;
  // split-K: this workgroup handles k-steps [k_begin, k_begin + nk)
  const int k_begin = slab ? blockIdx.y * ksteps_per_split : 0;
  const int nk = slab ? max(0, min(nk_all - k_begin, ksteps_per_split)) : nk_all;

  // two-piece mode: operand scales (wave-uniform, read once), and their exact inverse for the epilogue
  const float sc_a = MATH == 3 ? operand_scale(g.a_absmax) : 1.f, sc_b = MATH == 3 ? operand_scale(g.b_absmax) : 1.f;
  const float inv_a = 1.f / sc_a, inv_b = 1.f / sc_b;      // (exact: powers of two)
  const int chunk = tid & 7, row0 = tid >> 3;
  // two-piece math: the A rows of a thread are those of row0 with bits 0 and 2 swapped.  A 16-lane group of the piece stores
  // (ds_write_b64: rows r and r', 8 chunks each, 144-B rows = 36 dwords) then writes rows 4 apart -- banks 4r .. 4r+15 and
  // 4r+16 .. 4r+31 -- instead of neighbours, whose 16-bank windows overlap in 12 (SQ_LDS_BANK_CONFLICT was 15 - 17 % of the
  // LDS-active cycles of these kernels, all of it from these stores; profiles/r04_f16x2_coexec.csv)
  const int arow0 = MATH == 3 ? ((row0 & ~5) | ((row0 & 1) << 2) | ((row0 >> 2) & 1)) : row0;
  // 8-wave kernels only (measured: +3..4 % there; on 4 waves the doubled store count costs more than the reads gain)
  constexpr bool LDS_SHIFT = NTHR == 512;
  const int lds_shift_w = LDS_SHIFT ? ((row0 >> 3) & 1) * 2 : 0;
  const int lds_shift_r = LDS_SHIFT ? ((tid >> 3) & 1) * 2 : 0;   // (lane & 31) >> 3 & 1 == tid >> 3 & 1

  // Buffer descriptors over the gathered tensor and the weights: a tap outside the image (or a row
  // past M / Cout, or a chunk past K) is given an out-of-range offset and the hardware bounds check
  // returns zeros -- no branch, no select, and all loads of a k-step sit in one basic block.
  const __amdgpu_buffer_rsrc_t rsrc_a =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<st_t *>(in), 0, g.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<st_t *>(wgt), 0, g.w_bytes, 0x00020000);
  constexpr unsigned kOOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rsrc_sc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(g.in_scale), 0, INTF ? (unsigned)g.Cin * 4u : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_sh = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(g.in_shift), 0, INTF ? (unsigned)g.Cin * 4u : 0u, 0x00020000);
  float4 tf_sc = make_float4(0.f, 0.f, 0.f, 0.f), tf_sh = tf_sc;   // affine of this thread's 4 channels (INTF)
  float4 tf_sc2 = tf_sc, tf_sh2 = tf_sc;                           // ... channels 4..7 of its chunk (bf16 tensors)
  unsigned tf_mask = 0;                                            // bit i: A row i of the k-step is inside the image

  // ---- loader state of ONE output tile (re-initialised by setup_tile) -----------------------------
  int ld_m0 = 0, ld_n0 = 0;
  // (vector types, not arrays: the state is rewritten inside the tile loop and must stay in registers --
  // as arrays hipcc demoted it to scratch memory, whose loads serialise with the tile loads in vmcnt order)
  typedef int ivec8 __attribute__((ext_vector_type(8)));
  static_assert(A_LD <= 8 && B_LD <= 8, "loader state vectors hold 8 rows");
  ivec8 a_ih0 = {0, 0, 0, 0, 0, 0, 0, 0}, a_iw0 = a_ih0, a_eoff = a_ih0;   // element offset of tap (0,0) (may be negative)
  ivec8 b_eoff = a_ih0;
  // When a k-step (8 chunks = 32 channels) never straddles a tap, the tap is wave-uniform and is
  // tracked incrementally; otherwise (Cin = 4, 20, 36 ...) each lane derives its own tap.
  constexpr bool uniform_tap = UNIFORM_TAP;   // host guarantees (Cin/4) % 8 == 0
  int u_tr = 0, u_ts = 0, u_cq = 0;   // tap / channel-chunk of chunk 0 of the NEXT k-step to load
  int ld_kt = 0;                      // index (within this workgroup's K range) of the NEXT k-step to load
  // K order of the uniform path is CHANNEL-major: for each block of 32 channels all taps are visited
  // back to back, so the (overlapping) input pixels of neighbouring taps are re-read while they are
  // still in L1/L2 instead of once per sweep over all channels (3328-channel score3_conv: 9x less HBM).
  const int ntaps = g.TR * g.TS;
  // PRE: thread (row = tid >> 2, part = tid & 3) moves chunks 3 * part .. 3 * part + 2 of the 12 chunks (192 B) of tile
  // row `row` (+ NTHR / 4 per pass): one address register per pass, the three chunks at immediate offsets
  constexpr int B_PASS = PRE ? (4 * BN + NTHR - 1) / NTHR : 1;
  static_assert(!PRE || B_LD == NPC * B_PASS, "chunk count of the piece-plane loader");
  const int CB = g.Cin >> 5;                 // PRE: 32-channel blocks per tap
  auto setup_tile = [&](int t) __attribute__((always_inline)) {
    const int tile = xcd_remap(t, ntiles);
    const int mt = tile / n_tiles, nt = tile - mt * n_tiles;   // n fastest: A tile reuse in L2
    ld_m0 = mt * BM; ld_n0 = nt * BN;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      const int m = ld_m0 + arow0 + RSTEP * i;
      const int hw = g.Hg * g.Wg;
      const int n = m / hw, rem = m - n * hw;
      const int oi = rem / g.Wg, oj = rem - oi * g.Wg;
      const int ih0 = oi * g.ish + g.ioh, iw0 = oj * g.isw + g.iow;
      const bool mv = m < M;                       // rows past M: every tap fails the bounds test
      a_ih0[i] = mv ? ih0 : -0x40000000;
      a_iw0[i] = mv ? iw0 : 0;
      a_eoff[i] = mv ? ((n * g.Hin + ih0) * g.Win + iw0) * g.Cin : 0;
    }
    if constexpr (PRE) {
#pragma unroll
      for (int ps = 0; ps < B_PASS; ++ps) {
        const int br = (tid >> 2) + ps * (NTHR / 4), k = ld_n0 + br;
        b_eoff[ps] = ((B_EXACT || br < BN) && k < g.Cout) ? k * (g.WTAPS * CB * PB) + (tid & 3) * (NPC * 8) : -1;
      }
    } else {
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      const int k = ld_n0 + row0 + RSTEP * i;
      b_eoff[i] = k < g.Cout ? k * g.WTAPS * g.Cin : -1;
    }
    }
    ld_kt = 0;
    u_tr = 0; u_ts = 0; u_cq = 0;
    if (uniform_tap && ntaps > 0) {
      const int cb = k_begin / ntaps, tap0 = k_begin - cb * ntaps;
      u_cq = cb * 8; u_tr = tap0 / g.TS; u_ts = tap0 - u_tr * g.TS;
    }
  };

  float4 ra[A_LD], rb[B_LD];
  u32x4_t ha[A_LD], hb[B_LD];    // the same chunks as loaded, bf16 tensors (8 channels each)
  // timing-only ablation (bits 4096 / 8192, split modes): the A side of a k-step -- loads, affine, pieces, LDS stores -- only
  // every 3rd / 9th k-step: the upper bound of what an A tile kept in LDS across a kernel row's / all nine taps could save
  int abl_phase = 0;
  bool abl_a_on = true;
  auto load_tiles = [&]() __attribute__((always_inline)) {     // issues the global loads of k-step ld_kt of the tile set up last
    int tr, ts, cq, cq0 = 0;
    bool qv;
    if (dbg & (4096 | 8192)) {
      const int period = (dbg & 8192) ? 9 : 3;
      abl_a_on = abl_phase == 0;
      abl_phase = abl_phase + 1 == period ? 0 : abl_phase + 1;
    }
    if constexpr (uniform_tap) {
      tr = u_tr; ts = u_ts; cq0 = u_cq; cq = u_cq + chunk; qv = u_cq < CQ && nk > 0 && !(dbg & 128);   // 128: timing-only, no loads
      ++u_ts;                                   // branch-free wave-uniform advance: taps inner, channels outer
      const bool wrap = u_ts == g.TS;
      u_ts = wrap ? 0 : u_ts;
      u_tr += wrap ? 1 : 0;
      const bool wrap2 = u_tr == g.TR;
      u_tr = wrap2 ? 0 : u_tr;
      u_cq += wrap2 ? 8 : 0;
    } else {
      const int q = (k_begin + ld_kt) * 8 + chunk;
      qv = q < total_q && nk > 0;
      const int tap = q / CQ;
      cq = q - tap * CQ; tr = tap / g.TS; ts = tap - tr * g.TS;
    }
    ++ld_kt;
    const int dh = tr * g.idh, dw = ts * g.idw;
    const int a_off = (dh * g.Win + dw) * g.Cin + cq * kEPC;
    auto load_affine = [&]() __attribute__((always_inline)) {
      const unsigned coff = qv ? (unsigned)cq * (4u * kEPC) : kOOB;
      const auto s4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_sc, (int)coff, 0, 0);
      const auto h4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_sh, (int)coff, 0, 0);
      tf_sc = make_float4(__uint_as_float(s4[0]), __uint_as_float(s4[1]), __uint_as_float(s4[2]), __uint_as_float(s4[3]));
      tf_sh = make_float4(__uint_as_float(h4[0]), __uint_as_float(h4[1]), __uint_as_float(h4[2]), __uint_as_float(h4[3]));
      if constexpr (kHalf) {
        const auto s8 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_sc, (int)(coff + (qv ? 16u : 0u)), 0, 0);
        const auto h8 = __builtin_amdgcn_raw_buffer_load_b128(rsrc_sh, (int)(coff + (qv ? 16u : 0u)), 0, 0);
        tf_sc2 = make_float4(__uint_as_float(s8[0]), __uint_as_float(s8[1]), __uint_as_float(s8[2]), __uint_as_float(s8[3]));
        tf_sh2 = make_float4(__uint_as_float(h8[0]), __uint_as_float(h8[1]), __uint_as_float(h8[2]), __uint_as_float(h8[3]));
      }
    };
    if constexpr (INTF) {
      // (Round 4, measured and not kept: these two 16-byte loads per thread and k-step are half as many texture-path requests
      // again as the tiles themselves, and the same 16 bytes for the 64 threads of a chunk index -- fetched by 16 lanes of
      // each wave and handed on by ds_swizzle: 799 against 808 - 810 images/s, the eight swizzles cost more than the 48 idle
      // lanes save; fetched only at a block's first tap: eight registers more around the k-loop, scratch in the fused kernels)
      load_affine();
      tf_mask = 0;
    }
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      const int ih = a_ih0[i] + dh, iw = a_iw0[i] + dw;
      const bool v = qv && abl_a_on && (unsigned)ih < (unsigned)g.Hin && (unsigned)iw < (unsigned)g.Win;
      if constexpr (INTF) tf_mask |= v ? (1u << i) : 0u;
      // valid offsets are < 2^31; setting bit 31 pushes an invalid one past num_records
      const unsigned off = ((unsigned)(a_eoff[i] + a_off) * (unsigned)sizeof(st_t)) | (v ? 0u : kOOB);
      const auto t = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (int)off, 0, 0);
      if constexpr (kHalf) ha[i] = t;
      else ra[i] = make_float4(__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]),
                               __uint_as_float(t[3]));
    }
    const int wtap = (g.wr0 + tr * g.wrs) * g.WS + g.ws0 + ts * g.wss;
    if constexpr (PRE) {
      // piece planes: the 192 bytes of (row, tap, block of 32 channels) are contiguous; 12 chunks per row
      const int boff = (wtap * CB + (cq0 >> 3)) * PB;
#pragma unroll
      for (int ps = 0; ps < B_PASS; ++ps) {
        const bool v = qv && b_eoff[ps] >= 0;
        const unsigned off = ((unsigned)(b_eoff[ps] + boff) * 2u) | (v ? 0u : kOOB);
#pragma unroll
        for (int c = 0; c < NPC; ++c) hb[NPC * ps + c] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, (int)off + 16 * c, 0, 0);
      }
    } else {
    const int b_off = wtap * g.Cin + cq * kEPC;
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      const bool v = qv && b_eoff[i] >= 0;
      const unsigned off = ((unsigned)(b_eoff[i] + b_off) * (unsigned)sizeof(st_t)) | (v ? 0u : kOOB);
      const auto t = __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, (int)off, 0, 0);
      if constexpr (kHalf) hb[i] = t;
      else rb[i] = make_float4(__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]),
                               __uint_as_float(t[3]));
    }
    }
  };
  // float tensors: the input affine of this thread's A rows, in registers
  auto affine_tiles = [&]() __attribute__((always_inline)) {
    if constexpr (INTF) {   // u = x * scale[c] + shift[c] (ReLU), zero where the tap is outside the image
      const bool in_relu = g.flags & 32;
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        // fmaf, like every other place that evaluates this affine (nn.hip's apply / backward kernels, the
        // EPI == 2 mask below): the ReLU mask must come out identical in forward and backward
        float4 u = make_float4(fmaf(ra[i].x, tf_sc.x, tf_sh.x), fmaf(ra[i].y, tf_sc.y, tf_sh.y),
                               fmaf(ra[i].z, tf_sc.z, tf_sh.z), fmaf(ra[i].w, tf_sc.w, tf_sh.w));
        if (in_relu) u = make_float4(fmaxf(u.x, 0.f), fmaxf(u.y, 0.f), fmaxf(u.z, 0.f), fmaxf(u.w, 0.f));
        const bool v = (tf_mask >> i) & 1u;
        ra[i] = make_float4(v ? u.x : 0.f, v ? u.y : 0.f, v ? u.z : 0.f, v ? u.w : 0.f);
      }
    }
  };
  // split mode: the three bf16 pieces of the rows loaded last (after their affine), kept in registers until the LDS image
  // of the current k-step has been read by every wave.  Called in the middle of a k-step's MFMAs, whose issue slots the
  // ~25 vector instructions per 16-byte chunk then share, instead of between the two barriers where nothing overlaps them.
  bf16x4 pa[SPLIT ? A_LD : 1][3], pb[(SPLIT && !PRE) ? B_LD : 1][3];
  auto split_tiles = [&]() __attribute__((always_inline)) {
    if constexpr (SPLIT) {
      if (abl_a_on && !A_PL) {
      affine_tiles();
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        if constexpr (MATH == 3) split2h(ra[i], sc_a, pa[i][0], pa[i][1]);
        else split3(ra[i], pa[i][0], pa[i][1], pa[i][2]);
      }
      }
      if constexpr (!PRE) {
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
          if constexpr (MATH == 3) split2h(rb[i], sc_b, pb[i][0], pb[i][1]);
          else split3(rb[i], pb[i][0], pb[i][1], pb[i][2]);
        }
      }
    }
  };
  // bf16 tensors: the input affine of this thread's A chunks, in registers -- widen, u = x * scale[c] + shift[c] (ReLU), zero
  // outside the image, round back to bf16.  Called in the MIDDLE of a k-step's MFMAs (round 3; it used to sit in
  // store_tiles, after them: ~100 vector instructions per k-step that nothing overlapped beside 8 MFMAs -- the fused 1x1
  // forward took 3.4x the plain one)
  auto half_affine = [&]() __attribute__((always_inline)) {
    if constexpr (kHalf && INTF) {
      const bool in_relu = g.flags & 32;
      const float sc[8] = {tf_sc.x, tf_sc.y, tf_sc.z, tf_sc.w, tf_sc2.x, tf_sc2.y, tf_sc2.z, tf_sc2.w};
      const float sh[8] = {tf_sh.x, tf_sh.y, tf_sh.z, tf_sh.w, tf_sh2.x, tf_sh2.y, tf_sh2.z, tf_sh2.w};
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        float f[8];
        widen8(ha[i], f);
        const bool v = (tf_mask >> i) & 1u;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float u = fmaf(f[e], sc[e], sh[e]);      // the same fmaf as every other evaluation of this affine
          if (in_relu) u = fmaxf(u, 0.f);
          f[e] = v ? u : 0.f;
        }
        ha[i] = narrow8(f);
      }
    }
  };
  auto store_tiles = [&](int buf) __attribute__((always_inline)) {
    if constexpr (kHalf) {
      __bf16 *a = hA + buf * BM * kLdsRowH, *b = hB + buf * BN * kLdsRowH;
#pragma unroll
      for (int i = 0; i < A_LD; ++i)
        *reinterpret_cast<u32x4_t *>(a + (row0 + RSTEP * i) * kLdsRowH + chunk * 8) = ha[i];
#pragma unroll
      for (int i = 0; i < B_LD; ++i)
        *reinterpret_cast<u32x4_t *>(b + (row0 + RSTEP * i) * kLdsRowH + chunk * 8) = hb[i];
      return;
    }
    if constexpr (!SPLIT) affine_tiles();
    if constexpr (SPLIT) {
      // x = p0 + p1 + p2 (split_tiles): piece p of channel k of a row lies at row * ROWH + p * 32 + k
      __bf16 *a = hA + buf * BM * ROWH, *b = hB + buf * BN * ROWH;
      if constexpr (A_PL) {       // chunk c of the 128-byte record = bytes 16 c .. 16 c + 15 of the row image
#pragma unroll
        for (int i = 0; i < A_LD; ++i)
          *reinterpret_cast<float4 *>(a + (arow0 + RSTEP * i) * ROWH + chunk * 8) = ra[i];
      } else if (abl_a_on) {
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        __bf16 *d = a + (arow0 + RSTEP * i) * ROWH + chunk * 4;
#pragma unroll
        for (int pc = 0; pc < NPC; ++pc) *reinterpret_cast<bf16x4 *>(d + 32 * pc) = pa[i][pc];
      }
      }
      if constexpr (PRE) {   // three 16-byte chunks per 32 channels, as loaded
#pragma unroll
        for (int ps = 0; ps < B_PASS; ++ps) {
          const int br = (tid >> 2) + ps * (NTHR / 4);
          if (B_EXACT || br < BN) {
            __bf16 *d = b + br * ROWH + (tid & 3) * (NPC * 8);
#pragma unroll
            for (int c = 0; c < NPC; ++c) *reinterpret_cast<u32x4_t *>(d + 8 * c) = hb[NPC * ps + c];
          }
        }
      } else {
#pragma unroll
      for (int i = 0; i < B_LD; ++i) {
        __bf16 *d = b + (row0 + RSTEP * i) * ROWH + chunk * 4;
#pragma unroll
        for (int pc = 0; pc < NPC; ++pc) *reinterpret_cast<bf16x4 *>(d + 32 * pc) = pb[i][pc];
      }
      }
    } else if constexpr (BF16) {
      __bf16 *a = hA + buf * BM * kLdsRowH, *b = hB + buf * BN * kLdsRowH;
#pragma unroll
      for (int i = 0; i < A_LD; ++i)
        *reinterpret_cast<bf16x4 *>(a + (row0 + RSTEP * i) * kLdsRowH + chunk * 4) = to_bf16x4(ra[i]);
#pragma unroll
      for (int i = 0; i < B_LD; ++i)
        *reinterpret_cast<bf16x4 *>(b + (row0 + RSTEP * i) * kLdsRowH + chunk * 4) = to_bf16x4(rb[i]);
    } else {
      // rows whose bit 3 is set start 2 floats later (inside the 4-float pad): the 16 rows one ds_read_b64 pass
      // fetches then touch all 32 banks instead of 16 (row stride 36 = 4 mod 32 alone is a 2-way conflict on every
      // fragment read: SQ_LDS_BANK_CONFLICT was 42 % of the LDS-active cycles).  The shifted rows are only 8-byte
      // aligned, so the tile is stored as float2 pairs.
      float *a = sA + buf * BM * kLdsRow + lds_shift_w, *b = sB + buf * BN * kLdsRow + lds_shift_w;
      auto put = [&](float *d, const float4 v) __attribute__((always_inline)) {
        if constexpr (LDS_SHIFT) {
          reinterpret_cast<float2 *>(d)[0] = make_float2(v.x, v.y);
          reinterpret_cast<float2 *>(d)[1] = make_float2(v.z, v.w);
        } else {
          *reinterpret_cast<float4 *>(d) = v;
        }
      };
#pragma unroll
      for (int i = 0; i < A_LD; ++i) put(a + (row0 + RSTEP * i) * kLdsRow + chunk * 4, ra[i]);
#pragma unroll
      for (int i = 0; i < B_LD; ++i) put(b + (row0 + RSTEP * i) * kLdsRow + chunk * 4, rb[i]);
    }
  };

  const int wm = (wave / WAVES_N) * TM * 32, wn = (wave % WAVES_N) * TN * 32;
  const int frow = lane & 31, fk = (lane >> 5) * 2;
  const bool has_bias = g.flags & 1, relu = g.flags & 2, accum = g.flags & 4, has_res = g.flags & 8;

  // ---- persistent loop over output tiles ----------------------------------------------------------
  // The workgroup walks tiles t, t + gridDim.x, ...  While the last k-step of a tile is in the matrix
  // pipe the loads of the NEXT tile's first k-step are already in flight (they stay in registers across
  // the epilogue, whose staging uses all of the LDS), so neither the first-load latency nor the store
  // tail of the epilogue is exposed: short-K layers (1x1 expansions, K = 64..256) were serialised
  // load -> MFMA -> store per workgroup.
  const int nk1 = max(nk, 1);   // a K range without taps (parity class of a strided data gradient) runs one all-zero k-step
  int t = blockIdx.x;
  if (t >= ntiles) return;
  // diagnostic build path (dspn_debug_set bit 2048, results WRONG): shader clock held under load =
  // delta s_memtime / delta s_memrealtime x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6)
  const unsigned long long stamp_c0 = (dbg & 2048) ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long stamp_r0 = (dbg & 2048) ? __builtin_amdgcn_s_memrealtime() : 0ull;
  // diagnostic build path (dspn_debug_set bit 16384, results WRONG): where the cycles of a k-step go, per wave -- s_memtime
  // stamps between the phases of every k-step that is not a tile's last, summed per phase:
  //   0 requests issued | 1 first MFMA block | 2 wait for the requested rows | 3 piece arithmetic | 4 second MFMA block |
  //   5 LDS stores | 6 barrier;   slot 7 counts the k-steps.  Written over the head of `out` by lane 0 of every wave.
  unsigned ph_acc[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  unsigned ph_t = (dbg & 32768) ? (unsigned)__builtin_amdgcn_s_memtime() : 0u;
  auto stamp = [&](const int slot) __attribute__((always_inline)) {
    if (dbg & 16384) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned now = (unsigned)__builtin_amdgcn_s_memtime();
      if (slot >= 0) ph_acc[slot] += now - ph_t;
      ph_t = now;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // diagnostic build path (dspn_debug_set bit 32768; timing only): where the cycles of a tile's EPILOGUE go, per wave, in the
  // slots of the k-step stamps: 0 accumulators -> LDS staging | 1 barrier | 2 row loop (LDS read, arithmetic, global store,
  // statistics; INCLUDING the wait for its stores to be acknowledged) | 3 statistics / sums exchange and merge | 4 hand-over
  // to the next tile (first k-step's LDS stores + barrier) | 5 the tile's k-loop;   slot 7 counts the tiles
  auto estamp = [&](const int slot) __attribute__((always_inline)) {
    if (dbg & 32768) {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      const unsigned now = (unsigned)__builtin_amdgcn_s_memtime();
      if (slot >= 0) ph_acc[slot] += now - ph_t;
      ph_t = now;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  f32x16 acc[TM][TN];
  float gmx_all = 0.f;       // EPI == 2, two-piece math: largest |dx| stored by this thread over all its tiles (g.bn_dy_absmax)
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  // one k-step of matrix work on LDS buffer `buf`
  auto mma_step = [&](const int buf) __attribute__((always_inline)) {
    if constexpr (SPLIT) {
      // x * w = sum over piece pairs (p, q), p + q <= 2, of x_p * w_q: six bf16 MFMAs per 16-deep k block, every partial
      // product exact in the fp32 accumulator's input, the three dropped pairs at the level of one fp32 rounding (see kLdsRowS above).  Small terms
      // first.  Lane (row r = lane & 31, half h = lane >> 5) holds k = 8h .. 8h+7 of the block, as in the bf16 mode.
      const __bf16 *a = hA + buf * BM * ROWH + (wm + frow) * ROWH + (lane >> 5) * 8;
      const __bf16 *b = hB + buf * BN * ROWH + (wn + frow) * ROWH + (lane >> 5) * 8;
      auto block = [&](const int kk) __attribute__((always_inline)) {
        bf16x8 fa[NPC][TM], fb[NPC][TN];
#pragma unroll
        for (int p = 0; p < NPC; ++p) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
            fa[p][i] = *reinterpret_cast<const bf16x8 *>(a + i * 32 * ROWH + p * 32 + kk * 16);
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[p][j] = *reinterpret_cast<const bf16x8 *>(b + j * 32 * ROWH + p * 32 + kk * 16);
        }
        // piece pairs, smallest terms first: three-piece bf16 (p + q <= 2), two-piece fp16 (p + q <= 1)
        constexpr int NPROD = MATH == 3 ? 3 : 6;
        constexpr int PA[6] = {MATH == 3 ? 1 : 2, 0, MATH == 3 ? 0 : 1, 1, 0, 0}, PB[6] = {0, MATH == 3 ? 1 : 2, MATH == 3 ? 0 : 1, 0, 1, 0};
#pragma unroll
        for (int t6 = 0; t6 < NPROD; ++t6) {
          if ((dbg & 64) && t6 > 0) break;     // timing-only ablation: one of the products
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              if constexpr (MATH == 3)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[PA[t6]][i]),
                                                                   __builtin_bit_cast(f16x8, fb[PB[t6]][j]), acc[i][j], 0, 0, 0);
              else
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[t6]][i], fb[PB[t6]][j], acc[i][j], 0, 0, 0);
            }
        }
      };
      block(0);
      // (Round 4, measured and not kept: with pure-copy operands, the k-step's four 1-KiB requests per wave issued BETWEEN
      // this block's MFMAs -- sched_group_barrier MFMA / VMEM pairs -- instead of in a burst behind the barrier: 828.3 / 827.9
      // against 830.0 / 827.5 images/s on one box, the data-gradient layers within 1 %: the order of issue is not the limit)
      // the rows requested before this k-step have had the first block's MFMAs to arrive: their pieces are formed on the
      // vector ALU between the second block's MFMAs
      __builtin_amdgcn_sched_barrier(0);
      if (dbg & 16384) {
        stamp(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(2);
      }
      if (!(dbg & 256)) split_tiles();
      if (dbg & 16384) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i)
#pragma unroll
          for (int pc = 0; pc < NPC; ++pc) asm volatile("" : "+v"(pa[i][pc]));
        stamp(3);
      }
      block(1);
      // pin the pieces HERE: their only readers (the LDS stores) sit behind the barrier, and hipcc otherwise sinks the
      // whole piece arithmetic down there, next to them
#pragma unroll
      for (int i = 0; i < A_LD; ++i)
#pragma unroll
        for (int pc = 0; pc < NPC; ++pc) { if constexpr (!A_PL) asm volatile("" : "+v"(pa[i][pc])); }
      if constexpr (!PRE) {
#pragma unroll
      for (int i = 0; i < B_LD; ++i)
#pragma unroll
        for (int pc = 0; pc < NPC; ++pc) asm volatile("" : "+v"(pb[i][pc]));
      }
      __builtin_amdgcn_sched_group_barrier(0x100, NPC * (TM + TN), 0);
#pragma unroll
      for (int m = 0; m < (MATH == 3 ? 3 : 6) * TM * TN; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, PRE ? (kSplitValuPerMfma + 1) / 2 : (MATH == 3 ? 2 * kSplitValuPerMfma : kSplitValuPerMfma), 0);
      }
    } else if constexpr (BF16) {
      // lane (row r = lane & 31, half h = lane >> 5) holds k = 8h .. 8h+7 of each 16-wide MFMA k block
      const __bf16 *a = hA + buf * BM * kLdsRowH + (wm + frow) * kLdsRowH + (lane >> 5) * 8;
      const __bf16 *b = hB + buf * BN * kLdsRowH + (wn + frow) * kLdsRowH + (lane >> 5) * 8;
#pragma unroll
      for (int hk = 0; hk < kBK / 32; ++hk) {     // 32 k values (two 16-deep MFMA blocks) at a time
        if constexpr (kHalf && INTF) {
          if (hk == kBK / 32 - 1) {   // the rows requested before this k-step have had the first half's MFMAs to arrive
            __builtin_amdgcn_sched_barrier(0);
            half_affine();
          }
        }
        bf16x8 fa[2][TM], fb[2][TN];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
            fa[kk][i] = *reinterpret_cast<const bf16x8 *>(a + i * 32 * kLdsRowH + hk * 32 + kk * 16);
#pragma unroll
          for (int j = 0; j < TN; ++j)
            fb[kk][j] = *reinterpret_cast<const bf16x8 *>(b + j * 32 * kLdsRowH + hk * 32 + kk * 16);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk][i], fb[kk][j], acc[i][j], 0, 0, 0);
        if constexpr (kHalf && INTF) {
          if (hk == kBK / 32 - 1) {   // pin the transformed chunks here (their only readers, the LDS stores, sit behind the barrier)
#pragma unroll
            for (int i = 0; i < A_LD; ++i) asm volatile("" : "+v"(ha[i]));
#pragma unroll
            for (int m = 0; m < 2 * TM * TN; ++m) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
            }
          }
        }
      }
    } else {
      const float *a = sA + buf * BM * kLdsRow + (wm + frow) * kLdsRow + fk + lds_shift_r;
      const float *b = sB + buf * BN * kLdsRow + (wn + frow) * kLdsRow + fk + lds_shift_r;
      // fragments of group gq+1 are fetched from LDS before the MFMAs of group gq are issued
      float2 fa[2][TM], fb[2][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const float2 *>(a + i * 32 * kLdsRow);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const float2 *>(b + j * 32 * kLdsRow);
#pragma unroll
      for (int gq = 0; gq < kBKF / 4; ++gq) {
        const int cur = gq & 1, nxt = cur ^ 1;
        if (gq + 1 < kBKF / 4) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
            fa[nxt][i] = *reinterpret_cast<const float2 *>(a + i * 32 * kLdsRow + (gq + 1) * 4);
#pragma unroll
          for (int j = 0; j < TN; ++j)
            fb[nxt][j] = *reinterpret_cast<const float2 *>(b + j * 32 * kLdsRow + (gq + 1) * 4);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].x, fb[cur][j].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].y, fb[cur][j].y, acc[i][j], 0, 0, 0);
          }
        // pin the order: the LDS reads of the NEXT group are issued ahead of this group's MFMAs, so their
        // latency hides under 2*TM*TN MFMAs instead of one (hipcc otherwise sinks them next to their use)
        if (gq + 1 < kBKF / 4) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
        // ... and the next tile's global loads (with their address arithmetic) are spread over the
        // groups instead of delaying the first MFMA of the k-step
        __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, (A_LD + B_LD + 7) / 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * TM * TN, 0);
      }
    }
  };
  setup_tile(t);
  load_tiles();
  split_tiles();
  half_affine();
  store_tiles(0);
  __syncthreads();
  zero_acc();
  int m0 = ld_m0, n0 = ld_n0;
  int tn = t + gridDim.x;
  int kt = 0, buf = 0;
  // ONE loop over (tile, k-step): a single copy of the load + MFMA block; the end-of-tile work hangs off it
  while (true) {
    const bool last = kt == nk1 - 1;
    const bool has_next = tn < ntiles;
    // (the tile transitions are marked unlikely: hipcc then places its spill code there -- 74 scalar registers of the fused
    // data-gradient kernel live in lanes of a vector register -- instead of 24 v_readlane + 6 v_writelane per k-step)
    if (__builtin_expect(last && has_next, 0)) setup_tile(tn);
    stamp(-1);
    load_tiles();   // k-step kt+1 of this tile | k-step 0 of the next | past K without a next tile: out of range, zero-cost
    stamp(0);
    // nk == 0 (a parity class of a strided data gradient without taps): the loads return zeros, so the k-step may run
    // (accumulators stay 0) or be skipped.  The 8-wave build must NOT branch here: with the branch hipcc keeps the
    // loop-carried accumulators in other registers than the MFMA results and copies all 32 after every k-step
    // (32 v_mov + a full MFMA drain per k-step, seen in the ISA of the non-INTF 8-wave variants)
    if (NTHR == 512 || nk > 0) mma_step(buf);
    else { split_tiles(); half_affine(); }   // (the pieces / the affine of the rows just requested are otherwise formed inside mma_step)
    if ((dbg & 16384) && !last) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(acc[i][j]));
      stamp(4);
    }
    if (__builtin_expect(last, 0)) {
      __syncthreads();   // every wave has read its last fragments: the LDS becomes the staging area
      estamp(5);
      // ---- epilogue.  C/D layout: col = lane&31 (cout), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel)
      if (slab) {   // raw partial sums, dense [split][M][Cout]; bias / relu / accumulate happen in the reduce
        float *o = slab + (long long)blockIdx.y * M * g.Cout;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int co = n0 + wn + j * 32 + (lane & 31);
          if (co >= g.Cout) continue;
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
              if (m < M) o[(long long)m * g.Cout + co] = MATH == 3 ? acc[i][j][r] * inv_a * inv_b : acc[i][j][r];
            }
        }
      } else if (dbg & 32) {   // timing-only ablation: no epilogue (the impossible compare keeps the MFMAs alive)
        if (acc[0][0][0] == 1.2345e33f) out[0] = (st_t)acc[TM - 1][TN - 1][5];
      } else {
        // Output tile -> LDS (the barrier above has retired every fragment read) -> rows of float4: a
        // 128-wide row leaves as one 512-B contiguous store per 32 lanes.  The residual / accumulate operand
        // is read the same way, and ALL of a thread's rows are requested before the staging pass, so that
        // their HBM latency overlaps the LDS round trip instead of serialising with the stores (4 loads in
        // flight per thread held the residual convolutions to ~3 TB/s).
        // (Round 4, measured and not kept -- scratch/two_pass_epilogue_experiment_r04.patch: hipcc waits `vmcnt(0)` in front of
        // every chunk's operand rows, i.e. also for the previous chunk's STORES to be acknowledged (one counter for loads and
        // stores, which complete out of order with each other), and in-kernel stamps (dspn_debug_set bit 32768) charge the row
        // loop of the statistics epilogue with 40 - 65 thousand cycles per tile.  But a two-pass form -- operand rows,
        // statistics and the finished values back into the staged tile first, a store-only pass after it -- ran SLOWER: 767
        // against ~830 images/s, the nine-layer table 1.31 / 1.51 / 1.79 against 1.24 / 1.37 / 1.45 ms (plain / +statistics /
        // both).  The acknowledgements are waited for in either form -- by the next tile's first k-step, whose prefetched
        // rows sit behind the stores in the same counter -- and the other workgroup of the CU covers that wait in both.
        // Nor does skipping the staging pay: the plain epilogue stored straight from the accumulators (32 lanes x 4 bytes per
        // row segment) is 5 % slower over nine layers, 10 % on the output-heavy ones, than these 16-byte stores.)
        constexpr int SLD = BN + 4;
        constexpr int C4 = BN / 4, RPP = NTHR / C4, NP = BM / RPP;   // float4 columns per row, rows per pass, passes
        // rows are handled RC at a time: all of them, or half of them when the BatchNorm-backward sums hold a second
        // operand row in registers
        // 8 waves: the kernel has to fit in 128 VGPRs.  Rows in flight per thread and the unrolling of the chunk loop
        // are the settings that compile without scratch (hipcc 7.2): INTF + statistics 1 row / unrolled, the other
        // fused epilogues 2 rows / rolled, the plain epilogue 4 rows / unrolled
        // (round 4, measured: four rows in flight / an unrolled chunk loop for the statistics epilogue without an input affine
        // -- 118 - 120 registers, no scratch -- change nothing: 1.372 vs 1.367 ms over nine layers, 834 images/s either way)
        constexpr int RC8 = (EPI == 1 && INTF) ? 1 : (EPI != 0) ? 2 : 4;
        constexpr bool TIGHT = NTHR == 512 && TM * TN <= 2;     // the 128-register 8-wave kernels
        constexpr int RC = TIGHT ? (NP > RC8 ? RC8 : NP) : ((EPI == 2 && NP > 8) ? NP / 2 : NP);
        float *st = smem;
        const int c4 = tid % C4, er0 = tid / C4;
        const int co = n0 + c4 * 4;
        const bool cvalid = co < g.Cout;
        const bool vec = (g.flags & 16) && co + 3 < g.Cout;
        const st_t *addsrc = has_res ? residual : (accum ? out : nullptr);   // first additive operand
        int offs[RC];   // element offsets (the host checks that the output holds < 2^31 elements)
        float4 rq[RC], xq[EPI == 2 ? RC : 1];
        auto rows_begin = [&](const int ch) __attribute__((always_inline)) {   // addresses + additive operand of chunk ch
#pragma unroll
          for (int p = 0; p < RC; ++p) {
            const int m = m0 + er0 + (ch * RC + p) * RPP;
            if (g.dense) {
              offs[p] = m * g.ldc + co;
            } else {
              const int hw = g.Hg * g.Wg;
              const int n = m / hw, rem = m - n * hw;
              const int oi = rem / g.Wg, oj = rem - oi * g.Wg;
              offs[p] = n * (int)g.obs + ((oi * g.osh + g.ooh) * g.OW + (oj * g.osw + g.oow)) * g.ldc + co;
            }
            if (m >= M || !cvalid) offs[p] = -1;
            rq[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (addsrc && vec && offs[p] >= 0) rq[p] = ld4(addsrc + offs[p]);
          }
        };
        auto rows_bn_x = [&]() __attribute__((always_inline)) {   // the BatchNorm input rows of the current chunk
          if constexpr (EPI == 2) {
#pragma unroll
            for (int p = 0; p < RC; ++p) {
              xq[p] = make_float4(0.f, 0.f, 0.f, 0.f);
              if (vec && offs[p] >= 0) xq[p] = ld4(g.bn_x + offs[p]);
            }
          }
        };
        rows_begin(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              st[(wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * SLD + wn + j * 32 + (lane & 31)] =
                  MATH == 3 ? acc[i][j][r] * inv_a * inv_b : acc[i][j][r];      // (two-piece mode: undo the operand scales, exact)
        rows_bn_x();   // requested once the accumulators are staged (their registers are free), ahead of the barrier
        if (dbg & 32768) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned now = (unsigned)__builtin_amdgcn_s_memtime(); ph_acc[0] += now - ph_t; ph_t = now; }
        __syncthreads();
        estamp(1);
        float bsc[4] = {0.f, 0.f, 0.f, 0.f}, bsh[4] = {0.f, 0.f, 0.f, 0.f}, bmu[4] = {0.f, 0.f, 0.f, 0.f}, brs[4] = {0.f, 0.f, 0.f, 0.f};
        float gs[4] = {0.f, 0.f, 0.f, 0.f}, gss[4] = {0.f, 0.f, 0.f, 0.f};
        float gmx = 0.f;         // largest |dx| this thread stores (EPI == 2, two-piece math: g.bn_dy_absmax)
        if (EPI == 2 && vec) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            bmu[e] = g.bn_mean[co + e]; brs[e] = g.bn_rstd[co + e];
            if (g.bn_relu) { bsc[e] = g.bn_scale[co + e]; bsh[e] = g.bn_shift[co + e]; }
          }
        }
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (has_bias && cvalid) {
#pragma unroll
          for (int e = 0; e < 4; ++e) bv[e] = co + e < g.Cout ? bias[co + e] : 0.f;
        }
        // BatchNorm statistics of the stored values (g.stats): shifted sums about the thread's first row
        float sK[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
        int scnt = 0;
        constexpr bool MINMAX = EPI == 1 && MATH == 3;     // per-tile extremes of the stored values (g.minmax)
        constexpr float kInf = __builtin_huge_valf();
        float vmn[4] = {kInf, kInf, kInf, kInf}, vmx[4] = {-kInf, -kInf, -kInf, -kInf};
#pragma unroll (TIGHT && EPI != 0 && !(EPI == 1 && INTF) ? 1 : NP / RC)
        for (int ch = 0; ch < NP / RC; ++ch) {
          if (ch > 0) { rows_begin(ch); rows_bn_x(); }
#pragma unroll
          for (int p = 0; p < RC; ++p) {
            const int off = offs[p];
            if (off < 0) continue;
            const float4 tv = *reinterpret_cast<const float4 *>(st + (er0 + (ch * RC + p) * RPP) * SLD + c4 * 4);
            float v[4] = {tv.x + bv[0], tv.y + bv[1], tv.z + bv[2], tv.w + bv[3]};
            if (vec) {
              v[0] += rq[p].x; v[1] += rq[p].y; v[2] += rq[p].z; v[3] += rq[p].w;
              if (has_res && accum) {
                const float4 q = ld4(out + off);
                v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
              }
              if (relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
              }
              if constexpr (kHalf) {   // what is stored (and what the statistics / sums below describe) is the bf16 value
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = dspn::round_bf16(v[e]);
              }
              if (!(dbg & 16) || v[0] == 1.2345e33f) st4(out + off, make_float4(v[0], v[1], v[2], v[3]));
              if constexpr (EPI == 2) {
                const float xv[4] = {xq[p].x, xq[p].y, xq[p].z, xq[p].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const float gd = (!g.bn_relu || fmaf(xv[e], bsc[e], bsh[e]) > 0.f) ? v[e] : 0.f;
                  gs[e] += gd;
                  gss[e] += gd * ((xv[e] - bmu[e]) * brs[e]);
                }
                if constexpr (MATH == 3) gmx = fmaxf(gmx, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
              }
              if constexpr (EPI == 1) {
                if (scnt == 0) { sK[0] = v[0]; sK[1] = v[1]; sK[2] = v[2]; sK[3] = v[3]; }
                ++scnt;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[e] - sK[e]; s1[e] += d; s2[e] += d * d; }
                if constexpr (MINMAX) {
#pragma unroll
                  for (int e = 0; e < 4; ++e) { vmn[e] = fminf(vmn[e], v[e]); vmx[e] = fmaxf(vmx[e], v[e]); }
                }
              }
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                if (co + e >= g.Cout) break;
                float x = v[e];
                if (has_res) x += (float)residual[off + e];
                if (accum) x += (float)out[off + e];
                if (relu) x = x > 0.f ? x : 0.f;
                out[off + e] = (st_t)x;
              }
            }
          }
        }
        estamp(2);
        if constexpr (EPI == 1) {
          // per-thread (mean, M2) of its rows -> LDS -> one thread per column merges the RPP row groups with
          // Chan's update in a fixed order -> stats[m tile][mean | M2][column]; the extremes (g.minmax) travel with them in
          // the same exchange (round 4: they had a second one, two more barriers per tile)
          __syncthreads();               // every staged row has been read
          float *red = smem;             // [RPP][BN][2], then the extremes [RPP][BN][2]
          float *red2 = smem + RPP * BN * 2;
          bool mmx = false;
          if constexpr (MINMAX) mmx = g.minmax != nullptr;        // (kernel-uniform)
          if (vec) {
            const float inv = scnt > 0 ? 1.f / (float)scnt : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              red[((er0 * BN) + c4 * 4 + e) * 2] = sK[e] + s1[e] * inv;
              red[((er0 * BN) + c4 * 4 + e) * 2 + 1] = s2[e] - s1[e] * s1[e] * inv;
            }
            if constexpr (MINMAX) {
              if (mmx) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  red2[((er0 * BN) + c4 * 4 + e) * 2] = vmn[e];
                  red2[((er0 * BN) + c4 * 4 + e) * 2 + 1] = vmx[e];
                }
              }
            }
          }
          __syncthreads();
          if (tid < BN && n0 + tid < g.Cout) {
            // merge of the RPP row groups about the first group's mean (no division inside the loop):
            //   mean = m_0 + sum n_e d_e / n,  M2 = sum (M2_e + n_e d_e^2) - n (mean - m_0)^2,  d_e = mean_e - m_0
            const int lim = min(M - m0, BM);
            const float mref = red[tid * 2];            // row group 0 is never empty
            float n = 0.f, sd = 0.f, sq = 0.f;
#pragma unroll 4
            for (int er = 0; er < RPP; ++er) {
              const int ne_i = lim > er ? (lim - er + RPP - 1) / RPP : 0;   // rows er, er + RPP, ... of this tile below M
              const float ne = (float)ne_i;
              const float d = ne_i > 0 ? red[(er * BN + tid) * 2] - mref : 0.f;
              const float m2e = ne_i > 0 ? red[(er * BN + tid) * 2 + 1] : 0.f;
              n += ne; sd += ne * d; sq += m2e + ne * d * d;
            }
            const float dm = sd / n;
            const float mean = mref + dm;
            const float m2 = fmaxf(sq - n * dm * dm, 0.f);
            const long long mt_ = m0 / BM;
            g.stats[(mt_ * 2 + 0) * g.Cout + n0 + tid] = mean;
            g.stats[(mt_ * 2 + 1) * g.Cout + n0 + tid] = m2;
            if constexpr (MINMAX) {
              if (mmx) {
                float mn = kInf, mx = -kInf;
                for (int er = 0; er < RPP && er < lim; ++er) {      // row groups past the tile's last row hold nothing
                  mn = fminf(mn, red2[(er * BN + tid) * 2]); mx = fmaxf(mx, red2[(er * BN + tid) * 2 + 1]);
                }
                g.minmax[(mt_ * 2 + 0) * g.Cout + n0 + tid] = mn;
                g.minmax[(mt_ * 2 + 1) * g.Cout + n0 + tid] = mx;
              }
            }
          }
        }
        if constexpr (EPI == 2) {
          // per-thread sums -> LDS -> one thread per column adds the RPP row groups in a fixed order
          __syncthreads();               // every staged row has been read
          float *red = smem;             // [RPP][BN][2]
          if (vec) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              red[((er0 * BN) + c4 * 4 + e) * 2] = gs[e];
              red[((er0 * BN) + c4 * 4 + e) * 2 + 1] = gss[e];
            }
          }
          __syncthreads();
          if (tid < BN && n0 + tid < g.Cout) {
            float a = 0.f, b = 0.f;
            for (int er = 0; er < RPP; ++er) { a += red[(er * BN + tid) * 2]; b += red[(er * BN + tid) * 2 + 1]; }
            const long long mt_ = g.bn_tile_base + m0 / BM;
            g.bn_sums[(mt_ * 2 + 0) * g.Cout + n0 + tid] = a;
            g.bn_sums[(mt_ * 2 + 1) * g.Cout + n0 + tid] = b;
          }
          if constexpr (MATH == 3) gmx_all = fmaxf(gmx_all, gmx);     // (published once, when the workgroup has run out of tiles)
        }
      }
      estamp(3);
      if (dbg & 32768) ph_acc[7] += 1u;
      if (!has_next) {
        if constexpr (EPI == 2 && MATH == 3) {
          // g.bn_dy_absmax: the largest |dx| this workgroup stored, over ALL its tiles -- one atomic per workgroup and launch
          // (per tile and wave, the dependent look at the end of every epilogue cost the short-K data gradients 30 - 50 us each)
          if (g.bn_dy_absmax) {       // (kernel-uniform)
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) gmx_all = fmaxf(gmx_all, __shfl_xor(gmx_all, o, 64));
            __syncthreads();            // the epilogue's last LDS reads are done
            if (lane == 0) smem[wave] = gmx_all;
            __syncthreads();
            if (tid == 0) {             // ONE atomic per workgroup: the waves of a launch all end together, every one of
              float m = smem[0];        // them would find the slot still empty (4096 same-address atomics: +27 us per launch)
              for (int q = 1; q < NTHR / 64; ++q) m = fmaxf(m, smem[q]);
              if (m > 0.f) atomicMax(g.bn_dy_absmax + (blockIdx.x & 63u), __float_as_uint(m));
            }
          }
        }
        if ((dbg & (16384 | 32768)) && lane == 0 && blockIdx.y == 0) {
#ifdef DSPN_ABLATE
          unsigned *po = g_phase_stamps + ((blockIdx.x * (NTHR / 64) + wave) & 8191) * 8;
#pragma unroll
          for (int q = 0; q < 8; ++q) po[q] = ph_acc[q];
#endif
        }
        if ((dbg & 2048) && tid == 0 && blockIdx.y == 0) {
          const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
          // 4 words per workgroup: cycles, real-time ticks (100 MHz), start tick and end tick (low 32 bits, raw)
          float *po = reinterpret_cast<float *>(out);
          po[4 * blockIdx.x] = (float)(c1 - stamp_c0);
          po[4 * blockIdx.x + 1] = (float)(r1 - stamp_r0);
          po[4 * blockIdx.x + 2] = __uint_as_float((unsigned)stamp_r0);
          po[4 * blockIdx.x + 3] = __uint_as_float((unsigned)r1);
        }
        break;
      }
      zero_acc();
      t = tn; tn += gridDim.x;
      m0 = ld_m0; n0 = ld_n0;
      kt = -1;
      __syncthreads();   // every staged row has been read before the next tile's first k-step overwrites the LDS
    }
    if constexpr (STAGES == 1) {
      if (!last && !(dbg & 4)) __syncthreads();   // every wave has read the fragments of this k-step (the epilogue's barriers cover `last`)
      if (!(dbg & 2)) store_tiles(0);
      if (!(dbg & 4)) __syncthreads();
    } else {
      if (!(dbg & 2)) store_tiles(buf ^ 1);
      if ((dbg & 16384) && !last) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamp(5); }
      if (!(dbg & 4)) __syncthreads();
      if ((dbg & 16384) && !last) { stamp(6); ph_acc[7] += 1u; }
      if (last) estamp(4);
      buf ^= 1;
    }
    ++kt;
  }
}

// ---------------------------------------------------------------------------
// weight gradient: dW[k][tap][c] = sum_pix dY[pix][k] * X[pix@tap][c]
// ---------------------------------------------------------------------------
struct WgradGeom {
  int N, Hin, Win, Cin;      // X (gathered), Cin % 4 == 0
  int Ho, Wo, Cout;          // dY grid and channels (Cout % 4 == 0 physical row = ldy)
  int ldy;                   // dY pixel stride (floats)
  int sh, sw, ph, pw, dh, dw;  // ih = ho*sh - ph + r*dh
  int R, S;
  int pix_per_split;         // multiple of kPK
  unsigned x_bytes, dy_bytes;
  const float *in_scale, *in_shift;   // optional affine (+ReLU) on x, as in ConvGeom
  int in_relu;
  int bf16;                           // host side only: math mode of this call
  const float *dy_absmax, *x_absmax;  // DSPN_MATH_F32_F16X2: device scalars, largest magnitude of dy / of x after its affine (ConvGeom)
  int dy_planes, x_planes;            // host side only: dy / x are fp16 piece planes (MATHX = 4 / 5 / 6)
  // round 6: a BatchNorm-backward finalize riding in this launch (bn_final_job.h): the first job_rows rows of the grid
  // (blockIdx.y < job_rows: dispatched first) run it, sixteen channels per workgroup, and leave; the splits follow
  int job_rows;
  dspn::BnFinalJob job;
};
// -> (row, rows) of the weight gradient's own grid, or row < 0 for a workgroup that has run its share of the job
#define DSPN_WGRAD_JOB_ROWS(g, smem_)                                                                              \
  if (g.job_rows > 0 && (int)blockIdx.y < g.job_rows) {                                                           \
    const int jb_ = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;                                            \
    if (jb_ < g.job.blocks) dspn::bn_final_job_run(g.job, jb_, smem_);                                             \
    return;                                                                                                        \
  }                                                                                                                \
  const int grid_y = (int)gridDim.y - g.job_rows, block_y = (int)blockIdx.y - g.job_rows;

// bf16 mode of the weight gradient: LDS images stay [pixel][channel] (as loaded), rows padded so that the
// four pixel rows of a transposed block fall on disjoint banks; ds_read_b64_tr_b16 hands every lane the 4
// consecutive pixels (= MFMA k) of its channel, two reads per 8-k fragment.
constexpr int wg_row_bytes(int ch) { return ch == 32 ? 64 : ch * 2 + 64; }

// MATHX = 4: the two-piece math (3) with dy as fp16 PIECE PLANES (conv_nt_kernel, EPIX & 4): chunk q of a pixel's BM output
// channels -- the same byte address as the float chunk -- is 8 channels of ONE piece and goes to that piece's LDS plane as it is
template <int WAVES_M, int WAVES_N, int TM, int TN, int MATHX, bool INTF>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, (WAVES_M * WAVES_N == 8) ? 4 : 2) void conv_wgrad_kernel(
    const st_t *__restrict__ x, const st_t *__restrict__ dy, float *__restrict__ slab,
    const WgradGeom g, const int k_tiles, const int j_tiles) {
  // MATHX = 5: x as piece planes (a materialised BatchNorm output written by dspn_bn_apply_f32, y_planes), 6: both operands
  constexpr int MATH = MATHX >= 4 ? 3 : MATHX;
  constexpr bool A_PL = MATHX == 4 || MATHX == 6, B_PL = MATHX == 5 || MATHX == 6;
  static_assert(!(A_PL || B_PL) || !kHalf, "piece-plane operands: float tensors");
  static_assert(!B_PL || !INTF, "piece-plane x: the affine has been applied by the pass that wrote the planes");
  constexpr bool BF16 = MATH != 0, SPLIT = MATH >= 2;   // as in conv_nt_kernel
  constexpr int NPC = MATH == 3 ? 2 : 3;
  static_assert(!kHalf || MATH == 1, "bf16 tensors always run on the bf16 MFMA");
  constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;   // BM over cout, BN over (tap,c)
  constexpr int NTHR = WAVES_M * WAVES_N * 64;                     // 4 waves, or 8 (4 waves per SIMD with two workgroups per CU)
  constexpr int A_CH = BM / kEPC, B_CH = BN / kEPC;                // 16-B chunks per tile row
  // 16-B chunks per thread and k-step; a small tile of bf16 tensors has fewer chunks than threads (32 channels x 32 pixels
  // = 128 chunks on 256 threads): the surplus threads address past the pixel block, fetch nothing and store nothing
  constexpr int A_LD = (A_CH * kPK + NTHR - 1) / NTHR, B_LD = (B_CH * kPK + NTHR - 1) / NTHR;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *sA = smem;                      // [2][kPK][BM]
  float *sB = smem + 2 * kPK * BM;       // [2][kPK][BN]

  DSPN_WGRAD_JOB_ROWS(g, smem)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // (tile, split) of this workgroup.  Workgroups are dealt to the 8 XCDs round-robin in dispatch order (x fastest): taken as
  // is, the tiles of one split -- which all stream the SAME pixel range of dy and x -- land on all eight L2s and every L2
  // fetches that range for its four or five tiles.  xcd_remap gives each XCD a contiguous run of (split, tile) pairs
  // instead: a pixel range is then read through ONE L2 and shared there by all the tiles of its split
  const int lin = xcd_remap(block_y * gridDim.x + blockIdx.x, gridDim.x * grid_y);
  const int split = lin / (int)gridDim.x, tile = lin - split * (int)gridDim.x;
  const int kt_i = tile / j_tiles, jt_i = tile - kt_i * j_tiles;
  const int k0 = kt_i * BM, j0 = jt_i * BN;
  const int P = g.N * g.Ho * g.Wo;
  const int J = g.R * g.S * g.Cin;
  const int p_begin = split * g.pix_per_split;
  const int p_end = min(P, p_begin + g.pix_per_split);
  const int nk = p_end > p_begin ? (p_end - p_begin + kPK - 1) / kPK : 0;

  // fixed per-thread column chunk of the B (x) tile -> fixed tap and channel
  const int b_chunk = tid % B_CH, b_row0 = tid / B_CH;   // rows step by NTHR / B_CH
  constexpr int B_RSTEP = NTHR / B_CH;
  const int jq = j0 / kEPC + b_chunk;
  const int CQ = g.Cin / kEPC;
  const bool jv = jq * kEPC < J;
  const int tap = jq / CQ, cq = jq - tap * CQ;
  const int tr = tap / g.S, ts = tap - tr * g.S;
  const int tdh = tr * g.dh - g.ph, tdw = ts * g.dw - g.pw;
  const int a_chunk = tid % A_CH, a_row0 = tid / A_CH;
  constexpr int A_RSTEP = NTHR / A_CH;
  const bool kv = k0 + a_chunk * kEPC < g.Cout;

  const __amdgpu_buffer_rsrc_t rsrc_x =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<st_t *>(x), 0, g.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_dy =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<st_t *>(dy), 0, g.dy_bytes, 0x00020000);
  constexpr unsigned kOOB = 0x80000000u;
  // this thread's x chunk is a fixed (tap, 4 channels): its affine is loaded once
  float4 tf_sc = make_float4(0.f, 0.f, 0.f, 0.f), tf_sh = tf_sc, tf_sc2 = tf_sc, tf_sh2 = tf_sc;
  unsigned tf_mask = 0;
  if constexpr (INTF) {
    if (jv) {
      tf_sc = *reinterpret_cast<const float4 *>(g.in_scale + cq * kEPC);
      tf_sh = *reinterpret_cast<const float4 *>(g.in_shift + cq * kEPC);
      if constexpr (kHalf) {
        tf_sc2 = *reinterpret_cast<const float4 *>(g.in_scale + cq * kEPC + 4);
        tf_sh2 = *reinterpret_cast<const float4 *>(g.in_shift + cq * kEPC + 4);
      }
    }
  }
  // (image, row, column) of this thread's x rows, advanced by kPK pixels per k-step with carries instead of two
  // integer divisions per load (the divisions were ~120 of the ~270 VALU instructions beside the 64 MFMAs)
  typedef int ivec8 __attribute__((ext_vector_type(8)));
  static_assert(B_LD <= 8, "x-row state holds 8 rows");
  ivec8 b_n = {0, 0, 0, 0, 0, 0, 0, 0}, b_ho = b_n, b_wo = b_n;
  {
    const int hw = g.Ho * g.Wo;
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      const int p = p_begin + b_row0 + i * B_RSTEP;
      const int n = p / hw, rem = p - n * hw;
      b_n[i] = n; b_ho[i] = rem / g.Wo; b_wo[i] = rem - (rem / g.Wo) * g.Wo;
    }
  }
  const int adv_n = kPK / (g.Ho * g.Wo), adv_rem = kPK - adv_n * (g.Ho * g.Wo);
  const int adv_h = adv_rem / g.Wo, adv_w = adv_rem - adv_h * g.Wo;
  float4 ra[A_LD], rb[B_LD];
  u32x4_t ha[A_LD], hb[B_LD];    // the same chunks as loaded, bf16 tensors (8 channels each)
  auto load_tiles = [&](int kt) {
    const int pb = p_begin + kt * kPK;
    if constexpr (INTF) tf_mask = 0;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      const int p = pb + a_row0 + i * A_RSTEP;
      const bool in_blk = a_row0 + i * A_RSTEP < kPK;
      const unsigned off = ((unsigned)(p * g.ldy + k0 + a_chunk * kEPC) * (unsigned)sizeof(st_t)) |
                           ((kv && in_blk && p < p_end) ? 0u : kOOB);
      const auto t = __builtin_amdgcn_raw_buffer_load_b128(rsrc_dy, (int)off, 0, 0);
      if constexpr (kHalf) ha[i] = t;
      else ra[i] = make_float4(__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3]));
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      const int p = pb + b_row0 + i * B_RSTEP;
      const int n = b_n[i], ho = b_ho[i], wo = b_wo[i];
      {   // advance to the next k-step: + (adv_n images, adv_h rows, adv_w columns), one carry per level
        int w2 = wo + adv_w, h2 = ho + adv_h, n2 = n + adv_n;
        const bool cw = w2 >= g.Wo;
        w2 -= cw ? g.Wo : 0; h2 += cw ? 1 : 0;
        const bool chh = h2 >= g.Ho;
        h2 -= chh ? g.Ho : 0; n2 += chh ? 1 : 0;
        b_wo[i] = w2; b_ho[i] = h2; b_n[i] = n2;
      }
      const int ih = ho * g.sh + tdh, iw = wo * g.sw + tdw;
      const bool v = jv && p < p_end && (b_row0 + i * B_RSTEP < kPK) && (unsigned)ih < (unsigned)g.Hin &&
                     (unsigned)iw < (unsigned)g.Win;
      if constexpr (INTF) tf_mask |= v ? (1u << i) : 0u;
      const unsigned off = ((unsigned)(((n * g.Hin + ih) * g.Win + iw) * g.Cin + cq * kEPC) * (unsigned)sizeof(st_t)) |
                           (v ? 0u : kOOB);
      const auto t = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)off, 0, 0);
      if constexpr (kHalf) hb[i] = t;
      else rb[i] = make_float4(__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3]));
    }
  };
  constexpr int RAB = wg_row_bytes(BM), RBB = wg_row_bytes(BN);   // bf16 image row strides (bytes)
  // split mode: three piece planes per operand, single-buffered: [3][kPK][RAB] then [3][kPK][RBB]
  // LDS stages: the split-mode images are single-buffered (two barriers per k-step).  (Round 4, measured neutral and not kept:
  // two stages for the two-piece image of the 8-wave 128 x 128 tile -- 80 KiB, one barrier per k-step: 8.78 - 8.82 against 8.84 ms
  // of weight gradients per step.)
  constexpr int STAGES = SPLIT ? 1 : 2, PLANES = SPLIT ? NPC : 1;
  const float sc_a = MATH == 3 ? operand_scale(g.dy_absmax) : 1.f, sc_b = MATH == 3 ? operand_scale(g.x_absmax) : 1.f;
  const float inv_a = 1.f / sc_a, inv_b = 1.f / sc_b;
  const bool nf_a = MATH == 3 && operand_nonfinite(g.dy_absmax), nf_b = MATH == 3 && operand_nonfinite(g.x_absmax);
  char *hA = reinterpret_cast<char *>(smem);            // [2][kPK][RAB]
  char *hB = hA + STAGES * PLANES * kPK * RAB;          // [2][kPK][RBB]
  // the input affine (+ReLU, zero outside the image) applied to the x rows in registers
  auto transform_tiles = [&]() __attribute__((always_inline)) {
    if constexpr (INTF && kHalf) {
      const float sc[8] = {tf_sc.x, tf_sc.y, tf_sc.z, tf_sc.w, tf_sc2.x, tf_sc2.y, tf_sc2.z, tf_sc2.w};
      const float sh[8] = {tf_sh.x, tf_sh.y, tf_sh.z, tf_sh.w, tf_sh2.x, tf_sh2.y, tf_sh2.z, tf_sh2.w};
#pragma unroll
      for (int i = 0; i < B_LD; ++i) {
        float f[8];
        widen8(hb[i], f);
        const bool v = (tf_mask >> i) & 1u;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float u = fmaf(f[e], sc[e], sh[e]);
          if (g.in_relu) u = fmaxf(u, 0.f);
          f[e] = v ? u : 0.f;
        }
        hb[i] = narrow8(f);
      }
    } else if constexpr (INTF) {
#pragma unroll
      for (int i = 0; i < B_LD; ++i) {
        float4 u = make_float4(fmaf(rb[i].x, tf_sc.x, tf_sh.x), fmaf(rb[i].y, tf_sc.y, tf_sh.y),
                               fmaf(rb[i].z, tf_sc.z, tf_sh.z), fmaf(rb[i].w, tf_sc.w, tf_sh.w));
        if (g.in_relu) u = make_float4(fmaxf(u.x, 0.f), fmaxf(u.y, 0.f), fmaxf(u.z, 0.f), fmaxf(u.w, 0.f));
        const bool v = (tf_mask >> i) & 1u;
        rb[i] = make_float4(v ? u.x : 0.f, v ? u.y : 0.f, v ? u.z : 0.f, v ? u.w : 0.f);
      }
    }
  };
  auto store_tiles = [&](int buf, bool transformed = false) __attribute__((always_inline)) {
    if (!transformed) transform_tiles();
    if constexpr (kHalf) {       // [pixel][channel] images, as loaded: 16 bytes = this pixel's 8 channels
      char *a = hA + buf * kPK * RAB, *b = hB + buf * kPK * RBB;
#pragma unroll
      for (int i = 0; i < A_LD; ++i)
        if (a_row0 + i * A_RSTEP < kPK)
          *reinterpret_cast<u32x4_t *>(a + (a_row0 + i * A_RSTEP) * RAB + a_chunk * 16) = ha[i];
#pragma unroll
      for (int i = 0; i < B_LD; ++i)
        if (b_row0 + i * B_RSTEP < kPK)
          *reinterpret_cast<u32x4_t *>(b + (b_row0 + i * B_RSTEP) * RBB + b_chunk * 16) = hb[i];
    } else if constexpr (SPLIT) {
      char *sA_ = hA + (STAGES == 1 ? 0 : buf) * (PLANES * kPK * RAB), *sB_ = hB + (STAGES == 1 ? 0 : buf) * (PLANES * kPK * RBB);
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        if constexpr (A_PL) {
          // chunk q = a_chunk of the pixel's BM channels: 32-channel block q >> 3, piece (q >> 2) & 1, channels 8 (q & 3) .. + 7
          char *d = sA_ + ((a_chunk >> 2) & 1) * (kPK * RAB) + (a_row0 + i * A_RSTEP) * RAB + ((a_chunk >> 3) * 32 + (a_chunk & 3) * 8) * 2;
          *reinterpret_cast<float4 *>(d) = ra[i];
          continue;
        }
        bf16x4 p0, p1, p2;
        if constexpr (MATH == 3) { split2h(ra[i], sc_a, p0, p1); if (__builtin_expect(nf_a, 0)) repair_inf(p0, p1); }
        else split3(ra[i], p0, p1, p2);
        char *d = sA_ + (a_row0 + i * A_RSTEP) * RAB + a_chunk * 8;
        *reinterpret_cast<bf16x4 *>(d) = p0;
        *reinterpret_cast<bf16x4 *>(d + kPK * RAB) = p1;
        if constexpr (MATH != 3) *reinterpret_cast<bf16x4 *>(d + 2 * kPK * RAB) = p2;
      }
#pragma unroll
      for (int i = 0; i < B_LD; ++i) {
        if constexpr (B_PL) {       // (as A_PL above: chunk q = b_chunk of the tile row's BN columns)
          char *d = sB_ + ((b_chunk >> 2) & 1) * (kPK * RBB) + (b_row0 + i * B_RSTEP) * RBB + ((b_chunk >> 3) * 32 + (b_chunk & 3) * 8) * 2;
          *reinterpret_cast<float4 *>(d) = rb[i];
          continue;
        }
        bf16x4 p0, p1, p2;
        if constexpr (MATH == 3) { split2h(rb[i], sc_b, p0, p1); if (__builtin_expect(nf_b, 0)) repair_inf(p0, p1); }
        else split3(rb[i], p0, p1, p2);
        char *d = sB_ + (b_row0 + i * B_RSTEP) * RBB + b_chunk * 8;
        *reinterpret_cast<bf16x4 *>(d) = p0;
        *reinterpret_cast<bf16x4 *>(d + kPK * RBB) = p1;
        if constexpr (MATH != 3) *reinterpret_cast<bf16x4 *>(d + 2 * kPK * RBB) = p2;
      }
    } else if constexpr (BF16) {
      char *a = hA + buf * kPK * RAB, *b = hB + buf * kPK * RBB;
#pragma unroll
      for (int i = 0; i < A_LD; ++i)
        *reinterpret_cast<bf16x4 *>(a + (a_row0 + i * A_RSTEP) * RAB + a_chunk * 8) = to_bf16x4(ra[i]);
#pragma unroll
      for (int i = 0; i < B_LD; ++i)
        *reinterpret_cast<bf16x4 *>(b + (b_row0 + i * B_RSTEP) * RBB + b_chunk * 8) = to_bf16x4(rb[i]);
    } else {
      float *a = sA + buf * kPK * BM, *b = sB + buf * kPK * BN;
#pragma unroll
      for (int i = 0; i < A_LD; ++i)
        *reinterpret_cast<float4 *>(a + (a_row0 + i * A_RSTEP) * BM + a_chunk * 4) = ra[i];
#pragma unroll
      for (int i = 0; i < B_LD; ++i)
        *reinterpret_cast<float4 *>(b + (b_row0 + i * B_RSTEP) * BN + b_chunk * 4) = rb[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int wm = (wave / WAVES_N) * TM * 32, wn = (wave % WAVES_N) * TN * 32;
  if (nk > 0) { load_tiles(0); store_tiles(0); }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = STAGES == 1 ? 0 : (kt & 1);
    load_tiles(kt + 1);   // past the last k-step: every offset out of range, zero-cost
    // keep the requests HERE: without the fence hipcc sinks the four buffer loads below the 32 MFMAs (to shorten their
    // live ranges) and waits for them at once -- the whole global-memory latency of every k-step was exposed
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SPLIT) {
      // the bf16 mode's transposed reads, once per piece plane; six MFMAs per accumulator and 16-pixel block (conv_nt_kernel)
      const int gl = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
      const char *a = hA + buf * (PLANES * kPK * RAB) + (8 * (gl >> 1) + q) * RAB + (wm + 16 * (gl & 1) + 4 * pp) * 2;
      const char *b = hB + buf * (PLANES * kPK * RBB) + (8 * (gl >> 1) + q) * RBB + (wn + 16 * (gl & 1) + 4 * pp) * 2;
      auto frag = [](const char *base, int row_bytes) {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(base));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(base + 4 * row_bytes));
        bf16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return r;
      };
#pragma unroll
      for (int kk = 0; kk < kPK / 16; ++kk) {
        bf16x8 fa[NPC][TM], fb[NPC][TN];
#pragma unroll
        for (int p = 0; p < NPC; ++p) {
#pragma unroll
          for (int i = 0; i < TM; ++i) fa[p][i] = frag(a + p * kPK * RAB + kk * 16 * RAB + i * 64, RAB);
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[p][j] = frag(b + p * kPK * RBB + kk * 16 * RBB + j * 64, RBB);
        }
        constexpr int NPROD = MATH == 3 ? 3 : 6;     // piece pairs as in conv_nt_kernel
        constexpr int PA[6] = {MATH == 3 ? 1 : 2, 0, MATH == 3 ? 0 : 1, 1, 0, 0}, PB[6] = {0, MATH == 3 ? 1 : 2, MATH == 3 ? 0 : 1, 0, 1, 0};
#pragma unroll
        for (int t6 = 0; t6 < NPROD; ++t6)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              if constexpr (MATH == 3)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[PA[t6]][i]),
                                                                   __builtin_bit_cast(f16x8, fb[PB[t6]][j]), acc[i][j], 0, 0, 0);
              else
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[t6]][i], fb[PB[t6]][j], acc[i][j], 0, 0, 0);
            }
      }
    } else if constexpr (BF16) {
      // transposed-read addressing: 16-lane group gl = lane >> 4 covers channels 16*(gl&1) .. +15 of a 32-channel
      // block and pixels 8*(gl>>1) .. +7 of a 16-pixel MFMA k block; lane 4q+p of the group supplies the address
      // of pixel row q, channels 4p .. 4p+3, and receives its own channel's 4 pixels.
      const int gl = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
      const char *a = hA + buf * kPK * RAB + (8 * (gl >> 1) + q) * RAB + (wm + 16 * (gl & 1) + 4 * pp) * 2;
      const char *b = hB + buf * kPK * RBB + (8 * (gl >> 1) + q) * RBB + (wn + 16 * (gl & 1) + 4 * pp) * 2;
      auto frag = [](const char *base, int row_bytes) {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(base));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(base + 4 * row_bytes));
        bf16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return r;
      };
#pragma unroll
      for (int kk = 0; kk < kPK / 16; ++kk) {
        bf16x8 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = frag(a + kk * 16 * RAB + i * 64, RAB);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = frag(b + kk * 16 * RBB + j * 64, RBB);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    } else {
    const float *a = sA + buf * kPK * BM + (lane >> 5) * BM + wm + (lane & 31);
    const float *b = sB + buf * kPK * BN + (lane >> 5) * BN + wn + (lane & 31);
    float fa[2][TM], fb[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[0][i] = a[i * 32];
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[0][j] = b[j * 32];
#pragma unroll
    for (int ks = 0; ks < kPK / 2; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks + 1 < kPK / 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[nxt][i] = a[(ks + 1) * 2 * BM + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[nxt][j] = b[(ks + 1) * 2 * BN + j * 32];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
      if (ks + 1 < kPK / 2) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
      if constexpr (INTF) {
        // half way through the k-step the x rows requested above have arrived: their affine runs on the vector ALU
        // while the matrix pipe works through the MFMAs already issued, instead of after the last one
        if (ks == kPK / 4 - 1) {
          __builtin_amdgcn_sched_barrier(0);
          transform_tiles();
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    }
    // unconditional (the last iteration stores the zeros its out-of-range loads returned into the idle buffer): behind
    // `if (kt + 1 < nk)` hipcc sinks the loads of load_tiles into the branch, i.e. below the MFMAs
    if constexpr (STAGES == 1) __syncthreads();   // every wave has read this k-step's fragments
    store_tiles(buf ^ 1, INTF && !BF16);
    __syncthreads();
  }
  // slab[split][k][J]: the tile goes through LDS (free after the mainloop's last barrier) and leaves as
  // float4 rows along J (J % 4 == 0 because Cin % 4 == 0)
  float *o = slab + (long long)split * g.Cout * J;
  constexpr int SLD = BN + 4;
  float *st = smem;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        st[(wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * SLD + wn + j * 32 + (lane & 31)] =
            MATH == 3 ? acc[i][j][r] * inv_a * inv_b : acc[i][j][r];
  __syncthreads();
  constexpr int C4 = BN / 4, RPP = NTHR / C4;
  const int c4 = tid % C4, er0 = tid / C4;
  const int jj = j0 + c4 * 4;
  if (jj >= J) return;
#pragma unroll 4
  for (int p = 0; p < BM / RPP; ++p) {
    const int kl = er0 + p * RPP, k = k0 + kl;
    if (k >= g.Cout) break;
    *reinterpret_cast<float4 *>(o + (long long)k * J + jj) = *reinterpret_cast<const float4 *>(st + kl * SLD + c4 * 4);
  }
}


#ifndef DSPN_HALF
#include "conv_wgrad_wide.h"
#endif

// ---------------------------------------------------------------------------
// g[c] = sum over all input pixels of the data gradient, WITHOUT forming it:
//   sum_p dx[p,c] = sum_{k,tap} w[k,tap,c] * D[tap,k],
//   D[tap,k] = sum of dy[n,ho,wo,k] over the output positions whose tap lands inside the image.
// Used for the first convolution, whose input only feeds the beta of bn_data (symbol/resnet.py:91):
// the full data gradient there would be a GEMM with N = 3 useful columns.
// ---------------------------------------------------------------------------
__global__ void batch_sum_kernel(const dspn::CA4Ptr dy, float4 *__restrict__ p2, int N,
                                 long long per_image4) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < per_image4;
       i += (long long)gridDim.x * blockDim.x) {
    float4 s = dy[i];
    for (int n = 1; n < N; ++n) {
      const float4 v = dy[(long long)n * per_image4 + i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    p2[i] = s;
  }
}
struct SumGradGeom { int H, W, Ho, Wo, ldy, R, S, sh, sw, ph, pw, dh, dw, chunks; };
// partial[tap][chunk][k]: rows of the chunk that are valid for the tap, all valid columns
__global__ __launch_bounds__(256) void tap_sum_kernel(const float *__restrict__ p2, float *__restrict__ partial,
                                                      const SumGradGeom g) {
  __shared__ float sm[256];
  const int tap = blockIdx.x, chunk = blockIdx.y;
  const int r = tap / g.S, q = tap - r * g.S;
  auto lo = [](int num, int den) { return num <= 0 ? 0 : (num + den - 1) / den; };
  const int ho_lo = lo(g.ph - r * g.dh, g.sh), wo_lo = lo(g.pw - q * g.dw, g.sw);
  const int ho_hi = min(g.Ho - 1, (g.H - 1 + g.ph - r * g.dh) / g.sh);
  const int wo_hi = min(g.Wo - 1, (g.W - 1 + g.pw - q * g.dw) / g.sw);
  const int rows_per = (g.Ho + g.chunks - 1) / g.chunks;
  const int h0 = max(ho_lo, chunk * rows_per), h1 = min(ho_hi, (chunk + 1) * rows_per - 1);
  const int KL = min(g.ldy, 256), PL = 256 / KL;    // channel lanes x pixel lanes
  const int kl = threadIdx.x % KL, pl = threadIdx.x / KL;
  for (int kb = 0; kb < g.ldy; kb += KL) {
    const int k = kb + kl;
    float s = 0.f;
    if (pl < PL && k < g.ldy && (g.H - 1 + g.ph - r * g.dh) >= 0 && (g.W - 1 + g.pw - q * g.dw) >= 0)
      for (int ho = h0; ho <= h1; ++ho)
        for (int wo = wo_lo + pl; wo <= wo_hi; wo += PL) s += p2[((long long)ho * g.Wo + wo) * g.ldy + k];
    sm[threadIdx.x] = s;
    __syncthreads();
    if (pl == 0 && k < g.ldy) {
      for (int j = 1; j < PL; ++j) s += sm[j * KL + kl];
      partial[((long long)tap * g.chunks + chunk) * g.ldy + k] = s;
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(1024) void sum_grad_final_kernel(const float *__restrict__ partial,
                                                              const float *__restrict__ w, float *__restrict__ out,
                                                              int taps, int chunks, int ldy, int Cout, int Cin) {
  __shared__ double sm[1024];
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = threadIdx.x; i < taps * Cout; i += 1024) {
    const int tap = i / Cout, k = i - tap * Cout;
    double d = 0;
    for (int c = 0; c < chunks; ++c) d += partial[((long long)tap * chunks + c) * ldy + k];
    for (int c = 0; c < Cin; ++c) acc[c] += d * (double)w[((long long)k * taps + tap) * Cin + c];
  }
  for (int c = 0; c < Cin; ++c) {
    sm[threadIdx.x] = acc[c];
    __syncthreads();
    for (int st = 512; st >= 1; st >>= 1) {
      if ((int)threadIdx.x < st) sm[threadIdx.x] += sm[threadIdx.x + st];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = (float)sm[0];
    __syncthreads();
  }
}

// dw[i] (+)= sum_s slab[s][i], fixed order
__global__ void slab_reduce_kernel(const float4 *__restrict__ slab, float4 *__restrict__ dw,
                                   long long n4, int splits, int accumulate) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4;
       i += (long long)gridDim.x * blockDim.x) {
    float4 s = slab[i];
    int k = 1;
    for (; k + 16 <= splits; k += 16) {   // 16 rows in flight, added in order (see slab_reduce_batch_kernel)
      float4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = slab[(long long)(k + u) * n4 + i];
#pragma unroll
      for (int u = 0; u < 16; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; k < splits; ++k) {
      const float4 v = slab[(long long)k * n4 + i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (accumulate) { const float4 v = dw[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    dw[i] = s;
  }
}

// W[k][t][c] (float master) -> Wt[c][t][k] in the storage type (data-gradient operand); bf16 build: also the
// storage-type copy Wh[k][t][c] of W itself (forward operand) when wh != NULL
__global__ void weight_transpose_kernel(const float *__restrict__ w, st_t *__restrict__ wt, st_t *__restrict__ wh,
                                        int K, int T, int C, int Kp) {
  // wt has row length Kp >= K (Kp % kEPC == 0), zero padded
  const long long total = (long long)C * T * Kp;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % Kp);
    const long long ct = i / Kp;
    const int t = (int)(ct % T), c = (int)(ct / T);
    const long long src = ((long long)k * T + t) * C + c;
    const float v = k < K ? w[src] : 0.f;
    wt[i] = (st_t)v;
    if (wh && k < K) wh[src] = (st_t)v;
  }
}

// the split-K slab sums of many weight gradients in ONE launch (rows sorted by `begin`, found by binary search):
// dw[i] (+)= sum_s slab[s][i], fixed order
struct SlabDesc { const float4 *slab; float4 *dw; long long n4; int splits, accumulate; long long begin; };
__global__ void slab_reduce_batch_kernel(const SlabDesc *__restrict__ d, int n, long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (d[mid].begin <= i) lo = mid; else hi = mid - 1;
    }
    const SlabDesc e = d[lo];
    const long long j = i - e.begin;
    float4 s = e.slab[j];
    // 16 slab rows requested per batch, added in order: a small matrix cut into hundreds of slabs (64 x 64 weights,
    // 745 splits) is one long latency chain per thread otherwise
    int k = 1;
    for (; k + 16 <= e.splits; k += 16) {
      float4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = e.slab[(long long)(k + u) * e.n4 + j];
#pragma unroll
      for (int u = 0; u < 16; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; k < e.splits; ++k) {
      const float4 v = e.slab[(long long)k * e.n4 + j];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (e.accumulate) { const float4 v = e.dw[j]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    e.dw[j] = s;
  }
}

// all weight transposes of a step in ONE launch: table rows = {src, dst, K, T, C, Kp, first TILE of the row's range in the
// concatenated tile space, wh}; one workgroup per (32 k) x (32 c) tile of one tap, its row found by binary search on the
// range starts.  The tile is read once as 128-byte rows of the float master, leaves as it is for the storage-type copy wh,
// and goes through a 4-KB LDS image for the transpose: every access a whole run (the first version -- one thread per output
// element, i.e. a 4-byte read every T * C floats -- took 0.40 ms per step for the 24 M weights of the bf16-tensor graphs)
struct WtDesc { const float *w; st_t *wt; int K, T, C, Kp; long long begin; st_t *wh; };
__global__ __launch_bounds__(256) void weight_transpose_batch_kernel(const WtDesc *__restrict__ d, int n) {
  __shared__ float sm[32][33];
  const long long tile_g = blockIdx.x;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (d[mid].begin <= tile_g) lo = mid; else hi = mid - 1;
  }
  const WtDesc e = d[lo];
  const long long tile = tile_g - e.begin;
  const int cblocks = (e.C + 31) >> 5;
  const int cb = (int)(tile % cblocks);
  const long long kt = tile / cblocks;
  const int t = (int)(kt % e.T), kb = (int)(kt / e.T);
  const int r = threadIdx.x >> 3, q = threadIdx.x & 7;            // row of the tile, group of 4 consecutive columns
  {
    const int k = kb * 32 + r, c = cb * 32 + 4 * q;               // (C % 4 == 0: a group of four is inside or outside)
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < e.K && c < e.C) {
      const long long src = ((long long)k * e.T + t) * e.C + c;
      v = *reinterpret_cast<const float4 *>(e.w + src);
      if (e.wh) st4(e.wh + src, v);
    }
    sm[r][4 * q] = v.x; sm[r][4 * q + 1] = v.y; sm[r][4 * q + 2] = v.z; sm[r][4 * q + 3] = v.w;
  }
  __syncthreads();
  const int cc = cb * 32 + r, kq = kb * 32 + 4 * q;               // (Kp % 4 == 0 likewise)
  if (cc < e.C && kq < e.Kp)
    st4(e.wt + ((long long)cc * e.T + t) * e.Kp + kq, make_float4(sm[4 * q][r], sm[4 * q + 1][r], sm[4 * q + 2][r], sm[4 * q + 3][r]));
}

// Piece planes of a weight operand for the split mode (DSPN_MATH_F32_BF16X3), of a float master W[K][T][C]:
//   planes   = those of W itself      [K][T][C / 32][piece][32]        (forward operand; C % 32 == 0)
//   planes_t = those of its transpose [C][T][cols_t / 32][piece][32]   (data-gradient operand; cols_t >= K zero padded)
// every element cut into its three bf16 pieces by the same split3 the loaders use.  One workgroup per (32 k) x (32 c) tile of
// one tap: the tile is read once as 128-byte rows, the forward planes leave as 64-byte runs per row and piece, the
// transposed ones go through a 6-KB LDS image and leave the same way -- both operands of a layer from ONE read of its
// weights, every access a full line (the first version, one thread per element with 2-byte stores, took 0.51 ms per step;
// the float transpose it replaced 0.27).  Table rows sorted by `begin` (tile index), found by binary search per workgroup.
// npc = 3: bf16 pieces (DSPN_MATH_F32_BF16X3); npc = 2: fp16 pieces of w * 2^e, the scale operand_scale() derives from the
// weight's magnitude block `absmax` -- the same function of the same 64 floats the convolution kernels evaluate, so the planes
// and the kernels' epilogue agree on it by construction
struct WpDesc { const float *w; __bf16 *planes; __bf16 *planes_t; int K, T, C, cols_t; long long begin; const float *absmax; int npc, pad_; };
__device__ __forceinline__ void weight_planes_tile(const WpDesc &e, long long tile, __bf16 (*sm)[32][36]) {
  const int pb = e.npc * 32;
  const float scale = e.npc == 2 ? operand_scale(e.absmax) : 1.f;
  const int cblocks = (e.C + 31) >> 5;
  const int cb = (int)(tile % cblocks);
  const long long kt = tile / cblocks;
  const int t = (int)(kt % e.T), kb = (int)(kt / e.T);
  const int r = threadIdx.x >> 3, q = threadIdx.x & 7;          // row of the tile, group of 4 consecutive columns
  const int k = kb * 32 + r, c = cb * 32 + 4 * q;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (k < e.K && c < e.C) v = *reinterpret_cast<const float4 *>(e.w + ((long long)k * e.T + t) * e.C + c);   // C % 4 == 0
  bf16x4 p[3];
  if (e.npc == 2) { split2h(v, scale, p[0], p[1]); repair_inf(p[0], p[1]); p[2] = p[1]; }
  else split3(v, p[0], p[1], p[2]);
  if (e.planes && k < e.K && c < e.C) {      // (C % 32 == 0 whenever the forward planes exist)
    __bf16 *d = e.planes + (((long long)k * e.T + t) * (e.C >> 5) + cb) * pb + 4 * q;
#pragma unroll
    for (int pc = 0; pc < 3; ++pc)
      if (pc < e.npc) *reinterpret_cast<bf16x4 *>(d + 32 * pc) = p[pc];
  }
  if (!e.planes_t) return;                    // block-uniform
#pragma unroll
  for (int pc = 0; pc < 3; ++pc)
#pragma unroll
    for (int i = 0; i < 4; ++i) sm[pc][4 * q + i][r] = p[pc][i];    // [piece][c][k]
  __syncthreads();
  const int cc = cb * 32 + r;                 // this thread's row of the transposed operand, columns k = 4q .. 4q+3 of the block
  if (cc < e.C && kb * 32 < e.cols_t) {
    __bf16 *d = e.planes_t + (((long long)cc * e.T + t) * (e.cols_t >> 5) + kb) * pb + 4 * q;
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) {
      if (pc >= e.npc) break;
      bf16x4 o = {sm[pc][r][4 * q], sm[pc][r][4 * q + 1], sm[pc][r][4 * q + 2], sm[pc][r][4 * q + 3]};
      *reinterpret_cast<bf16x4 *>(d + 32 * pc) = o;
    }
  }
}
__global__ __launch_bounds__(256) void weight_planes_batch_kernel(const WpDesc *__restrict__ d, int n) {
  __shared__ __bf16 sm[3][32][36];
  const long long tile = blockIdx.x;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (d[mid].begin <= tile) lo = mid; else hi = mid - 1;
  }
  const WpDesc e = d[lo];
  weight_planes_tile(e, tile - e.begin, sm);
}
__global__ __launch_bounds__(256) void weight_planes_kernel(const WpDesc e) {
  __shared__ __bf16 sm[3][32][36];
  weight_planes_tile(e, blockIdx.x, sm);
}
// tiles of one table row: k blocks (of the padded transposed operand, if any) x taps x c blocks
long long weight_planes_tiles(int K, int T, int C, int cols_t, bool with_t) {
  return (long long)((std::max(K, with_t ? cols_t : K) + 31) / 32) * T * ((C + 31) / 32);
}

// Largest magnitude of a float tensor, for the operand scales of DSPN_MATH_F32_F16X2: u = x or (relu)(x * scale[c] + shift[c])
// (the operand a convolution with a folded BatchNorm actually multiplies).  The result is kAbsmaxSlots = 64 PARTIAL maxima:
// workgroup b folds its maximum into out[b & 63] with one integer atomicMax (non-negative floats order like their bit
// patterns; max is order independent: deterministic), and the consuming kernels take the maximum of the 64 with one
// coalesced load and a wave reduction.  (One shared word instead of 64: 16 000 same-address atomics per launch serialise
// in L2 -- 140 us per launch whatever the tensor's size, measured.)  The caller zeroes the 64 words once per step.
// NaNs are skipped (fmaxf), infinities propagate.
__device__ __forceinline__ float absmax4(const float4 v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }
__device__ __forceinline__ void absmax_commit(float m, unsigned *out, int slot) {
  __shared__ float sm[4];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    unsigned *o = out + (slot & (kAbsmaxSlots - 1));
    // (a plain look first: once a few workgroups have published, most maxima are not news and need no atomic at all)
    if (m > 0.f && __float_as_uint(m) > __builtin_nontemporal_load(o)) atomicMax(o, __float_as_uint(m));
  }
}
__global__ __launch_bounds__(256) void absmax_kernel(const float4 *__restrict__ x, long long n4, int C4,
                                                     const float4 *__restrict__ scale, const float4 *__restrict__ shift,
                                                     int relu, unsigned *__restrict__ out) {
  float m = 0.f;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long i0 = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (!scale) {
    for (long long i = i0; i < n4; i += stride) m = fmaxf(m, absmax4(x[i]));
  } else if (stride % C4 == 0) {            // every element of this thread lies in one channel group: coefficients loaded once
    const float4 a = scale[(int)(i0 % C4)], b = shift[(int)(i0 % C4)];
    for (long long i = i0; i < n4; i += stride) {
      const float4 t = x[i];
      float4 v = make_float4(fmaf(t.x, a.x, b.x), fmaf(t.y, a.y, b.y), fmaf(t.z, a.z, b.z), fmaf(t.w, a.w, b.w));   // the loaders' fmaf
      if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
      m = fmaxf(m, absmax4(v));
    }
  } else {
    for (long long i = i0; i < n4; i += stride) {
      const int c4 = (int)(i % C4);
      const float4 t = x[i], a = scale[c4], b = shift[c4];
      float4 v = make_float4(fmaf(t.x, a.x, b.x), fmaf(t.y, a.y, b.y), fmaf(t.z, a.z, b.z), fmaf(t.w, a.w, b.w));
      if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
      m = fmaxf(m, absmax4(v));
    }
  }
  absmax_commit(m, out, blockIdx.x);
}
// many small tensors (every weight of a graph) in one launch: chunk = 1024 float4 per workgroup, rows sorted by `begin` (chunks)
struct AmDesc { const float4 *x; unsigned *out; long long n4; long long begin; };
__global__ __launch_bounds__(256) void absmax_batch_kernel(const AmDesc *__restrict__ d, int n) {
  const long long chunk = blockIdx.x;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (d[mid].begin <= chunk) lo = mid; else hi = mid - 1;
  }
  const AmDesc e = d[lo];
  const long long base = (chunk - e.begin) * 1024;
  float m = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long i = base + k * 256 + threadIdx.x;
    if (i < e.n4) m = fmaxf(m, absmax4(e.x[i]));
  }
  absmax_commit(m, e.out, (int)(chunk - e.begin));
}

// out[m*ldc + co] (+)= relu(sum_s slab[s][m][co] + bias[co])   (dense outputs only)
__global__ void nt_split_reduce_kernel(const float *__restrict__ slab, const float *__restrict__ bias,
                                       st_t *__restrict__ out, long long M, int Cout, int ldc, int splits,
                                       int flags, const st_t *__restrict__ residual) {
  const long long total = M * Cout;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long m = i / Cout;
    const int co = (int)(i - m * Cout);
    float v = slab[i];
    for (int k = 1; k < splits; ++k) v += slab[(long long)k * total + i];
    if (flags & 1) v += bias[co];
    if (flags & 8) v += (float)residual[m * ldc + co];
    st_t *o = out + m * ldc + co;
    if (flags & 4) v += (float)*o;
    if (flags & 2) v = v > 0.f ? v : 0.f;
    *o = (st_t)v;
  }
}

// caller-provided scratch for split-K partial tiles (set per call by the C entry points)
struct SplitWs { float *ptr; size_t bytes; };



template <int WAVES_M, int WAVES_N, int TM, int TN, bool UNIFORM_TAP, int MATH, bool INTF, int EPI>
int launch_nt_impl(const st_t *in, const st_t *w, const float *bias, st_t *out, const ConvGeom &g,
                   hipStream_t s, int splits, int ksteps_per_split, float *slab, const st_t *residual) {
  constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
  const long long M = (long long)g.N * g.Hg * g.Wg;
  if (M <= 0) return 0;
  const int mt = (int)((M + BM - 1) / BM), nt = (g.Cout + BN - 1) / BN;
  // mainloop buffers | staged output tile of the epilogue
  const size_t lds = std::max<size_t>(MATH == 3   ? sizeof(__bf16) * 2 * (BM + BN) * (2 * 32 + 8)
                                      : MATH == 2 ? sizeof(__bf16) * (BM + BN) * kLdsRowS
                                      : MATH == 1 ? sizeof(__bf16) * 2 * (BM + BN) * kLdsRowH
                                                  : sizeof(float) * 2 * (BM + BN) * kLdsRow,
                                      sizeof(float) * BM * (BN + 4));
  auto kern = conv_nt_kernel<WAVES_M, WAVES_N, TM, TN, UNIFORM_TAP, MATH, INTF, EPI>;
  // persistent grid: as many workgroups as the chip holds at once (occupancy x CUs, a multiple of 8 so that
  // a workgroup's tiles t, t + grid, ... stay on its XCD's run of the tile order); each walks its tiles
  static dspn::KernelDeviceState st;
  const bool first = !st.slots[0] && !st.slots[1];
  const int dev = dspn::ensure_persistent_grid(reinterpret_cast<const void *>(kern), WAVES_M * WAVES_N * 64, lds, st, "conv_nt");
  if (dev < 0) return dev;
  const int slots = st.slots[dev], slots_per_cu = st.slots_per_cu[dev], slots_cus = st.cus[dev];
  if (first && getenv("DSPN_DEBUG_PRINT"))
    fprintf(stderr, "[dspn] conv_nt<%d,%d,%d,%d,uni=%d,bf16=%d,intf=%d,epi=%d>: %zu B LDS, occupancy %d/CU x %d CUs -> grid %d\n",
            WAVES_M, WAVES_N, TM, TN, (int)UNIFORM_TAP, MATH, (int)INTF, EPI, lds, slots_per_cu, slots_cus, slots);
  // dspn_conv_set_reserved_cus(k): the persistent grid leaves k CUs' worth of workgroup slots free, so that the kernels of
  // another queue (RCCL's all-reduce of the gradient buckets) find room beside a convolution instead of only between two
  const int reserved = dspn::reserved_cus();
  const int avail = reserved > 0 ? std::max(8, slots_per_cu * std::max(8, slots_cus - reserved) / 8 * 8) : slots;
  const int grid_x = (int)std::min<long long>((long long)mt * nt, avail);
  {
    dspn::ProfScope prof(0, s);
    hipLaunchKernelGGL(kern, dim3(grid_x, splits), dim3(WAVES_M * WAVES_N * 64), lds, s, in, w, bias, out, g, mt, nt,
                       ksteps_per_split, splits > 1 ? slab : nullptr, residual);
    if (splits > 1) {
      const long long total = M * g.Cout;
      const int blocks = (int)std::min<long long>((total + 255) / 256, 2048);
      hipLaunchKernelGGL(nt_split_reduce_kernel, dim3(blocks), dim3(256), 0, s, slab, bias, out, M, g.Cout,
                         g.ldc, splits, g.flags, residual);
    }
  }
  return dspn::check_launch("conv_nt");
}

template <int WAVES_M, int WAVES_N, int TM, int TN>
int launch_nt(const st_t *in, const st_t *w, const float *bias, st_t *out, const ConvGeom &g,
              hipStream_t s, int splits, int ksteps_per_split, float *slab, const st_t *residual) {
  const bool uni = ((g.Cin / kEPC) & 7) == 0;
#define DSPN_NT_(U, B, T, E) launch_nt_impl<WAVES_M, WAVES_N, TM, TN, U, B, T, E>(in, w, bias, out, g, s, splits, ksteps_per_split, slab, residual)
#ifdef DSPN_HALF
#define DSPN_NT_UB_(T, E) (uni ? DSPN_NT_(true, 1, T, E) : DSPN_NT_(false, 1, T, E))
#else
#define DSPN_NT_UB_(T, E) (g.bf16 == 3   ? (uni ? DSPN_NT_(true, 3, T, E) : DSPN_NT_(false, 3, T, E)) \
                           : g.bf16 == 2 ? (uni ? DSPN_NT_(true, 2, T, E) : DSPN_NT_(false, 2, T, E)) \
                           : g.bf16 == 1 ? (uni ? DSPN_NT_(true, 1, T, E) : DSPN_NT_(false, 1, T, E)) \
                                         : (uni ? DSPN_NT_(true, 0, T, E) : DSPN_NT_(false, 0, T, E)))
#endif
#ifndef DSPN_HALF
  if (g.a_planes) {      // (dispatch_nt has checked: two-piece math, uniform taps, no input affine)
    if (g.bn_sums) return DSPN_NT_(true, 3, false, 6);
    return g.stats ? DSPN_NT_(true, 3, false, 5) : DSPN_NT_(true, 3, false, 4);
  }
#endif
  if (g.bn_sums) return DSPN_NT_UB_(false, 2);                       // data gradient feeding a BatchNorm backward
  if (g.in_scale) return g.stats ? DSPN_NT_UB_(true, 1) : DSPN_NT_UB_(true, 0);
  return g.stats ? DSPN_NT_UB_(false, 1) : DSPN_NT_UB_(false, 0);
#undef DSPN_NT_UB_
#undef DSPN_NT_
}

// Tile choice: the largest tile that still yields >= one workgroup per CU; if even the smallest
// leaves most of the chip idle and K is long, split K across workgroups (dense outputs only).
#ifdef DSPN_ABLATE
int g_debug_bits = 0;       // timing experiments only (dspn_debug_set, csrc/dspn_debug.h); absent from the production library
#else
constexpr int g_debug_bits = 0;
#endif
static const int kNtBm[5] = {128, 128, 64, 256, 128}, kNtBn[5] = {128, 64, 64, 32, 256};
// Tile configuration (0: 128x128, 1: 128x64, 2: 64x64, 3: 256x32) for an M x Cout output
int nt_config(long long M, int Cout) {
  auto tiles = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((Cout + bn - 1) / bn); };
  int cfg;
  // measured on MI355X (scratch/cfgtest.py, scratch/x3_knobs.sh): 128x128 wins once it yields >= one workgroup per CU
  // in the split math (667 vs 656 images/s against the round-1 threshold of two per CU, which the fp32-MFMA and bf16-tensor
  // modes prefer by 0.3 .. 0.5 %); otherwise the 64x64 tile (4 workgroups of 36 KiB LDS per CU, 4 waves per SIMD)
  static const long long min_tiles = [] { const char *e = getenv("DSPN_NT_MINTILES"); return e ? atoll(e) : 256ll; }();   // experiments
  if (Cout <= 32) cfg = tiles(256, 32) >= min_tiles ? 3 : 2;
  else cfg = (Cout > 64 && tiles(128, 128) >= min_tiles) ? 0 : 2;
  // a Cout just past a multiple of 128 (171 = 19 classes x 9 taps) wastes up to half of the last
  // 128-wide column tile: 64-wide columns cut the padding to < 64
  if (cfg == 0 && (Cout + 63) / 64 * 64 < (Cout + 127) / 128 * 128) cfg = 1;
  static const int cfg64 = [] { const char *e = getenv("DSPN_NT_CFG64"); return e ? atoi(e) : -1; }();   // experiments
  if (cfg64 >= 0 && Cout > 32 && Cout <= 64 && tiles(kNtBm[cfg64], kNtBn[cfg64]) >= min_tiles) cfg = cfg64;
  if ((g_debug_bits >> 8) & 7) cfg = ((g_debug_bits >> 8) & 7) - 1;   // timing experiments only
  return cfg;
}
int dispatch_nt_impl(const st_t *in, const st_t *w, const float *bias, st_t *out, const ConvGeom &g_in,
                     hipStream_t s, SplitWs ws, const st_t *residual, bool *wide_used);
// one convolution launch; g.bn_dy_absmax without g.bn_sums = the magnitude block of the stored output (conv2d_forward_one):
// written by the wide family's epilogue, by a pass over the output behind any other kernel
int dispatch_nt(const st_t *in, const st_t *w, const float *bias, st_t *out, const ConvGeom &g_in,
                hipStream_t s, SplitWs ws, const st_t *residual = nullptr) {
  bool wide = false;
  const int rc = dispatch_nt_impl(in, w, bias, out, g_in, s, ws, residual, &wide);
#ifndef DSPN_HALF
  if (!rc && g_in.bn_dy_absmax && !g_in.bn_sums && !wide) {
    if (!(g_in.dense && g_in.obs == (long long)g_in.Hg * g_in.Wg * g_in.ldc && g_in.ldc % 4 == 0))
      return dspn::fail(DSPN_ERR_ARG_, "conv2d_forward: the output magnitude block needs a dense output with ldc %% 4 == 0");
    return dspn_absmax_f32(out, (long long)g_in.N * g_in.Hg * g_in.Wg, g_in.ldc, nullptr, nullptr, 0,
                           reinterpret_cast<float *>(g_in.bn_dy_absmax), s);
  }
#endif
  return rc;
}
int dispatch_nt_impl(const st_t *in, const st_t *w, const float *bias, st_t *out, const ConvGeom &g_in,
                     hipStream_t s, SplitWs ws, const st_t *residual, bool *wide_used) {
  ConvGeom g = g_in;
  g.dbg = g_debug_bits;
  // the 4-element-vector epilogue needs aligned rows (16 bytes float, 8 bytes bf16) in every operand it touches
  constexpr uintptr_t amask = 4 * sizeof(st_t) - 1;
  if (g.ldc % 4 == 0 && g.obs % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & amask) == 0 &&
      (!residual || (reinterpret_cast<uintptr_t>(residual) & amask) == 0))
    g.flags |= 16;
  {
    const long long ib = (long long)sizeof(st_t) * g.N * g.Hin * g.Win * g.Cin,
                    wb = (long long)sizeof(st_t) * g.Cout * g.WTAPS * g.Cin;
    if (ib >= (1ll << 31) || wb >= (1ll << 31))
      return dspn::fail(DSPN_ERR_ARG_, "conv: tensors of 2 GiB or more are not supported by the buffer-addressed kernel");
    g.in_bytes = (unsigned)ib; g.w_bytes = (unsigned)wb;
  }
  const long long M = (long long)g.N * g.Hg * g.Wg;
  if (M <= 0) return 0;
  // split mode with whole 32-channel blocks per tap: the kernels read the weights as piece planes
  const bool pre = !kHalf && g.bf16 >= 2 && ((g.Cin / kEPC) & 7) == 0;
  if (pre) {
    if (!g.w_planes)
      return dspn::fail(DSPN_ERR_ARG_, "conv: the split math modes with a multiple of 32 input channels (%d) need the weight operand as piece planes (dspn_conv2d_weight_planes_f32)", g.Cin);
    w = static_cast<const st_t *>(g.w_planes);
    g.w_bytes = (unsigned)((g.bf16 == 3 ? 4ll : 6ll) * g.Cout * g.WTAPS * g.Cin);
  }
  auto tiles = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((g.Cout + bn - 1) / bn); };
  const int cfg = nt_config(M, g.Cout);
  const int *bm_ = kNtBm, *bn_ = kNtBn;
  const long long nblk = tiles(bm_[cfg], bn_[cfg]);
  const int nk = (g.TR * g.TS * (g.Cin / kEPC) + 7) >> 3;
  int splits = 1, per = nk;
  if (g.a_planes && !(pre && g.bf16 == 3 && !g.in_scale))
    return dspn::fail(DSPN_ERR_ARG_, "conv: a piece-plane operand needs DSPN_MATH_F32_F16X2, a multiple of 32 channels (%d) and no input affine", g.Cin);
  if (g.stats && (!g.dense || !(g.flags & 16) || g.Cout % 4 != 0))
    return dspn::fail(DSPN_ERR_ARG_, "conv2d_forward: output statistics need a dense, 16-byte aligned output with Cout %% 4 == 0");
  if (g.bn_sums && (g.in_scale || g.stats))
    return dspn::fail(DSPN_ERR_ARG_, "conv: BatchNorm-backward sums cannot be combined with an input affine or output statistics");
  if (g.bn_sums && (!(g.flags & 16) || g.Cout % 4 != 0))
    return dspn::fail(DSPN_ERR_ARG_, "conv2d_dgrad: BatchNorm sums need a 16-byte aligned dx with Cin %% 4 == 0");
  if (g.dense && nblk < 192 && nk >= 16 && ws.ptr && !g.stats && !g.bn_sums) {
    splits = (int)std::min<long long>((384 + nblk - 1) / nblk, nk / 8);
    splits = std::max(1, std::min(splits, 32));
    while (splits > 1 && sizeof(float) * (size_t)splits * M * g.Cout > ws.bytes) --splits;
    per = (nk + splits - 1) / splits;
    splits = (nk + per - 1) / per;
  }
  // The 128x128 tile on 8 waves (4 x 2 of 32 x 64; two workgroups put 4 waves on every SIMD instead of 2): one wave
  // per SIMD reaches 67 % of the MFMA rate in this loop, two 81 %, four 93 % on a plain 1x1 layer (in-kernel stamps,
  // scratch/clock_probe.py) -- but the 128-VGPR budget leaves no room for the fused-BatchNorm epilogues, which lose
  // more than the main loop gains.  Used for the plain epilogue only (measured +4..6 %); DSPN_NT_8WAVE=1|0 forces it.
  static const char *eight_env = getenv("DSPN_NT_8WAVE");
  const int eight_mode = eight_env ? atoi(eight_env) : -1;   // -1 default, 0 never, 1 always, 2 default + 1x1 dgrads, 3 default + every 1x1
  const bool one_tap = g.TR * g.TS == 1;
  // bf16 tensors: the main loop is ~3x shorter, the fused epilogues cost relatively more, and the 8-wave form wins for
  // every epilogue (measured on the resnet-50 step: 28.6 -> 26.1 ms)
  // the split mode's kernels all fit the 128-register budget of the 8-wave form without scratch, fused epilogues included
  const bool eight = eight_mode == 0 ? false : eight_mode == 1 ? true : (kHalf || g.bf16 >= 2) ? true
                     : ((!g.stats && !g.bn_sums) || (eight_mode == 2 && one_tap && g.bn_sums) || (eight_mode == 3 && one_tap));
  // The wide family (conv_wide.h): both operands pure copies, 64 x 64 outputs per wave.  Legal when the vector epilogue applies,
  // there is at least one tap, no split-K, and the BatchNorm tables are per 128 rows (cfg 0), or per 64 rows with at most 64 output
  // columns (cfg 2) -- the layouts the wide epilogue writes -- and the A operand is a pure copy: fp16 piece planes in the float
  // build (=> two-piece math, whole 32-channel blocks, no input affine), the bf16 tensor itself in the bf16 build (whole
  // 64-channel blocks, no input affine).
  {
    bool wide_ok = splits == 1 && (cfg == 0 || (cfg == 2 && g.Cout > 32 && g.Cout <= 64)) && (g.flags & 16) && g.Cout % 4 == 0 &&
                   g.TR * g.TS > 0;
    // bf16 tensors: legal, bit-identical (tests/test_wide_tiles_gpu.py) and NOT faster -- the training step measured 1340 - 1343
    // images/s with the family against 1345 - 1347 without, every forced shape lower still (profiles/r05_bf16_wide_ab.txt): those
    // kernels are bound by the bytes they ask the L2 for, which a 128 x 128 tile does not change.  Only a forced mode routes them.
    if (kHalf) wide_ok = wide_ok && !g.in_scale && g.Cin % 64 == 0 && dspn::wide_tiles_mode() >= 2;
    else wide_ok = wide_ok && pre && g.bf16 == 3;       // (pre: whole 32-channel blocks, the weights as piece planes)
    if (wide_ok) {
      int shape = dspn::conv::wide_tile_choice(M, g.Cout, nk, (g.stats || g.bn_sums) ? 1 : 0);
      // a float A operand (with or without the folded BatchNorm affine) goes through the family's register-staged member:
      // shapes 12 / 13 / 14 = 128 x 256, 128 x 128 on four waves, 256 x 64 (conv_wide.h, conv_ntv_kernel)
      if (shape && !kHalf && !g.a_planes) shape = 10 + (shape == 1 ? 3 : shape);
      if (shape) { *wide_used = true; return dspn::conv::launch_wide(shape, in, w, bias, out, g, s, residual); }
    }
  }
  if (cfg == 0 && eight) return launch_nt<4, 2, 1, 2>(in, w, bias, out, g, s, splits, per, ws.ptr, residual);
  switch (cfg) {
    case 0: return launch_nt<2, 2, 2, 2>(in, w, bias, out, g, s, splits, per, ws.ptr, residual);
    case 1: return launch_nt<2, 2, 2, 1>(in, w, bias, out, g, s, splits, per, ws.ptr, residual);
    case 2: return launch_nt<2, 2, 1, 1>(in, w, bias, out, g, s, splits, per, ws.ptr, residual);
    default: return launch_nt<4, 1, 2, 1>(in, w, bias, out, g, s, splits, per, ws.ptr, residual);
  }
}

struct WgradPlan { int bm; int bn; int splits; int pps; };
// pixels per split are a multiple of this in BOTH builds (the float build steps 32 pixels per k-step, the bf16 build 64):
// dspn_conv2d_wgrad_splits() -- exported once -- must size the slab buffers of either
constexpr int kPlanPK = 64;
static_assert(kPlanPK % kPK == 0, "split sizes must be whole k-steps");
// Weight-gradient decomposition: tile height by Cout, split-K over pixels.  The number of splits is
// the one that minimises a small cost model: workgroups run in rounds of `slots` (workgroups the chip
// holds at once, set by LDS per workgroup), each costs its pixels plus a fixed prologue/epilogue, and
// every split adds one slab to write and re-read.  (A plain "about 1024 workgroups" rule lands just
// past a round boundary for the 3x3 layers: 36 tiles x 29 splits = 1044 = 2.04 rounds.)
WgradPlan wgrad_plan(long long P, int Cout, int J, long long x_bytes = 0, long long max_splits = 1 << 30) {
  WgradPlan p;
  p.bm = Cout <= 32 ? 32 : (Cout <= 64 ? 64 : 128);
  if (p.bm == 128 && (Cout + 63) / 64 * 64 < (Cout + 127) / 128 * 128) p.bm = 64;   // less row padding
  // J = taps x channels <= 64 (the 1x1 convolutions on 64 channels of stage 1): a 128-wide column tile would be
  // half padding
  p.bn = (J <= 64 && p.bm == 128) ? 64 : 128;   // (64 x 64 tiles measured slower than 64 x 128 for Cout <= 64)
  if (const char *e = getenv("DSPN_WG_TILE")) {   // experiments: 0 32x128, 1 64x128, 2 128x64, 3 128x128
    const int t = atoi(e);
    p.bm = t == 0 ? 32 : (t == 1 ? 64 : 128);
    p.bn = t == 2 ? 64 : 128;
  }
  const long long tiles = (long long)((Cout + p.bm - 1) / p.bm) * ((J + p.bn - 1) / p.bn);
  const long long lds = 4ll * std::max(2 * kPK * (p.bm + p.bn), p.bm * (p.bn + 4));
  const long long slots = 256 * std::min<long long>(8, (160ll << 10) / lds);
  // every tap re-reads the same pixels of x: keep one split's share of x within the Infinity Cache /
  // L2 so that only the first tap's workgroups fetch it from HBM
  long long s_min = 1;
  if (x_bytes > 0) s_min = (x_bytes + (32ll << 20) - 1) / (32ll << 20);
  // (max_splits: what the caller's slab workspace holds -- a batch chunk of a >= 2 GiB tensor can want more slabs than the
  // plan of the whole batch that sized the workspace)
  const long long s_max = std::max<long long>(1, std::min<long long>(std::min<long long>(P / 128, 1024), max_splits));
  s_min = std::min(s_min, s_max);
  const double flop_per_pix = 2.0 * p.bm * p.bn, slot_rate = 120e12 / (double)slots, ovh_pix = 160;
  double best = 1e30;
  long long splits = s_min;
  for (long long sp = s_min; sp <= s_max; ++sp) {
    const long long pps = ((P + sp - 1) / sp + kPlanPK - 1) / kPlanPK * kPlanPK;
    const long long real = (P + pps - 1) / pps;
    const long long rounds = (tiles * real + slots - 1) / slots;
    const double t = (double)rounds * ((double)pps + ovh_pix) * flop_per_pix / slot_rate +
                     (double)real * Cout * J * 8.0 / 4e12;
    if (t < best * 0.999) { best = t; splits = sp; }
    if (tiles * sp > 8 * slots) break;
  }
  if (const char *e = getenv("DSPN_WG_SPLITS")) splits = std::max<long long>(1, std::min<long long>(atoll(e), s_max));   // experiments
  splits = std::min(splits, s_max);
  long long pps = ((P + splits - 1) / splits + kPlanPK - 1) / kPlanPK * kPlanPK;
  p.splits = (int)((P + pps - 1) / pps);
  p.pps = (int)pps;
  return p;
}

}  // namespace

extern "C" {

#ifdef DSPN_ABLATE
#ifdef DSPN_HALF
int dspn_debug_set_bf16(int bits) { g_debug_bits = bits; return 0; }   // the bf16-tensor kernels have their own word
#else
int dspn_debug_set(int bits) { g_debug_bits = bits; return 0; }
int dspn_debug_read_stamps(unsigned *host, int words, int clear) {
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_phase_stamps), sizeof(unsigned) * (size_t)std::min(words, 8192 * 8)) != hipSuccess) return 1;
  if (clear) { static unsigned z[8192 * 8]; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_stamps), z, sizeof(z)); }
  return 0;
}
#endif
#endif

#ifndef DSPN_HALF
size_t dspn_conv2d_split_workspace_bytes(long long out_pixels, int Cout) {
  if (out_pixels <= 0 || Cout <= 0) return 0;
  // split-K is only taken below 192 workgroups (<= 192*64 x 192*64 outputs) with <= 32 splits
  const long long capped = std::min<long long>(out_pixels * Cout, 192ll * 64 * 64 * 4);
  return sizeof(float) * 32 * (size_t)capped;
}
#endif

struct InAffine { const float *scale, *shift; int relu; };
struct OpScales { const float *a, *b; int a_planes = 0; };   // device scalars: largest magnitudes of the two operands (DSPN_MATH_F32_F16X2); a_planes: DSPN_MATH_DY_PLANES

static int conv2d_forward_one(int math, OpScales scales, const st_t *x, InAffine tf, float *stats, float *minmax, const st_t *w, const void *w_planes, const float *bias, const st_t *residual, st_t *y, int N,
                            int H, int W, int Cin, int Cout, int R, int S, int stride, int pad_h, int pad_w,
                            int dil, int Ho, int Wo, long long y_batch_stride, int y_ldc,
                            int relu, int accumulate, void *workspace, size_t workspace_bytes,
                            void *stream) {
  DSPN_REQUIRE(x && (w || w_planes) && y, "conv2d_forward: null pointer");
  DSPN_REQUIRE(Cin % kEPC == 0, "conv2d_forward: Cin must be a multiple of %d (pad channels), got %d", kEPC, Cin);
  DSPN_REQUIRE(N > 0 && H > 0 && W > 0 && Cout > 0 && R > 0 && S > 0 && stride > 0 && dil > 0,
               "conv2d_forward: bad geometry");
  DSPN_REQUIRE(Ho == (H + 2 * pad_h - dil * (R - 1) - 1) / stride + 1 &&
                   Wo == (W + 2 * pad_w - dil * (S - 1) - 1) / stride + 1,
               "conv2d_forward: output size mismatch");
  ConvGeom g;
  memset(&g, 0, sizeof(g));
  g.N = N; g.Hin = H; g.Win = W; g.Cin = Cin; g.Hg = Ho; g.Wg = Wo;
  g.ish = stride; g.isw = stride; g.ioh = -pad_h; g.iow = -pad_w; g.idh = dil; g.idw = dil;
  g.TR = R; g.TS = S; g.WTAPS = R * S; g.WS = S; g.wr0 = 0; g.wrs = 1; g.ws0 = 0; g.wss = 1;
  g.Cout = Cout;
  g.ldc = y_ldc > 0 ? y_ldc : Cout;
  g.obs = y_batch_stride > 0 ? y_batch_stride : (long long)Ho * Wo * g.ldc;
  g.OW = Wo; g.osh = 1; g.osw = 1; g.ooh = 0; g.oow = 0;
  g.dense = (g.obs == (long long)Ho * Wo * g.ldc);
  g.flags = (bias ? 1 : 0) | (relu ? 2 : 0) | (accumulate ? 4 : 0) | (residual ? 8 : 0) | ((tf.scale && tf.relu) ? 32 : 0);
  g.in_scale = tf.scale; g.in_shift = tf.shift;
  g.stats = stats;
  g.minmax = (stats && !kHalf && math == DSPN_MATH_F32_F16X2) ? minmax : nullptr;
  // round 5: `minmax` WITHOUT `stats` = the 64-slot magnitude block of the output as stored (after bias / residual / ReLU): what the
  // next convolution of a graph without BatchNorm (vgg16_reduced, the SSD extra layers) needs instead of a pass over this tensor.
  // The wide family's plain epilogue takes it along (one atomic per workgroup and launch); any other kernel is followed by that pass.
  g.bn_dy_absmax = (!stats && minmax && !kHalf && math == DSPN_MATH_F32_F16X2) ? reinterpret_cast<unsigned *>(minmax) : nullptr;
  g.bf16 = kHalf ? 1 : math;
  g.w_planes = w_planes;
  g.a_absmax = scales.a; g.b_absmax = scales.b;
  g.a_planes = scales.a_planes;
#ifndef DSPN_HALF
  // the ResNet stem (7x7 / 2 on 4 physical channels -> 64, with BatchNorm statistics) has a kernel of its own (conv_stem.h)
  if (math == DSPN_MATH_F32_F16X2 && R == 7 && S == 7 && stride == 2 && pad_h == 3 && pad_w == 3 && dil == 1 && Cin == 4 &&
      Cout == 64 && !bias && !relu && !accumulate && !residual && !tf.scale && g.dense && g.ldc == Cout && scales.a && scales.b &&
      !scales.a_planes && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
      // the stem kernel writes neither the magnitude block of a statistics-free call (out_absmax) nor any table but the
      // 64-row one: both stay with the generic kernel (and its follow-up dspn_absmax_f32 pass)
      !g.bn_dy_absmax && (!stats || kNtBm[nt_config((long long)N * Ho * Wo, Cout)] == 64)) {
    const int rc = dspn::conv::launch_stem(x, w, y, N, H, W, Cin, Cout, Ho, Wo, scales.a, scales.b, stats, g.minmax, (hipStream_t)stream);
    if (rc <= 0) return rc;
  }
#endif
  return dispatch_nt(x, w, bias, y, g, (hipStream_t)stream,
                     SplitWs{static_cast<float *>(workspace), workspace ? workspace_bytes : 0}, residual);
}

// images per launch such that no per-launch tensor reaches 2 GiB (32-bit buffer offsets, bit 31 = out of range)
static int batch_chunk(int N, long long bytes_per_image) {
  const long long lim = (1ll << 31) - 1;
  if (bytes_per_image <= 0 || N * bytes_per_image <= lim) return N;
  return (int)std::max<long long>(1, lim / bytes_per_image);
}

static int stats_layout(long long out_pixels, int Cout, int *tile_rows) {
  if (out_pixels <= 0 || Cout <= 0 || Cout % 4 != 0) return 0;
  const int bm = kNtBm[nt_config(out_pixels, Cout)];
  if (tile_rows) *tile_rows = bm;
  return (int)((out_pixels + bm - 1) / bm);
}
#ifndef DSPN_HALF
/* the same tiling for both storage types */
int dspn_conv2d_stats_layout(long long out_pixels, int Cout, int *tile_rows) { return stats_layout(out_pixels, Cout, tile_rows); }
#endif

int DSPN_FN(dspn_conv2d_forward_bn)(const st_t *x, const float *in_scale, const float *in_shift, int in_relu,
                               const st_t *w, const void *w_planes, const float *bias, const st_t *residual, st_t *y, int N,
                               int H, int W, int Cin, int Cout, int R, int S, int stride, int pad_h, int pad_w,
                               int dil, int Ho, int Wo, long long y_batch_stride, int y_ldc,
                               int relu, int accumulate, float *out_stats, size_t out_stats_bytes, float *out_minmax,
                               int math, const float *x_absmax, const float *w_absmax,
                               void *workspace, size_t workspace_bytes, void *stream) {
  DSPN_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0, "conv2d_forward: bad geometry");
  const int math_vouch = math & DSPN_MATH_UNSCALED_OK, x_planes = (math & DSPN_MATH_X_PLANES) ? 1 : 0;
  math &= ~(DSPN_MATH_UNSCALED_OK | DSPN_MATH_X_PLANES);
  DSPN_REQUIRE(math >= DSPN_MATH_FP32 && math <= DSPN_MATH_F32_F16X2, "conv2d_forward: math is one of DSPN_MATH_*");
  DSPN_REQUIRE(!x_planes || (!dspn::kHalf && math == DSPN_MATH_F32_F16X2 && Cin % 32 == 0 && x_absmax && !in_scale),
               "conv2d_forward: DSPN_MATH_X_PLANES needs DSPN_MATH_F32_F16X2, Cin %% 32 == 0, no input affine and the block the planes were cut by (x_absmax)");
  DSPN_REQUIRE(dspn::kHalf || math != DSPN_MATH_F32_F16X2 || math_vouch || (x_absmax && w_absmax),
               "conv2d_forward: DSPN_MATH_F32_F16X2 needs the magnitude block of both operands (dspn_absmax_f32); a caller who knows that every |operand| < 65504 passes math | DSPN_MATH_UNSCALED_OK");
  DSPN_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv2d_forward: in_scale and in_shift go together");
  if (out_stats) {
    int tile_rows = 0;
    const int mt = stats_layout((long long)N * Ho * Wo, Cout, &tile_rows);
    DSPN_REQUIRE(mt > 0 && out_stats_bytes >= sizeof(float) * 2 * (size_t)mt * Cout,
                 "conv2d_forward: out_stats needs dspn_conv2d_stats_layout() tiles x 2 x Cout floats and Cout %% 4 == 0");
    DSPN_REQUIRE(batch_chunk(N, (long long)sizeof(st_t) * H * W * Cin) == N, "conv2d_forward: out_stats is not available for inputs of 2 GiB or more");
  }
  const int ldc = y_ldc > 0 ? y_ldc : Cout;
  const long long ybs = y_batch_stride > 0 ? y_batch_stride : (long long)Ho * Wo * ldc;
  const int nb = batch_chunk(N, (long long)sizeof(st_t) * H * W * Cin);
  for (int n0 = 0; n0 < N; n0 += nb) {
    const int n = std::min(nb, N - n0);
    const int rc = conv2d_forward_one(math, OpScales{x_absmax, w_absmax, x_planes}, x + (long long)n0 * H * W * Cin, InAffine{in_scale, in_shift, in_relu}, out_stats, out_minmax, w, w_planes, bias,
                                      residual ? residual + (long long)n0 * ybs : nullptr, y + (long long)n0 * ybs, n, H, W,
                                      Cin, Cout, R, S, stride, pad_h, pad_w, dil, Ho, Wo, y_batch_stride, y_ldc, relu,
                                      accumulate, workspace, workspace_bytes, stream);
    if (rc) return rc;
  }
  return 0;
}

#ifndef DSPN_HALF
int dspn_conv2d_forward_f32(const float *x, const float *w, const float *bias, const float *residual, float *y, int N,
                            int H, int W, int Cin, int Cout, int R, int S, int stride, int pad_h, int pad_w,
                            int dil, int Ho, int Wo, long long y_batch_stride, int y_ldc,
                            int relu, int accumulate, void *workspace, size_t workspace_bytes,
                            void *stream) {
  return dspn_conv2d_forward_bn_f32(x, nullptr, nullptr, 0, w, nullptr, bias, residual, y, N, H, W, Cin, Cout, R, S, stride, pad_h,
                                    pad_w, dil, Ho, Wo, y_batch_stride, y_ldc, relu, accumulate, nullptr, 0, nullptr,
                                    DSPN_MATH_FP32, nullptr, nullptr, workspace, workspace_bytes, stream);
}

int dspn_conv2d_weight_transpose_f32(const float *w, float *wt, int Cout, int taps, int Cin,
                                     int Cout_pad, void *stream) {
  DSPN_REQUIRE(w && wt && Cout > 0 && taps > 0 && Cin > 0 && Cout_pad >= Cout && Cout_pad % 4 == 0,
               "weight_transpose: bad argument");
  const long long total = (long long)Cin * taps * Cout_pad;
  const int blocks = (int)std::min<long long>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(weight_transpose_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w,
                     wt, static_cast<float *>(nullptr), Cout, taps, Cin, Cout_pad);
  return dspn::check_launch("weight_transpose");
}

int dspn_conv2d_weight_transpose_batch_f32(const void *table, int n, long long total_tiles, void *stream) {
  DSPN_REQUIRE(table && n > 0 && total_tiles > 0 && total_tiles < (1ll << 31), "weight_transpose_batch: bad argument");
  static_assert(sizeof(WtDesc) == 48, "table row layout: 2 pointers, 4 ints, 1 int64, 1 pointer (NULL here)");
  hipLaunchKernelGGL(weight_transpose_batch_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream,
                     static_cast<const WtDesc *>(table), n);
  return dspn::check_launch("weight_transpose_batch");
}
long long dspn_conv2d_weight_transpose_tiles(int Cout, int taps, int Cin, int Cout_pad) {
  if (Cout <= 0 || taps <= 0 || Cin <= 0 || Cout_pad < Cout) return 0;
  return (long long)((Cout_pad + 31) >> 5) * taps * ((Cin + 31) >> 5);
}

/* piece planes of a weight operand (DSPN_MATH_F32_BF16X3, include/dspn_nn.h) */
int dspn_conv2d_weight_planes_f32(const float *w, void *planes, void *planes_t, int Cout, int taps, int Cin, int cols_t,
                                  int pieces, const float *w_absmax, void *stream) {
  DSPN_REQUIRE(w && (planes || planes_t) && Cout > 0 && taps > 0 && Cin > 0 && Cin % 4 == 0, "weight_planes: bad argument");
  DSPN_REQUIRE(pieces == 3 || (pieces == 2 && w_absmax), "weight_planes: pieces is 3 (bf16) or 2 (fp16, with the weight's magnitude block)");
  DSPN_REQUIRE(!planes || Cin % 32 == 0, "weight_planes: the forward planes need Cin %% 32 == 0, got %d", Cin);
  DSPN_REQUIRE(!planes_t || (cols_t % 32 == 0 && cols_t >= Cout), "weight_planes: cols_t is the padded Cout, a multiple of 32");
  WpDesc e{w, static_cast<__bf16 *>(planes), static_cast<__bf16 *>(planes_t), Cout, taps, Cin, planes_t ? cols_t : 0, 0, w_absmax, pieces, 0};
  const long long tiles = weight_planes_tiles(Cout, taps, Cin, cols_t, planes_t != nullptr);
  DSPN_REQUIRE(tiles < (1ll << 31), "weight_planes: too many tiles");
  hipLaunchKernelGGL(weight_planes_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, e);
  return dspn::check_launch("weight_planes");
}

long long dspn_conv2d_weight_planes_tiles(int Cout, int taps, int Cin, int cols_t, int with_transposed) {
  if (Cout <= 0 || taps <= 0 || Cin <= 0) return 0;
  return weight_planes_tiles(Cout, taps, Cin, cols_t, with_transposed != 0);
}

/* operand magnitudes for DSPN_MATH_F32_F16X2 (include/dspn_nn.h) */
int dspn_absmax_f32(const float *x, long long rows, int C, const float *scale, const float *shift, int relu,
                    float *out_dev, void *stream) {
  DSPN_REQUIRE(x && out_dev && rows > 0 && C > 0 && C % 4 == 0, "absmax: bad argument (C must be a multiple of 4)");
  DSPN_REQUIRE((scale == nullptr) == (shift == nullptr), "absmax: scale and shift go together");
  const long long n4 = rows * (C / 4);
  const int blocks = (int)std::min<long long>((n4 + 255) / 256, 2048);     // (an even count: 512 float4 channel groups divide it)
  hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4 *>(x), n4,
                     C / 4, reinterpret_cast<const float4 *>(scale), reinterpret_cast<const float4 *>(shift), relu,
                     reinterpret_cast<unsigned *>(out_dev));
  return dspn::check_launch("absmax");
}

int dspn_absmax_batch_f32(const void *table, int n, long long total_chunks, void *stream) {
  DSPN_REQUIRE(table && n > 0 && total_chunks > 0 && total_chunks < (1ll << 31), "absmax_batch: bad argument");
  static_assert(sizeof(AmDesc) == 32, "table row layout: 2 pointers, 2 int64");
  hipLaunchKernelGGL(absmax_batch_kernel, dim3((unsigned)total_chunks), dim3(256), 0, (hipStream_t)stream,
                     static_cast<const AmDesc *>(table), n);
  return dspn::check_launch("absmax_batch");
}

int dspn_conv2d_weight_planes_batch_f32(const void *table, int n, long long total_tiles, void *stream) {
  DSPN_REQUIRE(table && n > 0 && total_tiles > 0 && total_tiles < (1ll << 31), "weight_planes_batch: bad argument");
  static_assert(sizeof(WpDesc) == 64, "table row layout: 3 pointers, 4 ints, 1 int64, 1 pointer, 2 ints");
  hipLaunchKernelGGL(weight_planes_batch_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream,
                     static_cast<const WpDesc *>(table), n);
  return dspn::check_launch("weight_planes_batch");
}
#else
/* bf16 operands of a float master weight w [Cout][taps][Cin]: wt [Cin][taps][Cout_pad] (data gradient) and, when
 * wh != NULL, the copy wh [Cout][taps][Cin] (forward).  Cout_pad % 8 == 0. */
int dspn_conv2d_weight_prepare_bf16(const float *w, st_t *wh, st_t *wt, int Cout, int taps, int Cin,
                                    int Cout_pad, void *stream) {
  DSPN_REQUIRE(w && wt && Cout > 0 && taps > 0 && Cin > 0 && Cout_pad >= Cout && Cout_pad % kEPC == 0,
               "weight_prepare: bad argument");
  const long long total = (long long)Cin * taps * Cout_pad;
  const int blocks = (int)std::min<long long>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(weight_transpose_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w,
                     wt, wh, Cout, taps, Cin, Cout_pad);
  return dspn::check_launch("weight_prepare");
}

int dspn_conv2d_weight_prepare_batch_bf16(const void *table, int n, long long total_tiles, void *stream) {
  DSPN_REQUIRE(table && n > 0 && total_tiles > 0 && total_tiles < (1ll << 31), "weight_prepare_batch: bad argument");
  static_assert(sizeof(WtDesc) == 48, "table row layout: 2 pointers, 4 ints, 1 int64, 1 pointer");
  hipLaunchKernelGGL(weight_transpose_batch_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream,
                     static_cast<const WtDesc *>(table), n);
  return dspn::check_launch("weight_prepare_batch");
}
#endif

// dx (N,H,W,Cin_x) from dy (N,Ho,Wo,ldy) and wt = transposed weights [Cin_x][R*S][ldy].
// Also the forward of a transposed convolution (x := dy).
struct BnBwd { const st_t *x; const float *scale, *shift, *mean, *rstd; int relu; float *sums; float *dy_absmax; };

// row tiles of the launches of one data gradient, in launch order (stride 2: up to 4 parity classes)
static int dgrad_tiles(int N, int H, int W, int Cin, int stride, int *per_class /* [4] or NULL */) {
  int total = 0;
  for (int ph = 0; ph < (stride == 1 ? 1 : 2); ++ph)
    for (int pw = 0; pw < (stride == 1 ? 1 : 2); ++pw) {
      const long long hg = stride == 1 ? H : (H - ph + 1) / 2, wg = stride == 1 ? W : (W - pw + 1) / 2;
      const long long M = (long long)N * hg * wg;
      const int t = M > 0 ? (int)((M + kNtBm[nt_config(M, Cin)] - 1) / kNtBm[nt_config(M, Cin)]) : 0;
      if (per_class) per_class[ph * 2 + pw] = t;
      total += t;
    }
  return total;
}

static int conv2d_dgrad_one(int math, OpScales scales, const st_t *dy, const st_t *wt, const void *wt_planes, st_t *dx, int N, int H, int W,
                          int Cin, int ldy, int R, int S, int stride, int pad_h, int pad_w, int dil, int Ho,
                          int Wo, int dx_ldc, int accumulate, BnBwd bn, void *workspace, size_t workspace_bytes,
                          void *stream) {
  DSPN_REQUIRE(dy && (wt || wt_planes) && dx, "conv2d_dgrad: null pointer");
  DSPN_REQUIRE(ldy % kEPC == 0, "conv2d_dgrad: dy channel stride must be a multiple of %d", kEPC);
  DSPN_REQUIRE(stride == 1 || (stride == 2 && dil == 1), "conv2d_dgrad: stride 1, or stride 2 with dilation 1");
  DSPN_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && R > 0 && S > 0, "conv2d_dgrad: bad geometry");
  ConvGeom g;
  memset(&g, 0, sizeof(g));
  g.N = N; g.Hin = Ho; g.Win = Wo; g.Cin = ldy; g.Cout = Cin;
  g.WTAPS = R * S; g.WS = S;
  g.ldc = dx_ldc > 0 ? dx_ldc : Cin;
  g.obs = (long long)H * W * g.ldc;
  g.OW = W;
  g.flags = accumulate ? 4 : 0;
  g.bf16 = kHalf ? 1 : math;
  g.w_planes = wt_planes;
  g.a_absmax = scales.a; g.b_absmax = scales.b;
  g.bn_x = bn.x; g.bn_scale = bn.scale; g.bn_shift = bn.shift; g.bn_mean = bn.mean; g.bn_rstd = bn.rstd;
  g.bn_relu = bn.relu; g.bn_sums = bn.sums; g.bn_tile_base = 0;
  g.bn_dy_absmax = (bn.sums && !kHalf && math == DSPN_MATH_F32_F16X2) ? reinterpret_cast<unsigned *>(bn.dy_absmax) : nullptr;
  g.a_planes = scales.a_planes;
  int class_tiles[4] = {0, 0, 0, 0};
  if (bn.sums) dgrad_tiles(N, H, W, Cin, stride, class_tiles);
  hipStream_t s = (hipStream_t)stream;
  const SplitWs sws{static_cast<float *>(workspace), workspace ? workspace_bytes : 0};
  if (stride == 1) {
    g.Hg = H; g.Wg = W; g.ish = 1; g.isw = 1; g.ioh = pad_h; g.iow = pad_w; g.idh = -dil; g.idw = -dil;
    g.TR = R; g.TS = S; g.wr0 = 0; g.wrs = 1; g.ws0 = 0; g.wss = 1;
    g.osh = 1; g.osw = 1; g.dense = 1;
    return dispatch_nt(dy, wt, nullptr, dx, g, s, sws);
  }
  for (int ph = 0; ph < 2; ++ph)
    for (int pw = 0; pw < 2; ++pw) {
      ConvGeom c = g;
      c.Hg = (H - ph + 1) / 2; c.Wg = (W - pw + 1) / 2;
      for (int q = 0; q < ph * 2 + pw; ++q) c.bn_tile_base += class_tiles[q];
      if (c.Hg <= 0 || c.Wg <= 0) continue;
      const int r0 = (ph + pad_h) & 1, s0 = (pw + pad_w) & 1;
      c.TR = r0 < R ? (R - r0 + 1) / 2 : 0;
      c.TS = s0 < S ? (S - s0 + 1) / 2 : 0;
      if (c.TR == 0 || c.TS == 0) {
        c.TR = 0; c.TS = 1;
        if (accumulate && !bn.sums) continue;   // nothing to add (with BatchNorm sums the class still has to be read)
      }
      c.ish = 1; c.isw = 1; c.ioh = (ph + pad_h - r0) / 2; c.iow = (pw + pad_w - s0) / 2;
      c.idh = -1; c.idw = -1;
      c.wr0 = r0; c.wrs = 2; c.ws0 = s0; c.wss = 2;
      c.osh = 2; c.osw = 2; c.ooh = ph; c.oow = pw; c.dense = 0;
      const int rc = dispatch_nt(dy, wt, nullptr, dx, c, s, sws);
      if (rc) return rc;
    }
  return 0;
}

static int dgrad_bn_tiles(int N, int H, int W, int Cin, int stride) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin % 4 != 0 || (stride != 1 && stride != 2)) return 0;
  return dgrad_tiles(N, H, W, Cin, stride, nullptr);
}
#ifndef DSPN_HALF
int dspn_conv2d_dgrad_bn_tiles(int N, int H, int W, int Cin, int stride) { return dgrad_bn_tiles(N, H, W, Cin, stride); }
#endif

int DSPN_FN(dspn_conv2d_dgrad_bn)(const st_t *dy, const st_t *wt, const void *wt_planes, st_t *dx, int N, int H, int W,
                             int Cin, int ldy, int R, int S, int stride, int pad_h, int pad_w, int dil, int Ho,
                             int Wo, int dx_ldc, int accumulate,
                             const st_t *bn_x, const float *bn_scale, const float *bn_shift, const float *bn_mean,
                             const float *bn_rstd, int bn_relu, float *bn_sums, size_t bn_sums_bytes, float *bn_dy_absmax,
                             int math, const float *dy_absmax, const float *w_absmax, void *workspace, size_t workspace_bytes,
                             void *stream) {
  DSPN_REQUIRE(N > 0 && Ho > 0 && Wo > 0 && ldy > 0, "conv2d_dgrad: bad geometry");
  const int math_vouch = math & DSPN_MATH_UNSCALED_OK, dy_planes = (math & DSPN_MATH_DY_PLANES) ? 1 : 0;
  math &= ~(DSPN_MATH_UNSCALED_OK | DSPN_MATH_DY_PLANES);
  DSPN_REQUIRE(!dy_planes || (!dspn::kHalf && math == DSPN_MATH_F32_F16X2 && ldy % 32 == 0 && dy_absmax),
               "conv2d_dgrad: DSPN_MATH_DY_PLANES needs DSPN_MATH_F32_F16X2, ldy %% 32 == 0 and the block the planes were cut by (dy_absmax)");
  DSPN_REQUIRE(math >= DSPN_MATH_FP32 && math <= DSPN_MATH_F32_F16X2, "conv2d_dgrad: math is one of DSPN_MATH_*");
  DSPN_REQUIRE(dspn::kHalf || math != DSPN_MATH_F32_F16X2 || math_vouch || (dy_absmax && w_absmax),
               "conv2d_dgrad: DSPN_MATH_F32_F16X2 needs the magnitude block of both operands (dspn_absmax_f32); a caller who knows that every |operand| < 65504 passes math | DSPN_MATH_UNSCALED_OK");
  const int ldc = dx_ldc > 0 ? dx_ldc : Cin;
  const int nb = batch_chunk(N, (long long)sizeof(st_t) * Ho * Wo * ldy);
  if (bn_sums) {
    DSPN_REQUIRE(bn_x && bn_mean && bn_rstd && (!bn_relu || (bn_scale && bn_shift)), "conv2d_dgrad: BatchNorm operands missing");
    DSPN_REQUIRE(ldc == Cin && nb == N, "conv2d_dgrad: BatchNorm sums need a dense dx and dy below 2 GiB");
    const int tiles = dgrad_bn_tiles(N, H, W, Cin, stride);
    DSPN_REQUIRE(tiles > 0 && bn_sums_bytes >= sizeof(float) * 2 * (size_t)tiles * Cin,
                 "conv2d_dgrad: bn_sums needs dspn_conv2d_dgrad_bn_tiles() x 2 x Cin floats");
  }
  for (int n0 = 0; n0 < N; n0 += nb) {
    const int n = std::min(nb, N - n0);
    const int rc = conv2d_dgrad_one(math, OpScales{dy_absmax, w_absmax, dy_planes}, dy + (long long)n0 * Ho * Wo * ldy, wt, wt_planes, dx + (long long)n0 * H * W * ldc, n, H,
                                    W, Cin, ldy, R, S, stride, pad_h, pad_w, dil, Ho, Wo, dx_ldc, accumulate,
                                    BnBwd{bn_x, bn_scale, bn_shift, bn_mean, bn_rstd, bn_relu, bn_sums, bn_dy_absmax}, workspace,
                                    workspace_bytes, stream);
    if (rc) return rc;
  }
  return 0;
}

#ifndef DSPN_HALF
int dspn_conv2d_dgrad_f32(const float *dy, const float *wt, float *dx, int N, int H, int W,
                          int Cin, int ldy, int R, int S, int stride, int pad_h, int pad_w, int dil, int Ho,
                          int Wo, int dx_ldc, int accumulate, void *workspace, size_t workspace_bytes,
                          void *stream) {
  return dspn_conv2d_dgrad_bn_f32(dy, wt, nullptr, dx, N, H, W, Cin, ldy, R, S, stride, pad_h, pad_w, dil, Ho, Wo, dx_ldc,
                                  accumulate, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr, DSPN_MATH_FP32,
                                  nullptr, nullptr, workspace, workspace_bytes, stream);
}

size_t dspn_conv2d_input_sum_grad_workspace_bytes(int Ho, int Wo, int ldy, int R, int S) {
  return sizeof(float) * ((size_t)Ho * Wo * ldy + (size_t)R * S * 32 * ldy);
}
#endif

int DSPN_FN(dspn_conv2d_input_sum_grad)(const st_t *dy, const float *w, float *out, int N, int H, int W,
                                   int Cin, int Cout, int ldy, int R, int S, int stride, int pad_h, int pad_w, int dil,
                                   int Ho, int Wo, void *workspace, size_t workspace_bytes, void *stream) {
  DSPN_REQUIRE(dy && w && out && workspace, "conv2d_input_sum_grad: null pointer");
  DSPN_REQUIRE(ldy % 4 == 0 && Cin >= 1 && Cin <= 8, "conv2d_input_sum_grad: Cin <= 8, ldy % 4 == 0");
  if (workspace_bytes < sizeof(float) * ((size_t)Ho * Wo * ldy + (size_t)R * S * 32 * ldy))
    return dspn::fail(DSPN_ERR_WORKSPACE_, "conv2d_input_sum_grad: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float *p2 = static_cast<float *>(workspace);
  float *partial = p2 + (size_t)Ho * Wo * ldy;
  const long long per4 = (long long)Ho * Wo * ldy / 4;
  hipLaunchKernelGGL(batch_sum_kernel, dim3((int)std::min<long long>((per4 + 255) / 256, 8192)), dim3(256), 0, s,
                     dspn::CA4Ptr(dy), reinterpret_cast<float4 *>(p2), N, per4);
  SumGradGeom g{H, W, Ho, Wo, ldy, R, S, stride, stride, pad_h, pad_w, dil, dil, 32};
  hipLaunchKernelGGL(tap_sum_kernel, dim3(R * S, 32), dim3(256), 0, s, p2, partial, g);
  hipLaunchKernelGGL(sum_grad_final_kernel, dim3(1), dim3(1024), 0, s, partial, w, out, R * S, 32, ldy, Cout, Cin);
  return dspn::check_launch("conv2d_input_sum_grad");
}

#ifndef DSPN_HALF
size_t dspn_conv2d_wgrad_workspace_bytes(int N, int Ho, int Wo, int Cin, int Cout, int R, int S) {
  const long long P = (long long)N * Ho * Wo;
  const int J = R * S * Cin;
  if (P <= 0 || J <= 0 || Cout <= 0) return 0;
  // the plan depends on (P, Cout, J) and, for R*S > 1, on the stride (1 or 2) through the x footprint
  size_t m = (size_t)wgrad_plan(P, Cout, J).splits;
  for (int stride = 1; stride <= 2; ++stride)
    m = std::max(m, (size_t)wgrad_plan(P, Cout, J, 4ll * P * Cin * stride * stride).splits);
  return sizeof(float) * m * Cout * J;
}
#endif

static int conv2d_wgrad_one(int math, OpScales scales, const st_t *x, InAffine tf, const st_t *dy, float *dw, int N, int H, int W, int Cin,
                          int Cout, int ldy, int R, int S, int stride, int pad_h, int pad_w, int dil, int Ho,
                          int Wo, int accumulate, void *workspace, size_t workspace_bytes,
                          void *stream) {
  DSPN_REQUIRE(x && dy && workspace, "conv2d_wgrad: null pointer");   // dw == NULL: leave the partial slabs in workspace
  DSPN_REQUIRE(Cin % kEPC == 0 && ldy % kEPC == 0, "conv2d_wgrad: channel strides must be multiples of %d", kEPC);
  WgradGeom g;
  g.N = N; g.Hin = H; g.Win = W; g.Cin = Cin; g.Ho = Ho; g.Wo = Wo; g.Cout = Cout; g.ldy = ldy;
  g.sh = stride; g.sw = stride; g.ph = pad_h; g.pw = pad_w; g.dh = dil; g.dw = dil; g.R = R; g.S = S;
  g.in_scale = tf.scale; g.in_shift = tf.shift; g.in_relu = tf.relu;
  g.bf16 = kHalf ? 1 : math;
  g.dy_absmax = scales.a; g.x_absmax = scales.b;
  g.dy_planes = scales.a_planes & 1; g.x_planes = (scales.a_planes >> 1) & 1;
  {
    const long long xb = (long long)sizeof(st_t) * N * H * W * Cin, yb = (long long)sizeof(st_t) * N * Ho * Wo * ldy;
    if (xb >= (1ll << 31) || yb >= (1ll << 31))
      return dspn::fail(DSPN_ERR_ARG_, "conv2d_wgrad: tensors of 2 GiB or more are not supported");
    g.x_bytes = (unsigned)xb; g.dy_bytes = (unsigned)yb;
  }
  const long long P = (long long)N * Ho * Wo;
  const int J = R * S * Cin;
  const long long ws_splits = (long long)(workspace_bytes / (sizeof(float) * (size_t)Cout * J));
  if (ws_splits < 1)
    return dspn::fail(DSPN_ERR_WORKSPACE_, "conv2d_wgrad: workspace %zu < %zu (one slab)", workspace_bytes, sizeof(float) * (size_t)Cout * J);
  const WgradPlan plan = wgrad_plan(P, Cout, J, R * S > 1 ? 4ll * P * Cin * std::min(stride, 2) * std::min(stride, 2) : 0, ws_splits);
  const int BM = plan.bm, BN = plan.bn;
  const int kt = (Cout + BM - 1) / BM, jt = (J + BN - 1) / BN;
  const long long splits = plan.splits;
  g.pix_per_split = plan.pps;
  const size_t need = sizeof(float) * (size_t)splits * Cout * J;
  if (workspace_bytes < need)
    return dspn::fail(DSPN_ERR_WORKSPACE_, "conv2d_wgrad: workspace %zu < %zu", workspace_bytes, need);
  hipStream_t s = (hipStream_t)stream;
  float *slab = static_cast<float *>(workspace);
  // round 6: a parked BatchNorm-backward finalize of this stream rides in front of the grid (bn_final_job.h)
  g.job = dspn::BnFinalJob{};
  g.job_rows = dspn::bn_job_take(s, &g.job) ? (g.job.blocks + kt * jt - 1) / (kt * jt) : 0;
  // mainloop buffers | staged output tile
  // (the float build sizes for its largest mode: float images 2 * kPK * (BM + BN) * 4 B; the three-piece bf16 image of the
  // split mode, single-buffered, 3 * kPK * row bytes, is smaller than the staged output tile for every tile shape but 32 x 128)
  const size_t lds = kHalf ? std::max<size_t>(2 * (size_t)kPK * (wg_row_bytes(BM) + wg_row_bytes(BN)), sizeof(float) * BM * (BN + 4))
                           : std::max<size_t>(sizeof(float) * std::max(2 * kPK * (BM + BN), BM * (BN + 4)),
                                              3 * (size_t)kPK * (wg_row_bytes(BM) + wg_row_bytes(BN)));
  dspn::ProfScope prof(1, s);
#ifdef DSPN_HALF
#define DSPN_WGRAD_LAUNCH(WM, WN, TM_, TN_)                                                              \
  {                                                                                                      \
    if (g.in_scale) DSPN_WGRAD_LAUNCH_(WM, WN, TM_, TN_, 1, true)                                        \
    else DSPN_WGRAD_LAUNCH_(WM, WN, TM_, TN_, 1, false)                                                  \
  }
#else
#define DSPN_WGRAD_LAUNCH(WM, WN, TM_, TN_)                                                              \
  {                                                                                                      \
    if (g.bf16 == 3 && g.x_planes) {                                                                     \
      if (g.dy_planes) DSPN_WGRAD_LAUNCH_(WM, WN, TM_, TN_, 6, false)                                    \
      else DSPN_WGRAD_LAUNCH_(WM, WN, TM_, TN_, 5, false)                                                \
    } else if (g.in_scale) {                                                                             \
      if (g.bf16 == 3 && g.dy_planes) DSPN_WGRAD_LAUNCH_(WM, WN, TM_, TN_, 4, true)                      \
      else if (g.bf16 == 3) DSPN_WGRAD_LAUNCH_(WM, WN, TM_, TN_, 3, true)                                \
      else if (g.bf16 == 2) DSPN_WGRAD_LAUNCH_(WM, WN, TM_, TN_, 2, true)                                \
      else if (g.bf16) DSPN_WGRAD_LAUNCH_(WM, WN, TM_, TN_, 1, true)                                     \
      else DSPN_WGRAD_LAUNCH_(WM, WN, TM_, TN_, 0, true)                                                 \
    } else {                                                                                             \
      if (g.bf16 == 3 && g.dy_planes) DSPN_WGRAD_LAUNCH_(WM, WN, TM_, TN_, 4, false)                     \
      else if (g.bf16 == 3) DSPN_WGRAD_LAUNCH_(WM, WN, TM_, TN_, 3, false)                               \
      else if (g.bf16 == 2) DSPN_WGRAD_LAUNCH_(WM, WN, TM_, TN_, 2, false)                               \
      else if (g.bf16) DSPN_WGRAD_LAUNCH_(WM, WN, TM_, TN_, 1, false)                                    \
      else DSPN_WGRAD_LAUNCH_(WM, WN, TM_, TN_, 0, false)                                                \
    }                                                                                                    \
  }
#endif
#define DSPN_WGRAD_LAUNCH_(WM, WN, TM_, TN_, BF, TF)                                                     \
  {                                                                                                      \
    auto kern = conv_wgrad_kernel<WM, WN, TM_, TN_, BF, TF>;                                             \
    static dspn::KernelDeviceState st;                                                                   \
    if (const int dev = dspn::ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds, st, "conv_wgrad"); dev < 0) return dev; \
    hipLaunchKernelGGL(kern, dim3(kt * jt, (int)splits + g.job_rows), dim3(WM * WN * 64), lds, s, x, dy, slab, g, kt, jt); \
  }
#ifndef DSPN_HALF
  // round 6: both operands as piece planes on a 128 x 128 tile -- global -> LDS directly, 64 x 64 per wave (conv_wgrad_wide.h);
  // same split plan, same slabs, same bits.  dspn_conv_set_wide_tiles(1) / DSPN_WGW=0: conv_wgrad_kernel (tests, same-box A/B)
  static const bool wgw_env = [] { const char *e = getenv("DSPN_WGW"); return !(e && atoi(e) == 0); }();
  // (eight waves on 128 x 256 with a three-slot ring measured no faster where the plan fills the chip -- stages 3 and 4: 0.37 - 0.39
  // of 833.3 either way -- and 1.5 x slower where it does not; profiles/r06_wgrad_wide_tile.txt)
  if (g.bf16 == 3 && g.x_planes && g.dy_planes && BM == 128 && BN == 128 && wgw_env && dspn::wide_tiles_mode() != 1) {
    auto kern = conv_wgw_kernel<2, 2, 2>;
    const size_t lds_w = std::max<size_t>(2 * (size_t)kPK * (128 + 128) * 4, sizeof(float) * 128 * (128 + 4));
    static dspn::KernelDeviceState st;
    if (const int dev = dspn::ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds_w, st, "conv_wgrad (wide)"); dev < 0) return dev;
    hipLaunchKernelGGL(kern, dim3(kt * jt, (int)splits + g.job_rows), dim3(256), lds_w, s, x, dy, slab, g, kt, jt);
  } else
#endif
  if (BM == 32) DSPN_WGRAD_LAUNCH(1, 4, 1, 1)                     // 32 x 128
  else if (BM == 64) DSPN_WGRAD_LAUNCH(2, 2, 1, 2)                // 64 x 128
  else if (BN == 64) DSPN_WGRAD_LAUNCH(2, 2, 2, 1)                // 128 x 64
  else DSPN_WGRAD_LAUNCH(4, 2, 1, 2)   // 128 x 128 on 8 waves: two workgroups = 4 waves per SIMD (+3..6 % over <2,2,2,2>)
#undef DSPN_WGRAD_LAUNCH
#undef DSPN_WGRAD_LAUNCH_
  int rc = dspn::check_launch("conv_wgrad");
  if (rc || !dw) return rc;
  const long long n4 = (long long)Cout * J / 4;
  const int blocks = (int)std::min<long long>((n4 + 255) / 256, 2048);
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, s,
                     reinterpret_cast<const float4 *>(slab), reinterpret_cast<float4 *>(dw), n4,
                     (int)splits, accumulate);
  return dspn::check_launch("conv_wgrad_reduce");
}

int DSPN_FN(dspn_conv2d_wgrad_bn)(const st_t *x, const float *in_scale, const float *in_shift, int in_relu,
                             const st_t *dy, float *dw, int N, int H, int W, int Cin,
                             int Cout, int ldy, int R, int S, int stride, int pad_h, int pad_w, int dil, int Ho,
                             int Wo, int accumulate, int math, const float *x_absmax, const float *dy_absmax,
                             void *workspace, size_t workspace_bytes, void *stream) {
  DSPN_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Ho > 0 && Wo > 0 && ldy > 0, "conv2d_wgrad: bad geometry");
  DSPN_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv2d_wgrad: in_scale and in_shift go together");
  const int math_vouch = math & DSPN_MATH_UNSCALED_OK;
  const int dy_planes = ((math & DSPN_MATH_DY_PLANES) ? 1 : 0) | ((math & DSPN_MATH_X_PLANES) ? 2 : 0);     // bit 0: dy, bit 1: x
  math &= ~(DSPN_MATH_UNSCALED_OK | DSPN_MATH_DY_PLANES | DSPN_MATH_X_PLANES);
  DSPN_REQUIRE(!(dy_planes & 1) || (!dspn::kHalf && math == DSPN_MATH_F32_F16X2 && ldy == Cout && Cout % 32 == 0 && dy_absmax),
               "conv2d_wgrad: DSPN_MATH_DY_PLANES needs DSPN_MATH_F32_F16X2, ldy == Cout, Cout %% 32 == 0 and the block the planes were cut by (dy_absmax)");
  DSPN_REQUIRE(!(dy_planes & 2) || (!dspn::kHalf && math == DSPN_MATH_F32_F16X2 && Cin % 32 == 0 && x_absmax && !in_scale),
               "conv2d_wgrad: DSPN_MATH_X_PLANES needs DSPN_MATH_F32_F16X2, Cin %% 32 == 0, no input affine and the block the planes were cut by (x_absmax)");
  DSPN_REQUIRE(math >= DSPN_MATH_FP32 && math <= DSPN_MATH_F32_F16X2, "conv2d_wgrad: math is one of DSPN_MATH_*");
  DSPN_REQUIRE(dspn::kHalf || math != DSPN_MATH_F32_F16X2 || math_vouch || (x_absmax && dy_absmax),
               "conv2d_wgrad: DSPN_MATH_F32_F16X2 needs the magnitude block of both operands (dspn_absmax_f32); a caller who knows that every |operand| < 65504 passes math | DSPN_MATH_UNSCALED_OK");
  const int nb = std::min(batch_chunk(N, (long long)sizeof(st_t) * H * W * Cin),
                          batch_chunk(N, (long long)sizeof(st_t) * Ho * Wo * ldy));
  for (int n0 = 0; n0 < N; n0 += nb) {
    const int n = std::min(nb, N - n0);
    const int rc = conv2d_wgrad_one(math, OpScales{dy_absmax, x_absmax, dy_planes}, x + (long long)n0 * H * W * Cin, InAffine{in_scale, in_shift, in_relu},
                                    dy + (long long)n0 * Ho * Wo * ldy, dw, n, H, W,
                                    Cin, Cout, ldy, R, S, stride, pad_h, pad_w, dil, Ho, Wo, accumulate || n0 > 0,
                                    workspace, workspace_bytes, stream);
    if (rc) return rc;
  }
  return 0;
}

#ifndef DSPN_HALF
/* number of split-K slabs ([splits][Cout][R*S*Cin] floats) the weight gradient of this geometry produces */
int dspn_conv2d_wgrad_splits(int N, int Ho, int Wo, int Cin, int Cout, int R, int S, int stride) {
  const long long P = (long long)N * Ho * Wo;
  const int J = R * S * Cin;
  if (P <= 0 || J <= 0 || Cout <= 0) return 0;
  return wgrad_plan(P, Cout, J, R * S > 1 ? 4ll * P * Cin * std::min(stride, 2) * std::min(stride, 2) : 0).splits;
}
#endif

/* the weight-gradient GEMM alone: the split-K partial sums stay in `slabs` (dspn_conv2d_wgrad_splits() x Cout x
 * R*S*Cin floats) for a later dspn_conv2d_slab_reduce_batch_f32 */
int DSPN_FN(dspn_conv2d_wgrad_slabs)(const st_t *x, const float *in_scale, const float *in_shift, int in_relu,
                                const st_t *dy, float *slabs, size_t slabs_bytes, int N, int H, int W, int Cin,
                                int Cout, int ldy, int R, int S, int stride, int pad_h, int pad_w, int dil, int Ho,
                                int Wo, int math, const float *x_absmax, const float *dy_absmax, void *stream) {
  DSPN_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Ho > 0 && Wo > 0 && ldy > 0, "conv2d_wgrad: bad geometry");
  DSPN_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv2d_wgrad: in_scale and in_shift go together");
  const int math_vouch = math & DSPN_MATH_UNSCALED_OK;
  const int dy_planes = ((math & DSPN_MATH_DY_PLANES) ? 1 : 0) | ((math & DSPN_MATH_X_PLANES) ? 2 : 0);     // bit 0: dy, bit 1: x
  math &= ~(DSPN_MATH_UNSCALED_OK | DSPN_MATH_DY_PLANES | DSPN_MATH_X_PLANES);
  DSPN_REQUIRE(!(dy_planes & 1) || (!dspn::kHalf && math == DSPN_MATH_F32_F16X2 && ldy == Cout && Cout % 32 == 0 && dy_absmax),
               "conv2d_wgrad: DSPN_MATH_DY_PLANES needs DSPN_MATH_F32_F16X2, ldy == Cout, Cout %% 32 == 0 and the block the planes were cut by (dy_absmax)");
  DSPN_REQUIRE(!(dy_planes & 2) || (!dspn::kHalf && math == DSPN_MATH_F32_F16X2 && Cin % 32 == 0 && x_absmax && !in_scale),
               "conv2d_wgrad: DSPN_MATH_X_PLANES needs DSPN_MATH_F32_F16X2, Cin %% 32 == 0, no input affine and the block the planes were cut by (x_absmax)");
  DSPN_REQUIRE(math >= DSPN_MATH_FP32 && math <= DSPN_MATH_F32_F16X2, "conv2d_wgrad: math is one of DSPN_MATH_*");
  DSPN_REQUIRE(dspn::kHalf || math != DSPN_MATH_F32_F16X2 || math_vouch || (x_absmax && dy_absmax),
               "conv2d_wgrad: DSPN_MATH_F32_F16X2 needs the magnitude block of both operands (dspn_absmax_f32); a caller who knows that every |operand| < 65504 passes math | DSPN_MATH_UNSCALED_OK");
  DSPN_REQUIRE(std::min(batch_chunk(N, (long long)sizeof(st_t) * H * W * Cin),
                        batch_chunk(N, (long long)sizeof(st_t) * Ho * Wo * ldy)) == N,
               "conv2d_wgrad_slabs: tensors of 2 GiB or more need dspn_conv2d_wgrad_f32");
  return conv2d_wgrad_one(math, OpScales{dy_absmax, x_absmax, dy_planes}, x, InAffine{in_scale, in_shift, in_relu}, dy, nullptr, N, H, W, Cin, Cout, ldy, R, S, stride,
                          pad_h, pad_w, dil, Ho, Wo, 0, slabs, slabs_bytes, stream);
}

#ifndef DSPN_HALF
/* table: n rows of 40 bytes in DEVICE memory, { const float *slabs; float *dw; int64 n4 (= Cout*R*S*Cin/4);
 * int32 splits, accumulate; int64 begin (= sum of n4 over the preceding rows) }; total4 = sum of n4 */
int dspn_conv2d_slab_reduce_batch_f32(const void *table, int n, long long total4, void *stream) {
  DSPN_REQUIRE(table && n > 0 && total4 > 0, "slab_reduce_batch: bad argument");
  static_assert(sizeof(SlabDesc) == 40, "table row layout: 2 pointers, int64, 2 ints, int64");
  const int blocks = (int)std::min<long long>((total4 + 255) / 256, 16384);
  dspn::ProfScope prof(1, (hipStream_t)stream);   // part of the weight-gradient family's time
  hipLaunchKernelGGL(slab_reduce_batch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                     static_cast<const SlabDesc *>(table), n, total4);
  return dspn::check_launch("slab_reduce_batch");
}

int dspn_conv2d_wgrad_f32(const float *x, const float *dy, float *dw, int N, int H, int W, int Cin,
                          int Cout, int ldy, int R, int S, int stride, int pad_h, int pad_w, int dil, int Ho,
                          int Wo, int accumulate, void *workspace, size_t workspace_bytes,
                          void *stream) {
  return dspn_conv2d_wgrad_bn_f32(x, nullptr, nullptr, 0, dy, dw, N, H, W, Cin, Cout, ldy, R, S, stride, pad_h, pad_w, dil,
                                  Ho, Wo, accumulate, DSPN_MATH_FP32, nullptr, nullptr, workspace, workspace_bytes, stream);
}

#endif

}  // extern "C"
