// The WIDE tile family of the two-piece convolution math (round 5).  Included by conv.hip (float build), inside its anonymous
// namespace, behind conv_nt_kernel: same ConvGeom, same K order, same epilogues, same results to the last bit of the
// accumulation order per output -- what changes is how a workgroup is fed.
//
// conv_nt_kernel's 128 x 128 tile on 8 waves of 32 x 64 (128 registers, two workgroups per CU) asks the texture path for
// 32 KiB per k-step and issues 12 LDS fragment reads for 12 MFMAs per wave; its stamps showed a
// k-step of ~3100 cycles of which ~980 are MFMA, the rest queueing for shared units.  Here:
//   * both operands are PURE COPIES (A: fp16 piece planes of the gathered tensor, B: piece planes of the weights), so both go
//     global -> LDS directly (buffer_load ... lds, 1 KiB per wave-instruction): no staging registers, no ds_write, no vector
//     arithmetic in the k-loop besides a handful of address instructions.  A tap outside the image (or a row past M) is an
//     out-of-range buffer offset: the hardware writes ZEROS into the LDS image for those lanes (measured on gfx950:
//     profiles/r05_lds_dma_out_of_range_probe.txt), so padding needs no branch and no second instruction;
//   * a wave owns 64 x 64 outputs (2 x 2 accumulators of 32 x 32): 16 fragment reads feed 24 MFMAs per k-step, and a
//     256 x 128 (128 x 256) workgroup asks for 48 KiB per k-step for TWICE the multiply-adds of the 128 x 128 tile;
//   * the LDS images are a ring of STAGES k-steps filled STAGES - 1 ahead behind a COUNTED s_waitcnt vmcnt(N) and a raw
//     s_barrier (a __syncthreads() would drain the ring: cdna_hip_programming.md section 5, "Pipelining across barriers"): one
//     barrier per k-step, the loads of the next k-steps stay in flight across it;
//   * the images are unpadded 128-byte rows (a DMA destination is lane-linear), XOR-swizzled on the SOURCE address and on
//     the fragment reads exactly as conv_nt_kernel's DMA_B image: 16-byte slot s = (row & 1) * 8 + chunk of bank line
//     L = row >> 1 holds what an unswizzled image keeps in slot s ^ (L & 7): conflict-free ds_read_b128.
// One workgroup per CU at 256 registers (8 waves), or two at 256 registers (4 waves, 128 x 128).
//
// BatchNorm tables keep their layout whatever the tile -- 128 rows per entry, 64 for the layers with at most 64 output columns
// (template parameter SR): a 256-row tile writes two (four) rows of g.stats / g.minmax / g.bn_sums, so
// dspn_conv2d_stats_layout / dspn_conv2d_dgrad_bn_tiles and every consumer are untouched by the routing.

// Phase stamps (scratch/r06/xt_bench.hip builds with -DDSPN_STAMPS; compiled out of the library): wave 0 of every workgroup
// adds the shader-clock time between consecutive DSPN_STAMP(i) points into g_wide_stamps[workgroup][i].  A stamp waits for the
// scalar read of the clock (lgkmcnt(0): the wave's LDS operations as well), not for vector memory.
#ifdef DSPN_STAMPS
__device__ unsigned long long g_wide_stamps[1024 * 16];
struct WideStamps {
  unsigned long long last, sum[16];
  __device__ __forceinline__ void begin() { for (int i = 0; i < 16; ++i) sum[i] = 0; last = __builtin_readcyclecounter(); }
  __device__ __forceinline__ void mark(int i) { const unsigned long long t = __builtin_readcyclecounter(); sum[i] += t - last; last = t; }
  __device__ __forceinline__ void flush() {
    if (threadIdx.x == 0 && blockIdx.x < 1024) for (int i = 0; i < 16; ++i) g_wide_stamps[blockIdx.x * 16 + i] = sum[i];
  }
};
#define DSPN_STAMP_DECL WideStamps stamps_; stamps_.begin()
#define DSPN_STAMP(i) stamps_.mark(i)
#define DSPN_STAMP_FLUSH stamps_.flush()
#define DSPN_STAMP_ARG , stamps_
#define DSPN_STAMP_PARAM , WideStamps &stamps_
#else
#define DSPN_STAMP_DECL
#define DSPN_STAMP(i)
#define DSPN_STAMP_FLUSH
#define DSPN_STAMP_ARG
#define DSPN_STAMP_PARAM
#endif

// dspn_conv_set_tile_spanning / DSPN_XT=0: the tile-spanning loop off (tests, same-box A/B runs; the results do not depend on it)
inline bool xt_enabled() { return dspn::tile_spanning() != 0; }
// The eight-wave 128 x 256 member on the tile-spanning loop too: bit-identical (tests/test_wide_tiles_gpu.py) and small on the step --
// its layers have 36+ k-steps and one to four tiles per workgroup, the epilogue is a tenth of a tile: 954.6 -> 958.8 images/s
// (+0.2 ... +0.5 % in each of four alternating pairs on one box, beside the float-operand kernels' direct epilogue).  Default;
// DSPN_XT8=0 keeps it on the round-5 loop (A/B runs).
inline bool xt_wide8_enabled() {
  static const bool on = [] { const char *e = getenv("DSPN_XT8"); return !e || atoi(e) != 0; }();
  return on;
}
// The 256 x 64 members (<= 64 output columns, 64-row statistics tiles: stage 1 of the ResNets) on the round-6 loops too: bit-identical
// stored tensors (tests), +0.2 ... +0.5 % on the step in each of three alternating pairs on one box.  Default; DSPN_XT64=0 keeps
// them on the round-5 loops (A/B runs).
inline bool xt_c64_enabled() {
  static const bool on = [] { const char *e = getenv("DSPN_XT64"); return !e || atoi(e) != 0; }();
  return on;
}
// ... whose direct epilogue addresses the output (and the tensors of its shape) as one buffer of M rows of ldc elements
inline bool xt_output_ok(const ConvGeom &g, const int bm = 128) {
  const long long M = (long long)g.N * g.Hg * g.Wg;
  return g.dense && (g.flags & 16) && M % bm == 0 && M * g.ldc * (long long)sizeof(st_t) < (1ll << 31);
}

// The tile epilogue of the wide family, from the staged fp32 tile in LDS (st[row * (BN + 4) + col], written by the caller, which
// has NOT yet met the barrier that publishes it) to the stored outputs and BatchNorm tables; ends with the barrier after which
// the LDS may be overwritten.  conv_nt_kernel's epilogue per 128-row half: same arithmetic, same tables.
struct NoStage {};
template <int BM, int BN, int NTHR, int EPI, int SR>
__device__ __forceinline__ void wide_epilogue(const ConvGeom &g, char *wsm, const int m0, const int n0, const int M, const int tid,
                                              const float *__restrict__ bias, st_t *__restrict__ out,
                                              const st_t *__restrict__ residual, float &gmx_all, NoStage
                                              DSPN_STAMP_PARAM) {
  constexpr int HALVES = BM / SR;          // statistics tiles (SR rows: 128, or 64 for the Cout <= 64 layers) per output tile
  static_assert(BM % SR == 0, "whole statistics tiles");
  const bool has_bias = g.flags & 1, relu = g.flags & 2, accum = g.flags & 4, has_res = g.flags & 8;
  constexpr int SLD = BN + 4;
  constexpr int C4 = BN / 4, RPP = NTHR / C4, NPH = SR / RPP;   // float4 columns per row, rows per pass, passes per half
  constexpr int RC = NPH > 8 ? 8 : NPH;                          // rows in flight per thread
  float *st = reinterpret_cast<float *>(wsm);
  const int c4 = tid % C4, er0 = tid / C4;
  const int co = n0 + c4 * 4;
  const bool cvalid = co < g.Cout;                               // (Cout % 4 == 0 and aligned rows: the host routes nothing else)
  const st_t *addsrc = has_res ? residual : (accum ? out : nullptr);
  auto ld4 = [](const st_t *q) __attribute__((always_inline)) { return dspn::CA1Ptr(q).vec4()[0]; };
  float bsc[4] = {0.f, 0.f, 0.f, 0.f}, bsh[4] = {0.f, 0.f, 0.f, 0.f}, bmu[4] = {0.f, 0.f, 0.f, 0.f}, brs[4] = {0.f, 0.f, 0.f, 0.f};
  if (EPI == 2 && cvalid) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bmu[e] = g.bn_mean[co + e]; brs[e] = g.bn_rstd[co + e];
      if (g.bn_relu) { bsc[e] = g.bn_scale[co + e]; bsh[e] = g.bn_shift[co + e]; }
    }
  }
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (has_bias && cvalid) {
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = bias[co + e];
  }
  constexpr float kInf = __builtin_huge_valf();
  float sK[HALVES][4], s1[HALVES][4], s2[HALVES][4], vmn[HALVES][4], vmx[HALVES][4], gs[HALVES][4], gss[HALVES][4];
  int scnt[HALVES];
  float gmx = 0.f;
#pragma unroll
  for (int h = 0; h < HALVES; ++h) {
    scnt[h] = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) { sK[h][e] = 0.f; s1[h][e] = 0.f; s2[h][e] = 0.f; vmn[h][e] = kInf; vmx[h][e] = -kInf; gs[h][e] = 0.f; gss[h][e] = 0.f; }
  }
  // The rows of a chunk (RC per thread) need their residual / BatchNorm-input rows from global memory; the requests of chunk
  // c + 1 are issued before chunk c is worked on (two register sets; the accumulators are dead by now), and those of chunk 0
  // above the barrier that publishes the staged tile: on the short-K layers of the residual stream the epilogue moves more
  // bytes than the k-loop, and a chunk that waits for its own requests leaves one memory latency per chunk exposed.
  constexpr int CPH = NPH / RC, NCH = HALVES * CPH;              // chunks per half, chunks per tile (2 or 4)
  int offs[2][RC];
  float4 rq[2][RC], xq[2][EPI == 2 ? RC : 1];
  auto issue = [&](const int c, const int b) __attribute__((always_inline)) {
    const int h = c / CPH, ch = c - h * CPH;
#pragma unroll
    for (int p = 0; p < RC; ++p) {
      const int m = m0 + h * SR + er0 + (ch * RC + p) * RPP;
      int off;
      if (g.dense) {
        off = m * g.ldc + co;
      } else {
        const int hw = g.Hg * g.Wg;
        const int n = m / hw, rem = m - n * hw;
        const int oi = rem / g.Wg, oj = rem - oi * g.Wg;
        off = n * (int)g.obs + ((oi * g.osh + g.ooh) * g.OW + (oj * g.osw + g.oow)) * g.ldc + co;
      }
      if (m >= M || !cvalid) off = -1;
      offs[b][p] = off;
      rq[b][p] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (addsrc && off >= 0) rq[b][p] = ld4(addsrc + off);
      if constexpr (EPI == 2) {
        xq[b][p] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (off >= 0) xq[b][p] = ld4(g.bn_x + off);
      }
    }
  };
  issue(0, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  DSPN_STAMP(4);       // staged tile published
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int h = c / CPH, ch = c % CPH, b = c & 1;              // (compile-time after unrolling)
    if (c + 1 < NCH) issue(c + 1, b ^ 1);
#pragma unroll
    for (int p = 0; p < RC; ++p) {
      const int off = offs[b][p];
      if (off < 0) continue;
      const float4 tv = *reinterpret_cast<const float4 *>(st + (h * SR + er0 + (ch * RC + p) * RPP) * SLD + c4 * 4);
      float v[4] = {tv.x + bv[0], tv.y + bv[1], tv.z + bv[2], tv.w + bv[3]};
      v[0] += rq[b][p].x; v[1] += rq[b][p].y; v[2] += rq[b][p].z; v[3] += rq[b][p].w;
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
      dspn::A1Ptr(out + off).vec4()[0] = make_float4(v[0], v[1], v[2], v[3]);
      if constexpr (EPI == 0) {      // the magnitude of the stored output, when asked for (g.bn_dy_absmax without BatchNorm sums)
        if (g.bn_dy_absmax) gmx = fmaxf(gmx, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
      }
      if constexpr (EPI == 2) {
        const float xv[4] = {xq[b][p].x, xq[b][p].y, xq[b][p].z, xq[b][p].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float gd = (!g.bn_relu || fmaf(xv[e], bsc[e], bsh[e]) > 0.f) ? v[e] : 0.f;
          gs[h][e] += gd;
          gss[h][e] += gd * ((xv[e] - bmu[e]) * brs[e]);
        }
        gmx = fmaxf(gmx, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
      }
      if constexpr (EPI == 1) {
        if (scnt[h] == 0) { sK[h][0] = v[0]; sK[h][1] = v[1]; sK[h][2] = v[2]; sK[h][3] = v[3]; }
        ++scnt[h];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = v[e] - sK[h][e];
          s1[h][e] += d; s2[h][e] += d * d;
          vmn[h][e] = fminf(vmn[h][e], v[e]); vmx[h][e] = fmaxf(vmx[h][e], v[e]);
        }
      }
    }
  }
  DSPN_STAMP(5);       // rows read, stores issued
  if constexpr (EPI == 1) {
    // per-thread (mean, M2) of its rows of each half -> LDS -> one thread per (half, column) merges the RPP row groups with
    // Chan's update in a fixed order -> stats[128-row tile][mean | M2][column]; the extremes (g.minmax) travel with them
    __builtin_amdgcn_s_barrier();      // every staged row has been read
    float *red = reinterpret_cast<float *>(wsm);             // [HALVES][RPP][BN][2], then the extremes, same shape
    float *red2 = red + HALVES * RPP * BN * 2;
    const bool mmx = !kHalf && g.minmax != nullptr;           // (kernel-uniform)
    if (cvalid) {
#pragma unroll
      for (int h = 0; h < HALVES; ++h) {
        const float inv = scnt[h] > 0 ? 1.f / (float)scnt[h] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          red[((h * RPP + er0) * BN + c4 * 4 + e) * 2] = sK[h][e] + s1[h][e] * inv;
          red[((h * RPP + er0) * BN + c4 * 4 + e) * 2 + 1] = s2[h][e] - s1[h][e] * s1[h][e] * inv;
          if (mmx) {
            red2[((h * RPP + er0) * BN + c4 * 4 + e) * 2] = vmn[h][e];
            red2[((h * RPP + er0) * BN + c4 * 4 + e) * 2 + 1] = vmx[h][e];
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int hc = tid; hc < HALVES * BN; hc += NTHR) {
      const int h = hc / BN, col = hc - h * BN;
      const int lim = min(M - (m0 + h * SR), SR);
      if (n0 + col >= g.Cout || lim <= 0) continue;
      // merge of the RPP row groups about the first group's mean (no division inside the loop):
      //   mean = m_0 + sum n_e d_e / n,  M2 = sum (M2_e + n_e d_e^2) - n (mean - m_0)^2,  d_e = mean_e - m_0
      const float *rh = red + (size_t)h * RPP * BN * 2, *rh2 = red2 + (size_t)h * RPP * BN * 2;
      const float mref = rh[col * 2];                        // row group 0 is never empty
      float n = 0.f, sd = 0.f, sq = 0.f;
#pragma unroll 4
      for (int er = 0; er < RPP; ++er) {
        const int ne_i = lim > er ? (lim - er + RPP - 1) / RPP : 0;
        const float ne = (float)ne_i;
        const float d = ne_i > 0 ? rh[(er * BN + col) * 2] - mref : 0.f;
        const float m2e = ne_i > 0 ? rh[(er * BN + col) * 2 + 1] : 0.f;
        n += ne; sd += ne * d; sq += m2e + ne * d * d;
      }
      const float dm = sd / n;
      const long long mt_ = m0 / SR + h;
      g.stats[(mt_ * 2 + 0) * g.Cout + n0 + col] = mref + dm;
      g.stats[(mt_ * 2 + 1) * g.Cout + n0 + col] = fmaxf(sq - n * dm * dm, 0.f);
      if (mmx) {
        float mn = kInf, mx = -kInf;
        for (int er = 0; er < RPP && er < lim; ++er) { mn = fminf(mn, rh2[(er * BN + col) * 2]); mx = fmaxf(mx, rh2[(er * BN + col) * 2 + 1]); }
        g.minmax[(mt_ * 2 + 0) * g.Cout + n0 + col] = mn;
        g.minmax[(mt_ * 2 + 1) * g.Cout + n0 + col] = mx;
      }
    }
  }
  if constexpr (EPI == 2) {
    __builtin_amdgcn_s_barrier();      // every staged row has been read
    float *red = reinterpret_cast<float *>(wsm);             // [HALVES][RPP][BN][2]
    if (cvalid) {
#pragma unroll
      for (int h = 0; h < HALVES; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          red[((h * RPP + er0) * BN + c4 * 4 + e) * 2] = gs[h][e];
          red[((h * RPP + er0) * BN + c4 * 4 + e) * 2 + 1] = gss[h][e];
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int hc = tid; hc < HALVES * BN; hc += NTHR) {
      const int h = hc / BN, col = hc - h * BN;
      if (n0 + col >= g.Cout || m0 + h * SR >= M) continue;
      const float *rh = red + (size_t)h * RPP * BN * 2;
      float a = 0.f, b = 0.f;
      for (int er = 0; er < RPP; ++er) { a += rh[(er * BN + col) * 2]; b += rh[(er * BN + col) * 2 + 1]; }
      const long long mt_ = g.bn_tile_base + m0 / SR + h;
      g.bn_sums[(mt_ * 2 + 0) * g.Cout + n0 + col] = a;
      g.bn_sums[(mt_ * 2 + 1) * g.Cout + n0 + col] = b;
    }
    gmx_all = fmaxf(gmx_all, gmx);
  }
  if constexpr (EPI == 0) gmx_all = fmaxf(gmx_all, gmx);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();        // the staging area / the exchange has been read: the next tile's images may land
  DSPN_STAMP(6);       // tables written, closing barrier
}

#ifndef DSPN_HALF
// ---- round 6: the DIRECT epilogue of the tile-spanning loops (XT) -- from the accumulators as they stand to memory, no staging
// in LDS, so the ring keeps receiving the next tile's images while this runs and nothing but a 4-KiB exchange of per-column
// partial sums is shared between the waves (one barrier per tile where the staged epilogue meets three or four).
// C/D layout of a 32 x 32 block: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5): one register of a block is two
// 128-byte row segments per wave instruction -- the shape MI355X_MICROARCH.md measures at the full store rate.  Dense outputs
// only (row m at m ldc) and whole row tiles (M % 128 == 0: the host routes nothing else here); columns past Cout are
// out-of-range buffer offsets: the hardware drops their stores and returns zeros for their loads, so every wave issues the same
// instructions (the counted waits of the k-loops rely on that).
// Arithmetic per element as wide_epilogue's (same order of the bias / residual / accumulate additions: the stored values are
// the same bits); the per-column reductions run over a lane's 32 rows, then over the four (wave row, half-wave) partials in a
// fixed order -- another summation order than the staged epilogue's, i.e. tables equal within rounding, deterministic.
// SR: rows of a statistics tile (128, or 64 on the <= 64-column layers: one wave row each); CLOSE: end with a barrier (the exchange
// lies in a ring slot the next tile's first requests are aimed at)
template <int WAVES_M, int WAVES_N, int EPI, int RB = 16, int SR = 128, bool CLOSE = false>       // RB: rows of a 32 x 32 block whose operand rows are in flight together
__device__ __forceinline__ void direct_epilogue(const ConvGeom &g, char *exch, f32x16 (&acc)[2][2], const float inv_a, const float inv_b, const int m0,
                                                const int n0, const int M, const int tid, const int wave,
                                                const float *__restrict__ bias, float *__restrict__ out,
                                                const float *__restrict__ residual, float &gmx_all) {
  constexpr int TM = 2, TN = 2, BN = WAVES_N * 64, NTHR = WAVES_M * WAVES_N * 64, PARTS = WAVES_M * 2;
  constexpr int WRT = SR / 64, TILES = WAVES_M / WRT, PPT = WRT * 2;       // wave rows per statistics tile, tiles per output tile, partials per tile
  static_assert(SR == 64 || SR == 128, "statistics tiles of 64 or 128 rows");
  static_assert(WAVES_M % WRT == 0, "whole statistics tiles per output tile");
  const int lane = tid & 63, half = lane >> 5, lc = lane & 31;
  const int wr = wave / WAVES_N, wm = wr * 64, wn = (wave % WAVES_N) * 64;
  const bool has_bias = g.flags & 1, relu = g.flags & 2, accum = g.flags & 4, has_res = g.flags & 8;
  const float *addsrc = has_res ? residual : (accum ? out : nullptr);
  const unsigned obytes = (unsigned)M * (unsigned)g.ldc * 4u;       // (the host checks M ldc 4 < 2^31)
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(out, 0, obytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(addsrc ? addsrc : out), 0, addsrc ? obytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(EPI == 2 ? g.bn_x : out), 0, EPI == 2 ? obytes : 0u, 0x00020000);
  constexpr unsigned kOOB = 0x80000000u;
  constexpr float kInf = __builtin_huge_valf();
  int col[TN];
  unsigned cbyte[TN];
  bool cv[TN];
  float bv[TN], bsc[TN], bsh[TN], bmu[TN], brs[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    col[j] = n0 + wn + j * 32 + lc;
    cv[j] = col[j] < g.Cout;
    cbyte[j] = cv[j] ? (unsigned)col[j] * 4u : kOOB;
    bv[j] = (has_bias && cv[j]) ? bias[col[j]] : 0.f;
    bsc[j] = bsh[j] = bmu[j] = brs[j] = 0.f;
    if (EPI == 2 && cv[j]) {
      bmu[j] = g.bn_mean[col[j]]; brs[j] = g.bn_rstd[col[j]];
      if (g.bn_relu) { bsc[j] = g.bn_scale[col[j]]; bsh[j] = g.bn_shift[col[j]]; }
    }
  }
  const int row0 = m0 + wm + 4 * half;                       // row of (i, r): row0 + 32 i + (r & 3) + 8 (r >> 2)
  const unsigned pitch = (unsigned)g.ldc * 4u;
  // per-column partials of this lane
  float sK[TN], s1[TN], s2[TN], vmn[TN], vmx[TN], gs[TN], gss[TN];
  int scnt = 0;
  float gmx = 0.f;
#pragma unroll
  for (int j = 0; j < TN; ++j) { sK[j] = 0.f; s1[j] = 0.f; s2[j] = 0.f; vmn[j] = kInf; vmx[j] = -kInf; gs[j] = 0.f; gss[j] = 0.f; }
  const bool both = has_res && accum;
  // The rows a block adds to (residual / accumulate) and the BatchNorm-input rows come from memory in batches of RB rows; the
  // batch after the one being worked on is requested first (two register sets, as the staged epilogue does with its chunks): a
  // batch that waited for its own requests would leave one memory latency per batch exposed.  Batch b = rows r0 .. r0 + RB - 1
  // of block i, b = i (16 / RB) + r0 / RB.
  constexpr int NB = TM * 16 / RB;
  const bool need_rows = addsrc != nullptr || EPI == 2;      // (kernel-uniform)
  if constexpr (EPI == 2) {
    // (one row at a time: the packed form below needs register pairs for the BatchNorm-input rows as well, which 256 registers
    // do not hold beside the accumulators and the next tile's rows)
    float rq[2][RB][TN], xq[2][RB][TN], oq[2][RB][TN];
    auto request = [&](const int bb, const int s_) __attribute__((always_inline)) {
      const int i = bb / (16 / RB), r0 = (bb % (16 / RB)) * RB;
#pragma unroll
      for (int rr = 0; rr < RB; ++rr) {
        const int r = r0 + rr;
        const unsigned rb = (unsigned)(row0 + 32 * i + (r & 3) + 8 * (r >> 2)) * pitch;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const unsigned off = rb + cbyte[j];
          rq[s_][rr][j] = addsrc ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_a, (int)off, 0, 0)) : 0.f;
          xq[s_][rr][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, (int)off, 0, 0));
          oq[s_][rr][j] = both ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_o, (int)off, 0, 0)) : 0.f;
        }
      }
    };
    request(0, 0);
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) {
      const int i = bb / (16 / RB), r0 = (bb % (16 / RB)) * RB, s_ = bb & 1;
      if (bb + 1 < NB) request(bb + 1, s_ ^ 1);
#pragma unroll
      for (int rr = 0; rr < RB; ++rr) {
        const int r = r0 + rr;
        const unsigned rb = (unsigned)(row0 + 32 * i + (r & 3) + 8 * (r >> 2)) * pitch;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float v = acc[i][j][r] * inv_a * inv_b + bv[j];
          v += rq[s_][rr][j];
          if (both) v += oq[s_][rr][j];
          if (relu) v = v > 0.f ? v : 0.f;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_o, (int)(rb + cbyte[j]), 0, 0);
          const float xv = xq[s_][rr][j];
          const float gd = (!g.bn_relu || fmaf(xv, bsc[j], bsh[j]) > 0.f) ? v : 0.f;
          gs[j] += gd;
          gss[j] += gd * ((xv - bmu[j]) * brs[j]);
          gmx = fmaxf(gmx, fabsf(v));
        }
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) { gs[j] = cv[j] ? gs[j] : 0.f; gss[j] = cv[j] ? gss[j] : 0.f; }
  } else {
    // every row of the tile exists (the host routes M % 128 == 0 here): no row tests, and the element arithmetic on PAIRS of rows
    // (r, r + 1: neighbouring accumulator registers) so that the scaling, the additions and the running sums are packed fp32
    // instructions -- the epilogue's vector arithmetic is of the order of its store time on the short-K layers
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 s1p[TN], s2p[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { s1p[j] = f32x2{0.f, 0.f}; s2p[j] = f32x2{0.f, 0.f}; }
    f32x2 rq[2][RB / 2][TN], oq[2][RB / 2][TN];
    auto request = [&](const int bb, const int s_) __attribute__((always_inline)) {
      const int i = bb / (16 / RB), r0 = (bb % (16 / RB)) * RB;
#pragma unroll
      for (int rr = 0; rr < RB; rr += 2) {
        const int r = r0 + rr;
        const unsigned rb = (unsigned)(row0 + 32 * i + (r & 3) + 8 * (r >> 2)) * pitch;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const unsigned off = rb + cbyte[j];
          rq[s_][rr / 2][j] = f32x2{0.f, 0.f}; oq[s_][rr / 2][j] = f32x2{0.f, 0.f};
          if (addsrc) rq[s_][rr / 2][j] = f32x2{__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_a, (int)off, 0, 0)),
                                               __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_a, (int)(off + pitch), 0, 0))};
          if (both) oq[s_][rr / 2][j] = f32x2{__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_o, (int)off, 0, 0)),
                                             __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_o, (int)(off + pitch), 0, 0))};
        }
      }
    };
    if (need_rows) request(0, 0);
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) {
      const int i = bb / (16 / RB), r0 = (bb % (16 / RB)) * RB, s_ = bb & 1;
      if (need_rows && bb + 1 < NB) request(bb + 1, s_ ^ 1);
#pragma unroll
      for (int rr = 0; rr < RB; rr += 2) {
        const int r = r0 + rr;
        const unsigned rb = (unsigned)(row0 + 32 * i + (r & 3) + 8 * (r >> 2)) * pitch;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          f32x2 t = f32x2{acc[i][j][r], acc[i][j][r + 1]} * inv_a * inv_b + bv[j];
          if (need_rows) t += rq[s_][rr / 2][j];
          else t += f32x2{0.f, 0.f};
          // (the two conditional steps element by element: hipcc 7.2 drops the second element of a select on a two-float vector)
          float v0 = t[0], v1 = t[1];
          if (both) { v0 += oq[s_][rr / 2][j][0]; v1 += oq[s_][rr / 2][j][1]; }
          if (relu) { v0 = v0 > 0.f ? v0 : 0.f; v1 = v1 > 0.f ? v1 : 0.f; }
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), rs_o, (int)(rb + cbyte[j]), 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v1), rs_o, (int)(rb + pitch + cbyte[j]), 0, 0);
          const f32x2 v = f32x2{v0, v1};
          if constexpr (EPI == 0) {
            if (g.bn_dy_absmax) gmx = fmaxf(fmaxf(gmx, fabsf(v0)), fabsf(v1));
          }
          if constexpr (EPI == 1) {
            if (bb == 0 && rr == 0) sK[j] = v0;
            const f32x2 d = v - sK[j];
            s1p[j] += d; s2p[j] += d * d;
            vmn[j] = fminf(fminf(vmn[j], v0), v1); vmx[j] = fmaxf(fmaxf(vmx[j], v0), v1);
          }
        }
      }
    }
    scnt = 32;
#pragma unroll
    for (int j = 0; j < TN; ++j) { s1[j] = s1p[j].x + s1p[j].y; s2[j] = s2p[j].x + s2p[j].y; }
  }
  if constexpr (EPI == 0) gmx_all = fmaxf(gmx_all, gmx);
  if constexpr (EPI == 2) gmx_all = fmaxf(gmx_all, gmx);
  if constexpr (EPI != 0) {
    // partial p = 2 (wave row) + half-wave of column c at red[(p * BN + c) * 4 ..]: EPI 1 (count, mean, M2) and, at red2, (min, max);
    // EPI 2 (sum, sum x-hat).  The previous tile's exchange was read before the k-loop barriers every wave has passed since.
    float *red = reinterpret_cast<float *>(exch);
    float *red2 = red + PARTS * BN * 4;
    const bool mmx = EPI == 1 && g.minmax != nullptr;
    const int p = wr * 2 + half;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float *q = red + (p * BN + wn + j * 32 + lc) * 4;
      if constexpr (EPI == 1) {
        const float inv_n = scnt > 0 ? 1.f / (float)scnt : 0.f;
        q[0] = (float)scnt; q[1] = sK[j] + s1[j] * inv_n; q[2] = s2[j] - s1[j] * s1[j] * inv_n;
        if (mmx) { float *q2 = red2 + (p * BN + wn + j * 32 + lc) * 2; q2[0] = vmn[j]; q2[1] = vmx[j]; }
      } else {
        q[0] = gs[j]; q[1] = gss[j];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int hc = tid; hc < TILES * BN; hc += NTHR) {
      const int h = hc / BN, c = hc - h * BN;                 // statistics tile h of this output tile: partials h PPT .. h PPT + PPT - 1
      if (n0 + c >= g.Cout || m0 + h * SR >= M) continue;
      const long long mt_ = m0 / SR + h;
      if constexpr (EPI == 1) {
        // Chan's update over the partials in a fixed order (empty partials -- rows past M -- skipped)
        float n = 0.f, mean = 0.f, m2 = 0.f, mn = kInf, mx = -kInf;
#pragma unroll
        for (int pq = 0; pq < PPT; ++pq) {
          const int pp = h * PPT + pq;
          const float *q = red + (pp * BN + c) * 4;
          const float nb = q[0];
          if (nb > 0.f) {
            const float d = q[1] - mean, nn = n + nb;
            mean += d * (nb / nn);
            m2 += q[2] + d * d * (n * nb / nn);
            n = nn;
            if (mmx) { mn = fminf(mn, red2[(pp * BN + c) * 2]); mx = fmaxf(mx, red2[(pp * BN + c) * 2 + 1]); }
          }
        }
        g.stats[(mt_ * 2 + 0) * g.Cout + n0 + c] = mean;
        g.stats[(mt_ * 2 + 1) * g.Cout + n0 + c] = fmaxf(m2, 0.f);
        if (mmx) {
          g.minmax[(mt_ * 2 + 0) * g.Cout + n0 + c] = mn;
          g.minmax[(mt_ * 2 + 1) * g.Cout + n0 + c] = mx;
        }
      } else {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int pq = 0; pq < PPT; ++pq) { a += red[((h * PPT + pq) * BN + c) * 4]; b += red[((h * PPT + pq) * BN + c) * 4 + 1]; }
        const long long t_ = g.bn_tile_base + mt_;
        g.bn_sums[(t_ * 2 + 0) * g.Cout + n0 + c] = a;
        g.bn_sums[(t_ * 2 + 1) * g.Cout + n0 + c] = b;
      }
    }
    if constexpr (CLOSE) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();        // the exchange has been read: the ring slot it lies in may be requested into
    }
  }
}
#endif   // !DSPN_HALF

// g.bn_dy_absmax: the largest |dx| this workgroup stored, over ALL its tiles -- one atomic per workgroup and launch
template <int NWV>
__device__ __forceinline__ void wide_publish_absmax(const ConvGeom &g, char *wsm, float gmx_all, const int tid) {
  const int lane = tid & 63, wave = tid >> 6;
    if (g.bn_dy_absmax) {       // (kernel-uniform)
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) gmx_all = fmaxf(gmx_all, __shfl_xor(gmx_all, o, 64));
      float *sm = reinterpret_cast<float *>(wsm);
      if (lane == 0) sm[wave] = gmx_all;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (tid == 0) {
        float m = sm[0];
        for (int q = 1; q < NWV; ++q) m = fmaxf(m, sm[q]);
        if (m > 0.f) atomicMax(g.bn_dy_absmax + (blockIdx.x & 63u), __float_as_uint(m));
      }
    }
}

// XT (round 6): the TILE-SPANNING loop (first for the short-K layers -- 1 x 1 convolutions of 2 .. 16 k-steps, where the epilogue
// is most of the kernel and the first images of a tile used to be requested only after the previous tile's last store -- then
// for every member with 128-row tiles): the ring runs on across tile boundaries -- the requests the plain loop issues "past
// the last k-step" (out of range, zeros) are here the LIVE requests of the next tile's first k-steps -- and the epilogue works
// from the accumulators (direct_epilogue), so the next tile's images are on their way while the current tile is written out.
// Same K order, same epilogue arithmetic per element: the bits of the plain loop.
template <int WAVES_M, int WAVES_N, int STAGES, int EPI, int SR = 128, bool XT = false>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, 2) void conv_ntw_kernel(
    const st_t *__restrict__ in, const st_t *__restrict__ wgt, const float *__restrict__ bias,
    st_t *__restrict__ out, const ConvGeom g, const int m_tiles, const int n_tiles,
    const st_t *__restrict__ residual) {
  constexpr int TM = 2, TN = 2;
  // a 128-byte row record of an image: float build -- two fp16 pieces of 32 channels (the planes); bf16 build -- 64 channels of
  // the tensor itself (its activations and weight copies ARE the operands: a k-step is four 16-deep MFMA blocks, one product)
  constexpr int KB = 128 / (int)sizeof(st_t);            // channels per record
  constexpr int NKK = kHalf ? 4 : 2, NPROD = kHalf ? 1 : 3;
  constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
  constexpr int NWV = WAVES_M * WAVES_N, NTHR = NWV * 64;
  constexpr int A_NI = BM / (8 * NWV), B_NI = BN / (8 * NWV), NI = A_NI + B_NI;   // 1-KiB pieces per wave and k-step
  static_assert(BM % (8 * NWV) == 0 && BN % (8 * NWV) == 0, "whole pieces per wave");
  constexpr int STG = (BM + BN) * 128;                   // bytes of one k-step's images: A rows, then B rows
  constexpr int D = STAGES - 1;                          // k-steps in flight ahead of the one being multiplied
  static_assert(STAGES >= 2 && STAGES <= 4, "ring depth");
  static_assert(BM % SR == 0, "BatchNorm tables are per SR rows");
  static_assert(!XT || !kHalf, "the tile-spanning loop: float build");
  extern __shared__ __attribute__((aligned(1024))) char wsm[];

#ifdef DSPN_ABLATE
  // timing-only ablation build (results WRONG when non-zero): 1 no requests inside the k-loop, 2 no MFMAs, 4 no fragment reads,
  // 8 no barrier, 16 no epilogue
  const int dbg = g.dbg;
#else
  constexpr int dbg = 0;
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = m_tiles * n_tiles;
  const int M = g.N * g.Hg * g.Wg;
  const int CB = g.Cin / KB;                             // channel blocks (records) per tap
  const int ntaps = g.TR * g.TS;
  const int nk = ntaps * CB;                             // (the host routes nk >= 1 only)
  const float sc_a = kHalf ? 1.f : operand_scale(g.a_absmax), sc_b = kHalf ? 1.f : operand_scale(g.b_absmax);
  const float inv_a = 1.f / sc_a, inv_b = 1.f / sc_b;    // exact: powers of two
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<st_t *>(in), 0, g.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<st_t *>(wgt), 0, g.w_bytes, 0x00020000);
  constexpr unsigned kOOB = 0x80000000u;

  // ---- loader: lane l of piece q (bank lines 4q .. 4q+3 = tile rows 8q .. 8q+7) fills slot l & 15 of line 4q + (l >> 4), i.e.
  // fetches chunk c of row r with (r & 1) * 8 + c = (l & 15) ^ (line & 7)
  int a_row[A_NI], b_row[B_NI], a_cb[A_NI], b_cb[B_NI];
#pragma unroll
  for (int i = 0; i < A_NI; ++i) {
    const int line = 4 * (wave * A_NI + i) + (lane >> 4), sl = (lane & 15) ^ (line & 7);
    a_row[i] = 2 * line + (sl >> 3); a_cb[i] = (sl & 7) * 16;
  }
#pragma unroll
  for (int i = 0; i < B_NI; ++i) {
    const int line = 4 * (wave * B_NI + i) + (lane >> 4), sl = (lane & 15) ^ (line & 7);
    b_row[i] = 2 * line + (sl >> 3); b_cb[i] = (sl & 7) * 16;
  }
  int a_ih0[A_NI], a_iw0[A_NI], a_boff[A_NI], b_boff[B_NI];   // state of ONE output tile
  int ld_m0 = 0, ld_n0 = 0;
  int l_tr = 0, l_ts = 0, l_cb = 0;                           // (tap, channel block) of the next k-step to request
  auto setup_tile = [&](const int t) __attribute__((always_inline)) {
    const int tile = xcd_remap(t, ntiles);
    const int mt = tile / n_tiles, nt = tile - mt * n_tiles;  // n fastest: the A rows are shared through L2
    ld_m0 = mt * BM; ld_n0 = nt * BN;
    const int hw = g.Hg * g.Wg;
#pragma unroll
    for (int i = 0; i < A_NI; ++i) {
      const int m = ld_m0 + a_row[i];
      const int n = m / hw, rem = m - n * hw;
      const int oi = rem / g.Wg, oj = rem - oi * g.Wg;
      const int ih0 = oi * g.ish + g.ioh, iw0 = oj * g.isw + g.iow;
      const bool mv = m < M;                                  // rows past M: every tap fails the bounds test
      a_ih0[i] = mv ? ih0 : -0x40000000;
      a_iw0[i] = mv ? iw0 : 0;
      a_boff[i] = mv ? (((n * g.Hin + ih0) * g.Win + iw0) * g.Cin) * (int)sizeof(st_t) + a_cb[i] : 0;   // BYTES of tap (0, 0) (may be negative)
    }
#pragma unroll
    for (int i = 0; i < B_NI; ++i) {
      // rows past Cout read the last row (finite values; their output columns are never stored)
      const int k = min(ld_n0 + b_row[i], g.Cout - 1);
      b_boff[i] = k * (g.WTAPS * CB * 128) + b_cb[i];
    }
    l_tr = 0; l_ts = 0; l_cb = 0;
  };
  // Requests of ONE k-step, one 1-KiB piece at a time so that mma_step can place them between its MFMAs.  K order as
  // conv_nt_kernel: channel blocks outer, taps inner.  Past the last k-step of the tile the pieces are still requested, with
  // every lane out of range (zeros into a ring slot nobody reads again, no memory traffic): the k-loop then has ONE instruction
  // stream and ONE counted wait, no tail cases.
  int q_dh = 0, q_dw = 0, q_atap = 0, q_bsoff = 0;            // (wave-uniform) constants of the k-step being requested
  unsigned q_oob = 0u;
  char *q_base = wsm;
  auto issue_begin = [&](const int slot, const bool live) __attribute__((always_inline)) {
    q_dh = l_tr * g.idh; q_dw = l_ts * g.idw;
    q_atap = ((q_dh * g.Win + q_dw) * g.Cin + l_cb * KB) * (int)sizeof(st_t);
    const int wtap = (g.wr0 + l_tr * g.wrs) * g.WS + g.ws0 + l_ts * g.wss;
    q_bsoff = (wtap * CB + l_cb) * 128;
    q_oob = live ? 0u : kOOB;
    q_base = wsm + slot * STG;
    ++l_ts;                                   // wave-uniform advance: taps inner, channel blocks outer
    const bool wrap = l_ts == g.TS;
    l_ts = wrap ? 0 : l_ts;
    l_tr += wrap ? 1 : 0;
    const bool wrap2 = l_tr == g.TR;
    l_tr = wrap2 ? 0 : l_tr;
    l_cb += wrap2 ? 1 : 0;
  };
  auto issue_piece = [&](const int i) __attribute__((always_inline)) {     // i: compile-time constant, 0 .. NI - 1 (A pieces first)
    if (dbg & 1) return;
    if (i < A_NI) {
      const int ih = a_ih0[i < A_NI ? i : 0] + q_dh, iw = a_iw0[i < A_NI ? i : 0] + q_dw;
      const bool v = (unsigned)ih < (unsigned)g.Hin && (unsigned)iw < (unsigned)g.Win;
      const unsigned off = (unsigned)(a_boff[i < A_NI ? i : 0] + q_atap) | (v ? q_oob : kOOB);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (__attribute__((address_space(3))) void *)(q_base + (wave * A_NI + i) * 1024),
                                               16, (int)off, 0, 0, 0);
    } else {
      const int j = i - A_NI;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (__attribute__((address_space(3))) void *)(q_base + BM * 128 + (wave * B_NI + j) * 1024),
                                               16, (int)((unsigned)b_boff[j < B_NI ? (j < 0 ? 0 : j) : 0] | q_oob), q_bsoff, 0, 0);
    }
  };
  auto issue = [&](const int slot, const bool live) __attribute__((always_inline)) {
    issue_begin(slot, live);
#pragma unroll
    for (int i = 0; i < NI; ++i) issue_piece(i);
  };

  // ---- fragments: row r = w? + 32 i + frow of an image sits in bank line r >> 1, chunk c = 4 p + 2 kk + (lane >> 5) of it in slot
  // ((r & 1) * 8 + c) ^ ((r >> 1) & 7); the wave's row offset is a whole number of 16 lines, so the row bits are frow's
  const int wm = (wave / WAVES_N) * (TM * 32), wn = (wave % WAVES_N) * (TN * 32);
  const int frow = lane & 31;
  const int fslot = ((((frow & 1) << 3) | (lane >> 5)) ^ ((frow >> 1) & 7)) << 4;
  const int foff = (frow >> 1) * 256 + fslot;
  f32x16 acc[TM][TN];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  // One k-step: ALL sixteen fragment reads first (64 registers: this kernel has them), in the order the MFMAs want them, then
  // the 24 MFMAs in groups between which the NI requests of the k-step D ahead are placed -- the matrix pipe works through a
  // group while the vector / scalar units form the next request's addresses.  x w = h0 g0 + h0 g1 + h1 g0, smallest terms
  // first (conv_nt_kernel's order per accumulator).
  auto mma_step = [&](const int slot) __attribute__((always_inline)) {
    const char *sa = wsm + slot * STG + wm * 128, *sb = wsm + slot * STG + BM * 128 + wn * 128;
    // fragment f of a row: chunk 2 f + (lane >> 5) of its record.  Float build: f = 2 piece + kk (pieces of 32 channels); bf16
    // build: f = kk (64 channels).  Read in the order the MFMAs want them.
    bf16x8 fa[4][TM], fb[4][TN];
    constexpr int ORD[4] = {kHalf ? 0 : 2, 0, kHalf ? 1 : 3, kHalf ? 2 : 1};     // float: (p1,kk0) (p0,kk0) (p1,kk1) (p0,kk1); bf16: kk 0 1 2 3
    constexpr int ORDH[4] = {0, 1, 2, 3};
    if (dbg & 4) {
#pragma unroll
      for (int f = 0; f < 4; ++f) {
#pragma unroll
        for (int i = 0; i < TM; ++i) { fa[f][i] = bf16x8{}; asm volatile("" : "+v"(fa[f][i])); }
#pragma unroll
        for (int j = 0; j < TN; ++j) { fb[f][j] = bf16x8{}; asm volatile("" : "+v"(fb[f][j])); }
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f = kHalf ? ORDH[q] : ORD[q];
        // float build: the first product of a block needs A piece 1 and B piece 0 (f and f ^ 2): read both sides of each
        const int fbq = kHalf ? f : (f ^ 2);
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[f][i] = *reinterpret_cast<const bf16x8 *>(sa + i * 4096 + (foff ^ ((2 * f) << 4)));
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[fbq][j] = *reinterpret_cast<const bf16x8 *>(sb + j * 4096 + (foff ^ ((2 * fbq) << 4)));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    constexpr int NMMA = NKK * NPROD * TM * TN;                      // MFMAs per k-step; request i goes behind MFMA (i + 1) NMMA / NI - 1
    static_assert(NI <= NMMA, "at most one request per MFMA");
    // float build: x w = h0 g0 + h0 g1 + h1 g0, smallest terms first (conv_nt_kernel's order per accumulator); fragment = 2 piece + kk
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
    for (int m = 0; m < NMMA; ++m) {
      const int kk = m / (NPROD * TM * TN), t3 = (m / (TM * TN)) % NPROD, i = (m / TN) % TM, j = m % TN;
      const int ia = kHalf ? kk : 2 * PA[t3] + kk, ib = kHalf ? kk : 2 * PB[t3] + kk;
      if (dbg & 2) { asm volatile("" :: "v"(fa[ia][i]), "v"(fb[ib][j])); }
      else if constexpr (kHalf)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ia][i], fb[ib][j], acc[i][j], 0, 0, 0);
      else
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[ia][i]),
                                                           __builtin_bit_cast(f16x8, fb[ib][j]), acc[i][j], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NI; ++q)
        if (((q + 1) * NMMA) / NI - 1 == m) {
          issue_piece(q);
          __builtin_amdgcn_sched_barrier(0);
        }
    }
  };

  float gmx_all = 0.f;       // EPI == 2: largest |dx| stored by this thread over all its tiles (g.bn_dy_absmax)

  if constexpr (XT) {
    // One ring for the whole walk: k-step s of the flattened (tile, k-step) sequence sits in slot s % STAGES and is requested D
    // k-steps ahead, across tile boundaries (the host routes nk >= STAGES: a tile's first D requests all belong to one tile).
    // The exchange of the direct epilogue's per-column partials lives behind the ring where that fits beside a second
    // workgroup (four waves), and IN the slot of the tile's last k-step -- the one slot no request is aimed at while the
    // epilogue runs -- where one workgroup owns the CU (eight waves, 144 KiB of ring): that costs one barrier per tile.
    constexpr bool EXCH_IN_RING = NWV == 8 || BM == 256;      // (256 x 64 on four waves: two workgroups of exactly 80 KiB)
    int t = blockIdx.x;
    if (t < ntiles) {
      setup_tile(t);
#pragma unroll
      for (int j = 0; j < D; ++j) issue(j, true);
    }
    int slot = 0, islot = D % STAGES;
    bool more = false;
    DSPN_STAMP_DECL;
    for (; t < ntiles; t += gridDim.x) {
      const int m0 = ld_m0, n0 = ld_n0;            // (the loaders moved on to this tile during the previous tile's last D k-steps)
      const int tn = t + (int)gridDim.x;
      const bool first = t == (int)blockIdx.x;
      zero_acc();
      int last_slot = 0;
      for (int kt = 0; kt < nk; ++kt) {
        // the images of this k-step have landed (this wave's pieces: the wait; the other waves': the barrier).  A tile's first D
        // k-steps were requested BEFORE the previous tile's epilogue, whose 64 stores per wave (always issued: out-of-range
        // lanes are dropped by the hardware) are younger, as are the D - 1 requests behind them: "at most 63 younger
        // operations outstanding" says they have landed without asking for those stores to be acknowledged
        if (!first && kt < D) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * NI) : "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + D < nk) {
          issue_begin(islot, true);
        } else {                         // (wave-uniform) the request belongs to the NEXT tile: k-step kt + D - nk of it
          if (kt + D == nk) { more = tn < ntiles; if (more) setup_tile(tn); }
          issue_begin(islot, more);
        }
        mma_step(slot);
        last_slot = slot;
        slot = slot + 1 == STAGES ? 0 : slot + 1;
        islot = islot + 1 == STAGES ? 0 : islot + 1;
      }
      DSPN_STAMP(2);
      char *exch = wsm + STAGES * STG;
      if constexpr (EXCH_IN_RING) {
        exch = wsm + last_slot * STG;
        if constexpr (EPI != 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }   // every wave has read its last fragments
      }
      if constexpr (!kHalf) direct_epilogue<WAVES_M, WAVES_N, EPI, (EPI == 2 ? 4 : 8), SR>(g, exch, acc, inv_a, inv_b, m0, n0, M, tid, wave, bias, out, residual, gmx_all);
      DSPN_STAMP(5);
    }
    DSPN_STAMP_FLUSH;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the last, out-of-range requests)
    if constexpr (EPI != 1) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                      // (the exchange of the last tile has been read)
      wide_publish_absmax<NWV>(g, wsm, gmx_all, tid);
    }
    return;
  }

  DSPN_STAMP_DECL;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    setup_tile(t);
    const int m0 = ld_m0, n0 = ld_n0;
#pragma unroll
    for (int j = 0; j < D; ++j) issue(j, j < nk);
    zero_acc();
    DSPN_STAMP(1);
    int slot = 0, islot = D % STAGES;
    for (int kt = 0; kt < nk; ++kt) {
      // the images of k-step kt have landed (this wave's pieces: the counted wait -- D - 1 k-steps stay in flight; the other
      // waves': the barrier), and every wave has read the fragments of k-step kt - 1, whose ring slot is requested into next
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * NI) : "memory");
      if (!(dbg & 8)) __builtin_amdgcn_s_barrier();
      issue_begin(islot, kt + D < nk);
      mma_step(slot);
      slot = slot + 1 == STAGES ? 0 : slot + 1;
      islot = islot + 1 == STAGES ? 0 : islot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the out-of-range requests past K: nothing may land in the staging area)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();     // every wave has read its last fragments (nothing is in flight): the LDS becomes the staging area
    DSPN_STAMP(2);
    if (dbg & 16) { if (acc[0][0][0] == 1.2345e33f) out[0] = (st_t)acc[TM - 1][TN - 1][5]; continue; }

    // ---- epilogue.  C/D layout: col = lane & 31 (cout), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    {
      constexpr int SLD = BN + 4;
      float *st = reinterpret_cast<float *>(wsm);
  #pragma unroll
      for (int i = 0; i < TM; ++i)
  #pragma unroll
        for (int j = 0; j < TN; ++j)
  #pragma unroll
          for (int r = 0; r < 16; ++r)
            st[(wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * SLD + wn + j * 32 + (lane & 31)] = acc[i][j][r] * inv_a * inv_b;
    }
    DSPN_STAMP(3);
    wide_epilogue<BM, BN, NTHR, EPI, SR>(g, wsm, m0, n0, M, tid, bias, out, residual, gmx_all, NoStage() DSPN_STAMP_ARG);
  }
  DSPN_STAMP_FLUSH;
  if constexpr (EPI != 1) wide_publish_absmax<NWV>(g, wsm, gmx_all, tid);
}

#ifndef DSPN_HALF
// ---- the same tiles for an A operand that is NOT a pure copy (round 5, second family member): a float tensor, optionally with
// the BatchNorm affine (+ ReLU) of the layer in front folded into the loader (conv_nt_kernel's INTF), cut into its two fp16
// pieces on the way into the SAME swizzled LDS image.  The A rows go global -> registers -> (affine, cut) -> LDS as in
// conv_nt_kernel; what the family changes for these layers is the rest: the weight planes still go global -> LDS directly, a
// wave owns 64 x 64 outputs (16 fragment reads per 24 MFMAs instead of 12 per 12), and the BatchNorm epilogues are the wide
// family's (statistics at no measurable cost where conv_nt_kernel's 128-register epilogue pays 15 - 30 %).
// A thread moves `A_U` units of (row, 8 channels): two 16-byte loads, two 16-byte LDS stores (one per piece).  Two LDS stages:
// the requests of k-step kt + 1 are issued before the MFMAs of k-step kt, the pieces are formed between its two MFMA halves.
typedef float ntv_f32x4_t __attribute__((ext_vector_type(4)));
// "at most NEWER requests are still outstanding", tied to the registers it guards (see wait_set in conv_ntv_kernel)
template <int NEWER>
__device__ __forceinline__ void ntv_wait2(ntv_f32x4_t &a, ntv_f32x4_t &b) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(NEWER));
}
template <int NEWER>
__device__ __forceinline__ void ntv_wait4(ntv_f32x4_t &a, ntv_f32x4_t &b, ntv_f32x4_t &c, ntv_f32x4_t &d) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(NEWER));
}
// XT: the tile-spanning loop (see conv_ntw_kernel): the weight pieces and the A rows of the NEXT tile's first two k-steps are
// requested before the epilogue of the current one, which stages chunk by chunk behind ring slot 0.
template <int WAVES_M, int WAVES_N, bool INTF, int EPI, int SR = 128, int XT = 0>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, 2) void conv_ntv_kernel(
    const float *__restrict__ in, const float *__restrict__ wgt, const float *__restrict__ bias,
    float *__restrict__ out, const ConvGeom g, const int m_tiles, const int n_tiles,
    const float *__restrict__ residual) {
  constexpr int TM = 2, TN = 2;
  constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
  constexpr int NWV = WAVES_M * WAVES_N, NTHR = NWV * 64;
  constexpr int A_U = BM * 4 / NTHR;                     // (row, 8-channel) units per thread and k-step
  constexpr int B_NI = BN / (8 * NWV);                   // 1-KiB pieces of the weight image per wave and k-step
  static_assert((BM * 4) % NTHR == 0 && BN % (8 * NWV) == 0, "whole units / pieces per thread / wave");
  constexpr int STG = (BM + BN) * 128;
  // rows in flight: with at most two units per thread the rows of k-step kt + 2 are requested while k-step kt is multiplied (a
  // second register set) -- the 1 x 1 layers of the residual stream are HBM-bound and a workgroup with ONE 16-KiB request in
  // flight leaves the memory latency (about one k-step) exposed: 3.4 - 3.9 TB/s before, see DESIGN.md
#ifdef DSPN_NTV_SHALLOW
  constexpr bool DEEP = false;
#else
  constexpr bool DEEP = A_U <= 2;
#endif
  constexpr int NSET = DEEP ? 2 : 1;
  constexpr int NA = A_U * 2 + (INTF ? 4 : 0);           // vector loads of one request of A rows
  extern __shared__ __attribute__((aligned(1024))) char wsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = m_tiles * n_tiles;
  const int M = g.N * g.Hg * g.Wg;
  const int CB = g.Cin >> 5;
  const int nk = g.TR * g.TS * CB;
  const float sc_a = operand_scale(g.a_absmax), sc_b = operand_scale(g.b_absmax);
  const float inv_a = 1.f / sc_a, inv_b = 1.f / sc_b;
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wgt), 0, g.w_bytes, 0x00020000);
  constexpr unsigned kOOB = 0x80000000u;
  const bool in_relu = g.flags & 32;
  // The A rows (and their affine) are requested by asm statements, not builtins: hipcc sinks a builtin load whose value is
  // first used behind the NEXT barrier down to that use (the rows of k-step kt + 2 would be requested one k-step late, and
  // behind the weight requests the counted waits rely on being older).  The compiler does not know these registers are
  // pending, so every read of a set is preceded by wait_set(), an s_waitcnt that names the set's registers.
  auto rsrc_words = [](const float *p, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    return u32x4_t{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a),
                   (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu)),
                   (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
  };
  const u32x4_t rw_a = rsrc_words(in, g.in_bytes);
  const u32x4_t rw_sc = rsrc_words(g.in_scale, INTF ? (unsigned)g.Cin * 4u : 0u), rw_sh = rsrc_words(g.in_shift, INTF ? (unsigned)g.Cin * 4u : 0u);
  typedef ntv_f32x4_t f32x4_t;
  auto load16 = [](const u32x4_t rw, const unsigned voff) __attribute__((always_inline)) {
    f32x4_t r;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r) : "v"(voff), "s"(rw));
    return r;
  };

  // ---- A units of this thread: unit u = tid + j NTHR -> row u >> 2, channel group q = tid & 3 (the same for all its units)
  const int aq = tid & 3;
  int a_lds[A_U];                                        // byte offset of the unit's piece-0 chunk inside the A image
#pragma unroll
  for (int j = 0; j < A_U; ++j) {
    const int r = (tid + j * NTHR) >> 2, line = r >> 1;
    a_lds[j] = line * 256 + (((((r & 1) << 3) | aq) ^ (line & 7)) << 4);      // piece 1: chunk aq + 4 -> this ^ 64
  }
  int b_row[B_NI], b_cb[B_NI];
#pragma unroll
  for (int i = 0; i < B_NI; ++i) {
    const int line = 4 * (wave * B_NI + i) + (lane >> 4), sl = (lane & 15) ^ (line & 7);
    b_row[i] = 2 * line + (sl >> 3); b_cb[i] = (sl & 7) * 16;
  }
  int a_ih0[A_U], a_iw0[A_U], a_boff[A_U], b_boff[B_NI];
  int l_tr = 0, l_ts = 0, l_cb = 0;                      // (tap row, tap column, channel block) of the next A request
  int lb_tr = 0, lb_ts = 0, lb_cb = 0;                   // ... and of the next weight request
  f32x4_t ra[NSET][A_U][2];                              // the rows requested last (per register set)
  f32x4_t tsc[NSET][2], tsh[NSET][2];                    // affine of this thread's 8 channels (INTF)
  unsigned a_mask[NSET] = {};                            // bit j: unit j of the set's k-step lies inside the image
  u32x4_t pa[A_U][2];                                    // its pieces, kept until the image being read has been released

  const int wm = (wave / WAVES_N) * (TM * 32), wn = (wave % WAVES_N) * (TN * 32);
  const int frow = lane & 31;
  const int fslot = ((((frow & 1) << 3) | (lane >> 5)) ^ ((frow >> 1) & 7)) << 4;
  const int foff = (frow >> 1) * 256 + fslot;
  f32x16 acc[TM][TN];
  float gmx_all = 0.f;

  int m0 = 0, n0 = 0;                                    // the tile being multiplied / written
  int nm0 = 0, nn0 = 0;                                  // the tile the loaders are set up for
  auto setup_tile = [&](const int t) __attribute__((always_inline)) {
    const int tile = xcd_remap(t, ntiles);
    const int mt = tile / n_tiles, nt = tile - mt * n_tiles;
    nm0 = mt * BM; nn0 = nt * BN;
    {
      const int hw = g.Hg * g.Wg;
#pragma unroll
      for (int j = 0; j < A_U; ++j) {
        const int m = nm0 + ((tid + j * NTHR) >> 2);
        const int n = m / hw, rem = m - n * hw;
        const int oi = rem / g.Wg, oj = rem - oi * g.Wg;
        const int ih0 = oi * g.ish + g.ioh, iw0 = oj * g.isw + g.iow;
        const bool mv = m < M;
        a_ih0[j] = mv ? ih0 : -0x40000000;
        a_iw0[j] = mv ? iw0 : 0;
        a_boff[j] = mv ? (((n * g.Hin + ih0) * g.Win + iw0) * g.Cin + aq * 8) * 4 : 0;
      }
#pragma unroll
      for (int i = 0; i < B_NI; ++i) {
        const int k = min(nn0 + b_row[i], g.Cout - 1);
        b_boff[i] = k * (g.WTAPS * CB * 128) + b_cb[i];
      }
      l_tr = 0; l_ts = 0; l_cb = 0;
      lb_tr = 0; lb_ts = 0; lb_cb = 0;
    }
  };
  {
    // requests: the A rows of a k-step into register set SET, the weight pieces of a k-step into ring slot `slot`; past the last
    // k-step every lane is out of range (zeros, no traffic).  Each kind walks the (tap, channel block) sequence with its own counters.
    auto request_a = [&](auto set_c, const bool live) __attribute__((always_inline)) {
      constexpr int SET = decltype(set_c)::value;
      const int dh = l_tr * g.idh, dw = l_ts * g.idw;
      const int a_tap = ((dh * g.Win + dw) * g.Cin + l_cb * 32) * 4;
      const unsigned oob = live ? 0u : kOOB;
      if constexpr (INTF) {
        const unsigned coff = (unsigned)(l_cb * 32 + aq * 8) * 4u | oob;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          tsc[SET][h] = load16(rw_sc, coff + 16u * h);
          tsh[SET][h] = load16(rw_sh, coff + 16u * h);
        }
      }
      a_mask[SET] = 0;
#pragma unroll
      for (int j = 0; j < A_U; ++j) {
        const int ih = a_ih0[j] + dh, iw = a_iw0[j] + dw;
        const bool v = live && (unsigned)ih < (unsigned)g.Hin && (unsigned)iw < (unsigned)g.Win;
        a_mask[SET] |= v ? (1u << j) : 0u;
        const unsigned off = (unsigned)(a_boff[j] + a_tap) | (v ? 0u : kOOB);
#pragma unroll
        for (int h = 0; h < 2; ++h) ra[SET][j][h] = load16(rw_a, off + 16u * h);
      }
      ++l_ts;
      const bool wrap = l_ts == g.TS;
      l_ts = wrap ? 0 : l_ts;
      l_tr += wrap ? 1 : 0;
      const bool wrap2 = l_tr == g.TR;
      l_tr = wrap2 ? 0 : l_tr;
      l_cb += wrap2 ? 1 : 0;
    };
    auto request_b = [&](const int slot, const bool live) __attribute__((always_inline)) {
      const int wtap = (g.wr0 + lb_tr * g.wrs) * g.WS + g.ws0 + lb_ts * g.wss;
      const int b_soff = (wtap * CB + lb_cb) * 128;
      const unsigned oob = live ? 0u : kOOB;
      char *base = wsm + slot * STG + BM * 128;
#pragma unroll
      for (int i = 0; i < B_NI; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (__attribute__((address_space(3))) void *)(base + (wave * B_NI + i) * 1024),
                                                 16, (int)((unsigned)b_boff[i] | oob), b_soff, 0, 0);
      ++lb_ts;
      const bool wrap = lb_ts == g.TS;
      lb_ts = wrap ? 0 : lb_ts;
      lb_tr += wrap ? 1 : 0;
      const bool wrap2 = lb_tr == g.TR;
      lb_tr = wrap2 ? 0 : lb_tr;
      lb_cb += wrap2 ? 1 : 0;
    };
    // affine (+ ReLU, zero outside the image: the padding is applied AFTER the affine, as in the forward of the layer in front) and
    // the cut of the rows requested last
    // the rows (and affine) of register set SET have arrived once at most NEWER requests are outstanding (requests complete in
    // order); the statement names the set's registers so that no read of them is scheduled above it
    auto wait_set = [&](auto set_c, auto newer_c) __attribute__((always_inline)) {
      constexpr int SET = decltype(set_c)::value, NEWER = decltype(newer_c)::value;
      if constexpr (INTF) ntv_wait4<NEWER>(tsc[SET][0], tsc[SET][1], tsh[SET][0], tsh[SET][1]);
#pragma unroll
      for (int j = 0; j < A_U; ++j) ntv_wait2<NEWER>(ra[SET][j][0], ra[SET][j][1]);
    };
    auto cut = [&](auto set_c) __attribute__((always_inline)) {
      constexpr int SET = decltype(set_c)::value;
#pragma unroll
      for (int j = 0; j < A_U; ++j) {
        bf16x4 p0[2], p1[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float4 u = make_float4(ra[SET][j][h][0], ra[SET][j][h][1], ra[SET][j][h][2], ra[SET][j][h][3]);
          if constexpr (INTF) {
            // fmaf, like every other place that evaluates this affine: the ReLU mask must come out identical in forward and backward
            u = make_float4(fmaf(u.x, tsc[SET][h][0], tsh[SET][h][0]), fmaf(u.y, tsc[SET][h][1], tsh[SET][h][1]),
                            fmaf(u.z, tsc[SET][h][2], tsh[SET][h][2]), fmaf(u.w, tsc[SET][h][3], tsh[SET][h][3]));
            if (in_relu) u = make_float4(fmaxf(u.x, 0.f), fmaxf(u.y, 0.f), fmaxf(u.z, 0.f), fmaxf(u.w, 0.f));
            const bool v = (a_mask[SET] >> j) & 1u;
            u = make_float4(v ? u.x : 0.f, v ? u.y : 0.f, v ? u.z : 0.f, v ? u.w : 0.f);
          }
          split2h(u, sc_a, p0[h], p1[h]);
        }
        const dspn::u32x2_t a0 = __builtin_bit_cast(dspn::u32x2_t, p0[0]), a1 = __builtin_bit_cast(dspn::u32x2_t, p0[1]);
        const dspn::u32x2_t c0 = __builtin_bit_cast(dspn::u32x2_t, p1[0]), c1 = __builtin_bit_cast(dspn::u32x2_t, p1[1]);
        pa[j][0] = u32x4_t{a0[0], a0[1], a1[0], a1[1]};
        pa[j][1] = u32x4_t{c0[0], c0[1], c1[0], c1[1]};
      }
    };
    auto store_a = [&](const int slot) __attribute__((always_inline)) {
      char *base = wsm + slot * STG;
#pragma unroll
      for (int j = 0; j < A_U; ++j) {
        *reinterpret_cast<u32x4_t *>(base + a_lds[j]) = pa[j][0];
        *reinterpret_cast<u32x4_t *>(base + (a_lds[j] ^ 64)) = pa[j][1];
      }
    };
    typedef std::integral_constant<int, 0> set0_t;
    typedef std::integral_constant<int, 1> set1_t;
    // the first requests of a tile (setup_tile has run): the weight pieces of k-step 0, the rows of k-steps 0 and 1
    auto tile_requests_a = [&](const bool live) __attribute__((always_inline)) {
      request_a(set0_t{}, live);
      if constexpr (DEEP) request_a(set1_t{}, live && 1 < nk);
    };
    // ... and what turns them into the images of k-step 0 (weights, rows 0, rows 1 are requested in this order)
    auto tile_first_requests = [&](const bool live) __attribute__((always_inline)) {
      request_b(0, live);
      __builtin_amdgcn_sched_barrier(0);      // the weight requests FIRST: the counted waits let the NA newest requests fly
      tile_requests_a(live);
    };
    // pre (XT, every tile but a workgroup's first): the requests were issued before the previous tile's epilogue, whose 64 stores
    // per wave (always issued) are younger than all of them: "at most 63 younger operations outstanding" says the rows of k-step
    // 0 -- and the weight pieces, which are older -- have landed without asking for those stores to be acknowledged
    auto tile_head = [&](const bool pre) __attribute__((always_inline)) {
      if (!pre) tile_first_requests(true);
      if (pre) wait_set(set0_t{}, std::integral_constant<int, 63>{});
      else wait_set(set0_t{}, std::integral_constant<int, DEEP ? NA : 0>{});
      cut(set0_t{});
      store_a(0);
      // the weight pieces of k-step 0 have landed and the A image is written; the rows of k-step 1 (DEEP) may still be on their way
      if (pre) asm volatile("s_waitcnt vmcnt(63) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(DEEP ? NA : 0) : "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    // one k-step; PAR = kt & 1 picks the register sets at compile time: the rows of k-step kt + 1 are cut out of set CS, the rows
    // of k-step kt + 2 (DEEP; kt + 1 otherwise) are requested into set RS
    auto kstep = [&](auto par_c, const int kt) __attribute__((always_inline)) {
      constexpr int PAR = decltype(par_c)::value;
      typedef std::integral_constant<int, DEEP ? PAR : 0> rs_t;
      typedef std::integral_constant<int, DEEP ? (PAR ^ 1) : 0> cs_t;
      const int cur = PAR;
      request_b(cur ^ 1, kt + 1 < nk);      // (every wave has read ring slot cur ^ 1 before the barrier it has just passed)
      // the weight requests are issued FIRST and stay first: the wait at the end of the k-step counts on the NA newest
      // requests being the rows (hipcc otherwise hoists the row loads above the LDS-DMA instructions)
      __builtin_amdgcn_sched_barrier(0);
      request_a(rs_t{}, kt + (DEEP ? 2 : 1) < nk);
      __builtin_amdgcn_sched_barrier(0);
      const char *sa = wsm + cur * STG + wm * 128, *sb = wsm + cur * STG + BM * 128 + wn * 128;
      bf16x8 fa[4][TM], fb[4][TN];          // fragment f = 2 piece + kk: chunk 2 f + (lane >> 5) of the row's record
      constexpr int ORD[4] = {2, 0, 3, 1};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f = ORD[q], fbq = f ^ 2;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[f][i] = *reinterpret_cast<const bf16x8 *>(sa + i * 4096 + (foff ^ ((2 * f) << 4)));
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[fbq][j] = *reinterpret_cast<const bf16x8 *>(sb + j * 4096 + (foff ^ ((2 * fbq) << 4)));
      }
      __builtin_amdgcn_sched_barrier(0);
      constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        if (kk == 1) {                      // the rows to cut have had (at least) the first half's MFMAs to arrive
          __builtin_amdgcn_sched_barrier(0);
          wait_set(cs_t{}, std::integral_constant<int, DEEP ? B_NI + NA : 0>{});
          cut(cs_t{});
        }
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[2 * PA[t3] + kk][i]),
                                                                 __builtin_bit_cast(f16x8, fb[2 * PB[t3] + kk][j]), acc[i][j], 0, 0, 0);
      }
      // pin the pieces here: their only readers (the LDS stores) sit behind the MFMAs, and hipcc otherwise sinks the whole piece
      // arithmetic down there, next to them; then spread it over the second half's MFMAs
#pragma unroll
      for (int j = 0; j < A_U; ++j) { asm volatile("" : "+v"(pa[j][0])); asm volatile("" : "+v"(pa[j][1])); }
#pragma unroll
      for (int m = 0; m < 3 * TM * TN; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, (INTF ? 14 : 8) * A_U / 2 + 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      store_a(cur ^ 1);
      // the weight pieces requested at the top have landed (they were issued BEFORE the rows, which may stay in flight)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(DEEP ? NA : 0) : "memory");
      __builtin_amdgcn_s_barrier();
    };
    auto kloop = [&]() __attribute__((always_inline)) {
      for (int kt = 0; kt < nk; kt += 2) {
        kstep(set0_t{}, kt);
        if (kt + 1 < nk) kstep(set1_t{}, kt + 1);
      }
      // the last requests (past the last k-step: out of range, zeros) still name their registers: nothing else may be given
      // those registers before they have landed -- the compiler does not know they are pending
      wait_set(set0_t{}, std::integral_constant<int, 0>{});
      if constexpr (DEEP) wait_set(set1_t{}, std::integral_constant<int, 0>{});
    };
    // the whole tile staged in LDS (the ring is done with: every k-step ends with a barrier behind its last fragment read)
    auto stage_tile = [&]() __attribute__((always_inline)) {
      constexpr int SLD = BN + 4;
      float *st = reinterpret_cast<float *>(wsm);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            st[(wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * SLD + wn + j * 32 + (lane & 31)] = acc[i][j][r] * inv_a * inv_b;
    };
    if constexpr (XT == 1) {
      // the weight pieces of the NEXT tile's first k-step and the A rows of its first two are requested before the epilogue of
      // the current tile, which works from the accumulators (direct_epilogue: nothing of it touches the ring).  (DEEP: two
      // register sets; the host routes an even nk >= 2 and a dense output here.)
      static_assert(DEEP, "two register sets");
      int t = blockIdx.x;
      if (t < ntiles) setup_tile(t);
      bool pre = false;
      DSPN_STAMP_DECL;
      for (; t < ntiles; t += gridDim.x) {
        m0 = nm0; n0 = nn0;
        tile_head(pre);
        DSPN_STAMP(1);
        kloop();
        DSPN_STAMP(2);
        const int tn = t + (int)gridDim.x;
        if (tn < ntiles) setup_tile(tn);
        tile_first_requests(tn < ntiles);
        __builtin_amdgcn_sched_barrier(0);
        DSPN_STAMP(3);
        direct_epilogue<WAVES_M, WAVES_N, EPI, ((EPI == 2 || INTF) ? 2 : 4)>(g, wsm + 2 * STG, acc, inv_a, inv_b, m0, n0, M, tid, wave, bias, out, residual, gmx_all);
        DSPN_STAMP(5);
        pre = true;
      }
      DSPN_STAMP_FLUSH;
      wait_set(set0_t{}, std::integral_constant<int, 0>{});
      wait_set(set1_t{}, std::integral_constant<int, 0>{});
    } else {
      // XT == 2 (round 6): the round-5 loop with the DIRECT epilogue -- no staging, no publishing barrier, the table partials in the
      // lanes -- and nothing held in registers across it, so its operand rows run 8 deep like the plane-fed kernel's
      DSPN_STAMP_DECL;
      for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        setup_tile(t);
        m0 = nm0; n0 = nn0;
        tile_head(false);
        DSPN_STAMP(1);
        kloop();
        DSPN_STAMP(2);
        if constexpr (XT == 2) {
          // the exchange of the per-column partials lies in ring slot 1: nothing is in flight after the k-loop, and the next writes
          // into that slot (k-step 0's weight request, its closing A store) sit behind the barrier of the next tile's head, which
          // a wave passes only after its part of the merge
          direct_epilogue<WAVES_M, WAVES_N, EPI, (EPI == 2 ? 4 : 8), SR>(g, wsm + STG, acc, inv_a, inv_b, m0, n0, M, tid, wave, bias, out, residual, gmx_all);
          DSPN_STAMP(5);
        } else {
          stage_tile();
          DSPN_STAMP(3);
          wide_epilogue<BM, BN, NTHR, EPI, SR>(g, wsm, m0, n0, M, tid, bias, out, residual, gmx_all, NoStage() DSPN_STAMP_ARG);
        }
      }
      DSPN_STAMP_FLUSH;
    }
    if constexpr (EPI != 1) wide_publish_absmax<NWV>(g, wsm, gmx_all, tid);
  }
}

template <int WAVES_M, int WAVES_N, bool INTF, int EPI, int SR, int XT = 0>
int launch_ntv_impl(const float *in, const float *w, const float *bias, float *out, const ConvGeom &g, hipStream_t s,
                    const float *residual) {
  constexpr int BM = WAVES_M * 64, BN = WAVES_N * 64;
  const long long M = (long long)g.N * g.Hg * g.Wg;
  const int mt = (int)((M + BM - 1) / BM), nt = (g.Cout + BN - 1) / BN;
  // XT: the ring and, behind it, the exchange of the direct epilogue's per-column partials (4 x BN x 6 floats)
  const size_t lds = XT == 1 ? (size_t)2 * (BM + BN) * 128 + sizeof(float) * 4 * BN * 6
                     : XT == 2 ? (size_t)2 * (BM + BN) * 128         // (the exchange lies inside ring slot 1)
                               : std::max<size_t>((size_t)2 * (BM + BN) * 128, sizeof(float) * BM * (BN + 4));
  static_assert(XT != 2 || (size_t)(BM + BN) * 128 >= sizeof(float) * 2 * WAVES_M * BN * 6, "the exchange fits a ring slot");
  auto kern = conv_ntv_kernel<WAVES_M, WAVES_N, INTF, EPI, SR, XT>;
  static dspn::KernelDeviceState st;
  const bool first = !st.slots[0] && !st.slots[1];
  const int dev = dspn::ensure_persistent_grid(reinterpret_cast<const void *>(kern), WAVES_M * WAVES_N * 64, lds, st, "conv_ntv");
  if (dev < 0) return dev;
  const int slots = st.slots[dev], slots_per_cu = st.slots_per_cu[dev], slots_cus = st.cus[dev];
  if (first && getenv("DSPN_DEBUG_PRINT"))
    fprintf(stderr, "[dspn] conv_ntv<%d,%d,intf=%d,epi=%d,xt=%d>: %zu B LDS, occupancy %d/CU x %d CUs -> grid %d\n", WAVES_M, WAVES_N,
            (int)INTF, EPI, (int)XT, lds, slots_per_cu, slots_cus, slots);
  const int reserved = dspn::reserved_cus();
  const int avail = reserved > 0 ? std::max(8, slots_per_cu * std::max(8, slots_cus - reserved) / 8 * 8) : slots;
  const int grid_x = (int)std::min<long long>((long long)mt * nt, avail);
  {
    dspn::ProfScope prof(0, s);
    hipLaunchKernelGGL(kern, dim3(grid_x), dim3(WAVES_M * WAVES_N * 64), lds, s, in, w, bias, out, g, mt, nt, residual);
  }
  return dspn::check_launch("conv_ntv");
}
template <int WAVES_M, int WAVES_N, int SR = 128>
int launch_ntv(const float *in, const float *w, const float *bias, float *out, const ConvGeom &g, hipStream_t s, const float *residual) {
#define DSPN_NTV_(T, X) \
  (g.bn_sums ? launch_ntv_impl<WAVES_M, WAVES_N, T, 2, SR, X>(in, w, bias, out, g, s, residual) \
   : g.stats ? launch_ntv_impl<WAVES_M, WAVES_N, T, 1, SR, X>(in, w, bias, out, g, s, residual) \
             : launch_ntv_impl<WAVES_M, WAVES_N, T, 0, SR, X>(in, w, bias, out, g, s, residual))
  if constexpr (WAVES_M == 2 && WAVES_N == 2 && SR == 128) {
    const int nk = g.TR * g.TS * (g.Cin / 32);
    // Measured (scratch/r06/xt_bench.hip, profiles/r06_xt_*): the float-operand member gains 5 - 15 % in isolation where the
    // epilogue is a large share of the tile (K <= 256) and a workgroup walks several tiles, and LOSES inside the training step:
    // its hot calls add a residual (the conv3 of every unit), and with the next tile's rows held in registers across the
    // epilogue there is room for 2 - 4 residual rows in flight per lane where the staged epilogue keeps 16 (281 -> 305 us on
    // the stage-1 conv3 layers).  Setting 2 of dspn_conv_set_tile_spanning routes it (experiments); the default does not.
    const long long tiles = (((long long)g.N * g.Hg * g.Wg + 127) / 128) * ((g.Cout + 127) / 128);
    if (dspn::tile_spanning() >= 2 && nk >= 2 && nk <= 8 && nk % 2 == 0 && tiles >= 2048 && xt_output_ok(g))
      return g.in_scale ? DSPN_NTV_(true, 1) : DSPN_NTV_(false, 1);
    // The direct epilogue ALONE (XT = 2: the round-5 loop, nothing held across the epilogue): measured on the step, three
    // alternating runs on one box: 948.4 -> 953.9 images/s, conv family 27.04 -> 26.66 ms.  Default at setting >= 1;
    // DSPN_NTV_DIRECT=0 keeps the staged epilogue (A/B runs).
    static const bool direct_only = [] { const char *e = getenv("DSPN_NTV_DIRECT"); return !e || atoi(e) != 0; }();
    if (direct_only && xt_enabled() && nk >= 2 && xt_output_ok(g))
      return g.in_scale ? DSPN_NTV_(true, 2) : DSPN_NTV_(false, 2);
  }
  if constexpr (WAVES_M == 2 && WAVES_N == 4 && SR == 128) {
    // ... and on the eight-wave 128 x 256 member: 953.9 -> 955.6 images/s (+0.1 ... +0.3 % in each of three alternating pairs);
    // DSPN_NTV_DIRECT8=0 keeps the staged epilogue
    static const bool direct8 = [] { const char *e = getenv("DSPN_NTV_DIRECT8"); return !e || atoi(e) != 0; }();
    const int nk = g.TR * g.TS * (g.Cin / 32);
    if (direct8 && xt_enabled() && nk >= 2 && xt_output_ok(g))
      return g.in_scale ? DSPN_NTV_(true, 2) : DSPN_NTV_(false, 2);
  }
  if constexpr (WAVES_M == 4 && WAVES_N == 1 && SR == 64) {      // the 256 x 64 member (xt_c64_enabled)
    const int nk = g.TR * g.TS * (g.Cin / 32);
    if (xt_c64_enabled() && xt_enabled() && nk >= 2 && xt_output_ok(g, 256))
      return g.in_scale ? DSPN_NTV_(true, 2) : DSPN_NTV_(false, 2);
  }
  return g.in_scale ? DSPN_NTV_(true, 0) : DSPN_NTV_(false, 0);
#undef DSPN_NTV_
}
#endif   // !DSPN_HALF

// host side: one launch of the wide family.  Persistent grid as conv_nt_kernel's (occupancy x CUs, a multiple of 8).
template <int WAVES_M, int WAVES_N, int STAGES, int EPI, int SR, bool XT = false>
int launch_ntw_impl(const st_t *in, const st_t *w, const float *bias, st_t *out, const ConvGeom &g, hipStream_t s,
                    const st_t *residual) {
  constexpr int BM = WAVES_M * 64, BN = WAVES_N * 64;
  const long long M = (long long)g.N * g.Hg * g.Wg;
  const int mt = (int)((M + BM - 1) / BM), nt = (g.Cout + BN - 1) / BN;
  // XT: the ring and -- four waves -- behind it the exchange of the direct epilogue's per-column partials (4 x BN x 6 floats;
  // eight waves keep it in the ring slot of the tile's last k-step)
  const size_t lds = XT ? (size_t)STAGES * (BM + BN) * 128 + ((WAVES_M * WAVES_N == 8 || BM == 256) ? 0 : sizeof(float) * 4 * BN * 6)
                        : std::max<size_t>((size_t)STAGES * (BM + BN) * 128, sizeof(float) * BM * (BN + 4));
  auto kern = conv_ntw_kernel<WAVES_M, WAVES_N, STAGES, EPI, SR, XT>;
  static dspn::KernelDeviceState st;
  const bool first = !st.slots[0] && !st.slots[1];
  const int dev = dspn::ensure_persistent_grid(reinterpret_cast<const void *>(kern), WAVES_M * WAVES_N * 64, lds, st, "conv_ntw");
  if (dev < 0) return dev;
  const int slots = st.slots[dev], slots_per_cu = st.slots_per_cu[dev], slots_cus = st.cus[dev];
  if (first && getenv("DSPN_DEBUG_PRINT"))
    fprintf(stderr, "[dspn] conv_ntw<%d,%d,stages=%d,epi=%d,xt=%d>: %zu B LDS, occupancy %d/CU x %d CUs -> grid %d\n", WAVES_M, WAVES_N,
            STAGES, EPI, (int)XT, lds, slots_per_cu, slots_cus, slots);
  const int reserved = dspn::reserved_cus();
  const int avail = reserved > 0 ? std::max(8, slots_per_cu * std::max(8, slots_cus - reserved) / 8 * 8) : slots;
  const int grid_x = (int)std::min<long long>((long long)mt * nt, avail);
  {
    dspn::ProfScope prof(0, s);
    hipLaunchKernelGGL(kern, dim3(grid_x), dim3(WAVES_M * WAVES_N * 64), lds, s, in, w, bias, out, g, mt, nt, residual);
  }
  return dspn::check_launch("conv_ntw");
}

template <int WAVES_M, int WAVES_N, int STAGES, int SR = 128>
int launch_ntw(const st_t *in, const st_t *w, const float *bias, st_t *out, const ConvGeom &g, hipStream_t s, const st_t *residual) {
#ifndef DSPN_HALF
  if constexpr ((WAVES_M == 2 && SR == 128) || (WAVES_M == 4 && WAVES_N == 1 && SR == 64)) {
    // the four-wave 128 x 128 tile, the eight-wave 128 x 256 tile (xt_wide8_enabled) and the 256 x 64 tile of the <= 64-column
    // layers (xt_c64_enabled) on layers of at least STAGES k-steps
    const int nk = g.TR * g.TS * (g.Cin / 32);
    if (xt_enabled() && nk >= STAGES && xt_output_ok(g, WAVES_M * 64) &&
        (WAVES_M == 4 ? xt_c64_enabled() : (WAVES_N == 2 || xt_wide8_enabled()))) {
      if (g.bn_sums) return launch_ntw_impl<WAVES_M, WAVES_N, STAGES, 2, SR, true>(in, w, bias, out, g, s, residual);
      if (g.stats) return launch_ntw_impl<WAVES_M, WAVES_N, STAGES, 1, SR, true>(in, w, bias, out, g, s, residual);
      return launch_ntw_impl<WAVES_M, WAVES_N, STAGES, 0, SR, true>(in, w, bias, out, g, s, residual);
    }
  }
#endif
  if (g.bn_sums) return launch_ntw_impl<WAVES_M, WAVES_N, STAGES, 2, SR>(in, w, bias, out, g, s, residual);
  if (g.stats) return launch_ntw_impl<WAVES_M, WAVES_N, STAGES, 1, SR>(in, w, bias, out, g, s, residual);
  return launch_ntw_impl<WAVES_M, WAVES_N, STAGES, 0, SR>(in, w, bias, out, g, s, residual);
}

