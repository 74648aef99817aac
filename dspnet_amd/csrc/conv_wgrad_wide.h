// The weight gradient of the plane-fed layers on the wide family's footing (round 6).  Included by conv.hip
// behind conv_wgrad_kernel (float build only), whose geometry (WgradGeom), split plan and slab layout it shares.
//
// Both operands are fp16 PIECE PLANES -- dy [pixel][Cout / 32][piece][32] written by the BatchNorm backward, x
// [pixel][Cin / 32][piece][32] written by the BatchNorm apply -- so a k-step (kPK = 32 pixels) of either operand is a pure copy:
// global -> LDS directly (buffer_load ... lds, 16 bytes per lane, 1 KiB per wave instruction), no registers, no vector work.
// The MFMA k dimension is the PIXEL, the images arrive pixel-major: fragments come out of LDS through ds_read_b64_tr_b16, as in
// conv_wgrad_kernel's bf16 mode, but from unpadded rows -- an LDS-DMA instruction writes 1 KiB contiguously, there is no room
// for row padding -- made conflict-free by an XOR on the SOURCE side instead:
//   row r (a pixel) of an image is BX * 4 bytes = BX / 32 records of [piece 0: 64 B][piece 1: 64 B] in memory order; its 64-byte
//   window u lands in window u ^ (r & 3) of the row.  The transposed read of a half-wave touches pixel rows q = 0 .. 3 at the
//   same logical window, i.e. four DIFFERENT physical windows = 256 consecutive bytes of bank space: no conflict.
// A wave owns 64 x 64 outputs (16 transposed reads of 8 bytes feed 12 MFMAs per 16-pixel block), four waves a 128 x 128 tile,
// two-slot ring, two workgroups per CU -- or eight waves a 128 x 256 tile, three slots, one workgroup: the requests of k-step
// kt + STAGES - 1 go out between the MFMAs of k-step kt (one counted wait and one raw barrier per k-step).  Per accumulator the pixel blocks and the three piece products come
// in conv_wgrad_kernel's order: the slabs are bit-identical to that kernel's (tests/test_nn_gpu.py, tests/test_wide_tiles_gpu.py).
#pragma once

template <int WAVES_M, int WAVES_N, int STAGES>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, 2) void conv_wgw_kernel(
    const st_t *__restrict__ x, const st_t *__restrict__ dy, float *__restrict__ slab, const WgradGeom g, const int k_tiles, const int j_tiles) {
  constexpr int BM = WAVES_M * 64, BN = WAVES_N * 64;       // BM over cout, BN over (tap, c)
  constexpr int NWV = WAVES_M * WAVES_N, NTHR = NWV * 64;
  constexpr int RA = BM * 4, RB = BN * 4;                   // bytes of one pixel's row of the A / B image
  constexpr int A_LPR = RA / 16, B_LPR = RB / 16;           // lanes (16-byte chunks) per row
  constexpr int A_RPI = 64 / A_LPR, B_RPI = 64 / B_LPR;     // rows per wave instruction
  constexpr int A_NI = kPK / A_RPI / NWV, B_NI = kPK / B_RPI / NWV, NI = A_NI + B_NI;
  static_assert(A_LPR >= 16 && A_LPR <= 64 && B_LPR >= 16 && B_LPR <= 64, "rows of 256 .. 1024 bytes");
  static_assert(kPK % (A_RPI * NWV) == 0 && kPK % (B_RPI * NWV) == 0, "whole instructions per wave");
  static_assert((NWV * A_RPI) % 4 == 0 && (NWV * B_RPI) % 4 == 0, "a lane's rows share r & 3");
  constexpr int STG = kPK * (RA + RB);
  constexpr int D = STAGES - 1;                             // k-steps in flight ahead of the one being multiplied
  static_assert(STAGES >= 2 && STAGES <= 4, "ring depth");
  extern __shared__ __attribute__((aligned(1024))) char wsm[];

  DSPN_WGRAD_JOB_ROWS(g, wsm)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lin = xcd_remap(block_y * gridDim.x + blockIdx.x, gridDim.x * grid_y);     // (as conv_wgrad_kernel)
  const int split = lin / (int)gridDim.x, tile = lin - split * (int)gridDim.x;
  const int kt_i = tile / j_tiles, jt_i = tile - kt_i * j_tiles;
  const int k0 = kt_i * BM, j0 = jt_i * BN;
  const int P = g.N * g.Ho * g.Wo;
  const int J = g.R * g.S * g.Cin;
  const int p_begin = split * g.pix_per_split;
  const int p_end = min(P, p_begin + g.pix_per_split);
  const int nk = p_end > p_begin ? (p_end - p_begin + kPK - 1) / kPK : 0;
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<st_t *>(x), 0, g.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<st_t *>(dy), 0, g.dy_bytes, 0x00020000);
  constexpr unsigned kOOB = 0x80000000u;

  // ---- loader.  Instruction n = i * NWV + wave of an image fills rows n * RPI .. + RPI - 1 (1 KiB); lane l writes chunk l % LPR
  // of row n * RPI + l / LPR and FETCHES chunk (l % LPR) ^ ((r & 3) << 2) of that pixel's row.  r & 3 is the same for all of a
  // lane's rows (NWV * RPI is a multiple of 4): the chunk a lane fetches -- channel block, piece, tap -- is fixed for the kernel
  const int a_h = lane / A_LPR, a_r3 = (wave * A_RPI + a_h) & 3, a_c = (lane % A_LPR) ^ (a_r3 << 2);
  const int b_h = lane / B_LPR, b_r3 = (wave * B_RPI + b_h) & 3, b_c = (lane % B_LPR) ^ (b_r3 << 2);
  // A (dy): row of pixel p = bytes [p * Cout * 4 + k0 * 4, + RA) of the planes; 32-channel blocks past Cout are not fetched
  const bool a_kv = k0 + (a_c >> 3) * 32 < g.Cout;
  unsigned a_off[A_NI];
  int a_left[A_NI];                       // pixels from this lane's row to the end of the split (<= 0: the row is not fetched)
#pragma unroll
  for (int i = 0; i < A_NI; ++i) {
    const int r = (i * NWV + wave) * A_RPI + a_h;
    a_off[i] = (unsigned)((p_begin + r) * g.Cout + k0) * 4u + (unsigned)a_c * 16u;
    a_left[i] = p_end - p_begin - r;
  }
  const unsigned a_step = (unsigned)(kPK * g.Cout) * 4u;
  // B (x): column block b_c >> 3 of the tile is (tap, 32 channels) -- fixed per lane; the pixel moves
  const int jb = j0 + (b_c >> 3) * 32;
  const bool b_jv = jb < J;
  const int tap = jb / g.Cin, cb = (jb - tap * g.Cin) >> 5;
  const int tr = tap / g.S, ts = tap - tr * g.S;
  const int tdh = tr * g.dh - g.ph, tdw = ts * g.dw - g.pw;
  const unsigned b_in = (unsigned)cb * 128u + (unsigned)(b_c & 7) * 16u;
  int b_n[B_NI], b_ho[B_NI], b_wo[B_NI], b_left[B_NI];
  {
    const int hw = g.Ho * g.Wo;
#pragma unroll
    for (int i = 0; i < B_NI; ++i) {
      const int r = (i * NWV + wave) * B_RPI + b_h;
      const int p = p_begin + r;
      const int n = p / hw, rem = p - n * hw;
      b_n[i] = n; b_ho[i] = rem / g.Wo; b_wo[i] = rem - (rem / g.Wo) * g.Wo;
      b_left[i] = p_end - p_begin - r;
    }
  }
  const int adv_n = kPK / (g.Ho * g.Wo), adv_rem = kPK - adv_n * (g.Ho * g.Wo);
  const int adv_h = adv_rem / g.Wo, adv_w = adv_rem - adv_h * g.Wo;

  int q_done = 0;                         // pixels of the split before the k-step being requested
  char *q_base = wsm;
  auto issue_begin = [&](const int slot, const int kt) __attribute__((always_inline)) {
    q_done = kt * kPK;
    q_base = wsm + slot * STG;
  };
  auto issue_piece = [&](const int i) __attribute__((always_inline)) {     // i: compile-time constant, A pieces first
    if (i < A_NI) {
      const int ii = i < A_NI ? i : 0;
      const unsigned off = a_off[ii] | ((a_kv && q_done < a_left[ii]) ? 0u : kOOB);
      a_off[ii] += a_step;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_dy, (__attribute__((address_space(3))) void *)(q_base + (ii * NWV + wave) * 1024),
                                               16, (int)off, 0, 0, 0);
    } else {
      const int j = i - A_NI < 0 ? 0 : (i - A_NI < B_NI ? i - A_NI : 0);
      const int n = b_n[j], ho = b_ho[j], wo = b_wo[j];
      {   // the next k-step's pixel: + (adv_n images, adv_h rows, adv_w columns), one carry per level (conv_wgrad_kernel)
        int w2 = wo + adv_w, h2 = ho + adv_h, n2 = n + adv_n;
        const bool cw = w2 >= g.Wo;
        w2 -= cw ? g.Wo : 0; h2 += cw ? 1 : 0;
        const bool chh = h2 >= g.Ho;
        h2 -= chh ? g.Ho : 0; n2 += chh ? 1 : 0;
        b_wo[j] = w2; b_ho[j] = h2; b_n[j] = n2;
      }
      const int ih = ho * g.sh + tdh, iw = wo * g.sw + tdw;
      const bool v = b_jv && q_done < b_left[j] && (unsigned)ih < (unsigned)g.Hin && (unsigned)iw < (unsigned)g.Win;
      const unsigned off = ((unsigned)(((n * g.Hin + ih) * g.Win + iw) * g.Cin) * 4u + b_in) | (v ? 0u : kOOB);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void *)(q_base + kPK * RA + (j * NWV + wave) * 1024),
                                               16, (int)off, 0, 0, 0);
    }
  };

  // ---- fragments (conv_wgrad_kernel's transposed reads): 16-lane group gl = lane >> 4 covers channels 16 (gl & 1) .. + 15 of a
  // 32-channel block and pixels 8 (gl >> 1) .. + 7 of a 16-pixel block; lane 4 q + pp of the group supplies the address of pixel
  // row q, channels 4 pp .. + 3, and receives its own channel's 4 pixels; the second read is 4 pixel rows further.  Window u =
  // 2 block + piece of the row sits at u ^ q: the lane's base carries q in the window bits, the fragment XORs its u in
  const int wm = (wave / WAVES_N) * 64, wn = (wave % WAVES_N) * 64;
  const int gl = lane >> 4, fq = (lane & 15) >> 2, fpp = lane & 3;
  const int fa0 = (8 * (gl >> 1) + fq) * RA + wm * 4 + ((fq << 6) | ((gl & 1) * 32 + fpp * 8));
  const int fb0 = kPK * RA + (8 * (gl >> 1) + fq) * RB + wn * 4 + ((fq << 6) | ((gl & 1) * 32 + fpp * 8));
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto frag = [](const char *p, const int row_bytes) __attribute__((always_inline)) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(p));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4 *)(p + 4 * row_bytes));
    bf16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return r;
  };
  auto mma_step = [&](const int slot) __attribute__((always_inline)) {
    const char *s = wsm + slot * STG;
    bf16x8 fa[2][2][2], fb[2][2][2];            // [16-pixel block][piece][32-channel block]
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int pc = 0; pc < 2; ++pc)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          fa[kk][pc][i] = frag(s + (fa0 ^ ((2 * i + pc) << 6)) + kk * 16 * RA, RA);
          fb[kk][pc][i] = frag(s + (fb0 ^ ((2 * i + pc) << 6)) + kk * 16 * RB, RB);
        }
    __builtin_amdgcn_sched_barrier(0);
    // x w = h1 g0 + h0 g1 + h0 g0 per 16-pixel block, smallest terms first (conv_wgrad_kernel's order per accumulator)
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
    constexpr int NMMA = 2 * 3 * 4;
    static_assert(NI <= NMMA, "at most one request per MFMA");
#pragma unroll
    for (int m = 0; m < NMMA; ++m) {
      const int kk = m / 12, t3 = (m / 4) % 3, i = (m / 2) % 2, j = m % 2;
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[kk][PA[t3]][i]),
                                                         __builtin_bit_cast(f16x8, fb[kk][PB[t3]][j]), acc[i][j], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NI; ++q)
        if (((q + 1) * NMMA) / NI - 1 == m) {
          issue_piece(q);
          __builtin_amdgcn_sched_barrier(0);
        }
    }
  };

  // (k-steps past the split are requested all the same, every lane out of range: zeros into a slot nobody reads, no memory
  // traffic -- one instruction stream, one counted wait)
#pragma unroll
  for (int j = 0; j < D; ++j) {
    issue_begin(j, j);
#pragma unroll
    for (int q = 0; q < NI; ++q) issue_piece(q);
  }
  int slot = 0, islot = D % STAGES;
  for (int kt = 0; kt < nk; ++kt) {
    // the images of k-step kt have landed (this wave's pieces: the counted wait -- D - 1 k-steps stay in flight; the other
    // waves': the barrier) and every wave has read the fragments of k-step kt - 1, whose slot the requests of k-step kt + D go into
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * NI) : "memory");
    __builtin_amdgcn_s_barrier();
    issue_begin(islot, kt + D);
    mma_step(slot);
    slot = slot + 1 == STAGES ? 0 : slot + 1;
    islot = islot + 1 == STAGES ? 0 : islot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();           // nothing in flight, every fragment read: the LDS becomes the staging area

  // slab[split][k][J], as conv_wgrad_kernel: through LDS, float4 rows along J
  const float inv_a = 1.f / operand_scale(g.dy_absmax), inv_b = 1.f / operand_scale(g.x_absmax);
  float *o = slab + (long long)split * g.Cout * J;
  constexpr int SLD = BN + 4;
  float *st = reinterpret_cast<float *>(wsm);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        st[(wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * SLD + wn + j * 32 + (lane & 31)] = acc[i][j][r] * inv_a * inv_b;
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
