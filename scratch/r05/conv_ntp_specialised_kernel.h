// Round 5 experiment, NOT part of the library: the wide tiles with specialised waves (4 multiplying + 4 requesting).  Measured
// against conv_ntw_kernel on the stage-3 3x3 layer (gpurun_out/r05_ntw_ablate2.txt -> profiles/r05_ntw_ablation.txt): k-loop 107 vs 112 us,
// whole kernel 125 vs 124 us; MFMAs alone 85 vs 73.5 us (one multiplying wave per SIMD issues slower than two), requests alone 63 vs 67.
// To build it again: paste behind conv_ntw_kernel in dspnet_amd/csrc/conv_wide.h (scratch/r05/ntw_bench.hip has the runp<> driver).
// ---- the same tiles with the waves SPECIALISED (round 5, second step).  Ablations of conv_ntw_kernel on the stage-3 3x3 layer
// (scratch/r05/ntw_bench.hip, profiles/r05_ntw_ablation.txt): requests alone 67 us, MFMAs alone 72 us, both 112 us -- a wave that
// is waiting for the texture path to accept its next 1-KiB request is not issuing MFMAs, and all eight waves meet that queue
// together.  Here waves 0 - 3 (one per SIMD) only multiply -- 128 x 64 (64 x 128) outputs each, 48 MFMAs and 24 fragment reads per
// k-step, fragments double-buffered ACROSS k-steps so that no LDS latency is exposed -- and waves 4 - 7 (their SIMD partners)
// only request: 12 pieces per k-step each, two k-steps ahead.  One barrier per k-step, placed in the MIDDLE of a multiplying
// wave's k-step: by then it has all fragments of k-step kt in registers (ring slot kt is free for k-step kt + 3) and is about to
// read k-step kt + 1, which the requesting waves have waited for.
template <int TM, int TN, int EPI>
__global__ __launch_bounds__(512, 2) void conv_ntp_kernel(
    const float *__restrict__ in, const float *__restrict__ wgt, const float *__restrict__ bias,
    float *__restrict__ out, const ConvGeom g, const int m_tiles, const int n_tiles,
    const float *__restrict__ residual) {
  constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32, NTHR = 512, NWV = 8, STAGES = 3;
  constexpr int A_NI = BM / 32, B_NI = BN / 32, NI = A_NI + B_NI;      // 1-KiB pieces per REQUESTING wave (4 of them) and k-step
  constexpr int STG = (BM + BN) * 128;
#ifdef DSPN_ABLATE
  const int dbg = g.dbg;      // timing only: 1 no requests in the k-loop, 2 no MFMAs, 4 no fragment reads, 8 no barrier, 16 no epilogue
#else
  constexpr int dbg = 0;
#endif
  extern __shared__ __attribute__((aligned(1024))) char wsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= 4;
  const int ntiles = m_tiles * n_tiles;
  const int M = g.N * g.Hg * g.Wg;
  const int CB = g.Cin >> 5;
  const int nk = g.TR * g.TS * CB;
  const float sc_a = operand_scale(g.a_absmax), sc_b = operand_scale(g.b_absmax);
  const float inv_a = 1.f / sc_a, inv_b = 1.f / sc_b;
  float gmx_all = 0.f;

  if (loader) {
    // ================= requesting waves =================
    const int lw = wave - 4;
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, g.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wgt), 0, g.w_bytes, 0x00020000);
    constexpr unsigned kOOB = 0x80000000u;
    int a_row[A_NI], b_row[B_NI], a_cb[A_NI], b_cb[B_NI];
#pragma unroll
    for (int i = 0; i < A_NI; ++i) {
      const int line = 4 * (lw * A_NI + i) + (lane >> 4), sl = (lane & 15) ^ (line & 7);
      a_row[i] = 2 * line + (sl >> 3); a_cb[i] = (sl & 7) * 16;
    }
#pragma unroll
    for (int i = 0; i < B_NI; ++i) {
      const int line = 4 * (lw * B_NI + i) + (lane >> 4), sl = (lane & 15) ^ (line & 7);
      b_row[i] = 2 * line + (sl >> 3); b_cb[i] = (sl & 7) * 16;
    }
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
      const int tile = xcd_remap(t, ntiles);
      const int mt = tile / n_tiles, nt = tile - mt * n_tiles;
      const int m0 = mt * BM, n0 = nt * BN;
      int a_ih0[A_NI], a_iw0[A_NI], a_boff[A_NI], b_boff[B_NI];
      const int hw = g.Hg * g.Wg;
#pragma unroll
      for (int i = 0; i < A_NI; ++i) {
        const int m = m0 + a_row[i];
        const int n = m / hw, rem = m - n * hw;
        const int oi = rem / g.Wg, oj = rem - oi * g.Wg;
        const int ih0 = oi * g.ish + g.ioh, iw0 = oj * g.isw + g.iow;
        const bool mv = m < M;
        a_ih0[i] = mv ? ih0 : -0x40000000;
        a_iw0[i] = mv ? iw0 : 0;
        a_boff[i] = mv ? (((n * g.Hin + ih0) * g.Win + iw0) * g.Cin) * 4 + a_cb[i] : 0;
      }
#pragma unroll
      for (int i = 0; i < B_NI; ++i) {
        const int k = min(n0 + b_row[i], g.Cout - 1);
        b_boff[i] = k * (g.WTAPS * CB * 128) + b_cb[i];
      }
      int l_tr = 0, l_ts = 0, l_cb = 0;
      auto issue = [&](const int slot, const bool live) __attribute__((always_inline)) {
        const int dh = l_tr * g.idh, dw = l_ts * g.idw;
        const int a_tap = ((dh * g.Win + dw) * g.Cin + l_cb * 32) * 4;
        const int wtap = (g.wr0 + l_tr * g.wrs) * g.WS + g.ws0 + l_ts * g.wss;
        const int b_soff = (wtap * CB + l_cb) * 128;
        const unsigned oob = live ? 0u : kOOB;
        char *base = wsm + slot * STG;
        if (!(dbg & 1)) {
#pragma unroll
        for (int i = 0; i < A_NI; ++i) {
          const int ih = a_ih0[i] + dh, iw = a_iw0[i] + dw;
          const bool v = (unsigned)ih < (unsigned)g.Hin && (unsigned)iw < (unsigned)g.Win;
          const unsigned off = (unsigned)(a_boff[i] + a_tap) | (v ? oob : kOOB);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (__attribute__((address_space(3))) void *)(base + (lw * A_NI + i) * 1024),
                                                   16, (int)off, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_NI; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (__attribute__((address_space(3))) void *)(base + BM * 128 + (lw * B_NI + i) * 1024),
                                                   16, (int)((unsigned)b_boff[i] | oob), b_soff, 0, 0);
        }
        ++l_ts;
        const bool wrap = l_ts == g.TS;
        l_ts = wrap ? 0 : l_ts;
        l_tr += wrap ? 1 : 0;
        const bool wrap2 = l_tr == g.TR;
        l_tr = wrap2 ? 0 : l_tr;
        l_cb += wrap2 ? 1 : 0;
      };
      issue(0, true);
      issue(1, 1 < nk);
      int islot = 2;
      for (int kt = 0; kt < nk; ++kt) {
        // barrier kt: k-step kt has landed (k-step kt + 1 may still be in flight), and the multiplying waves hold every fragment
        // of k-step kt - 1: its ring slot takes k-step kt + 2
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
        if (!(dbg & 8)) __builtin_amdgcn_s_barrier();
        issue(islot, kt + 2 < nk);
        islot = islot + 1 == STAGES ? 0 : islot + 1;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // nothing may land in the staging area
      if (!(dbg & 8)) __builtin_amdgcn_s_barrier();         // barrier nk of the multiplying waves (their last fragments are in registers)
      __builtin_amdgcn_s_barrier();                         // the accumulators are staged
      if (!(dbg & 16)) wide_epilogue<BM, BN, NTHR, EPI>(g, wsm, m0, n0, M, tid, bias, out, residual, gmx_all);
    }
  } else {
    // ================= multiplying waves =================
    const int wm = (wave >> 1) * (TM * 32), wn = (wave & 1) * (TN * 32);
    const int frow = lane & 31;
    const int fslot = ((((frow & 1) << 3) | (lane >> 5)) ^ ((frow >> 1) & 7)) << 4;
    const int foff = (frow >> 1) * 256 + fslot;
    f32x16 acc[TM][TN];
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
      const int tile = xcd_remap(t, ntiles);
      const int mt = tile / n_tiles, nt = tile - mt * n_tiles;
      const int m0 = mt * BM, n0 = nt * BN;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      bf16x8 fa[2][2][TM], fb[2][2][TN];                    // [kk][piece][...]
      auto read_frags = [&](const int slot, const int kk) __attribute__((always_inline)) {    // kk: compile-time constant
        const char *sa = wsm + slot * STG + wm * 128, *sb = wsm + slot * STG + BM * 128 + wn * 128;
        if (dbg & 4) {
#pragma unroll
          for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int i = 0; i < TM; ++i) { fa[kk][p][i] = bf16x8{}; asm volatile("" : "+v"(fa[kk][p][i])); }
#pragma unroll
            for (int j = 0; j < TN; ++j) { fb[kk][p][j] = bf16x8{}; asm volatile("" : "+v"(fb[kk][p][j])); }
          }
          return;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[kk][1][i] = *reinterpret_cast<const bf16x8 *>(sa + i * 4096 + (foff ^ ((4 + 2 * kk) << 4)));
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[kk][0][j] = *reinterpret_cast<const bf16x8 *>(sb + j * 4096 + (foff ^ ((2 * kk) << 4)));
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[kk][0][i] = *reinterpret_cast<const bf16x8 *>(sa + i * 4096 + (foff ^ ((2 * kk) << 4)));
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[kk][1][j] = *reinterpret_cast<const bf16x8 *>(sb + j * 4096 + (foff ^ ((4 + 2 * kk) << 4)));
      };
      // x w = h0 g0 + h0 g1 + h1 g0, smallest terms first (conv_nt_kernel's order per accumulator)
      auto mma_half = [&](const int kk) __attribute__((always_inline)) {
        constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              if (dbg & 2) { asm volatile("" :: "v"(fa[kk][PA[t3]][i]), "v"(fb[kk][PB[t3]][j])); }
              else
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[kk][PA[t3]][i]),
                                                                 __builtin_bit_cast(f16x8, fb[kk][PB[t3]][j]), acc[i][j], 0, 0, 0);
            }
      };
      if (!(dbg & 8)) __builtin_amdgcn_s_barrier();          // barrier 0: k-step 0 has landed
      read_frags(0, 0);
      int slot = 0;
      for (int kt = 0; kt < nk; ++kt) {
        const int nslot = slot + 1 == STAGES ? 0 : slot + 1;
        read_frags(slot, 1);                                 // second half of k-step kt, under the first half's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        mma_half(0);
        __builtin_amdgcn_sched_barrier(0);
        // every fragment of k-step kt is in registers (the second half's MFMAs need them anyway): barrier kt + 1 -- k-step kt + 1
        // has landed, ring slot kt may be refilled
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!(dbg & 8)) __builtin_amdgcn_s_barrier();
        if (kt + 1 < nk) read_frags(nslot, 0);               // first half of k-step kt + 1, under the second half's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        mma_half(1);
        __builtin_amdgcn_sched_barrier(0);
        slot = nslot;
      }
      if (dbg & 16) { if (acc[0][0][0] == 1.2345e33f) out[0] = acc[TM - 1][TN - 1][5]; __builtin_amdgcn_s_barrier(); continue; }
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
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                          // the accumulators are staged
      wide_epilogue<BM, BN, NTHR, EPI>(g, wsm, m0, n0, M, tid, bias, out, residual, gmx_all);
    }
  }
  if constexpr (EPI == 2) wide_publish_absmax<NWV>(g, wsm, gmx_all, tid);
}


template <int TM, int TN, int EPI>
int launch_ntp_impl(const float *in, const float *w, const float *bias, float *out, const ConvGeom &g, hipStream_t s,
                    const float *residual) {
  constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32;
  const long long M = (long long)g.N * g.Hg * g.Wg;
  const int mt = (int)((M + BM - 1) / BM), nt = (g.Cout + BN - 1) / BN;
  const size_t lds = std::max<size_t>((size_t)3 * (BM + BN) * 128, sizeof(float) * BM * (BN + 4));
  auto kern = conv_ntp_kernel<TM, TN, EPI>;
  static int slots = 0, slots_cus = 0;
  if (!slots) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int dev = 0, cus = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    slots_cus = std::max(1, cus);
    slots = std::max(8, slots_cus / 8 * 8);                  // one workgroup per CU
  }
  const int reserved = dspn::reserved_cus();
  const int avail = reserved > 0 ? std::max(8, std::max(8, slots_cus - reserved) / 8 * 8) : slots;
  const int grid_x = (int)std::min<long long>((long long)mt * nt, avail);
  {
    dspn::ProfScope prof(0, s);
    hipLaunchKernelGGL(kern, dim3(grid_x), dim3(512), lds, s, in, w, bias, out, g, mt, nt, residual);
  }
  return dspn::check_launch("conv_ntp");
}
template <int TM, int TN>
int launch_ntp(const float *in, const float *w, const float *bias, float *out, const ConvGeom &g, hipStream_t s, const float *residual) {
  if (g.bn_sums) return launch_ntp_impl<TM, TN, 2>(in, w, bias, out, g, s, residual);
  if (g.stats) return launch_ntp_impl<TM, TN, 1>(in, w, bias, out, g, s, residual);
  return launch_ntp_impl<TM, TN, 0>(in, w, bias, out, g, s, residual);
}
