// EXPERIMENT, NOT BUILT INTO THE LIBRARY (round 3; DESIGN.md section 5 "Round 3" (e)).  Kept for the next round.
// It was wired into igemm.hip as its own sr_igemm_args.tile value (13 at the time; 13-15 are the deep-ring tiles now) (dispatch: `if (force == 13) { if (geglu_xs_ok(a, M)) return launch_geglu_xs(a, M, st); }`).
// Status when it was taken out:
//   * timing (M 65536, K 320, N 2560, one box): 201-208 us vs 199 us (256x320 tile) and 196 us (128x320 tile): no gain;
//     ablations: without the epilogue 114.5 us = 938 TF/s (the GEGLU epilogue is 43 % of the kernel), without W staging 178 us;
//   * correctness: parity-clean at K = 320 (8 chunks) in every run, but at K = 128, N = 640 (KS = 4, two chunks) 190-200 of 200
//     launches produced wrong values (lane 31 of waves 4..7, or NaN).  RESOLVED after it was taken out: the epilogue's
//     `global_store_dwordx4` statements were issued from inline asm WITHOUT the `s_nop 1` that hipcc pads a >64-bit store with
//     (the instruction after it may overwrite the store's data registers before the store has read them; hipcc does not model
//     what is inside an asm string).  With `s_nop 1` inside both store strings: 0 of 200 bad launches at K = 128 and K = 320,
//     output bit-identical to the 256x320 tile's.  The counted vmcnt waits were right.  (The earlier observation "vmcnt(0) at
//     every step fixes it" changed the schedule around the stores, not the ordering.)
//   * so the kernel is ordered, but still no faster than the tile kernels => not shipped; next step = deferred epilogue.
// ---- X-stationary GEGLU (tile 13): 128 x (all of N), 8 waves -----------------------------------------------------------------
// The K-short GEGLU projection of a transformer block (ff.net.0: K = C = 320, N = 8C; attention.py:60-90) through the tile kernels
// above is one workgroup per 256 x 320 output tile: ~20 us of lifetime of which ~10 are MFMA (3 us of ramp -- arguments, first
// stage in flight -- and 6.3 us of epilogue), and the 256 x 320 activation tile is staged again by each of the N / 320 workgroups
// that share it.  Here a workgroup keeps its 128 x K activation tile RESIDENT in LDS (80 KB at K = 320) and streams all of W past
// it: N / 320 chunks of 320 interleaved (value, gate) columns, each chunk KS = K / 32 half-steps of [320 rows][64 B] through a
// 3-slot ring that never drains between chunks -- one ramp per workgroup instead of one per tile, the activations staged once.
//   LDS: [bias | colsum of the chunk: 2560 B] [X: KS x 128 x 64 B] [W ring: 3 x 20 KB] [epilogue staging: 8 x 16 rows x 96 B].
//   Every vector-memory operation inside the loop is issued from inline asm (LDS-DMA loads, the epilogue's stores), so the counted
//   waits below are exact: at the top of step s = (chunk c, half-step j) the operations younger than stage s are
//     stage s+1 (3), plus, right behind a chunk boundary, the previous epilogue's 8 stores (j = 0, 1) and the bias piece (j = 1).
//   Needs M % 128 == 0, N % 320 == 0, 128 <= K <= 320, fp16, no residual / rowvec (what the 64 x 64 level's ff.net.0 is).
// Measured (M 65536, K 320, N 2560, one box, tools/bench_igemm.py): 201-208 us against 199 us for the 256 x 320 tile and 196 us for
// 128 x 320 -- no gain yet -- but the ablations say where the layer's time is: without the epilogue (SR_GX_DBG=1) the loop runs in
// 114.5 us = 938 TF/s, i.e. the GEGLU epilogue (80 accumulators per lane: scale / fold / bias, 40 erf-GELUs, fp16 pack, LDS
// staging, stores) is 43 % of the kernel; without the W staging (SR_GX_DBG=2) 178 us.  The next step is the deferred epilogue (the
// previous chunk's 80 values packed to fp16 pairs and finished in slices under this chunk's MFMAs); the tuner does not offer tile 13
// until then.
// DBG (development, SR_GX_DBG): 1 = no epilogue arithmetic / stores, 2 = no W staging after the prologue (MFMA + fragment reads only)
template <int KS, int DBG = 0>
__global__ __launch_bounds__(512, 1) void geglu_xs_kernel(const sr_igemm_args p, const int M, const int nchunks) {
  using T = _Float16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int VEC_B = 2 * 320 * 4, XB = KS * 128 * 64, WSTAGE = 320 * 64, R = 3, D = R - 1;
  constexpr int TM = 4, TN = 5, ST = 8, PS = 3;            // fragments of the 64 x 80 wave tile; stores per epilogue; DMA pieces per stage
  constexpr int STE = DBG == 1 ? 0 : ST;
  constexpr int STG_ROWB = 96, STG_WAVE = 16 * STG_ROWB;
  float* const lvec = (float*)smem;                          // [320 bias][320 colsum] of the current chunk
  char* const sX = smem + VEC_B;
  char* const sW = sX + XB;
  char* const sStg = sW + R * WSTAGE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * 128;
  const int C1 = p.C1, N = p.N, ldo = N >> 1;
  const int c16 = lane & 15, g4 = lane >> 4;
  const int wm = wv >> 2, wn = wv & 3, pm0 = wm * 64, qn0 = wn * 80;
  // folded-LayerNorm row statistics of this lane's four rows: loaded (by the compiler) before any LDS-DMA is in flight
  float2 rs[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    rs[tm] = make_float2(1.f, 0.f);
    if (p.row_stats) rs[tm] = *(const float2*)(p.row_stats + 2 * (int64_t)(m0 + pm0 + tm * 16 + c16));
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) asm volatile("" : "+v"(rs[tm].x), "+v"(rs[tm].y));      // the compiler's wait for them sits HERE, not in the loop
  const float scale = p.scale;
  // ---- source addresses (64-byte K-steps: 16 rows x 4 chunks per LDS-DMA instruction; source-side swizzle of igemm_kernel)
  const int lrow = lane >> 2;
  const int lchunk = (0x1320 >> (4 * ((lane & 3) ^ ((lrow >> 2) & 3)))) & 3;
  const char* xsrc = (const char*)p.a + ((int64_t)(m0 + wv * 16 + lrow) * C1) * 2 + lchunk * 16;      // row group wv, all half-steps
  const char* wsrc[PS];
#pragma unroll
  for (int i = 0; i < PS; ++i) {
    const int gi = (i * 8 + wv) < 20 ? (i * 8 + wv) : 19;
    wsrc[i] = (const char*)p.w + ((int64_t)(gi * 16 + lrow) * C1) * 2 + lchunk * 16;
  }
  const unsigned ldsX = __builtin_amdgcn_readfirstlane(sr_lds_addr(sX)), ldsW = __builtin_amdgcn_readfirstlane(sr_lds_addr(sW));
  const unsigned ldsV = __builtin_amdgcn_readfirstlane(sr_lds_addr(smem));
  const int64_t chunk_adv = (int64_t)320 * C1 * 2 - (int64_t)KS * 64;        // from the end of a chunk's K range to the next chunk's rows
  int lj = 0;                                                                  // half-step of the next stage to issue
  auto issue_stage = [&](int slot) {
#pragma unroll
    for (int i = 0; i < PS; ++i) {
      const int gi = (i * 8 + wv) < 20 ? (i * 8 + wv) : 19;
      sr_glds16_asm_nosave(wsrc[i], ldsW + slot * WSTAGE + gi * 1024);
      wsrc[i] += 64;
    }
    if (++lj == KS) {
      lj = 0;
#pragma unroll
      for (int i = 0; i < PS; ++i) wsrc[i] += chunk_adv;
    }
  };
  const float* const vec_bias = p.bias;                      // (uniform values: a per-lane choice between the FIELDS of `p` makes the
  const float* const vec_csum = p.row_stats ? p.colsum : nullptr;   //  compiler load them per lane from the argument block, in the loop)
  auto issue_bias = [&](int c) {
    // [bias | colsum] of chunk c = 160 sixteen-byte slots; waves 0..2 cover them, waves 3..7 repeat wave 2's piece (identical
    // bytes), so every wave issues exactly one operation; lanes past slot 159 are masked (they would land in the X tile)
    const int w3 = wv < 3 ? wv : 2;
    const int sl = w3 * 64 + lane;
    if (sl < 160) {
      const bool second = sl >= 80;
      const char* sb = vec_bias ? (const char*)(vec_bias + c * 320 + 4 * sl) : (const char*)p.zero_page;
      const char* sc = vec_csum ? (const char*)(vec_csum + c * 320 + 4 * (sl - 80)) : (const char*)p.zero_page;
      sr_glds16_asm_nosave(second ? sc : sb, ldsV + w3 * 1024);
    }
  };
  // ---- prologue: the resident X tile (KS pieces per wave), the first chunk's bias, the first D stages
  {
    const unsigned m0k = sr_m0_save();
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) sr_glds16_asm_nosave(xsrc + ks * 64, ldsX + ks * 8192 + wv * 1024);
    issue_bias(0);
    const int total0 = nchunks * KS;
#pragma unroll
    for (int i = 0; i < D; ++i) if (i < total0) issue_stage(i);
    sr_m0_restore(m0k);
  }
  // ---- fragment read offsets (64-byte rows: igemm_kernel's BKB = 64 layout)
  const int foff = c16 * 64 + ((((c16 >> 2) & 3) ^ ((0x2130 >> (4 * g4)) & 3)) << 4);
  const char* fX = sX + pm0 * 64 + foff;
  const char* fW = sW + qn0 * 64 + foff;
  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  char* const stg = sStg + wv * STG_WAVE;
  // read-back geometry of a 16-row pass: 80 sixteen-byte chunks (5 per row); instruction 0 takes chunks 0..63, instruction 1 chunks
  // 64..79 in its first 16 lanes -- its other lanes repeat them (identical bytes to identical addresses: every lane stores, so both
  // instructions always execute and the operation count per epilogue is exactly ST)
  const int ch0 = lane, ch1 = 64 + (lane & 15);
  const int r0 = ch0 / 5, k0 = ch0 - r0 * 5, r1 = ch1 / 5, k1 = ch1 - r1 * 5;
  const int total = nchunks * KS;
  int s = 0, slot = 0;
  for (int c = 0; c < nchunks; ++c) {
    auto step = [&](auto jc) {
      constexpr int j = decltype(jc)::value;
      // operations of this wave younger than stage s (see the header comment); never more than the real count
      const bool more = s + 1 < total && DBG != 2;          // stage s + 1 exists (it was issued one step ago)
      if constexpr (DBG == 3) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      else if constexpr (j == 0) {
        if (c > 0) { if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PS + STE) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STE) : "memory"); }
        else       { if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PS) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      } else if constexpr (j == 1) {
        if (c > 0) { if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PS + STE + 1) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STE + 1) : "memory"); }
        else       { if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PS) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      } else {
        if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PS) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();                         // stage s visible to all; everyone is done with step s - 1 (and its epilogue)
      {
        const unsigned m0k = sr_m0_save();
        if (j == 0 && c > 0) issue_bias(c);
        if (s + D < total && DBG != 2) { int nb = slot + D; if (nb >= R) nb -= R; issue_stage(nb); }
        sr_m0_restore(m0k);
      }
      uint4 xf[TM], wf[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) xf[t] = *(const uint4*)(fX + j * 8192 + t * 1024);
#pragma unroll
      for (int t = 0; t < TN; ++t) wf[t] = *(const uint4*)(fW + slot * WSTAGE + t * 1024);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) sr_mma(acc[tn][tm], wf[tn], xf[tm], T());
      ++s;
      if (++slot == R) slot = 0;
    };
    if constexpr (KS > 0) step(std::integral_constant<int, 0>{});
    if constexpr (KS > 1) step(std::integral_constant<int, 1>{});
    if constexpr (KS > 2) step(std::integral_constant<int, 2>{});
    if constexpr (KS > 3) step(std::integral_constant<int, 3>{});
    if constexpr (KS > 4) step(std::integral_constant<int, 4>{});
    if constexpr (KS > 5) step(std::integral_constant<int, 5>{});
    if constexpr (KS > 6) step(std::integral_constant<int, 6>{});
    if constexpr (KS > 7) step(std::integral_constant<int, 7>{});
    if constexpr (KS > 8) step(std::integral_constant<int, 8>{});
    if constexpr (KS > 9) step(std::integral_constant<int, 9>{});
    // ---- epilogue of chunk c: bias / folded LayerNorm / GEGLU in registers, 16 rows at a time through the wave's staging rows,
    // 16-byte stores issued from asm (exactly ST = 8 per wave)
    char* const obase = (char*)p.out + ((int64_t)(m0 + pm0) * ldo + c * 160 + wn * 40) * 2;
    if constexpr (DBG == 1) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) { asm volatile("" ::"v"(acc[tn][tm])); acc[tn][tm] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      continue;
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int nl = qn0 + tn * 16 + 4 * g4;
        float v[4] = {acc[tn][tm][0] * scale, acc[tn][tm][1] * scale, acc[tn][tm][2] * scale, acc[tn][tm][3] * scale};
        const float4 cs = *(const float4*)(lvec + 320 + nl);
        const float4 bv = *(const float4*)(lvec + nl);
        v[0] = fmaf(rs[tm].x, v[0], rs[tm].y * cs.x) + bv.x; v[1] = fmaf(rs[tm].x, v[1], rs[tm].y * cs.y) + bv.y;
        v[2] = fmaf(rs[tm].x, v[2], rs[tm].y * cs.z) + bv.z; v[3] = fmaf(rs[tm].x, v[3], rs[tm].y * cs.w) + bv.w;
        h16x2 hv = {(_Float16)(v[0] * sr_gelu_f(v[1])), (_Float16)(v[2] * sr_gelu_f(v[3]))};
        *(h16x2*)(stg + c16 * STG_ROWB + (tn * 8 + 2 * g4) * 2) = hv;
        acc[tn][tm] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);                    // lgkmcnt(0): the wave's own rows are in LDS
      __builtin_amdgcn_wave_barrier();
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      const u32x4 d0 = *(const u32x4*)(stg + r0 * STG_ROWB + k0 * 16);
      const u32x4 d1 = *(const u32x4*)(stg + r1 * STG_ROWB + k1 * 16);
      char* a0 = obase + (int64_t)(tm * 16 + r0) * ldo * 2 + k0 * 16;
      char* a1 = obase + (int64_t)(tm * 16 + r1) * ldo * 2 + k1 * 16;
      asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(a0), "v"(d0) : "memory");   // s_nop 1: see the header
      asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(a1), "v"(d1) : "memory");
      __builtin_amdgcn_wave_barrier();                       // (LDS is in order per wave: the next pass may overwrite)
    }
    if constexpr (DBG == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static bool geglu_xs_ok(const sr_igemm_args& a, int M) {
  return a.dtype == SR_F16 && a.act == 2 && a.KH == 1 && a.stride == 1 && !a.upsample && !a.C2 && !a.transpose_out && !a.residual &&
         !a.rowvec && !a.out_f32 && !a.pad_br && a.N % 320 == 0 && M % 128 == 0 && a.C1 % 64 == 0 && a.C1 <= 320 && a.C1 >= 128;   // (KS >= 3: the chunk's bias piece is covered by the wait of half-step 2)
}

static int launch_geglu_xs(const sr_igemm_args& a, int M, hipStream_t st) {
  const int KS = a.C1 / 32;
  const int lds = 2 * 320 * 4 + KS * 128 * 64 + 3 * 320 * 64 + 8 * 16 * 96;
  const dim3 grid(M / 128), block(512);
#define SR_GX(K) { auto k = geglu_xs_kernel<K>; static bool set_ = false; if (!set_) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds); set_ = true; } \
    hipLaunchKernelGGL(k, grid, block, lds, st, a, M, a.N / 320); }
  static const int dbg = getenv("SR_GX_DBG") ? atoi(getenv("SR_GX_DBG")) : 0;
  if (dbg == 1 && KS == 10) { auto k = geglu_xs_kernel<10, 1>; (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds); hipLaunchKernelGGL(k, grid, block, lds, st, a, M, a.N / 320); return SR_OK; }
  if (dbg == 4 && KS == 4) { auto k = geglu_xs_kernel<4, 4>; (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds); hipLaunchKernelGGL(k, grid, block, lds, st, a, M, a.N / 320); return SR_OK; }
  if (dbg == 3 && KS == 4) { auto k = geglu_xs_kernel<4, 3>; (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds); hipLaunchKernelGGL(k, grid, block, lds, st, a, M, a.N / 320); return SR_OK; }
  if (dbg == 2 && KS == 10) { auto k = geglu_xs_kernel<10, 2>; (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds); hipLaunchKernelGGL(k, grid, block, lds, st, a, M, a.N / 320); return SR_OK; }
  switch (KS) {
    case 4: SR_GX(4) break; case 6: SR_GX(6) break; case 8: SR_GX(8) break; case 10: SR_GX(10) break;
    default: SR_FAIL(SR_ERR_INVALID, "sr_igemm: tile 13 (X-stationary GEGLU) is built for K = 128, 192, 256, 320 (K = %d)", a.C1);
  }
#undef SR_GX
  SR_CHECK_LAUNCH("sr_igemm(geglu_xs)");
  return SR_OK;
}

