// Implicit-GEMM convolution / linear layer on gfx950 MFMA (see include/sr_hip.h: sr_igemm).
//
// GEMM view:  out[m, n] = sum_k X[m, k] * Wt[n, k],  m = output pixel (b, oy, ox), k = (ky, kx, c).
// Both operands are K-contiguous, so both LDS tiles are [rows][128 bytes] and are read with the same
// 16-byte fragment pattern.  Per K-step (128 bytes of K = 64 halfs / 32 floats):
//   * every wave issues global_load_lds (16 B/lane, async, no VGPR round trip) for its share of the X and W
//     tiles into the *other* LDS buffer.  The LDS image is lane-linear (8 rows x 8 chunks per wave
//     instruction), so the XOR swizzle (chunk ^= row&7, conflict-free ds_read_b128) is applied to the SOURCE
//     address; the im2col gather (tap offsets, zero padding via a zero page, fused nearest-upsample,
//     stride 2, channel-concat of two sources) is just the per-lane source address.
//   * each wave multiplies its (BM/WAVES_M) x (BN/WAVES_N) sub-tile with v_mfma_f32_16x16x32_f16 (fp16) or
//     v_mfma_f32_16x16x4_f32 (exact fp32) out of the current buffer.
// One barrier per K-step, a ring of 2..4 LDS slots with the loads 1..3 K-steps ahead.  The LDS-DMA is issued from inline asm
// and ordered by hand (counted s_waitcnt vmcnt + raw s_barrier): hipcc's own wait-count pass would drain every in-flight
// LDS-DMA before the next ds_read.  Weights are the MFMA "A" operand so that every lane ends up
// holding 4 consecutive output channels of one pixel (8/16-byte stores, vector bias/residual loads); with
// transpose_out the roles swap and a lane holds 4 consecutive pixels of one channel (V^T for attention).
#include "sr_common.h"
#ifndef SR_IGEMM_TRACE
#define SR_IGEMM_TRACE 0        // development only: wave 0 of every workgroup writes wall-clock stamps (100 MHz) of its phases to p.workspace
#endif
#if SR_IGEMM_TRACE
#define SR_TS(k) do { if (threadIdx.x == 0) ts_[k] = wall_clock64(); } while (0)
#else
#define SR_TS(k) do { } while (0)
#endif
#include <stdlib.h>
#include <type_traits>

namespace {

// Folded LayerNorm (sr_igemm_args.row_stats / .ln_inline): the (rstd, -rstd * mean) pair of input row m.  With ln_inline the
// kernel has left the pairs of its BM rows at the front of the dynamic LDS (the prefetch landing zone, dead by then).
#define SR_FOLD(p) ((p).row_stats != nullptr || (p).ln_inline != 0)
__device__ __forceinline__ float2 sr_fold_stat(const sr_igemm_args& p, const int m, const int m0) {
  extern __shared__ __attribute__((aligned(16))) char sr_lds_front[];
  if (p.ln_inline) return *(const float2*)(sr_lds_front + 8 * (m - m0));
  return *(const float2*)(p.row_stats + 2 * (int64_t)m);
}

// fp16-output form of the row-major epilogue (the coalesced path of epilogue_rows below, same arithmetic and the same stores),
// restructured after a per-workgroup timeline of the K-short layers (tools/trace_igemm.py) showed the epilogue taking 4.4 us of a
// 12.5 us workgroup lifetime on the 128x160 tile and 8.4 of 23.5 us on 128x320: on CDNA4 vmcnt counts stores too, so every pass
// that loaded its bias (and every read-back iteration that loaded its residual chunk) AFTER the previous pass's global stores
// waited for those stores to retire -- ~2 us per pass.  Here no global LOAD of the epilogue follows a store: the tile's bias /
// colsum rows go through LDS once (read back with ds_read, which vmcnt does not cover), the wave's residual chunks are loaded
// up front when the register budget VCAP allows, and a pass covers as many 16-row groups as the staging LDS holds at the fp16
// row size (the old bound assumed fp32 rows).
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int LDS_AVAIL, bool GEGLU, int VCAP, bool VECPRE>
__device__ __forceinline__ void epilogue_rows_h(const sr_igemm_args& p, f32x4 (&acc)[BN / WAVES_N / 16][BM / WAVES_M / 16], char* smem, const int M,
                                                const int rpb, const int m0, const int n0, const int pm0, const int qn0, const int wv,
                                                const int lane) {
  constexpr int NW = WAVES_M * WAVES_N, TM = BM / WAVES_M / 16, TN = BN / WAVES_N / 16;
  constexpr int WCOLS = BN / WAVES_N, OCOLS = GEGLU ? WCOLS / 2 : WCOLS;
  constexpr int ROWB = OCOLS * 2 + 16;                       // padded LDS row (bank spread, keeps 16-B alignment)
  constexpr int CPR = OCOLS * 2 / 16, RPI = 64 / CPR;        // 16-B chunks per row, rows per wave instruction
  constexpr int VEC_B = 2 * BN * 4;                          // bias + colsum rows of the tile (fp32)
  constexpr int LDS_ROWS = VECPRE ? LDS_AVAIL : LDS_AVAIL - VEC_B;   // VECPRE: the kernel put them in front of `smem` by LDS-DMA at its start
  // 16-row groups per pass: as many as the staging LDS holds -- unless a smaller pass is what lets the residual chunks be
  // requested ahead of the stores within the register cap (PRE: all of them up front; ROLL: the next pass's, two rolling sets)
  constexpr int ACC = TM * TN * 4;
  constexpr auto iters = [](int tmp) { return (tmp * 16 + RPI - 1) / RPI; };
  constexpr auto fits = [](int tmp) { return tmp >= 1 && TM % tmp == 0 && NW * tmp * 16 * ROWB <= LDS_ROWS; };
  constexpr auto pre_ok = [](int tmp) { return ACC + (TM / tmp) * ((tmp * 16 + RPI - 1) / RPI) * 4 + 48 <= VCAP; };
  constexpr auto roll_ok = [](int tmp) { return TM / tmp > 1 && ACC + 2 * ((tmp * 16 + RPI - 1) / RPI) * 4 + 40 <= VCAP; };
  constexpr int T_LDS = fits(TM) ? TM : (fits(TM / 2) ? TM / 2 : 1);
  constexpr int TMP = pre_ok(T_LDS) ? T_LDS : (roll_ok(T_LDS) ? T_LDS : ((T_LDS > 1 && roll_ok(T_LDS / 2) && fits(T_LDS / 2)) ? T_LDS / 2 : ((T_LDS > 1 && roll_ok(1)) ? 1 : T_LDS)));
  constexpr int ROWS = TMP * 16, ITER = iters(TMP), NPASS = TM / TMP;
  constexpr bool PRE = pre_ok(TMP);
  constexpr bool ROLL = !PRE && roll_ok(TMP);
  static_assert(OCOLS % 8 == 0 && NW * 16 * ROWB <= LDS_ROWS, "epilogue staging");
  const int c16 = lane & 15, g4 = lane >> 4;
  const float scale = p.scale;
  const int N = p.N;
  const int ldr = p.rowvec_ld ? p.rowvec_ld : N;
  const int ldo = GEGLU ? (N >> 1) : N;
  float* lvec = (float*)(VECPRE ? smem - VEC_B : smem);      // [BN] bias, [BN] colsum
  char* wl = smem + (VECPRE ? 0 : VEC_B) + wv * (ROWS * ROWB);
  const int lr = lane / CPR, lc = lane - lr * CPR;
  const int ncol = (GEGLU ? ((n0 + qn0) >> 1) : (n0 + qn0)) + lc * 8;
  const bool lane_on = lr < RPI && ncol < ldo;
  const int tid = wv * 64 + lane;

  // every global load first: the tile's bias / colsum chunk of this thread, the wave's residual chunks
  float4 vb = make_float4(0.f, 0.f, 0.f, 0.f), vc = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (!VECPRE) {
    const bool vec_lane = tid < BN / 4 && n0 + tid * 4 < N;
    if (vec_lane) {
      if (p.bias) vb = *(const float4*)(p.bias + n0 + tid * 4);
      if (SR_FOLD(p)) vc = *(const float4*)(p.colsum + n0 + tid * 4);
    }
  }
  uint4 resid[PRE ? NPASS * ITER : 1];
  if constexpr (PRE) {
    if (p.residual) {
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
          const int row = it * RPI + lr, m = m0 + pm0 + ps * ROWS + row;
          resid[ps * ITER + it] = make_uint4(0, 0, 0, 0);
          if (lane_on && row < ROWS && m < M) resid[ps * ITER + it] = *(const uint4*)((const _Float16*)p.residual + (int64_t)m * ldo + ncol);
        }
    }
  }
  uint4 rcur[ROLL ? ITER : 1], rnext[ROLL ? ITER : 1];
  auto roll_load = [&](uint4 (&dst)[ROLL ? ITER : 1], int ps) {
#pragma unroll
    for (int it = 0; it < (ROLL ? ITER : 1); ++it) {
      const int row = it * RPI + lr, m = m0 + pm0 + ps * ROWS + row;
      dst[it] = make_uint4(0, 0, 0, 0);
      if (lane_on && row < ROWS && m < M) dst[it] = *(const uint4*)((const _Float16*)p.residual + (int64_t)m * ldo + ncol);
    }
  };
  if constexpr (ROLL) { if (p.residual) roll_load(rcur, 0); }
  __syncthreads();                                           // every wave is done reading the staging tiles
  if constexpr (!VECPRE) {
    if (tid < BN / 4) { *(float4*)(lvec + tid * 4) = vb; *(float4*)(lvec + BN + tid * 4) = vc; }
    __syncthreads();
  }
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
    for (int t = 0; t < TMP; ++t) {
      const int tm = ps * TMP + t;
      const int m = m0 + pm0 + tm * 16 + c16;
      const int b = (p.rowvec && m < M) ? m / rpb : 0;
      float2 rs = make_float2(1.f, 0.f);                     // folded LayerNorm: (rstd, -rstd*mean) of input row m
      if (SR_FOLD(p) && m < M) rs = sr_fold_stat(p, m, m0);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int nl = qn0 + tn * 16 + 4 * g4, n = n0 + nl;
        float v[4] = {acc[tn][tm][0] * scale, acc[tn][tm][1] * scale, acc[tn][tm][2] * scale, acc[tn][tm][3] * scale};
        if (n < N) {                                         // N % 4 == 0 on this path
          if (SR_FOLD(p)) {
            const float4 cs = *(const float4*)(lvec + BN + nl);
            v[0] = fmaf(rs.x, v[0], rs.y * cs.x); v[1] = fmaf(rs.x, v[1], rs.y * cs.y);
            v[2] = fmaf(rs.x, v[2], rs.y * cs.z); v[3] = fmaf(rs.x, v[3], rs.y * cs.w);
          }
          if (p.bias) { const float4 bv = *(const float4*)(lvec + nl); v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w; }
          if (p.rowvec) { const float4 rv = *(const float4*)(p.rowvec + (int64_t)b * ldr + n); v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w; }
        }
        if (p.act == 1) { for (int r = 0; r < 4; ++r) v[r] = sr_silu_f(v[r]); }
        else if (p.act == 3) { for (int r = 0; r < 4; ++r) v[r] = sr_gelu_f(v[r]); }
        else if (p.act == 4) { for (int r = 0; r < 4; ++r) v[r] = fminf(fmaxf((v[r] + 1.0f) * 0.5f, 0.0f), 1.0f); }
        char* dst = wl + (t * 16 + c16) * ROWB;
        if constexpr (GEGLU) {
          const float o0 = v[0] * sr_gelu_f(v[1]), o1 = v[2] * sr_gelu_f(v[3]);
          h16x2 hv = {(_Float16)o0, (_Float16)o1};
          *(h16x2*)(dst + (tn * 8 + 2 * g4) * 2) = hv;
        } else {
          h16x4 hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
          *(h16x4*)(dst + (tn * 16 + 4 * g4) * 2) = hv;
        }
      }
    }
    // read back row-wise (same wave wrote it: no block barrier needed, only the LDS write->read ordering)
    __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    if constexpr (ROLL) { if (p.residual && ps + 1 < NPASS) roll_load(rnext, ps + 1); }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int row = it * RPI + lr, m = m0 + pm0 + ps * ROWS + row;
      if (!(lane_on && row < ROWS && m < M)) continue;
      uint4 raw = *(const uint4*)(wl + row * ROWB + lc * 16);
      const int64_t oi = (int64_t)m * ldo + ncol;
      if (p.residual) {
        const h16x8 rr = PRE ? __builtin_bit_cast(h16x8, resid[PRE ? ps * ITER + it : 0])
                       : ROLL ? __builtin_bit_cast(h16x8, rcur[ROLL ? it : 0]) : *(const h16x8*)((const _Float16*)p.residual + oi);
        h16x8 hv = __builtin_bit_cast(h16x8, raw);
#pragma unroll
        for (int e = 0; e < 8; ++e) hv[e] = (_Float16)((float)hv[e] + (float)rr[e]);
        raw = __builtin_bit_cast(uint4, hv);
      }
      *(uint4*)((_Float16*)p.out + oi) = raw;
    }
    if constexpr (ROLL) {
#pragma unroll
      for (int it = 0; it < ITER; ++it) rcur[it] = rnext[it];
    }
    __builtin_amdgcn_wave_barrier();                         // (LDS is in order per wave: the next pass may overwrite)
  }
}

// Row-major epilogue shared by the tile kernels: bias / time-embedding slice / activation in registers, the wave's sub-tile
// staged through LDS (`smem`, LDS_AVAIL bytes free for it) for 16-byte coalesced stores and residual loads.
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int LDS_AVAIL, bool BLOCK_SYNC = true, int VCAP = 256, bool VECPRE = false>
__device__ __forceinline__ void epilogue_rows(const sr_igemm_args& p, f32x4 (&acc)[BN / WAVES_N / 16][BM / WAVES_M / 16], char* smem, const int M,
                                              const int rpb, const int m0, const int n0, const int pm0, const int qn0, const int wv,
                                              const int lane) {
  constexpr int NW = WAVES_M * WAVES_N, TM = BM / WAVES_M / 16, TN = BN / WAVES_N / 16;
  const int c16 = lane & 15, g4 = lane >> 4;
  const float scale = p.scale;
  const int N = p.N;
  const int ldr = p.rowvec_ld ? p.rowvec_ld : N;
  {
    const int ldo = (p.act == 2) ? (N >> 1) : N;
    const bool o32 = p.out_f32 || sizeof(T) == 4;
    const int oes = o32 ? 4 : 2;                             // output element size
    if constexpr (sizeof(T) == 2) {
      if (!o32 && (ldo * 2) % 16 == 0) {
        static_assert(BLOCK_SYNC, "the fp16 epilogue synchronises the workgroup itself");
        if (p.act == 2) epilogue_rows_h<T, BM, BN, WAVES_M, WAVES_N, LDS_AVAIL, true, VCAP, VECPRE>(p, acc, smem, M, rpb, m0, n0, pm0, qn0, wv, lane);
        else            epilogue_rows_h<T, BM, BN, WAVES_M, WAVES_N, LDS_AVAIL, false, VCAP, VECPRE>(p, acc, smem, M, rpb, m0, n0, pm0, qn0, wv, lane);
        return;
      }
    }
    if ((ldo * oes) % 16 == 0) {
      // Coalesced path: bias / time-embedding / activation in registers, then the wave's sub-tile goes through LDS
      // (the staging buffers are free now) so that global stores -- and the residual loads -- are whole 16-byte
      // chunks of contiguous rows instead of 4..8-byte pieces scattered over 16 rows per instruction.
      constexpr int WCOLS = BN / WAVES_N;                    // columns of this wave's sub-tile before GEGLU
      // rows of the wave's sub-tile staged per pass: all of them when that fits the staging LDS, else 16 (the 256x320 tile)
      constexpr int TMP = (NW * (BM / WAVES_M) * (WCOLS * 4 + 16) <= LDS_AVAIL) ? TM : 1;
      const int ocols = (p.act == 2) ? WCOLS / 2 : WCOLS;
      const int rowb = ocols * oes + 16;                     // padded LDS row (bank spread, keeps 16-B alignment)
      if constexpr (BLOCK_SYNC) __syncthreads();             // every wave is done reading the staging tiles
      char* wl = smem + wv * ((TMP * 16) * rowb);
      const int cpr = ocols * oes / 16;                      // 16-B chunks per row
      const int rpi = 64 / cpr;                              // rows per wave instruction
      const int lr = lane / cpr, lc = lane - lr * cpr;
      const int ncol0 = (p.act == 2) ? ((n0 + qn0) >> 1) : (n0 + qn0);
      const int epc_o = 16 / oes;
#pragma unroll
     for (int tm0 = 0; tm0 < TM; tm0 += TMP) {
#pragma unroll
      for (int tm = tm0; tm < tm0 + TMP; ++tm) {
        const int m = m0 + pm0 + tm * 16 + c16;
        const int b = (m < M) ? m / rpb : 0;
        float2 rs = make_float2(1.f, 0.f);                   // folded LayerNorm: (rstd, -rstd*mean) of input row m
        if (SR_FOLD(p) && m < M) rs = sr_fold_stat(p, m, m0);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const int n = n0 + qn0 + tn * 16 + 4 * g4;
          float v[4] = {acc[tn][tm][0] * scale, acc[tn][tm][1] * scale, acc[tn][tm][2] * scale, acc[tn][tm][3] * scale};
          if (n < N) {                                       // N % 4 == 0 on this path
            if (SR_FOLD(p)) {
              const float4 cs = *(const float4*)(p.colsum + n);
              v[0] = fmaf(rs.x, v[0], rs.y * cs.x); v[1] = fmaf(rs.x, v[1], rs.y * cs.y);
              v[2] = fmaf(rs.x, v[2], rs.y * cs.z); v[3] = fmaf(rs.x, v[3], rs.y * cs.w);
            }
            if (p.bias) { const float4 bv = *(const float4*)(p.bias + n); v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w; }
            if (p.rowvec) { const float4 rv = *(const float4*)(p.rowvec + (int64_t)b * ldr + n); v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w; }
          }
          if (p.act == 1) { for (int r = 0; r < 4; ++r) v[r] = sr_silu_f(v[r]); }
          else if (p.act == 3) { for (int r = 0; r < 4; ++r) v[r] = sr_gelu_f(v[r]); }
          else if (p.act == 4) { for (int r = 0; r < 4; ++r) v[r] = fminf(fmaxf((v[r] + 1.0f) * 0.5f, 0.0f), 1.0f); }
          char* dst = wl + ((tm - tm0) * 16 + c16) * rowb;
          if (p.act == 2) {
            const float o0 = v[0] * sr_gelu_f(v[1]), o1 = v[2] * sr_gelu_f(v[3]);
            const int col = tn * 8 + 2 * g4;
            if (o32) *(float2*)(dst + col * 4) = make_float2(o0, o1);
            else { h16x2 hv = {(_Float16)o0, (_Float16)o1}; *(h16x2*)(dst + col * 2) = hv; }
          } else {
            const int col = tn * 16 + 4 * g4;
            if (o32) *(float4*)(dst + col * 4) = make_float4(v[0], v[1], v[2], v[3]);
            else { h16x4 hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]}; *(h16x4*)(dst + col * 2) = hv; }
          }
        }
      }
      // read back row-wise (same wave wrote it: no block barrier needed, only the LDS write->read ordering)
      __builtin_amdgcn_s_waitcnt(0xC07F);                    // lgkmcnt(0)
      __builtin_amdgcn_wave_barrier();
      for (int r0 = 0; r0 < TMP * 16; r0 += rpi) {
        const int row = r0 + lr;
        if (lr >= rpi || row >= TMP * 16) continue;
        const int m = m0 + pm0 + tm0 * 16 + row, ncol = ncol0 + lc * epc_o;
        if (m >= M || ncol >= ldo) continue;
        uint4 raw = *(const uint4*)(wl + row * rowb + lc * 16);
        const int64_t oi = (int64_t)m * ldo + ncol;
        if (p.residual) {
          if constexpr (sizeof(T) == 2) {
            if (o32) {                                     // fp32 out + fp16 residual: 4 values per 16-byte out chunk
              float4 f = __builtin_bit_cast(float4, raw);
              const h16x4 r4 = *(const h16x4*)((const _Float16*)p.residual + oi);
              f.x += (float)r4[0]; f.y += (float)r4[1]; f.z += (float)r4[2]; f.w += (float)r4[3];
              raw = __builtin_bit_cast(uint4, f);
            } else {
              const h16x8 rr = *(const h16x8*)((const _Float16*)p.residual + oi);
              h16x8 hv = __builtin_bit_cast(h16x8, raw);
#pragma unroll
              for (int e = 0; e < 8; ++e) hv[e] = (_Float16)((float)hv[e] + (float)rr[e]);
              raw = __builtin_bit_cast(uint4, hv);
            }
          } else {
            float4 f = __builtin_bit_cast(float4, raw);
            const float4 rr = *(const float4*)((const float*)p.residual + oi);
            f.x += rr.x; f.y += rr.y; f.z += rr.z; f.w += rr.w;
            raw = __builtin_bit_cast(uint4, f);
          }
        }
        *(uint4*)((char*)p.out + oi * oes) = raw;
      }
      __builtin_amdgcn_wave_barrier();                       // (LDS is in order per wave: the next pass may overwrite)
     }
      return;
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int m = m0 + pm0 + tm * 16 + c16;
      if (m >= M) continue;
      const int b = m / rpb;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + qn0 + tn * 16 + 4 * g4;
        if (n >= N) continue;
        float v[4] = {acc[tn][tm][0] * scale, acc[tn][tm][1] * scale, acc[tn][tm][2] * scale, acc[tn][tm][3] * scale};
        const int nv = (N - n) < 4 ? (N - n) : 4;
        if (SR_FOLD(p)) {
          const float2 rs = sr_fold_stat(p, m, m0);
          for (int r = 0; r < nv; ++r) v[r] = fmaf(rs.x, v[r], rs.y * p.colsum[n + r]);
        }
        if (p.bias) { for (int r = 0; r < nv; ++r) v[r] += p.bias[n + r]; }
        if (p.rowvec) { for (int r = 0; r < nv; ++r) v[r] += p.rowvec[(int64_t)b * ldr + n + r]; }
        if (p.act == 1) { for (int r = 0; r < 4; ++r) v[r] = sr_silu_f(v[r]); }
        else if (p.act == 3) { for (int r = 0; r < 4; ++r) v[r] = sr_gelu_f(v[r]); }
        else if (p.act == 4) { for (int r = 0; r < 4; ++r) v[r] = fminf(fmaxf((v[r] + 1.0f) * 0.5f, 0.0f), 1.0f); }
        if (p.act == 2) {                                // GEGLU: (value, gate) pairs interleaved along n
          float o0 = v[0] * sr_gelu_f(v[1]), o1 = v[2] * sr_gelu_f(v[3]);
          const int64_t oi = (int64_t)m * ldo + (n >> 1);
          if (p.residual) { o0 += sr_load_f((const T*)p.residual + oi); o1 += sr_load_f((const T*)p.residual + oi + 1); }
          if (p.out_f32) { float* o = (float*)p.out + oi; o[0] = o0; o[1] = o1; }
          else { T* o = (T*)p.out + oi; sr_store_f(o, o0); sr_store_f(o + 1, o1); }
          continue;
        }
        const int64_t oi = (int64_t)m * ldo + n;
        if (nv == 4) {
          if (p.residual) {
            if constexpr (sizeof(T) == 2) {
              const h16x4 rr = *(const h16x4*)((const T*)p.residual + oi);
              v[0] += (float)rr[0]; v[1] += (float)rr[1]; v[2] += (float)rr[2]; v[3] += (float)rr[3];
            } else {
              const float4 rr = *(const float4*)((const float*)p.residual + oi);
              v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
            }
          }
          if (p.out_f32 || sizeof(T) == 4) {
            *(float4*)((float*)p.out + oi) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
            h16x4 hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            *(h16x4*)((_Float16*)p.out + oi) = hv;
          }
        } else {
          for (int r = 0; r < nv; ++r) {
            float o = v[r];
            if (p.residual) o += sr_load_f((const T*)p.residual + oi + r);
            if (p.out_f32) ((float*)p.out)[oi + r] = o; else sr_store_f((T*)p.out + oi + r, o);
          }
        }
      }
    }
  }
}

// the drain of a STAGES-deep ring: `y` (wave-uniform, < D - 1) younger stages stay in flight -- s_waitcnt takes an immediate
template <int PS, int D>
__device__ __forceinline__ void wait_leaving(int y) {
#define SR_WL(k) if constexpr (D - 1 > k) { if (y == k) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PS * k) : "memory"); return; } }
  SR_WL(1) SR_WL(2) SR_WL(3) SR_WL(4) SR_WL(5) SR_WL(6)
#undef SR_WL
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// BKB = bytes of K per LDS stage: 128 (two MFMA k-substeps, 8 rows x 8 chunks per LDS-DMA instruction) or 64 (one k-substep,
// 16 rows x 4 chunks) -- the 64-byte form halves a stage so that the 256x320 tile gets a FOUR-deep ring in 144 KB (three
// K-steps of loads in flight instead of one: that tile is otherwise bound by the exposed load latency of every K-step).
// The kernel body: `bid` of `nblk` is the workgroup's index among the workgroups of THIS problem (blockIdx.x / gridDim.x for an
// ordinary launch; a grouped launch runs several problems side by side, see igemm_group_kernel)
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, bool TRANS, bool SPLIT = false, int BKB = 128, int SPREAD = 0, int MINB = 1, bool LNI = false>
__device__ __forceinline__ void igemm_body(const sr_igemm_args& p, const int M, const int Ho, const int Wo,
                                           const int NT, const int nwg, const int tile0, const int bid, const int nblk) {
  extern __shared__ __attribute__((aligned(16))) char smem_all[];
  // [256 B per wave: landing zone of the weight prefetch (sr_igemm_args.prefetch), never read] [bias / colsum rows] [ring]
  constexpr int PF_B = WAVES_M * WAVES_N * 256;
  char* const smem_raw = smem_all + PF_B;
  // fp16, unsplit: the tile's bias and colsum rows (2 x BN floats) sit in front of the staging ring, fetched by LDS-DMA below
  constexpr bool VECPRE = sizeof(T) == 2 && !SPLIT;
  constexpr int VEC_B = VECPRE ? 2 * BN * 4 : 0;
  char* const smem = smem_raw + VEC_B;
  constexpr int KE = BKB / (int)sizeof(T);            // elements of K per step
  constexpr int NW = WAVES_M * WAVES_N;               // waves per workgroup (4 or 8)
  constexpr int RPI = 1024 / BKB;                     // tile rows one LDS-DMA wave instruction fills (1 KiB)
  constexpr int GP = BM / RPI, GQ = BN / RPI;         // row groups of the X / W tiles
  constexpr int NIP = (GP + NW - 1) / NW, NIQ = (GQ + NW - 1) / NW;   // glds instructions per wave per K-step; when the
                                                      // groups do not split evenly the surplus instructions re-fetch the last group
  constexpr int TM = BM / WAVES_M / 16, TN = BN / WAVES_N / 16;
  constexpr int STAGE_BYTES = (BM + BN) * BKB;
  constexpr int FBLK = 16 * BKB;                      // bytes of one 16-row fragment block
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  static_assert(BKB == 128 || BKB == 64, "K bytes per stage");
  static_assert(BM % RPI == 0 && BN % RPI == 0, "tile rows must be whole LDS-DMA groups");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#if SR_IGEMM_TRACE
  unsigned long long ts_[6] = {0, 0, 0, 0, 0, 0};
#endif
  SR_TS(0);
  if (p.prefetch) {
    const int64_t nthr = (int64_t)nblk * gridDim.y * (NW * 64);
    sr_prefetch_touch(p.prefetch, p.prefetch_bytes, ((int64_t)blockIdx.y * nblk + bid) * (NW * 64) + tid, nthr,
                      __builtin_amdgcn_readfirstlane(sr_lds_addr(smem_all) + wv * 256));
  }
  const int wg = tile0 + sr_xcd_remap(bid, nwg);             // this launch covers tiles [tile0, tile0 + nwg)
  int mt, nt;
  sr_tile_of(wg, NT, (M + BM - 1) / BM, p.tile_order, mt, nt);
  const int m0 = mt * BM, n0 = nt * BN;
  if constexpr (VECPRE) {
    // [bias n0..n0+BN) | colsum n0..n0+BN)] -> LDS, 16 bytes per lane; absent rows / columns past N come from the zero page.
    // These are the oldest LDS-DMA requests of the wave: the first counted vmcnt wait of the K loop covers them, its barrier
    // publishes them; the epilogue reads them with ds_read (no global load behind a store there: see epilogue_rows_h)
    constexpr int CH = BN / 2;                               // 16-byte chunks in all
    const int c = wv * 64 + lane;
    if (wv * 64 < CH) {
      const bool second = c >= BN / 4;
      const int n = n0 + 4 * (second ? c - BN / 4 : c);
      const float* base = second ? (SR_FOLD(p) ? p.colsum : nullptr) : p.bias;
      const char* src = (base && n < p.N) ? (const char*)(base + n) : (const char*)p.zero_page;
      if (c < CH) sr_glds16_asm(src, __builtin_amdgcn_readfirstlane(sr_lds_addr(smem_raw) + wv * 1024));
    }
  }
  // LDS-DMA writes lane-linearly (row = lane / chunks-per-row, slot = lane % chunks-per-row), so the bank swizzle is applied to
  // the SOURCE: the lane fetches the logical 16-B chunk whose swizzled slot it fills.  128-B rows: slot = chunk ^ (row & 7).
  // 64-B rows (rows r and r+4 share banks): slot = ((row >> 2) & 3) ^ F[chunk], F = {0,3,1,2}, which makes every 16-lane
  // group of ds_read_b128 ({0-3,12-15,20-27}, ...) hit 16 distinct (row & 3, slot) pairs = all 64 banks once.
  const int lrow = BKB == 128 ? lane >> 3 : lane >> 2;
  int lchunk;
  if constexpr (BKB == 128) lchunk = (lane & 7) ^ lrow;
  else { const int q = (lane & 3) ^ ((lrow >> 2) & 3); lchunk = (0x1320 >> (4 * q)) & 3; }   // F^-1 = {0,2,3,1}

  const int C1 = p.C1, C2 = p.C2, Ctot = C1 + C2;
  const int K1 = C1 / KE, KPT = Ctot / KE;             // K-steps from source a / per tap
  const int ntaps = p.KH * p.KH;
  const int KT = ntaps * KPT;
  const int H = p.H, W = p.W, pad = p.pad_br ? 0 : (p.KH >> 1);
  const int rpb = Ho * Wo;

  // ---- per-lane row bookkeeping for the X (pixel) tile
  int pb[NIP], py[NIP], px[NIP];
#pragma unroll
  for (int i = 0; i < NIP; ++i) {
    const int gi = (i * NW + wv) < GP ? (i * NW + wv) : GP - 1;
    const int m = m0 + gi * RPI + lrow;
    if (m < M) {
      if (rpb == 1) { pb[i] = m; py[i] = -pad; px[i] = -pad; }      // linear layers (one "pixel" per row): no divisions
      else {
        const int b = m / rpb, rem = m - b * rpb, oy = rem / Wo;
        pb[i] = b; py[i] = oy * p.stride - pad; px[i] = (rem - oy * Wo) * p.stride - pad;
      }
    } else { pb[i] = -1; py[i] = 0; px[i] = 0; }
  }
  const char* zp = (const char*)p.zero_page + lchunk * 16;
  const char* rowA[NIP];
  const char* rowB[NIP];
  auto set_tap = [&](int tap) {
    const int ky = tap / p.KH, kx = tap - ky * p.KH;
#pragma unroll
    for (int i = 0; i < NIP; ++i) {
      int iy = py[i] + ky, ix = px[i] + kx;
      bool ok = pb[i] >= 0;
      if (p.upsample) {
        // nearest upsample fused into the gather: x2, or to an arbitrary (up_h, up_w) with torch's rule src = floor(dst * in/out)
        // (fp32 scale) -- Upsample.forward(x, output_shape) interpolates to the skip tensor's size (openaimodel.py:109-121)
        if (p.up_h > 0) {
          ok = ok && iy >= 0 && ix >= 0 && iy < p.up_h && ix < p.up_w;
          iy = min((int)floorf((float)iy * ((float)H / (float)p.up_h)), H - 1);
          ix = min((int)floorf((float)ix * ((float)W / (float)p.up_w)), W - 1);
          if (!ok) { iy = 0; ix = 0; }
        } else { ok = ok && iy >= 0 && ix >= 0 && iy < 2 * H && ix < 2 * W; iy >>= 1; ix >>= 1; }
      }
      else            { ok = ok && iy >= 0 && ix >= 0 && iy < H && ix < W; }
      const int64_t pix = ((int64_t)pb[i] * H + iy) * W + ix;
      rowA[i] = ok ? (const char*)p.a + (pix * C1) * (int64_t)sizeof(T) + lchunk * 16 : zp;
      rowB[i] = (ok && C2 > 0) ? (const char*)p.a2 + (pix * C2) * (int64_t)sizeof(T) + lchunk * 16 : zp;
    }
  };
  // ---- W tile rows
  const char* wrow[NIQ];
#pragma unroll
  for (int i = 0; i < NIQ; ++i) {
    const int gi = (i * NW + wv) < GQ ? (i * NW + wv) : GQ - 1;
    const int n = n0 + gi * RPI + lrow;
    wrow[i] = (const char*)p.w + ((int64_t)n * KT * KE) * (int64_t)sizeof(T) + lchunk * 16;
  }

  // split-K: blockIdx.y owns the K-steps [kt0, kt1) and writes raw fp32 partials to the workspace
  int kt0 = 0, kt1 = KT;
  if constexpr (SPLIT) {
    kt0 = (int)((int64_t)KT * blockIdx.y / gridDim.y);
    kt1 = (int)((int64_t)KT * (blockIdx.y + 1) / gridDim.y);
#pragma unroll
    for (int i = 0; i < NIQ; ++i) wrow[i] += (int64_t)kt0 * BKB;
  }
  int s_tap = kt0 / KPT, s_kk = kt0 - s_tap * KPT;
  // (measured and dropped: starting every tile's K loop at a different K-step, so that concurrent tiles do not ask for the same
  // weight / activation lines at the same moment: neutral to -10 % on the K-short 1x1 layers -- they are not bound by hot lines)
  auto advance = [&]() { if (++s_kk == KPT) { s_kk = 0; if (++s_tap < ntaps) set_tap(s_tap); } };
  set_tap(s_tap);
  auto stage = [&](int buf) {                            // issue the async loads of the next K-step
    char* tP = smem + buf * STAGE_BYTES;
    char* tQ = tP + BM * BKB;
    const bool fromA = s_kk < K1;
    const int off = (fromA ? s_kk : s_kk - K1) * BKB;
    auto gp = [&](int i) { return ((i * NW + wv) < GP ? (i * NW + wv) : GP - 1) * 1024; };   // LDS offset of the row group
    auto gq = [&](int i) { return ((i * NW + wv) < GQ ? (i * NW + wv) : GQ - 1) * 1024; };
    {
      const unsigned lP = __builtin_amdgcn_readfirstlane(sr_lds_addr(tP)), lQ = __builtin_amdgcn_readfirstlane(sr_lds_addr(tQ));
      const unsigned m0 = sr_m0_save();
#pragma unroll
      for (int i = 0; i < NIP; ++i) sr_glds16_asm_nosave((fromA ? rowA[i] : rowB[i]) + off, lP + gp(i));
#pragma unroll
      for (int i = 0; i < NIQ; ++i) { sr_glds16_asm_nosave(wrow[i], lQ + gq(i)); wrow[i] += BKB; }
      sr_m0_restore(m0);
    }
    advance();
  };

  // ---- fragment read offsets
  const int c16 = lane & 15, g4 = lane >> 4;
  const int wm = wv / WAVES_N, wn = wv - wm * WAVES_N;
  const int pm0 = wm * (BM / WAVES_M), qn0 = wn * (BN / WAVES_N);
  int foff[2];
  if constexpr (BKB == 128) {
    foff[0] = c16 * 128 + (((0 + g4) ^ (c16 & 7)) << 4);
    foff[1] = c16 * 128 + (((4 + g4) ^ (c16 & 7)) << 4);
  } else {
    foff[0] = foff[1] = c16 * 64 + ((((c16 >> 2) & 3) ^ ((0x2130 >> (4 * g4)) & 3)) << 4);        // F = {0,3,1,2}
  }

  constexpr int TA = TRANS ? TM : TN, TB = TRANS ? TN : TM;
  f32x4 acc[TA][TB];
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- in-launch LayerNorm statistics (sr_igemm_args.ln_inline, kernels instantiated with LNI): K = C1 is the whole normalised row
  // and every K-step of it passes through this workgroup, so x and x^2 are summed on the way (fp32 accumulators, v_dot2_f32_f16).
  // The WAVES_N waves that share a row block split its TM fragments (wave wn takes tm = wn, wn + WAVES_N, ...): each reads ITS
  // fragments once more from LDS at a wave-uniform offset -- a compile-time register index would need a branch per fragment, and
  // branches in this loop cost the 128- and 256-VGPR tiles their last registers -- i.e. TM / WAVES_N extra ds_read_b128 and 8 VALU
  // instructions per fragment and k-substep beside 10..40 MFMAs.
  constexpr int TMS = LNI ? TM / WAVES_N : 1;
  static_assert(!LNI || (TM % WAVES_N == 0 && !SPLIT), "in-launch LayerNorm statistics: fragments split evenly over the waves of a row block");
  float lsum[TMS], lsq[TMS];
#pragma unroll
  for (int i = 0; i < TMS; ++i) { lsum[i] = 0.f; lsq[i] = 0.f; }
  auto ln_acc = [&](const char* tP, const int fo) {          // tP: this wave's row block of the X stage
#pragma unroll
    for (int i = 0; i < TMS; ++i) {
      const uint4 x = *(const uint4*)(tP + (wn + i * WAVES_N) * FBLK + fo);
      if constexpr (sizeof(T) == 2) {
        const h16x2 one = {(_Float16)1.0f, (_Float16)1.0f};
        const h16x2 h0 = __builtin_bit_cast(h16x2, x.x), h1 = __builtin_bit_cast(h16x2, x.y);
        const h16x2 h2 = __builtin_bit_cast(h16x2, x.z), h3 = __builtin_bit_cast(h16x2, x.w);
        lsum[i] = __builtin_amdgcn_fdot2(h0, one, lsum[i], false); lsq[i] = __builtin_amdgcn_fdot2(h0, h0, lsq[i], false);
        lsum[i] = __builtin_amdgcn_fdot2(h1, one, lsum[i], false); lsq[i] = __builtin_amdgcn_fdot2(h1, h1, lsq[i], false);
        lsum[i] = __builtin_amdgcn_fdot2(h2, one, lsum[i], false); lsq[i] = __builtin_amdgcn_fdot2(h2, h2, lsq[i], false);
        lsum[i] = __builtin_amdgcn_fdot2(h3, one, lsum[i], false); lsq[i] = __builtin_amdgcn_fdot2(h3, h3, lsq[i], false);
      } else {
        const float f0 = __uint_as_float(x.x), f1 = __uint_as_float(x.y), f2 = __uint_as_float(x.z), f3 = __uint_as_float(x.w);
        lsum[i] += f0; lsum[i] += f1; lsum[i] += f2; lsum[i] += f3;
        lsq[i] = fmaf(f0, f0, lsq[i]); lsq[i] = fmaf(f1, f1, lsq[i]); lsq[i] = fmaf(f2, f2, lsq[i]); lsq[i] = fmaf(f3, f3, lsq[i]);
      }
    }
  };

  auto compute = [&](int buf, int j0 = 0, int j1 = BKB / 64) {
    const char* tP = smem + buf * STAGE_BYTES + pm0 * BKB;
    const char* tQ = smem + buf * STAGE_BYTES + BM * BKB + qn0 * BKB;
#pragma unroll
    for (int j = j0; j < j1; ++j) {
      uint4 xf[TM], wf[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) xf[t] = *(const uint4*)(tP + t * FBLK + foff[j]);
#pragma unroll
      for (int t = 0; t < TN; ++t) wf[t] = *(const uint4*)(tQ + t * FBLK + foff[j]);
      if constexpr (LNI) ln_acc(tP, foff[j]);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          if constexpr (TRANS) sr_mma(acc[tm][tn], xf[tm], wf[tn], T());
          else                 sr_mma(acc[tn][tm], wf[tn], xf[tm], T());
        }
    }
  };
  // compute(buf) with the LDS-DMA of a later K-step (ring slot nbuf) issued piecewise BETWEEN the MFMA groups of the first
  // k-substep instead of in one burst after the barrier: a wave that issues 6..9 LDS-DMA instructions back to back sits
  // ~100 cycles on each while its SIMD's matrix pipe idles (both waves of a SIMD do this at the same moment).
  auto compute_staging = [&](int buf, int nbuf) {
    const char* tP = smem + buf * STAGE_BYTES + pm0 * BKB;
    const char* tQ = smem + buf * STAGE_BYTES + BM * BKB + qn0 * BKB;
    char* nP = smem + nbuf * STAGE_BYTES;
    const unsigned lP = __builtin_amdgcn_readfirstlane(sr_lds_addr(nP)), lQ = __builtin_amdgcn_readfirstlane(sr_lds_addr(nP + BM * BKB));
    const bool fromA = s_kk < K1;
    const int off = (fromA ? s_kk : s_kk - K1) * BKB;
    constexpr int PIECES = NIP + NIQ, PPS = (PIECES + TN - 1) / TN;       // LDS-DMA pieces per MFMA group
    auto piece = [&](int i) {
      if (i < NIP) {
        const int g = ((i * NW + wv) < GP ? (i * NW + wv) : GP - 1) * 1024;
        sr_glds16_asm_nosave((fromA ? rowA[i] : rowB[i]) + off, lP + g);
      } else if (i < PIECES) {
        const int q = i - NIP;
        const int g = ((q * NW + wv) < GQ ? (q * NW + wv) : GQ - 1) * 1024;
        sr_glds16_asm_nosave(wrow[q], lQ + g);
        wrow[q] += BKB;
      }
    };
    const unsigned m0_keep = sr_m0_save();                   // (MFMA / ds_read between the pieces do not use M0)
#pragma unroll
    for (int j = 0; j < BKB / 64; ++j) {
      uint4 xf[TM], wf[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) xf[t] = *(const uint4*)(tP + t * FBLK + foff[j]);
#pragma unroll
      for (int t = 0; t < TN; ++t) wf[t] = *(const uint4*)(tQ + t * FBLK + foff[j]);
      if constexpr (LNI) ln_acc(tP, foff[j]);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          if constexpr (TRANS) sr_mma(acc[tm][tn], xf[tm], wf[tn], T());
          else                 sr_mma(acc[tn][tm], wf[tn], xf[tm], T());
        }
        if (j == 0) {
#pragma unroll
          for (int q = 0; q < PPS; ++q) piece(tn * PPS + q);
        }
      }
    }
    sr_m0_restore(m0_keep);
    advance();
  };
  {
    // STAGES-deep LDS ring, loads run D = STAGES-1 K-steps ahead: wait only for the oldest stage (counted vmcnt, never 0
    // while a younger stage is in flight) and synchronise with a raw s_barrier so the in-flight LDS-DMA is not drained
    constexpr int PER_STAGE = NIP + NIQ;                 // glds instructions one wave issues per stage (same for every wave)
    // With two slots (D = 1) this is the plain double buffer, but the LDS-DMA MUST be issued from inline asm: for the
    // __builtin_amdgcn_global_load_lds form hipcc puts s_waitcnt vmcnt(0) in front of the first ds_read of compute(),
    // i.e. it drains the prefetch it has just issued and every K-step pays the full load latency (found in the ISA of the
    // 2-slot kernels; SQ_WAIT_ANY was 49 % of the wave cycles).  asm-issued: 64x64 C320 3x3 conv 146 -> 133 us, big conv
    // 1094 -> 1186 TF/s, 64x64 GEGLU 265 -> 251 us, and every 128-wide tile likewise.
    constexpr int D = STAGES - 1;
    static_assert(STAGES >= 2 && STAGES <= 8, "ring depth");
    static_assert(PER_STAGE * (D - 1) < 64, "vmcnt immediate");
    const int nk = kt1 - kt0;                            // (split-K: this workgroup's share of the K-steps)
    SR_TS(1);
#pragma unroll
    for (int i = 0; i < D; ++i) if (i < nk) stage(i);
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
      // the asm-issued LDS-DMA is invisible to hipcc's wait-count pass, so these counted waits are the only ordering:
      // leave the min(D-1, remaining) younger stages in flight
      const int younger = nk - 1 - kt;
      if (younger >= D - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STAGE * (D - 1)) : "memory");
      else                  wait_leaving<PER_STAGE, D>(younger);
      __builtin_amdgcn_s_barrier();                      // stage kt visible to all; everyone finished step kt-1
      if (kt == 0) SR_TS(2);
      if constexpr (SPREAD == 1) {
        if (kt + D < nk) { int nb = cur + D; if (nb >= STAGES) nb -= STAGES; compute_staging(cur, nb); }
        else compute(cur);
      } else if constexpr (SPREAD == 2 && BKB == 128) {
        // the matrix pipe gets the first k-substep right after the barrier; the LDS-DMA burst (~150 scalar / address
        // instructions for 9 pieces) is issued in its shadow
        compute(cur, 0, 1);
        if (kt + D < nk) { int nb = cur + D; if (nb >= STAGES) nb -= STAGES; stage(nb); }
        compute(cur, 1, 2);
      } else {
        if (kt + D < nk) { int nb = cur + D; if (nb >= STAGES) nb -= STAGES; stage(nb); }
        compute(cur);
      }
      if (++cur == STAGES) cur = 0;
    }
  }

  SR_TS(3);
  if constexpr (LNI) {
    // a lane holds the sums of its 8 (4) k-slots per K-step: add the four k-slot lane groups, finish (rstd, -rstd * mean) and leave
    // the pair of row pm0 + tm * 16 + c16 at the front of the LDS for the epilogue (sr_fold_stat)
    const float invk = 1.0f / (float)p.C1;
#pragma unroll
    for (int i = 0; i < TMS; ++i) {
      float su = lsum[i], sq = lsq[i];
      su += __shfl_xor(su, 16); sq += __shfl_xor(sq, 16);
      su += __shfl_xor(su, 32); sq += __shfl_xor(sq, 32);
      const int tm = wn + i * WAVES_N;
      if (g4 == 0) {
        const float mean = su * invk, var = fmaxf(fmaf(-mean, mean, sq * invk), 0.f);
        const float rstd = rsqrtf(var + p.ln_eps);
        *(float2*)(smem_all + 8 * (pm0 + tm * 16 + c16)) = make_float2(rstd, -rstd * mean);
      }
    }
    __syncthreads();
  }
  // ---- epilogue
  const float scale = p.scale;
  const int N = p.N;
  const int ldr = p.rowvec_ld ? p.rowvec_ld : N;
  if constexpr (SPLIT) {
    // fragment-major partials: ws[z][tile][wave][tn][tm][lane] (float4) -> every store instruction writes 1 KiB contiguous
    f32x4* ws = (f32x4*)p.workspace + (((int64_t)blockIdx.y * nwg + (wg - tile0)) * NW + wv) * (TN * TM * 64) + lane;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) ws[(tn * TM + tm) * 64] = acc[tn][tm];
    if (p.split_counters == nullptr) return;                 // two-launch form: splitk_reduce_kernel finishes the tile
    // ---- fix-up inside the GEMM (sr_igemm_args.split_counters): every workgroup publishes its partial (agent-scope release),
    // counts itself in, and the LAST of the tile's gridDim.y workgroups to arrive sums all partials in z order -- its own is read
    // back like the others, so the bits do not depend on who is last -- and runs the ordinary epilogue.  Nobody waits for anybody;
    // the last arriver re-zeroes the counter for the next launch.  Saves the reduce launch (5-11 us each, 30 per B = 16 UNet
    // evaluation, 67 at B = 2) and the partials' second trip through a cold kernel.
    __threadfence();
    __syncthreads();                                         // all partial stores of the workgroup are published; the ring is dead
    int* const flag = (int*)smem_all;
    if (tid == 0) {
      const int prev = __hip_atomic_fetch_add(p.split_counters + (wg - tile0), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *flag = prev == (int)gridDim.y - 1;
    }
    __syncthreads();
    if (!*flag) return;
    __threadfence();                                         // acquire: the other workgroups' partials
    {
      const f32x4* wr = (const f32x4*)p.workspace + ((int64_t)(wg - tile0) * NW + wv) * (TN * TM * 64) + lane;
      const int64_t zs = (int64_t)nwg * NW * (TN * TM * 64);
      const int S = gridDim.y;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) acc[tn][tm] = wr[(tn * TM + tm) * 64];
      for (int z = 1; z < S; ++z) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int tm = 0; tm < TM; ++tm) acc[tn][tm] += wr[z * zs + (tn * TM + tm) * 64];
      }
      if (tid == 0) p.split_counters[wg - tile0] = 0;
    }
    epilogue_rows<T, BM, BN, WAVES_M, WAVES_N, STAGES * STAGE_BYTES, true, (MINB >= 4 ? 128 : 256), VECPRE>(p, acc, smem, M, rpb, m0, n0, pm0, qn0, wv, lane);
    return;
  } else if constexpr (!TRANS) {
    epilogue_rows<T, BM, BN, WAVES_M, WAVES_N, STAGES * STAGE_BYTES, true, (MINB >= 4 ? 128 : 256), VECPRE>(p, acc, smem, M, rpb, m0, n0, pm0, qn0, wv, lane);
#if SR_IGEMM_TRACE
    SR_TS(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SR_TS(5);
    if (threadIdx.x == 0 && p.workspace && !SPLIT) { unsigned long long* w_ = (unsigned long long*)p.workspace + (size_t)bid * 8; for (int k_ = 0; k_ < 6; ++k_) w_[k_] = ts_[k_]; w_[6] = __smid(); }
#endif
  } else {
    const bool vec = (rpb % 4 == 0) && (p.ldt % 4 == 0);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const int m = m0 + pm0 + tm * 16 + 4 * g4;
      if (m >= M) continue;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + qn0 + tn * 16 + c16;
        if (n >= N) continue;
        const float bz = p.bias ? p.bias[n] : 0.f;
        float v[4] = {acc[tm][tn][0] * scale, acc[tm][tn][1] * scale, acc[tm][tn][2] * scale, acc[tm][tn][3] * scale};
        if (SR_FOLD(p)) {                                  // a lane holds 4 consecutive input rows of one output channel
          const float cs = p.colsum[n];
          for (int r = 0; r < 4; ++r)
            if (m + r < M) { const float2 rs = sr_fold_stat(p, m + r, m0); v[r] = fmaf(rs.x, v[r], rs.y * cs); }
        }
        for (int r = 0; r < 4; ++r) v[r] += bz;
        if (vec) {
          const int b = m / rpb, t = m - b * rpb;
          const int64_t oi = ((int64_t)b * N + n) * p.ldt + t;
          if (p.out_f32 || sizeof(T) == 4) *(float4*)((float*)p.out + oi) = make_float4(v[0], v[1], v[2], v[3]);
          else { h16x4 hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]}; *(h16x4*)((_Float16*)p.out + oi) = hv; }
        } else {
          for (int r = 0; r < 4; ++r) {
            const int mm = m + r;
            if (mm >= M) break;
            const int b = mm / rpb, t = mm - b * rpb;
            const int64_t oi = ((int64_t)b * N + n) * p.ldt + t;
            if (p.out_f32) ((float*)p.out)[oi] = v[r]; else sr_store_f((T*)p.out + oi, v[r]);
          }
        }
      }
    }
  }
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, bool TRANS, bool SPLIT = false, int BKB = 128, int SPREAD = 0, int MINB = 1, bool LNI = false>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, MINB) void igemm_kernel(const sr_igemm_args p, const int M, const int Ho, const int Wo,
                                                                        const int NT, const int nwg, const int tile0) {
  igemm_body<T, BM, BN, WAVES_M, WAVES_N, STAGES, TRANS, SPLIT, BKB, SPREAD, MINB, LNI>(p, M, Ho, Wo, NT, nwg, tile0, blockIdx.x, gridDim.x);
}

// Several INDEPENDENT problems in one launch (sr_igemm_group): the workgroups of problem i are the blocks [start[i], start[i+1]).
// Why: at small batches (one view per GPU of a shard: B = 2) almost every UNet kernel is a latency chain on a fraction of the
// CUs -- launch, first stage, a short K loop, epilogue -- and the kernels of one stream run strictly one after the other; two
// parallel hipGraph branches cost more in fork / join than they return (measured: 6.62 vs 6.41 ms per B = 2 evaluation).  A block
// range per problem gives the same concurrency inside ONE dispatch: Q, K and V^T projections of a transformer block (three launches
// of 8-15 us each) run as one, a ResBlock's skip convolution beside its first 3x3 convolution.
struct igemm_group_k {
  sr_igemm_args p[SR_IGEMM_GROUP_MAX];
  int M[SR_IGEMM_GROUP_MAX], Ho[SR_IGEMM_GROUP_MAX], Wo[SR_IGEMM_GROUP_MAX], NT[SR_IGEMM_GROUP_MAX], nwg[SR_IGEMM_GROUP_MAX];
  int start[SR_IGEMM_GROUP_MAX + 1];
  int n;
};
// (LNI: the variant that takes LayerNorm statistics inside the launch for the members that ask for them -- sr_igemm_args.ln_inline;
//  the statistics are summed for every member, applied where the member's flag is set)
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, int BKB = 128, int SPREAD = 0, int MINB = 1, bool LNI = false>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, MINB) void igemm_group_kernel(const igemm_group_k g) {
  int i = 0;
#pragma unroll
  for (int j = 1; j < SR_IGEMM_GROUP_MAX; ++j) if (j < g.n && (int)blockIdx.x >= g.start[j]) i = j;
  igemm_body<T, BM, BN, WAVES_M, WAVES_N, STAGES, false, false, BKB, SPREAD, MINB, LNI>(g.p[i], g.M[i], g.Ho[i], g.Wo[i], g.NT[i], g.nwg[i], 0,
                                                                                          (int)blockIdx.x - g.start[i], g.nwg[i]);
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, int BKB = 128, int SPREAD = 0, int MINB = 1, bool LNI = false>
int launch_group_impl(const sr_igemm_args* const* as, const int* Ms, const int* Hos, const int* Wos, int n, hipStream_t st) {
  igemm_group_k g;
  g.n = n;
  int tot = 0;
  for (int i = 0; i < SR_IGEMM_GROUP_MAX; ++i) {
    const int k = i < n ? i : n - 1;
    g.p[i] = *as[k];
    g.p[i].prefetch = nullptr;
    g.M[i] = Ms[k]; g.Ho[i] = Hos[k]; g.Wo[i] = Wos[k];
    g.NT[i] = (as[k]->N + BN - 1) / BN;
    g.nwg[i] = ((Ms[k] + BM - 1) / BM) * g.NT[i];
    g.start[i] = tot;
    if (i < n) tot += g.nwg[i];
  }
  g.start[SR_IGEMM_GROUP_MAX] = tot;
  constexpr int lds_stage = STAGES * (BM + BN) * BKB;
  constexpr int lds_epi_all = WAVES_M * WAVES_N * (BM / WAVES_M) * ((BN / WAVES_N) * 4 + 16);
  constexpr int lds_epi = lds_epi_all <= lds_stage ? lds_epi_all : WAVES_M * WAVES_N * 16 * ((BN / WAVES_N) * 4 + 16);
  constexpr int lds = (lds_stage > lds_epi ? lds_stage : lds_epi) + (sizeof(T) == 2 ? 2 * BN * 4 : 0) + WAVES_M * WAVES_N * 256;
  auto k = igemm_group_kernel<T, BM, BN, WAVES_M, WAVES_N, STAGES, BKB, SPREAD, MINB, LNI>;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_set = true; }
  hipLaunchKernelGGL(k, dim3(tot), dim3(WAVES_M * WAVES_N * 64), lds, st, g);
  SR_CHECK_LAUNCH("sr_igemm_group");
  return SR_OK;
}
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, int BKB = 128, int SPREAD = 0, int MINB = 1>
int launch_group(const sr_igemm_args* const* as, const int* Ms, const int* Hos, const int* Wos, int n, hipStream_t st) {
  bool lni = false;
  for (int i = 0; i < n; ++i) lni = lni || as[i]->ln_inline;
  if (lni) return launch_group_impl<T, BM, BN, WAVES_M, WAVES_N, STAGES, BKB, SPREAD, MINB, true>(as, Ms, Hos, Wos, n, st);
  return launch_group_impl<T, BM, BN, WAVES_M, WAVES_N, STAGES, BKB, SPREAD, MINB, false>(as, Ms, Hos, Wos, n, st);
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, bool TRANS, int BKB, int SPREAD, int MINB, bool LNI>
int launch_impl(const sr_igemm_args& a, int M, int Ho, int Wo, hipStream_t st) {
  const int Npad = (a.N + 127) / 128 * 128;
  const int MT = (M + BM - 1) / BM, NT = Npad / BN;
  // skip all-padding n-tiles
  const int NTv = (a.N + BN - 1) / BN;
  const int nwg = MT * NTv;
  constexpr int lds_stage = STAGES * (BM + BN) * BKB;
  constexpr int lds_epi_all = WAVES_M * WAVES_N * (BM / WAVES_M) * ((BN / WAVES_N) * 4 + 16);   // fp32 output sub-tiles of all waves
  constexpr int lds_epi = lds_epi_all <= lds_stage ? lds_epi_all : WAVES_M * WAVES_N * 16 * ((BN / WAVES_N) * 4 + 16);
  constexpr int lds = (lds_stage > lds_epi ? lds_stage : lds_epi) + (sizeof(T) == 2 ? 2 * BN * 4 : 0)    // + the bias / colsum rows (fp16 kernels)
                      + WAVES_M * WAVES_N * 256;                                                           // + the prefetch landing zone
  auto k = igemm_kernel<T, BM, BN, WAVES_M, WAVES_N, STAGES, TRANS, false, BKB, SPREAD, MINB, LNI>;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_set = true; }
  (void)NT;
  hipLaunchKernelGGL(k, dim3(nwg), dim3(WAVES_M * WAVES_N * 64), lds, st, a, M, Ho, Wo, NTv, nwg, 0);
  SR_CHECK_LAUNCH("sr_igemm");
  return SR_OK;
}
// tiles without a variant that takes the LayerNorm statistics in the launch (sr_igemm_args.ln_inline): the tuner skips them
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, bool TRANS, int BKB = 128, int SPREAD = 0, int MINB = 1>
int launch(const sr_igemm_args& a, int M, int Ho, int Wo, hipStream_t st) {
  if (a.ln_inline) SR_FAIL(SR_ERR_INVALID, "sr_igemm: ln_inline is not built for this tile (%d x %d, %d stages)", BM, BN, STAGES);
  return launch_impl<T, BM, BN, WAVES_M, WAVES_N, STAGES, TRANS, BKB, SPREAD, MINB, false>(a, M, Ho, Wo, st);
}
// ... and the tiles that have one: 2, 3, 4, 13 (fp16 / fp32, row-major and transposed), 5, 7, 9, 10, 11, 12 (fp16)
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, bool TRANS, int BKB = 128, int SPREAD = 0, int MINB = 1>
int launch_ln(const sr_igemm_args& a, int M, int Ho, int Wo, hipStream_t st) {
  if (a.ln_inline) return launch_impl<T, BM, BN, WAVES_M, WAVES_N, STAGES, TRANS, BKB, SPREAD, MINB, true>(a, M, Ho, Wo, st);
  return launch_impl<T, BM, BN, WAVES_M, WAVES_N, STAGES, TRANS, BKB, SPREAD, MINB, false>(a, M, Ho, Wo, st);
}

// (A persistent "K-step stream" form of the 256x320 tile -- one workgroup per CU walking its tiles through the two LDS
//  buffers, the next tile's first K-step prefetched under the epilogue -- was built and measured for the K-short GEGLU layers:
//  parity-clean but slower, 383-431 vs 276 us at M65536 K320 N2560: keeping the next tile's bookkeeping alive beside 160
//  accumulators and the epilogue temporaries spills 276-412 bytes per lane into the epilogue, and the asm-issued LDS-DMA loop
//  alone is 9 % behind the builtin one.  Left out; the remaining lever for those layers is a smaller accumulator footprint.)
// (Second persistent attempt, also measured and dropped: the 256x128 tile (64 accumulators per lane, 163 VGPRs, no spills) as a
//  one-workgroup-per-CU stream through the 3-slot ring -- load cursor two K-steps ahead ACROSS tile boundaries, epilogue out
//  of the slot just consumed, store-aware counted vmcnt.  Parity-clean incl. ragged tiles, but 277 vs 280 us on the 64x64
//  GEGLU layer and 2074 vs 1956 us on the big conv: with all eight waves in lockstep the epilogue is still a phase in which
//  the MFMA pipe idles; overlapping it needs a second, independently phased workgroup on the CU, which LDS does not allow
//  for tiles with enough FLOP per staged byte.)
// out = act(scale * sum_z ws[z] + bias + rowvec) + residual for the tiles [tile0, tile0+nwg) of a split launch; a block
// owns one (tn, tm) fragment of each of the tile's 4 waves and reads the partials in the layout the GEMM wrote (1 KiB
// per wave instruction), fixed z order.
template <typename T, int BM, int BN>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const sr_igemm_args p, const int M, const int rpb, const int S,
                                                            const int NT, const int nwg, const int tile0) {
  constexpr int TM = BM / 2 / 16, TN = BN / 2 / 16, FR = TN * TM;
  const int tile_l = blockIdx.x / FR, fr = blockIdx.x - tile_l * FR;
  const int tn = fr / TM, tm = fr - tn * TM;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, c16 = lane & 15, g4 = lane >> 4;
  const int wg = tile0 + tile_l;
  int mt, nt;
  sr_tile_of(wg, NT, (M + BM - 1) / BM, p.tile_order, mt, nt);
  const int m = mt * BM + (wv >> 1) * (BM / 2) + tm * 16 + c16;
  const int n = nt * BN + (wv & 1) * (BN / 2) + tn * 16 + 4 * g4;
  const int N = p.N;
  if (m >= M || n >= N) return;
  const f32x4* ws = (const f32x4*)p.workspace + ((int64_t)tile_l * 4 + wv) * (FR * 64) + fr * 64 + lane;
  const int64_t zs = (int64_t)nwg * 4 * FR * 64;
  f32x4 acc = ws[0];
  for (int z = 1; z < S; ++z) acc += ws[z * zs];
  float v[4] = {acc[0] * p.scale, acc[1] * p.scale, acc[2] * p.scale, acc[3] * p.scale};
  if (p.row_stats) {
    const float2 rs = *(const float2*)(p.row_stats + 2 * (int64_t)m);
    const float4 cs = *(const float4*)(p.colsum + n);
    v[0] = fmaf(rs.x, v[0], rs.y * cs.x); v[1] = fmaf(rs.x, v[1], rs.y * cs.y);
    v[2] = fmaf(rs.x, v[2], rs.y * cs.z); v[3] = fmaf(rs.x, v[3], rs.y * cs.w);
  }
  if (p.bias) { const float4 bv = *(const float4*)(p.bias + n); v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w; }
  if (p.rowvec) {
    const int ldr = p.rowvec_ld ? p.rowvec_ld : N;
    const float4 rv = *(const float4*)(p.rowvec + (int64_t)(m / rpb) * ldr + n);
    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
  }
  if (p.act == 1) { for (int r = 0; r < 4; ++r) v[r] = sr_silu_f(v[r]); }
  else if (p.act == 3) { for (int r = 0; r < 4; ++r) v[r] = sr_gelu_f(v[r]); }
  else if (p.act == 4) { for (int r = 0; r < 4; ++r) v[r] = fminf(fmaxf((v[r] + 1.0f) * 0.5f, 0.0f), 1.0f); }
  const int64_t oi = (int64_t)m * N + n;
  if (p.residual) {
    if constexpr (sizeof(T) == 2) {
      const h16x4 rr = *(const h16x4*)((const _Float16*)p.residual + oi);
      v[0] += (float)rr[0]; v[1] += (float)rr[1]; v[2] += (float)rr[2]; v[3] += (float)rr[3];
    } else {
      const float4 rr = *(const float4*)((const float*)p.residual + oi);
      v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
    }
  }
  if (p.out_f32 || sizeof(T) == 4) *(float4*)((float*)p.out + oi) = make_float4(v[0], v[1], v[2], v[3]);
  else { h16x4 hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]}; *(h16x4*)((_Float16*)p.out + oi) = hv; }
}

// tiles [0, tile0) full-K (ordinary epilogue), tiles [tile0, ntiles) split S ways over K + reduce
template <typename T, int BM, int BN, int STAGES = 2>
int launch_split(const sr_igemm_args& a, int M, int Ho, int Wo, int tile0, int S, hipStream_t st) {
  const int MT = (M + BM - 1) / BM, NTv = (a.N + BN - 1) / BN, ntiles = MT * NTv, ntail = ntiles - tile0;
  constexpr int lds_stage = STAGES * (BM + BN) * 128;
  constexpr int lds_epi = 4 * (BM / 2) * ((BN / 2) * 4 + 16);
  constexpr int lds = (lds_stage > lds_epi ? lds_stage : lds_epi) + (sizeof(T) == 2 ? 2 * BN * 4 : 0) + 4 * 256;   // (the unsplit head tiles' bias / colsum rows, the prefetch landing zone)
  auto kf = igemm_kernel<T, BM, BN, 2, 2, STAGES, false, false>;
  auto ks = igemm_kernel<T, BM, BN, 2, 2, STAGES, false, true>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)ks, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  if (tile0 > 0) hipLaunchKernelGGL(kf, dim3(tile0), dim3(256), lds, st, a, M, Ho, Wo, NTv, tile0, 0);
  sr_igemm_args as = a;
  if (ntail > SR_IGEMM_SPLIT_COUNTERS) as.split_counters = nullptr;
  hipLaunchKernelGGL(ks, dim3(ntail, S), dim3(256), lds, st, as, M, Ho, Wo, NTv, ntail, tile0);
  if (!as.split_counters) {
    constexpr int FR = (BM / 32) * (BN / 32);
    hipLaunchKernelGGL((splitk_reduce_kernel<T, BM, BN>), dim3(ntail * FR), dim3(256), 0, st, a, M, Ho * Wo, S, NTv, ntail, tile0);
  }
  SR_CHECK_LAUNCH("sr_igemm(split-K)");
  return SR_OK;
}

// ---- patch-stationary 3x3 convolution (tile 8): 256 x 320, 8 waves ------------------------------------------------------
// The implicit GEMM above re-stages the activation tile for each of the nine filter taps (PMC: 375 MB fetched per launch of
// the 64x64 C320 conv against 43.8 MB of input).  Here a workgroup's 256 output pixels are whole image rows (or whole 8x8
// images), so the pixels any tap needs are the tile's own rows plus a one-pixel halo: that halo patch is staged ONCE per
// 32-channel chunk and the nine taps read it at nine LDS offsets.  Per chunk and workgroup 32 KB of activations + 9 x 20 KB of
// weights go through the 64 B/clk TCP->LDS path instead of 9 x (16 + 20) KB.
//   LDS: activation halo, double buffered, pitch 80 B (64 B of channels + 16 B pad: 16 consecutive rows fall on 16 distinct
//   16-byte bank groups, so a tap shift is a plain address offset with no swizzle to re-derive); weight stages in a 4-slot
//   ring of 64-byte K-steps (the BKB = 64 layout of igemm_kernel).  2 x 32 KB + 4 x 20 KB = 144 KB.
//   Pipeline: step s = (chunk c, tap t); weights run three steps ahead, the four halo pieces of chunk c+1 are issued at taps
//   0..3 of chunk c; all LDS-DMA from inline asm with counted vmcnt (the count per tap is a compile-time constant because the
//   tap loop is unrolled).
__global__ __launch_bounds__(512, 1) void conv3p_kernel(const sr_igemm_args p, const int M, const int NT, const int nwg) {
  using T = _Float16;
  extern __shared__ __attribute__((aligned(16))) char smem_all[];
  char* const smem = smem_all + 8 * 256;               // (in front: the prefetch landing zone, 256 B per wave)
  constexpr int BM = 256, BN = 320, WAVES_M = 4, WAVES_N = 2, TM = 4, TN = 10;
  constexpr int APITCH = 80, ABYTES = 32768, BBYTES = BN * 64, BOFF = 2 * ABYTES, LDS_TOTAL = BOFF + 4 * BBYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (p.prefetch)
    sr_prefetch_touch(p.prefetch, p.prefetch_bytes, (int64_t)blockIdx.x * 512 + tid, (int64_t)gridDim.x * 512,
                      __builtin_amdgcn_readfirstlane(sr_lds_addr(smem_all) + wv * 256));
  const int wg = sr_xcd_remap(blockIdx.x, nwg);
  const int mt = wg / NT, nt = wg - mt * NT;
  const int m0 = mt * BM, n0 = nt * BN;
  const int H = p.H, W = p.W, C1 = p.C1, rpb = H * W;
  const int ipt = rpb >= 256 ? 1 : 256 / rpb;          // images per tile
  const int Ri = rpb >= 256 ? 256 / W : H;             // image rows (per image) in the tile
  const int PW = W + 2, HPI = (Ri + 2) * PW, HR = ipt * HPI;
  const int b0 = m0 / rpb, y0 = (m0 - b0 * rpb) / W;

  // ---- halo pieces: LDS-DMA instruction k = wv*4 + i fills the 64 sixteen-byte slots [k*64, k*64+64) of the halo buffer.
  // Slots outside the image (and the pad chunk of every row) are the same for every channel chunk: they are zeroed once in
  // both buffers and the LDS-DMA runs with those lanes masked off.  Sources are 32-bit offsets from a wave-uniform base
  // (SGPR pair) that advances by 64 bytes per chunk: one VGPR per piece.
  unsigned aoffg[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int u = (wv * 4 + i) * 64 + lane;           // slot -> (halo row, 16-B chunk; chunk 4 is the pad)
    const int hrow = u / 5, ch = u - hrow * 5;
    const int img = hrow / HPI, rem = hrow - img * HPI, hr = rem / PW, hc = rem - hr * PW;
    const int y = y0 + hr - 1, x = hc - 1, b = b0 + img;
    const bool ok = ch < 4 && hrow < HR && y >= 0 && y < H && x >= 0 && x < W && b < p.B;
    aoffg[i] = ok ? (unsigned)(((((int64_t)b * H + y) * W + x) * C1) * 2 + ch * 16) : 0xFFFFFFFFu;
    if (!ok) {
      *(uint4*)(smem + (wv * 4 + i) * 1024 + lane * 16) = make_uint4(0, 0, 0, 0);
      *(uint4*)(smem + ABYTES + (wv * 4 + i) * 1024 + lane * 16) = make_uint4(0, 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the zero fill is in LDS before this wave's first barrier
  // ---- weight rows (64-byte K-steps: 16 rows per instruction, source-side swizzle as in igemm_kernel): the per-lane part
  // (row within the 16-row group, chunk) is the same for the three instructions of a stage, the group is wave-uniform
  const int lrow = lane >> 2;
  const int lchunk = (0x1320 >> (4 * ((lane & 3) ^ ((lrow >> 2) & 3)))) & 3;
  const unsigned wlane = (unsigned)(lrow * 9 * C1 * 2 + lchunk * 16);
  const unsigned ldsA = __builtin_amdgcn_readfirstlane(sr_lds_addr(smem)), ldsB = ldsA + BOFF;
  int ac = 0;                                           // halo chunk cursor
  auto issueA = [&](int buf, int i) {
    if (aoffg[i] != 0xFFFFFFFFu) sr_glds16_asm_saddr(aoffg[i], (const char*)p.a + (int64_t)ac * 64, ldsA + buf * ABYTES + (wv * 4 + i) * 1024);
  };
  int lc = 0, lt = 0;                                   // load cursor (chunk, tap) of the next weight stage
  auto issueB = [&](int slot) {
    const char* wb = (const char*)p.w + ((int64_t)n0 * 9 * C1 + lt * C1 + lc * 32) * 2;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int gi = (i * 8 + wv) < 20 ? (i * 8 + wv) : 19;
      sr_glds16_asm_saddr(wlane, wb + (int64_t)gi * 16 * 9 * C1 * 2, ldsB + slot * BBYTES + gi * 1024);
    }
    if (++lt == 9) { lt = 0; ++lc; }
  };

  // ---- fragment offsets
  const int c16 = lane & 15, g4 = lane >> 4;
  const int wm = wv >> 1, wn = wv & 1;
  const int pm0 = wm * 64, qn0 = wn * 160;
  // fragment rows of the wave's four 16-pixel groups: per-lane base + wave-uniform deltas (16 | W, or two 8-pixel rows)
  int aoff0, adel[TM];
  {
    const int riw = Ri * W;
    auto hrow_of = [&](int pi) { const int img = pi / riw, q = pi - img * riw, r = q / W, cx = q - r * W; return img * HPI + r * PW + cx; };
    aoff0 = hrow_of(pm0 + c16) * APITCH + g4 * 16;
#pragma unroll
    for (int j = 0; j < TM; ++j) adel[j] = __builtin_amdgcn_readfirstlane((hrow_of(pm0 + j * 16) - hrow_of(pm0)) * APITCH);
  }
  const int foffB = qn0 * 64 + c16 * 64 + ((((c16 >> 2) & 3) ^ ((0x2130 >> (4 * g4)) & 3)) << 4);
  const int pw80 = PW * APITCH;

  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int NC = C1 / 32;
  {
    const unsigned m0k = sr_m0_save();
#pragma unroll
    for (int i = 0; i < 4; ++i) issueA(0, i);
    ac = 1;
    issueB(0); issueB(1); issueB(2);
    sr_m0_restore(m0k);
  }
  // One barrier per step (one tap of one chunk): wait for this wave's pieces of weight stage s, barrier, start the 14 fragment
  // reads, issue the stage three steps ahead (+ a halo piece) behind them, 40 MFMAs.
  // Measured alternatives on the 64x64 C320 conv (this form: 118 us; the re-staging 256x320 tile: 124 us): two taps per
  // barrier with one pair of stages in flight 126 us (prefetch depth matters more than barrier count); ping-pong of the two
  // waves of each SIMD (memory phase | barrier | MFMA phase, waves 4..7 half a step late) 137 us, and 163 us with the groups
  // chosen so that both waves of a SIMD share a phase (which shows waves w and w+4 share a SIMD, and that the second barrier
  // and the drained fragment reads cost more than the overlap returns: the 112 KB of fragment reads per step are ~70 % of the
  // step's MFMA time on this tile and already overlap the MFMAs within a wave).
  int s = 0;
  auto chunk = [&](auto lastc, const int c) {
    constexpr bool LASTC = decltype(lastc)::value;
    auto step = [&](auto tc) {
      constexpr int t = decltype(tc)::value;
      // this wave's LDS-DMA instructions younger than weight stage s: stages s+1, s+2 and the halo pieces issued at taps 0..3
      // among the last three steps (unrolled taps -> compile-time counts)
      constexpr int nx = LASTC ? (t <= 6 ? 6 : (t == 7 ? 3 : 0))
                               : 6 + (t >= 1 && t <= 4) + (t >= 2 && t <= 5) + (t >= 3 && t <= 6);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(nx) : "memory");
      __builtin_amdgcn_s_barrier();                       // stage s (at t == 0 also the halo of chunk c) visible; step s-1 done
      const char* Bq = smem + BOFF + (s & 3) * BBYTES + foffB;
      // (opaque to the optimiser: otherwise the 9 x 4 loop-invariant fragment addresses are hoisted out of the chunk loop into
      //  VGPRs the kernel does not have, and their reloads from scratch carry s_waitcnt vmcnt(0) into the main loop)
      int to = (t / 3) * pw80 + (t % 3) * APITCH + (c & 1) * ABYTES;
      asm volatile("" : "+s"(to));
      uint4 xf[TM], wf[TN];
#pragma unroll
      for (int j = 0; j < TM; ++j) xf[j] = *(const uint4*)(smem + aoff0 + (adel[j] + to));
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) wf[tn] = *(const uint4*)(Bq + tn * 1024);
      if (!LASTC || t < 6) {
        const unsigned m0k = sr_m0_save();
        issueB((s + 3) & 3);
        if constexpr (!LASTC) { if (t < 4) issueA((c + 1) & 1, t); if (t == 3) ++ac; }
        sr_m0_restore(m0k);
      }
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) sr_mma(acc[tn][tm], wf[tn], xf[tm], T());
      ++s;
    };
    step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
    step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
    step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
  };
  for (int c = 0; c < NC - 1; ++c) chunk(std::false_type{}, c);
  chunk(std::true_type{}, NC - 1);

  epilogue_rows<T, BM, BN, WAVES_M, WAVES_N, LDS_TOTAL>(p, acc, smem, M, rpb, m0, n0, pm0, qn0, wv, lane);
}

static bool conv3p_ok(const sr_igemm_args& a, int M) {
  if (a.dtype != SR_F16 || a.KH != 3 || a.stride != 1 || a.upsample || a.C2 || a.transpose_out || a.row_stats || a.ln_inline || a.pad_br) return false;
  if (a.C1 % 64 || a.N % 320 || M % 256 || a.act == 2) return false;
  const int rpb = a.H * a.W;
  if (a.W < 8 || 256 % a.W) { if (!(rpb < 256 && 256 % rpb == 0)) return false; }
  if (rpb >= 256 ? (rpb % 256 != 0 || 256 % a.W != 0) : (256 % rpb != 0)) return false;
  const int ipt = rpb >= 256 ? 1 : 256 / rpb, Ri = rpb >= 256 ? 256 / a.W : a.H;
  if ((int64_t)a.B * rpb * a.C1 * 2 >= 0xFFFFFFFFLL) return false;          // 32-bit source offsets
  return ipt * (Ri + 2) * (a.W + 2) * 80 <= 32768;
}

static int launch_conv3p(const sr_igemm_args& a, int M, hipStream_t st) {
  constexpr int lds = 2 * 32768 + 4 * 320 * 64 + 8 * 256;
  const int NT = a.N / 320, nwg = (M / 256) * NT;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)conv3p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_set = true; }
  hipLaunchKernelGGL(conv3p_kernel, dim3(nwg), dim3(512), lds, st, a, M, NT, nwg);
  SR_CHECK_LAUNCH("sr_igemm(conv3p)");
  return SR_OK;
}

// Wave quantisation model for the 4-wave tiles (calibrated on MI355X): `slots` workgroups are co-resident
// chip-wide, a round of co-resident workgroups takes t_k microseconds per K-step, partials cost their HBM round trip.
// Returns the split count for the tail round (1 = none) and sets tile0 (tiles before it run unsplit).
static inline int plan_split(int64_t tiles, int KT, int slots, double t_k, int64_t tile_ws_bytes, int64_t ws_bytes, int* tile0, int forceS = 0) {
  const int64_t full = tiles / slots, tail = tiles - full * slots;
  *tile0 = (int)(full * slots);
  if (tail == 0) return 1;
  if (forceS > 0)                                            // sr_igemm_args.split = S: the per-shape tuner measures S instead of modelling it
    return (KT / forceS >= 4 && (int64_t)forceS * tail * tile_ws_bytes <= ws_bytes) ? forceS : -1;
  if (const char* fs = getenv("SR_SPLIT_S")) {               // tuning aid: force the split count of the tail round
    const int S = atoi(fs);
    if (S >= 1 && KT / S >= 4 && (int64_t)S * tail * tile_ws_bytes <= ws_bytes) return S;
  }
  double best = (double)KT * t_k;                            // the tail as one more (partly empty) round
  int bestS = 1;
  for (int S = 2; S <= 16; ++S) {
    if (KT / S < 8 || (int64_t)S * tail * tile_ws_bytes > ws_bytes) break;
    const int64_t rounds = (tail * S + slots - 1) / slots;
    const double t = (double)rounds * ((KT + S - 1) / S) * t_k + (double)(2 * S * tail * tile_ws_bytes) / 4.0e6 + 14.0;   // + two more launches
    if (t < best * 0.9) { best = t; bestS = S; }
  }
  return bestS;
}

template <typename T, bool TRANS>
int dispatch(const sr_igemm_args& a, int M, int Ho, int Wo, hipStream_t st) {
  const int n128 = (a.N + 127) / 128, n64 = (a.N + 63) / 64;
  const int64_t wg_128x128 = (int64_t)((M + 127) / 128) * n128;
  const bool waste128 = (n128 * 128 - a.N) * 8 > a.N;          // >12.5% padded columns with BN=128
  const char* fenv = getenv("SR_IGEMM_TILE");                 // tuning aid, overrides args.tile
  const int force = fenv ? atoi(fenv) : a.tile;              // 1=256x128x3, 2=128x128, 3=128x64, 4=64x64, 5=256x320
  static const bool split_off = getenv("SR_SPLITK") && atoi(getenv("SR_SPLITK")) == 0;      // tuning / A-B aid
  const int KT = a.KH * a.KH * ((a.C1 + a.C2) / (int)(128 / sizeof(T)));
  const int forceS = a.split > 1 ? a.split : 0;
  const bool can_split = !TRANS && !split_off && a.split >= 0 && a.workspace && a.act != 2 && a.N % 4 == 0 && !a.ln_inline;
  const bool may_split = can_split && KT >= 32;
  // Deep-ring tiles for grids of less than one workgroup per CU (13 = 64x64 x 8 stages, 14 = 128x64 x 6, 15 = 128x128 x 4; 128-144
  // KB of LDS, one workgroup per CU).  The 2-stage tiles above have ONE K-step of loads in flight per workgroup and rely on two
  // or three co-resident workgroups to hide the rest of the latency; a launch with 40..160 workgroups (the 16x16 / 8x8 levels, any
  // level of a one-view batch: M = 128..8192) has no co-resident partner, so every K-step cost a full L2 / HBM round trip --
  // 20.8 us for M 512 K 1280 N 1280 (1.7 GFLOP, 4.6 MB).  Here the whole ring is requested up front (up to seven K-steps in
  // flight per workgroup, counted vmcnt as above); 14 / 15 also split K (from 8 K-steps on) so that a short loop is entirely in
  // flight before its first MFMA.
  const bool deep_split = can_split && KT >= 8 && (force == 14 || force == 15);
  if (forceS && !((may_split && (force == 2 || force == 3)) || deep_split))
    SR_FAIL(SR_ERR_INVALID, "sr_igemm: split = %d needs tile 2 / 3 (K >= 32 steps) or 14 / 15 (K >= 8 steps), a workspace, no GEGLU / transposed output", a.split);
  if constexpr (!TRANS) {
    if (deep_split && forceS) {
      int tile0 = 0;
      const int64_t tiles = force == 15 ? wg_128x128 : (int64_t)((M + 127) / 128) * n64;
      const int S = plan_split(tiles, KT, 256, 1.0, 128 * (force == 15 ? 128 : 64) * 4, a.workspace_bytes, &tile0, forceS);
      if (S < 0) SR_FAIL(SR_ERR_INVALID, "sr_igemm: split = %d does not fit this shape (K steps %d, workspace %lld B)", a.split, KT, (long long)a.workspace_bytes);
      if (S > 1) return force == 15 ? launch_split<T, 128, 128, 4>(a, M, Ho, Wo, tile0, S, st) : launch_split<T, 128, 64, 6>(a, M, Ho, Wo, tile0, S, st);
      SR_FAIL(SR_ERR_INVALID, "sr_igemm: split = %d: nothing to split (whole rounds of workgroups only)", a.split);
    }
    if (may_split && (force == 2 || force == 3)) {           // tuned tile + modelled (split 0) or given (split S) tail split
      int tile0 = 0;
      if (force == 2) {
        const int S = plan_split(wg_128x128, KT, 512, 1.2, 128 * 128 * 4, a.workspace_bytes, &tile0, forceS);
        if (S < 0) SR_FAIL(SR_ERR_INVALID, "sr_igemm: split = %d does not fit this shape (K steps %d, workspace %lld B)", a.split, KT, (long long)a.workspace_bytes);
        if (S > 1) return launch_split<T, 128, 128>(a, M, Ho, Wo, tile0, S, st);
      } else {
        const int S = plan_split((int64_t)((M + 127) / 128) * n64, KT, 768, 0.93, 128 * 64 * 4, a.workspace_bytes, &tile0, forceS);
        if (S < 0) SR_FAIL(SR_ERR_INVALID, "sr_igemm: split = %d does not fit this shape (K steps %d, workspace %lld B)", a.split, KT, (long long)a.workspace_bytes);
        if (S > 1) return launch_split<T, 128, 64>(a, M, Ho, Wo, tile0, S, st);
      }
      if (forceS) SR_FAIL(SR_ERR_INVALID, "sr_igemm: split = %d: nothing to split (whole rounds of workgroups only)", a.split);
    }
  }
  if (force == 2) return launch_ln<T, 128, 128, 2, 2, 2, TRANS>(a, M, Ho, Wo, st);
  if (force == 3) return launch_ln<T, 128, 64, 2, 2, 2, TRANS>(a, M, Ho, Wo, st);
  if (force == 4) return launch_ln<T, 64, 64, 2, 2, 2, TRANS>(a, M, Ho, Wo, st);
  if (force == 13) return launch_ln<T, 64, 64, 2, 2, 8, TRANS>(a, M, Ho, Wo, st);
  if (force == 14) return launch_ln<T, 128, 64, 2, 2, 6, TRANS>(a, M, Ho, Wo, st);
  if (force == 15) return launch_ln<T, 128, 128, 2, 2, 4, TRANS>(a, M, Ho, Wo, st);
  if (force == 8) {
    if constexpr (!TRANS && sizeof(T) == 2) { if (conv3p_ok(a, M)) return launch_conv3p(a, M, st); }
    SR_FAIL(SR_ERR_INVALID, "sr_igemm: tile 8 (patch-stationary 3x3) needs fp16, KH=3, stride 1, one source, N %% 320 == 0, "
                            "256-pixel tiles of whole rows / images (H=%d W=%d M=%d N=%d)", a.H, a.W, M, a.N);
  }
  if constexpr (!TRANS && sizeof(T) == 2) {
    // (4-wave 128x128 / 128x64 tiles with a 4-deep ring of 64-byte K-steps measured 5..30 % slower than their 2 x 128-byte
    //  form on every UNet shape: twice the barriers per K, and those tiles already overlap through co-resident workgroups)
    // 256x320 tile (8 waves, 64x160 per wave): every UNet layer width is a multiple of 320, so no padded columns, the
    // activation tile is fetched once for N = 320, and 142 FLOP per byte staged through the 64 B/clk TCP->LDS path (a
    // 128x128 tile: 64 FLOP/B = exactly the MFMA rate, so that path saturates first).  Needs a full round of workgroups:
    // measured 830 vs 627 TF/s on the 64x64 C320 3x3 conv, 1086 vs 1000 on C1280, but 556 vs 678 at 32x32 C640.
    if ((force >= 5 && force <= 7) && a.N % 320) SR_FAIL(SR_ERR_INVALID, "sr_igemm: tile %d (256x320) needs N %% 320 == 0, N=%d", force, a.N);
    if (force == 6) return launch<T, 256, 320, 4, 2, 4, TRANS, 64>(a, M, Ho, Wo, st);
    // Two co-resident workgroups per CU for the K-short linear layers (to_q / to_out / proj_in / proj_out, GEGLU: K = 320..1280,
    // 30 launches each per UNet evaluation): with one 8-wave workgroup per CU every workgroup of the single round loads, computes
    // and stores in the same phase, so HBM reads, MFMA and HBM writes never overlap; two workgroups (<= 80 KB LDS, <= 128 VGPRs
    // each) drift apart and overlap them.  9 = 128x160 (8 waves of 32x80, 74 KB, 92 VGPRs): 29.5 vs 36.2 us at M65536 K320
    // N320, 23.3 vs 25.1 at M16384 K640 N640, 24.3 vs 26.6 at M4096 K1280 N1280, 233 vs 263 us on the 64x64 GEGLU.
    // 10 = 128x320 with 64-byte K-steps (57 KB; reads the activation tile once for N = 320): 28.8 us on the first, 174-180 us
    // on the 32x32 GEGLU.  (A 3-slot ring of 64-byte K-steps under the 128x160 tile measured 5..10 % behind the 2 x 128-byte form.)
    if (force == 9) {
      if (a.N % 160) SR_FAIL(SR_ERR_INVALID, "sr_igemm: tile 9 (128x160) needs N %% 160 == 0, N=%d", a.N);
      return launch_ln<T, 128, 160, 4, 2, 2, TRANS, 128, 0, 4>(a, M, Ho, Wo, st);
    }
    if (force == 10) {
      if (a.N % 320) SR_FAIL(SR_ERR_INVALID, "sr_igemm: tile 10 (128x320) needs N %% 320 == 0, N=%d", a.N);
      return launch_ln<T, 128, 320, 2, 4, 2, TRANS, 64, 0, 4>(a, M, Ho, Wo, st);
    }
    // the same idea for layer widths that are multiples of 128 but not of 160 (the VAE decoder): 11 = 128x128 as 8 waves of
    // 32x64 (64 KB), 12 = 256x128 with 64-byte K-steps (49 KB).  844 vs 899 us on the 512x512 C128 conv, 369 vs 457 us on the
    // 2M-row C256 -> 128 1x1 layer; level with the 4-wave / 3-stage forms elsewhere -- the tuner decides per shape.
    // (3- and 4-deep rings under the 128x160 tile change nothing on the K-long few-tile layers -- 24.4 us at M4096 K1280 N1280
    //  either way: with 128x160 tiles that layer moves 191 MB from L2 to the CUs, i.e. it runs at the L2 read bandwidth -- and
    //  lose 25 % where two workgroups per CU mattered)
    if (force == 11) return launch_ln<T, 128, 128, 4, 2, 2, TRANS, 128, 0, 4>(a, M, Ho, Wo, st);
    if (force == 12) return launch_ln<T, 256, 128, 4, 2, 2, TRANS, 64, 0, 4>(a, M, Ho, Wo, st);
    // 128x320 (8 waves as 2x4, 64x80 per wave, 112 KB): the same full-width rows for layers with half as many pixels -- the
    // 32x32 level (M = 16384, N = 640) gets exactly one round of 256 workgroups: 130 vs 161 us on its 3x3 conv (926 TF/s),
    // 61 vs 80 us on the K = 2560 linear.  (A 64x320 tile for the 16x16 level measured no better than split-K 128x128.)
    if (force == 7 || (force == 0 && a.N % 320 == 0 && (int64_t)((M + 255) / 256) * (a.N / 320) < 256 &&
                       (int64_t)((M + 127) / 128) * (a.N / 320) >= 256))
      return launch_ln<T, 128, 320, 2, 4, 2, TRANS>(a, M, Ho, Wo, st);
    // (a 256x160 tile with FOUR waves of 128x80 and 64-byte K-steps -- 53 KB, two workgroups per CU so that one's epilogue
    //  overlaps the other's main loop -- measured 636 vs 1100 TF/s on the big conv and 431 vs 273 us on the 64x64 GEGLU layer)

    // (LDS-DMA burst issued between the two k-substeps -- SPREAD 2 -- so the matrix pipe has work right after the barrier:
    //  +3..9 %, 134 -> 125 us on the 64x64 C320 conv, 1200 -> 1286 TF/s on the big one, 252 -> 236 us on the 64x64 GEGLU;
    //  the 128x320 tile is indifferent to it and the 4-wave tiles lose 5..10 %)
    if (force == 5 || (force == 0 && a.N % 320 == 0 && (int64_t)((M + 255) / 256) * (a.N / 320) >= 256))
      return launch_ln<T, 256, 320, 4, 2, 2, TRANS, 128, 2>(a, M, Ho, Wo, st);
  } else {
    if (force >= 5 && force <= 12) SR_FAIL(SR_ERR_INVALID, "sr_igemm: tile %d is fp16, non-transposed only", force);
  }
  const int64_t wg_256x128 = (int64_t)((M + 255) / 256) * n128;
  const bool big = !waste128 && wg_256x128 >= 512;
  // split-K of the last, partly empty round of workgroups (all of them when the whole grid is less than one round: the
  // 8x8 / 16x16 UNet levels).  fp32 partials go through the caller's workspace; a fixed-order reduce kernel applies the
  // epilogue (bit-reproducible, no float atomics).
  if constexpr (!TRANS) {
    if (force == 0 && may_split && !big) {
      int tile0 = 0;
      if (!waste128) {
        const int S = plan_split(wg_128x128, KT, 512, 1.2, 128 * 128 * 4, a.workspace_bytes, &tile0);
        if (S > 1) return launch_split<T, 128, 128>(a, M, Ho, Wo, tile0, S, st);
      } else {
        const int S = plan_split((int64_t)((M + 127) / 128) * n64, KT, 768, 0.93, 128 * 64 * 4, a.workspace_bytes, &tile0);
        if (S > 1) return launch_split<T, 128, 64>(a, M, Ho, Wo, tile0, S, st);
      }
    }
  }
  // 256x128 tile, 8 waves, 3-deep LDS ring with counted vmcnt: +5..17 % on the large-M layers (measured 1003 vs 858 TF/s
  // on the 64x64x1280 3x3 conv); smaller problems keep the 4-wave 2-stage tiles (finer granularity, same rate there)
  // (its LDS-DMA is issued piecewise between the MFMA groups -- compute_staging -- which is worth +5..8 % on this 8-wave
  //  tile: 193 -> 176 us on the 64x64 C320 conv, 965 -> 1020 TF/s on the big one; the 4-wave tiles lose 10..20 % with it and
  //  the 256x320 tile has no registers left for it)
  if (force == 1 || (force == 0 && big && !a.ln_inline))
    return launch<T, 256, 128, 4, 2, 3, TRANS, 128, 1>(a, M, Ho, Wo, st);
  // (a 128x160 tile for N = 320 measured slower than five 64-wide tiles on MI355X: 487 vs 612 TF/s on the 3x3 conv)
  // (1x1 layers with few 128x128 tiles run faster on 128x64: three co-resident workgroups per CU hide the short K loop's
  //  ramp; 26.8 vs 31.3 us at M4096 K1280 N1280, 30.8 vs 36.0 at M16384 K640 N640 -- what the per-shape tuner picks too)
  if (!waste128 && wg_128x128 >= 192 && !(a.KH == 1 && wg_128x128 <= 768)) return launch_ln<T, 128, 128, 2, 2, 2, TRANS>(a, M, Ho, Wo, st);
  const int64_t wg_128x64 = (int64_t)((M + 127) / 128) * n64;
  if (wg_128x64 >= 192) return launch_ln<T, 128, 64, 2, 2, 2, TRANS>(a, M, Ho, Wo, st);
  if ((int64_t)((M + 63) / 64) * n64 <= 256 && KT >= 4) return launch_ln<T, 64, 64, 2, 2, 8, TRANS>(a, M, Ho, Wo, st);   // no co-resident partner: deep ring
  return launch_ln<T, 64, 64, 2, 2, 2, TRANS>(a, M, Ho, Wo, st);
}

}  // namespace

// argument checks shared by sr_igemm and sr_igemm_group; -> M, Ho, Wo of the GEMM view
static int igemm_check(const sr_igemm_args* a, int* M_, int* Ho_, int* Wo_) {
  if (!a || !a->a || !a->w || !a->out || !a->zero_page) SR_FAIL(SR_ERR_INVALID, "sr_igemm: null pointer");
  const int ke = a->dtype == SR_F16 ? 64 : 32;
  if (a->dtype != SR_F16 && a->dtype != SR_F32) SR_FAIL(SR_ERR_INVALID, "sr_igemm: bad dtype %d", a->dtype);
  if (a->C1 <= 0 || a->C1 % ke || a->C2 % ke || a->C2 < 0) SR_FAIL(SR_ERR_INVALID, "sr_igemm: C1=%d C2=%d must be multiples of %d", a->C1, a->C2, ke);
  if (a->C2 > 0 && !a->a2) SR_FAIL(SR_ERR_INVALID, "sr_igemm: C2>0 without a2");
  if (a->KH != 1 && a->KH != 3) SR_FAIL(SR_ERR_INVALID, "sr_igemm: KH=%d", a->KH);
  if (a->stride != 1 && a->stride != 2) SR_FAIL(SR_ERR_INVALID, "sr_igemm: stride=%d", a->stride);
  if (a->upsample && a->stride != 1) SR_FAIL(SR_ERR_INVALID, "sr_igemm: upsample with stride");
  if (a->pad_br && (a->KH != 3 || a->upsample)) SR_FAIL(SR_ERR_INVALID, "sr_igemm: pad_br is for 3x3 convs without upsample");
  if (a->N <= 0 || a->B <= 0 || a->H <= 0 || a->W <= 0) SR_FAIL(SR_ERR_INVALID, "sr_igemm: bad sizes");
  if (a->act == 2 && (a->N % 4 || a->transpose_out)) SR_FAIL(SR_ERR_INVALID, "sr_igemm: GEGLU needs N%%4==0");
  if (a->row_stats && !a->colsum) SR_FAIL(SR_ERR_INVALID, "sr_igemm: row_stats without colsum");
  if (a->ln_inline && (a->row_stats || !a->colsum)) SR_FAIL(SR_ERR_INVALID, "sr_igemm: ln_inline takes colsum and no row_stats");
  if ((a->row_stats || a->ln_inline) && (a->KH != 1 || a->stride != 1 || a->upsample || a->C2)) SR_FAIL(SR_ERR_INVALID, "sr_igemm: folded LayerNorm is for 1x1 single-source layers");
  if (a->tile < 0 || a->tile > 15 || a->split < -1 || a->split == 1 || a->split > 16) SR_FAIL(SR_ERR_INVALID, "sr_igemm: tile=%d split=%d", a->tile, a->split);
  if (a->tile_order != 0 && a->tile_order != 1) SR_FAIL(SR_ERR_INVALID, "sr_igemm: tile_order=%d (0 rows first, 1 columns first)", a->tile_order);
  int Ho, Wo;
  if (a->upsample && ((a->up_h > 0) != (a->up_w > 0))) SR_FAIL(SR_ERR_INVALID, "sr_igemm: up_h / up_w go together");
  if (!a->upsample && (a->up_h || a->up_w)) SR_FAIL(SR_ERR_INVALID, "sr_igemm: up_h / up_w without upsample");
  if (a->upsample) { Ho = a->up_h > 0 ? a->up_h : 2 * a->H; Wo = a->up_w > 0 ? a->up_w : 2 * a->W; }
  else if (a->stride == 2) {
    const int ptot = a->pad_br ? a->KH / 2 : 2 * (a->KH / 2);       // total zero rows / columns added per dimension
    Ho = (a->H + ptot - a->KH) / 2 + 1; Wo = (a->W + ptot - a->KH) / 2 + 1;
  }
  else { Ho = a->H; Wo = a->W; }
  const int64_t M64 = (int64_t)a->B * Ho * Wo;
  if (M64 > 0x7fffffffLL / 2) SR_FAIL(SR_ERR_INVALID, "sr_igemm: M too large");
  if (a->transpose_out && a->ldt < Ho * Wo) SR_FAIL(SR_ERR_INVALID, "sr_igemm: ldt < pixels per batch");
  *M_ = (int)M64; *Ho_ = Ho; *Wo_ = Wo;
  return SR_OK;
}

extern "C" int sr_igemm(const sr_igemm_args* a, void* stream) {
  int M, Ho, Wo;
  const int rc = igemm_check(a, &M, &Ho, &Wo);
  if (rc != SR_OK) return rc;
  hipStream_t st = sr_stream(stream);
  if (a->dtype == SR_F16) {
    return a->transpose_out ? dispatch<_Float16, true>(*a, M, Ho, Wo, st) : dispatch<_Float16, false>(*a, M, Ho, Wo, st);
  }
  return a->transpose_out ? dispatch<float, true>(*a, M, Ho, Wo, st) : dispatch<float, false>(*a, M, Ho, Wo, st);
}

// tiles a grouped launch can run (the non-split 4-wave tiles and the two-per-CU 8-wave tiles): 0 = no
template <typename T>
static int group_launch(int tile, const sr_igemm_args* const* as, const int* Ms, const int* Hos, const int* Wos, int n, hipStream_t st) {
  switch (tile) {
    case 2: return launch_group<T, 128, 128, 2, 2, 2>(as, Ms, Hos, Wos, n, st);
    case 3: return launch_group<T, 128, 64, 2, 2, 2>(as, Ms, Hos, Wos, n, st);
    case 4: return launch_group<T, 64, 64, 2, 2, 2>(as, Ms, Hos, Wos, n, st);
    case 13: return launch_group<T, 64, 64, 2, 2, 8>(as, Ms, Hos, Wos, n, st);
    case 14: return launch_group<T, 128, 64, 2, 2, 6>(as, Ms, Hos, Wos, n, st);
    case 15: return launch_group<T, 128, 128, 2, 2, 4>(as, Ms, Hos, Wos, n, st);
    default: break;
  }
  if constexpr (sizeof(T) == 2) {
    if (tile == 9) return launch_group<T, 128, 160, 4, 2, 2, 128, 0, 4>(as, Ms, Hos, Wos, n, st);
    if (tile == 10) return launch_group<T, 128, 320, 2, 4, 2, 64, 0, 4>(as, Ms, Hos, Wos, n, st);
  }
  return -100;
}

extern "C" int sr_igemm_group(const sr_igemm_args* const* args, int32_t n, void* stream) {
  if (!args || n < 1 || n > SR_IGEMM_GROUP_MAX) SR_FAIL(SR_ERR_INVALID, "sr_igemm_group: 1..%d problems", SR_IGEMM_GROUP_MAX);
  int M[SR_IGEMM_GROUP_MAX], Ho[SR_IGEMM_GROUP_MAX], Wo[SR_IGEMM_GROUP_MAX];
  bool one = n > 1;
  for (int i = 0; i < n; ++i) {
    const int rc = igemm_check(args[i], &M[i], &Ho[i], &Wo[i]);
    if (rc != SR_OK) return rc;
    const sr_igemm_args* a = args[i];
    // one launch needs one kernel: same dtype and pinned tile, row-major outputs, no split-K (its workspace and second grid
    // dimension belong to a single problem), and a tile that is legal for every member's width
    one = one && a->dtype == args[0]->dtype && a->tile == args[0]->tile && a->tile != 0 && !a->transpose_out && a->split == -1;
    if (a->tile == 9) one = one && a->N % 160 == 0;
    if (a->tile == 10) one = one && a->N % 320 == 0;
  }
  hipStream_t st = sr_stream(stream);
  if (one) {
    const int rc = args[0]->dtype == SR_F16 ? group_launch<_Float16>(args[0]->tile, args, M, Ho, Wo, n, st)
                                            : group_launch<float>(args[0]->tile, args, M, Ho, Wo, n, st);
    if (rc != -100) return rc;
  }
  for (int i = 0; i < n; ++i) {                               // not groupable as given: the same results, one launch each
    const int rc = sr_igemm(args[i], stream);
    if (rc != SR_OK) return rc;
  }
  return SR_OK;
}
