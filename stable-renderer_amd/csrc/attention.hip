// Fused attention  O = softmax(Q K^T * scale) V  on gfx950 MFMA (see include/sr_hip.h: sr_attention).
//
// Orientation chosen so that NO data ever changes lanes between the two products:
//   S^T[key, q] = K[key, :] . Q[q, :]      MFMA A = K rows (LDS), B = Q^T (registers, loaded once)
//   O^T[d,  q] = V^T[d, key] . P^T[key, q] MFMA A = V^T rows (LDS), B = P^T = the S^T accumulators themselves
// With the 16x16 MFMA accumulator layout (col = lane&15 = query, row = 4*(lane>>4)+reg = key) a lane's four
// S^T registers are four consecutive keys of ONE query, which is exactly the k-slot layout the B operand of
// the next MFMA wants (any k permutation is legal as long as the A operand uses the same one, so V^T is read
// with the matching key order).  Softmax row statistics are therefore per-lane + two xor-shuffles (16, 32).
// V is consumed transposed ([head][d][key]); sr_igemm writes it that way (transpose_out), so no transpose
// pass exists anywhere.  fp16: v_mfma_f32_16x16x32_f16, fp32: v_mfma_f32_16x16x4_f32 (exact), fp32 softmax.
#include "sr_common.h"
#include <type_traits>
#include <stdlib.h>

namespace {

constexpr int KV_TILE = 64;     // keys per iteration (4 MFMA key tiles)

// SR ("sum row"): when d is not a multiple of 16 the last V^T tile has spare rows; row d is filled with ones, so the MFMA
// that accumulates O^T also accumulates the softmax denominator (of the SAME fp16-rounded probabilities as the numerator)
// and the per-element v_add of the row sum disappears from the VALU-bound d=40 loop.
// SHORT (Tk <= 2 tiles, the 77-token prompt): a workgroup stages the whole K / V^T ONCE into the two LDS buffers and then walks
// several 128-query blocks (grid-stride over blockIdx.x) with no barrier inside the walk, the next block's Q fragments fetched
// under the current block's work.  With one query block per workgroup the launch was 4096 workgroups that each paid two
// dependent global->register->LDS round trips for 128 x 77 scores of work (53 us at B16 Tq4096 d40).
template <typename T, int DQ, int DT, int QT, bool SR, int MB, bool SHORT = false>
__global__ __launch_bounds__(256, MB) void attn_kernel(const sr_attention_args p) {
  constexpr int EPC = sr_traits<T>::EPC;
  constexpr int NCH = 4 * DQ;                               // 16-B chunks per K row in LDS (zero padded)
  constexpr int KROW = NCH * 16;                            // bytes
  constexpr int VROW = KV_TILE * (int)sizeof(T) + (sizeof(T) == 2 ? 8 : 16);   // padded V^T row (bank spread)
  constexpr int K_BYTES = KV_TILE * KROW;
  constexpr int V_BYTES = DT * 16 * VROW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;
  char* sV = smem + K_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, g4 = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const int bk = p.Bk == 1 ? 0 : b;
  const int d = p.d;
  int q0 = blockIdx.x * (64 * QT) + wv * (16 * QT);

  // ---- Q fragments (B operand): lane (g,c) <- Q[q0 + qt*16 + c][h*d + s*4*EPC + g*EPC ..]
  uint4 qf[QT][DQ];
  auto load_q = [&](uint4 (&dst)[QT][DQ], int qb0) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const int q = qb0 + qt * 16 + c16;
#pragma unroll
      for (int s = 0; s < DQ; ++s) {
        const int di = s * 4 * EPC + g4 * EPC;
        if (q < p.Tq && di < d)
          dst[qt][s] = *(const uint4*)((const T*)p.q + ((int64_t)b * p.Tq + q) * p.q_stride + h * d + di);
        else
          dst[qt][s] = make_uint4(0, 0, 0, 0);
      }
    }
  };
  load_q(qf, q0);

  f32x4 o[DT][QT];
  float mrow[QT], lrow[QT];
  auto reset = [&]() {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) o[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { mrow[qt] = -INFINITY; lrow[qt] = 0.f; }
  };
  reset();
  const float sl2 = p.scale * 1.4426950408889634f;          // fold log2(e): p = exp2(s*sl2 - m)

  const T* kbase = (const T*)p.k + (int64_t)bk * p.Tk * p.k_stride + h * d;
  const T* vbase = (const T*)p.vt + ((int64_t)bk * p.heads + h) * (int64_t)d * p.ldt;
  constexpr int VCH = KV_TILE * (int)sizeof(T) / 16;        // 16-B chunks per V^T row

  // K / V^T tiles: global -> registers one tile AHEAD of use, registers -> LDS (double buffered) after the barrier, so
  // the HBM/L2 latency of tile t+2 hides under the MFMA + softmax work of tile t and there is ONE barrier per tile.
  constexpr int KPT = (KV_TILE * NCH + 255) / 256, VPT = (DT * 16 * VCH + 255) / 256;
  uint4 rk[KPT], rv[VPT];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int idx = tid + i * 256;
      const int key = idx / NCH, ch = idx - key * NCH;
      rk[i] = make_uint4(0, 0, 0, 0);
      if (idx < KV_TILE * NCH && k0 + key < p.Tk && ch * EPC < d) rk[i] = *(const uint4*)(kbase + (int64_t)(k0 + key) * p.k_stride + ch * EPC);
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / VCH, ch = idx - row * VCH;
      const int key = k0 + ch * EPC;
      rv[i] = make_uint4(0, 0, 0, 0);
      if (idx < DT * 16 * VCH && row < d && key < p.ldt) rv[i] = *(const uint4*)(vbase + (int64_t)row * p.ldt + key);
      if constexpr (SR) { if (row == d) rv[i] = make_uint4(0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u); }   // fp16 ones
    }
  };
  auto lstore = [&](int buf) {
    char* bK = sK + buf * (K_BYTES + V_BYTES);
    char* bV = sV + buf * (K_BYTES + V_BYTES);
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int idx = tid + i * 256;
      const int key = idx / NCH, ch = idx - key * NCH;
      const int pch = (NCH == 8) ? (ch ^ (key & 7)) : ch;
      if (idx < KV_TILE * NCH) *(uint4*)(bK + key * KROW + pch * 16) = rk[i];
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int idx = tid + i * 256;
      const int row = idx / VCH, ch = idx - row * VCH;
      if (idx < DT * 16 * VCH) *(uint4*)(bV + row * VROW + ch * 16) = rv[i];
    }
  };
  constexpr bool DB = 2 * (K_BYTES + V_BYTES) <= 144 * 1024;   // double-buffer LDS when it fits (all fp16 shapes)
  int cur = 0, k0 = 0;
    auto tile = [&](auto tail_tag) {
      constexpr bool TAIL = decltype(tail_tag)::value;
      const char* cK = sK + cur * (K_BYTES + V_BYTES);
      const char* cV = sV + cur * (K_BYTES + V_BYTES);

      // ---- S^T = K Q^T
      f32x4 s[4][QT];
  #pragma unroll
      for (int kt = 0; kt < 4; ++kt)
  #pragma unroll
        for (int qt = 0; qt < QT; ++qt) s[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  #pragma unroll
      for (int st = 0; st < DQ; ++st) {
        uint4 kf[4];
  #pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          const int key = kt * 16 + c16, ch = 4 * st + g4;
          const int pch = (NCH == 8) ? (ch ^ (key & 7)) : ch;
          kf[kt] = *(const uint4*)(cK + key * KROW + pch * 16);
        }
  #pragma unroll
        for (int kt = 0; kt < 4; ++kt)
  #pragma unroll
          for (int qt = 0; qt < QT; ++qt) sr_mma(s[kt][qt], kf[kt], qf[qt][st], T());
      }

      // ---- online softmax (per query = per lane column; keys over kt, reg and the 4 lane groups)
  #pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float mx = -INFINITY;
        if constexpr (TAIL) {
  #pragma unroll
          for (int kt = 0; kt < 4; ++kt)
  #pragma unroll
            for (int r = 0; r < 4; ++r)
              if (k0 + kt * 16 + 4 * g4 + r >= p.Tk) s[kt][qt][r] = -INFINITY;
        }
  #pragma unroll
        for (int kt = 0; kt < 4; ++kt)
  #pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][qt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mnew = fmaxf(mrow[qt], mx * sl2);          // running max in the exp2 domain (sl2 > 0)
        const float alpha = __builtin_amdgcn_exp2f(mrow[qt] - mnew);           // first tile: exp2(-inf) = 0
        mrow[qt] = mnew;
        float ps = 0.f;
        const f32x2 sl2v = {sl2, sl2}, nmv = {-mnew, -mnew};
  #pragma unroll
        for (int kt = 0; kt < 4; ++kt)
  #pragma unroll
          for (int r = 0; r < 4; r += 2) {
            f32x2 t = {s[kt][qt][r], s[kt][qt][r + 1]};
            t = __builtin_elementwise_fma(t, sl2v, nmv);          // v_pk_fma_f32: two scores per VALU op
            const float e0 = __builtin_amdgcn_exp2f(t[0]), e1 = __builtin_amdgcn_exp2f(t[1]);   // raw v_exp_f32 (args <= 0)
            s[kt][qt][r] = e0; s[kt][qt][r + 1] = e1;
            if constexpr (!SR) ps += e0 + e1;
          }
        if constexpr (!SR) lrow[qt] = lrow[qt] * alpha + ps;
        if (__any(alpha != 1.0f)) {                            // wave-uniform: most tiles leave every row max unchanged
  #pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            o[dt][qt][0] *= alpha; o[dt][qt][1] *= alpha; o[dt][qt][2] *= alpha; o[dt][qt][3] *= alpha;
          }
        }
      }

      // ---- O^T += V^T P^T
      if constexpr (sizeof(T) == 2) {
  #pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
          uint4 pf[QT];
  #pragma unroll
          for (int qt = 0; qt < QT; ++qt) {
            h16x8 hv;
  #pragma unroll
            for (int r = 0; r < 4; ++r) { hv[r] = (_Float16)s[2 * kp][qt][r]; hv[4 + r] = (_Float16)s[2 * kp + 1][qt][r]; }
            pf[qt] = __builtin_bit_cast(uint4, hv);
          }
  #pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const char* vr = cV + (dt * 16 + c16) * VROW + kp * 64 + g4 * 8;
            const uint2 lo = *(const uint2*)vr;               // keys kp*32 + 4g .. +3
            const uint2 hi = *(const uint2*)(vr + 32);        // keys kp*32 + 16 + 4g .. +3
            const uint4 vf = make_uint4(lo.x, lo.y, hi.x, hi.y);
  #pragma unroll
            for (int qt = 0; qt < QT; ++qt) sr_mma(o[dt][qt], vf, pf[qt], T());
          }
        }
      } else {
  #pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
  #pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const uint4 vf = *(const uint4*)(cV + (dt * 16 + c16) * VROW + kt * 64 + g4 * 16);
  #pragma unroll
            for (int qt = 0; qt < QT; ++qt) sr_mma(o[dt][qt], vf, __builtin_bit_cast(uint4, s[kt][qt]), T());
          }
        }
      }
    };
  // ---- normalise and store: lane holds O^T[d = dt*16 + 4g + r][q = qt*16 + c]
  auto finish = [&]() {
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float l = lrow[qt];
    if constexpr (SR) {
      l = __shfl(o[DT - 1][qt][0], (((d & 15) >> 2) << 4) + c16);   // O^T row d (the ones row) of this lane's query column
    } else {
      l += __shfl_xor(l, 16);
      l += __shfl_xor(l, 32);
    }
    const float inv = 1.0f / l;
    const int q = q0 + qt * 16 + c16;
    if (q >= p.Tq) continue;
    T* orow = (T*)p.o + ((int64_t)b * p.Tq + q) * p.q_stride + h * d;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int di = dt * 16 + 4 * g4;
      if (di >= d) continue;
      if constexpr (sizeof(T) == 2) {
        h16x4 hv = {(_Float16)(o[dt][qt][0] * inv), (_Float16)(o[dt][qt][1] * inv), (_Float16)(o[dt][qt][2] * inv),
                    (_Float16)(o[dt][qt][3] * inv)};
        *(h16x4*)(orow + di) = hv;
      } else {
        *(float4*)(orow + di) = make_float4(o[dt][qt][0] * inv, o[dt][qt][1] * inv, o[dt][qt][2] * inv, o[dt][qt][3] * inv);
      }
    }
  }
  };

  if constexpr (SHORT) {
    static_assert(DB, "the short-sequence walk keeps both tiles resident");
    gload(0);
    lstore(0);
    if (KV_TILE < p.Tk) { gload(KV_TILE); lstore(1); }
    __syncthreads();
    const int qstep = gridDim.x * (64 * QT);
    for (;;) {
      const int qn = q0 + qstep;
      uint4 qnext[QT][DQ];
      const bool more = qn - wv * (16 * QT) < p.Tq;         // (wave-uniform: the block start)
      if (more) load_q(qnext, qn);
      cur = 0; k0 = 0;
      if (p.Tk <= KV_TILE) tile(std::true_type{});
      else { tile(std::false_type{}); cur = 1; k0 = KV_TILE; tile(std::true_type{}); }
      finish();
      if (!more) break;
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int s = 0; s < DQ; ++s) qf[qt][s] = qnext[qt][s];
      q0 = qn;
      reset();
    }
  } else {
    gload(0);
    lstore(0);
    if (KV_TILE < p.Tk) gload(KV_TILE);
    int it = 0;
    for (k0 = 0; k0 < p.Tk; k0 += KV_TILE, ++it) {
      __syncthreads();                                      // tile `it` visible; everyone is done with tile it-1
      cur = DB ? (it & 1) : 0;
      if constexpr (DB) {
        if (k0 + KV_TILE < p.Tk) {
          lstore(cur ^ 1);
          if (k0 + 2 * KV_TILE < p.Tk) gload(k0 + 2 * KV_TILE);
        }
      }
      if (k0 + KV_TILE > p.Tk) tile(std::true_type{}); else tile(std::false_type{});
      if constexpr (!DB) {                                  // one LDS buffer: refill it once every wave is done reading
        if (k0 + KV_TILE < p.Tk) {
          __syncthreads();
          lstore(0);
          if (k0 + 2 * KV_TILE < p.Tk) gload(k0 + 2 * KV_TILE);
        }
      }
    }
    finish();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Software-pipelined fp16 variant (QT = 2) for the small head dims of the 64x64 / 32x32 UNet levels, where the loop is
// bound by the softmax VALU work and by dependency latency, not by MFMA rate (d = 40: 160 MFMA FLOP but ~4 VALU ops per
// score).  Each wave overlaps the two inside ITS OWN instruction stream, because an in-order wave cannot rely on its one
// SIMD partner to fill every stall:
//   phase A(t):  S(t+1) = K(t+1) Q^T  [MFMA]   interleaved with   P(t) = exp2(S(t)*c - m)  -> fp16   [VALU]
//   phase B(t):  O += V(t)^T P(t)     [MFMA]   interleaved with   row max / alpha of S(t+1)           [VALU + 2 shuffles]
// S lives in two register sets that swap roles every tile (loop unrolled by two); K/V tiles go through a 3-slot LDS ring
// (tile t+2 is written while t and t+1 are read) with ONE barrier per tile; global loads run one more tile ahead in
// registers.  The interleave itself is requested with sched_group_barrier (1 MFMA : n VALU).
// LAZY (default): the softmax shift rides inside the QK^T product and is only moved when it has to be.
//   * Q is pre-scaled by scale*log2(e) once (fp16), channel d of every staged K row is 1.0 and channel d of the Q fragment holds
//     -m (the running shift, kept fp16-representable so the MFMA sees exactly the value the bookkeeping uses): the MFMA delivers
//     S' = log2-scaled score - m and the per-score fused multiply-subtract disappears from the VALU-bound loop;
//   * m follows the row maximum lazily: it is moved (O rescaled, the pending S' tile shifted, the fragment rewritten) only when a
//     tile's maximum exceeds the current shift by more than 2^TAU -- probabilities stay <= 256, comfortably inside fp16 / fp32
//     accumulation -- and always after the first tile; softmax is shift invariant, so only rounding differs from the exact-max form;
//   * the two cross-row max reductions are v_permlane16_swap / v_permlane32_swap (VALU) instead of ds_bpermute round trips.
#ifndef SR_ATTN_TRACE
#define SR_ATTN_TRACE 0      // development only: every wave accumulates the shader-clock time of the three parts of an iteration
#endif
#ifndef SR_ATTN_DBG
#define SR_ATTN_DBG 0        // development only: 1 = no per-tile barrier / loads (timing of the compute alone; wrong results),
#endif                       // 2 = exp replaced by a multiply, 3 = both
// STG: tiles per workgroup barrier.  The ring holds 3*STG slots; tile u (slot u % (3*STG)) is written during iteration u - 2*STG
// and read in iterations u-1 (QK^T) and u (PV), the barrier stands at every STG-th iteration: between the last read of a slot's
// previous tile (iteration u - 3*STG) and its rewrite, and between the write and the first read, there is always one.
template <int DQ, int DT, bool SR, int MB, int NTHR = 256, bool LAZY = false, int KTB = 4, int QT = 2, int STG = 1>
__global__ __launch_bounds__(NTHR, MB) void attn_pipe_kernel(const sr_attention_args p) {
  using T = _Float16;
  constexpr int EPC = 8;
  constexpr int KV = 16 * KTB;                               // keys per tile: 64 (KTB 4) or 32 (KTB 2: half the S registers)
  constexpr int NCH = 4 * DQ, KROW = NCH * 16, VROW = KV * 2 + 8, VCH = KV * 2 / 16;
  constexpr int K_BYTES = KV * KROW, V_BYTES = DT * 16 * VROW, TILE_B = K_BYTES + V_BYTES;
  constexpr int KPT = (KV * NCH + NTHR - 1) / NTHR, VPT = (DT * 16 * VCH + NTHR - 1) / NTHR;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, g4 = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const int bk = p.Bk == 1 ? 0 : b;
  const int d = p.d, Tk = p.Tk;
  const int q0 = blockIdx.x * ((NTHR / 64) * 16 * QT) + wv * (16 * QT);
  const int NT = (Tk + KV - 1) / KV;

  uint4 qf[QT][DQ];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = q0 + qt * 16 + c16;
#pragma unroll
    for (int st = 0; st < DQ; ++st) {
      const int di = st * 4 * EPC + g4 * EPC;
      qf[qt][st] = make_uint4(0, 0, 0, 0);
      if (q < p.Tq && di < d) qf[qt][st] = *(const uint4*)((const T*)p.q + ((int64_t)b * p.Tq + q) * p.q_stride + h * d + di);
    }
  }
  f32x4 o[DT][QT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) o[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float mrow[QT], lrow[QT], alpha[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) { mrow[qt] = LAZY ? 0.f : -INFINITY; lrow[qt] = 0.f; alpha[qt] = 1.f; }
  const float sl2 = p.scale * 1.4426950408889634f;
  // LAZY: where channel d sits in the Q fragment (d is a multiple of 8 and < 32*DQ: st = d/32, lane group (d%32)/8, element 0)
  const int m_st = d >> 5, m_g4 = (d & 31) >> 3;
  float delta[QT];                                           // pending move of the shift (0 = none)
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) delta[qt] = 0.f;
  bool first_tile = true;
  constexpr float TAU = 8.0f;
  if constexpr (LAZY) {
    const _Float16 hs = (_Float16)sl2;
    const h16x8 hsv = {hs, hs, hs, hs, hs, hs, hs, hs};
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int st = 0; st < DQ; ++st) qf[qt][st] = __builtin_bit_cast(uint4, __builtin_bit_cast(h16x8, qf[qt][st]) * hsv);
  }

  // ---- per-thread load slots (loop invariant): which K / V^T chunk this thread moves every tile.  Sources are a workgroup-uniform
  // base (SGPR pair) + a 32-bit byte offset per lane: no 64-bit address arithmetic or pointer registers in the loop
  const char* kbase = (const char*)((const T*)p.k + (int64_t)bk * Tk * p.k_stride + h * d);
  const char* vbase = (const char*)((const T*)p.vt + ((int64_t)bk * p.heads + h) * (int64_t)d * p.ldt);
  const unsigned krow_b = (unsigned)p.k_stride * 2u;
  int k_key[KPT], k_lds[KPT]; unsigned k_boff[KPT]; bool k_ok[KPT], k_one[KPT];
#pragma unroll
  for (int i = 0; i < KPT; ++i) {
    const int idx = tid + i * NTHR, key = idx / NCH, ch = idx - key * NCH;
    k_key[i] = key; k_boff[i] = (unsigned)(ch * EPC) * 2u;
    k_one[i] = LAZY && idx < KV * NCH && ch * EPC == d;     // the shift channel: 1.0 in element 0
    k_ok[i] = idx < KV * NCH && ch * EPC < d;
    k_lds[i] = idx < KV * NCH ? key * KROW + ((NCH == 8) ? (ch ^ (key & 7)) : ch) * 16 : -1;
  }
  unsigned v_boff[VPT]; int v_lds[VPT], v_k8[VPT]; bool v_ok[VPT], v_one[VPT];
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int idx = tid + i * NTHR, row = idx / VCH, ch = idx - row * VCH;
    v_ok[i] = idx < DT * 16 * VCH && row < d;
    v_one[i] = SR && row == d;
    v_lds[i] = idx < DT * 16 * VCH ? K_BYTES + row * VROW + ch * 16 : -1;
    v_k8[i] = ch * EPC;
    v_boff[i] = (unsigned)(row < d ? row : 0) * (unsigned)p.ldt * 2u;
  }
  // The padding chunks of a tile (channels >= d of K with the LAZY shift channel = 1.0, rows >= d of V^T with the ones row) are the
  // same for every tile: written once into every ring slot here, so that in the loop only lanes with real data load and store.
  // (Merging a constant into the loaded register instead made hipcc wait for the load right where it was issued -- the merged
  // value lived in non-consecutive registers -- which exposed one full L2 round trip per tile: 750 -> 657 us at B16 T4096 d40.)
  constexpr int NS = 3 * STG;
#pragma unroll
  for (int sl = 0; sl < NS; ++sl) {
    char* bs = smem + sl * TILE_B;
#pragma unroll
    for (int i = 0; i < KPT; ++i)
      if (k_lds[i] >= 0 && !k_ok[i]) *(uint4*)(bs + k_lds[i]) = k_one[i] ? make_uint4(0x00003C00u, 0, 0, 0) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < VPT; ++i)
      if (v_lds[i] >= 0 && !v_ok[i])
        *(uint4*)(bs + v_lds[i]) = v_one[i] ? make_uint4(0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u) : make_uint4(0, 0, 0, 0);
  }
  uint4 rk[KPT], rv[VPT];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int key = min(k0 + k_key[i], Tk - 1);            // rows past Tk: any valid row, their scores are masked to -inf
      if (k_ok[i]) rk[i] = *(const uint4*)(kbase + ((unsigned)key * krow_b + k_boff[i]));
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int key = min(k0 + v_k8[i], p.ldt - EPC);        // columns past Tk meet p = 0
      if (v_ok[i]) rv[i] = *(const uint4*)(vbase + (v_boff[i] + (unsigned)key * 2u));
    }
  };
  auto lstore = [&](int slot) {
    char* bs = smem + slot * TILE_B;
#pragma unroll
    for (int i = 0; i < KPT; ++i) if (k_ok[i]) *(uint4*)(bs + k_lds[i]) = rk[i];
#pragma unroll
    for (int i = 0; i < VPT; ++i) if (v_ok[i]) *(uint4*)(bs + v_lds[i]) = rv[i];
  };

  // ---- pipeline pieces
  auto qk = [&](f32x4 (&s)[KTB][QT], int slot) {
    const char* cK = smem + slot * TILE_B;
#pragma unroll
    for (int kt = 0; kt < KTB; ++kt)
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) s[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < DQ; ++st) {
      uint4 kf[KTB];
#pragma unroll
      for (int kt = 0; kt < KTB; ++kt) {
        const int key = kt * 16 + c16, ch = 4 * st + g4;
        kf[kt] = *(const uint4*)(cK + key * KROW + ((NCH == 8) ? (ch ^ (key & 7)) : ch) * 16);
      }
#pragma unroll
      for (int kt = 0; kt < KTB; ++kt)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) sr_mma(s[kt][qt], kf[kt], qf[qt][st], T());
    }
  };
  // max over the 4 lanes (g4 = 0..3) that hold one query: two permlane swaps + two max (VALU only)
  auto max_over_g4 = [&](float x) -> float {
    if constexpr (LAZY) {
      auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
      x = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
      auto c = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
      return fmaxf(__uint_as_float(c[0]), __uint_as_float(c[1]));
    } else {
      x = fmaxf(x, __shfl_xor(x, 16));
      return fmaxf(x, __shfl_xor(x, 32));
    }
  };
  // row max of a fresh S tile -> running max, alpha (applied to O and l at the start of the next iteration)
  auto rowmax = [&](f32x4 (&s)[KTB][QT], int k0, bool mask) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      if (mask) {
#pragma unroll
        for (int kt = 0; kt < KTB; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (k0 + kt * 16 + 4 * g4 + r >= Tk) s[kt][qt][r] = -INFINITY;
      }
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < KTB; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][qt][r]);
      mx = max_over_g4(mx);
      if constexpr (LAZY) {
        // s already is (log2 score - m): move m only if this tile tops it by more than TAU (always after the first tile)
        float dl = 0.f;
        if (first_tile || mx > TAU) {
          const float mnew = (float)(_Float16)(mrow[qt] + mx);     // keep the shift fp16 representable
          dl = mnew - mrow[qt];
          mrow[qt] = mnew;
        }
        delta[qt] = dl;
      } else {
        const float mnew = fmaxf(mrow[qt], mx * sl2);
        alpha[qt] = __builtin_amdgcn_exp2f(mrow[qt] - mnew);
        mrow[qt] = mnew;
      }
    }
    first_tile = false;
  };
  // P = exp2(S*c - m) packed to the fp16 B-operand layout of the PV product
  auto expo = [&](f32x4 (&s)[KTB][QT], uint4 (&pf)[KTB / 2][QT]) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const f32x2 sl2v = {sl2, sl2}, nmv = {-mrow[qt], -mrow[qt]};
      float ps = 0.f;
#pragma unroll
      for (int kt = 0; kt < KTB; ++kt)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          f32x2 t = {s[kt][qt][r], s[kt][qt][r + 1]};
          if constexpr (!LAZY) t = __builtin_elementwise_fma(t, sl2v, nmv);
          const float e0 = (SR_ATTN_DBG & 2) ? t[0] * 0.001f : __builtin_amdgcn_exp2f(t[0]), e1 = (SR_ATTN_DBG & 2) ? t[1] * 0.001f : __builtin_amdgcn_exp2f(t[1]);
          s[kt][qt][r] = e0; s[kt][qt][r + 1] = e1;
          if constexpr (!SR) ps += e0 + e1;
        }
      if constexpr (!SR) lrow[qt] += ps;
#pragma unroll
      for (int kp = 0; kp < KTB / 2; ++kp) {
        h16x8 hv;
#pragma unroll
        for (int r = 0; r < 4; ++r) { hv[r] = (_Float16)s[2 * kp][qt][r]; hv[4 + r] = (_Float16)s[2 * kp + 1][qt][r]; }
        pf[kp][qt] = __builtin_bit_cast(uint4, hv);
      }
    }
  };
  auto rescale = [&]() {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      if constexpr (!SR) lrow[qt] *= alpha[qt];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) { o[dt][qt][0] *= alpha[qt]; o[dt][qt][1] *= alpha[qt]; o[dt][qt][2] *= alpha[qt]; o[dt][qt][3] *= alpha[qt]; }
    }
  };
  auto pv = [&](const uint4 (&pf)[KTB / 2][QT], int slot) {
    const char* cV = smem + slot * TILE_B + K_BYTES;
#pragma unroll
    for (int kp = 0; kp < KTB / 2; ++kp) {
      uint4 vf[DT];                                          // one burst of V^T fragment reads per key half, then the MFMAs
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const char* vr = cV + (dt * 16 + c16) * VROW + kp * 64 + g4 * 8;
        const uint2 lo = *(const uint2*)vr;
        const uint2 hi = *(const uint2*)(vr + 32);
        vf[dt] = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) sr_mma(o[dt][qt], vf[dt], pf[kp][qt], T());
    }
  };

  // ---- prologue: tiles 0..2*STG into the ring, tile 2*STG+1 in registers, S(0) and its row max
#pragma unroll
  for (int u = 0; u <= 2 * STG; ++u)
    if (u < NT) { gload(u * KV); lstore(u); }
  if (2 * STG + 1 < NT) gload((2 * STG + 1) * KV);
  __syncthreads();
  f32x4 sA[KTB][QT], sB[KTB][QT];
  qk(sA, 0);
  rowmax(sA, 0, NT == 1);

#if SR_ATTN_TRACE
  long long trc[3] = {0, 0, 0};
#endif
  int slot = 0;                                              // ring slot of tile t
  // STEADY: tile t+1 exists and is a full tile (compile-time straight-line body); otherwise wave-uniform run-time flags
  auto iter = [&](f32x4 (&sc)[KTB][QT], f32x4 (&sn)[KTB][QT], int t, auto steady_tag) {
    constexpr bool STEADY = decltype(steady_tag)::value;
    const bool has_next = STEADY || t + 1 < NT, mask = !STEADY && t + 2 == NT;
    const int s1 = slot == NS - 1 ? 0 : slot + 1;            // slot of tile t+1
    const int s2 = slot + 2 * STG >= NS ? slot + 2 * STG - NS : slot + 2 * STG;   // slot of tile t + 2*STG (its old tile: t - STG)
#if SR_ATTN_TRACE
    const long long tr0 = clock64();
#endif
    if (t > 0 && !(SR_ATTN_DBG & 1)) {
      if (STG == 1 || t % STG == 0) __syncthreads();         // every wave is past iteration t-1 (see the slot timing above)
      if (t + 2 * STG < NT) lstore(s2);
      if (t + 2 * STG + 1 < NT) gload((t + 2 * STG + 1) * KV);
    }
#if SR_ATTN_TRACE
    __builtin_amdgcn_sched_barrier(0);
    const long long tr1 = clock64();
#endif
    uint4 pf[KTB / 2][QT];
    // alpha of tile t (from the previous iteration's rowmax); once the running maxima settle every alpha is exactly 1 and
    // the 12 packed multiplies are skipped (wave-uniform branch ahead of the pipelined body, which stays one block)
    if constexpr (LAZY) {
      bool moved = false;
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) moved = moved || delta[qt] != 0.0f;
      if (__any(moved)) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          alpha[qt] = __builtin_amdgcn_exp2f(-delta[qt]);
#pragma unroll
          for (int kt = 0; kt < KTB; ++kt) { sc[kt][qt][0] -= delta[qt]; sc[kt][qt][1] -= delta[qt]; sc[kt][qt][2] -= delta[qt]; sc[kt][qt][3] -= delta[qt]; }
          // channel d of the Q fragment = -m: rewritten in the lanes that hold it
          const unsigned hb = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)(-mrow[qt]));
#pragma unroll
          for (int st = 0; st < DQ; ++st)
            if (st == m_st && g4 == m_g4) qf[qt][st].x = (qf[qt][st].x & 0xffff0000u) | hb;
          delta[qt] = 0.f;
        }
        rescale();
      }
    } else {
      bool resc = false;
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) resc = resc || alpha[qt] != 1.0f;
      if (__any(resc)) rescale();
    }
    if (has_next) qk(sn, s1);
    expo(sc, pf);
    if constexpr (STEADY) {
#pragma unroll
      for (int i = 0; i < KTB * QT * DQ; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // 4 VALU
      }
    }
#if SR_ATTN_TRACE
    __builtin_amdgcn_sched_barrier(0);
    const long long tr2 = clock64();
#endif
    pv(pf, slot);
    if (has_next) rowmax(sn, (t + 1) * KV, mask);
    if constexpr (STEADY) {
#pragma unroll
      for (int i = 0; i < (KTB / 2) * DT * QT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 1);
      }
    }
#if SR_ATTN_TRACE
    __builtin_amdgcn_sched_barrier(0);
    const long long tr3 = clock64();
    trc[0] += tr1 - tr0; trc[1] += tr2 - tr1; trc[2] += tr3 - tr2;
#endif
    slot = s1;
  };
  int t = 0;
  for (; t + 3 < NT; t += 2) {                               // both iterations have a full next tile
    iter(sA, sB, t, std::true_type{});
    iter(sB, sA, t + 1, std::true_type{});
  }
  for (bool cur_a = true; t < NT; ++t, cur_a = !cur_a) {
    if (cur_a) iter(sA, sB, t, std::false_type{});
    else       iter(sB, sA, t, std::false_type{});
  }

#if SR_ATTN_TRACE
  if (lane == 0 && blockIdx.x < 64 && blockIdx.y == 0 && blockIdx.z == 0) {
    long long* o_ = (long long*)((_Float16*)p.k + (int64_t)p.Bk * p.Tk * p.k_stride) + (blockIdx.x * (NTHR / 64) + wv) * 4;   // rows the caller left behind K (tools/trace_attn.py)
    o_[0] = trc[0]; o_[1] = trc[1]; o_[2] = trc[2]; o_[3] = NT;
  }
#endif
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float l;
    if constexpr (SR) {
      l = __shfl(o[DT - 1][qt][0], (((d & 15) >> 2) << 4) + c16);
    } else {
      l = lrow[qt];
      l += __shfl_xor(l, 16);
      l += __shfl_xor(l, 32);
    }
    const float inv = 1.0f / l;
    const int q = q0 + qt * 16 + c16;
    if (q >= p.Tq) continue;
    T* orow = (T*)p.o + ((int64_t)b * p.Tq + q) * p.q_stride + h * d;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int di = dt * 16 + 4 * g4;
      if (di >= d) continue;
      h16x4 hv = {(_Float16)(o[dt][qt][0] * inv), (_Float16)(o[dt][qt][1] * inv), (_Float16)(o[dt][qt][2] * inv),
                  (_Float16)(o[dt][qt][3] * inv)};
      *(h16x4*)(orow + di) = hv;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Occupancy variant for the d = 40 self-attention (fp16, 32 < d < 48, d % 16 != 0, long Tk): what the timelines of the pipelined
// kernel above asked for -- fewer vector-issue cycles per score and more waves per SIMD instead of an in-wave software pipeline.
//   * 32x32x16 MFMAs: S^T tiles of 32 keys x 32 queries (16 accumulator registers), QK^T over three 16-channel steps (48 padded
//     channels instead of 64), O^T as two 32-row tiles; 14 MFMAs per 64 keys x 32 queries instead of 28, each holding the vector
//     issue port for the same 8 cycles.  The accumulators of S^T are again the B operand of the PV product: lane (q = lane & 31,
//     g = lane >> 5) register i holds key (i/4)*8 + 4g + i%4 of its query, so the 16-key step s' of a tile packs registers
//     8s'..8s'+7 and V^T is read in the same key order (two 8-byte pieces per lane).
//   * one S set (no QK^T(t+1) | exp(t) overlap inside a wave), K / V^T tiles brought in by LDS-DMA (no staging registers):
//     <= 168 registers, so three 4-wave workgroups share a CU and fill each other's LDS / dependency waits; they are not
//     synchronised with each other, so the SIMD partners drift apart by themselves.
//   * shift-in-the-MFMA softmax as above (channel d of K = 1.0, of Q = -m, lazy moves), the denominator from the ones row d of V^T.
// One barrier per tile: [wait my DMA of tile t+1] [barrier: tile t+1 visible, everyone is done with tile t] [DMA of tile t+2 into
// the buffer of tile t] [compute tile t+1].
// Measured around it (B16 T4096 d40, 507-524 us): no DMA / barrier at all 457 us; exp2 replaced by a multiply: no change; two 64-key
// tiles per staged buffer and barrier 515 us; tile order rotated per workgroup: no change; four waves per workgroup 534-545 us.
// Per wave and tile (trace build): DMA issue + K reads + QK^T 817 cycles, row max 173, exp2 / pack / PV 650, wait + barrier 363.
typedef float f32x16 __attribute__((ext_vector_type(16)));
// Row r of a 32-key S^T tile holds key pi(r): the rows lane group g packs for PV step s' -- {16s' + 4g + x, 16s' + 8 + 4g + x},
// x = 0..3 -- are keys 8(2s'+g) .. +7, i.e. ONE 16-byte chunk of a V^T row (the K rows are simply staged in that order).
__device__ __forceinline__ int attn32_pi(int r) { return ((r >> 4) * 2 + ((r >> 2) & 1)) * 8 + ((r >> 3) & 1) * 4 + (r & 3); }
// KS: 16-channel steps of QK^T (covers d channels + the shift channel d), OT: 32-row tiles of O^T (covers d rows + the ones row d)
template <int NWAVE, int MINW, int KS = 3, int OT = 2>
__global__ __launch_bounds__(NWAVE * 64, MINW) void attn32_kernel(const sr_attention_args p) {
  using T = _Float16;
  constexpr int NTHR = NWAVE * 64, KV = 64, KCH = (2 * KS <= 8) ? 8 : 16, KROW = KCH * 16, VROW = 128, VR = OT * 32;
  constexpr int K_BYTES = KV * KROW, V_BYTES = VR * VROW, TILE_B = K_BYTES + V_BYTES;
  constexpr int KI = (KV * KCH + NTHR - 1) / NTHR, VI = (VR * 8 + NTHR - 1) / NTHR;    // 16-byte slots per thread and tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q32 = lane & 31, g = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int bk = p.Bk == 1 ? 0 : b;
  const int d = p.d, Tk = p.Tk;
  const int q0 = blockIdx.x * (NWAVE * 32) + wv * 32;
  const int NT = (Tk + KV - 1) / KV;
  const float sl2 = p.scale * 1.4426950408889634f;

  // ---- Q fragments (B operand): lane (q, g) holds channels 16j + 8g .. +7 of its query, pre-scaled; channel d carries -m
  uint4 qf[KS];
  {
    const _Float16 hs = (_Float16)sl2;
    const h16x8 hsv = {hs, hs, hs, hs, hs, hs, hs, hs};
    const int q = q0 + q32;
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      const int ch = 16 * j + 8 * g;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (q < p.Tq && ch < d) v = *(const uint4*)((const T*)p.q + ((int64_t)b * p.Tq + q) * p.q_stride + h * d + ch);
      qf[j] = __builtin_bit_cast(uint4, __builtin_bit_cast(h16x8, v) * hsv);
    }
  }
  const int m_j = d >> 4, m_g = (d & 15) >> 3;               // where channel d sits: k-step d/16, lane group (d%16)/8, element 0
  f32x16 o[OT];
#pragma unroll
  for (int dt = 0; dt < OT; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
  float mrow = 0.f;
  constexpr float TAU = 8.0f;

  // ---- staging by LDS-DMA: 16-byte slot sigma of a tile = (row, s); the slot holds logical chunk c = s ^ ((row >> 1) & 7)
  // (the swizzle that makes the 32-row fragment reads conflict free, applied on the source side); padding chunks and rows are
  // constants written once into both buffers and never overwritten (their lanes are masked out of the DMA)
  const char* kbase = (const char*)((const T*)p.k + (int64_t)bk * Tk * p.k_stride + h * d);
  const char* vbase = (const char*)((const T*)p.vt + ((int64_t)bk * p.heads + h) * (int64_t)d * p.ldt);
  const unsigned krow_b = (unsigned)p.k_stride * 2u, vrow_b = (unsigned)p.ldt * 2u;
  int k_key[KI]; unsigned k_off[KI]; bool k_ok[KI];
  unsigned v_off[VI]; int v_c8[VI]; bool v_ok[VI];
#pragma unroll
  for (int i = 0; i < KI; ++i) {
    const int sg = i * NTHR + tid, row = sg / KCH, sl = sg - row * KCH, c = (sl & ~7) | ((sl & 7) ^ ((row >> 1) & 7));
    k_key[i] = (row & 32) + attn32_pi(row & 31); k_off[i] = (unsigned)c * 16u; k_ok[i] = sg < KV * KCH && c * 8 < d;
  }
#pragma unroll
  for (int i = 0; i < VI; ++i) {
    const int sg = i * NTHR + tid, row = sg >> 3, sl = sg & 7, c = sl ^ ((row >> 1) & 7);
    v_off[i] = (unsigned)row * vrow_b; v_c8[i] = c * 8; v_ok[i] = sg < VR * 8 && row < d;
  }
#pragma unroll
  for (int bf = 0; bf < 2; ++bf) {
    char* bs = smem + bf * TILE_B;
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      const int sg = i * NTHR + tid, row = sg / KCH, sl = sg - row * KCH, c = (sl & ~7) | ((sl & 7) ^ ((row >> 1) & 7));
      if (sg < KV * KCH && c * 8 >= d) *(uint4*)(bs + sg * 16) = (c * 8 == d) ? make_uint4(0x00003C00u, 0, 0, 0) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < VI; ++i) {
      const int sg = i * NTHR + tid, row = sg >> 3;
      if (sg < VR * 8 && row >= d)
        *(uint4*)(bs + K_BYTES + sg * 16) = (row == d) ? make_uint4(0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u) : make_uint4(0, 0, 0, 0);
    }
  }
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(sr_lds_addr(smem));
  auto dma = [&](int u, int bf) {                            // tile u -> buffer bf
    const int k0 = u * KV;
    const unsigned m0keep = sr_m0_save();
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      const int key = min(k0 + k_key[i], Tk - 1);            // rows past Tk: any valid row, their scores are masked
      const unsigned la = lds0 + bf * TILE_B + (i * NTHR + wv * 64) * 16;
      if (k_ok[i]) sr_glds16_asm_saddr((unsigned)key * krow_b + k_off[i], kbase, la);
    }
#pragma unroll
    for (int i = 0; i < VI; ++i) {
      const int key = min(k0 + v_c8[i], p.ldt - 8);          // columns past Tk meet p = 0
      const unsigned la = lds0 + bf * TILE_B + K_BYTES + (i * NTHR + wv * 64) * 16;
      if (v_ok[i]) sr_glds16_asm_saddr(v_off[i] + (unsigned)key * 2u, vbase, la);
    }
    sr_m0_restore(m0keep);
  };

  // ---- fragment read offsets (bytes inside a tile buffer); +32 rows leave (row >> 1) & 7 unchanged, so kt / dt are plain offsets
  const int swz = (q32 >> 1) & 7;
  int k_rd[KS], v_rd[4];                                      // K: row kt*32 + q32, chunk 2j + g;  V^T: row dt*32 + q32, chunk kt*4 + 2sp + g
#pragma unroll
  for (int j = 0; j < KS; ++j) k_rd[j] = q32 * KROW + ((((2 * j + g) & ~7) | (((2 * j + g) & 7) ^ swz)) * 16);
#pragma unroll
  for (int c = 0; c < 4; ++c) v_rd[c] = q32 * VROW + (((2 * (c & 1) + g + 4 * (c >> 1)) ^ swz) * 16);   // c = kt*2 + sp

#if SR_ATTN_TRACE
  long long trc[4] = {0, 0, 0, 0};
#endif
  auto tile = [&](int t, auto masked_tag) {
    constexpr bool MASKED = decltype(masked_tag)::value;
#if SR_ATTN_TRACE
    const long long a0 = clock64();
#endif
    const int bf = t & 1;
#ifndef SR_A32_DBG
#define SR_A32_DBG 0          // development only: 1 = no per-tile DMA / barrier (timing of the compute alone), 2 = no exp2
#endif
    if (t + 1 < NT && !(SR_A32_DBG & 1)) dma(t + 1, bf ^ 1);
    const char* cK = smem + bf * TILE_B;
    const char* cV = cK + K_BYTES;
    // ---- S^T = K Q^T (two 32-key tiles)
    f32x16 sacc[2];
    {
      uint4 kf[2][KS];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int j = 0; j < KS; ++j) kf[kt][j] = *(const uint4*)(cK + kt * 32 * KROW + k_rd[j]);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc[kt][i] = 0.f;
#pragma unroll
      for (int j = 0; j < KS; ++j)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
          sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, kf[kt][j]), __builtin_bit_cast(h16x8, qf[j]), sacc[kt], 0, 0, 0);
    }
#if SR_ATTN_TRACE
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 0" :: "v"(sacc[0][0]), "v"(sacc[1][15]));
    const long long a1 = clock64();
#endif
    // ---- row max / lazy shift / exp2 / pack
    if constexpr (MASKED) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (t * KV + kt * 32 + attn32_pi((i >> 2) * 8 + 4 * g + (i & 3)) >= Tk) sacc[kt][i] = -INFINITY;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sacc[kt][i]);
    {
      auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    if (t == 0 || mx > TAU) {                                // (per-lane condition, identical in the two lanes of a query)
      const float mnew = (float)(_Float16)(mrow + mx);       // fp16 representable: the MFMA sees exactly this value
      const float dl = mnew - mrow;
      mrow = mnew;
      const float alpha = __builtin_amdgcn_exp2f(-dl);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc[kt][i] -= dl;
#pragma unroll
      for (int dt = 0; dt < OT; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
      const unsigned hb = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)(-mrow));
#pragma unroll
      for (int j = 0; j < KS; ++j)
        if (j == m_j && g == m_g) qf[j].x = (qf[j].x & 0xffff0000u) | hb;
    }
#if SR_ATTN_TRACE
    __builtin_amdgcn_sched_barrier(0);
    const long long a2 = clock64();
#endif
    // exp2 / pack of key tile 0, then PV of key tile 0 interleaved with exp2 / pack of key tile 1, then PV of key tile 1.
    // Step (kt, sp) of lane group g covers keys kt*32 + 8(2sp + g) .. +7 = one 16-byte chunk of a V^T row.
    uint4 pf[2][2], vf[2][2][OT];
    auto pack = [&](int kt) {
#pragma unroll
      for (int sp = 0; sp < 2; ++sp) {
        h16x8 hv;
#pragma unroll
        for (int e = 0; e < 8; ++e) hv[e] = (_Float16)((SR_A32_DBG & 2) ? sacc[kt][8 * sp + e] * 0.001f : __builtin_amdgcn_exp2f(sacc[kt][8 * sp + e]));
        pf[kt][sp] = __builtin_bit_cast(uint4, hv);
      }
    };
    auto vread = [&](int kt) {
#pragma unroll
      for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int dt = 0; dt < OT; ++dt) vf[kt][sp][dt] = *(const uint4*)(cV + dt * 32 * VROW + v_rd[kt * 2 + sp]);
    };
    auto pvmma = [&](int kt) {
#pragma unroll
      for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int dt = 0; dt < OT; ++dt)
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, vf[kt][sp][dt]), __builtin_bit_cast(h16x8, pf[kt][sp]), o[dt], 0, 0, 0);
    };
    vread(0);
    pack(0);
    vread(1);
    __builtin_amdgcn_sched_barrier(0);
    pvmma(0);
    pack(1);
#pragma unroll
    for (int i = 0; i < 2 * OT; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, (24 + 2 * OT - 1) / (2 * OT), 0);     // its share of the 16 exp + 8 pack
    }
    __builtin_amdgcn_sched_barrier(0);
    pvmma(1);
#if SR_ATTN_TRACE
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 0" :: "v"(o[0][0]), "v"(o[1][15]));
    const long long a3 = clock64();
#endif
    if (!(SR_A32_DBG & 1)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // my pieces of tile t+1 have landed
      __syncthreads();                                       // ... everyone's have, and everyone is done with tile t
    }
#if SR_ATTN_TRACE
    const long long a4 = clock64();
    trc[0] += a1 - a0; trc[1] += a2 - a1; trc[2] += a3 - a2; trc[3] += a4 - a3;
#endif
  };

  dma(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int NFULL = (Tk % KV) ? NT - 1 : NT;                 // the ragged tail tile (if any) runs the masked body
  for (int t = 0; t < NFULL; ++t) tile(t, std::false_type{});
  if (NFULL < NT) tile(NT - 1, std::true_type{});

#if SR_ATTN_TRACE
  if (lane == 0 && blockIdx.x < 16 && blockIdx.y == 0 && blockIdx.z == 0) {
    long long* o_ = (long long*)((_Float16*)p.k + (int64_t)p.Bk * p.Tk * p.k_stride) + (blockIdx.x * NWAVE + wv) * 5;
    o_[0] = trc[0]; o_[1] = trc[1]; o_[2] = trc[2]; o_[3] = trc[3]; o_[4] = NT;
  }
#endif
  // ---- output: O^T[dt] register i of lane (q, g) is channel dt*32 + (i/4)*8 + 4g + i%4; the denominator is channel d
  const int den_i = ((d & 31) >> 3) * 4 + (d & 3), den_g = (d & 7) >> 2, den_dt = d >> 5;
  float l = 1.f;
#pragma unroll
  for (int dt = 0; dt < OT; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (dt == den_dt && i == den_i) l = o[dt][i];
  l = __shfl(l, den_g * 32 + q32);
  const float inv = 1.0f / l;
  const int q = q0 + q32;
  if (q < p.Tq) {
    T* orow = (T*)p.o + ((int64_t)b * p.Tq + q) * p.q_stride + h * d;
#pragma unroll
    for (int dt = 0; dt < OT; ++dt)
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) {
        const int di = dt * 32 + i4 * 8 + 4 * g;
        if (di >= d) continue;
        h16x4 hv = {(_Float16)(o[dt][4 * i4] * inv), (_Float16)(o[dt][4 * i4 + 1] * inv), (_Float16)(o[dt][4 * i4 + 2] * inv),
                    (_Float16)(o[dt][4 * i4 + 3] * inv)};
        *(h16x4*)(orow + di) = hv;
      }
  }
}

template <int NWAVE, int MINW, int KS = 3, int OT = 2>
int launch_attn32(const sr_attention_args& a, hipStream_t st) {
  dim3 grid(sr_cdiv(a.Tq, NWAVE * 32), a.heads, a.B);
  constexpr int lds = 2 * (64 * ((2 * KS <= 8) ? 128 : 256) + OT * 32 * 128);
  auto k = attn32_kernel<NWAVE, MINW, KS, OT>;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_set = true; }
  hipLaunchKernelGGL(k, grid, dim3(NWAVE * 64), lds, st, a);
  SR_CHECK_LAUNCH("sr_attention");
  return SR_OK;
}

template <int DQ, int DT, bool SR, int MB = 2, int NTHR = 256, bool LAZY = false, int KTB = 4, int QT = 2, int STG = 1>
int launch_pipe(const sr_attention_args& a, hipStream_t st) {
  dim3 grid(sr_cdiv(a.Tq, (NTHR / 64) * 16 * QT), a.heads, a.B);
  constexpr int KV = 16 * KTB;
  constexpr int lds = 3 * STG * (KV * 4 * DQ * 16 + DT * 16 * (KV * 2 + 8));
  auto k = attn_pipe_kernel<DQ, DT, SR, MB, NTHR, LAZY, KTB, QT, STG>;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_set = true; }
  hipLaunchKernelGGL(k, grid, dim3(NTHR), lds, st, a);
  SR_CHECK_LAUNCH("sr_attention");
  return SR_OK;
}

template <typename T, int DQ, int DT, int QT, bool SR = false, int MB = ((QT <= 2 && DT <= 5 && sizeof(T) == 2) ? 2 : 1)>
int launch_short(const sr_attention_args& a, hipStream_t st) {
  constexpr int tile_b = KV_TILE * 4 * DQ * 16 + DT * 16 * (KV_TILE * (int)sizeof(T) + (sizeof(T) == 2 ? 8 : 16));
  constexpr int lds = 2 * tile_b;
  const int nqb = sr_cdiv(a.Tq, 64 * QT);
  int gx = sr_cdiv(nqb, 8);                                  // ~8 query blocks per workgroup, but keep >= 2 workgroups per CU
  while (gx < nqb && (int64_t)gx * a.heads * a.B < 512) ++gx;
  auto k = attn_kernel<T, DQ, DT, QT, SR, MB, true>;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_set = true; }
  hipLaunchKernelGGL(k, dim3(gx, a.heads, a.B), dim3(256), lds, st, a);
  SR_CHECK_LAUNCH("sr_attention");
  return SR_OK;
}

template <typename T, int DQ, int DT, int QT, bool SR = false, int MB = ((QT <= 2 && DT <= 5 && sizeof(T) == 2) ? 2 : 1)>
int launch(const sr_attention_args& a, hipStream_t st) {
  dim3 grid(sr_cdiv(a.Tq, 64 * QT), a.heads, a.B);
  constexpr int tile_b = KV_TILE * 4 * DQ * 16 + DT * 16 * (KV_TILE * (int)sizeof(T) + (sizeof(T) == 2 ? 8 : 16));
  constexpr int lds = (2 * tile_b <= 144 * 1024) ? 2 * tile_b : tile_b;
  auto k = attn_kernel<T, DQ, DT, QT, SR, MB>;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_set = true; }
  hipLaunchKernelGGL(k, grid, dim3(256), lds, st, a);
  SR_CHECK_LAUNCH("sr_attention");
  return SR_OK;
}

}  // namespace

extern "C" int sr_attention(const sr_attention_args* a, void* stream) {
  if (!a || !a->q || !a->k || !a->vt || !a->o) SR_FAIL(SR_ERR_INVALID, "sr_attention: null pointer");
  if (a->Bk != 1 && a->Bk != a->B) SR_FAIL(SR_ERR_INVALID, "sr_attention: Bk must be 1 or B");
  const int epc = a->dtype == SR_F16 ? 8 : 4;
  if (a->d % epc || a->q_stride % epc || a->k_stride % epc || a->ldt % epc || a->ldt < a->Tk)
    SR_FAIL(SR_ERR_INVALID, "sr_attention: d/strides must be multiples of %d and ldt >= Tk", epc);
  hipStream_t st = sr_stream(stream);
  const int d = a->d;
  if (a->dtype == SR_F16) {
    if (d <= 16) return launch<_Float16, 1, 1, 4>(*a, st);
    if (d <= 32) return launch<_Float16, 1, 2, 4>(*a, st);
    // QT = 2 (32 queries per wave): everything fits in 128 VGPRs -> no AGPR<->VGPR copies around the softmax and two
    // workgroups per CU, so one wave's MFMAs overlap the other's exp/max/sum VALU work
    static const bool pipe = !(getenv("SR_ATTN_PIPE") && atoi(getenv("SR_ATTN_PIPE")) == 0);   // A-B aid
    static const bool short_on = !(getenv("SR_ATTN_SHORT") && atoi(getenv("SR_ATTN_SHORT")) == 0);   // A-B aid
    const bool shortk = short_on && a->Tk <= 2 * KV_TILE && a->Tq >= 512;
    if (d <= 48) {
      // long key sequences (64x64 self-attention): the software-pipelined loop, 819 vs 895 us (Bk=1) / 722 vs 815 (Bk=B) at
      // B16 T4096 d40; short ones (the 77-token prompt) stay on the simple loop, whose prologue is cheaper
      if (pipe && a->Tk >= 512) {
        static const int mb = getenv("SR_ATTN_MB") ? atoi(getenv("SR_ATTN_MB")) : 2;      // tuning aid
        if (mb == 3 && (d & 15)) return launch_pipe<2, 3, true, 3>(*a, st);
        // eight waves per workgroup (one workgroup per CU instead of two of four waves): every staged K / V^T tile serves
        // twice as many queries, 733 vs 791 us at B16 T4096 d40 on the same box
        static const int nthr = getenv("SR_ATTN_NTHR") ? atoi(getenv("SR_ATTN_NTHR")) : 512;   // tuning aid
        static const bool lazy = !(getenv("SR_ATTN_LAZY") && atoi(getenv("SR_ATTN_LAZY")) == 0);   // A-B aid
        static const int ktb = getenv("SR_ATTN_KTB") ? atoi(getenv("SR_ATTN_KTB")) : 4;           // A-B aid: keys per tile / 16
        // (measured and dropped: 64 queries per wave at one wave per SIMD -- QT = 4 -- needs 512 registers and still spills 1 KB;
        //  3 or 4 waves per SIMD spill 190-700 B; 32-key tiles (KTB = 2) fit in 218 registers without a spill but double the
        //  barriers: 847 vs 744 us.  Timing with the per-tile barrier + loads compiled out: 551 us of the 752, with v_exp_f32 replaced by
        //  a multiply: no change -> the loop is bound by the tile hand-off and by LDS-read / MFMA dependency waits at two waves per
        //  SIMD, not by the transcendental rate; profiles/r02_attention_pmc.txt.  After the staging fix above (657 us): rotating the
        //  tile order per workgroup 665, s_setprio 1 for waves 4-7 655, all V^T fragments of a tile read under the QK^T / exp phase
        //  699 -- the hand-off is no longer what binds; the vector issue sum of a wave-tile (28 MFMA x 8 + 32 v_exp x 8 + ~55 plain
        //  x 4 = 700 cycles) is used to 50 % by the two lock-stepped waves of a SIMD.  A ping-pong form of the same loop -- waves 0-3
        //  in a matrix-only phase (QK^T of tile t+1, PV of tile t, all 14 LDS fragment reads issued up front) while waves 4-7 are in
        //  the vector-only phase (row max, lazy shift, exp2, pack, staging) and vice versa, one s_barrier per phase -- is parity clean
        //  and slower: 725 us.  With the vector phase compiled out it takes 481 us, without the staging 593 us: the vector work of one
        //  wave does NOT hide under the partner's MFMAs, because every MFMA holds the SIMD's vector issue port for 8 of its 16 cycles;
        //  both forms share the issue floor of ~1430 cycles per pair of wave-tiles (= 330 us), and what is left to remove is issue
        //  cost itself: 32x32x16 MFMAs (half the holds per FLOP) and fewer max / pack instructions -- a different fragment layout.
        //  Also parity clean and slower: waves 4-7 taking the per-tile barrier in the middle of their iteration (ring of 4 slots), so
        //  that the two waves of a SIMD run half a tile apart: 686 vs 648 us.)
        static const int stg = getenv("SR_ATTN_STG") ? atoi(getenv("SR_ATTN_STG")) : 1;           // A-B aid: tiles per barrier
        // d = 40 (32 < d < 48, not a multiple of 16): the 32x32-MFMA occupancy kernel, 507-516 vs 647-658 us at B16 T4096 (SR_ATTN_32=0:
        // the software-pipelined kernel below; 1/2: four waves per workgroup; 3: eight waves, cap 256 registers)
        static const int a32 = getenv("SR_ATTN_32") ? atoi(getenv("SR_ATTN_32")) : 4;             // A-B aid
        if (a32 && d > 32 && d < 48 && (d & 15) && (d & 7) == 0) {
          if (a32 == 1) return launch_attn32<4, 3>(*a, st);
          if (a32 == 2) return launch_attn32<4, 4>(*a, st);
          if (a32 == 3) return launch_attn32<8, 2>(*a, st);
          return launch_attn32<8, 4>(*a, st);
        }
        if (nthr == 512 && lazy && ktb == 4 && stg == 2) return (d & 15) ? launch_pipe<2, 3, true, 2, 512, true, 4, 2, 2>(*a, st) : launch_pipe<2, 3, false, 2, 512, true, 4, 2, 2>(*a, st);
        if (nthr == 512 && lazy && ktb == 4 && stg == 3) return (d & 15) ? launch_pipe<2, 3, true, 2, 512, true, 4, 2, 3>(*a, st) : launch_pipe<2, 3, false, 2, 512, true, 4, 2, 3>(*a, st);
        if (nthr == 512 && lazy && ktb == 2 && stg == 4) return (d & 15) ? launch_pipe<2, 3, true, 2, 512, true, 2, 2, 4>(*a, st) : launch_pipe<2, 3, false, 2, 512, true, 2, 2, 4>(*a, st);
        if (nthr == 512 && lazy && ktb == 2) return (d & 15) ? launch_pipe<2, 3, true, 2, 512, true, 2>(*a, st) : launch_pipe<2, 3, false, 2, 512, true, 2>(*a, st);
        if (nthr == 512 && lazy) return (d & 15) ? launch_pipe<2, 3, true, 2, 512, true>(*a, st) : launch_pipe<2, 3, false, 2, 512, true>(*a, st);
        if (nthr == 512) return (d & 15) ? launch_pipe<2, 3, true, 2, 512>(*a, st) : launch_pipe<2, 3, false, 2, 512>(*a, st);
        return (d & 15) ? launch_pipe<2, 3, true>(*a, st) : launch_pipe<2, 3, false>(*a, st);
      }
      if (shortk) return (d & 15) ? launch_short<_Float16, 2, 3, 2, true>(*a, st) : launch_short<_Float16, 2, 3, 2>(*a, st);
      static const int smb = getenv("SR_ATTN_SIMPLE_MB") ? atoi(getenv("SR_ATTN_SIMPLE_MB")) : 2;   // probe: occupancy of the simple loop
      if (smb == 3 && (d & 15)) return launch<_Float16, 2, 3, 2, true, 3>(*a, st);
      if (smb == 4 && (d & 15)) return launch<_Float16, 2, 3, 2, true, 4>(*a, st);
      return (d & 15) ? launch<_Float16, 2, 3, 2, true>(*a, st) : launch<_Float16, 2, 3, 2>(*a, st);
    }
    // wider heads with long key sequences (SDXL d = 64 at 64x64 / 32x32, SD1.5 d = 80 at 32x32): the same occupancy kernel with more
    // channel steps / output tiles, 165-168 registers -> three waves per SIMD.  d = 80, B16 T1024 h8: 64.7 vs 91.9 us for the simple
    // loop; d = 64, B16 T4096 h10: 961 vs 1206 us, B16 T1024 h20: 150 vs 186 us.  (A 128-register cap spills: 2x slower.)
    // SR_ATTN_32W: 0 = the simple loop, 1 = four waves per workgroup, 2 = eight; default: 1 for d = 80, 2 for d = 64
    static const int a32w = getenv("SR_ATTN_32W") ? atoi(getenv("SR_ATTN_32W")) : -1;
    if (a32w && a->Tk >= 512) {
      if (d == 64) return a32w == 1 ? launch_attn32<4, 3, 5, 3>(*a, st) : launch_attn32<8, 2, 5, 3>(*a, st);
      if (d == 80) return a32w == 2 ? launch_attn32<8, 2, 6, 3>(*a, st) : launch_attn32<4, 3, 6, 3>(*a, st);
    }
    if (d <= 64) return launch<_Float16, 2, 4, 2>(*a, st);
    if (d <= 80) return shortk ? launch_short<_Float16, 3, 5, 2>(*a, st) : launch<_Float16, 3, 5, 2>(*a, st);
    if (d <= 160) return shortk ? launch_short<_Float16, 5, 10, 2>(*a, st) : launch<_Float16, 5, 10, 2>(*a, st);
  } else if (a->dtype == SR_F32) {
    if (d <= 16) return launch<float, 1, 1, 4>(*a, st);
    if (d <= 32) return launch<float, 2, 2, 4>(*a, st);
    if (d <= 48) return launch<float, 3, 3, 2>(*a, st);
    if (d <= 64) return launch<float, 4, 4, 2>(*a, st);
    if (d <= 80) return launch<float, 5, 5, 2>(*a, st);
    if (d <= 160) return launch<float, 10, 10, 1>(*a, st);
  }
  SR_FAIL(SR_ERR_UNSUPPORTED, "sr_attention: head dim %d / dtype %d unsupported", d, a->dtype);
}
