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

namespace {

constexpr int KV_TILE = 64;     // keys per iteration (4 MFMA key tiles)

template <typename T, int DQ, int DT, int QT>
__global__ __launch_bounds__(256, (QT <= 2 && DT <= 5 && sizeof(T) == 2) ? 2 : 1) void attn_kernel(const sr_attention_args p) {
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
  const int q0 = blockIdx.x * (64 * QT) + wv * (16 * QT);

  // ---- Q fragments (B operand): lane (g,c) <- Q[q0 + qt*16 + c][h*d + s*4*EPC + g*EPC ..]
  uint4 qf[QT][DQ];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = q0 + qt * 16 + c16;
#pragma unroll
    for (int s = 0; s < DQ; ++s) {
      const int di = s * 4 * EPC + g4 * EPC;
      if (q < p.Tq && di < d)
        qf[qt][s] = *(const uint4*)((const T*)p.q + ((int64_t)b * p.Tq + q) * p.q_stride + h * d + di);
      else
        qf[qt][s] = make_uint4(0, 0, 0, 0);
    }
  }

  f32x4 o[DT][QT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) o[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float mrow[QT], lrow[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) { mrow[qt] = -INFINITY; lrow[qt] = 0.f; }
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
  gload(0);
  lstore(0);
  if (KV_TILE < p.Tk) gload(KV_TILE);
  int it = 0;
  for (int k0 = 0; k0 < p.Tk; k0 += KV_TILE, ++it) {
    __syncthreads();                                        // tile `it` visible; everyone is done with tile it-1
    const int cur = DB ? (it & 1) : 0;
    if constexpr (DB) {
      if (k0 + KV_TILE < p.Tk) {
        lstore(cur ^ 1);
        if (k0 + 2 * KV_TILE < p.Tk) gload(k0 + 2 * KV_TILE);
      }
    }
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
  #pragma unroll
        for (int kt = 0; kt < 4; ++kt)
  #pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(fmaf(s[kt][qt][r], sl2, -mnew));   // raw v_exp_f32 (args <= 0)
            s[kt][qt][r] = e;
            ps += e;
          }
        lrow[qt] = lrow[qt] * alpha + ps;
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
    if (k0 + KV_TILE > p.Tk) tile(std::true_type{}); else tile(std::false_type{});
    if constexpr (!DB) {                                    // one LDS buffer: refill it once every wave is done reading
      if (k0 + KV_TILE < p.Tk) {
        __syncthreads();
        lstore(0);
        if (k0 + 2 * KV_TILE < p.Tk) gload(k0 + 2 * KV_TILE);
      }
    }
  }

  // (single-buffer variant: see the end of the loop body)
  // ---- normalise and store: lane holds O^T[d = dt*16 + 4g + r][q = qt*16 + c]
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float l = lrow[qt];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
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
}

template <typename T, int DQ, int DT, int QT>
int launch(const sr_attention_args& a, hipStream_t st) {
  dim3 grid(sr_cdiv(a.Tq, 64 * QT), a.heads, a.B);
  constexpr int tile_b = KV_TILE * 4 * DQ * 16 + DT * 16 * (KV_TILE * (int)sizeof(T) + (sizeof(T) == 2 ? 8 : 16));
  constexpr int lds = (2 * tile_b <= 144 * 1024) ? 2 * tile_b : tile_b;
  auto k = attn_kernel<T, DQ, DT, QT>;
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
    if (d <= 48) return launch<_Float16, 2, 3, 2>(*a, st);
    if (d <= 64) return launch<_Float16, 2, 4, 2>(*a, st);
    if (d <= 80) return launch<_Float16, 3, 5, 2>(*a, st);
    if (d <= 160) return launch<_Float16, 5, 10, 2>(*a, st);
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
