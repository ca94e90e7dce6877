// GroupNorm(+SiLU) and LayerNorm over NHWC activations — HBM-bound streaming kernels (16 B/lane accesses).
//
// GroupNorm: pass 1 accumulates per-channel sum / sum-of-squares in registers (every thread owns a fixed
// 16-byte channel chunk and walks pixels, so loads are fully coalesced along C), folds them to the 32 groups
// in LDS and writes one fp32 partial per (batch, pixel-chunk, group) — no float atomics, so results are
// bit-reproducible.  Pass 2 reduces the partials of its batch entry in a fixed order, builds per-channel
// scale/shift in LDS and applies y = x*sc + sh (+SiLU).  Both passes can read the channel concat of two
// sources, which is how the decoder's th.cat([h, skip]) is consumed without materialising it.
#include "sr_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

// pixels per workgroup: 64 for UNet-sized maps, grows with HW so that a batch entry never has more than 64 chunks
// (every apply block re-reduces its batch entry's partials; 64 chunks keeps that at 16 KB of L2 reads)
// Small batches get up to four times as many (smaller) chunks: with 64 chunks per entry a one-image launch is 64 workgroups on 256 CUs
// (VAE decoder of one view: 0.93 TB/s on the 512 x 512 x 128 map against 4.5 TB/s at eight views; partials then 64 KB per entry).
__host__ __device__ inline int gn_ppc(int HW, int B) {
  const int chunks = B >= 8 ? 64 : B >= 4 ? 128 : 256;
  const int p = (HW + chunks - 1) / chunks;
  return p < 64 ? 64 : p;                                   // (16-pixel chunks for small batches measured the same: two launches bound those)
}

template <typename T>
__device__ __forceinline__ void load_chunk(const T* p, float (&v)[sr_traits<T>::EPC]) {
  if constexpr (sizeof(T) == 2) {
    const h16x8 h = *(const h16x8*)p;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)h[e];
  } else {
    const float4 f = *(const float4*)p;
    v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
  }
}
template <typename T>
__device__ __forceinline__ void store_chunk(T* p, const float (&v)[sr_traits<T>::EPC]) {
  if constexpr (sizeof(T) == 2) {
    h16x8 h;
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = (_Float16)v[e];
    *(h16x8*)p = h;
  } else {
    *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// grid (nchunk, B), block 256.  partials[b][chunk][group][2]
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x1, const T* __restrict__ x2, float* __restrict__ partials,
                                                       int HW, int C1, int C2, int groups, int nchunk) {
  constexpr int EPC = sr_traits<T>::EPC;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int C = C1 + C2, cpt = C / EPC, cpg = C / groups;
  const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const int pp_ = 256 / cpt;
  const int nps = pp_ >= 1 ? pp_ : 1;               // pixel slices held in LDS
  float* chs = (float*)smem_raw;                    // [nps][C] per-channel sums
  float* chq = chs + nps * C;                       // [nps][C] per-channel sums of squares
  const int ppc = gn_ppc(HW, gridDim.y);
  const int p0 = chunk * ppc;
  const int p1 = min(HW, p0 + ppc);
  const int pp = 256 / cpt;                        // pixels processed in parallel when cpt <= 256
  if (pp >= 1) {
    const int cc = tid % cpt, ps = tid / cpt;
    if (ps < pp) {
      float s[EPC], q[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) { s[e] = 0.f; q[e] = 0.f; }
      const int c0 = cc * EPC;
      const T* src = c0 < C1 ? x1 + (int64_t)b * HW * C1 + c0 : x2 + (int64_t)b * HW * C2 + (c0 - C1);
      const int cs = c0 < C1 ? C1 : C2;
      // four pixels per trip: four independent 16-byte loads in flight per lane (the walk is otherwise one load, a dependent
      // accumulate, the next load); the accumulation order per lane is unchanged, so results are too
      int p = p0 + ps;
      for (; p + 3 * pp < p1; p += 4 * pp) {
        float v0[EPC], v1[EPC], v2[EPC], v3[EPC];
        load_chunk<T>(src + (int64_t)p * cs, v0);
        load_chunk<T>(src + (int64_t)(p + pp) * cs, v1);
        load_chunk<T>(src + (int64_t)(p + 2 * pp) * cs, v2);
        load_chunk<T>(src + (int64_t)(p + 3 * pp) * cs, v3);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          s[e] += v0[e]; q[e] += v0[e] * v0[e];
          s[e] += v1[e]; q[e] += v1[e] * v1[e];
          s[e] += v2[e]; q[e] += v2[e] * v2[e];
          s[e] += v3[e]; q[e] += v3[e] * v3[e];
        }
      }
      for (; p < p1; p += pp) {
        float v[EPC];
        load_chunk<T>(src + (int64_t)p * cs, v);
#pragma unroll
        for (int e = 0; e < EPC; ++e) { s[e] += v[e]; q[e] += v[e] * v[e]; }
      }
#pragma unroll
      for (int e = 0; e < EPC; ++e) { chs[ps * C + c0 + e] = s[e]; chq[ps * C + c0 + e] = q[e]; }
    }
  } else {
    for (int cc = tid; cc < cpt; cc += 256) {
      float s[EPC], q[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) { s[e] = 0.f; q[e] = 0.f; }
      const int c0 = cc * EPC;
      const T* src = c0 < C1 ? x1 + (int64_t)b * HW * C1 + c0 : x2 + (int64_t)b * HW * C2 + (c0 - C1);
      const int cs = c0 < C1 ? C1 : C2;
      for (int p = p0; p < p1; ++p) {
        float v[EPC];
        load_chunk<T>(src + (int64_t)p * cs, v);
#pragma unroll
        for (int e = 0; e < EPC; ++e) { s[e] += v[e]; q[e] += v[e] * v[e]; }
      }
#pragma unroll
      for (int e = 0; e < EPC; ++e) { chs[c0 + e] = s[e]; chq[c0 + e] = q[e]; }
    }
  }
  __syncthreads();
  if (tid < groups) {                               // fixed summation order -> bit-reproducible
    float gs = 0.f, gq = 0.f;
    for (int ps = 0; ps < nps; ++ps)
      for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { gs += chs[ps * C + c]; gq += chq[ps * C + c]; }
    float* o = partials + (((int64_t)b * nchunk + chunk) * groups + tid) * 2;
    o[0] = gs; o[1] = gq;
  }
}

// grid (nchunk, B), block 256; dynamic LDS: 2*C floats (scale/shift per channel)
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x1, const T* __restrict__ x2, const float* __restrict__ partials,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ y,
                                                       int HW, int C1, int C2, int groups, int nchunk, float eps, int silu) {
  constexpr int EPC = sr_traits<T>::EPC;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sc = (float*)smem_raw;
  const int C = C1 + C2, cpt = C / EPC, cpg = C / groups;
  float* sh = sc + C;
  float* gm = sh + C;      // [groups] mean, [groups] rstd
  const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const int ppc = gn_ppc(HW, gridDim.y);
  const int p0 = chunk * ppc;
  const int p1 = min(HW, p0 + ppc);
  const int total = (p1 - p0) * cpt;
  auto src_of = [&](int idx, int& c0, int& p) -> const T* {
    p = p0 + idx / cpt;
    c0 = (idx - (idx / cpt) * cpt) * EPC;
    return c0 < C1 ? x1 + ((int64_t)b * HW + p) * C1 + c0 : x2 + ((int64_t)b * HW + p) * C2 + (c0 - C1);
  };
  auto finish = [&](float (&v)[EPC], int c0, int p) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      float t = v[e] * sc[c0 + e] + sh[c0 + e];
      v[e] = silu ? sr_silu_f(t) : t;
    }
    store_chunk<T>(y + ((int64_t)b * HW + p) * C + c0, v);
  };
  // The block's first trip of data loads goes out BEFORE the statistics preamble (partials reduction, two barriers, scale /
  // shift table): at 64 pixels per block the preamble re-reads 16 KB of partials to process 20 KB of activations and used to
  // sit, latency-exposed, in front of the first load (UNet 64x64 maps: 32 -> ~20 us per launch).  The arithmetic is unchanged.
  // (raw 16-byte vectors, converted after the preamble behind an opaque asm so that no wait for them is placed earlier)
  using VEC = typename std::conditional<sizeof(T) == 2, h16x8, f32x4>::type;
  VEC pr[4];
  int pc0[4], pq0[4];
  int pre_n = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    pr[i] = VEC{};
    if (tid + i * 256 < total) { pr[i] = *(const VEC*)src_of(tid + i * 256, pc0[i], pq0[i]); pre_n = i + 1; }
  }
  // fixed-order reduction of this batch entry's partials, spread over the whole block: thread (g, part) sums every
  // 8th chunk, then thread g adds the 8 parts in order (bit-reproducible, no serial latency chain)
  {
    float* red = gm + 2 * groups;                  // [8][groups][2]
    const int g = tid & 31, part = tid >> 5;       // groups <= 32 on this path (asserted by the host), 8 parts
    if (g < groups) {
      float s = 0.f, q = 0.f;
      const float* pp = partials + ((int64_t)b * nchunk * groups + g) * 2;
      for (int i = part; i < nchunk; i += 8) { s += pp[(int64_t)i * groups * 2]; q += pp[(int64_t)i * groups * 2 + 1]; }
      red[(part * groups + g) * 2] = s; red[(part * groups + g) * 2 + 1] = q;
    }
    __syncthreads();
    if (tid < groups) {
      double s = 0.0, q = 0.0;
      for (int p8 = 0; p8 < 8; ++p8) { s += red[(p8 * groups + tid) * 2]; q += red[(p8 * groups + tid) * 2 + 1]; }
      const double cnt = (double)HW * cpg;
      const double mean = s / cnt;
      double var = q / cnt - mean * mean;
      if (var < 0.0) var = 0.0;
      gm[tid] = (float)mean;
      gm[groups + tid] = (float)(1.0 / sqrt(var + (double)eps));
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    const int g = c / cpg;
    const float a = gm[groups + g] * gamma[c];
    sc[c] = a; sh[c] = beta[c] - gm[g] * a;
  }
  __syncthreads();
  int idx = tid;
#pragma unroll
  for (int i = 0; i < 4; ++i) {                            // the trip whose loads were issued before the preamble
    VEC r = pr[i];
    asm volatile("" : "+v"(r));
    if (i < pre_n) {
      float v[EPC];
      load_chunk<T>((const T*)&r, v);
      finish(v, pc0[i], pq0[i]);
    }
  }
  idx += pre_n * 256;
  for (; idx + 3 * 256 < total; idx += 4 * 256) {          // four 16-byte loads in flight per lane
    int c0[4], pq[4];
    float v0[EPC], v1[EPC], v2[EPC], v3[EPC];
    load_chunk<T>(src_of(idx, c0[0], pq[0]), v0);
    load_chunk<T>(src_of(idx + 256, c0[1], pq[1]), v1);
    load_chunk<T>(src_of(idx + 512, c0[2], pq[2]), v2);
    load_chunk<T>(src_of(idx + 768, c0[3], pq[3]), v3);
    finish(v0, c0[0], pq[0]); finish(v1, c0[1], pq[1]); finish(v2, c0[2], pq[2]); finish(v3, c0[3], pq[3]);
  }
  for (; idx < total; idx += 256) {
    int c0, pq;
    float v[EPC];
    load_chunk<T>(src_of(idx, c0, pq), v);
    finish(v, c0, pq);
  }
}

// Single-launch GroupNorm for maps whose (batch entry, group bundle) slab fits the workgroup's registers (all UNet levels
// up to 32x32): grid (groups/GB * B).  A workgroup owns GB consecutive groups (GB*cpg
// channels, a multiple of one 16-byte chunk, contiguous per pixel), reads its slab ONCE into registers, reduces mean and
// then the centred second moment through LDS in a fixed order (bit-reproducible, no E[x^2]-mean^2 cancellation), and
// writes y = (x-mean)*rstd*gamma+beta (+SiLU).  One read + one write of the tensor and one launch instead of two reads,
// a write, a partials round trip and two launches; the two-pass kernels above remain for large maps (VAE).
template <typename T, int NV, int BLOCK>
__global__ __launch_bounds__(BLOCK) void gn_fused_kernel(const T* __restrict__ x1, const T* __restrict__ x2, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, T* __restrict__ y, int HW, int C1, int C2,
                                                         int groups, int GB, float eps, int silu) {
  constexpr int EPC = sr_traits<T>::EPC;
  using VEC = typename std::conditional<sizeof(T) == 2, h16x8, f32x4>::type;   // native vectors: register asm operands
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int C = C1 + C2, cpg = C / groups, span = GB * cpg, vpp = span / EPC, pp = BLOCK / vpp;
  float* chs = (float*)smem_raw;                    // [pp][span] per-thread partials
  float* chsum = chs + pp * span;                   // [span]
  float* gstat = chsum + span;                      // [GB]
  float* strip = gstat + GB;                        // [NS][span] first-level sums
  const int NS = BLOCK / span;                      // strips of the two-level column sum (span <= 256 <= BLOCK)
  // bundles of one batch entry interleave inside the same cache lines: keep them on one XCD (one L2), close in time
  const int nb = groups / GB, wg = sr_xcd_remap(blockIdx.x, gridDim.x);
  const int b = wg / nb, bundle = wg - b * nb, tid = threadIdx.x;
  const int cc = tid % vpp, ps = tid / vpp;
  const bool active = ps < pp;
  const int cl = cc * EPC, c0 = bundle * span + cl;         // chunk offset inside the bundle / absolute channel
  const T* src = c0 < C1 ? x1 + (int64_t)b * HW * C1 + c0 : x2 + (int64_t)b * HW * C2 + (c0 - C1);
  const int cs = c0 < C1 ? C1 : C2;
  VEC raw[NV];
  float acc[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int p = ps + i * pp;
    if (active && p < HW) {
      raw[i] = *(const VEC*)(src + (int64_t)p * cs);
      float v[EPC];
      load_chunk<T>((const T*)&raw[i], v);
#pragma unroll
      for (int e = 0; e < EPC; ++e) acc[e] += v[e];
    }
  }
  const float inv_cnt = 1.0f / ((float)HW * (float)cpg);
  float mean_e[EPC], rstd_e[EPC];
  // two rounds of the same fixed-order reduction: sum -> mean, centred squares -> rstd
  for (int round = 0; round < 2; ++round) {
    if (active) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) chs[ps * span + cl + e] = acc[e];
    }
    __syncthreads();
    // column sums in two fixed-order levels: strip s adds rows s, s + NS, ... (all threads busy), then span threads add the NS
    // strips -- pp / NS + NS dependent adds instead of pp (512 for the 1024-thread form: a third of the kernel's time at small
    // batches, where nothing else runs beside the 16..64 workgroups of a launch)
    if (tid < NS * span) {
      const int c = tid % span, s0 = tid / span;
      float t = 0.f;
      for (int q = s0; q < pp; q += NS) t += chs[q * span + c];
      strip[s0 * span + c] = t;
    }
    __syncthreads();
    if (tid < span) {
      float t = 0.f;
      for (int q = 0; q < NS; ++q) t += strip[q * span + tid];
      chsum[tid] = t;
    }
    __syncthreads();
    if (tid < GB) {
      float t = 0.f;
      for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) t += chsum[c];
      gstat[tid] = round == 0 ? t * inv_cnt : rsqrtf(t * inv_cnt + eps);
    }
    __syncthreads();
    if (round == 0) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) { mean_e[e] = gstat[(cl + e) / cpg]; acc[e] = 0.f; }
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int p = ps + i * pp;
        if (active && p < HW) {
          float v[EPC];
          VEC r = raw[i];
          asm volatile("" : "+v"(r));                // keep the slab packed in registers: no hoisted fp32 copies
          load_chunk<T>((const T*)&r, v);
#pragma unroll
          for (int e = 0; e < EPC; ++e) { const float d = v[e] - mean_e[e]; acc[e] += d * d; }
        }
      }
      __syncthreads();                               // gstat is rewritten by the next round
    } else {
#pragma unroll
      for (int e = 0; e < EPC; ++e) rstd_e[e] = gstat[(cl + e) / cpg];
    }
  }
  if (!active) return;
  float sc[EPC], sh[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    sc[e] = rstd_e[e] * gamma[c0 + e];
    sh[e] = beta[c0 + e] - mean_e[e] * sc[e];
  }
  T* dst = y + (int64_t)b * HW * C + c0;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int p = ps + i * pp;
    if (p < HW) {
      float v[EPC];
      VEC r = raw[i];
      asm volatile("" : "+v"(r));
      load_chunk<T>((const T*)&r, v);
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const float t = v[e] * sc[e] + sh[e];
        v[e] = silu ? sr_silu_f(t) : t;
      }
      store_chunk<T>(dst + (int64_t)p * C, v);
    }
  }
}

template <typename T, int NV, int BLOCK>
void launch_gn_fused(const sr_groupnorm_args* a, int GB, hipStream_t st) {
  constexpr int EPC = sr_traits<T>::EPC;
  const int C = a->C1 + a->C2, span = GB * (C / a->groups), vpp = span / EPC, pp = BLOCK / vpp;
  const size_t lds = (size_t)(pp * span + span + GB + (BLOCK / span) * span) * sizeof(float);
  hipLaunchKernelGGL((gn_fused_kernel<T, NV, BLOCK>), dim3((a->groups / GB) * a->B), dim3(BLOCK), lds, st, (const T*)a->x, (const T*)a->x2,
                     a->gamma, a->beta, (T*)a->y, a->HW, a->C1, a->C2, a->groups, GB, a->eps, a->silu);
}

// -> true when the fused kernel took the job
template <typename T>
bool try_gn_fused(const sr_groupnorm_args* a, hipStream_t st) {
  constexpr int EPC = sr_traits<T>::EPC;
  const int C = a->C1 + a->C2, cpg = C / a->groups;
  int GB = 0;
  for (int g = 1; g <= 8; ++g)
    if ((g * cpg) % EPC == 0 && a->groups % g == 0) { GB = g; break; }
  if (!GB) return false;
  const int vpp = GB * cpg / EPC;
  if (vpp > 256 || GB * cpg > 256) return false;             // (the column sums are taken by `span` threads of a >= 256-thread block)
  const int need256 = sr_cdiv(a->HW, 256 / vpp), need1024 = sr_cdiv(a->HW, 1024 / vpp);
  if (need256 <= 4) launch_gn_fused<T, 4, 256>(a, GB, st);
  else if (need256 <= 8) launch_gn_fused<T, 8, 256>(a, GB, st);
  else if (need1024 <= 8) launch_gn_fused<T, 8, 1024>(a, GB, st);
  else if (need1024 <= 16) launch_gn_fused<T, 16, 1024>(a, GB, st);
  else return false;
  return true;
}

// The same slab-in-registers GroupNorm with the reductions done by wave shuffles: ONE workgroup barrier per moment instead of four
// (write per-thread sums -> strip sums -> column sums -> group statistic), no serial LDS loops, the affine parameters requested
// with the slab.  Written for small batches (a one-view rank of the 8-GPU shard evaluates B = 2): there a launch is 16-64
// workgroups, nothing overlaps a workgroup's dependent chain, and the chain -- not bandwidth -- is the kernel's time
// (B 2, 8x8 x 1280: 12.6 us with the LDS form for 330 KB of traffic).  A thread folds its 8-channel sums to the (at most GB)
// groups they belong to, every group's value is reduced across the wave by xor-shuffles, lane 0 publishes one value per group and
// wave, and after the barrier every thread adds the waves in wave order: fixed order, bit-reproducible.  Mean first, then the
// centred second moment (no E[x^2] - mean^2 cancellation), as above.
template <typename T, int NV, int BLOCK>
__global__ __launch_bounds__(BLOCK) void gn_wave_kernel(const T* __restrict__ x1, const T* __restrict__ x2, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ y, int HW, int C1, int C2,
                                                        int groups, int GB, float eps, int silu) {
  constexpr int EPC = sr_traits<T>::EPC, NW = BLOCK / 64, GMAX = 8;
  using VEC = typename std::conditional<sizeof(T) == 2, h16x8, f32x4>::type;
  __shared__ float wsum[2][NW][GMAX];
  __shared__ float aff[2][512];                              // gamma / beta of the bundle's channels (span <= 512)
  const int C = C1 + C2, cpg = C / groups, span = GB * cpg, vpp = span / EPC, pp = BLOCK / vpp;
  const int nb = groups / GB, wg = sr_xcd_remap(blockIdx.x, gridDim.x);
  const int b = wg / nb, bundle = wg - b * nb, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int cc = tid % vpp, ps = tid / vpp;
  const bool active = ps < pp;
  const int cl = cc * EPC, c0 = bundle * span + cl;
  const T* src = c0 < C1 ? x1 + (int64_t)b * HW * C1 + c0 : x2 + (int64_t)b * HW * C2 + (c0 - C1);
  const int cs = c0 < C1 ? C1 : C2;
  VEC raw[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int p = ps + i * pp;
    if (active && p < HW) raw[i] = *(const VEC*)(src + (int64_t)p * cs);
  }
  for (int c = tid; c < span; c += BLOCK) { aff[0][c] = gamma[bundle * span + c]; aff[1][c] = beta[bundle * span + c]; }   // (visible after the barriers below)
  // the thread's 8 (4) channels lie in at most two groups (cpg >= EPC / 2): g0 and, from element eb on, g0 + 1
  const int g0 = cl / cpg, eb = (g0 + 1) * cpg - cl;
  const float inv_cnt = 1.0f / ((float)HW * (float)cpg);
  float mean0 = 0.f, mean1 = 0.f, rstd0 = 0.f, rstd1 = 0.f;
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int p = ps + i * pp;
      if (active && p < HW) {
        float v[EPC];
        VEC r = raw[i];
        asm volatile("" : "+v"(r));                        // keep the slab packed in registers: no hoisted fp32 copies
        load_chunk<T>((const T*)&r, v);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          if (round == 0) acc[e] += v[e];
          else { const float d = v[e] - (e < eb ? mean0 : mean1); acc[e] += d * d; }
        }
      }
    }
    float t0 = 0.f, t1 = 0.f;
#pragma unroll
    for (int e = 0; e < EPC; ++e) { if (e < eb) t0 += acc[e]; else t1 += acc[e]; }
    for (int g = 0; g < GB; ++g) {                           // (GB is uniform)
      float t = (g == g0 ? t0 : 0.f) + (g == g0 + 1 ? t1 : 0.f);
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off);
      if (lane == 0) wsum[round][wv][g] = t;
    }
    __syncthreads();
    float a0 = 0.f, a1 = 0.f;
    const int g1 = g0 + 1 < GB ? g0 + 1 : g0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { a0 += wsum[round][w][g0]; a1 += wsum[round][w][g1]; }
    if (round == 0) { mean0 = a0 * inv_cnt; mean1 = a1 * inv_cnt; }
    else { rstd0 = rsqrtf(a0 * inv_cnt + eps); rstd1 = rsqrtf(a1 * inv_cnt + eps); }
  }
  if (!active) return;
  float sc[EPC], sh[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    sc[e] = (e < eb ? rstd0 : rstd1) * aff[0][cl + e];
    sh[e] = aff[1][cl + e] - (e < eb ? mean0 : mean1) * sc[e];
  }
  T* dst = y + (int64_t)b * HW * C + c0;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int p = ps + i * pp;
    if (p < HW) {
      float v[EPC];
      VEC r = raw[i];
      asm volatile("" : "+v"(r));
      load_chunk<T>((const T*)&r, v);
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const float t = v[e] * sc[e] + sh[e];
        v[e] = silu ? sr_silu_f(t) : t;
      }
      store_chunk<T>(dst + (int64_t)p * C, v);
    }
  }
}

template <typename T, int NV, int BLOCK>
void launch_gn_wave(const sr_groupnorm_args* a, int GB, hipStream_t st) {
  hipLaunchKernelGGL((gn_wave_kernel<T, NV, BLOCK>), dim3((a->groups / GB) * a->B), dim3(BLOCK), 0, st, (const T*)a->x, (const T*)a->x2,
                     a->gamma, a->beta, (T*)a->y, a->HW, a->C1, a->C2, a->groups, GB, a->eps, a->silu);
}

// -> true when the shuffle-reduced kernel took the job: launches of few workgroups (small batches), where the dependent chain of
// one workgroup is the whole kernel; maps up to 32x32
template <typename T>
bool try_gn_wave(const sr_groupnorm_args* a, hipStream_t st) {
  constexpr int EPC = sr_traits<T>::EPC;
  const int C = a->C1 + a->C2, cpg = C / a->groups;
  int GB = 0;
  for (int g = 1; g <= 8; ++g)
    if ((g * cpg) % EPC == 0 && a->groups % g == 0) { GB = g; break; }
  if (!GB) return false;
  const int vpp = GB * cpg / EPC;
  if (vpp > 64 || GB * cpg > 512 || 2 * cpg < EPC) return false;      // (a thread's chunk may straddle two groups, not more)
  static const int max_wg = getenv("SR_GN_WAVE_MAX_WG") ? atoi(getenv("SR_GN_WAVE_MAX_WG")) : 256;   // tuning / A-B aid (0 = off)
  if ((a->groups / GB) * a->B > max_wg) return false;
  auto need = [&](int block) { return sr_cdiv(a->HW, block / vpp); };
  if (need(64) <= 8) launch_gn_wave<T, 8, 64>(a, GB, st);
  else if (need(256) <= 4) launch_gn_wave<T, 4, 256>(a, GB, st);
  else if (need(256) <= 8) launch_gn_wave<T, 8, 256>(a, GB, st);
  else if (need(1024) <= 8) launch_gn_wave<T, 8, 1024>(a, GB, st);
  else if (need(1024) <= 16) launch_gn_wave<T, 16, 1024>(a, GB, st);
  // (64x64 x 320 maps need 21 chunks per thread of a 1024-thread block: 20-24 spilled registers and 16 workgroups per batch pair --
  //  38 us against 20.5 us for the two-pass kernels at B = 2; left to them)
  else return false;
  return true;
}

// optional row gather in front of the LayerNorm (sr_layernorm_gather): output row r is the LayerNorm of row
// sel[r / frame_rows] * frame_rows + r % frame_rows of x -- the ONE K/V-injected frame of a batch, picked by a device index
// (OverlapCorresponder.pre_atten_inject, corresponder.py:204-214) and normalised in the same launch.  An index outside the batch
// gives zero rows and raises *err, as sr_gather_rows does.
struct ln_gather { const int* sel; int* err; int frame_rows, n_frames; };
__device__ __forceinline__ int64_t ln_src_row(const ln_gather& g, int row, bool& ok) {
  if (!g.sel) return row;
  const int j = row / g.frame_rows, f = g.sel[j];
  if (f < 0 || f >= g.n_frames) {
    ok = false;
    if (g.err) atomicOr(g.err, 1);
    return 0;
  }
  return (int64_t)f * g.frame_rows + (row - j * g.frame_rows);
}

// one wave per row
// One wave per ROWS consecutive rows (MAXC 16-byte chunks per lane and row): with one 640-byte row per wave the kernel is a chain
// load -> two shuffle trees -> store with ~20 KB in flight per CU (3.2 TB/s at rows 65536, C 320); ROWS independent rows per
// wave put ROWS times as many loads in flight.  The arithmetic per row (per-lane partials in chunk order, xor-shuffle tree,
// centred second moment) is unchanged.
template <typename T, int MAXC, int ROWS>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        T* __restrict__ y, int rows, int C, float eps, const ln_gather g) {
  constexpr int EPC = sr_traits<T>::EPC;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS, lane = threadIdx.x & 63;
  if (row0 >= rows) return;
  const int cpt = C / EPC;
  float v[ROWS][MAXC][EPC];
  float s[ROWS], q[ROWS];
  bool okr[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    s[r] = 0.f;
    okr[r] = true;
    const int64_t srow = row0 + r < rows ? ln_src_row(g, row0 + r, okr[r]) : 0;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int cc = lane + i * 64;
      if (cc < cpt && row0 + r < rows) {
        if (okr[r]) load_chunk<T>(x + srow * C + cc * EPC, v[r][i]);
        else { for (int e = 0; e < EPC; ++e) v[r][i][e] = 0.f; }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      if (lane + i * 64 < cpt && row0 + r < rows) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) s[r] += v[r][i][e];
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int r = 0; r < ROWS; ++r) s[r] += __shfl_xor(s[r], o);
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const float mean = s[r] / (float)C;
    s[r] = mean;
    q[r] = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      if (lane + i * 64 < cpt && row0 + r < rows) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) { const float d = v[r][i][e] - mean; q[r] += d * d; }
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int r = 0; r < ROWS; ++r) q[r] += __shfl_xor(q[r], o);
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const float mean = s[r], rstd = rsqrtf(q[r] / (float)C + eps);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int cc = lane + i * 64;
      if (cc < cpt && row0 + r < rows) {
        float o[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) o[e] = okr[r] ? (v[r][i][e] - mean) * rstd * gamma[cc * EPC + e] + beta[cc * EPC + e] : 0.f;
        store_chunk<T>(y + (int64_t)(row0 + r) * C + cc * EPC, o);
      }
    }
  }
}

// Sub-wave rows: LPR lanes per row (a power of two), NCH 16-byte chunks per lane -- for C = LPR * NCH chunks.  The wave-per-row
// form above leaves 24 of 64 lanes idle at C = 320 (40 chunks) and runs two 6-step shuffle trees per row; here a wave owns
// 64 / LPR rows at once (C 320: eight rows of 8 lanes x 5 chunks), every lane is busy, each row's 128-byte pieces are read by
// LPR consecutive lanes (whole lines), NCH independent loads are in flight per lane and the trees are log2(LPR) steps.
// Mean then centred second moment, as above.
template <typename T, int LPR, int NCH>
__global__ __launch_bounds__(256) void layernorm_sub_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            T* __restrict__ y, int rows, int C, float eps, const ln_gather g) {
  constexpr int EPC = sr_traits<T>::EPC, RPW = 64 / LPR;
  using VEC = typename std::conditional<sizeof(T) == 2, h16x8, f32x4>::type;
  const int lane = threadIdx.x & 63, sub = lane & (LPR - 1);
  const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
  const bool on = row < rows;
  bool ok = true;
  const int64_t srow = on ? ln_src_row(g, row, ok) : 0;
  VEC raw[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) raw[i] = (on && ok) ? *(const VEC*)(x + srow * C + (sub + i * LPR) * EPC) : VEC{};
  float v[NCH][EPC];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    load_chunk<T>((const T*)&raw[i], v[i]);
#pragma unroll
    for (int e = 0; e < EPC; ++e) s += v[i][e];
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int e = 0; e < EPC; ++e) { const float d = v[i][e] - mean; q += d * d; }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o);
  if (!on) return;
  const float rstd = rsqrtf(q / (float)C + eps);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c0 = (sub + i * LPR) * EPC;
    float o[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) o[e] = ok ? (v[i][e] - mean) * rstd * gamma[c0 + e] + beta[c0 + e] : 0.f;
    store_chunk<T>(y + (int64_t)row * C + c0, o);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void row_stats_kernel(const T* __restrict__ x, float* __restrict__ stats, int rows, int C, float eps) {
  constexpr int EPC = sr_traits<T>::EPC;
  constexpr int MAXC = 5;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int cpt = C / EPC;
  float v[MAXC][EPC];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int cc = lane + i * 64;
    if (cc < cpt) {
      load_chunk<T>(x + (int64_t)row * C + cc * EPC, v[i]);
#pragma unroll
      for (int e = 0; e < EPC; ++e) s += v[i][e];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int cc = lane + i * 64;
    if (cc < cpt) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) { const float d = v[i][e] - mean; q += d * d; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  if (lane == 0) {
    const float rstd = rsqrtf(q / (float)C + eps);
    *(float2*)(stats + 2 * (int64_t)row) = make_float2(rstd, -rstd * mean);
  }
}

}  // namespace

extern "C" int sr_row_stats(const void* x, float* stats, int32_t rows, int32_t C, float eps, int32_t dtype, void* stream) {
  if (!x || !stats) SR_FAIL(SR_ERR_INVALID, "sr_row_stats: null pointer");
  const int epc = dtype == SR_F16 ? 8 : 4;
  if (C % epc || C / epc > 64 * 5) SR_FAIL(SR_ERR_INVALID, "sr_row_stats: C=%d unsupported", C);
  hipStream_t st = sr_stream(stream);
  dim3 grid(sr_cdiv(rows, 4));
  if (dtype == SR_F16) hipLaunchKernelGGL(row_stats_kernel<_Float16>, grid, dim3(256), 0, st, (const _Float16*)x, stats, rows, C, eps);
  else if (dtype == SR_F32) hipLaunchKernelGGL(row_stats_kernel<float>, grid, dim3(256), 0, st, (const float*)x, stats, rows, C, eps);
  else SR_FAIL(SR_ERR_INVALID, "sr_row_stats: dtype");
  SR_CHECK_LAUNCH("sr_row_stats");
  return SR_OK;
}

extern "C" int64_t sr_groupnorm_scratch_floats(int32_t B, int32_t HW) {
  // small batches cut an entry into more chunks (gn_ppc), so the need is not monotonic in B: return the largest need of any batch
  // up to B -- a host that sized `partials` once for its largest batch stays safe for every smaller one
  int64_t need = 0;
  for (int b = 1; b <= B; ++b) {
    const int64_t n = (int64_t)b * sr_cdiv(HW, gn_ppc(HW, b)) * 64 * 2;
    need = n > need ? n : need;
  }
  return need;
}

extern "C" int sr_groupnorm(const sr_groupnorm_args* a, void* stream) {
  if (!a || !a->x || !a->y || !a->gamma || !a->beta || !a->partials) SR_FAIL(SR_ERR_INVALID, "sr_groupnorm: null pointer");
  const int epc = a->dtype == SR_F16 ? 8 : 4;
  const int C = a->C1 + a->C2;
  if (a->groups <= 0 || a->groups > 64 || C % a->groups) SR_FAIL(SR_ERR_INVALID, "sr_groupnorm: C=%d groups=%d", C, a->groups);
  if (a->C1 % epc || a->C2 % epc) SR_FAIL(SR_ERR_INVALID, "sr_groupnorm: channels must be multiples of %d", epc);
  if (a->C2 > 0 && !a->x2) SR_FAIL(SR_ERR_INVALID, "sr_groupnorm: C2>0 without x2");
  const int nchunk = sr_cdiv(a->HW, gn_ppc(a->HW, a->B));
  if (a->groups > 32) SR_FAIL(SR_ERR_INVALID, "sr_groupnorm: at most 32 groups");
  const size_t lds = (size_t)(2 * C + 2 * a->groups + 16 * a->groups) * sizeof(float);
  if (lds > 64 * 1024) SR_FAIL(SR_ERR_INVALID, "sr_groupnorm: C too large");
  hipStream_t st = sr_stream(stream);
  static const bool no_fused = getenv("SR_GN_TWO_PASS") != nullptr;      // tuning / A-B aid
  if (!no_fused && (a->dtype == SR_F16 ? try_gn_wave<_Float16>(a, st) : a->dtype == SR_F32 ? try_gn_wave<float>(a, st) : false)) {
    SR_CHECK_LAUNCH("sr_groupnorm");
    return SR_OK;
  }
  if (!no_fused && (a->dtype == SR_F16 ? try_gn_fused<_Float16>(a, st) : a->dtype == SR_F32 ? try_gn_fused<float>(a, st) : false)) {
    SR_CHECK_LAUNCH("sr_groupnorm");
    return SR_OK;
  }
  dim3 grid(nchunk, a->B);
  const int cpt = C / epc, nps = 256 / cpt >= 1 ? 256 / cpt : 1;
  const size_t lds_stats = (size_t)2 * nps * C * sizeof(float);
  if (a->dtype == SR_F16) {
    hipLaunchKernelGGL(gn_stats_kernel<_Float16>, grid, dim3(256), lds_stats, st, (const _Float16*)a->x, (const _Float16*)a->x2, a->partials, a->HW, a->C1, a->C2, a->groups, nchunk);
    hipLaunchKernelGGL(gn_apply_kernel<_Float16>, grid, dim3(256), lds, st, (const _Float16*)a->x, (const _Float16*)a->x2, a->partials, a->gamma, a->beta,
                       (_Float16*)a->y, a->HW, a->C1, a->C2, a->groups, nchunk, a->eps, a->silu);
  } else if (a->dtype == SR_F32) {
    hipLaunchKernelGGL(gn_stats_kernel<float>, grid, dim3(256), lds_stats, st, (const float*)a->x, (const float*)a->x2, a->partials, a->HW, a->C1, a->C2, a->groups, nchunk);
    hipLaunchKernelGGL(gn_apply_kernel<float>, grid, dim3(256), lds, st, (const float*)a->x, (const float*)a->x2, a->partials, a->gamma, a->beta,
                       (float*)a->y, a->HW, a->C1, a->C2, a->groups, nchunk, a->eps, a->silu);
  } else SR_FAIL(SR_ERR_INVALID, "sr_groupnorm: dtype");
  SR_CHECK_LAUNCH("sr_groupnorm");
  return SR_OK;
}

static int layernorm_impl(const void* x, const float* gamma, const float* beta, void* y, int32_t rows, int32_t C,
                          float eps, int32_t dtype, void* stream, const ln_gather g) {
  if (!x || !y || !gamma || !beta) SR_FAIL(SR_ERR_INVALID, "sr_layernorm: null pointer");
  const int epc = dtype == SR_F16 ? 8 : 4;
  if (C % epc || C / epc > 64 * 5) SR_FAIL(SR_ERR_INVALID, "sr_layernorm: C=%d unsupported", C);
  hipStream_t st = sr_stream(stream);
  const int cpt = C / epc;
  auto go = [&](auto t, auto maxc, auto nrows) {
    using T = decltype(t);
    constexpr int MAXC = decltype(maxc)::value, ROWS = decltype(nrows)::value;
    hipLaunchKernelGGL((layernorm_kernel<T, MAXC, ROWS>), dim3(sr_cdiv(rows, 4 * ROWS)), dim3(256), 0, st, (const T*)x, gamma, beta, (T*)y, rows, C, eps, g);
  };
  using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
  // sub-wave rows when the chunk count factors as (power of two <= 32) x (1..5): every UNet / VAE width does
  static const bool no_sub = getenv("SR_LN_WAVE_ROWS") != nullptr;       // tuning / A-B aid
  auto sub = [&](auto t) -> bool {
    using T = decltype(t);
    if (no_sub) return false;
    for (int nch = 5; nch >= 1; --nch) {
      if (cpt % nch) continue;
      const int lpr = cpt / nch;
      if (lpr > 32 || (lpr & (lpr - 1))) continue;
      const dim3 grid(sr_cdiv(rows, 4 * (64 / lpr)));
#define SR_LN_SUB(L, N) hipLaunchKernelGGL((layernorm_sub_kernel<T, L, N>), grid, dim3(256), 0, st, (const T*)x, gamma, beta, (T*)y, rows, C, eps, g); return true
#define SR_LN_SUBN(L) switch (nch) { case 1: SR_LN_SUB(L, 1); case 2: SR_LN_SUB(L, 2); case 3: SR_LN_SUB(L, 3); case 4: SR_LN_SUB(L, 4); default: SR_LN_SUB(L, 5); }
      switch (lpr) {
        case 1: SR_LN_SUBN(1)
        case 2: SR_LN_SUBN(2)
        case 4: SR_LN_SUBN(4)
        case 8: SR_LN_SUBN(8)
        case 16: SR_LN_SUBN(16)
        default: SR_LN_SUBN(32)
      }
#undef SR_LN_SUBN
#undef SR_LN_SUB
    }
    return false;
  };
  if ((dtype == SR_F16 && sub(_Float16())) || (dtype == SR_F32 && sub(float()))) {
    SR_CHECK_LAUNCH("sr_layernorm");
    return SR_OK;
  }
  if (dtype == SR_F16) {
    if (cpt <= 64) go(_Float16(), I1{}, I4{}); else if (cpt <= 128) go(_Float16(), I2{}, I2{});
    else if (cpt <= 192) go(_Float16(), I3{}, I2{}); else go(_Float16(), I5{}, I1{});
  } else if (dtype == SR_F32) {
    if (cpt <= 64) go(float(), I1{}, I4{}); else if (cpt <= 128) go(float(), I2{}, I2{});
    else if (cpt <= 192) go(float(), I3{}, I2{}); else go(float(), I5{}, I1{});
  }
  else SR_FAIL(SR_ERR_INVALID, "sr_layernorm: dtype");
  SR_CHECK_LAUNCH("sr_layernorm");
  return SR_OK;
}

extern "C" int sr_layernorm(const void* x, const float* gamma, const float* beta, void* y, int32_t rows, int32_t C,
                            float eps, int32_t dtype, void* stream) {
  return layernorm_impl(x, gamma, beta, y, rows, C, eps, dtype, stream, ln_gather{nullptr, nullptr, 1, 0});
}

extern "C" int sr_layernorm_gather(const void* x, const int32_t* sel, int32_t nsel, int32_t frame_rows, int32_t n_frames, int32_t* err_flag,
                                   const float* gamma, const float* beta, void* y, int32_t C, float eps, int32_t dtype, void* stream) {
  if (!sel || nsel < 1 || frame_rows < 1 || n_frames < 1) SR_FAIL(SR_ERR_INVALID, "sr_layernorm_gather: bad selection");
  return layernorm_impl(x, gamma, beta, y, nsel * frame_rows, C, eps, dtype, stream, ln_gather{sel, err_flag, frame_rows, n_frames});
}
