// Stable-rendering kernels: id-map masks, latent-overlap (OverlapCorresponder.step_finished), AdaIN, engine-noise
// pooling and CorrespondMap update.  All are HBM/L2-bound gather/scatter work on int32 id maps and small fp32
// latents; duplicate scatter targets are resolved deterministically as "last row wins" (= the reference's
// sequential CPU index_put_) with an integer atomicMax on the row index instead of racing stores.
#include "sr_common.h"

namespace {

constexpr int NON_AI = 2048;

__device__ __forceinline__ bool id_valid(const int4 v) {
  return v.z != NON_AI && (v.x != 0 || v.y != 0 || v.z != 0 || v.w != 0);
}

__global__ void idmap_masks_kernel(const int4* __restrict__ ids, float* __restrict__ masks, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int4 v = ids[i];
  masks[i] = (v.z == NON_AI || (v.x == 0 && v.y == 0 && v.z == 0 && v.w == 0)) ? 1.0f : 0.0f;
}

// pixel -> latent cell with the reference's fp32 arithmetic: x_ratio = x / H, y_ratio = y / W (sic,
// corrmap.py:243,252), cell = int(ratio * latent_size) (corresponder.py:312-313)
__device__ __forceinline__ int cell_of(int f, int y, int x, int H, int W, int lh, int lw, bool* ok) {
  const float xr = (float)x / (float)H, yr = (float)y / (float)W;
  const int sx = (int)(xr * (float)lw), sy = (int)(yr * (float)lh);
  *ok = sx < lw && sy < lh;
  return (f * lh + sy) * lw + sx;
}

// info[0] = max vid, info[1] = out-of-range flag, info[2] = #valid pixels
__global__ void overlap_build1(const int4* __restrict__ ids, int N, int H, int W, int lh, int lw, int* __restrict__ pix_cell,
                               int* __restrict__ cell_win, int* __restrict__ info) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)N * H * W;
  // the two scalars every valid pixel contributes to (largest vertex id, number of valid pixels) are reduced across the wave first:
  // with one atomic per PIXEL on these two addresses the kernel ran 632 us per 8 x 512^2 call (2 x 1.5 M serialised L2 atomics;
  // profiles/r03_bench_kernel_stats.csv) for 33.5 MB of ids -- 100x its HBM time
  int vmax = -1, valid = 0;
  if (i < total) {
    const int4 v = ids[i];
    int cell = -1;
    if (id_valid(v)) {
      const int x = (int)(i % W), y = (int)((i / W) % H), f = (int)(i / ((int64_t)W * H));
      bool ok;
      cell = cell_of(f, y, x, H, W, lh, lw, &ok);
      if (!ok || v.w < 0) { atomicOr(&info[1], 1); cell = -1; }
      else {
        atomicMax(&cell_win[cell], (int)i);                // last (f,y,x) pixel of the cell wins the scatter
        vmax = v.w;
        valid = 1;
      }
    }
    pix_cell[i] = cell;
  }
  const int nvalid = __popcll(__ballot(valid));
  for (int o = 32; o >= 1; o >>= 1) vmax = max(vmax, __shfl_xor(vmax, o));
  if ((threadIdx.x & 63) == 0 && nvalid > 0) {
    atomicMax(&info[0], vmax);
    atomicAdd(&info[2], nvalid);
  }
}
__global__ void overlap_build2(const int4* __restrict__ ids, int ncell, const int* __restrict__ cell_win, int* __restrict__ cell_vid) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncell) return;
  const int w = cell_win[c];
  cell_vid[c] = w >= 0 ? ids[w].w : -1;
}

// ---- vertexID -> entries CSR (built once per call) ---------------------------------------------------------------------------
// cnt[vid] = number of valid pixels carrying vid (integer atomics: the result does not depend on their order)
__global__ void overlap_count(const int4* __restrict__ ids, const int* __restrict__ pix_cell, int64_t npix, int cap, int* __restrict__ cnt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix || pix_cell[i] < 0) return;
  const int vid = ids[i].w;
  if (vid < cap) atomicAdd(&cnt[vid], 1);
}
// exclusive prefix sum of n ints in three launches: 1024-element block-local scans + block totals, one block over the totals
// (sequential chunks with a carry), add back.
constexpr int SCAN_B = 1024;
__global__ __launch_bounds__(256) void scan_local(const int* __restrict__ in, int* __restrict__ out, int* __restrict__ bsum, int n) {
  __shared__ int wsum[4];
  const int base = blockIdx.x * SCAN_B + threadIdx.x * 4;
  int v[4], s = 0;
  for (int k = 0; k < 4; ++k) { v[k] = base + k < n ? in[base + k] : 0; s += v[k]; }
  int inc = s;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  int pre = inc - s;
  for (int w = 0; w < wv; ++w) pre += wsum[w];
  for (int k = 0; k < 4; ++k) { if (base + k < n) out[base + k] = pre; pre += v[k]; }
  if (threadIdx.x == 255) bsum[blockIdx.x] = pre;
}
__global__ __launch_bounds__(256) void scan_bsum(int* __restrict__ bsum, int nb) {
  __shared__ int wsum[4];
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int c0 = 0; c0 < nb; c0 += 256) {
    const int i = c0 + threadIdx.x;
    const int s = i < nb ? bsum[i] : 0;
    int inc = s;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int pre = carry_s + inc - s;
    for (int w = 0; w < wv; ++w) pre += wsum[w];
    if (i < nb) bsum[i] = pre;
    __syncthreads();
    if (threadIdx.x == 255) carry_s = pre + s;
    __syncthreads();
  }
}
__global__ void scan_add(int* __restrict__ out, const int* __restrict__ bsum, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] += bsum[i / SCAN_B];
}
// entries[off[vid] .. off[vid+1]) = latent cells of the pixels carrying vid.  The order inside a segment depends on the
// cursor atomics; the step sums a segment in exact integer arithmetic, so its result does not.
__global__ void overlap_fill(const int4* __restrict__ ids, const int* __restrict__ pix_cell, int64_t npix, int cap, const int* __restrict__ off,
                             int* __restrict__ cursor, int* __restrict__ entries) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  const int cell = pix_cell[i];
  if (cell < 0) return;
  const int vid = ids[i].w;
  if (vid >= cap) return;
  entries[off[vid] + atomicAdd(&cursor[vid], 1)] = cell;
}

// ---- per step ---------------------------------------------------------------------------------------------------------------
// blended[n,c,cell] = (1-r)*x + r*mean over every pixel that carries the cell's winning vertexID (corresponder.py:339-369).
// Sixteen lanes per latent cell walk the vertex's segment; the sum is taken in 2^-28 fixed point (int64: exact, so neither the
// segment order nor the lane split can change a bit of the result; finite |x| saturates at 4096 -- stated in sr_hip.h -- so 2^21
// pixels of one vertex still fit).  A NaN or an infinity must not be laundered into a finite mean (the reference's float mean,
// corresponder.py:339-369, propagates it): every non-finite input raises bit c of `bad`, which travels through the same
// shuffle reduction and turns the vertex's mean of channel c into NaN.
constexpr float FIX_SCALE = 268435456.0f;                   // 2^28
__device__ __forceinline__ long long to_fix(float v, int c, int& bad) {
  if (!(fabsf(v) <= 3.0e38f)) { bad |= 1 << c; v = 0.f; }   // NaN fails every comparison
  v = fminf(fmaxf(v, -4096.0f), 4096.0f);
  return __float2ll_rn(v * FIX_SCALE);                      // power-of-two scaling is exact; round to nearest below 2^-28
}
constexpr int BLEND_LANES = 16;                            // lanes that share one latent cell's segment walk
template <int C>
__global__ __launch_bounds__(256) void overlap_blend(const float* __restrict__ x, const int* __restrict__ cell_vid, const int* __restrict__ off,
                                                     const int* __restrict__ entries, int ncell, int lhw, int cap, float ratio,
                                                     float* __restrict__ blended) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int cell = t / BLEND_LANES, l = t % BLEND_LANES;
  if (cell >= ncell) return;                                // (whole lane groups leave together)
  const int vid = cell_vid[cell];
  const int f = cell / lhw, p = cell - f * lhw;
  long long acc[C];
  for (int c = 0; c < C; ++c) acc[c] = 0;
  int bad = 0;
  int b = 0, e = 0;
  if (vid >= 0 && vid < cap) { b = off[vid]; e = off[vid + 1]; }
  // two independent entry -> latent chains in flight per lane (the walk is a chain of dependent L2 round trips)
  int i = b + l;
  for (; i + BLEND_LANES < e; i += 2 * BLEND_LANES) {
    const int c0 = entries[i], c1 = entries[i + BLEND_LANES];
    const int f0 = c0 / lhw, f1 = c1 / lhw;
    const float* x0 = x + (int64_t)f0 * C * lhw + (c0 - f0 * lhw);
    const float* x1 = x + (int64_t)f1 * C * lhw + (c1 - f1 * lhw);
    float v0[C], v1[C];
    for (int c = 0; c < C; ++c) { v0[c] = x0[(int64_t)c * lhw]; v1[c] = x1[(int64_t)c * lhw]; }
    for (int c = 0; c < C; ++c) acc[c] += to_fix(v0[c], c, bad) + to_fix(v1[c], c, bad);
  }
  if (i < e) {
    const int ce = entries[i];
    const int fe = ce / lhw, pe = ce - fe * lhw;
    const float* xe = x + (int64_t)fe * C * lhw + pe;
    for (int c = 0; c < C; ++c) acc[c] += to_fix(xe[(int64_t)c * lhw], c, bad);
  }
  for (int c = 0; c < C; ++c)
    for (int o = 1; o < BLEND_LANES; o <<= 1) acc[c] += __shfl_xor(acc[c], o, BLEND_LANES);
  for (int o = 1; o < BLEND_LANES; o <<= 1) bad |= __shfl_xor(bad, o, BLEND_LANES);
  if (l < C) {
    long long mine = acc[0];
    for (int c = 1; c < C; ++c) mine = (l == c) ? acc[c] : mine;
    const int64_t at = ((int64_t)f * C + l) * lhw + p;
    const float xv = x[at];
    float out = xv;
    if (e > b) {
      float mean = (float)((double)mine / ((double)(e - b) * (double)FIX_SCALE));
      if ((bad >> l) & 1) mean = __builtin_nanf("");
      out = (1.0f - ratio) * xv + ratio * mean;
    }
    blended[at] = out;
  }
}

// block reduction helpers (fixed order -> reproducible)
__device__ __forceinline__ float block_sum(float v, float* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
  return t;
}

// one block per (n, c) plane: AdaIN(content = x, style = blended) -> x in place (corresponder.py:371-376).  1024 threads; a plane
// of up to APPLY_REG * 1024 values (64x64 .. 128x128 latents) is read ONCE into registers, both moments come from there.
constexpr int APPLY_T = 1024, APPLY_REG = 16;
__global__ __launch_bounds__(APPLY_T) void overlap_apply(float* x, const float* __restrict__ blended, int lhw, float eps) {
  __shared__ float red[APPLY_T / 64];
  float* xp = x + (int64_t)blockIdx.x * lhw;
  const float* bp = blended + (int64_t)blockIdx.x * lhw;
  const bool in_regs = lhw <= APPLY_REG * APPLY_T;
  float xv[APPLY_REG], bv[APPLY_REG];
  float sc = 0.f, ss = 0.f;
  if (in_regs) {
#pragma unroll
    for (int k = 0; k < APPLY_REG; ++k) {
      const int p = threadIdx.x + k * APPLY_T;
      xv[k] = p < lhw ? xp[p] : 0.f; bv[k] = p < lhw ? bp[p] : 0.f;
      sc += xv[k]; ss += bv[k];
    }
  } else {
    for (int p = threadIdx.x; p < lhw; p += APPLY_T) { sc += xp[p]; ss += bp[p]; }
  }
  const float mc = block_sum(sc, red) / (float)lhw;
  const float ms = block_sum(ss, red) / (float)lhw;
  float qc = 0.f, qs = 0.f;
  if (in_regs) {
#pragma unroll
    for (int k = 0; k < APPLY_REG; ++k) {
      const int p = threadIdx.x + k * APPLY_T;
      const float a = xv[k] - mc, b = bv[k] - ms;
      if (p < lhw) { qc += a * a; qs += b * b; }
    }
  } else {
    for (int p = threadIdx.x; p < lhw; p += APPLY_T) {
      const float a = xp[p] - mc, b = bp[p] - ms;
      qc += a * a; qs += b * b;
    }
  }
  const float stdc = sqrtf(block_sum(qc, red) / (float)(lhw - 1) + eps);
  const float stds = sqrtf(block_sum(qs, red) / (float)(lhw - 1) + eps);
  if (in_regs) {
#pragma unroll
    for (int k = 0; k < APPLY_REG; ++k) {
      const int p = threadIdx.x + k * APPLY_T;
      if (p < lhw) xp[p] = (xv[k] - mc) / stdc * stds + ms;
    }
  } else {
    __syncthreads();
    for (int p = threadIdx.x; p < lhw; p += APPLY_T) xp[p] = (xp[p] - mc) / stdc * stds + ms;
  }
}

// generic AdaIN: one block per (n,c)
template <typename TS>
__global__ __launch_bounds__(256) void adain_kernel(const float* __restrict__ content, int64_t c_ps, int64_t c_cs, int64_t c_ns, int HWc,
                                                    const TS* __restrict__ style, int64_t s_ps, int64_t s_cs, int64_t s_ns, int HWs,
                                                    float* __restrict__ out, int C, float eps, int half_stats,
                                                    const float* __restrict__ style_part = nullptr, int nblk = 0) {
  __shared__ float red[4];
  const int n = blockIdx.x / C, c = blockIdx.x - n * C;
  const float* cp = content + n * c_ns + c * c_cs;
  const TS* sp = style + n * s_ns + c * s_cs;
  float a = 0.f, b = 0.f;
  for (int p = threadIdx.x; p < HWc; p += 256) a += cp[p * c_ps];
  if (style_part) { for (int i = threadIdx.x; i < nblk; i += 256) b += style_part[i * C + c]; }      // per-block sums (pass A)
  else            { for (int p = threadIdx.x; p < HWs; p += 256) b += sr_load_f(sp + p * s_ps); }
  const float mc = block_sum(a, red) / (float)HWc;
  float ms = block_sum(b, red) / (float)HWs;
  a = 0.f; b = 0.f;
  for (int p = threadIdx.x; p < HWc; p += 256) { const float t = cp[p * c_ps] - mc; a += t * t; }
  if (style_part) { for (int i = threadIdx.x; i < nblk; i += 256) b += style_part[(nblk + i) * C + c]; }   // centred squares (pass B)
  else            { for (int p = threadIdx.x; p < HWs; p += 256) { const float t = sr_load_f(sp + p * s_ps) - ms; b += t * t; } }
  const float varc = block_sum(a, red) / (float)(HWc - 1) + eps;
  float vars = block_sum(b, red) / (float)(HWs - 1);
  float stds;
  if (half_stats) {
    // style is an fp16 tensor in the reference: var / +eps / mean are rounded to fp16, sqrt in fp32, std -> fp16
    const _Float16 v16 = (_Float16)((float)(_Float16)vars + (float)(_Float16)eps);
    stds = (float)(_Float16)sqrtf((float)v16);
    ms = (float)(_Float16)ms;
  } else {
    stds = sqrtf(vars + eps);
  }
  const float stdc = sqrtf(varc);
  float* op = out + (int64_t)blockIdx.x * HWc;
  for (int p = threadIdx.x; p < HWc; p += 256) op[p] = (cp[p * c_ps] - mc) / stdc * stds + ms;
}

// Style statistics of an RGBA16F plane (HW pixels x 4 channels) spread over the chip instead of one workgroup per channel:
// pass A writes per-block channel sums part[blk][4]; pass B re-reduces them (fixed order) to the means and writes the centred
// sums of squares part[nblk + blk][4].  adain_kernel then only sums 2 x nblk partials per channel.
__global__ __launch_bounds__(256) void style_partial_rgba16(const _Float16* __restrict__ style, int HW, float* __restrict__ part, int nblk,
                                                            int pass) {
  __shared__ float red[4];
  __shared__ float mean[4];
  if (pass == 1) {
    float m[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < nblk; i += 256)
      for (int c = 0; c < 4; ++c) m[c] += part[i * 4 + c];
    for (int c = 0; c < 4; ++c) { const float t = block_sum(m[c], red); if (threadIdx.x == 0) mean[c] = t / (float)HW; }
    __syncthreads();
  }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += nblk * 256) {
    const h16x4 v = *(const h16x4*)(style + (int64_t)p * 4);
    for (int c = 0; c < 4; ++c) {
      const float t = pass == 0 ? (float)v[c] : (float)v[c] - mean[c];
      acc[c] += pass == 0 ? t : t * t;
    }
  }
  for (int c = 0; c < 4; ++c) {
    const float t = block_sum(acc[c], red);
    if (threadIdx.x == 0) part[((int64_t)pass * nblk + blockIdx.x) * 4 + c] = t;
  }
}

// strip means (`strip` consecutive pixels of the flattened image: 64 in the engine, renderManager.py:929-932; m*m in the loader,
// _nodes/loaders.py:131-146) of noise*(1-mask) + bg*mask
__global__ void noise_pool_kernel(const _Float16* __restrict__ noise, const _Float16* __restrict__ alpha, const float* __restrict__ bg,
                                  float* __restrict__ pooled, int ngroups, int strip) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;      // over ngroups*4
  if (i >= ngroups * 4) return;
  const int g = i >> 2, c = i & 3;
  float s = 0.f;
  for (int k = 0; k < strip; ++k) {
    const int64_t px = (int64_t)g * strip + k;
    const _Float16 m = (_Float16)(1.0f - (float)alpha[px]);            // mask = 1 - alpha (fp16)
    const _Float16 om = (_Float16)(1.0f - (float)m);
    const _Float16 a = (_Float16)((float)noise[px * 4 + c] * (float)om); // fp16 product
    s += (float)a + bg[px * 4 + c] * (float)m;
  }
  pooled[i] = s / (float)strip;
}

__global__ void corrmap_pass1(const int4* __restrict__ ids, const float* __restrict__ mask, int n, int sprite, int material, int chk_s,
                              int chk_m, int mode_first, const uint8_t* __restrict__ writtens, int kk, int V, int* __restrict__ winner,
                              int* __restrict__ err) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (mask && !(mask[i] > 0.f)) return;
  const int4 v = ids[i];
  if (chk_s && v.x != sprite) return;
  if (chk_m && v.y != material) return;
  if (v.z < 0 || v.z >= kk || v.w < 0 || v.w >= V) { atomicOr(err, 1); return; }
  const int cell = v.z * V + v.w;
  if (mode_first && writtens[cell]) return;
  atomicMax(&winner[cell], i);
}
__global__ void corrmap_pass2(const float* __restrict__ frame, int Cf, const int4* __restrict__ ids, const int* __restrict__ src_index, int n,
                              int V, int kk, const int* __restrict__ winner, const int* __restrict__ err,
                              _Float16* __restrict__ values, uint8_t* __restrict__ writtens) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (*err) return;                                        // the reference raises before writing anything of this frame
  const int4 v = ids[i];
  if (v.z < 0 || v.z >= kk || v.w < 0 || v.w >= V) return;
  const int cell = v.z * V + v.w;
  if (winner[cell] != i) return;                           // (cells of filtered rows keep winner == -1 / other)
  const int64_t src = src_index ? src_index[i] : i;
  const float* cp = frame + src * Cf;
  _Float16* o = values + (int64_t)cell * 4;
  o[0] = (_Float16)cp[0]; o[1] = (_Float16)cp[1]; o[2] = (_Float16)cp[2];
  o[3] = Cf >= 4 ? (_Float16)cp[3] : (_Float16)1.0f;
  writtens[cell] = 1;
}

// legacy Overlap / ResizeOverlap fused at latent resolution: one thread per output cell
__global__ void legacy_overlap_kernel(const float* __restrict__ x, float* __restrict__ y, const int* __restrict__ pix_vert,
                                      const int* __restrict__ offsets, const int* __restrict__ tr_f, const int* __restrict__ tr_y,
                                      const int* __restrict__ tr_x, const float* __restrict__ vn, int T, int C, int h, int w, int H, int W,
                                      float alpha, int radius, int algo, int keep_nonzero) {
  const int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= (int64_t)T * h * w) return;
  const int j = (int)(cell % w), i = (int)((cell / w) % h), f = (int)(cell / ((int64_t)w * h));
  // F.interpolate(mode='nearest'): src = floor(dst * in/out) (float scale as torch: in/out computed in fp32)
  const float sy = (float)H / (float)h, sx = (float)W / (float)w;
  const int py = min((int)floorf((float)i * sy), H - 1), px = min((int)floorf((float)j * sx), W - 1);
  const int64_t lhw = (int64_t)h * w;
  const float* xf = x + (int64_t)f * C * lhw;
  float* yo = y + (int64_t)f * C * lhw + (int64_t)i * w + j;
  const int v = pix_vert[((int64_t)f * H + py) * W + px];
  const int b = v >= 0 ? offsets[v] : 0, e = v >= 0 ? offsets[v + 1] : 0;
  if (v < 0 || e - b < 2) {                                   // no id / vertex seen once: unchanged
    for (int c = 0; c < C; ++c) yo[c * lhw] = xf[c * lhw + (int64_t)i * w + j];
    return;
  }
  // up-sampled latent value of full-res pixel (yy,xx) of frame ff: nearest up-sampling = cell (floor(yy*h/H), floor(xx*w/W))
  const float uy = (float)h / (float)H, ux = (float)w / (float)W;
  auto cell_of_px = [&](int yy, int xx) -> int64_t {
    const int ci = min((int)floorf((float)yy * uy), h - 1), cj = min((int)floorf((float)xx * ux), w - 1);
    return (int64_t)ci * w + cj;
  };
  // which trace entry is this pixel?  (the one with the same (f,py,px))
  int me = b;
  for (int t = b; t < e; ++t) if (tr_f[t] == f && tr_y[t] == py && tr_x[t] == px) { me = t; break; }
  const float my_vn = (algo == 3) ? vn[((int64_t)f * H + py) * W + px] : 0.f;
  float norm = 0.f;
  float acc[8];
  for (int c = 0; c < C && c < 8; ++c) acc[c] = 0.f;
  const float inv_r = 1.0f / (float)(2 * radius + 1);
  for (int t = b; t < e; ++t) {
    const int tf = tr_f[t], ty = tr_y[t], tx = tr_x[t];
    float wgt;
    if (algo == 0) wgt = 1.0f;
    else if (algo == 1) wgt = 1.0f / (fabsf((float)tf - (float)f) + 1.0f);
    else if (algo == 2) wgt = 1.0f / (fabsf((float)tx - (float)px) + fabsf((float)ty - (float)py) + 1.0f);
    else wgt = 1.0f / (fabsf(1.0f - vn[((int64_t)tf * H + ty) * W + tx]) + 1.0f);
    norm += (algo == 3) ? 0.f : wgt;
    const float* xt = x + (int64_t)tf * C * lhw;
    for (int c = 0; c < C && c < 8; ++c) {
      float pooled = 0.f;
      for (int k = -radius; k <= radius; ++k)
        pooled += xt[c * lhw + cell_of_px(min(max(ty + k, 0), H - 1), min(max(tx + k, 0), W - 1))];
      acc[c] += wgt * (pooled * inv_r);
    }
  }
  if (algo == 3) norm = (float)(e - b) * (1.0f / (fabsf(1.0f - my_vn) + 1.0f));   // column sum of w[.,me] applied to row me
  (void)me;
  for (int c = 0; c < C && c < 8; ++c) {
    const float orig = xf[c * lhw + (int64_t)i * w + j];
    const float val = alpha * (acc[c] / norm) + (1.0f - alpha) * orig;
    yo[c * lhw] = (keep_nonzero && val == 0.0f) ? orig : val;
  }
}

inline dim3 g1(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

// ---- legacy Overlap in the REFERENCE'S IN-PLACE ORDER (kernel radius > 0) --------------------------------------------------------
// The reference walks its vertex dict and writes into a tensor that aliases the one it reads (overlap.py:103 `.detach()`), so a
// vertex sees the updates of every vertex before it wherever its (2r+1)-pixel diagonal windows touch their pixels.  The host sorts
// the vertices into LEVELS (sr_legacy_levels): two vertices of one level neither read what the other writes nor write what the
// other reads, and every conflict with an earlier vertex puts the later one in a later level -- replaying the levels in order,
// compute-then-write inside each, gives bit for bit what the sequential loop gives.
__global__ void legacy_seq_compute(const float* __restrict__ U, const int* __restrict__ lvl_vert, int nlv, const int* __restrict__ offsets,
                                   const int* __restrict__ tr_f, const int* __restrict__ tr_y, const int* __restrict__ tr_x,
                                   const float* __restrict__ vn, int C, int H, int W, float alpha, int radius, int algo,
                                   float* __restrict__ newval, int max_len) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int vi = (int)(gid / max_len), t = (int)(gid % max_len);
  if (vi >= nlv) return;
  const int v = lvl_vert[vi];
  const int b = offsets[v], e = offsets[v + 1];
  if (b + t >= e) return;
  const int64_t hw = (int64_t)H * W;
  const int f = tr_f[b + t], y = tr_y[b + t], x = tr_x[b + t];
  const float inv_r = (float)(2 * radius + 1);
  float acc[8], norm = 0.f;
  for (int c = 0; c < C; ++c) acc[c] = 0.f;
  const float my_vn = (algo == 3) ? vn[((int64_t)f * H + y) * W + x] : 0.f;
  for (int s = b; s < e; ++s) {
    const int sf = tr_f[s], sy = tr_y[s], sx = tr_x[s];
    float wgt;
    if (algo == 0) wgt = 1.0f;
    else if (algo == 1) wgt = 1.0f / (fabsf((float)f - (float)sf) + 1.0f);
    else if (algo == 2) wgt = 1.0f / (fabsf((float)x - (float)sx) + fabsf((float)y - (float)sy) + 1.0f);
    else wgt = 1.0f / (fabsf(1.0f - vn[((int64_t)sf * H + sy) * W + sx]) + 1.0f);
    norm += (algo == 3) ? 0.f : wgt;
    const float* us = U + (int64_t)sf * C * hw;
    for (int c = 0; c < C; ++c) {
      float pooled = 0.f;
      for (int k = -radius; k <= radius; ++k)
        pooled += us[c * hw + (int64_t)min(max(sy + k, 0), H - 1) * W + min(max(sx + k, 0), W - 1)];
      acc[c] += wgt * (pooled / inv_r);
    }
  }
  if (algo == 3) norm = (float)(e - b) * (1.0f / (fabsf(1.0f - my_vn) + 1.0f));
  const float* uf = U + (int64_t)f * C * hw + (int64_t)y * W + x;
  for (int c = 0; c < C; ++c) newval[(int64_t)(b + t) * C + c] = alpha * (acc[c] / norm) + (1.0f - alpha) * uf[c * hw];
}
__global__ void legacy_seq_write(float* __restrict__ U, const int* __restrict__ lvl_vert, int nlv, const int* __restrict__ offsets,
                                 const int* __restrict__ tr_f, const int* __restrict__ tr_y, const int* __restrict__ tr_x, int C, int H,
                                 int W, const float* __restrict__ newval, int max_len) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int vi = (int)(gid / max_len), t = (int)(gid % max_len);
  if (vi >= nlv) return;
  const int v = lvl_vert[vi];
  const int b = offsets[v], e = offsets[v + 1];
  if (b + t >= e) return;
  const int64_t hw = (int64_t)H * W;
  float* uf = U + (int64_t)tr_f[b + t] * C * hw + (int64_t)tr_y[b + t] * W + tr_x[b + t];
  for (int c = 0; c < C; ++c) uf[c * hw] = newval[(int64_t)(b + t) * C + c];
}
// F.interpolate(mode="nearest"): dst (T,C,Ho,Wo) <- src (T,C,Hi,Wi), src index = min(floor(dst * in/out), in-1) (fp32 scale)
__global__ void nearest_resize_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t planes, int Hi, int Wi, int Ho, int Wo,
                                      const float* __restrict__ keep_if_zero) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= planes * Ho * Wo) return;
  const int xo = (int)(i % Wo), yo = (int)((i / Wo) % Ho);
  const int64_t pl = i / ((int64_t)Wo * Ho);
  const int yi = min((int)floorf((float)yo * ((float)Hi / (float)Ho)), Hi - 1), xi = min((int)floorf((float)xo * ((float)Wi / (float)Wo)), Wi - 1);
  const float v = src[(pl * Hi + yi) * Wi + xi];
  dst[i] = (keep_if_zero && v == 0.0f) ? keep_if_zero[i] : v;      // ResizeOverlap: where(ovlp != 0, ovlp, orig) (overlap.py:218-221)
}

}  // namespace

extern "C" int sr_legacy_overlap(const float* x, float* y, const int32_t* pix_vert, const int32_t* offsets, const int32_t* tr_f,
                                 const int32_t* tr_y, const int32_t* tr_x, const float* view_normal, int32_t T, int32_t C, int32_t h,
                                 int32_t w, int32_t H, int32_t W, float alpha, int32_t radius, int32_t algo, int32_t keep_nonzero,
                                 void* stream) {
  if (!x || !y || !pix_vert || !offsets || !tr_f || !tr_y || !tr_x) SR_FAIL(SR_ERR_INVALID, "sr_legacy_overlap: null");
  if (C > 8 || radius < 0 || algo < 0 || algo > 3 || (algo == 3 && !view_normal)) SR_FAIL(SR_ERR_INVALID, "sr_legacy_overlap: bad args");
  hipLaunchKernelGGL(legacy_overlap_kernel, g1((int64_t)T * h * w), dim3(256), 0, sr_stream(stream), x, y, pix_vert, offsets, tr_f, tr_y, tr_x,
                     view_normal, T, C, h, w, H, W, alpha, radius, algo, keep_nonzero);
  SR_CHECK_LAUNCH("sr_legacy_overlap");
  return SR_OK;
}

extern "C" int sr_legacy_overlap_seq(float* U, float* newval, const int32_t* lvl_vert, const int32_t* lvl_off_host, int32_t n_levels,
                                     int32_t max_len, const int32_t* offsets, const int32_t* tr_f, const int32_t* tr_y, const int32_t* tr_x,
                                     const float* view_normal, int32_t C, int32_t H, int32_t W, float alpha, int32_t radius, int32_t algo,
                                     void* stream) {
  if (!U || !newval || !lvl_vert || !lvl_off_host || !offsets || !tr_f || !tr_y || !tr_x) SR_FAIL(SR_ERR_INVALID, "sr_legacy_overlap_seq: null");
  if (C > 8 || radius < 0 || algo < 0 || algo > 3 || (algo == 3 && !view_normal) || max_len < 1) SR_FAIL(SR_ERR_INVALID, "sr_legacy_overlap_seq: bad args");
  hipStream_t st = sr_stream(stream);
  for (int l = 0; l < n_levels; ++l) {
    const int b = lvl_off_host[l], n = lvl_off_host[l + 1] - b;
    if (n <= 0) continue;
    const dim3 grid = g1((int64_t)n * max_len);
    hipLaunchKernelGGL(legacy_seq_compute, grid, dim3(256), 0, st, U, lvl_vert + b, n, offsets, tr_f, tr_y, tr_x, view_normal, C, H, W, alpha,
                       radius, algo, newval, max_len);
    hipLaunchKernelGGL(legacy_seq_write, grid, dim3(256), 0, st, U, lvl_vert + b, n, offsets, tr_f, tr_y, tr_x, C, H, W, newval, max_len);
  }
  SR_CHECK_LAUNCH("sr_legacy_overlap_seq");
  return SR_OK;
}

extern "C" int sr_nearest_resize(const float* src, float* dst, int64_t planes, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo,
                                 const float* keep_if_zero, void* stream) {
  if (!src || !dst || planes < 1 || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1) SR_FAIL(SR_ERR_INVALID, "sr_nearest_resize: bad args");
  hipLaunchKernelGGL(nearest_resize_kernel, g1(planes * Ho * Wo), dim3(256), 0, sr_stream(stream), src, dst, planes, Hi, Wi, Ho, Wo, keep_if_zero);
  SR_CHECK_LAUNCH("sr_nearest_resize");
  return SR_OK;
}

extern "C" int sr_idmap_masks(const int32_t* ids, float* masks, int64_t n, void* stream) {
  if (!ids || !masks) SR_FAIL(SR_ERR_INVALID, "sr_idmap_masks: null");
  hipLaunchKernelGGL(idmap_masks_kernel, g1(n), dim3(256), 0, sr_stream(stream), (const int4*)ids, masks, n);
  SR_CHECK_LAUNCH("sr_idmap_masks");
  return SR_OK;
}

extern "C" int sr_overlap_build(const int32_t* ids, int32_t N, int32_t H, int32_t W, int32_t lh, int32_t lw, int32_t* pix_cell,
                                int32_t* cell_vid, int32_t* info, void* stream) {
  if (!ids || !pix_cell || !cell_vid || !info) SR_FAIL(SR_ERR_INVALID, "sr_overlap_build: null");
  hipStream_t st = sr_stream(stream);
  const int ncell = N * lh * lw;
  // cell_vid doubles as the winner table during pass 1
  if (hipMemsetAsync(cell_vid, 0xFF, (size_t)ncell * 4, st) != hipSuccess) SR_FAIL(SR_ERR_LAUNCH, "memset");
  if (hipMemsetAsync(info, 0, 16, st) != hipSuccess) SR_FAIL(SR_ERR_LAUNCH, "memset");
  const int64_t npix = (int64_t)N * H * W;
  hipLaunchKernelGGL(overlap_build1, g1(npix), dim3(256), 0, st, (const int4*)ids, N, H, W, lh, lw, pix_cell, cell_vid, info);
  hipLaunchKernelGGL(overlap_build2, g1(ncell), dim3(256), 0, st, (const int4*)ids, ncell, cell_vid, cell_vid);
  SR_CHECK_LAUNCH("sr_overlap_build");
  return SR_OK;
}

extern "C" int64_t sr_overlap_csr_scratch_ints(int32_t vid_capacity) {
  const int64_t n = (int64_t)vid_capacity + 1;
  return n + (n + SCAN_B - 1) / SCAN_B;                      // counts / cursors + block totals of the scan
}

extern "C" int sr_overlap_csr(const int32_t* ids, const int32_t* pix_cell, int32_t N, int32_t H, int32_t W, int32_t vid_capacity,
                              int32_t* vid_off, int32_t* entries, int32_t* scratch, void* stream) {
  if (!ids || !pix_cell || !vid_off || !entries || !scratch || vid_capacity < 1) SR_FAIL(SR_ERR_INVALID, "sr_overlap_csr: bad args");
  hipStream_t st = sr_stream(stream);
  const int n = vid_capacity + 1, nb = (n + SCAN_B - 1) / SCAN_B;
  int* cnt = scratch;
  int* bsum = scratch + n;
  const int64_t npix = (int64_t)N * H * W;
  if (hipMemsetAsync(cnt, 0, (size_t)n * 4, st) != hipSuccess) SR_FAIL(SR_ERR_LAUNCH, "memset");
  hipLaunchKernelGGL(overlap_count, g1(npix), dim3(256), 0, st, (const int4*)ids, pix_cell, npix, vid_capacity, cnt);
  hipLaunchKernelGGL(scan_local, dim3(nb), dim3(256), 0, st, cnt, vid_off, bsum, n);
  hipLaunchKernelGGL(scan_bsum, dim3(1), dim3(256), 0, st, bsum, nb);
  hipLaunchKernelGGL(scan_add, g1(n), dim3(256), 0, st, vid_off, bsum, n);
  if (hipMemsetAsync(cnt, 0, (size_t)n * 4, st) != hipSuccess) SR_FAIL(SR_ERR_LAUNCH, "memset");
  hipLaunchKernelGGL(overlap_fill, g1(npix), dim3(256), 0, st, (const int4*)ids, pix_cell, npix, vid_capacity, vid_off, cnt, entries);
  SR_CHECK_LAUNCH("sr_overlap_csr");
  return SR_OK;
}

extern "C" int sr_overlap_step(float* x, const int32_t* cell_vid, const int32_t* vid_off, const int32_t* entries, int32_t N, int32_t C,
                               int32_t lh, int32_t lw, int32_t vid_capacity, float ratio, float* blended, void* stream) {
  if (!x || !cell_vid || !vid_off || !entries || !blended) SR_FAIL(SR_ERR_INVALID, "sr_overlap_step: null");
  if (lh * lw < 2) SR_FAIL(SR_ERR_INVALID, "sr_overlap_step: latent too small");
  if (C < 1 || C > 8) SR_FAIL(SR_ERR_INVALID, "sr_overlap_step: 1 <= C <= 8");
  hipStream_t st = sr_stream(stream);
  const int ncell = N * lh * lw, lhw = lh * lw;
  const dim3 grid((unsigned)(((int64_t)ncell * BLEND_LANES + 255) / 256));
#define SR_BLEND(CC) hipLaunchKernelGGL(overlap_blend<CC>, grid, dim3(256), 0, st, x, cell_vid, vid_off, entries, ncell, lhw, vid_capacity, ratio, blended)
  switch (C) {
    case 1: SR_BLEND(1); break; case 2: SR_BLEND(2); break; case 3: SR_BLEND(3); break; case 4: SR_BLEND(4); break;
    case 5: SR_BLEND(5); break; case 6: SR_BLEND(6); break; case 7: SR_BLEND(7); break; default: SR_BLEND(8); break;
  }
#undef SR_BLEND
  hipLaunchKernelGGL(overlap_apply, dim3(N * C), dim3(APPLY_T), 0, st, x, blended, lhw, 1e-5f);
  SR_CHECK_LAUNCH("sr_overlap_step");
  return SR_OK;
}

extern "C" int sr_adain(const float* content, int64_t c_ps, int64_t c_cs, int64_t c_ns, int32_t HWc, const void* style,
                        int32_t style_dtype, int64_t s_ps, int64_t s_cs, int64_t s_ns, int32_t HWs, float* out, int32_t N, int32_t C,
                        float eps, float* stats, void* stream) {
  (void)stats;
  if (!content || !style || !out || HWc < 2 || HWs < 2) SR_FAIL(SR_ERR_INVALID, "sr_adain: bad args");
  hipStream_t st = sr_stream(stream);
  if (style_dtype == SR_F16)
    hipLaunchKernelGGL(adain_kernel<_Float16>, dim3(N * C), dim3(256), 0, st, content, c_ps, c_cs, c_ns, HWc, (const _Float16*)style, s_ps,
                       s_cs, s_ns, HWs, out, C, eps, 1);
  else
    hipLaunchKernelGGL(adain_kernel<float>, dim3(N * C), dim3(256), 0, st, content, c_ps, c_cs, c_ns, HWc, (const float*)style, s_ps, s_cs,
                       s_ns, HWs, out, C, eps, 0);
  SR_CHECK_LAUNCH("sr_adain");
  return SR_OK;
}

extern "C" int sr_noise_pool(const void* noise_f16, const void* alpha_f16, const float* bg, float* pooled, float* out, int32_t H,
                             int32_t W, float* stats, void* stream) {
  if (H % 8 || W % 8 || (H * W) % 64) SR_FAIL(SR_ERR_INVALID, "sr_noise_pool: H,W must be multiples of 8");
  return sr_noise_pool_strips(noise_f16, alpha_f16, bg, pooled, out, H, W, 64, stats, stream);
}

extern "C" int sr_noise_pool_strips(const void* noise_f16, const void* alpha_f16, const float* bg, float* pooled, float* out, int32_t H,
                                    int32_t W, int32_t strip, float* stats, void* stream) {
  if (!noise_f16 || !alpha_f16 || !bg || !pooled || !out) SR_FAIL(SR_ERR_INVALID, "sr_noise_pool: null");
  if (strip < 1 || ((int64_t)H * W) % strip) SR_FAIL(SR_ERR_INVALID, "sr_noise_pool: %d x %d pixels are not whole strips of %d", H, W, strip);
  hipStream_t st = sr_stream(stream);
  const int ng = H * W / strip;
  hipLaunchKernelGGL(noise_pool_kernel, g1((int64_t)ng * 4), dim3(256), 0, st, (const _Float16*)noise_f16, (const _Float16*)alpha_f16, bg, pooled, ng, strip);
  // AdaIN(content = pooled NHWC (1,h,w,4), style = full-res fp16 noise NHWC) -> (1,4,h,w)
  if (stats) {                                              // scratch of 2*256*4 floats: style statistics computed chip-wide
    constexpr int NBLK = 256;
    hipLaunchKernelGGL(style_partial_rgba16, dim3(NBLK), dim3(256), 0, st, (const _Float16*)noise_f16, H * W, stats, NBLK, 0);
    hipLaunchKernelGGL(style_partial_rgba16, dim3(NBLK), dim3(256), 0, st, (const _Float16*)noise_f16, H * W, stats, NBLK, 1);
    hipLaunchKernelGGL(adain_kernel<_Float16>, dim3(4), dim3(256), 0, st, pooled, (int64_t)4, (int64_t)1, (int64_t)0, ng,
                       (const _Float16*)noise_f16, (int64_t)4, (int64_t)1, (int64_t)0, H * W, out, 4, 1e-5f, 1, stats, NBLK);
  } else
  hipLaunchKernelGGL(adain_kernel<_Float16>, dim3(4), dim3(256), 0, st, pooled, (int64_t)4, (int64_t)1, (int64_t)0, ng, (const _Float16*)noise_f16,
                     (int64_t)4, (int64_t)1, (int64_t)0, H * W, out, 4, 1e-5f, 1);
  SR_CHECK_LAUNCH("sr_noise_pool");
  return SR_OK;
}

extern "C" int sr_corrmap_update(const float* frame, int32_t Cf, const int32_t* ids, const float* mask, const int32_t* src_index,
                                 int32_t n, int32_t sprite, int32_t material, int32_t chk_s, int32_t chk_m, int32_t mode_first,
                                 void* values, uint8_t* writtens, int32_t kk, int32_t V, int32_t* winner, int32_t* err, void* stream) {
  if (!frame || !ids || !values || !writtens || !winner || !err) SR_FAIL(SR_ERR_INVALID, "sr_corrmap_update: null");
  if (Cf != 3 && Cf != 4) SR_FAIL(SR_ERR_INVALID, "sr_corrmap_update: Cf must be 3 or 4");
  hipStream_t st = sr_stream(stream);
  if (hipMemsetAsync(winner, 0xFF, (size_t)kk * V * 4, st) != hipSuccess) SR_FAIL(SR_ERR_LAUNCH, "memset");
  hipLaunchKernelGGL(corrmap_pass1, g1(n), dim3(256), 0, st, (const int4*)ids, mask, n, sprite, material, chk_s, chk_m, mode_first, writtens, kk, V,
                     winner, err);
  hipLaunchKernelGGL(corrmap_pass2, g1(n), dim3(256), 0, st, frame, Cf, (const int4*)ids, src_index, n, V, kk, winner, err, (_Float16*)values, writtens);
  SR_CHECK_LAUNCH("sr_corrmap_update");
  return SR_OK;
}
