// Small element-wise kernels: layout/dtype conversion at the UNet/VAE boundary, timestep embedding,
// row softmax (VAE mid attention) and the sampler arithmetic (EPS scaling, CFG, euler / ddpm / lcm updates).
// All HBM-bound and tiny next to the MFMA kernels; they exist so the per-step loop never leaves the device.
#include "sr_common.h"

namespace {

template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int B, int C, int HW, int Cpad,
                                    float scale, const float* __restrict__ pbs) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // over B*HW*Cpad
  const int64_t total = (int64_t)B * HW * Cpad;
  if (i >= total) return;
  const int c = (int)(i % Cpad);
  const int64_t bp = i / Cpad;
  const int p = (int)(bp % HW), b = (int)(bp / HW);
  float v = 0.f;
  if (c < C) v = x[((int64_t)b * C + c) * HW + p] * scale * (pbs ? pbs[b] : 1.0f);
  sr_store_f(y + i, v);
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ x, float* __restrict__ y, int B, int C, int HW, int ldc) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // over B*C*HW (output order)
  const int64_t total = (int64_t)B * C * HW;
  if (i >= total) return;
  const int p = (int)(i % HW);
  const int64_t bc = i / HW;
  const int c = (int)(bc % C), b = (int)(bc / C);
  y[i] = sr_load_f(x + ((int64_t)b * HW + p) * ldc + c);
}

template <typename T>
__global__ void temb_kernel(const float* __restrict__ t, T* __restrict__ y, int B, int dim) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * dim) return;
  const int b = i / dim, j = i - b * dim, half = dim / 2;
  const int k = j < half ? j : j - half;
  // freqs = exp(-ln(10000) * k / half) in fp32, args = t * freqs (util.py:250-256)
  const float freq = expf(-9.210340371976184f * (float)k / (float)half);
  const float a = t[b] * freq;
  sr_store_f(y + i, j < half ? cosf(a) : sinf(a));
}

template <typename T>
__global__ void silu_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) sr_store_f(y + i, sr_silu_f(sr_load_f(x + i)));
}

template <typename TS, typename TD>
__global__ void cast_kernel(const TS* __restrict__ x, TD* __restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) sr_store_f(y + i, sr_load_f(x + i));
}

// in-place row softmax, one block per row (VAE mid-block attention, 4096 keys)
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(T* __restrict__ x, int cols) {
  __shared__ float red[4];
  T* row = x + (int64_t)blockIdx.x * cols;
  const int tid = threadIdx.x;
  float mx = -INFINITY;
  for (int i = tid; i < cols; i += 256) mx = fmaxf(mx, sr_load_f(row + i));
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int i = tid; i < cols; i += 256) s += __expf(sr_load_f(row + i) - mx);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  s = (red[0] + red[1]) + (red[2] + red[3]);
  const float inv = 1.0f / s;
  for (int i = tid; i < cols; i += 256) sr_store_f(row + i, __expf(sr_load_f(row + i) - mx) * inv);
}

template <typename T>
__global__ void add_scaled_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, int64_t n, float s) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) sr_store_f(y + i, sr_load_f(a + i) + s * sr_load_f(b + i));
}

// y[j] = x[sel[j]]; an index outside [0, n_rows) never reaches memory: the row is zero-filled and *err is raised
__global__ void gather_rows_kernel(const uint4* __restrict__ x, const int* __restrict__ sel, uint4* __restrict__ y, int nsel, int n_rows,
                                   int64_t row_chunks, int* __restrict__ err) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)nsel * row_chunks) return;
  const int j = (int)(i / row_chunks);
  const int64_t c = i - (int64_t)j * row_chunks;
  const int r = sel[j];
  if (r < 0 || r >= n_rows) {
    y[i] = make_uint4(0u, 0u, 0u, 0u);
    if (c == 0 && err) atomicOr(err, 1);
    return;
  }
  y[i] = x[(int64_t)r * row_chunks + c];
}

__global__ void eps_scale_kernel(const float* __restrict__ x, float* __restrict__ xin, int64_t n, int copies, float inv) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i] * inv;
  xin[i] = v;
  if (copies == 2) xin[n + i] = v;
}

__global__ void cfg_denoise_kernel(const float* __restrict__ x, const float* __restrict__ eps, float* __restrict__ den,
                                   float* __restrict__ d, int64_t n, int copies, float sigma, float cfg) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float xv = x[i];
  float r;
  if (copies == 2) {
    const float u = xv - eps[i] * sigma, c = xv - eps[n + i] * sigma;       // calculate_denoised per chunk
    r = u + (c - u) * cfg;                                                  // samplers.py:351
  } else {
    r = xv - eps[i] * sigma;
  }
  den[i] = r;
  if (d) d[i] = (xv - r) / sigma;                                           // to_d
}

// ---- conditioning composition (comfy/samplers.py:50-127 get_area_and_mult, :176-320 calc_cond_uncond_batch) -----------------
// xin[(j*N + n), c, yy, xx] = x[n, c, y0+yy, x0+xx] * inv for every chunk j of the group (all chunks see the same crop)
__global__ void cond_crop_scale_kernel(const float* __restrict__ x, float* __restrict__ xin, int N, int C, int h, int w, int ah, int aw,
                                       int y0, int x0, int chunks, float inv) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per = (int64_t)N * C * ah * aw;
  if (i >= per) return;
  const int xx = (int)(i % aw), yy = (int)((i / aw) % ah);
  const int64_t nc = i / ((int64_t)aw * ah);
  const float v = x[(nc * h + (y0 + yy)) * w + (x0 + xx)] * inv;
  for (int j = 0; j < chunks; ++j) xin[(int64_t)j * per + i] = v;
}
// for the chunks of one model call, IN BATCH ORDER:  out_kind[area] += (x - eps_j*sigma) * mult_j ; cnt_kind[area] += mult_j
__global__ void cond_accumulate_kernel(const float* __restrict__ x, const float* __restrict__ eps, const float* __restrict__ mult,
                                       const int* __restrict__ kinds, float* __restrict__ out_c, float* __restrict__ cnt_c,
                                       float* __restrict__ out_u, float* __restrict__ cnt_u, int N, int C, int h, int w, int ah, int aw,
                                       int y0, int x0, int chunks, float sigma) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per = (int64_t)N * C * ah * aw;
  if (i >= per) return;
  const int xx = (int)(i % aw), yy = (int)((i / aw) % ah);
  const int64_t nc = i / ((int64_t)aw * ah);
  const int64_t at = (nc * h + (y0 + yy)) * w + (x0 + xx);
  const float xv = x[at];
  float oc = out_c[at], cc = cnt_c[at], ou = out_u[at], cu = cnt_u[at];
  for (int j = 0; j < chunks; ++j) {
    const float den = xv - eps[(int64_t)j * per + i] * sigma;             // EPS.calculate_denoised on the cropped input
    const float m = mult[(int64_t)j * per + i];
    const float t = den * m;                                              // (output * mult) rounded, then added (:309-312)
    if (kinds[j] == 0) { oc = oc + t; cc = cc + m; } else { ou = ou + t; cu = cu + m; }
  }
  out_c[at] = oc; cnt_c[at] = cc; out_u[at] = ou; cnt_u[at] = cu;
}
// cond_pred = out_c/cnt_c, uncond_pred = out_u/cnt_u (counts start at 1e-37), cfg = u + (c-u)*scale, d = (x - cfg)/sigma
__global__ void cfg_combine_kernel(const float* __restrict__ x, const float* __restrict__ out_c, const float* __restrict__ cnt_c,
                                   const float* __restrict__ out_u, const float* __restrict__ cnt_u, float* __restrict__ den,
                                   float* __restrict__ d, int64_t n, float sigma, float cfg) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float c = out_c[i] / cnt_c[i], u = out_u[i] / cnt_u[i];
  const float r = u + (c - u) * cfg;
  den[i] = r;
  if (d) d[i] = (x[i] - r) / sigma;
}

// DiagonalGaussianDistribution.sample (distributions.py:24-37): moments NHWC [mean(zc) | logvar(zc)] -> z NCHW
__global__ void vae_sample_kernel(const float* __restrict__ mom, const float* __restrict__ noise, float* __restrict__ z, int B, int zc, int HW) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * zc * HW) return;
  const int p = (int)(i % HW), c = (int)((i / HW) % zc), b = (int)(i / ((int64_t)HW * zc));
  const float* m = mom + ((int64_t)b * HW + p) * (2 * zc);
  const float logvar = fminf(fmaxf(m[zc + c], -30.0f), 20.0f);
  z[i] = m[c] + expf(0.5f * logvar) * noise[i];
}

__global__ void euler_kernel(float* __restrict__ x, const float* __restrict__ d, int64_t n, float dt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = x[i] + d[i] * dt;
}

// generic_step_sampler + DDPMSampler_step (k_diffusion/sampling.py:749-776), scalars precomputed on the host
__global__ void ddpm_kernel(float* __restrict__ x, const float* __restrict__ den, const float* __restrict__ noise, int64_t n,
                            float sigma, float in_scale, float c_mu, float c_eps, float c_noise, float out_scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float xv = x[i];
  const float e = (xv - den[i]) / sigma;
  float mu = c_mu * (xv * in_scale - c_eps * e);
  if (noise) mu += c_noise * noise[i];
  x[i] = mu * out_scale;
}

__global__ void lcm_kernel(float* __restrict__ x, const float* __restrict__ den, const float* __restrict__ noise, int64_t n, float sn) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = den[i];
  if (noise) v += sn * noise[i];
  x[i] = v;
}

__global__ void axpby_kernel(float* __restrict__ y, const float* __restrict__ x, int64_t n, float a, float b) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = a * x[i] + b * y[i];
}

inline dim3 g1(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

extern "C" int sr_nchw_to_nhwc(const float* x, void* y, int32_t B, int32_t C, int32_t HW, int32_t Cpad, float scale_mul,
                               const float* pbs, int32_t dtype, void* stream) {
  if (!x || !y || Cpad < C) SR_FAIL(SR_ERR_INVALID, "sr_nchw_to_nhwc: bad args");
  const int64_t n = (int64_t)B * HW * Cpad;
  if (dtype == SR_F16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<_Float16>, g1(n), dim3(256), 0, sr_stream(stream), x, (_Float16*)y, B, C, HW, Cpad, scale_mul, pbs);
  else hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, g1(n), dim3(256), 0, sr_stream(stream), x, (float*)y, B, C, HW, Cpad, scale_mul, pbs);
  SR_CHECK_LAUNCH("sr_nchw_to_nhwc");
  return SR_OK;
}

extern "C" int sr_nhwc_to_nchw(const void* x, float* y, int32_t B, int32_t C, int32_t HW, int32_t ldc, int32_t dtype, void* stream) {
  if (!x || !y || ldc < C) SR_FAIL(SR_ERR_INVALID, "sr_nhwc_to_nchw: bad args");
  const int64_t n = (int64_t)B * HW * C;
  if (dtype == SR_F16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<_Float16>, g1(n), dim3(256), 0, sr_stream(stream), (const _Float16*)x, y, B, C, HW, ldc);
  else hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, g1(n), dim3(256), 0, sr_stream(stream), (const float*)x, y, B, C, HW, ldc);
  SR_CHECK_LAUNCH("sr_nhwc_to_nchw");
  return SR_OK;
}

extern "C" int sr_timestep_embedding(const float* t, void* y, int32_t B, int32_t dim, int32_t dtype, void* stream) {
  if (!t || !y || dim % 2) SR_FAIL(SR_ERR_INVALID, "sr_timestep_embedding: bad args");
  if (dtype == SR_F16) hipLaunchKernelGGL(temb_kernel<_Float16>, g1((int64_t)B * dim), dim3(256), 0, sr_stream(stream), t, (_Float16*)y, B, dim);
  else hipLaunchKernelGGL(temb_kernel<float>, g1((int64_t)B * dim), dim3(256), 0, sr_stream(stream), t, (float*)y, B, dim);
  SR_CHECK_LAUNCH("sr_timestep_embedding");
  return SR_OK;
}

extern "C" int sr_silu(const void* x, void* y, int64_t n, int32_t dtype, void* stream) {
  if (!x || !y) SR_FAIL(SR_ERR_INVALID, "sr_silu: null");
  if (dtype == SR_F16) hipLaunchKernelGGL(silu_kernel<_Float16>, g1(n), dim3(256), 0, sr_stream(stream), (const _Float16*)x, (_Float16*)y, n);
  else hipLaunchKernelGGL(silu_kernel<float>, g1(n), dim3(256), 0, sr_stream(stream), (const float*)x, (float*)y, n);
  SR_CHECK_LAUNCH("sr_silu");
  return SR_OK;
}

// reads one 16-byte chunk of every 64 bytes of [p, p + bytes): the range ends up in L2 / Infinity Cache as if a producer kernel had
// just written it.  The host-side tile tuner (ops.tune_igemm, SR_TUNE_COLD) times a layer behind a cache flush + this pass over its
// ACTIVATIONS, i.e. in the state it meets inside a plan: inputs fresh from the previous kernel, weights in HBM.
__global__ __launch_bounds__(256) void cache_touch_kernel(const uint4* __restrict__ p, const int64_t n64, unsigned* sink) {
  unsigned acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n64; i += (int64_t)gridDim.x * 256) acc ^= p[i * 4].x;
  if (acc == 0x9e3779b9u && sink) *sink = acc;               // (keeps the loads; practically never taken)
}

extern "C" int sr_cache_touch(const void* p, int64_t bytes, void* stream) {
  if (!p || bytes < 0 || ((uintptr_t)p & 15)) SR_FAIL(SR_ERR_INVALID, "sr_cache_touch: null / unaligned pointer or negative size");
  const int64_t n64 = bytes >> 6;
  if (n64 == 0) return SR_OK;
  const int blocks = (int)((n64 + 255) / 256 < 4096 ? (n64 + 255) / 256 : 4096);
  hipLaunchKernelGGL(cache_touch_kernel, dim3(blocks), dim3(256), 0, sr_stream(stream), (const uint4*)p, n64, (unsigned*)nullptr);
  SR_CHECK_LAUNCH("sr_cache_touch");
  return SR_OK;
}

extern "C" int sr_cast(const void* x, int32_t sd, void* y, int32_t dd, int64_t n, void* stream) {
  if (!x || !y) SR_FAIL(SR_ERR_INVALID, "sr_cast: null");
  hipStream_t st = sr_stream(stream);
  if (sd == SR_F32 && dd == SR_F16) hipLaunchKernelGGL((cast_kernel<float, _Float16>), g1(n), dim3(256), 0, st, (const float*)x, (_Float16*)y, n);
  else if (sd == SR_F16 && dd == SR_F32) hipLaunchKernelGGL((cast_kernel<_Float16, float>), g1(n), dim3(256), 0, st, (const _Float16*)x, (float*)y, n);
  else if (sd == SR_F32 && dd == SR_F32) hipLaunchKernelGGL((cast_kernel<float, float>), g1(n), dim3(256), 0, st, (const float*)x, (float*)y, n);
  else hipLaunchKernelGGL((cast_kernel<_Float16, _Float16>), g1(n), dim3(256), 0, st, (const _Float16*)x, (_Float16*)y, n);
  SR_CHECK_LAUNCH("sr_cast");
  return SR_OK;
}

extern "C" int sr_softmax_rows(void* x, int32_t rows, int32_t cols, int32_t dtype, void* stream) {
  if (!x) SR_FAIL(SR_ERR_INVALID, "sr_softmax_rows: null");
  if (dtype == SR_F16) hipLaunchKernelGGL(softmax_rows_kernel<_Float16>, dim3(rows), dim3(256), 0, sr_stream(stream), (_Float16*)x, cols);
  else hipLaunchKernelGGL(softmax_rows_kernel<float>, dim3(rows), dim3(256), 0, sr_stream(stream), (float*)x, cols);
  SR_CHECK_LAUNCH("sr_softmax_rows");
  return SR_OK;
}

extern "C" int sr_add_scaled(const void* a, const void* b, void* y, int64_t n, float s, int32_t dtype, void* stream) {
  if (!a || !b || !y) SR_FAIL(SR_ERR_INVALID, "sr_add_scaled: null");
  if (dtype == SR_F16) hipLaunchKernelGGL(add_scaled_kernel<_Float16>, g1(n), dim3(256), 0, sr_stream(stream), (const _Float16*)a, (const _Float16*)b, (_Float16*)y, n, s);
  else hipLaunchKernelGGL(add_scaled_kernel<float>, g1(n), dim3(256), 0, sr_stream(stream), (const float*)a, (const float*)b, (float*)y, n, s);
  SR_CHECK_LAUNCH("sr_add_scaled");
  return SR_OK;
}

extern "C" int sr_gather_rows(const void* x, const int32_t* sel, void* y, int32_t nsel, int32_t n_rows, int64_t row_bytes,
                              int32_t* err_flag, void* stream) {
  if (!x || !sel || !y || row_bytes <= 0 || row_bytes % 16 || nsel < 0 || n_rows < 1) SR_FAIL(SR_ERR_INVALID, "sr_gather_rows: bad args");
  if (nsel == 0) return SR_OK;
  const int64_t rc = row_bytes / 16;
  hipLaunchKernelGGL(gather_rows_kernel, g1((int64_t)nsel * rc), dim3(256), 0, sr_stream(stream), (const uint4*)x, sel, (uint4*)y, nsel, n_rows, rc,
                     err_flag);
  SR_CHECK_LAUNCH("sr_gather_rows");
  return SR_OK;
}

extern "C" int sr_eps_scale_input(const float* x, float* xin, int64_t n, int32_t copies, float sigma, void* stream) {
  if (!x || !xin || (copies != 1 && copies != 2)) SR_FAIL(SR_ERR_INVALID, "sr_eps_scale_input: bad args");
  const float inv = 1.0f / sqrtf(sigma * sigma + 1.0f);     // x / (sigma^2 + 1)^0.5
  hipLaunchKernelGGL(eps_scale_kernel, g1(n), dim3(256), 0, sr_stream(stream), x, xin, n, copies, inv);
  SR_CHECK_LAUNCH("sr_eps_scale_input");
  return SR_OK;
}

extern "C" int sr_cfg_denoise(const float* x, const float* eps, float* den, float* d, int64_t n, int32_t copies, float sigma,
                              float cfg, void* stream) {
  if (!x || !eps || !den || (copies != 1 && copies != 2)) SR_FAIL(SR_ERR_INVALID, "sr_cfg_denoise: bad args");
  hipLaunchKernelGGL(cfg_denoise_kernel, g1(n), dim3(256), 0, sr_stream(stream), x, eps, den, d, n, copies, sigma, cfg);
  SR_CHECK_LAUNCH("sr_cfg_denoise");
  return SR_OK;
}

extern "C" int sr_cond_crop_scale(const float* x, float* xin, int32_t N, int32_t C, int32_t h, int32_t w, int32_t ah, int32_t aw, int32_t y0,
                                  int32_t x0, int32_t chunks, float sigma, void* stream) {
  if (!x || !xin || chunks < 1 || ah < 1 || aw < 1 || y0 < 0 || x0 < 0 || y0 + ah > h || x0 + aw > w)
    SR_FAIL(SR_ERR_INVALID, "sr_cond_crop_scale: bad args (area outside the latent?)");
  const float inv = 1.0f / sqrtf(sigma * sigma + 1.0f);
  hipLaunchKernelGGL(cond_crop_scale_kernel, g1((int64_t)N * C * ah * aw), dim3(256), 0, sr_stream(stream), x, xin, N, C, h, w, ah, aw, y0, x0,
                     chunks, inv);
  SR_CHECK_LAUNCH("sr_cond_crop_scale");
  return SR_OK;
}

extern "C" int sr_cond_accumulate(const float* x, const float* eps, const float* mult, const int32_t* kinds, float* out_c, float* cnt_c,
                                  float* out_u, float* cnt_u, int32_t N, int32_t C, int32_t h, int32_t w, int32_t ah, int32_t aw,
                                  int32_t y0, int32_t x0, int32_t chunks, float sigma, void* stream) {
  if (!x || !eps || !mult || !kinds || !out_c || !cnt_c || !out_u || !cnt_u || chunks < 1 || ah < 1 || aw < 1 || y0 < 0 || x0 < 0 ||
      y0 + ah > h || x0 + aw > w)
    SR_FAIL(SR_ERR_INVALID, "sr_cond_accumulate: bad args (area outside the latent?)");
  hipLaunchKernelGGL(cond_accumulate_kernel, g1((int64_t)N * C * ah * aw), dim3(256), 0, sr_stream(stream), x, eps, mult, kinds, out_c, cnt_c,
                     out_u, cnt_u, N, C, h, w, ah, aw, y0, x0, chunks, sigma);
  SR_CHECK_LAUNCH("sr_cond_accumulate");
  return SR_OK;
}

extern "C" int sr_cfg_combine(const float* x, const float* out_c, const float* cnt_c, const float* out_u, const float* cnt_u,
                              float* denoised, float* d, int64_t n, float sigma, float cfg, void* stream) {
  if (!x || !out_c || !cnt_c || !out_u || !cnt_u || !denoised) SR_FAIL(SR_ERR_INVALID, "sr_cfg_combine: null");
  hipLaunchKernelGGL(cfg_combine_kernel, g1(n), dim3(256), 0, sr_stream(stream), x, out_c, cnt_c, out_u, cnt_u, denoised, d, n, sigma, cfg);
  SR_CHECK_LAUNCH("sr_cfg_combine");
  return SR_OK;
}

extern "C" int sr_vae_sample(const float* moments, const float* noise, float* z, int32_t B, int32_t zc, int32_t HW, void* stream) {
  if (!moments || !noise || !z || B < 1 || zc < 1 || HW < 1) SR_FAIL(SR_ERR_INVALID, "sr_vae_sample: bad args");
  hipLaunchKernelGGL(vae_sample_kernel, g1((int64_t)B * zc * HW), dim3(256), 0, sr_stream(stream), moments, noise, z, B, zc, HW);
  SR_CHECK_LAUNCH("sr_vae_sample");
  return SR_OK;
}

extern "C" int sr_euler_step(float* x, const float* d, int64_t n, float dt, void* stream) {
  if (!x || !d) SR_FAIL(SR_ERR_INVALID, "sr_euler_step: null");
  hipLaunchKernelGGL(euler_kernel, g1(n), dim3(256), 0, sr_stream(stream), x, d, n, dt);
  SR_CHECK_LAUNCH("sr_euler_step");
  return SR_OK;
}

extern "C" int sr_ddpm_step(float* x, const float* den, const float* noise, int64_t n, float sigma, float sigma_next, void* stream) {
  if (!x || !den) SR_FAIL(SR_ERR_INVALID, "sr_ddpm_step: null");
  // scalar algebra of DDPMSampler_step in fp32, as torch does on 0-d fp32 tensors
  const float in_scale = 1.0f / sqrtf(1.0f + sigma * sigma);
  const float ac = 1.0f / (sigma * sigma + 1.0f), acp = 1.0f / (sigma_next * sigma_next + 1.0f);
  const float alpha = ac / acp;
  const float c_mu = sqrtf(1.0f / alpha);
  const float c_eps = (1.0f - alpha) / sqrtf(1.0f - ac);
  float c_noise = 0.f;
  const float* nz = nullptr;
  if (sigma_next > 0.f) {
    if (!noise) SR_FAIL(SR_ERR_INVALID, "sr_ddpm_step: noise required when sigma_next > 0");
    c_noise = sqrtf((1.0f - alpha) * (1.0f - acp) / (1.0f - ac));
    nz = noise;
  }
  const float out_scale = sigma_next != 0.f ? sqrtf(1.0f + sigma_next * sigma_next) : 1.0f;
  hipLaunchKernelGGL(ddpm_kernel, g1(n), dim3(256), 0, sr_stream(stream), x, den, nz, n, sigma, in_scale, c_mu, c_eps, c_noise, out_scale);
  SR_CHECK_LAUNCH("sr_ddpm_step");
  return SR_OK;
}

extern "C" int sr_lcm_step(float* x, const float* den, const float* noise, int64_t n, float sigma_next, void* stream) {
  if (!x || !den) SR_FAIL(SR_ERR_INVALID, "sr_lcm_step: null");
  const float* nz = sigma_next > 0.f ? noise : nullptr;
  if (sigma_next > 0.f && !noise) SR_FAIL(SR_ERR_INVALID, "sr_lcm_step: noise required");
  hipLaunchKernelGGL(lcm_kernel, g1(n), dim3(256), 0, sr_stream(stream), x, den, nz, n, sigma_next);
  SR_CHECK_LAUNCH("sr_lcm_step");
  return SR_OK;
}

extern "C" int sr_axpby(float* y, const float* x, int64_t n, float a, float b, void* stream) {
  if (!x || !y) SR_FAIL(SR_ERR_INVALID, "sr_axpby: null");
  hipLaunchKernelGGL(axpby_kernel, g1(n), dim3(256), 0, sr_stream(stream), y, x, n, a, b);
  SR_CHECK_LAUNCH("sr_axpby");
  return SR_OK;
}
