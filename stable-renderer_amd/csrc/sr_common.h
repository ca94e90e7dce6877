// Shared device/host helpers for libsr_hip (gfx950 only: wave64, MFMA 16x16x32 f16 / 16x16x4 f32).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/sr_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

void sr_set_error(const char* fmt, ...);
#define SR_FAIL(code, ...) do { sr_set_error(__VA_ARGS__); return (code); } while (0)
#define SR_CHECK_LAUNCH(name) do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { \
    sr_set_error("%s: %s", name, hipGetErrorString(e_)); return SR_ERR_LAUNCH; } } while (0)

template <typename T> struct sr_traits;
template <> struct sr_traits<_Float16> { static constexpr int EPC = 8;  /* elements per 16-byte chunk */ };
template <> struct sr_traits<float>    { static constexpr int EPC = 4; };

// One "macro step" of the 16x16 MFMA family on a pair of 16-byte operand chunks (lane (g = lane>>4,
// c = lane&15) holds row/col c, k-slots [g*EPC, g*EPC+EPC) of a 4*EPC deep slab):
//   fp16: one v_mfma_f32_16x16x32_f16 (K = 32);  fp32: four v_mfma_f32_16x16x4_f32 (element s of every lane
//   forms k-slab s; any k partition is valid as long as A and B use the same one).
__device__ __forceinline__ void sr_mma(f32x4& acc, const uint4& a, const uint4& b, _Float16) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), acc, 0, 0, 0);
}
__device__ __forceinline__ void sr_mma(f32x4& acc, const uint4& a, const uint4& b, float) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

// async global -> LDS, 16 bytes per lane; LDS destination = wave-uniform base + lane*16.
__device__ __forceinline__ void sr_glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Same transfer issued from inline asm: hipcc then does NOT see a pending LDS write, so it does not drain the whole
// pipeline (s_waitcnt vmcnt(0)) in front of the next ds_read; the caller owns the ordering: counted s_waitcnt vmcnt(N)
// for the stage it is about to read, then a raw s_barrier.  M0 (LDS base of the wave) is saved/restored inside the
// statement because it is compiler-reserved.  lds_addr must be wave-uniform.
__device__ __forceinline__ void sr_glds16_asm(const void* gsrc, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}
// burst form: M0 is saved once before and restored once after a run of LDS-DMA pieces (nothing between the pieces of a
// burst touches M0: address selects and 64-bit adds only)
__device__ __forceinline__ unsigned sr_m0_save() {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0" : "=s"(keep));
  return keep;
}
__device__ __forceinline__ void sr_m0_restore(unsigned keep) { asm volatile("s_mov_b32 m0, %0" ::"s"(keep)); }
__device__ __forceinline__ void sr_glds16_asm_nosave(const void* gsrc, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_addr) : "memory");
}
// same with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset: one VGPR per stream instead of two
__device__ __forceinline__ void sr_glds16_asm_saddr(unsigned voff, const void* sbase, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
// Cache warm-up of bytes a LATER launch will stream (sr_igemm_args.prefetch): the threads of the grid share the range, one
// 4-byte LDS-DMA read per 64 bytes into a 256-byte per-wave scratch nobody reads (no VGPR destination, so nothing the compiler
// could reuse while the load is in flight).  These are the oldest vector-memory requests of the wave: every counted
// s_waitcnt vmcnt(N) that follows also covers them (requests retire in issue order).
__device__ __forceinline__ void sr_prefetch_touch(const void* base, int64_t bytes, int64_t gtid, int64_t gthreads, unsigned lds_scratch_wave) {
  const int64_t n = bytes >> 6;                              // 64-byte sectors
  const unsigned keep = sr_m0_save();
  for (int64_t i = gtid; i < n; i += gthreads) {
    const char* src = (const char*)base + (i << 6);
    // M0 is set in the SAME statement as the load that reads it: between two asm statements the compiler may use M0 itself
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(src), "s"(lds_scratch_wave) : "memory");
  }
  sr_m0_restore(keep);
}
__device__ __forceinline__ unsigned sr_lds_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}

__device__ __forceinline__ float sr_silu_f(float x) { return x / (1.0f + __expf(-x)); }
// exact-erf GELU (torch F.gelu default).  erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, below fp32 GEMM noise and
// far below fp16 resolution): one v_rcp, one v_exp, six FMAs and no divergent branch -- libm's erff costs about three
// times that in the GEGLU epilogue (84 M evaluations per 64x64-level FF layer).
__device__ __forceinline__ float sr_gelu_f(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float e = poly * t * __expf(-z * z);               // 1 - erf(|x|/sqrt2)
  const float cdf = x >= 0.0f ? 1.0f - 0.5f * e : 0.5f * e; // Phi(x)
  return x * cdf;
}

// (A polynomial erf without v_rcp / v_exp for the fp16 GEGLU epilogue -- degree 9 in u = 2 z^2 / 3.7^2 - 1, |GELU error| <= 1.8e-4 --
//  was built and timed in round 4: 190.6 vs 194.3 us on the 64x64 GEGLU projection, 162.1 vs 159.6 and 137.1 vs 138.4 on the other
//  two: noise.  The epilogue's cost is not the GELU arithmetic; the 1.5e-7 form above stays.)
__device__ __forceinline__ float sr_load_f(const _Float16* p) { return (float)*p; }
__device__ __forceinline__ float sr_load_f(const float* p) { return *p; }
__device__ __forceinline__ void sr_store_f(_Float16* p, float v) { *p = (_Float16)v; }
__device__ __forceinline__ void sr_store_f(float* p, float v) { *p = v; }

// bijective XCD-aware remap of a linear workgroup id (blocks b and b+8 share an XCD; give each XCD a
// contiguous chunk of the tile space so neighbouring tiles hit the same L2).
__device__ __forceinline__ int sr_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// tile sequence number -> (M-tile, N-tile): rows first (an XCD's contiguous run covers whole bands of M-tiles: activations stay in
// its L2) or columns first (bands of N-tiles: the packed weights stay) -- sr_igemm_args.tile_order
__device__ __forceinline__ void sr_tile_of(int wg, int NT, int MT, int order, int& mt, int& nt) {
  if (order == 1) { nt = wg / MT; mt = wg - nt * MT; }
  else            { mt = wg / NT; nt = wg - mt * NT; }
}

static inline hipStream_t sr_stream(void* s) { return (hipStream_t)s; }
static inline int sr_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
