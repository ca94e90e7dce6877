// Micro-benchmark (development tool): how many bytes per second can all CUs stage from an L2-resident window
//   mode 0: global_load_lds_dwordx4 (LDS-DMA, what igemm.hip uses)      mode 1: global_load_dwordx4 into registers
// Each workgroup (512 threads) streams `iters` x 8 KiB pieces from a window of `win_kb` KiB that all workgroups of an XCD share
// (so after the first pass everything hits L2), `depth` wave-instructions in flight per wave.
// build: hipcc --offload-arch=gfx950 -O3 -o stage_bw tools/micro/stage_bw.hip ; run: ./stage_bw [win_kb] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_addr) : "memory");
}

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void stage_kernel(const char* __restrict__ src, size_t win_bytes, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const size_t xcd_win = (size_t)(blockIdx.x & 7) * win_bytes;     // consecutive workgroups land on different XCDs
  const char* base = src + xcd_win;
  unsigned acc = 0;
  size_t off = ((size_t)(blockIdx.x >> 3) * 8192 * 7) % win_bytes;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wv * 1024 * DEPTH;
  for (int it = 0; it < iters; it += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const char* p = base + ((off + (size_t)d * 8192) % win_bytes) + wv * 1024 + lane * 16;
      if (MODE == 0) glds16(p, __builtin_amdgcn_readfirstlane(lds0 + d * 1024));
      else { const uint4 v = *(const uint4*)p; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    off = (off + (size_t)DEPTH * 8192) % win_bytes;
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (MODE == 0) acc = *(volatile unsigned*)(smem + tid * 4);
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int DEPTH>
double run(const char* src, size_t win, int iters, unsigned* sink, int wgs, int lds) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)stage_kernel<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  stage_kernel<MODE, DEPTH><<<wgs, 512, lds>>>(src, win, iters, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) stage_kernel<MODE, DEPTH><<<wgs, 512, lds>>>(src, win, iters, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return (double)wgs * iters * 8192.0 * 5 / (ms * 1e-3) / 1e12;
}

int main(int argc, char** argv) {
  const size_t win_kb = argc > 1 ? atoi(argv[1]) : 2048;
  const int iters = argc > 2 ? atoi(argv[2]) : 2048;
  const size_t win = win_kb * 1024;
  char* src; unsigned* sink;
  hipMalloc(&src, win * 8 + (1 << 20)); hipMemset(src, 1, win * 8 + (1 << 20)); hipMalloc(&sink, 64);
  for (int wgs : {256, 512}) {
    printf("window %zu KiB per XCD, %d workgroups x 512 threads, %d x 8 KiB each\n", win_kb, wgs, iters);
    printf("  LDS-DMA  depth 2: %.2f TB/s   depth 4: %.2f TB/s   depth 8: %.2f TB/s\n", run<0, 2>(src, win, iters, sink, wgs, 16384 * 2),
           run<0, 4>(src, win, iters, sink, wgs, 32768 * 2), run<0, 8>(src, win, iters, sink, wgs, 65536));
    printf("  register depth 2: %.2f TB/s   depth 4: %.2f TB/s   depth 8: %.2f TB/s\n", run<1, 2>(src, win, iters, sink, wgs, 1024),
           run<1, 4>(src, win, iters, sink, wgs, 1024), run<1, 8>(src, win, iters, sink, wgs, 1024));
  }
  return 0;
}
