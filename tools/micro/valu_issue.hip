// Micro-benchmark (development tool): issue cost of a few VALU instructions, one wave per SIMD, 16 independent chains.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_issue tools/micro/valu_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = (float)(threadIdx.x + i) * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#define EXP32(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
#define EXP16(i) asm volatile("v_exp_f16 %0, %0" : "+v"(v[i]));
#define EXP16HI(i) asm volatile("v_exp_f16_sdwa %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(v[i]));
#define ADD32(i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v[i]));
#define CVTPK(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(v[i]));
#define PKMAX(i) asm volatile("v_pk_max_f16 %0, %0, %0" : "+v"(v[i]));
#define MAX3(i) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(v[i]));
#define PKFMA16(i) asm volatile("v_pk_fma_f16 %0, %0, %0, %0" : "+v"(v[i]));
    if (MODE == 0) { REP16(EXP32) REP16(EXP32) }
    if (MODE == 1) { REP16(EXP16) REP16(EXP16) }
    if (MODE == 2) { REP16(EXP16HI) REP16(EXP16HI) }
    if (MODE == 3) { REP16(ADD32) REP16(ADD32) }
    if (MODE == 4) { REP16(CVTPK) REP16(CVTPK) }
    if (MODE == 5) { REP16(PKMAX) REP16(PKMAX) }
    if (MODE == 6) { REP16(MAX3) REP16(MAX3) }
    if (MODE == 7) { REP16(PKFMA16) REP16(PKFMA16) }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += v[i];
  if (s == 123.456f) out[0] = s;
}

template <int MODE>
void run(const char* name, float* out) {
  const int iters = 4000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<256, 256>>>(out, iters);                     // 256 workgroups x 4 waves: one wave per SIMD
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<MODE><<<256, 256>>>(out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-22s %.2f ns per instruction per wave (x clock GHz = cycles)\n", name, ms * 1e6 / (iters * 32.0));
}

int main() {
  float* out; (void)hipMalloc(&out, 64);
  run<3>("v_add_f32", out); run<0>("v_exp_f32", out); run<1>("v_exp_f16", out); run<2>("v_exp_f16 op_sel hi", out);
  run<4>("v_cvt_pk_f16_f32", out); run<5>("v_pk_max_f16", out); run<6>("v_max3_f32", out); run<7>("v_pk_fma_f16", out);
  run<3>("v_add_f32 (again)", out);
  return 0;
}
