// mfma_fp6_rate.hip - issue rate of v_mfma_f32_16x16x128_f8f6f4 by operand format AND operand register-tuple width.
// Round 1 measured FP6 (E2M3) at the FP8 rate (20 ns per SIMD, 3.4 PFLOP/s) where MI355X_MICROARCH.md says FP6 issues at
// the FP4 rate - through the builtin, whose source-level operands are 8-element vectors whatever the format (the assembler
// rejects anything but the format's own tuple width, so the compiler narrows them).  Here the instruction is written in
// assembly with the tuple width the format needs (FP8: 8 VGPRs, FP6: 6, FP4: 4), beside the builtin, both FP6
// selectors (cbsz / blgp 2 = E2M3, 3 = E3M2), the unscaled and the scaled form, one wavefront per SIMD, 8 independent
// accumulators.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_fp6_rate mfma_fp6_rate.hip && ./mfma_fp6_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v6i __attribute__((ext_vector_type(6)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define ASM8(fmt) asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0 cbsz:" #fmt " blgp:" #fmt : "+v"(acc[i]) : "v"(a8), "v"(b8))
#define ASM6(fmt) asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0 cbsz:" #fmt " blgp:" #fmt : "+v"(acc[i]) : "v"(a6), "v"(b6))
#define ASM4(fmt) asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0 cbsz:" #fmt " blgp:" #fmt : "+v"(acc[i]) : "v"(a4), "v"(b4))
#define ASM6S(fmt) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:" #fmt " blgp:" #fmt : "+v"(acc[i]) : "v"(a6), "v"(b6), "v"(sc))

template <int MODE>
__global__ __launch_bounds__(256) void rate(float* out, int iters) {
  const int t = threadIdx.x;
  v8i a8 = {t, 1, 2, 3, 4, 5, 6, 7}, b8 = {5, t, 7, 8, 9, 1, 2, 3};
  v6i a6 = {t, 1, 2, 3, 4, 5}, b6 = {5, t, 7, 8, 9, 1};
  v4i a4 = {t, 1, 2, 3}, b4 = {5, t, 7, 8};
  int sc = 127;
  v4f acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = v4f{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) ASM8(0);        // FP8 E4M3, 8 registers
      if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[i], 2, 2, 0, 0, 0, 0);   // the builtin (8-element vectors in the source)
      if (MODE == 2) ASM6(2);        // FP6 E2M3, 6 registers
      if (MODE == 3) ASM6(3);        // FP6 E3M2, 6 registers
      if (MODE == 4) ASM4(4);        // FP4 E2M1, 4 registers
      if (MODE == 5) ASM6S(2);       // FP6 E2M3, 6 registers, scaled form
    }
  float r = 0;
  for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + t] = r;
}

template <int MODE>
void run(const char* what, float* out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 2000;
  hipLaunchKernelGGL(rate<MODE>, dim3(256), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(rate<MODE>, dim3(256), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %6.2f ns per MFMA per SIMD => %5.2f PFLOP/s chip\n", what, ms * 1e6 / (iters * 8.0), 65536.0 * iters * 8 * 1024 / (ms * 1e-3) / 1e15);
}

int main() {
  float* out;
  (void)hipMalloc(&out, 4 * 256 * 256);
  run<0>("FP8 E4M3, 8-register operands", out);
  run<1>("FP6 E2M3 through the builtin", out);
  run<2>("FP6 E2M3, 6-register operands", out);
  run<3>("FP6 E3M2, 6-register operands", out);
  run<5>("FP6 E2M3, 6-register operands, scaled form", out);
  run<4>("FP4 E2M1, 4-register operands", out);
  return 0;
}
