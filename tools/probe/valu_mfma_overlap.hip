// Probe: do VALU ops (v_pk_fma_f32 / v_fma_f32 / v_pk_mul_f32) overlap with v_mfma_scale_f32_16x16x128_f8f6f4 (FP4)
// on one SIMD, within a wave and across two waves?  Every sequence is inline asm, no data dependencies
// between the MFMAs and the VALU ops.  Reports wall time per loop body and SIMD cycles at the measured clock.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

#ifdef UNSCALED
#define MFMA(acc) asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0 cbsz:4 blgp:4" : "+v"(acc) : "v"(a), "v"(b), "v"(sc))
#else
#define MFMA(acc) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:4 blgp:4" : "+v"(acc) : "v"(a), "v"(b), "v"(sc))
#endif
#define PKFMA(p) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(q), "v"(r))
#define PKMUL(p) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(q))
#define FMA(ff) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(ff) : "v"(q1), "v"(r1))

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {5, (int)threadIdx.x, 7, 8};
  int sc = 127;
  v4f acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  v2f p[8], q = {1.0f, 1.0f}, r = {0.0f, 0.0f};
  float f[8], q1 = 1.0f, r1 = 0.0f;
  for (int i = 0; i < 8; ++i) { p[i] = v2f{(float)i, 1.0f}; f[i] = (float)i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (MODE != 1 && MODE != 2 && MODE != 7) MFMA(acc[i]);
      if (MODE == 1 || MODE == 3) { PKFMA(p[2 * i]); PKFMA(p[2 * i + 1]); PKFMA(p[(2 * i + 2) & 7]); PKFMA(p[(2 * i + 3) & 7]); }
      if (MODE == 2 || MODE == 4) { FMA(f[2 * i]); FMA(f[2 * i + 1]); FMA(f[(2 * i + 2) & 7]); FMA(f[(2 * i + 3) & 7]); }
      if (MODE == 5) { PKFMA(p[2 * i]); PKFMA(p[2 * i + 1]); }
      if (MODE == 6) { FMA(f[2 * i]); FMA(f[2 * i + 1]); FMA(f[(2 * i + 2) & 7]); FMA(f[(2 * i + 3) & 7]); FMA(f[(2 * i + 4) & 7]); FMA(f[(2 * i + 5) & 7]); FMA(f[(2 * i + 6) & 7]); FMA(f[(2 * i + 7) & 7]); }
      if (MODE == 7 || MODE == 8) { PKMUL(p[2 * i]); PKMUL(p[2 * i + 1]); PKMUL(p[(2 * i + 2) & 7]); PKMUL(p[(2 * i + 3) & 7]); }
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1] + f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* tag, int per_body_mfma, int per_body_valu) {
  float* out;
  hipMalloc(&out, 4 * 512 * 256);
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int threads = 256; threads <= 512; threads *= 2) {
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ns_body = ms * 1e6 / (iters * 4.0);   // one "body" = 1 MFMA slot + its VALU ops, per wave
    printf("%-34s %d wave/SIMD: %.2f ns per body per wave => %.2f ns per body per SIMD  (%d mfma + %d valu)\n", tag, threads / 256,
           ns_body, ns_body / (threads / 256), per_body_mfma, per_body_valu);
  }
  hipFree(out);
}

int main() {
  run<0>("mfma only", 1, 0);
  run<1>("4 pk_fma only", 0, 4);
  run<2>("4 v_fma only", 0, 4);
  run<7>("4 pk_mul only", 0, 4);
  run<3>("mfma + 4 pk_fma", 1, 4);
  run<4>("mfma + 4 v_fma", 1, 4);
  run<8>("mfma + 4 pk_mul", 1, 4);
  run<5>("mfma + 2 pk_fma", 1, 2);
  run<6>("mfma + 8 v_fma", 1, 8);
  return 0;
}
