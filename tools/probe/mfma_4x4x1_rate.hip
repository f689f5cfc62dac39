// Probe: issue rate of v_mfma_f32_4x4x1_16b_f32 (fp32 outer products of 16 blocks: one instruction = a 16x16 tile of
// products a[row] * b[col] in the 16x16 MFMA's result layout) alone, beside the FP4 v_mfma_f32_16x16x128_f8f6f4, and with
// 4 / 8 v_fma_f32 - would building the scale product sa x sw of the FP4 GEMM on the matrix pipe (and applying it with ONE
// fma per element instead of a multiply and an fma) pay?   hipcc --offload-arch=gfx950 -O3 -o /tmp/p tools/probe/mfma_4x4x1_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define BIG(acc) asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0 cbsz:4 blgp:4" : "+v"(acc) : "v"(a), "v"(b))
#define SMALL(acc) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(fa), "v"(fb))
#define FMA(ff) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(ff) : "v"(q1), "v"(r1))

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {5, (int)threadIdx.x, 7, 8};
  float fa = (float)threadIdx.x, fb = 1.0f + threadIdx.x;
  v4f acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  v4f cc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float f[8], q1 = 1.0f, r1 = 0.0f;
  for (int i = 0; i < 8; ++i) f[i] = (float)i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (MODE == 0 || MODE == 2 || MODE == 3 || MODE == 4) BIG(acc[i]);
      if (MODE == 1 || MODE == 2 || MODE == 3) SMALL(cc[i]);
      if (MODE == 3 || MODE == 5) { FMA(f[2 * i]); FMA(f[2 * i + 1]); FMA(f[(2 * i + 2) & 7]); FMA(f[(2 * i + 3) & 7]); }
      if (MODE == 4) { FMA(f[0]); FMA(f[1]); FMA(f[2]); FMA(f[3]); FMA(f[4]); FMA(f[5]); FMA(f[6]); FMA(f[7]); }
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + cc[i][0] + cc[i][1] + cc[i][2] + cc[i][3];
  for (int i = 0; i < 8; ++i) s += f[i];
  if (s == 12345.678f) out[0] = s;
}

template <int MODE>
static void run(const char* name) {
  float* out;
  hipMalloc(&out, 4);
  const int iters = 20000, blocks = 256;   // 512 threads = 8 wavefronts per CU = 2 per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: 2 wavefronts x iters x 4 bodies
  printf("%-44s %7.2f ns per body and SIMD\n", name, ms * 1e6 / (2.0 * iters * 4));
  hipFree(out);
}

int main() {
  run<0>("FP4 16x16x128 mfma only");
  run<1>("4x4x1 f32 mfma only");
  run<2>("FP4 mfma + 4x4x1 mfma");
  run<5>("4 v_fma_f32 only");
  run<3>("FP4 mfma + 4x4x1 mfma + 4 v_fma_f32");
  run<4>("FP4 mfma + 8 v_fma_f32 (today's body)");
  return 0;
}
