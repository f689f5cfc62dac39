// Probe: issue rate of v_mfma_scale_f32_16x16x128_f8f6f4 (FP4 operands) per SIMD, alone and with packed-fp32
// VALU work interleaved, at 1 / 2 waves per SIMD.  Prints cycles per MFMA per SIMD from s_memtime deltas.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int NV>   // NV packed VALU ops per MFMA
__global__ __launch_bounds__(512) void rate(float* out, long long* cyc, int iters, float s) {
  v8i a = {(int)threadIdx.x, 1, 2, 3, 0, 0, 0, 0}, b = {5, (int)threadIdx.x, 7, 8, 0, 0, 0, 0};
  v4f acc[8];
  v2f p[8];
  for (int i = 0; i < 8; ++i) { acc[i] = v4f{0, 0, 0, 0}; p[i] = v2f{s, s + i}; }
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 4, 4, 0, 127, 0, 127);
#pragma unroll
      for (int v = 0; v < NV; ++v) p[(i + v) & 7] = __builtin_elementwise_fma(p[(i + v) & 7], v2f{s, s}, v2f{1.0f, 2.0f});
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float r = 0;
  for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][3] + p[i][0] + p[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NV>
void run(int threads, const char* tag) {
  float* out; long long* cyc;
  hipMalloc(&out, 4 * 512 * 256); hipMalloc(&cyc, 8);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(rate<NV>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 0.5f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(rate<NV>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double mf = (double)iters * 8;
  const int waves_per_simd = threads / 256;
  printf("%-28s threads %3d: %.3f ms, %.1f ns/MFMA/wave, counter ticks/MFMA %.2f, => %.1f PFLOP/s chip (FP4)\n", tag, threads, ms,
         ms * 1e6 / mf, (double)c / mf, 65536.0 * mf * (threads / 64) * 256 / (ms * 1e-3) / 1e15);
  (void)waves_per_simd;
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0>(256, "mfma only");
  run<0>(512, "mfma only");
  run<1>(256, "mfma + 1 pk_fma");
  run<2>(256, "mfma + 2 pk_fma");
  run<3>(256, "mfma + 3 pk_fma");
  run<4>(256, "mfma + 4 pk_fma");
  run<4>(512, "mfma + 4 pk_fma");
  run<6>(256, "mfma + 6 pk_fma");
  run<6>(512, "mfma + 6 pk_fma");
  return 0;
}
