// mfma_valu_overlap.hip - do matrix-core instructions and vector instructions of DIFFERENT wavefronts of a SIMD overlap?
// Every wavefront runs a loop whose body is NM independent v_mfma_f32_16x16x32_f16 (own accumulators) and NV v_fma_f32
// (8 rotating registers), either in two blocks (the shape of the producers' row loop: 8 MFMAs, then the epilogue) or
// interleaved.  Cases: MFMAs alone, FMAs alone, both.  1, 2, 4 and 5 wavefronts per SIMD on every CU.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

template <int NM, int NV, bool INTERLEAVE>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
  float d[8];
  const float a = 1.0f + threadIdx.x * 1e-7f, b = 0.999f;
#pragma unroll
  for (int i = 0; i < 8; ++i) d[i] = (float)i;
  v4f acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
  v8h ha, hb;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    ha[i] = (_Float16)(1.0f + 0.001f * (threadIdx.x & 7));
    hb[i] = (_Float16)(0.5f);
  }
  for (int it = 0; it < iters; ++it) {
    if constexpr (!INTERLEAVE) {
#pragma unroll
      for (int m = 0; m < NM; ++m)
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(ha), "v"(hb));
#pragma unroll
      for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(d[v & 7]) : "v"(a), "v"(b));
    } else {
      constexpr int PER = NM > 0 ? NV / NM : NV;
#pragma unroll
      for (int m = 0; m < (NM > 0 ? NM : 1); ++m) {
        if (NM > 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(ha), "v"(hb));
#pragma unroll
        for (int v = 0; v < PER; ++v) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(d[v & 7]) : "v"(a), "v"(b));
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += d[i] + acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int NM, int NV, bool IL>
static void run(const char* name, float* out) {
  const int iters = 4000;
  for (int waves : {1, 2, 4, 5}) {   // per SIMD
    dim3 block(256), grid(256 * waves);   // 4 wavefronts per workgroup = one per SIMD; `waves` workgroups per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NM, NV, IL>), grid, block, 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NM, NV, IL>), grid, block, 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // time per loop body per SIMD (all `waves` wavefronts of a SIMD execute one body each in this time)
    printf("%-44s %d wavefronts/SIMD: %8.1f ns per body per wavefront, %8.1f ns per SIMD per (body of each of its wavefronts) / waves = %7.1f\n",
           name, waves, ms * 1e6 / iters, ms * 1e6 / iters, ms * 1e6 / iters / waves);
  }
}

int main() {
  float* out;
  hipMalloc(&out, 4096);
  run<8, 0, false>("8 MFMA", out);
  run<0, 128, false>("128 v_fma_f32", out);
  run<8, 128, false>("8 MFMA, then 128 v_fma_f32", out);
  run<8, 128, true>("8 x (MFMA, 16 v_fma_f32)", out);
  run<32, 0, false>("32 MFMA", out);
  run<32, 128, false>("32 MFMA, then 128 v_fma_f32", out);
  run<32, 128, true>("32 x (MFMA, 4 v_fma_f32)", out);
  run<0, 512, false>("512 v_fma_f32", out);
  run<32, 512, false>("32 MFMA, then 512 v_fma_f32", out);
  run<32, 512, true>("32 x (MFMA, 16 v_fma_f32)", out);
  return 0;
}
