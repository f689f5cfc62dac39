// gelu_probe.hip - candidate formulas for F.gelu(x, approximate="tanh") on an fp16 tensor (torch computes it in fp32:
// aten/src/ATen/native/cuda/ActivationGeluKernel.cu) evaluated on ALL 65536 fp16 patterns; tools/gelu_probe_check.py
// compares each with torch's own result on the same GPU (bit-equal inputs, ulp histogram).  The fc1 tail of
// fpq_gemm_fp4.h uses the cheapest variant that stays within one fp16 ulp of torch everywhere.
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -o gelu_probe gelu_probe.hip && ./gelu_probe out.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define NV 8
__device__ __forceinline__ float tanh_devlib(float x) {   // __ocml_tanh_f32 restated
  const float y = __builtin_fabsf(x);
  float z;
  if (y < 0.625f) {
    const float y2 = x * x;
    float p = __builtin_fmaf(y2, -0x1.758e7ap-8f, 0x1.521192p-6f);
    p = __builtin_fmaf(y2, p, -0x1.b8389cp-5f);
    p = __builtin_fmaf(y2, p, 0x1.110704p-3f);
    p = __builtin_fmaf(y2, p, -0x1.555532p-2f);
    z = __builtin_fmaf(y2, y * p, y);
  } else {
    const float t = __builtin_expf(2.0f * y);
    z = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(t + 1.0f), 1.0f);
  }
  return __builtin_copysignf(z, x);
}
__device__ float variant(int v, float x) {
  const float kBeta = (float)(1.41421356237309504880 * 1.12837916709551257390 * 0.5), kKappa = 0.044715f;
  const float L2E2 = 2.8853900817779268f;   // 2 log2(e)
  switch (v) {
    case 0: {   // torch's order, x + kappa x^3 contracted into an fma, the device library's tanh
      const float x3 = x * x * x;
      return 0.5f * x * (1.0f + tanh_devlib(kBeta * __builtin_fmaf(kKappa, x3, x)));
    }
    case 1: {   // ... not contracted
      const float x3 = x * x * x;
      return 0.5f * x * (1.0f + tanh_devlib(kBeta * (x + kKappa * x3)));
    }
    case 2: {   // ... the library call itself
      const float x3 = x * x * x;
      return 0.5f * x * (1.0f + tanhf(kBeta * __builtin_fmaf(kKappa, x3, x)));
    }
    case 3: {   // torch's u; tanh by its exp branch alone with a bare v_exp_f32, same 1 + copysign structure
      const float x3 = x * x * x;
      const float u = kBeta * __builtin_fmaf(kKappa, x3, x);
      const float t = __builtin_amdgcn_exp2f(__builtin_fabsf(u) * L2E2);
      const float z = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(t + 1.0f), 1.0f);
      return 0.5f * x * (1.0f + __builtin_copysignf(z, u));
    }
    case 4: {   // sigmoid form: x * w, w = 1 / (1 + exp2(-2 log2e u)), snapped to the grid of torch's 1 + tanh (w - 0.5 + 0.5)
      const float x3 = x * x * x;
      const float u = kBeta * __builtin_fmaf(kKappa, x3, x);
      const float e = __builtin_amdgcn_exp2f(-L2E2 * u);
      const float w = __builtin_amdgcn_rcpf(1.0f + e);
      const float ws = (w - 0.5f) + 0.5f;
      return x * ws;
    }
    case 5: {   // sigmoid form without the snap
      const float x3 = x * x * x;
      const float u = kBeta * __builtin_fmaf(kKappa, x3, x);
      const float e = __builtin_amdgcn_exp2f(-L2E2 * u);
      return x * __builtin_amdgcn_rcpf(1.0f + e);
    }
    case 6: {   // three-operation argument: m = x (c0 + c1 x^2), sigmoid form with the snap
      const float c0 = -L2E2 * kBeta, c1 = c0 * kKappa;
      const float m = x * __builtin_fmaf(x * x, c1, c0);
      const float w = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(m));
      return x * ((w - 0.5f) + 0.5f);
    }
    default: {  // variant 3 with the argument of exp2 split hi / lo (what llvm.exp.f32's lowering buys)
      const float x3 = x * x * x;
      const float u = kBeta * __builtin_fmaf(kKappa, x3, x);
      const float au = __builtin_fabsf(u);
      const float ph = au * L2E2;
      const float pl = __builtin_fmaf(au, L2E2, -ph) + au * (float)(2.8853900817779268 - (double)L2E2);
      const float e0 = __builtin_amdgcn_exp2f(ph);
      const float t = __builtin_fmaf(e0, pl * 0.6931471805599453f, e0);
      const float z = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(t + 1.0f), 1.0f);
      return 0.5f * x * (1.0f + __builtin_copysignf(z, u));
    }
  }
}
__global__ void probe(uint16_t* out) {
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;   // fp16 pattern
  const float x = (float)__builtin_bit_cast(_Float16, (uint16_t)u);
  for (int v = 0; v < NV; ++v) {
    float g = variant(v, x);
    asm volatile("" : "+v"(g));   // the fp32 value exists before it is narrowed (no fused narrowing)
    out[v * 65536 + u] = __builtin_bit_cast(uint16_t, (_Float16)g);
  }
}
int main(int argc, char** argv) {
  uint16_t* d;
  hipMalloc(&d, NV * 65536 * 2);
  probe<<<256, 256>>>(d);
  std::vector<uint16_t> h(NV * 65536);
  hipMemcpy(h.data(), d, h.size() * 2, hipMemcpyDeviceToHost);
  FILE* f = fopen(argc > 1 ? argv[1] : "gelu_probe.bin", "wb");
  fwrite(h.data(), 2, h.size(), f);
  fclose(f);
  printf("wrote %d variants x 65536 fp16 results\n", NV);
  return 0;
}
