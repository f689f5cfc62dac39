// cvt_fp6_probe.hip - what the FP6 conversion hardware of gfx950 does, for the E2M3 / E3M2 quantizers (fpq_fast16.h):
//   v_cvt_scalef32_pk32_{fp6,bf6}_f16   32 packed halves (16 registers)        -> 32 six-bit codes (6 registers)
//   v_cvt_scalef32_2xpk16_{fp6,bf6}_f32 2 x 16 floats (two 16-register tuples) -> 32 six-bit codes
//   v_cvt_scalef32_pk32_f16_{fp6,bf6}   32 codes -> 32 packed halves
// (1) every fp16 pattern through encode + decode, directly and through float(x) + 2^-17 (the bias that turns the
//     hardware's round-to-nearest-even into the reference scan's "ties to the larger value"), against the nearest level
//     computed on the host; (2) where element i of the source lands in the 192 bits; (3) SIMD time per instruction.
//   hipcc -O3 --offload-arch=gfx950 -o cvt_fp6_probe cvt_fp6_probe.hip && ./cvt_fp6_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef uint32_t u16v __attribute__((ext_vector_type(16)));
typedef uint32_t u6v __attribute__((ext_vector_type(6)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <bool BF6>
__device__ __forceinline__ u6v enc_f16(u16v s) {
  u6v d;
  if (BF6) asm volatile("v_cvt_scalef32_pk32_bf6_f16 %0, %1, 1.0" : "=v"(d) : "v"(s));
  else asm volatile("v_cvt_scalef32_pk32_fp6_f16 %0, %1, 1.0" : "=v"(d) : "v"(s));
  return d;
}
template <bool BF6>
__device__ __forceinline__ u6v enc_f32(f16v a, f16v b) {
  u6v d;
  if (BF6) asm volatile("v_cvt_scalef32_2xpk16_bf6_f32 %0, %1, %2, 1.0" : "=v"(d) : "v"(a), "v"(b));
  else asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, 1.0" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
template <bool BF6>
__device__ __forceinline__ u16v dec_f16(u6v c) {
  u16v d;
  if (BF6) asm volatile("v_cvt_scalef32_pk32_f16_bf6 %0, %1, 1.0" : "=v"(d) : "v"(c));
  else asm volatile("v_cvt_scalef32_pk32_f16_fp6 %0, %1, 1.0" : "=v"(d) : "v"(c));
  return d;
}

// thread t: patterns 32 t .. 32 t + 31; out_direct / out_bias: 32 halves each, codes: the 6 words of the direct form
template <bool BF6>
__global__ void sem(uint16_t* out_direct, uint16_t* out_bias, uint32_t* codes, uint32_t* codes_bias) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  u16v s;
  f16v a, b;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const uint32_t lo = (uint32_t)(32 * t + 2 * j), hi = lo + 1;
    s[j] = lo | (hi << 16);
    const float fl = (float)__builtin_bit_cast(_Float16, (uint16_t)lo) + 0x1p-17f;
    const float fh = (float)__builtin_bit_cast(_Float16, (uint16_t)hi) + 0x1p-17f;
    if (j < 8) { a[2 * j] = fl; a[2 * j + 1] = fh; } else { b[2 * (j - 8)] = fl; b[2 * (j - 8) + 1] = fh; }
  }
  const u6v c = enc_f16<BF6>(s);
  const u16v d = dec_f16<BF6>(c);
  const u6v cb = enc_f32<BF6>(a, b);
  const u16v db = dec_f16<BF6>(cb);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    ((uint32_t*)out_direct)[16 * t + j] = d[j];
    ((uint32_t*)out_bias)[16 * t + j] = db[j];
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    codes[6 * t + j] = c[j];
    codes_bias[6 * t + j] = cb[j];
  }
}

enum { ENC16, ENC32, DEC16 };
template <int WHICH>
__global__ __launch_bounds__(256) void cost(uint32_t* sink, int iters) {
  u16v s;
  f16v a, b;
  u6v c = {1, 2, 3, 4, 5, 6};
#pragma unroll
  for (int j = 0; j < 16; ++j) { s[j] = 0x3C003800u + threadIdx.x + j; a[j] = 1.0f + j; b[j] = 0.5f + j; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (WHICH == ENC16) { c = enc_f16<false>(s); s[0] ^= c[0]; }
      if (WHICH == ENC32) { c = enc_f32<false>(a, b); a[0] += __builtin_bit_cast(float, c[0] & 0x3F800000u); }
      if (WHICH == DEC16) { s = dec_f16<false>(c); c[0] ^= s[0]; }
    }
  }
  if (s[0] + c[0] + __builtin_bit_cast(uint32_t, a[0]) == 0x12345u) sink[threadIdx.x] = 1;
}

static float h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }

int main() {
  const int N = 65536;
  uint16_t *od, *ob;
  uint32_t *cd, *cb;
  hipMalloc(&od, N * 2); hipMalloc(&ob, N * 2); hipMalloc(&cd, N / 32 * 6 * 4); hipMalloc(&cb, N / 32 * 6 * 4);
  for (int bf6 = 0; bf6 < 2; ++bf6) {
    if (bf6) hipLaunchKernelGGL(sem<true>, dim3(N / 32 / 64), dim3(64), 0, 0, od, ob, cd, cb);
    else hipLaunchKernelGGL(sem<false>, dim3(N / 32 / 64), dim3(64), 0, 0, od, ob, cd, cb);
    hipDeviceSynchronize();
    std::vector<uint16_t> d(N), b(N);
    std::vector<uint32_t> c(N / 32 * 6), c2(N / 32 * 6);
    hipMemcpy(d.data(), od, N * 2, hipMemcpyDeviceToHost); hipMemcpy(b.data(), ob, N * 2, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), cd, c.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(c2.data(), cb, c2.size() * 4, hipMemcpyDeviceToHost);
    // levels of E2M3 (bias 1) / E3M2 (bias 3)
    std::vector<float> lv;
    const int M = bf6 ? 2 : 3, E = bf6 ? 3 : 2, bias = bf6 ? 3 : 1;
    for (int e = 0; e < (1 << E); ++e)
      for (int m = 0; m < (1 << M); ++m)
        lv.push_back(e == 0 ? std::ldexp((float)m / (1 << M), 1 - bias) : std::ldexp(1.0f + (float)m / (1 << M), e - bias));
    int mis_direct_tie = 0, mis_direct_other = 0, mis_bias = 0, nonfinite_seen = 0, shown = 0;
    for (int p = 0; p < N; ++p) {
      const float x = h2f((uint16_t)p);
      if (!(std::fabs(x) <= 65504.0f)) {   // NaN / Inf: report what comes out
        if (nonfinite_seen++ < 4 || p == 0x7C00 || p == 0xFC00 || p == 0x7E00)
          printf("  %s non-finite input 0x%04x -> direct 0x%04x (%g), biased 0x%04x (%g)\n", bf6 ? "E3M2" : "E2M3", p, d[p], h2f(d[p]), b[p], h2f(b[p]));
        continue;
      }
      // nearest level, ties to the LARGER value (the reference scan), and whether x is an exact tie
      float best = 0, bd = 1e30f;
      bool tie = false;
      for (int sgn = 0; sgn < 2; ++sgn)
        for (float l : lv) {
          const float v = sgn ? -l : l, dist = std::fabs(x - v);
          if (dist < bd || (dist == bd && v > best)) { tie = (dist == bd && v != best); bd = dist; best = v; }
          else if (dist == bd && v != best) tie = true;
        }
      const float gd = h2f(d[p]), gb = h2f(b[p]);
      if (gd != best) { if (tie) ++mis_direct_tie; else { ++mis_direct_other; if (shown++ < 6) printf("  direct: x=%g (0x%04x) -> %g, nearest %g\n", x, p, gd, best); } }
      if (gb != best || (gb == 0 && std::signbit(gb) && !std::signbit(best))) { ++mis_bias; if (shown++ < 12) printf("  biased: x=%g (0x%04x) -> %g (0x%04x), nearest %g\n", x, p, gb, b[p], best); }
    }
    printf("%s: direct f16 path: %d mismatches at exact ties, %d elsewhere; float(x) + 2^-17 path: %d mismatches (of 63488 finite patterns)\n",
           bf6 ? "E3M2 (bf6)" : "E2M3 (fp6)", mis_direct_tie, mis_direct_other, mis_bias);
    // layout: thread 480 holds patterns 0x3C00 .. 0x3C1F (1.0 ...): print its code words; thread 0: zeros and denormals
    const int t1 = 0x3C00 / 32;
    printf("  codes of patterns 0x3C00.. (all 1.0 -> code %s): %08x %08x %08x %08x %08x %08x\n", bf6 ? "001100" : "001000",
           c[6 * t1], c[6 * t1 + 1], c[6 * t1 + 2], c[6 * t1 + 3], c[6 * t1 + 4], c[6 * t1 + 5]);
    const int t2 = 0x4000 / 32;   // 2.0, 2.002, ... : element 0 = 2.0 exactly; patterns +16.. cross 2.0156?  all round to 2.0
    printf("  codes of patterns 0x4000.. : %08x %08x ...   same through the f32 form: %08x %08x\n", c[6 * t2], c[6 * t2 + 1], c2[6 * t2], c2[6 * t2 + 1]);
    // one element differs: patterns 0x4500 + i (5.0 ..): thread index
    const int t3 = 0xBC00 / 32;
    printf("  codes of patterns 0xBC00.. (all -1.0): %08x %08x %08x\n", c[6 * t3], c[6 * t3 + 1], c[6 * t3 + 2]);
  }
  // cost
  uint32_t* sink;
  hipMalloc(&sink, 4096);
  const char* names[3] = {"v_cvt_scalef32_pk32_fp6_f16", "v_cvt_scalef32_2xpk16_fp6_f32", "v_cvt_scalef32_pk32_f16_fp6"};
  for (int w = 0; w < 3; ++w)
    for (int waves : {1, 2, 4}) {
      const int iters = 2000;
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      auto launch = [&](int n) {
        if (w == 0) hipLaunchKernelGGL(cost<ENC16>, dim3(256 * waves), dim3(256), 0, 0, sink, n);
        if (w == 1) hipLaunchKernelGGL(cost<ENC32>, dim3(256 * waves), dim3(256), 0, 0, sink, n);
        if (w == 2) hipLaunchKernelGGL(cost<DEC16>, dim3(256 * waves), dim3(256), 0, 0, sink, n);
      };
      launch(50);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      launch(iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("%-34s %d wavefronts/SIMD: %7.1f ns per instruction per SIMD (32 elements each)\n", names[w], waves, ms * 1e6 / (iters * 8.0) / waves);
    }
  return 0;
}
