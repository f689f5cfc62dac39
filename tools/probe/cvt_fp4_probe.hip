// cvt_fp4_probe.hip - what v_cvt_scalef32_pk_fp4_f32 / v_cvt_scalef32_pk_f16_fp4 do with every fp16 value, and whether
//     level(x) = f16_fp4( fp4_f32( float(x) + 2^-14 ) )
// reproduces the reference's nearest-entry scan on the E2M1 table (quant/quant_kernel.cu:25-37: ties go to the LARGER
// value, i.e. towards +inf on both sides; |x| beyond the table saturates to +-6; NaN / Inf give 0.0).  The bias removes
// every tie (fp16 values >= 0.25 are multiples of 2^-12, the rounding boundaries multiples of 0.25) and pushes each one
// to the side the scan picks; below 0.25 nothing is a boundary.
//   hipcc -O3 --offload-arch=gfx950 -o cvt_fp4_probe cvt_fp4_probe.hip && ./cvt_fp4_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef _Float16 h2 __attribute__((ext_vector_type(2)));

__global__ void probe(uint32_t* code_f32, uint32_t* back, uint32_t* code_f16) {
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;   // fp16 pattern
  const _Float16 h = __builtin_bit_cast(_Float16, (uint16_t)u);
  const float t = (float)h + 0x1p-14f;
  const uint32_t c = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(0u, t, t, 1.0f, 0);
  const h2 b = __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(c, 1.0f, 0);
  code_f32[u] = c;
  back[u] = __builtin_bit_cast(uint32_t, b);
  const h2 hh = {h, h};
  code_f16[u] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(0u, hh, 1.0f, 0);
}

static float h2f_host(uint16_t u) {
  const int s = u >> 15, e = (u >> 10) & 31, m = u & 1023;
  float v = e == 0 ? std::ldexp((float)m, -24) : e == 31 ? (m ? NAN : INFINITY) : std::ldexp((float)(m + 1024), e - 25);
  return s ? -v : v;
}

int main() {
  const float tab[15] = {-6, -4, -3, -2, -1.5, -1, -0.5, 0, 0.5, 1, 1.5, 2, 3, 4, 6};
  uint32_t *c32, *bk, *c16;
  hipMalloc(&c32, 65536 * 4); hipMalloc(&bk, 65536 * 4); hipMalloc(&c16, 65536 * 4);
  probe<<<256, 256>>>(c32, bk, c16);
  std::vector<uint32_t> hc(65536), hb(65536), h16(65536);
  hipMemcpy(hc.data(), c32, 65536 * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hb.data(), bk, 65536 * 4, hipMemcpyDeviceToHost);
  hipMemcpy(h16.data(), c16, 65536 * 4, hipMemcpyDeviceToHost);
  int bad = 0, bad_nonfinite = 0, neg_zero = 0, rne_differs = 0;
  for (uint32_t u = 0; u < 65536; ++u) {
    const float x = h2f_host((uint16_t)u);
    float best = 102400.0f, z = 0.0f;               // the reference's scan
    for (int i = 0; i < 15; ++i) {
      const float d = std::fabs(x - tab[i]);
      if (d <= best) { best = d; z = tab[i]; }
    }
    const float got = h2f_host((uint16_t)(hb[u] & 0xFFFFu));
    const bool same_val = (got == z);
    if (!std::isfinite(x)) {
      if (!same_val) {
        if (bad_nonfinite < 6) printf("non-finite x=%04x: code %x -> %g, scan gives %g\n", u, hc[u] & 0xFF, got, z);
        ++bad_nonfinite;
      }
      continue;
    }
    if (!same_val) {
      if (bad < 10) printf("MISMATCH x=%04x (%g): code %x -> %g, scan gives %g\n", u, x, hc[u] & 0xFF, got, z);
      ++bad;
    } else if (z == 0.0f && (hb[u] & 0x8000u)) {
      ++neg_zero;                                    // right level, but -0 where the table's zero is +0
    }
    if ((hc[u] & 0xF) != (h16[u] & 0xF)) ++rne_differs;
    if ((hc[u] & 0xF) != ((hc[u] >> 4) & 0xF)) { printf("lo/hi nibble differ at %04x\n", u); ++bad; }
  }
  printf("finite fp16 values whose biased hardware level != the scan's: %d\n", bad);
  printf("levels that came back as -0 (scan: +0): %d\n", neg_zero);
  printf("non-finite patterns not mapped to 0: %d of 2048\n", bad_nonfinite);
  printf("patterns where the unbiased fp16 -> fp4 conversion (round to nearest even) gives another code: %d\n", rne_differs);
  return bad != 0;
}
