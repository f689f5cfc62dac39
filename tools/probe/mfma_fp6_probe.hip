// Probe: operand layout of v_mfma_f32_16x16x128_f8f6f4 with FP6 E2M3 operands (cbsz = blgp = 2) on gfx950, unscaled.
// Hypothesis: lane l holds row (l & 15), k-block (l >> 4) of 32 consecutive k as 32 x 6 bits = 192 bits (6 VGPRs),
// element j in bits [6j, 6j+6) of the lane's little-endian bit string.  Code: bit 5 sign, bits 4:3 exponent, 2:0 mantissa
// (bias 1: e = 0 subnormal m/8, else (1 + m/8) * 2^(e-1)).  Also the issue rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void probe(const uint32_t* a, const uint32_t* b, float* d) {
  int l = threadIdx.x;
  v8i av = {0, 0, 0, 0, 0, 0, 0, 0}, bv = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 6; ++i) { av[i] = (int)a[l * 6 + i]; bv[i] = (int)b[l * 6 + i]; }
  v4f c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 2, 2, 0, 0, 0, 0);
  for (int i = 0; i < 4; ++i) d[l * 4 + i] = c[i];
}

__global__ __launch_bounds__(256) void rate(float* out, int iters) {
  v8i a = {(int)threadIdx.x, 1, 2, 3, 4, 5, 0, 0}, b = {5, (int)threadIdx.x, 7, 8, 9, 1, 0, 0};
  v4f acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = v4f{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 2, 2, 0, 0, 0, 0);
  float r = 0;
  for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

static float e2m3(uint8_t v) {
  int s = (v >> 5) & 1, e = (v >> 3) & 3, m = v & 7;
  float r = e == 0 ? (float)m / 8.0f : ldexpf(1.0f + (float)m / 8.0f, e - 1);
  return s ? -r : r;
}

int main() {
  static uint8_t A[16][128], B[128][16];
  srand(3);
  for (int r = 0; r < 16; ++r) for (int k = 0; k < 128; ++k) A[r][k] = rand() & 63;
  for (int k = 0; k < 128; ++k) for (int c = 0; c < 16; ++c) B[k][c] = rand() & 63;
  uint32_t ha[64 * 6] = {0}, hb[64 * 6] = {0};
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 32; ++j) {
      int k = 32 * (l >> 4) + j;
      uint64_t bit = 6 * j;
      for (int t = 0; t < 6; ++t) {
        if ((A[l & 15][k] >> t) & 1) ha[l * 6 + (bit + t) / 32] |= 1u << ((bit + t) % 32);
        if ((B[k][l & 15] >> t) & 1) hb[l * 6 + (bit + t) / 32] |= 1u << ((bit + t) % 32);
      }
    }
  uint32_t *da, *db; float* dd;
  (void)hipMalloc(&da, sizeof(ha)); (void)hipMalloc(&db, sizeof(hb)); (void)hipMalloc(&dd, 64 * 4 * 4);
  (void)hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd);
  float hd[256];
  (void)hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int i = 0; i < 4; ++i) {
      int row = 4 * (l >> 4) + i, col = l & 15;
      double ref = 0;
      for (int k = 0; k < 128; ++k) ref += (double)e2m3(A[row][k]) * e2m3(B[k][col]);
      if ((float)ref != hd[l * 4 + i]) { if (bad < 5) printf("mismatch l=%d i=%d got %g want %g\n", l, i, hd[l * 4 + i], ref); ++bad; }
    }
  printf("fp6 e2m3 layout: %d mismatches of 256\n", bad);
  float* out; (void)hipMalloc(&out, 4 * 256 * 256);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 2000;
  hipLaunchKernelGGL(rate, dim3(256), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(rate, dim3(256), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("fp6 16x16x128: %.2f ns per MFMA per SIMD => %.2f PFLOP/s chip\n", ms * 1e6 / (iters * 8.0), 65536.0 * iters * 8 * 1024 / (ms * 1e-3) / 1e15);
  return 0;
}
