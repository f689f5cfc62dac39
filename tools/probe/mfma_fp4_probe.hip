// Probe: operand layout of v_mfma_scale_f32_16x16x128_f8f6f4 with FP4 (E2M1) operands on gfx950.
// Hypothesis: lane l holds row (l & 15) of A (col of B), k-block (l >> 4) of 32 consecutive k,
// element j of the block in nibble j (low nibble of byte 0 first).  D: col = l & 15, row = 4*(l>>4)+reg.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void probe_unscaled(const uint32_t* a, const uint32_t* b, float* d) {
  int l = threadIdx.x;
  v8i av = {0, 0, 0, 0, 0, 0, 0, 0}, bv = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) { av[i] = (int)a[l * 4 + i]; bv[i] = (int)b[l * 4 + i]; }
  v4f c = {0, 0, 0, 0};
  // literal zero scales: the compiler selects v_mfma_f32_16x16x128_f8f6f4 (no v_mfma_ld_scale_b32 prefix)
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 4, 4, 0, 0, 0, 0);
  for (int i = 0; i < 4; ++i) d[l * 4 + i] = c[i];
}

__global__ void probe(const uint32_t* a, const uint32_t* b, float* d, int scale_a, int scale_b) {
  int l = threadIdx.x;
  v8i av = {0, 0, 0, 0, 0, 0, 0, 0}, bv = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) { av[i] = (int)a[l * 4 + i]; bv[i] = (int)b[l * 4 + i]; }
  v4f c = {0, 0, 0, 0};
  // cbsz = 4 (A fp4), blgp = 4 (B fp4), opsel 0, scales as given (E8M0 byte in bits 7:0)
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 4, 4, 0, scale_a, 0, scale_b);
  for (int i = 0; i < 4; ++i) d[l * 4 + i] = c[i];
}

static const float kVal[16] = {0, .5f, 1, 1.5f, 2, 3, 4, 6, -0.f, -.5f, -1, -1.5f, -2, -3, -4, -6};

int main() {
  uint8_t A[16][128], B[128][16];   // codes
  srand(1);
  for (int r = 0; r < 16; ++r) for (int k = 0; k < 128; ++k) A[r][k] = rand() & 15;
  for (int k = 0; k < 128; ++k) for (int c = 0; c < 16; ++c) B[k][c] = rand() & 15;
  uint32_t ha[64 * 4] = {0}, hb[64 * 4] = {0};
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 32; ++j) {
      int k = 32 * (l >> 4) + j;
      ha[l * 4 + j / 8] |= (uint32_t)A[l & 15][k] << (4 * (j % 8));
      hb[l * 4 + j / 8] |= (uint32_t)B[k][l & 15] << (4 * (j % 8));
    }
  uint32_t *da, *db; float* dd;
  hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dd, 64 * 4 * 4);
  hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
  for (int sc = 0; sc < 3; ++sc) {
    int sa = sc ? 128 : 127, sb = sc ? 126 : 127;   // 2^1 and 2^-1 in the second run; third run: unscaled instruction
    if (sc < 2) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd, sa, sb);
    else hipLaunchKernelGGL(probe_unscaled, dim3(1), dim3(64), 0, 0, da, db, dd);
    float hd[256];
    hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int i = 0; i < 4; ++i) {
        int row = 4 * (l >> 4) + i, col = l & 15;
        double ref = 0;
        for (int k = 0; k < 128; ++k) ref += (double)kVal[A[row][k]] * kVal[B[k][col]];
        if ((float)ref != hd[l * 4 + i]) { if (bad < 5) printf("mismatch l=%d i=%d got %g want %g\n", l, i, hd[l * 4 + i], ref); ++bad; }
      }
    printf("scale run %d: %d mismatches of 256\n", sc, bad);
  }
  return 0;
}
