// valu_issue_cost.hip - SIMD time per vector instruction, for the instruction mix of the fused producers (fpq_adaln.h,
// fpq_rotate_mfma.h).  Each case is a loop of 32 independent copies of ONE instruction (8 rotating destination
// registers) in inline asm; measured with 1, 2 and 4 wavefronts per SIMD on every CU.  Reports cycles per instruction
// per SIMD at the clock measured from s_memrealtime, i.e. what one more such instruction in a wavefront's row loop
// costs the vector pipe (SQ_ACTIVE_INST_VALU counts the same thing in quad-cycles).
//   hipcc -O3 --offload-arch=gfx950 -o valu_issue_cost valu_issue_cost.hip && ./valu_issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

enum { FMA_F32, FMA_MIX_F32, FMA_MIXLO_F16, PK_MUL_F16, PK_FMA_F16, PK_ADD_U16, DOT2C, MAX3, ADD_DPP, PERMLANE16_SWAP,
       CVT_FP4_F32, CVT_F16_FP4, CVT_PK_F16_F32, AND_B32, BFE_U32, PK_ADD_F32, PERM_B32, RCP_F32, CVT_FP4_F16, PK_MAX_U16, LSHRREV,
       MUL_F32, ADD_F32, FMAC_F32, CVT_F32_F16, CVT_F16_F32, CVT_F32_F16_SDWA, MAX_F32_ABS, XOR_B32, ADD_U32, LSHL_OR, BITOP3, CNDMASK,
       PK_LSHRREV_B16, PK_SUB_U16, PK_FMA_F32, DOT2_F32_F16, MOV_DPP, MAX_U32, MFMA_16, FMA_MIXLO_ONEHALF, PK_MUL_F32, MED3, SUB_F32_E64,
       N_CASES };
static const char* kNames[N_CASES] = {"v_fma_f32", "v_fma_mix_f32", "v_fma_mixlo_f16", "v_pk_mul_f16", "v_pk_fma_f16", "v_pk_add_u16",
                                      "v_dot2c_f32_f16", "v_max3_f32 |a| |b|", "v_add_f32 dpp quad_perm", "v_permlane16_swap_b32",
                                      "v_cvt_scalef32_pk_fp4_f32", "v_cvt_scalef32_pk_f16_fp4", "v_cvt_pk_f16_f32", "v_and_b32",
                                      "v_bfe_u32", "v_pk_add_f32", "v_perm_b32", "v_rcp_f32", "v_cvt_scalef32_pk_fp4_f16", "v_pk_max_u16",
                                      "v_lshrrev_b32", "v_mul_f32", "v_add_f32", "v_fmac_f32", "v_cvt_f32_f16", "v_cvt_f16_f32",
                                      "v_cvt_f32_f16 sdwa WORD_1", "v_max_f32 |a| (e64)", "v_xor_b32", "v_add_u32", "v_lshl_or_b32",
                                      "v_bitop3_b32", "v_cndmask_b32", "v_pk_lshrrev_b16", "v_pk_sub_u16", "v_pk_fma_f32", "v_dot2_f32_f16 (vop3p)",
                                      "v_mov_b32 dpp", "v_max_u32", "v_mfma_f32_16x16x32_f16", "v_fma_mixlo_f16 (f16 srcs)", "v_pk_mul_f32",
                                      "v_med3_f32", "v_sub_f32 (e64: neg mod)"};

template <int CASE>
__global__ __launch_bounds__(1024) void k(float* out, int iters, unsigned long long* clk) {
  float d[8], a = 1.0f + threadIdx.x * 1e-7f, b = 0.999f, c = 1e-9f;
#pragma unroll
  for (int i = 0; i < 8; ++i) d[i] = (float)i;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#define ONE(i)                                                                                                              \
  if (CASE == FMA_F32) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(d[i]) : "v"(a), "v"(b));                               \
  if (CASE == FMA_MIX_F32) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(d[i]) : "v"(a), "v"(b));     \
  if (CASE == FMA_MIXLO_F16) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3" : "+v"(d[i]) : "v"(a), "v"(b), "v"(c));           \
  if (CASE == PK_MUL_F16) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(d[i]) : "v"(b));                                     \
  if (CASE == PK_FMA_F16) asm volatile("v_pk_fma_f16 %0, %0, %1, 0" : "+v"(d[i]) : "v"(b));                                  \
  if (CASE == PK_ADD_U16) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(d[i]) : "v"(b));                                     \
  if (CASE == DOT2C) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(d[i]) : "v"(a), "v"(b));                               \
  if (CASE == MAX3) asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(d[i]) : "v"(a), "v"(b));                             \
  if (CASE == ADD_DPP) asm volatile("v_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(d[i]) : "v"(a)); \
  if (CASE == PERMLANE16_SWAP) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(d[i]), "+v"(d[(i + 1) & 7]));               \
  if (CASE == CVT_FP4_F32) asm volatile("v_cvt_scalef32_pk_fp4_f32 %0, %1, %2, 1.0" : "+v"(d[i]) : "v"(a), "v"(b));          \
  if (CASE == CVT_F16_FP4) asm volatile("v_cvt_scalef32_pk_f16_fp4 %0, %1, 1.0" : "=v"(d[i]) : "v"(a));                      \
  if (CASE == CVT_PK_F16_F32) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d[i]) : "v"(a), "v"(b));                     \
  if (CASE == AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(d[i]) : "v"(a));                                           \
  if (CASE == BFE_U32) asm volatile("v_bfe_u32 %0, %0, 7, 9" : "+v"(d[i]));                                                  \
  if (CASE == PERM_B32) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(d[i]) : "v"(a), "v"(b));                             \
  if (CASE == RCP_F32) asm volatile("v_rcp_f32 %0, %1" : "=v"(d[i]) : "v"(a));                                               \
  if (CASE == CVT_FP4_F16) asm volatile("v_cvt_scalef32_pk_fp4_f16 %0, %1, 1.0" : "+v"(d[i]) : "v"(a));                      \
  if (CASE == PK_MAX_U16) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(d[i]) : "v"(a));                                     \
  if (CASE == LSHRREV) asm volatile("v_lshrrev_b32 %0, 7, %0" : "+v"(d[i]));                                                 \
  if (CASE == MUL_F32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(b));                                           \
  if (CASE == ADD_F32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(d[i]) : "v"(b));                                           \
  if (CASE == FMAC_F32) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(d[i]) : "v"(a), "v"(b));                                 \
  if (CASE == CVT_F32_F16) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(d[i]) : "v"(a));                                       \
  if (CASE == CVT_F16_F32) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(d[i]) : "v"(a));                                       \
  if (CASE == CVT_F32_F16_SDWA) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(d[i]) : "v"(a)); \
  if (CASE == MAX_F32_ABS) asm volatile("v_max_f32_e64 %0, %0, |%1|" : "+v"(d[i]) : "v"(a));                                 \
  if (CASE == XOR_B32) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(d[i]) : "v"(a));                                           \
  if (CASE == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(d[i]) : "v"(a));                                           \
  if (CASE == LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(d[i]) : "v"(a));                                    \
  if (CASE == BITOP3) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x6c" : "+v"(d[i]) : "v"(a), "v"(b));                  \
  if (CASE == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(d[i]) : "v"(a));                                  \
  if (CASE == PK_LSHRREV_B16) asm volatile("v_pk_lshrrev_b16 %0, 15, %0" : "+v"(d[i]));                                      \
  if (CASE == PK_SUB_U16) asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(d[i]) : "v"(a));                                     \
  if (CASE == DOT2_F32_F16) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(d[i]) : "v"(a), "v"(b));                     \
  if (CASE == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(d[i]) : "v"(a)); \
  if (CASE == MAX_U32) asm volatile("v_max_u32 %0, %0, %1" : "+v"(d[i]) : "v"(a));                                           \
  if (CASE == FMA_MIXLO_ONEHALF) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,1,1]" : "+v"(d[i]) : "v"(a), "v"(b), "v"(c)); \
  if (CASE == MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(d[i]) : "v"(a), "v"(b));                                 \
  if (CASE == SUB_F32_E64) asm volatile("v_add_f32_e64 %0, %0, -%1" : "+v"(d[i]) : "v"(a));
    REP32(ONE)
#undef ONE
    if (CASE == MFMA_16) {
      typedef float v4f __attribute__((ext_vector_type(4)));
      typedef _Float16 h8 __attribute__((ext_vector_type(8)));
      v4f acc[8];
      h8 ha, hb;
#pragma unroll
      for (int j = 0; j < 8; ++j) { ha[j] = (_Float16)a; hb[j] = (_Float16)b; acc[j] = v4f{d[j], 0, 0, 0}; }
#pragma unroll
      for (int j = 0; j < 32; ++j) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[j & 7]) : "v"(ha), "v"(hb));
#pragma unroll
      for (int j = 0; j < 8; ++j) d[j] = acc[j][0];
    }
    if (CASE == PK_FMA_F32 || CASE == PK_MUL_F32) {
      typedef float v2f __attribute__((ext_vector_type(2)));
      v2f p[4] = {{d[0], d[1]}, {d[2], d[3]}, {d[4], d[5]}, {d[6], d[7]}}, q = {a, b};
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        if (CASE == PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[j & 3]) : "v"(q));
        else asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[j & 3]) : "v"(q));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { d[2 * j] = p[j][0]; d[2 * j + 1] = p[j][1]; }
    }
    if (CASE == PK_ADD_F32) {
      typedef float v2f __attribute__((ext_vector_type(2)));
      v2f p[4] = {{d[0], d[1]}, {d[2], d[3]}, {d[4], d[5]}, {d[6], d[7]}}, q = {a, b};
#pragma unroll
      for (int j = 0; j < 32; ++j) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[j & 3]) : "v"(q));
#pragma unroll
      for (int j = 0; j < 4; ++j) { d[2 * j] = p[j][0]; d[2 * j + 1] = p[j][1]; }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += d[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int CASE>
void run(float* out, unsigned long long* clk) {
  const int iters = 2000;
  printf("%-28s", kNames[CASE]);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int threads = 256; threads <= 1024; threads *= 2) {
    hipLaunchKernelGGL(k<CASE>, dim3(256), dim3(threads), 0, 0, out, iters, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CASE>, dim3(256), dim3(threads), 0, 0, out, iters, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2];
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / (double)h[1] * 0.1;
    // wavefront 0 of a SIMD is its oldest and wins every arbitration (its own view, h[0], is that of a wavefront alone):
    // what the SIMD sustains is the whole launch's time over the instructions of ALL its wavefronts
    const double cyc_first = (double)h[0] / (iters * 32.0);
    const int waves = threads / 256;
    const double cyc_simd = (ms * 1e6 * ghz) / (iters * 32.0 * waves);
    printf("  %dw/SIMD: %5.2f cyc/inst/SIMD (oldest wavefront: %4.1f per inst)", waves, cyc_simd, cyc_first);
  }
  printf("\n");
}

int main() {
  float* out; unsigned long long* clk;
  hipMalloc(&out, 4 * 1024 * 256); hipMalloc(&clk, 16);
  run<FMA_F32>(out, clk); run<FMA_MIX_F32>(out, clk); run<FMA_MIXLO_F16>(out, clk); run<PK_MUL_F16>(out, clk); run<PK_FMA_F16>(out, clk);
  run<PK_ADD_U16>(out, clk); run<PK_MAX_U16>(out, clk); run<DOT2C>(out, clk); run<MAX3>(out, clk); run<ADD_DPP>(out, clk); run<PERMLANE16_SWAP>(out, clk);
  run<CVT_FP4_F32>(out, clk); run<CVT_FP4_F16>(out, clk); run<CVT_F16_FP4>(out, clk); run<CVT_PK_F16_F32>(out, clk); run<AND_B32>(out, clk); run<BFE_U32>(out, clk);
  run<LSHRREV>(out, clk); run<PERM_B32>(out, clk); run<PK_ADD_F32>(out, clk); run<RCP_F32>(out, clk);
  run<MUL_F32>(out, clk); run<ADD_F32>(out, clk); run<SUB_F32_E64>(out, clk); run<FMAC_F32>(out, clk); run<CVT_F32_F16>(out, clk); run<CVT_F16_F32>(out, clk);
  run<CVT_F32_F16_SDWA>(out, clk); run<MAX_F32_ABS>(out, clk); run<MED3>(out, clk); run<XOR_B32>(out, clk); run<ADD_U32>(out, clk); run<LSHL_OR>(out, clk);
  run<BITOP3>(out, clk); run<CNDMASK>(out, clk); run<PK_LSHRREV_B16>(out, clk); run<PK_SUB_U16>(out, clk); run<PK_FMA_F32>(out, clk); run<PK_MUL_F32>(out, clk);
  run<DOT2_F32_F16>(out, clk); run<MOV_DPP>(out, clk); run<MAX_U32>(out, clk); run<FMA_MIXLO_ONEHALF>(out, clk); run<MFMA_16>(out, clk);
  return 0;
}
