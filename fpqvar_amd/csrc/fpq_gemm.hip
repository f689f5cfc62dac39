// fpq_gemm.hip - the consumers on the far side of the quantizers and of the KV path:
// FP4 / FP6 / FP8 matrix-core GEMMs on quantizer codes, attention over the cache, the gated residual tail.
// Second translation unit of libfpq_hip.so; the quantizers (and the code-emitting kernels that live in the
// fpq_gemm_*.h headers) are compiled in fpq_kernels.hip.
#include "fpq_common.h"

namespace {
#include "fpq_fast16.h"
#include "fpq_gemm_fp4.h"
#include "fpq_gemm_fp8.h"
#include "fpq_gemm_fp6.h"
#include "fpq_attention.h"

// Per-group scales [rows, groups] (fp16 or fp32) -> the fp32 k-major scale image [groups][image_rows] of the FP4 GEMM (include/fpq.h):
// image_rows = rows rounded up to 4 (activation side) or to 64 (weight side; natural row order), padding = 0.
template <typename Ts>
__global__ __launch_bounds__(256) void scales_to_kmajor_kernel(const Ts* __restrict__ scales, float* __restrict__ image, int64_t rows,
                                                               int64_t image_rows, int groups) {
  const int64_t n = (int64_t)groups * image_rows;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t g = i / image_rows, r = i - g * image_rows;
    image[i] = r < rows ? (float)scales[r * groups + g] : 0.0f;
  }
}
}  // namespace

extern "C" {

int fpq_attention_blhc(const void* q, const void* k, const void* v, void* out, int64_t batch, int64_t lq, int64_t lkv,
                       int64_t heads, int64_t head_dim, int64_t q_batch_pitch, int64_t q_token_pitch,
                       int64_t kv_batch_pitch, int64_t kv_token_pitch, float scale, fpq_stream_t stream) {
  if (batch < 0 || lq < 0 || lkv < 0 || heads <= 0) return FPQ_ERR_ARG;
  if (head_dim != 64) return FPQ_ERR_SHAPE;
  if (batch == 0 || lq == 0) return FPQ_OK;
  if (lkv == 0 || !(scale > 0.0f)) return FPQ_ERR_ARG;            // softmax over nothing
  if (!q || !k || !v || !out) return FPQ_ERR_ARG;
  if ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) != 0) return FPQ_ERR_ARG;
  if (q_batch_pitch % 8 != 0 || q_token_pitch % 8 != 0 || kv_batch_pitch % 8 != 0 || kv_token_pitch % 8 != 0)
    return FPQ_ERR_SHAPE;
  if (lq > 0x7FFFFFFF || lkv > 0x7FFFFFFF || batch * heads > 0x7FFFFFFF) return FPQ_ERR_SHAPE;
  AttnArgs a;
  a.q = (const uint16_t*)q;
  a.k = (const uint16_t*)k;
  a.v = (const uint16_t*)v;
  a.out = (uint16_t*)out;
  a.q_batch = q_batch_pitch;
  a.q_token = q_token_pitch;
  a.kv_batch = kv_batch_pitch;
  a.kv_token = kv_token_pitch;
  a.batch = (int)batch;
  a.heads = (int)heads;
  a.lq = (int)lq;
  a.lkv = (int)lkv;
  a.q_tiles = (int)((lq + 127) / 128);
  a.scale_log2e = scale * 1.4426950408889634f;
  const int64_t groups = (batch * heads + 7) / 8;
  const int64_t n_wg = groups * a.q_tiles * 8;
  if (n_wg > 0x7FFFFFFF) return FPQ_ERR_SHAPE;
  hipLaunchKernelGGL(attn_fwd64_kernel, dim3((unsigned)n_wg), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch();
}

int fpq_gemm_fp4_mx(const uint8_t* a_codes, const void* a_scales, const uint8_t* w_codes, const void* w_scales,
                    int w_scale_dtype, const void* bias, void* out, int64_t tokens, int64_t outs, int64_t k,
                    fpq_stream_t stream) {
  return fpq_gemm_fp4_mx_ex(a_codes, a_scales, w_codes, w_scales, w_scale_dtype, bias, out, tokens, outs, k, nullptr, stream);
}

int fpq_gemm_fp6_rows(const uint8_t* a_codes, const void* a_scales, int a_scale_dtype, const uint8_t* w_codes,
                      const void* w_scales, int w_scale_dtype, const void* bias, void* out, int64_t tokens, int64_t outs,
                      int64_t k, fpq_stream_t stream) {
  return fpq_gemm_fp6_rows_ex(a_codes, a_scales, a_scale_dtype, w_codes, w_scales, w_scale_dtype, bias, out, tokens, outs, k,
                              nullptr, stream);
}

int fpq_gemm_fp8_rows(const uint8_t* a_codes, const void* a_scales, int a_scale_dtype, const uint8_t* w_codes,
                      const void* w_scales, int w_scale_dtype, const void* bias, void* out, int64_t tokens, int64_t outs,
                      int64_t k, fpq_stream_t stream) {
  return fpq_gemm_fp8_rows_ex(a_codes, a_scales, a_scale_dtype, w_codes, w_scales, w_scale_dtype, bias, out, tokens, outs, k,
                              nullptr, stream);
}

// out = resid + y * gate[row / rows_per_gate, :], fp16 with one rounding per operation (the GEMM epilogues' tail as a
// kernel of its own, for Linears that run elsewhere - e.g. fc2's fp16 GEMM)
__global__ __launch_bounds__(kBlock) void gate_residual_kernel(const u32x4* y, const u32x4* __restrict__ gate,
                                                              const u32x4* resid, u32x4* out, int64_t n_vec, int row_vec,
                                                              int rows_per_gate) {
  for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < n_vec; v += (int64_t)gridDim.x * kBlock) {
    const int64_t row = v / row_vec;
    const int c = (int)(v - row * row_vec);
    u32x4 a = __builtin_nontemporal_load(y + v);
    const u32x4 g = gate[(row / rows_per_gate) * row_vec + c];
    const u32x4 r = resid[v];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const fpq_h2_t p = __builtin_bit_cast(fpq_h2_t, (uint32_t)a[i]) * __builtin_bit_cast(fpq_h2_t, (uint32_t)g[i]);
      a[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(fpq_h2_t, (uint32_t)r[i]) + p);
    }
    out[v] = a;
  }
}

int fpq_gate_residual(const void* y, const void* gate, const void* residual, void* out, int64_t rows, int64_t cols,
                      int64_t rows_per_gate, fpq_stream_t stream) {
  if (rows < 0 || cols < 0 || rows_per_gate < 1 || rows_per_gate > 0x7FFFFFFF) return FPQ_ERR_ARG;
  if (cols % 8 != 0 || cols / 8 > 0x7FFFFFFF) return FPQ_ERR_SHAPE;
  if (rows == 0 || cols == 0) return FPQ_OK;
  if (!y || !gate || !residual || !out) return FPQ_ERR_ARG;
  if ((((uintptr_t)y | (uintptr_t)gate | (uintptr_t)residual | (uintptr_t)out) & 15) != 0) return FPQ_ERR_ARG;
  const int64_t n_vec = rows * (cols / 8);
  const int64_t wgs = (n_vec + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(gate_residual_kernel, dim3(grid_for(wgs, 1 << 20)), dim3(kBlock), 0, (hipStream_t)stream,
                     (const u32x4*)y, (const u32x4*)gate, (const u32x4*)residual, (u32x4*)out, n_vec, (int)(cols / 8),
                     (int)rows_per_gate);
  return check_launch();
}

// validates an optional epilogue descriptor and turns it into the kernels' form
static int gemm_epilogue(const fpq_gemm_epilogue_t* ep, int64_t tokens, GemmEpi* epi) {
  epi->gate = nullptr;
  epi->resid = nullptr;
  epi->rows_per_gate = 1;
  epi->km_w_rows = 0;
  epi->sp_cols = 0;
  epi->sp_rpb = 1;
  for (int p = 0; p < 3; ++p) {
    epi->sp_out[p] = nullptr;
    epi->sp_stride[p] = epi->sp_bstride[p] = epi->sp_row0[p] = 0;
  }
  if (!ep) return FPQ_OK;
  if (ep->gate && (ep->rows_per_gate < 1 || ep->rows_per_gate > 0x7FFFFFFF)) return FPQ_ERR_ARG;
  if ((((uintptr_t)ep->gate | (uintptr_t)ep->residual) & 15) != 0) return FPQ_ERR_ARG;
  epi->gate = (const _Float16*)ep->gate;
  epi->resid = (const _Float16*)ep->residual;
  if (ep->gate) epi->rows_per_gate = (int)ep->rows_per_gate;
  (void)tokens;
  return FPQ_OK;
}

// km: both operands are k-major images (include/fpq.h); only the LDS-DMA kernels read them
static int gemm_fp4_mx_impl(const uint8_t* a_codes, const void* a_scales, const uint8_t* w_codes, const void* w_scales,
                            int w_scale_dtype, const void* bias, void* out, int64_t tokens, int64_t outs, int64_t k,
                            const fpq_gemm_epilogue_t* epilogue, bool km, fpq_stream_t stream, const fpq_gemm_split_t* split = nullptr) {
  if (tokens < 0 || outs < 0 || k < 0) return FPQ_ERR_ARG;
  GemmEpi epi;
  if (int rc = gemm_epilogue(epilogue, tokens, &epi)) return rc;
  if (split) {   // the outputs leave in column parts, each to its own rows (include/fpq.h): the LDS-DMA kernels' plain epilogue only
    if (epilogue || split->n_parts < 1 || split->n_parts > 3 || split->part_cols <= 0 || split->part_cols % 128 != 0 ||
        outs != split->n_parts * split->part_cols || split->rows_per_batch < 1 || split->rows_per_batch > 0x7FFFFFFF)
      return FPQ_ERR_ARG;
    epi.sp_cols = (int)split->part_cols;
    epi.sp_rpb = (int)split->rows_per_batch;
    for (int p = 0; p < split->n_parts; ++p) {
      if (!split->out[p] || ((uintptr_t)split->out[p] & 7) != 0 || split->row_stride[p] < split->part_cols || split->row_stride[p] % 4 != 0 ||
          split->batch_stride[p] < 0 || split->row0[p] < 0)
        return FPQ_ERR_ARG;
      epi.sp_out[p] = (_Float16*)split->out[p];
      epi.sp_stride[p] = split->row_stride[p];
      epi.sp_bstride[p] = split->batch_stride[p];
      epi.sp_row0[p] = split->row0[p];
    }
    if (!out) out = split->out[0];   // (never written: every tile belongs to a part)
  }
  if (km) {   // scales come as fp32 k-major images too (include/fpq.h); their lane offsets are 32-bit: 3 planes of rows * 4 bytes
    if (w_scale_dtype != FPQ_F32) return FPQ_ERR_DTYPE;
    if (tokens >= (1ll << 28) || outs >= (1ll << 28)) return FPQ_ERR_SHAPE;
    if ((((uintptr_t)a_scales | (uintptr_t)w_scales) & 15) != 0) return FPQ_ERR_ARG;
    epi.km_w_rows = (int)((outs + 63) / 64 * 64);
  }
  if (w_scale_dtype != FPQ_F16 && w_scale_dtype != FPQ_F32) return FPQ_ERR_DTYPE;
  if (k % 128 != 0 || k > 128 * 64 || outs % 8 != 0 || tokens > 0x7FFFFFFF || outs > 0x7FFFFFFF) return FPQ_ERR_SHAPE;
  if (tokens == 0 || outs == 0) return FPQ_OK;
  if (k == 0 || !a_codes || !a_scales || !w_codes || !w_scales || !out) return FPQ_ERR_ARG;
  if ((((uintptr_t)a_codes | (uintptr_t)w_codes | (uintptr_t)out) & 15) != 0) return FPQ_ERR_ARG;
  const int G = (int)(k / 128);
  hipStream_t st = (hipStream_t)stream;
  // Default: the LDS-DMA kernel with 128 x 128 tiles (three workgroups per CU); 256 x 128 tiles (two per CU) from 4000 of them
  // on (round 4: 2 - 5 % faster at [65536 x 1920] x {1920, 5760, 7680} and from 16 900 tokens on for the wide Linears, 3 - 5 %
  // slower between 1000 and 4000 tiles) while two of them fit a CU's 160 KB of LDS (the scale tiles grow with K: from K = 3840 on
  // only one would, and the smaller tile is 15 % faster there - tools/gemm_k_sweep.py); 64 x 128 tiles while the 128 x 128 ones
  // would fill less than half of the chip's 768 slots (the first scale steps of a generation: 8.4 against 11.4 us at 100 tokens;
  // tools/gemm_small_steps.py); the register-staged kernel when the LDS image does not fit (very long K).
  // FPQ_GEMM_CFG (experiments, tests): 0..2 register-staged tilings, 10 / 20 / 30 LDS-DMA tilings (256x128, 128x128, 64x128).
    const int64_t big_tiles = ((tokens + 255) / 256) * ((outs + 127) / 128);
  const int64_t mid_tiles = ((tokens + 127) / 128) * ((outs + 127) / 128);
  const bool big_fits_twice = 2 * GemmGldsCfg<8, 4>::lds(G) <= 160 * 1024;
  // (the LDS-DMA kernel reads the bias four outputs at a time: a bias that is not 8-byte aligned goes to the other kernel)
  int cfg = ((uintptr_t)bias & 7) != 0 ? 0 : fpq_opt_set(OPT_FPQ_GEMM_CFG) ? fpq_opt(OPT_FPQ_GEMM_CFG, 0) : mid_tiles <= 384 ? 30 : (big_tiles >= 4000 && big_fits_twice) ? 10 : 20;
  if (km || split) {
    if (((uintptr_t)bias & 7) != 0 || outs + 63 > 0x7FFFFFFF) return FPQ_ERR_ARG;
    if (cfg != 10 && cfg != 20 && cfg != 30) cfg = mid_tiles <= 384 ? 30 : (big_tiles >= 4000 && big_fits_twice) ? 10 : 20;
  }
#define FPQ_GEMM_LAUNCH(MT, NT, WR, WC)                                                                              \
  do {                                                                                                               \
    using Cfg = GemmCfg<MT, NT, WR, WC>;                                                                             \
    const int64_t n_col = (outs + Cfg::BN - 1) / Cfg::BN, n_row = (tokens + Cfg::BM - 1) / Cfg::BM;                  \
    const int64_t n_wg = 8 * ((n_col + 7) / 8) * n_row;                                                              \
    if (n_wg > 0x7FFFFFFF) return FPQ_ERR_SHAPE;                                                                     \
    if (w_scale_dtype == FPQ_F16)                                                                                    \
      hipLaunchKernelGGL((gemm_fp4_kernel<_Float16, MT, NT, WR, WC>), dim3((unsigned)n_wg), dim3(Cfg::NTHR),        \
                         Cfg::lds(G), st, a_codes, (const _Float16*)a_scales, w_codes, (const _Float16*)w_scales,   \
                         (const _Float16*)bias, (_Float16*)out, (int)tokens, (int)outs, (int)k, epi);                     \
    else                                                                                                             \
      hipLaunchKernelGGL((gemm_fp4_kernel<float, MT, NT, WR, WC>), dim3((unsigned)n_wg), dim3(Cfg::NTHR),           \
                         Cfg::lds(G), st, a_codes, (const _Float16*)a_scales, w_codes, (const float*)w_scales,      \
                         (const _Float16*)bias, (_Float16*)out, (int)tokens, (int)outs, (int)k, epi);                     \
  } while (0)
#define FPQ_GEMM_GLDS(MT, NT)                                                                                        \
  do {                                                                                                               \
    using Cfg = GemmGldsCfg<MT, NT>;                                                                                 \
    const size_t lds = Cfg::lds(G);                                                                                  \
    if (lds <= 160 * 1024) {                                                                                         \
      const int64_t n_col = (outs + Cfg::BN - 1) / Cfg::BN, n_row = (tokens + Cfg::BM - 1) / Cfg::BM;                \
      const int64_t n_wg = 8 * ((n_col + 7) / 8) * n_row;                                                            \
      if (n_wg > 0x7FFFFFFF) return FPQ_ERR_SHAPE;                                                                   \
      if (w_scale_dtype == FPQ_F16)                                                                                  \
        hipLaunchKernelGGL((gemm_fp4_glds_kernel<_Float16, MT, NT>), dim3((unsigned)n_wg), dim3(256), lds, st,      \
                           a_codes, (const _Float16*)a_scales, w_codes, (const _Float16*)w_scales,                   \
                           (const _Float16*)bias, (_Float16*)out, (int)tokens, (int)outs, (int)k, epi, GemmNoFc1{});      \
      else                                                                                                           \
        hipLaunchKernelGGL((gemm_fp4_glds_kernel<float, MT, NT>), dim3((unsigned)n_wg), dim3(256), lds, st,         \
                           a_codes, (const _Float16*)a_scales, w_codes, (const float*)w_scales,                      \
                           (const _Float16*)bias, (_Float16*)out, (int)tokens, (int)outs, (int)k, epi, GemmNoFc1{});      \
      return check_launch();                                                                                         \
    }                                                                                                                \
  } while (0)
  if (cfg == 30) FPQ_GEMM_GLDS(2, 4);   // 64 x 128 tiles
  if (cfg == 10) FPQ_GEMM_GLDS(8, 4);
  if (cfg == 10 || cfg == 20 || cfg == 30) FPQ_GEMM_GLDS(4, 4);   // (the larger tile's LDS image may not fit where the smaller one's does)
#undef FPQ_GEMM_GLDS
  if (km || split) return FPQ_ERR_SHAPE;   // K too long for the LDS-DMA kernel's scale tiles: the register-staged kernels read row-major codes, write one tensor
  if (cfg == 1) FPQ_GEMM_LAUNCH(2, 4, 4, 2);
  else if (cfg == 2) FPQ_GEMM_LAUNCH(4, 4, 2, 4);
  else FPQ_GEMM_LAUNCH(4, 4, 2, 2);
#undef FPQ_GEMM_LAUNCH
  return check_launch();
}

int fpq_gemm_fp4_mx_ex(const uint8_t* a_codes, const void* a_scales, const uint8_t* w_codes, const void* w_scales,
                       int w_scale_dtype, const void* bias, void* out, int64_t tokens, int64_t outs, int64_t k,
                       const fpq_gemm_epilogue_t* epilogue, fpq_stream_t stream) {
  return gemm_fp4_mx_impl(a_codes, a_scales, w_codes, w_scales, w_scale_dtype, bias, out, tokens, outs, k, epilogue, false, stream);
}
int fpq_gemm_fp4_mx_split(const uint8_t* a_codes, const void* a_scales, const uint8_t* w_codes, const void* w_scales, int w_scale_dtype,
                          const void* bias, int64_t tokens, int64_t outs, int64_t k, const fpq_gemm_split_t* split, int kmajor,
                          fpq_stream_t stream) {
  if (!split) return FPQ_ERR_ARG;
  return gemm_fp4_mx_impl(a_codes, a_scales, w_codes, w_scales, w_scale_dtype, bias, nullptr, tokens, outs, k, nullptr, kmajor != 0, stream, split);
}
int fpq_gemm_fp4_mx_km(const uint8_t* a_image, const void* a_scales, const uint8_t* w_image, const void* w_scales,
                       int w_scale_dtype, const void* bias, void* out, int64_t tokens, int64_t outs, int64_t k,
                       const fpq_gemm_epilogue_t* epilogue, fpq_stream_t stream) {
  return gemm_fp4_mx_impl(a_image, a_scales, w_image, w_scales, w_scale_dtype, bias, out, tokens, outs, k, epilogue, true, stream);
}

#ifdef FPQ_GEMM6_STAMPS
// diagnostic builds only: where the FP6 GEMM's wavefronts put their per-phase stamp sums ([wavefronts][8] uint64, zeroed by the caller)
int fpq_debug_gemm6_stamp_buffer(void* device_buffer) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_gemm6_stamps), &device_buffer, sizeof(void*)) == hipSuccess ? FPQ_OK : FPQ_ERR_LAUNCH;
}
#endif

// fc1 with GELU and fc2's dual-format input quantizer in the GEMM's epilogue (fpq_gemm_fp4.h, GemmFc1)
static int gemm_fp4_gelu_dual_impl(const uint8_t* a_codes, const void* a_scales, const uint8_t* w_codes, const void* w_scales,
                                   int w_scale_dtype, const void* bias, void* out, void* gelu_out, int64_t tokens, int64_t outs,
                                   int64_t k, void* nan_flag, bool km, fpq_stream_t stream) {
  if (tokens < 0 || outs < 0 || k < 0) return FPQ_ERR_ARG;
  if (w_scale_dtype != FPQ_F16 && w_scale_dtype != FPQ_F32) return FPQ_ERR_DTYPE;
  // outs % 128: an output tile is one quantization group wide
  if (k % 128 != 0 || k > 128 * 64 || outs % 128 != 0 || tokens > 0x7FFFFFFF || outs > 0x7FFFFFFF) return FPQ_ERR_SHAPE;
  if (tokens == 0 || outs == 0) return FPQ_OK;
  if (k == 0 || !a_codes || !a_scales || !w_codes || !w_scales || !out) return FPQ_ERR_ARG;
  if ((((uintptr_t)a_codes | (uintptr_t)w_codes | (uintptr_t)out | (uintptr_t)gelu_out) & 15) != 0 || ((uintptr_t)bias & 7) != 0 ||
      ((uintptr_t)nan_flag & 7) != 0)
    return FPQ_ERR_ARG;
  // the dual quantizer's table and arguments, built once by the quantizers' translation unit (immutable afterwards)
  struct Dual { GemmFc1 xe; int rc; };
  static const Dual dual = [] {
    Dual d;
    d.rc = fpq_internal_dual_lut(FPQ_E1M2_NEG, FPQ_E2M1_POS, &d.xe.a, sizeof(d.xe.a), &d.xe.tab, sizeof(d.xe.tab));
    d.xe.h_out = nullptr;
    d.xe.nan_flag = nullptr;
    return d;
  }();
  if (dual.rc != FPQ_OK) return dual.rc;
  GemmFc1 xe = dual.xe;
  xe.h_out = (_Float16*)gelu_out;
  xe.nan_flag = (uint32_t*)nan_flag;
  const int G = (int)(k / 128);
  hipStream_t st = (hipStream_t)stream;
  GemmEpi epi{nullptr, nullptr, 1, km ? (int)outs : 0};   // (outs % 128 == 0: the weight image has exactly outs rows)
  if (km && (w_scale_dtype != FPQ_F32 || tokens >= (1ll << 28) || (((uintptr_t)a_scales | (uintptr_t)w_scales) & 15) != 0)) return FPQ_ERR_ARG;
  // tile choice as fpq_gemm_fp4_mx_ex (FPQ_GEMM_CFG 10 / 20 / 30 forces one of the three LDS-DMA tilings)
  const int64_t big_tiles = ((tokens + 255) / 256) * (outs / 128), mid_tiles = ((tokens + 127) / 128) * (outs / 128);
  const bool big_fits_twice = 2 * GemmGldsCfg<8, 4>::lds_fc1(G, xe.a.shift) <= 160 * 1024;
  const int want = fpq_opt(OPT_FPQ_GEMM_CFG, 0);
  const int cfg = (want == 10 || want == 20 || want == 30) ? want : mid_tiles <= 384 ? 30 : (big_tiles >= 4000 && big_fits_twice) ? 10 : 20;
#define FPQ_GEMM_FC1(MT, NT)                                                                                         \
  do {                                                                                                               \
    using Cfg = GemmGldsCfg<MT, NT>;                                                                                 \
    const size_t lds = Cfg::lds_fc1(G, xe.a.shift);                                                                  \
    if (lds <= 160 * 1024) {                                                                                         \
      const int64_t n_col = outs / Cfg::BN, n_row = (tokens + Cfg::BM - 1) / Cfg::BM;                                \
      const int64_t n_wg = 8 * ((n_col + 7) / 8) * n_row;                                                            \
      if (n_wg > 0x7FFFFFFF) return FPQ_ERR_SHAPE;                                                                   \
      if (w_scale_dtype == FPQ_F16)                                                                                  \
        hipLaunchKernelGGL((gemm_fp4_glds_kernel<_Float16, MT, NT, GemmFc1>), dim3((unsigned)n_wg), dim3(256), lds, st, \
                           a_codes, (const _Float16*)a_scales, w_codes, (const _Float16*)w_scales,                   \
                           (const _Float16*)bias, (_Float16*)out, (int)tokens, (int)outs, (int)k, epi, xe);          \
      else                                                                                                           \
        hipLaunchKernelGGL((gemm_fp4_glds_kernel<float, MT, NT, GemmFc1>), dim3((unsigned)n_wg), dim3(256), lds, st, \
                           a_codes, (const _Float16*)a_scales, w_codes, (const float*)w_scales,                      \
                           (const _Float16*)bias, (_Float16*)out, (int)tokens, (int)outs, (int)k, epi, xe);          \
      if (int rc = check_launch()) return rc;                                                                        \
      return nan_flag ? fpq_internal_zero_if_flag(out, tokens * outs * 2, nan_flag, stream) : FPQ_OK;                \
    }                                                                                                                \
  } while (0)
  if (cfg == 30) FPQ_GEMM_FC1(2, 4);
  if (cfg == 10) FPQ_GEMM_FC1(8, 4);
  FPQ_GEMM_FC1(4, 4);
#undef FPQ_GEMM_FC1
  return FPQ_ERR_SHAPE;   // K too long for the LDS-DMA kernel's scale tiles
}
int fpq_gemm_fp4_gelu_dual(const uint8_t* a_codes, const void* a_scales, const uint8_t* w_codes, const void* w_scales,
                           int w_scale_dtype, const void* bias, void* out, void* gelu_out, int64_t tokens, int64_t outs,
                           int64_t k, void* nan_flag, fpq_stream_t stream) {
  return gemm_fp4_gelu_dual_impl(a_codes, a_scales, w_codes, w_scales, w_scale_dtype, bias, out, gelu_out, tokens, outs, k, nan_flag, false, stream);
}
int fpq_gemm_fp4_gelu_dual_km(const uint8_t* a_image, const void* a_scales, const uint8_t* w_image, const void* w_scales,
                              int w_scale_dtype, const void* bias, void* out, void* gelu_out, int64_t tokens, int64_t outs,
                              int64_t k, void* nan_flag, fpq_stream_t stream) {
  return gemm_fp4_gelu_dual_impl(a_image, a_scales, w_image, w_scales, w_scale_dtype, bias, out, gelu_out, tokens, outs, k, nan_flag, true, stream);
}

static int gemm_fp6_rows_impl(const uint8_t* a_codes, const void* a_scales, int a_scale_dtype, const uint8_t* w_codes,
                              const void* w_scales, int w_scale_dtype, const void* bias, void* out, int64_t tokens,
                              int64_t outs, int64_t k, const fpq_gemm_epilogue_t* epilogue, bool km, fpq_stream_t stream) {
  if (tokens < 0 || outs < 0 || k < 0) return FPQ_ERR_ARG;
  GemmEpi epi;
  if (int rc = gemm_epilogue(epilogue, tokens, &epi)) return rc;
  if (km) {
    if (outs + 63 > 0x7FFFFFFF) return FPQ_ERR_SHAPE;
    epi.km_w_rows = (int)((outs + 63) / 64 * 64);
  }
  if (k % 128 != 0 || outs % 8 != 0 || tokens > 0x7FFFFFFF || outs > 0x7FFFFFFF || k > 0x7FFFFFFF) return FPQ_ERR_SHAPE;
  if ((a_scale_dtype != FPQ_F16 && a_scale_dtype != FPQ_F32) || (w_scale_dtype != FPQ_F16 && w_scale_dtype != FPQ_F32))
    return FPQ_ERR_DTYPE;
  if (tokens == 0 || outs == 0) return FPQ_OK;
  if (k == 0 || !a_codes || !a_scales || !w_codes || !w_scales || !out) return FPQ_ERR_ARG;
  if ((((uintptr_t)a_codes | (uintptr_t)w_codes | (uintptr_t)out) & 15) != 0) return FPQ_ERR_ARG;
  // the LDS-DMA pieces address a tile by a 32-bit lane offset (row inside the tile x row bytes + chunk): the 256-row tile's
  // last row must stay below 2^32 (k above ~22 M would wrap and read wrong rows silently)
  if (255 * (k * 3 / 4) + 128 >= (1ll << 32)) return FPQ_ERR_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  // FPQ_GEMM6_CFG 0: 128 x 128 tiles, 1: 256 x 128 (default for tall problems)
  // 256 x 128 tiles from 4096 tokens on for the wide Linears, from 32768 on for outs < 4096 (tools/gemm_small_steps.py fp6)
  const int cfg6 = fpq_opt_set(OPT_FPQ_GEMM6_CFG) ? fpq_opt(OPT_FPQ_GEMM6_CFG, 0) : (tokens >= 4096 && (outs >= 4096 || tokens >= 32768) ? 1 : 0);
#define FPQ_GO6(TA, TW, MT, NT)                                                                                     \
  do {                                                                                                               \
    using Cfg = GemmFp6Cfg<MT, NT>;                                                                                  \
    const int64_t n_col = (outs + Cfg::BN - 1) / Cfg::BN, n_row = (tokens + Cfg::BM - 1) / Cfg::BM;                  \
    const int64_t n_wg = 8 * ((n_col + 7) / 8) * n_row;                                                              \
    if (n_wg > 0x7FFFFFFF) return FPQ_ERR_SHAPE;                                                                     \
    hipLaunchKernelGGL((gemm_fp6_rows_kernel<TA, TW, MT, NT>), dim3((unsigned)n_wg), dim3(256), Cfg::lds(), st,     \
                       a_codes, (const TA*)a_scales, w_codes, (const TW*)w_scales, (const _Float16*)bias,            \
                       (_Float16*)out, (int)tokens, (int)outs, (int)k, epi);                                              \
  } while (0)
#define FPQ_GO6T(MT, NT)                                                                                             \
  do {                                                                                                               \
    if (a_scale_dtype == FPQ_F16 && w_scale_dtype == FPQ_F16) FPQ_GO6(_Float16, _Float16, MT, NT);                   \
    else if (a_scale_dtype == FPQ_F16) FPQ_GO6(_Float16, float, MT, NT);                                             \
    else if (w_scale_dtype == FPQ_F16) FPQ_GO6(float, _Float16, MT, NT);                                             \
    else FPQ_GO6(float, float, MT, NT);                                                                              \
  } while (0)
  if (cfg6 == 1) FPQ_GO6T(8, 4);
  else FPQ_GO6T(4, 4);
#undef FPQ_GO6T
#undef FPQ_GO6
  return check_launch();
}
int fpq_gemm_fp6_rows_ex(const uint8_t* a_codes, const void* a_scales, int a_scale_dtype, const uint8_t* w_codes,
                         const void* w_scales, int w_scale_dtype, const void* bias, void* out, int64_t tokens,
                         int64_t outs, int64_t k, const fpq_gemm_epilogue_t* epilogue, fpq_stream_t stream) {
  return gemm_fp6_rows_impl(a_codes, a_scales, a_scale_dtype, w_codes, w_scales, w_scale_dtype, bias, out, tokens, outs, k, epilogue, false, stream);
}
int fpq_gemm_fp6_rows_km(const uint8_t* a_image, const void* a_scales, int a_scale_dtype, const uint8_t* w_image,
                         const void* w_scales, int w_scale_dtype, const void* bias, void* out, int64_t tokens,
                         int64_t outs, int64_t k, const fpq_gemm_epilogue_t* epilogue, fpq_stream_t stream) {
  return gemm_fp6_rows_impl(a_image, a_scales, a_scale_dtype, w_image, w_scales, w_scale_dtype, bias, out, tokens, outs, k, epilogue, true, stream);
}

// Row-major codes -> k-major image (include/fpq.h): one thread per 16-byte chunk of the image.  seg = bytes of a row per K step of
// 128 elements (64: FP4 nibbles, 96: dense 6-bit codes); dealt: the weight side's row order, image rows = rows rounded up to 64
// (rows past the tensor's end are zero).
__global__ __launch_bounds__(256) void codes_to_kmajor_kernel(const u32x4* __restrict__ codes, u32x4* __restrict__ image, int64_t rows,
                                                              int64_t image_rows, int steps, int seg, int dealt) {
  const int cps = seg >> 4;   // chunks per segment
  const int64_t n = (int64_t)steps * image_rows * cps;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int pc = (int)(i % cps);
    const int64_t sj = i / cps;
    const int64_t j = sj % image_rows;
    const int s = (int)(sj / image_rows);
    const int c = seg == 64 ? (pc ^ glds_chunk_perm((int)(j & 15))) : (pc - fp6_rot((int)(j & 31)) + 6) % 6;
    const int64_t row = dealt ? (j & ~(int64_t)63) + 4 * (j & 15) + ((j >> 4) & 3) : j;
    u32x4 v = u32x4{0, 0, 0, 0};
    if (row < rows) v = codes[(row * steps + s) * cps + c];
    image[i] = v;
  }
}
int fpq_scales_to_kmajor(const void* scales, int scale_dtype, float* image, int64_t rows, int64_t groups, int weight_side,
                         fpq_stream_t stream) {
  if (rows < 0 || groups < 0 || groups > 0x7FFFFFFF) return FPQ_ERR_ARG;
  if (scale_dtype != FPQ_F16 && scale_dtype != FPQ_F32) return FPQ_ERR_DTYPE;
  if (rows == 0 || groups == 0) return FPQ_OK;
  if (!scales || !image || ((uintptr_t)image & 15) != 0) return FPQ_ERR_ARG;
  const int64_t image_rows = weight_side ? (rows + 63) / 64 * 64 : (rows + 3) / 4 * 4;
  const dim3 grid(grid_for((groups * image_rows + 255) / 256, 1 << 16));
  if (scale_dtype == FPQ_F16)
    hipLaunchKernelGGL(scales_to_kmajor_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, (const _Float16*)scales, image, rows,
                       image_rows, (int)groups);
  else
    hipLaunchKernelGGL(scales_to_kmajor_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)scales, image, rows,
                       image_rows, (int)groups);
  return check_launch();
}
int fpq_codes_to_kmajor(const uint8_t* codes, uint8_t* image, int64_t rows, int64_t k, int code_bits, int dealt, fpq_stream_t stream) {
  if (rows < 0 || k < 0 || (code_bits != 4 && code_bits != 6)) return FPQ_ERR_ARG;
  if (k % 128 != 0 || k / 128 > 0x7FFFFFFF) return FPQ_ERR_SHAPE;
  if (rows == 0 || k == 0) return FPQ_OK;
  if (!codes || !image || (((uintptr_t)codes | (uintptr_t)image) & 15) != 0) return FPQ_ERR_ARG;
  const int seg = code_bits == 4 ? 64 : 96;
  const int64_t image_rows = dealt ? (rows + 63) / 64 * 64 : rows;
  const int64_t n = (k / 128) * image_rows * (seg / 16);
  hipLaunchKernelGGL(codes_to_kmajor_kernel, dim3(grid_for((n + 255) / 256, 1 << 16)), dim3(256), 0, (hipStream_t)stream,
                     (const u32x4*)codes, (u32x4*)image, rows, image_rows, (int)(k / 128), seg, dealt ? 1 : 0);
  return check_launch();
}

int fpq_gemm_fp8_rows_ex(const uint8_t* a_codes, const void* a_scales, int a_scale_dtype, const uint8_t* w_codes,
                         const void* w_scales, int w_scale_dtype, const void* bias, void* out, int64_t tokens,
                         int64_t outs, int64_t k, const fpq_gemm_epilogue_t* epilogue, fpq_stream_t stream) {
  if (tokens < 0 || outs < 0 || k < 0) return FPQ_ERR_ARG;
  GemmEpi epi;
  if (int rc = gemm_epilogue(epilogue, tokens, &epi)) return rc;
  if (k % 128 != 0 || outs % 8 != 0 || tokens > 0x7FFFFFFF || outs > 0x7FFFFFFF || k > 0x7FFFFFFF) return FPQ_ERR_SHAPE;
  if ((a_scale_dtype != FPQ_F16 && a_scale_dtype != FPQ_F32) || (w_scale_dtype != FPQ_F16 && w_scale_dtype != FPQ_F32))
    return FPQ_ERR_DTYPE;
  if (tokens == 0 || outs == 0) return FPQ_OK;
  if (k == 0 || !a_codes || !a_scales || !w_codes || !w_scales || !out) return FPQ_ERR_ARG;
  if ((((uintptr_t)a_codes | (uintptr_t)w_codes | (uintptr_t)out) & 15) != 0) return FPQ_ERR_ARG;
  if (255 * k + 128 >= (1ll << 32)) return FPQ_ERR_SHAPE;   // 32-bit lane offsets inside a tile, as in fpq_gemm_fp6_rows_ex
  hipStream_t st = (hipStream_t)stream;
  const int cfg8 = fpq_opt(OPT_FPQ_GEMM8_CFG, 0);
#define FPQ_GO8(TA, TW, MT, NT)                                                                                     \
  do {                                                                                                               \
    using Cfg = GemmFp8Cfg<MT, NT>;                                                                                  \
    const int64_t n_col = (outs + Cfg::BN - 1) / Cfg::BN, n_row = (tokens + Cfg::BM - 1) / Cfg::BM;                  \
    const int64_t n_wg = 8 * ((n_col + 7) / 8) * n_row;                                                              \
    if (n_wg > 0x7FFFFFFF) return FPQ_ERR_SHAPE;                                                                     \
    hipLaunchKernelGGL((gemm_fp8_rows_kernel<TA, TW, MT, NT>), dim3((unsigned)n_wg), dim3(256), Cfg::lds(), st,     \
                       a_codes, (const TA*)a_scales, w_codes, (const TW*)w_scales, (const _Float16*)bias,            \
                       (_Float16*)out, (int)tokens, (int)outs, (int)k, epi);                                              \
  } while (0)
#define FPQ_GO8T(MT, NT)                                                                                             \
  do {                                                                                                               \
    if (a_scale_dtype == FPQ_F16 && w_scale_dtype == FPQ_F16) FPQ_GO8(_Float16, _Float16, MT, NT);                   \
    else if (a_scale_dtype == FPQ_F16) FPQ_GO8(_Float16, float, MT, NT);                                             \
    else if (w_scale_dtype == FPQ_F16) FPQ_GO8(float, _Float16, MT, NT);                                             \
    else FPQ_GO8(float, float, MT, NT);                                                                              \
  } while (0)
  if (cfg8 == 1) FPQ_GO8T(8, 4);
  else FPQ_GO8T(4, 4);
#undef FPQ_GO8T
#undef FPQ_GO8
  return check_launch();
}

}  // extern "C"
