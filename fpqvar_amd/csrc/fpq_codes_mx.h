// fpq_codes_mx.h - the operand-emitting quantizer of the FP4 matrix-core GEMM (fpq_gemm_fp4.h): per-group(128) E2M1 codes +
// scales.  Included by fpq_kernels.hip (the quantizers' translation unit) only - the GEMM kernels live in fpq_gemm.hip.
#pragma once

typedef float v2f_t __attribute__((ext_vector_type(2)));   // (fpq_gemm_fp4.h has its own: the two headers never meet in one translation unit)

// per-group(128) E2M1 quantization of fp16 rows straight to hardware nibbles + fp16 scales: the fused
// activation quantizer of fpq_fast16.h with the level table replaced by a code table
__global__ __launch_bounds__(kBlock) void rows16_codes_mx_kernel(const u32x4* __restrict__ x, uint32_t* __restrict__ codes,
                                                                uint16_t* __restrict__ scales, int64_t n_vec,
                                                                Lut16Args a, Lut16Tab tab, uint32_t km_rows, FastDiv groups_per_row) {
  __shared__ __attribute__((aligned(16))) uint16_t lut[kLutLdsEntries];   // static: a compile-time LDS address (a dynamic base is not folded into the ds_read offsets)
  {
    lut16_stage(lut, tab, a.shift);
    __syncthreads();
  }
  for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < n_vec; v += (int64_t)gridDim.x * kBlock) {
    const u32x4 w = __builtin_nontemporal_load(x + v);
    const uint32_t m = row_max_dpp<16>(vec_absmax16(w));
    const RowScale16 s = row_scale16(m, a.fpos.gmax, a.inv_gpos);
    if ((threadIdx.x & 15) == 0) scales[v >> 4] = (uint16_t)(s.s16x2 & 0xFFFFu);
    const uint32_t packed = codes_vec16(w, lut, a.shift, s.inv, s.inv_lo);
    if (km_rows) {   // k-major image (include/fpq.h): vector v = 4 bytes at byte 4 (v & 15) of group (v >> 4) % G of row (v >> 4) / G
      const uint32_t gi = (uint32_t)(v >> 4), t = fast_div_q(gi, groups_per_row), g = gi - t * groups_per_row.d;
      codes[(km4_off(t, g, ((uint32_t)v & 15u) >> 2, km_rows) >> 2) + ((uint32_t)v & 3u)] = packed;
    } else {
      codes[v] = packed;
    }
  }
}
