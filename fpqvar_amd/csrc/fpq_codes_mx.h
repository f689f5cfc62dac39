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
  if (km_rows) {
    // k-major image (include/fpq.h): a workgroup's 256 vectors are ONE group of 16 consecutive rows (16 lanes per row as before), so
    // that what it writes - 64 bytes per row into the group's plane - is 1 KiB in a row; a flat run of 16 groups would scatter 64-byte
    // halves of lines over 15 planes, with the other half written by a workgroup on another XCD.  unit u = row block u / G, group u % G.
    const uint32_t G = groups_per_row.d, lane16 = threadIdx.x & 15u, row_in = threadIdx.x >> 4;
    const uint32_t units = ((km_rows + 15u) >> 4) * G;
    for (uint32_t u = blockIdx.x; u < units; u += gridDim.x) {
      const uint32_t rb = fast_div_q(u, groups_per_row), g = u - rb * G, t = rb * 16u + row_in;
      if (t >= km_rows) continue;   // whole 16-lane clusters drop out
      const int64_t v = ((int64_t)t * G + g) * 16 + lane16;
      const u32x4 w = __builtin_nontemporal_load(x + v);
      const uint32_t m = row_max_dpp<16>(vec_absmax16(w));
      const RowScale16 s = row_scale16(m, a.fpos.gmax, a.inv_gpos);
      if (lane16 == 0)   // the scale as fp32 into the k-major scale image [G][rows rounded up to 4] (include/fpq.h)
        ((float*)scales)[(int64_t)g * ((km_rows + 3u) & ~3u) + t] = (float)__builtin_bit_cast(_Float16, (uint16_t)(s.s16x2 & 0xFFFFu));
      codes[(km4_off(t, g, lane16 >> 2, km_rows) >> 2) + (lane16 & 3u)] = codes_vec16(w, lut, a.shift, s.inv, s.inv_lo);
    }
    return;
  }
  for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < n_vec; v += (int64_t)gridDim.x * kBlock) {
    const u32x4 w = __builtin_nontemporal_load(x + v);
    const uint32_t m = row_max_dpp<16>(vec_absmax16(w));
    const RowScale16 s = row_scale16(m, a.fpos.gmax, a.inv_gpos);
    if ((threadIdx.x & 15) == 0) scales[v >> 4] = (uint16_t)(s.s16x2 & 0xFFFFu);
    codes[v] = codes_vec16(w, lut, a.shift, s.inv, s.inv_lo);
  }
}
