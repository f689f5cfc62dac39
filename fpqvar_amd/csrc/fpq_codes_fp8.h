// fpq_codes_fp8.h - the operand-emitting quantizers of the row-scaled FP8 GEMM (fpq_gemm_fp8.h): one E4M3 byte per element +
// one scale per row.  Included by fpq_kernels.hip only.
#pragma once

// Per-row quantization straight to E4M3 bytes + one scale per row (x's dtype): same scale / normalise / rounding
// arithmetic as fpq_quant_rows, i.e. level(code) * scale reproduces fp6_quant_*_per_token_cuda exactly.
// One workgroup per row, two passes over the row (the second one is served by L2).
template <typename Tin>
__global__ __launch_bounds__(kBlock) void rows_codes_fp8_kernel(const Tin* __restrict__ x, uint8_t* __restrict__ codes,
                                                               Tin* __restrict__ scales, int64_t rows, int64_t cols, Fmt f) {
  __shared__ uint32_t sh[kBlock / 64];
  const bool vec = (cols & 3) == 0 && (((uintptr_t)x | (uintptr_t)codes) & 15) == 0;
  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    const Tin* xr = x + row * cols;
    uint32_t m = 0;
    for (int64_t c = threadIdx.x; c < cols; c += kBlock) {
      const uint32_t ab = DT<Tin>::absbits(load_scalar<Tin>(xr + c));
      m = m > ab ? m : ab;
    }
    m = block_max(m, sh);
    const float s = scale_of<Tin>(m, f.gmax);
    if (threadIdx.x == 0) store_scalar<Tin>(scales + row, s);
    auto level = [&](float e) {
      const float xn = div_round<Tin>(e, s);
      const uint32_t neg = (xn < 0.0f) ? 1u : 0u;
      const float qm = quant_mag(fabsf(xn), neg, f);
      return (neg && qm != 0.0f) ? -qm : qm;
    };
    if (vec) {
      for (int64_t c = (int64_t)threadIdx.x * 4; c < cols; c += (int64_t)kBlock * 4) {
        const float q0 = level(load_scalar<Tin>(xr + c)), q1 = level(load_scalar<Tin>(xr + c + 1));
        const float q2 = level(load_scalar<Tin>(xr + c + 2)), q3 = level(load_scalar<Tin>(xr + c + 3));
        int w = __builtin_amdgcn_cvt_pk_fp8_f32(q0, q1, 0, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(q2, q3, w, true);
        *(uint32_t*)(codes + row * cols + c) = (uint32_t)w;
      }
    } else {
      for (int64_t c = threadIdx.x; c < cols; c += kBlock) {
        const int w = __builtin_amdgcn_cvt_pk_fp8_f32(level(load_scalar<Tin>(xr + c)), 0.0f, 0, false);
        codes[row * cols + c] = (uint8_t)(w & 0xFF);
      }
    }
  }
}

// Fast form for fp16 rows of up to 4096 elements (per-token activations, C = 1920 / 2304): one wavefront owns a row,
// the row stays in registers between the reduction and the rounding, levels come out of the same bucket table as the
// fake-quant kernels with E4M3 bytes as entries (fpq_fast16.h).
__device__ __forceinline__ void codes8_vec16(const u32x4& w, const uint16_t* lut, int shift, float inv_hi, float inv_lo,
                                             uint32_t& lo4, uint32_t& hi4) {
  uint32_t c[8];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t wk = w[k];
    const uint32_t rb = div_pair16(wk, inv_hi, inv_lo, inv_hi, inv_lo);
    const uint32_t u = pk_sub_u16(rb, pk_lshr_u16(rb, 15));
    c[2 * k] = lut[(u & 0xFFFFu) >> shift];
    c[2 * k + 1] = lut[u >> (16 + shift)];
  }
  lo4 = c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24);
  hi4 = c[4] | (c[5] << 8) | (c[6] << 16) | (c[7] << 24);
}

template <int MAXC>
__global__ __launch_bounds__(kBlock) void rows16_codes8_wave_kernel(const uint16_t* __restrict__ x, uint8_t* __restrict__ codes,
                                                                   uint16_t* __restrict__ scales, int64_t rows, int64_t cols,
                                                                   Lut16Args a, Lut16Tab tab) {
  __shared__ __attribute__((aligned(16))) uint16_t lut[kLutLdsEntries];   // static: a compile-time LDS address (a dynamic base is not folded into the ds_read offsets)
  {
    lut16_stage(lut, tab, a.shift);
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int64_t vpr = cols >> 3;
  constexpr int R = kBlock / 64;
  for (int64_t base = (int64_t)blockIdx.x * R; base < rows; base += (int64_t)gridDim.x * R) {
    const int64_t row = base + (threadIdx.x >> 6);
    if (row >= rows) continue;   // whole wavefront skips
    const u32x4* xr = (const u32x4*)(x + row * cols);
    u32x2* crow = (u32x2*)(codes + row * cols);
    u32x4 raw[MAXC];
    uint32_t m = 0;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int64_t v = (int64_t)c * 64 + lane;
      raw[c] = (v < vpr) ? __builtin_nontemporal_load(xr + v) : u32x4{0, 0, 0, 0};
      const uint32_t t = vec_absmax16(raw[c]);
      m = m > t ? m : t;
    }
    m = row_max_dpp<64>(m);
    const RowScale16 s = row_scale16(m, a.fpos.gmax, a.inv_gpos);
    if (lane == 0) scales[row] = (uint16_t)(s.s16x2 & 0xFFFFu);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int64_t v = (int64_t)c * 64 + lane;
      if (v < vpr) {
        uint32_t c_lo, c_hi;
        codes8_vec16(raw[c], lut, a.shift, s.inv, s.inv_lo, c_lo, c_hi);
        __builtin_nontemporal_store(u32x2{c_lo, c_hi}, crow + v);
      }
    }
  }
}
