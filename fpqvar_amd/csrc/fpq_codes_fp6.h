// fpq_codes_fp6.h - the operand-emitting quantizers of the row-scaled FP6 GEMM (fpq_gemm_fp6.h): dense 6-bit E2M3 codes + one
// scale per row.  Included by fpq_kernels.hip only.
#pragma once

// Fast form for fp16 rows (per-token activations): one wavefront per row, every lane owns whole 32-element
// k-blocks (64 bytes in, 24 bytes out), bucket table with 6-bit codes as entries.
template <int MAXC>
__global__ __launch_bounds__(kBlock) void rows16_codes6_wave_kernel(const uint16_t* __restrict__ x, uint8_t* __restrict__ codes,
                                                                   uint16_t* __restrict__ scales, int64_t rows, int64_t cols,
                                                                   Lut16Args a, Lut16Tab tab, uint32_t km_rows) {
  __shared__ __attribute__((aligned(16))) uint16_t lut[kLutLdsEntries];   // static: a compile-time LDS address (a dynamic base is not folded into the ds_read offsets)
  {
    lut16_stage(lut, tab, a.shift);
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int64_t nblk = cols >> 5;                 // 32-element k-blocks per row
  constexpr int R = kBlock / 64;
  for (int64_t base = (int64_t)blockIdx.x * R; base < rows; base += (int64_t)gridDim.x * R) {
    const int64_t row = base + (threadIdx.x >> 6);
    if (row >= rows) continue;   // whole wavefront skips
    const u32x4* xr = (const u32x4*)(x + row * cols);
    u32x4 raw[MAXC][4];
    uint32_t m = 0;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int64_t b = (int64_t)c * 64 + lane;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        raw[c][v] = (b < nblk) ? xr[4 * b + v] : u32x4{0, 0, 0, 0};
        const uint32_t t = vec_absmax16(raw[c][v]);
        m = m > t ? m : t;
      }
    }
    m = row_max_dpp<64>(m);
    const RowScale16 s = row_scale16(m, a.fpos.gmax, a.inv_gpos);
    if (lane == 0) scales[row] = (uint16_t)(s.s16x2 & 0xFFFFu);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int64_t b = (int64_t)c * 64 + lane;
      if (b < nblk) {
        uint32_t o[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          uint32_t lo4, hi4;                       // eight 6-bit codes, one per byte
          codes8_vec16(raw[c][v], lut, a.shift, s.inv, s.inv_lo, lo4, hi4);
          // 8 codes = 48 bits at bit offset 48 * v
          const uint64_t p48 = (uint64_t)((lo4 & 0x3Fu) | ((lo4 >> 2) & 0xFC0u) | ((lo4 >> 4) & 0x3F000u) | ((lo4 >> 6) & 0xFC0000u)) |
                               ((uint64_t)((hi4 & 0x3Fu) | ((hi4 >> 2) & 0xFC0u) | ((hi4 >> 4) & 0x3F000u) | ((hi4 >> 6) & 0xFC0000u)) << 24);
          const int bit = 48 * v;
          o[bit >> 5] |= (uint32_t)(p48 << (bit & 31));
          o[(bit >> 5) + 1] |= (uint32_t)(p48 >> (32 - (bit & 31)));
          if ((bit & 31) + 48 > 64) o[(bit >> 5) + 2] |= (uint32_t)(p48 >> (64 - (bit & 31)));
        }
        if (km_rows) {   // k-major image (include/fpq.h): block b = bytes 24 (b & 3) .. + 23 of K step b >> 2, three 8-byte halves of chunks
          // (plain stores: a row leaves 96 bytes per plane, the workgroup's four consecutive rows fill three whole 128-byte lines
          // between them - L2 has to be allowed to merge them; streamed past it the partial lines cost 2 x the kernel's time)
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const uint32_t w = 24u * ((uint32_t)b & 3u) + 8u * j;
            *(u32x2*)(codes + km6_off((uint32_t)row, (uint32_t)b >> 2, w >> 4, km_rows) + (w & 15u)) = u32x2{o[2 * j], o[2 * j + 1]};
          }
        } else {
          u32x2* dst = (u32x2*)(codes + row * (nblk * 24) + b * 24);
          __builtin_nontemporal_store(u32x2{o[0], o[1]}, dst);
          __builtin_nontemporal_store(u32x2{o[2], o[3]}, dst + 1);
          __builtin_nontemporal_store(u32x2{o[4], o[5]}, dst + 2);
        }
      }
    }
  }
}

// level (exactly an E2M3 number, sign included) -> 6-bit code
__host__ __device__ __forceinline__ uint32_t e2m3_of_level(float q) {
  const uint32_t sgn = (q < 0.0f) ? 32u : 0u;
  const float a = q < 0.0f ? -q : q;
  uint32_t mag;
  if (a < 1.0f) mag = (uint32_t)(a * 8.0f);                                        // subnormal: m / 8
  else if (a < 2.0f) mag = (1u << 3) | (uint32_t)((a - 1.0f) * 8.0f);
  else if (a < 4.0f) mag = (2u << 3) | (uint32_t)((a * 0.5f - 1.0f) * 8.0f);
  else mag = (3u << 3) | (uint32_t)((a * 0.25f - 1.0f) * 8.0f);
  return sgn | mag;
}

// Generic form (fp32 weights, long or unaligned rows): one workgroup per row, a thread packs whole 32-element blocks.
template <typename Tin>
__global__ __launch_bounds__(kBlock) void rows_codes_fp6_kernel(const Tin* __restrict__ x, uint8_t* __restrict__ codes,
                                                               Tin* __restrict__ scales, int64_t rows, int64_t cols, Fmt f,
                                                               uint32_t km_rows) {
  __shared__ uint32_t sh[kBlock / 64];
  const int64_t nblk = cols >> 5;
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
    for (int64_t b = threadIdx.x; b < nblk; b += kBlock) {
      uint32_t o[6] = {0, 0, 0, 0, 0, 0};
      for (int j = 0; j < 32; ++j) {
        const float xn = div_round<Tin>(load_scalar<Tin>(xr + b * 32 + j), s);
        const uint32_t neg = (xn < 0.0f) ? 1u : 0u;
        const float qm = quant_mag(fabsf(xn), neg, f);
        const uint32_t code = e2m3_of_level((neg && qm != 0.0f) ? -qm : qm);
        const int bit = 6 * j;
        o[bit >> 5] |= code << (bit & 31);
        if ((bit & 31) > 26) o[(bit >> 5) + 1] |= code >> (32 - (bit & 31));
      }
      if (km_rows) {
        for (int i = 0; i < 6; ++i) {
          const uint32_t w = 24u * ((uint32_t)b & 3u) + 4u * i;
          *(uint32_t*)(codes + km6_off((uint32_t)row, (uint32_t)(b >> 2), w >> 4, km_rows) + (w & 15u)) = o[i];
        }
      } else {
        uint32_t* dst = (uint32_t*)(codes + row * (nblk * 24) + b * 24);
        for (int i = 0; i < 6; ++i) dst[i] = o[i];
      }
    }
  }
}
