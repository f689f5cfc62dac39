// fpq_gemm_fp4.h - F2: a REAL low-precision consumer for the quantized
// activations and weights.  Included by fpq_kernels.hip inside its anonymous namespace.
//
// The reference fake-quantizes and then runs an fp16 GEMM on the de-quantized tensors
// (tr/quant_utils.py:765-767: F.linear(act_quant(x), W_q)).  With per-group(128) FP4-E2M1 on both
// sides the same product is
//     y[t,o] = sum_g  s_a[t,g] * s_w[o,g] * ( sum_{k in g} La[t,k] * Lw[o,k] )
// and the inner 128-term dot product of FP4 levels is exactly ONE gfx950 block-scaled MFMA
// (v_mfma_scale_f32_16x16x128_f8f6f4, FP4 operands, unit E8M0 scales): exact products, fp32
// accumulation.  The two per-group scales (arbitrary fp16/fp32 numbers, not powers of two) are
// applied to the 16x16 partial tile in the packed-fp32 VALU.
//
// Operand layout (probed on hardware, tools/probe/mfma_fp4_probe.hip): lane l supplies row l&15
// of A (column l&15 of B), k-block l>>4 = 32 consecutive k as 32 nibbles (16 bytes, low nibble
// first); D: column l&15, rows 4*(l>>4) + reg.  Nibble = OCP E2M1: bit 3 sign, bits 2:0 magnitude
// index into {0, .5, 1, 1.5, 2, 3, 4, 6}.
//
// Numerics: more exact than the reference (it rounds every de-quantized value to fp16 before its
// GEMM); the parity contract for this entry point is a tolerance, not bit equality.
#pragma once

typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef float v2f_t __attribute__((ext_vector_type(2)));

constexpr int kGemmBM = 128, kGemmBN = 128;   // workgroup tile (tokens x outputs), 4 wavefronts as 2 x 2
constexpr int kGemmMT = 4, kGemmNT = 4;       // 16x16 MFMA tiles per wavefront: 64 x 64

// Byte offset of (row, 16-byte chunk kb) in the 8 KiB LDS image of a 128-row operand tile.
// ds_read_b128 serves a wavefront in four fixed 16-lane groups - {0-3,12-15,20-27}, {4-11,16-19,28-31} and
// the same +32 (MI355X_MICROARCH.md, LDS) - and with lane = (kb << 4) | row each group mixes two k-blocks.
// Layout: per 16 rows one 1 KiB block = 4 planes (one per kb) of 16 slots; row r sits in slot
// (r&3)*4 + (r>>2) of its plane, planes 2 and 3 rotated by two slots.  Every read group then covers 16
// distinct 16-byte slots (conflict-free); the staging writes (4 lanes per row) are 2-way.
__device__ __forceinline__ int gemm_lds_off(int row, int kb) {
  const int r = row & 15;
  return ((row >> 4) << 10) + (kb << 8) + (((((r & 3) << 2) + (r >> 2) + ((kb >> 1) << 1)) & 15) << 4);
}

template <typename Tsw>
__global__ __launch_bounds__(256) void gemm_fp4_kernel(const uint8_t* __restrict__ A, const _Float16* __restrict__ sa,
                                                      const uint8_t* __restrict__ W, const Tsw* __restrict__ sw,
                                                      const _Float16* __restrict__ bias, _Float16* __restrict__ out,
                                                      int T, int O, int C) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int G = C >> 7, row_bytes = C >> 1;
  // two operand buffers (A 8 KiB + B 8 KiB each): group g+1 is written while group g is multiplied,
  // one barrier per group
  uint8_t* lAB = smem;                        // [2][A 8 KiB | B 8 KiB]
  float* lsa = (float*)(smem + 32768);        // [G][128]
  float* lsw = lsa + G * 128;                 // [G][128]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order (workgroups are dealt round-robin over the 8 XCDs, each with its own 4 MiB L2):
  // XCD k owns a contiguous band of `cpx` column tiles and walks all row tiles of that band, so its
  // slice of W (cpx * 128 rows) stays L2-resident and an A tile is re-used by cpx consecutive workgroups.
  const int n_col = (O + kGemmBN - 1) / kGemmBN, n_row = (T + kGemmBM - 1) / kGemmBM;
  const int cpx = (n_col + 7) >> 3;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int col_blk = xcd * cpx + local % cpx, row_blk = local / cpx;
  if (col_blk >= n_col || row_blk >= n_row) return;   // uniform over the workgroup
  const int t0 = row_blk * kGemmBM, o0 = col_blk * kGemmBN;

  // scales of this tile, transposed to [g][row] and widened to fp32
  for (int i = tid; i < G * 128; i += 256) {
    const int r = i & 127, g = i >> 7;
    const int t = t0 + r, o = o0 + r;
    lsa[g * 128 + r] = (t < T) ? (float)sa[(int64_t)t * G + g] : 0.0f;
    lsw[g * 128 + r] = (o < O) ? (float)sw[(int64_t)o * G + g] : 0.0f;
  }

  v4f_t acc[kGemmMT][kGemmNT];
#pragma unroll
  for (int m = 0; m < kGemmMT; ++m)
#pragma unroll
    for (int n = 0; n < kGemmNT; ++n) acc[m][n] = v4f_t{0, 0, 0, 0};

  // staging assignment: 512 chunks of 16 B per operand tile, 2 per thread.  Software pipeline: the
  // global loads of group g+1 are in flight while group g is multiplied out of LDS.
  const int sr0 = tid >> 2, sc = tid & 3;     // rows sr0 and sr0 + 64, chunk sc
  const uint8_t* pa[2];
  const uint8_t* pb[2];
  bool va[2], vb[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = sr0 + 64 * h;
    va[h] = t0 + r < T;
    vb[h] = o0 + r < O;
    pa[h] = A + (int64_t)(va[h] ? t0 + r : 0) * row_bytes + sc * 16;
    pb[h] = W + (int64_t)(vb[h] ? o0 + r : 0) * row_bytes + sc * 16;
  }
  u32x4 ga[2], gb[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    ga[h] = va[h] ? *(const u32x4*)(pa[h]) : u32x4{0, 0, 0, 0};
    gb[h] = vb[h] ? *(const u32x4*)(pb[h]) : u32x4{0, 0, 0, 0};
  }
  // prologue: group 0 into buffer 0, group 1 into the registers
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = sr0 + 64 * h;
    *(u32x4*)(lAB + gemm_lds_off(r, sc)) = ga[h];
    *(u32x4*)(lAB + 8192 + gemm_lds_off(r, sc)) = gb[h];
  }
  if (G > 1) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      ga[h] = va[h] ? *(const u32x4*)(pa[h] + 64) : u32x4{0, 0, 0, 0};
      gb[h] = vb[h] ? *(const u32x4*)(pb[h] + 64) : u32x4{0, 0, 0, 0};
    }
  }
  __syncthreads();   // buffer 0 and the scale tiles are visible
  for (int g = 0; g < G; ++g) {
    const uint8_t* lA = lAB + (g & 1) * 16384;
    const uint8_t* lB = lA + 8192;
    if (g + 1 < G) {   // stage group g+1 into the other buffer (last read in iteration g-1, barrier since)
      uint8_t* nA = lAB + ((g + 1) & 1) * 16384;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = sr0 + 64 * h;
        *(u32x4*)(nA + gemm_lds_off(r, sc)) = ga[h];
        *(u32x4*)(nA + 8192 + gemm_lds_off(r, sc)) = gb[h];
      }
      if (g + 2 < G) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          ga[h] = va[h] ? *(const u32x4*)(pa[h] + (g + 2) * 64) : u32x4{0, 0, 0, 0};
          gb[h] = vb[h] ? *(const u32x4*)(pb[h] + (g + 2) * 64) : u32x4{0, 0, 0, 0};
        }
      }
    }
    v8i_t af[kGemmMT], bf[kGemmNT];
#pragma unroll
    for (int m = 0; m < kGemmMT; ++m) {
      const u32x4 v = *(const u32x4*)(lA + gemm_lds_off(wm * 64 + m * 16 + (lane & 15), lane >> 4));
      af[m] = v8i_t{(int)v[0], (int)v[1], (int)v[2], (int)v[3], 0, 0, 0, 0};
    }
#pragma unroll
    for (int n = 0; n < kGemmNT; ++n) {
      const u32x4 v = *(const u32x4*)(lB + gemm_lds_off(wn * 64 + n * 16 + (lane & 15), lane >> 4));
      bf[n] = v8i_t{(int)v[0], (int)v[1], (int)v[2], (int)v[3], 0, 0, 0, 0};
    }
    v4f_t sa4[kGemmMT];
    float sw1[kGemmNT];
#pragma unroll
    for (int m = 0; m < kGemmMT; ++m) sa4[m] = *(const v4f_t*)(lsa + g * 128 + wm * 64 + m * 16 + 4 * (lane >> 4));
#pragma unroll
    for (int n = 0; n < kGemmNT; ++n) sw1[n] = lsw[g * 128 + wn * 64 + n * 16 + (lane & 15)];
    // four independent MFMAs per tile row, then their scale-and-accumulate in the packed-fp32 VALU:
    // the matrix pipe works on row m+1 while the VALU finishes row m
#pragma unroll
    for (int m = 0; m < kGemmMT; ++m) {
      v4f_t d[kGemmNT];
#pragma unroll
      for (int n = 0; n < kGemmNT; ++n)
        d[n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af[m], bf[n], v4f_t{0, 0, 0, 0}, 4, 4, 0, 127, 0, 127);
#pragma unroll
      for (int n = 0; n < kGemmNT; ++n) {
        const v4f_t p = sa4[m] * sw1[n];
        acc[m][n] = __builtin_elementwise_fma(d[n], p, acc[m][n]);
      }
    }
    __syncthreads();   // group g consumed, group g+1 staged
  }

  // epilogue: bias, fp16, transpose each 64x64 wavefront tile through LDS for 16-byte row stores
  _Float16* lo = (_Float16*)smem + wave * (64 * 72);   // 64 rows x (64 + 8 pad) halves = 9216 B per wavefront
#pragma unroll
  for (int m = 0; m < kGemmMT; ++m)
#pragma unroll
    for (int n = 0; n < kGemmNT; ++n) {
      const int col = n * 16 + (lane & 15);
      const int o = o0 + wn * 64 + col;
      const float b = (bias && o < O) ? (float)bias[o] : 0.0f;
#pragma unroll
      for (int i = 0; i < 4; ++i) lo[(m * 16 + 4 * (lane >> 4) + i) * 72 + col] = (_Float16)(acc[m][n][i] + b);
    }
  __syncthreads();
  // every lane writes one 16-byte piece (8 halves) per pass: 64 rows x 8 pieces = 512 pieces, 8 passes
#pragma unroll
  for (int pass = 0; pass < 8; ++pass) {
    const int piece = pass * 64 + lane;
    const int r = piece >> 3, cpc = piece & 7;
    const int t = t0 + wm * 64 + r, o = o0 + wn * 64 + cpc * 8;
    if (t < T && o + 8 <= O) {
      const u32x4 v = *(const u32x4*)(lo + r * 72 + cpc * 8);
      *(u32x4*)(out + (int64_t)t * O + o) = v;
    } else if (t < T) {
      for (int e = 0; e < 8; ++e)
        if (o + e < O) out[(int64_t)t * O + o + e] = lo[r * 72 + cpc * 8 + e];
    }
  }
}

// per-group(128) E2M1 quantization of fp16 rows straight to hardware nibbles + fp16 scales: the fused
// activation quantizer of fpq_fast16.h with the level table replaced by a code table
__global__ __launch_bounds__(kBlock) void rows16_codes_mx_kernel(const u32x4* __restrict__ x, uint32_t* __restrict__ codes,
                                                                uint16_t* __restrict__ scales, int64_t n_vec,
                                                                Lut16Args a, Lut16Tab tab) {
  extern __shared__ __attribute__((aligned(16))) uint16_t lut[];
  {
    const int n = 1 << (16 - a.shift);
    for (int i = threadIdx.x; i < n; i += kBlock) lut[i] = tab.e[i];
    __syncthreads();
  }
  for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < n_vec; v += (int64_t)gridDim.x * kBlock) {
    const u32x4 w = __builtin_nontemporal_load(x + v);
    const uint32_t m = row_max_dpp<16>(vec_absmax16(w));
    const RowScale16 s = row_scale16(m, a.fpos.gmax, a.inv_gpos);
    if ((threadIdx.x & 15) == 0) scales[v >> 4] = (uint16_t)(s.s16x2 & 0xFFFFu);
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t wk = w[k];
      const float x0 = h2f(wk & 0xFFFFu), x1 = h2f(wk >> 16);
      const float y0 = mul_h_lo(wk, s.inv), y1 = mul_h_hi(wk, s.inv);
      const float e0 = __builtin_fmaf(-y0, s.sf, x0), e1 = __builtin_fmaf(-y1, s.sf, x1);
      const float r0 = __builtin_fmaf(e0, s.inv, y0), r1 = __builtin_fmaf(e1, s.inv, y1);
      const uint32_t rb = f2h2(r0, r1);
      const uint32_t u = pk_sub_u16(rb, pk_lshr_u16(rb, 15));
      const uint32_t c0 = lut[(u & 0xFFFFu) >> a.shift], c1 = lut[u >> (16 + a.shift)];
      packed |= (c0 | (c1 << 4)) << (8 * k);
    }
    codes[v] = packed;
  }
}
