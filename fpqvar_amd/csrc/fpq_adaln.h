// fpq_adaln.h - the complete producer of tr/basic_var.py:263 / :266 for fp16 activations, second generation:
//     h  = half( fma( LN(x), A, B ) ),   A = half(scale + 1) * s,  B = shift * s      (fp32; LN without affine)
//     x1 = half( FWHT128( c_h * (h * D) ) )
//     q  = per-group(128) quant(x1)
// Included by fpq_kernels.hip after fpq_fast16.h (whose quantizer, butterfly and reductions it uses).
//
// What the counters said about the first generation (adaln_rotate_quant16_kernel, profiles/r02_pmc_adaln_before.txt):
// 41 VALU instructions per element, but the vector pipe only 38 % busy; 134-140 VGPRs = 3 wavefronts per SIMD, each
// walking its rows strictly load -> reduce -> compute -> store, 42 % of the wavefront-cycles waiting on memory.
// Latency-bound, not throughput-bound.  Hence:
//   * a workgroup owns ROWS consecutive rows of ONE batch entry and stages the modulation of that entry - already
//     folded with the smoothing vector: A, B above - in LDS once; the per-element modulate is one fma instead of
//     add, mul, add, mul plus two 16-byte modulation loads and their conversions per vector;
//   * rows stay packed fp16 in registers (16 VGPRs per row instead of 32 + 16): LayerNorm reads them through
//     v_dot2_f32_f16 (sum, sum of squares) and v_fma_mix_f32 (normalise) - the widening rides on the arithmetic;
//   * the rotation's sign vector is folded into the staged modulation, the first butterfly stage reads the packed
//     halves directly (fwht128_h_n): no sign flips and no fp16 -> fp32 conversions in the row loop;
//   * the NEXT row's loads are issued before the current row is processed (software prefetch);
//   * wavefront sums are DPP + v_permlane16/32_swap: no LDS round trip, no address arithmetic;
//   * straight-line code: the launch picks MAXC = ceil(vectors per row / 64) exactly, so only the LAST vector of a
//     lane can fall outside the row; it is loaded from a clamped address and zeroed, and its store is the one
//     exec-masked branch of the row loop.  Vectors go through the stages two at a time (stage-major source order:
//     two independent dependency chains per wavefront for the in-order issue to interleave).
// (Tried and dropped: scaling by c_h in front of the butterfly, where it would ride on the fp16 -> fp32 widening.
//  The butterfly of raw fp16 values is nearly always EXACT in fp32 - 11-bit inputs of similar magnitude - while
//  19-bit inputs are not: rotated values went from <= 1 to 3 fp16 ulp off the fp64 product.)
// LayerNorm statistics: one pass (E[x^2] - mean^2) with a centred second pass for rows where that cancels; fp32
// rounding of the folded modulate differs from torch's three separate ops by a few 2^-24 - the parity contract of this entry point
// is the fuzzy one of SURVEY.md section 7 (tests/test_gpu_parity.py::test_adaln_rotate_quant_fused: h within half an
// fp16 ulp + 4e-6 relative, rotated values and quantization bit-exact given h).
#pragma once

// D = (float)half_lo(w) * b + c   /   (float)half_hi(w) * b + c
__device__ __forceinline__ float fma_h_lo(uint32_t w, float b, float c) {
  float d;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(w), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float fma_h_hi(uint32_t w, float b, float c) {
  float d;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(w), "v"(b), "v"(c));
  return d;
}

// sum over the 64 lanes, result in every lane: 4 DPP adds inside the rows of 16, then the rows trade sums
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
  auto r = __builtin_amdgcn_permlane16_swap(fbits(v), fbits(v), false, false);
  v = u2f(r[0]) + u2f(r[1]);
  r = __builtin_amdgcn_permlane32_swap(fbits(v), fbits(v), false, false);
  return u2f(r[0]) + u2f(r[1]);
}

// two sums at once (the steps interleave)
__device__ __forceinline__ void wave_sum2_dpp(float& a, float& b) {
#define FPQ_DPP_ADD(ctrl)                                                                         \
  a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), ctrl, 0xF, 0xF, true));   \
  b += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b), ctrl, 0xF, 0xF, true));
  FPQ_DPP_ADD(0xB1) FPQ_DPP_ADD(0x4E) FPQ_DPP_ADD(0x141) FPQ_DPP_ADD(0x140)
#undef FPQ_DPP_ADD
  auto ra = __builtin_amdgcn_permlane16_swap(fbits(a), fbits(a), false, false);
  auto rb = __builtin_amdgcn_permlane16_swap(fbits(b), fbits(b), false, false);
  a = u2f(ra[0]) + u2f(ra[1]);
  b = u2f(rb[0]) + u2f(rb[1]);
  ra = __builtin_amdgcn_permlane32_swap(fbits(a), fbits(a), false, false);
  rb = __builtin_amdgcn_permlane32_swap(fbits(b), fbits(b), false, false);
  a = u2f(ra[0]) + u2f(ra[1]);
  b = u2f(rb[0]) + u2f(rb[1]);
}

typedef _Float16 h2v_t __attribute__((ext_vector_type(2)));

// build-time knobs for A/B experiments (tools/ab_quant.py); the defaults are the measured best
#ifndef FPQ_ADALN_PREFETCH
#define FPQ_ADALN_PREFETCH 1
#endif
#ifndef FPQ_ADALN_N2
#define FPQ_ADALN_N2 2
#endif
#ifndef FPQ_ADALN_WAVES
#define FPQ_ADALN_WAVES 0
#endif
#if FPQ_ADALN_WAVES > 0
#define FPQ_ADALN_OCC __attribute__((amdgpu_waves_per_eu(FPQ_ADALN_WAVES, 8)))
#else
#define FPQ_ADALN_OCC
#endif

#ifndef FPQ_ADALN_PREFETCH32
#define FPQ_ADALN_PREFETCH32 1
#endif
#ifndef FPQ_ADALN_X32_WAVES   // fp32 rows of 13 .. 16 groups with the next row prefetched: 130 registers = 3 wavefronts per SIMD; capped at 128 (= 4, with 3 - 5 spilled) measured the same 134 - 138 us: left uncapped
#define FPQ_ADALN_X32_WAVES 1
#endif

// MFMA: the rotation on the matrix cores (fpq_rotate_mfma.h: a row of up to 16 groups is one tile; value output only).
// X32: fp32 rows - the model's case: the residual stream is fp32 under the reference's autocast (tr/var.py:168,209:
// fp16 Linear output + fp32 position embedding; tr/basic_var.py:264,267: fp32 x + fp16 branch).  A lane then loads
// 4-float vectors n * 64 + lane of the row (n = 0 .. 2 MAXC - 1, every load a fully coalesced 16 bytes per lane; a
// 32-bytes-per-lane mapping would leave each load instruction half of every 128-byte line, see fpq_fast32.h): that is
// half h = lane & 1 of the 8-element chunk 32 n + lane / 2.  Statistics do not care, the modulation planes are kept by
// halves anyway, and the lane's fp16 results (8 bytes) meet their other half in LDS - straight in the matrix-core
// operand image, or in a row buffer that is read back in the one-chunk-per-lane order of the butterfly forms.  No
// cross-lane exchange, no second kernel.
template <typename Tmod, int MAXC, bool CODES, bool EMIT, bool TOKEN = false, bool X32 = false>
__global__ __launch_bounds__(kBlock, 1) FPQ_ADALN_OCC void adaln_rq16_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ out,
                                                           u32x4* __restrict__ h_out, u32x4* __restrict__ y_out,
                                                           int64_t rows, AdaLnArgs ad, RotArgs r, Lut16Args a, Lut16Tab tab,
                                                           int rows_per_wg, int wgs_per_batch) {
  __shared__ __attribute__((aligned(16))) uint16_t lut[kLutLdsEntries];   // static LDS: addresses known at compile time
  __shared__ u32x4 planes[4 * 64 * 5];                                      // 64 bytes per vector of the row (<= 320 vectors)
  constexpr bool MOD16 = sizeof(Tmod) == 2;
  constexpr int W = kBlock / 64;
  constexpr int RV = X32 ? 2 * MAXC : MAXC;      // 16-byte registers of one row per lane
  constexpr bool PREFETCH = X32 ? (FPQ_ADALN_PREFETCH32 != 0) : (FPQ_ADALN_PREFETCH != 0);
  const int vpr = (int)r.vec_per_row;            // host: (MAXC - 1) * 64 < vpr <= MAXC * 64
  FPQ_PHASE("workgroup_prologue");
  // four planes of vpr x 16 bytes: A[8v..8v+3], A[8v+4..8v+7], B[8v..8v+3], B[8v+4..8v+7]
  // (a lane reads 16 bytes of each plane at 16 * v: consecutive lanes, consecutive banks)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  u32x4* buf = nullptr;
  if constexpr (X32) {
    __shared__ u32x4 rowbuf[W][64 * MAXC];   // the modulated row, halves in, whole chunks out
    buf = rowbuf[wave];
  }
  const int64_t b = blockIdx.x / wgs_per_batch;
  const int chunk = blockIdx.x % wgs_per_batch;
  const int64_t L = ad.rows_per_batch;
  const int64_t row0 = b * L + (int64_t)chunk * rows_per_wg;
  int64_t n_here = L - (int64_t)chunk * rows_per_wg;
  if (n_here > rows_per_wg) n_here = rows_per_wg;
  if (row0 + n_here > rows) n_here = rows - row0;

  // vector c of this lane is vector c * 64 + lane of the row; only c = MAXC - 1 can lie outside
  const bool last_live = (MAXC - 1) * 64 + lane < vpr;
  int vidx[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) vidx[c] = c * 64 + lane;
  if (!last_live) vidx[MAXC - 1] = vpr - 1;      // any valid address: the value is zeroed, the store masked

  // X32: 4-float vector n * 64 + lane; only the last two can lie outside (a whole half-wave at a time: vpr % 16 == 0)
  int qidx[RV];
  bool qlive[RV];
#pragma unroll
  for (int n = 0; n < RV; ++n) {
    qidx[n] = n * 64 + lane;
    qlive[n] = !X32 || qidx[n] < 2 * vpr;
    if (!qlive[n]) qidx[n] = 2 * vpr - 1;
  }
  auto load_row = [&](u32x4 (&dst)[RV], int64_t row) {
    if constexpr (X32) {
      const u32x4* p = x + row * (2 * vpr);
#pragma unroll
      for (int n = 0; n < RV; ++n) dst[n] = __builtin_nontemporal_load(p + qidx[n]);
    } else {
      const u32x4* p = x + row * vpr;
#pragma unroll
      for (int c = 0; c < MAXC; ++c) dst[c] = __builtin_nontemporal_load(p + vidx[c]);
    }
  };
  u32x4 cur[RV];
#pragma unroll
  for (int c = 0; c < RV; ++c) cur[c] = u32x4{0, 0, 0, 0};
  if (wave < n_here) load_row(cur, row0 + wave);   // requested before the staging below

  // ---- stage the table and the folded modulation of batch entry b ----
  lut16_stage(lut, tab, a.shift);
  for (int v = threadIdx.x; v < vpr; v += kBlock) {
    const int64_t col = (int64_t)v * 8;
    float sc[8], sh[8];
    if constexpr (MOD16) {
      const u32x4 ws = *(const u32x4*)((const _Float16*)ad.scale + b * ad.cols + col);
      const u32x4 wh = *(const u32x4*)((const _Float16*)ad.shift + b * ad.cols + col);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t s1p = pk_add_f16(ws[k], 0x3C003C00u);   // scale.add(1) is an fp16 op in the reference
        sc[2 * k] = h2f(s1p & 0xFFFFu);
        sc[2 * k + 1] = h2f(s1p >> 16);
        sh[2 * k] = h2f(wh[k] & 0xFFFFu);
        sh[2 * k + 1] = h2f(wh[k] >> 16);
      }
    } else {
      const u32x4* ap = (const u32x4*)((const float*)ad.scale + b * ad.cols + col);
      const u32x4* bp = (const u32x4*)((const float*)ad.shift + b * ad.cols + col);
      const u32x4 a0 = ap[0], a1 = ap[1], b0 = bp[0], b1 = bp[1];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        sc[k] = u2f(a0[k]) + 1.0f;
        sc[4 + k] = u2f(a1[k]) + 1.0f;
        sh[k] = u2f(b0[k]);
        sh[4 + k] = u2f(b1[k]);
      }
    }
    if (r.smooth) {
      const u32x4* sp = (const u32x4*)(r.smooth + col);
      const u32x4 s0 = sp[0], s1 = sp[1];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        sc[k] *= u2f(s0[k]);
        sh[k] *= u2f(s0[k]);
        sc[4 + k] *= u2f(s1[k]);
        sh[4 + k] *= u2f(s1[k]);
      }
    }
    // the rotation's sign vector D rides on the modulation: half(-t) == -half(t), so h * D = half(fma(ln, A*D, B*D))
    const int j0 = (v * 8) & 127;
    const uint32_t dbits = (r.sign[j0 >> 5] >> (j0 & 31)) & 0xFFu;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t flip = ((dbits >> k) & 1u) << 31;
      sc[k] = u2f(fbits(sc[k]) ^ flip);
      sh[k] = u2f(fbits(sh[k]) ^ flip);
    }
    planes[v] = u32x4{fbits(sc[0]), fbits(sc[1]), fbits(sc[2]), fbits(sc[3])};
    planes[vpr + v] = u32x4{fbits(sc[4]), fbits(sc[5]), fbits(sc[6]), fbits(sc[7])};
    planes[2 * vpr + v] = u32x4{fbits(sh[0]), fbits(sh[1]), fbits(sh[2]), fbits(sh[3])};
    planes[3 * vpr + v] = u32x4{fbits(sh[4]), fbits(sh[5]), fbits(sh[6]), fbits(sh[7])};
  }
  __syncthreads();

  const int lg = lane & 15;
  const uint32_t sb = (r.sign[lg >> 2] >> ((lg & 3) * 8)) & 0xFFu;
  uint32_t sx[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) sx[k] = (((sb >> (2 * k)) & 1u) << 15) | (((sb >> (2 * k + 1)) & 1u) << 31);
  const float inv_c = 1.0f / (float)ad.cols;
  const h2v_t ones = {(_Float16)1.0f, (_Float16)1.0f};

  // One row: `cur` holds it, the wavefront's next row (if any) is requested into `nxt` first (software prefetch).
  // The row loop below alternates two register sets, so no row is ever copied from register to register.
  auto do_row = [&](u32x4 (&cur)[RV], u32x4 (&nxt)[RV], int i) {
    const int64_t row = row0 + i;
    FPQ_PHASE("prefetch_next_row");
    if constexpr (PREFETCH) {
      if (i + W < n_here) load_row(nxt, row + W);      // wave-uniform branch
    } else {
      (void)nxt;
      if (i != wave) load_row(cur, row);               // no prefetch: the row is requested when its turn comes
    }
    if constexpr (X32) {
#pragma unroll
      for (int n = (RV > 2 ? RV - 2 : 0); n < RV; ++n)
        if (!qlive[n]) cur[n] = u32x4{0, 0, 0, 0};
    } else {
      if (!last_live) cur[MAXC - 1] = u32x4{0, 0, 0, 0};
    }

    // ---- LayerNorm statistics: sum and sum of squares in one pass over the packed row (v_dot2_f32_f16: exact
    // products, fp32 accumulation; the zeroed padding vector adds nothing), var = E[x^2] - mean^2.  That
    // subtraction cancels when |mean| >> sigma: rows with mean^2 >= 64 var (6 of the 24 bits gone; also NaN / Inf
    // rows) take the centred second pass instead - wave-uniform branch, rare.
    // fp32 rows: plain sums, the squares are rounded: the centred pass already when mean^2 >= 8 var.
    FPQ_PHASE("ln_stats");
    float a1[RV], a2[RV];
#pragma unroll
    for (int c = 0; c < RV; ++c) a1[c] = a2[c] = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int c = 0; c < RV; ++c) {
        const uint32_t xw = cur[c][k];   // NOT __builtin_bit_cast(h2v_t, cur[c][k]): hipcc 7.2 then reads element 0 for every k
        if constexpr (X32) {
          const float xf = u2f(xw);
          a1[c] += xf;
          a2[c] = __builtin_fmaf(xf, xf, a2[c]);
        } else {
          const h2v_t xv = __builtin_bit_cast(h2v_t, xw);
          a1[c] = __builtin_amdgcn_fdot2(xv, ones, a1[c], false);
          a2[c] = __builtin_amdgcn_fdot2(xv, xv, a2[c], false);
        }
      }
    float s1 = a1[0], s2 = a2[0];
#pragma unroll
    for (int c = 1; c < RV; ++c) {
      s1 += a1[c];
      s2 += a2[c];
    }
    FPQ_PHASE("ln_reduce_rstd");
    wave_sum2_dpp(s1, s2);
    const float mean = s1 * inv_c;
    float var = __builtin_fmaf(-mean, mean, s2 * inv_c);
    if (!(mean * mean < (X32 ? 8.0f : 64.0f) * var)) {
#pragma unroll
      for (int c = 0; c < RV; ++c) a2[c] = 0.0f;
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < RV; ++c) {
          if constexpr (X32) {
            const float d0 = u2f(cur[c][k]) - mean;
            a2[c] = __builtin_fmaf(d0, d0, a2[c]);
          } else {
            const float d0 = fma_h_lo(cur[c][k], 1.0f, -mean), d1 = fma_h_hi(cur[c][k], 1.0f, -mean);
            a2[c] = __builtin_fmaf(d0, d0, a2[c]);
            a2[c] = __builtin_fmaf(d1, d1, a2[c]);
          }
        }
      if constexpr (X32) {                            // the zeroed padding is not part of the row
#pragma unroll
        for (int n = (RV > 2 ? RV - 2 : 0); n < RV; ++n)
          if (!qlive[n]) a2[n] = 0.0f;
      } else {
        if (!last_live) a2[MAXC - 1] = 0.0f;
      }
      s2 = a2[0];
#pragma unroll
      for (int c = 1; c < RV; ++c) s2 += a2[c];
      var = wave_sum_dpp(s2) * inv_c;
    }
    // rstd = 1 / sqrt(var + eps): v_rsq_f32 (1 ulp) + one Newton step, ~2^-23 relative - four instructions instead of
    // the IEEE sqrt and division sequences (~25); what it feeds is rounded to fp16
    const float ve = var + ad.eps;
    float rstd = __builtin_amdgcn_rsqf(ve);
    rstd = __builtin_fmaf(rstd * __builtin_fmaf(-ve * rstd, rstd, 1.0f), 0.5f, rstd);
    const float nm = -mean * rstd;

    // per-token operand output of ONE chunk per lane (the butterfly forms, and groups 16 .. 19 of the hybrid): E4M3 bytes
    // of the levels (fpq_gemm_fp8.h), 8 per vector, or dense 6-bit codes (fpq_gemm_fp6.h)
    auto token_codes_out = [&](int c, const u32x4& yv, const RowScale16& s, int v) {
        uint32_t cb[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t wk = yv[k];
          const uint32_t rb = div_pair16(wk, s.inv, s.inv_lo, s.inv, s.inv_lo);
          const uint32_t u = pk_sub_u16(rb, pk_lshr_u16(rb, 15));
          cb[2 * k] = lut[(u & 0xFFFFu) >> a.shift];
          cb[2 * k + 1] = lut[u >> (16 + a.shift)];
        }
        if (r.code_bits == 6) {
          // 8 six-bit codes = 48 bits per lane, rows packed densely: the four lanes of a quad own 24 contiguous
          // bytes; lane q of the quad takes the (3 - q) upper 16-bit words of its own string and the q + 1 lower
          // words of its right neighbour's, so that lanes 0..2 each store 8 aligned bytes (cols % 32 == 0: a quad
          // is live or dead as a whole).
          const uint64_t own = (uint64_t)(cb[0] | (cb[1] << 6) | (cb[2] << 12) | (cb[3] << 18)) |
                               ((uint64_t)(cb[4] | (cb[5] << 6) | (cb[6] << 12) | (cb[7] << 18)) << 24);
          const uint32_t nlo = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)own, 0xF9, 0xF, 0xF, false);   // quad_perm [1,2,3,3]
          const uint32_t nhi = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(own >> 32), 0xF9, 0xF, 0xF, false);
          const uint64_t nb = ((uint64_t)nhi << 32) | nlo;
          const int qp = lane & 3, sr = 16 * qp;
          const uint64_t w = (own >> sr) | (nb << (48 - sr));
          if (qp < 3) {
            uint8_t* dst = (uint8_t*)out + row * ((int64_t)vpr * 6) + (int64_t)c * (64 * 6) + 24 * (lane >> 2) + 8 * qp;
            __builtin_nontemporal_store(u32x2{(uint32_t)w, (uint32_t)(w >> 32)}, (u32x2*)dst);
          }
        } else {
          const u32x2 o2 = {cb[0] | (cb[1] << 8) | (cb[2] << 16) | (cb[3] << 24), cb[4] | (cb[5] << 8) | (cb[6] << 16) | (cb[7] << 24)};
          __builtin_nontemporal_store(o2, (u32x2*)out + row * vpr + v);
        }
    };
    FPQ_PHASE("modulate_to_image");
    if constexpr (X32) {
      // ---- fp32 rows: modulate this lane's half-chunks, 8 bytes of fp16 each, into LDS ----
      const int lane_x = rq_opaque(lane);
      const int hsel = lane_x & 1, k2 = lane_x >> 1;
      u32x2* img = (u32x2*)buf;
#pragma unroll
      for (int n = 0; n < RV; ++n) {
        const int v = 32 * n + k2;                       // chunk of the row
        u32x2 hw2 = {0, 0};
        if (n < RV) {
          const u32x4 A = planes[hsel * vpr + v], B = planes[(2 + hsel) * vpr + v];   // beyond the row: in bounds, unused
          const u32x4 w = cur[n];
          hw2[0] = f2h2(__builtin_fmaf(__builtin_fmaf(u2f(w[0]), rstd, nm), u2f(A[0]), u2f(B[0])), __builtin_fmaf(__builtin_fmaf(u2f(w[1]), rstd, nm), u2f(A[1]), u2f(B[1])));
          hw2[1] = f2h2(__builtin_fmaf(__builtin_fmaf(u2f(w[2]), rstd, nm), u2f(A[2]), u2f(B[2])), __builtin_fmaf(__builtin_fmaf(u2f(w[3]), rstd, nm), u2f(A[3]), u2f(B[3])));
          if (n >= RV - 2 && !qlive[n]) hw2 = u32x2{0, 0};
        }
        img[v * 2 + hsel] = hw2;
      }
      __builtin_amdgcn_wave_barrier();
    }
    {
    // ---- modulate, rotate, quantize: vectors two at a time, stage by stage ----
    u32x4 ys[TOKEN ? MAXC : 1];   // per-token scale: the rotated row waits here for the row maximum
    uint32_t mrow = 0;
    (void)ys;
    (void)mrow;
#pragma unroll
    for (int c0 = 0; c0 < MAXC; c0 += FPQ_ADALN_N2) {
      constexpr int N2 = FPQ_ADALN_N2;
      const int n = (MAXC - c0) < N2 ? (MAXC - c0) : N2;
      u32x4 hw[N2], y[N2], o[N2];
      float t[N2][8];
#pragma unroll
      for (int j = 0; j < n; ++j) {
        const int v = vidx[c0 + j];
        if constexpr (X32) {
          hw[j] = buf[v];                                // the modulated row, written by halves above
        } else {
          const u32x4 A0 = planes[v], B0 = planes[2 * vpr + v];
          const u32x4 A1 = planes[vpr + v], B1 = planes[3 * vpr + v];
          const u32x4 w = cur[c0 + j];
          hw[j][0] = f2h2(__builtin_fmaf(fma_h_lo(w[0], rstd, nm), u2f(A0[0]), u2f(B0[0])), __builtin_fmaf(fma_h_hi(w[0], rstd, nm), u2f(A0[1]), u2f(B0[1])));
          hw[j][1] = f2h2(__builtin_fmaf(fma_h_lo(w[1], rstd, nm), u2f(A0[2]), u2f(B0[2])), __builtin_fmaf(fma_h_hi(w[1], rstd, nm), u2f(A0[3]), u2f(B0[3])));
          hw[j][2] = f2h2(__builtin_fmaf(fma_h_lo(w[2], rstd, nm), u2f(A1[0]), u2f(B1[0])), __builtin_fmaf(fma_h_hi(w[2], rstd, nm), u2f(A1[1]), u2f(B1[1])));
          hw[j][3] = f2h2(__builtin_fmaf(fma_h_lo(w[3], rstd, nm), u2f(A1[2]), u2f(B1[2])), __builtin_fmaf(fma_h_hi(w[3], rstd, nm), u2f(A1[3]), u2f(B1[3])));
        }
      }
      if (!last_live && c0 + n == MAXC) hw[n - 1] = u32x4{0, 0, 0, 0};
      fwht128_h_n<N2>(hw, t, n, lg);                     // hw already carries the rotation's signs
#pragma unroll
      for (int j = 0; j < n; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) y[j][k] = mul2_to_h2(t[j][2 * k], t[j][2 * k + 1], r.c_h);
      uint32_t m[N2];
#pragma unroll
      for (int j = 0; j < n; ++j) m[j] = vec_absmax16(y[j]);
      if constexpr (TOKEN) {
#pragma unroll
        for (int j = 0; j < n; ++j) {
          const int c = c0 + j;
          ys[c] = y[j];
          mrow = mrow > m[j] ? mrow : m[j];
          if constexpr (EMIT) {
            if (c < MAXC - 1 || last_live) {
              const int64_t at = row * vpr + vidx[c];
              if (h_out) __builtin_nontemporal_store(u32x4{hw[j][0] ^ sx[0], hw[j][1] ^ sx[1], hw[j][2] ^ sx[2], hw[j][3] ^ sx[3]}, h_out + at);
              if (y_out) __builtin_nontemporal_store(y[j], y_out + at);
            }
          }
        }
        continue;
      }
      row_max_dpp16_n<N2>(m, n);
      uint32_t cd[N2];
      RowScale16 s[N2];
#pragma unroll
      for (int j = 0; j < n; ++j) s[j] = row_scale16(m[j], a.fpos.gmax, a.inv_gpos);
#pragma unroll
      for (int j = 0; j < n; ++j) {
        if constexpr (CODES) cd[j] = codes_vec16(y[j], lut, a.shift, s[j].inv, s[j].inv_lo);
        else o[j] = quant_vec16<false>(y[j], lut, a.shift, s[j].inv, s[j].inv_lo, s[j].s16x2, 0.f, 0.f, 0u);
      }
#pragma unroll
      for (int j = 0; j < n; ++j) {
        const int c = c0 + j;
        if (c < MAXC - 1 || last_live) {
          const int64_t at = row * vpr + vidx[c];
          if constexpr (EMIT) {
            if (h_out) __builtin_nontemporal_store(u32x4{hw[j][0] ^ sx[0], hw[j][1] ^ sx[1], hw[j][2] ^ sx[2], hw[j][3] ^ sx[3]}, h_out + at);
            if (y_out) __builtin_nontemporal_store(y[j], y_out + at);
          }
          if constexpr (CODES) {
            ((uint32_t*)out)[at] = cd[j];
            if (lg == 0) r.code_scales[at >> 4] = (uint16_t)(s[j].s16x2 & 0xFFFFu);
          } else {
            __builtin_nontemporal_store(o[j], out + at);
          }
        }
      }
    }
    if constexpr (TOKEN) {
      // fp6_quant_*_per_token_cuda on the rotated row (tr/quant_utils.py:503-534): one scale for the whole row
      mrow = row_max_dpp<64>(mrow);
      const RowScale16 s = row_scale16(mrow, a.fpos.gmax, a.inv_gpos);
      if (r.code_scales && lane == 0) r.code_scales[row] = (uint16_t)(s.s16x2 & 0xFFFFu);
#pragma unroll
      for (int c = 0; c < MAXC; ++c) {
        if (c == MAXC - 1 && !last_live) continue;
        const int v = vidx[c];
        if constexpr (CODES) {
          token_codes_out(c, ys[c], s, v);
        } else {
          const u32x4 o = quant_vec16<false>(ys[c], lut, a.shift, s.inv, s.inv_lo, s.s16x2, 0.f, 0.f, 0u);
          __builtin_nontemporal_store(o, out + row * vpr + v);
        }
      }
    }
    }
    if constexpr (X32) __builtin_amdgcn_wave_barrier();   // the row buffer is rewritten by the next row
  };
  u32x4 alt[RV];
  if constexpr (PREFETCH) {
    for (int i = wave; i < n_here; i += 2 * W) {   // no barrier below: wavefronts run their rows independently
      do_row(cur, alt, i);
      if (i + W < n_here) do_row(alt, cur, i + W);
    }
  } else {
    for (int i = wave; i < n_here; i += W) do_row(cur, alt, i);
  }
}
// ==================================================================================================================
// Third generation (round 3): the matrix-core form of the producer, rebuilt around the instruction census of the second
// one (profiles/r03_adaln_isa_census.txt: 585 vector instructions per row and wavefront at C = 1920 in the row loop +
// 62 of workgroup prologue spread over 4 rows; of the 585, ~110 were LDS / global address arithmetic, 16 the v_perm that
// packs two 16-bit table reads, 14 register copies between the two prefetch sets).
//   * every LDS access is a lane-constant base + an immediate: padded image rows (fpq_rotate_mfma.h), modulation planes
//     with a compile-time plane stride; the bases live in registers across the kernel (the LDS footprint allows 4
//     wavefronts per SIMD = 128 registers each);
//   * rows are loaded and stored through per-row buffer resources: no per-lane liveness - loads beyond the row return
//     zeros, stores beyond it are dropped, and the modulation of the padding is zero, so padding lanes need no selects;
//   * ONE register set per row: the next row is requested right after the current one has been modulated into the
//     operand image - its raw registers are dead from there on - and lands during the transform / quantize / store half
//     of the row; no duplicated loop body, 16 (fp16 rows) / 32 (fp32 rows) registers fewer;
//   * workgroup = (batch entry, chunk of rows), the chunks shrinking towards the end of the grid (AdalnTiers below).
//     Tried and dropped: ONE generation of 1024 resident workgroups, each
//     with a contiguous balanced share of ~64 rows and the modulation re-staged at batch boundaries - 105 us against 93
//     for chunks of 16 rows dispatched in address order (the resident wavefronts then sweep a compact window of the
//     tensor instead of 4096 streams a quarter of a megabyte apart), 2048 / 4096 shares: 100 / 93 us
//     (profiles/r03_adaln_partition.txt);
//   * E2M1 levels come from the FP4 conversion hardware (HW4): level = f16_fp4(fp4_f32(float(xn) + 2^-14)).  The bias
//     removes every tie (fp16 values >= 0.25 are multiples of 2^-12, every rounding boundary of E2M1 is a multiple of
//     0.25) and sends each to the side the reference's scan picks (the larger value); tools/probe/cvt_fp4_probe.hip checks all 63 488
//     finite fp16 values against the scan on the hardware (profiles/r03_cvt_fp4_probe.txt).  Levels that round to zero
//     from below come back as -0: the dequantizing multiply is an fma with +0.  Non-finite quotients only occur under a
//     non-finite scale, which is replaced by NaN (the reference's 0 * inf).  5 instructions per pair instead of 7, and
//     no table, no LDS traffic.
// Same results as the second generation on every path except that rows of 17 .. 20 groups on fp32 input form their
// slot first (the order of operations inside a row, not the values).
// ==================================================================================================================
// Workgroup -> rows.  A workgroup owns a chunk of rows of ONE batch entry (its modulation is staged once per workgroup).
// Long chunks amortise that prologue, but the launch ends with the chip draining for about half a workgroup's lifetime
// (the same kernel as a plain row copy: 81 us with 4 .. 8 rows per workgroup, 86 with 16, 94 with 32 -
// profiles/r03_adaln_partition.txt).  Hence chunks that shrink towards the end of the grid: the last batch entries in
// dispatch order are cut into chunks of rows[2], the ones before them into rows[1], the bulk into rows[0].
struct AdalnTiers {
  int rows[3];        // rows per workgroup in each tier
  int per_batch[3];   // workgroups per batch entry = ceil(L / rows)
  int batches[2];     // batch entries in tiers 0 and 1 (tier 2: the rest)
  uint32_t magic[3];  // ceil(2^32 / per_batch) when (id * magic) >> 32 == id / per_batch for every workgroup id of the tier
                      // (host-checked), else 0: the division then runs in the scalar unit instead of ~14 vector instructions
};

// 4 wavefronts per SIMD (128 registers) wherever the row fits: d30 on fp16 rows takes 96, on fp32 rows 118.  fp32 rows of
// 17 .. 20 groups hold 40 registers of raw row and get 168 (3 wavefronts); the emitting forms are for tests and
// calibration dumps.
// TIGHT (fp16 rows of exactly 15 groups = VAR-d30, hardware levels: no table): planes of 240 vectors + four 4096-byte
// images of 15 groups x 272 bytes = 31744 bytes of LDS - the most with which FIVE workgroups are placed on a CU (LDS is
// handed out in 1280-byte granules, profiles/r03_occupancy_census.txt).  From ~14 resident wavefronts per CU on the
// launch is throughput-bound (the stream itself: profiles/r03_adaln_ab.txt), so the fifth workgroup alone changes
// nothing; what it buys is that workgroups of 8 rows no longer cost throughput and drain faster at the end of the grid
// (profiles/r03_adaln_partition.txt, state C).  The padding lanes read past the planes (finite garbage, or the images
// behind them): their group 15 is dead weight in every phase, writes nothing into the image, and its stores are dropped.
#ifndef FPQ_ADALN_TIGHT
#define FPQ_ADALN_TIGHT 1
#endif
// -DFPQ_ADALN_STAMPS: diagnostic build (tools/adaln_stamps.py).  s_memtime stamps between the phases of a row, summed per
// wavefront in scalar registers and written - to `y_out`, which the non-emitting form never touches otherwise - once at
// the end: where a wavefront's row time goes.  Each stamp drains the LDS queue and the first one of a row the memory
// queue: read the SHARES, not the length.  No stamp executes in a regular build.
#ifdef FPQ_ADALN_STAMPS
#define FPQ_STAMP(k)                                                                  \
  do {                                                                                \
    unsigned long long t_;                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : : "memory");    \
    __builtin_amdgcn_sched_barrier(0);                                                \
    st_sum[k] += t_ - st_last;                                                        \
    st_last = t_;                                                                     \
  } while (0)
#else
#define FPQ_STAMP(k) do { } while (0)
#endif
#ifndef FPQ_ADALN_DB       // 1: two register sets, the next row is requested at the START of the current one (same-process A/B: 95.4 - 97.0 us against 91.9 - 97.6 for the single set, profiles/r03_adaln_ab.txt: not the default)
#define FPQ_ADALN_DB 0
#endif
// NW: wavefronts per workgroup.  (8 sharing one set of planes - the staging paid once per 8 wavefronts - was measured
// no faster than 4 and is not built any more, profiles/r03_adaln_partition.txt.)
// PAIR2 (rows of exactly 8 groups = C 1024, VAR-d16; instantiated with MAXC = 4): TWO consecutive rows of a batch entry
// fill the 16 group slots of one tile - chunks 0, 1 of the lane registers hold the first row, chunks 2, 3 the second
// (they are contiguous in memory: the loads, the image, the transform and the stores do not know) - with LayerNorm
// statistics and the modulation per half.  One row per tile leaves half of every epilogue instruction's lanes idle:
// 0.53 of 8 TB/s at [32768 x 1024] (profiles/r03_survey_shapes.txt).
// HW6 (1: E2M3, 2: E3M2; value output, rows of at most 16 groups): the levels of a lane's 32 outputs from the FP6 conversion
// hardware (fpq_fast16.h, fp6_levels_hw32) - per group and per token alike; no table is staged.
template <typename Tmod, int MAXC, bool CODES, bool EMIT, bool TOKEN, bool X32, bool HW4, bool TIGHT = false, int NW = 4, bool PAIR2 = false, int HW6 = 0>
__global__ __launch_bounds__(64 * NW, (EMIT || (X32 && MAXC == 5)) ? 3 : TIGHT ? 5 : 4) void adaln_mfma_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ out,
                                                              u32x4* __restrict__ h_out, u32x4* __restrict__ y_out,
                                                              int64_t rows, AdaLnArgs ad, RotArgs r, Lut16Args a,
                                                              Lut16Tab tab, AdalnTiers tiers) {
  static_assert(MAXC >= 1 && MAXC <= 5, "rows of at most 20 groups");
  static_assert(!HW4 || !TOKEN, "hardware E2M1 levels / codes: per group only");
  static_assert(HW6 == 0 || (!HW4 && !CODES && !PAIR2), "hardware FP6 levels: value outputs");
  static_assert(!TIGHT || (HW4 && !X32 && !EMIT && MAXC == 4), "the 31 KiB form: fp16 rows of 15 groups, no table");
  constexpr bool MOD16 = sizeof(Tmod) == 2;
  static_assert(NW == 4, "four wavefronts per workgroup");
  static_assert(!PAIR2 || (MAXC == 4 && !X32 && !EMIT && !TOKEN && !TIGHT), "two rows per tile: fp16 rows of 8 groups, per group");
  constexpr int W = NW;
  constexpr int RPU = PAIR2 ? 2 : 1;             // rows per unit of work of a wavefront
  constexpr int RV = X32 ? 2 * MAXC : MAXC;      // 16-byte registers of one row per lane
#ifdef FPQ_ADALN_PV_TEST   // timing experiment only (wrong results): a smaller LDS footprint
  constexpr int PV = FPQ_ADALN_PV_TEST;
#else
  constexpr int PV = TIGHT ? 240 : PAIR2 ? 128 : MAXC * 64;   // vectors per modulation plane (not TIGHT: the padding carries zeros)
#endif
  constexpr int INS = TIGHT ? kRqOutStride : kRqInStride;
  constexpr bool DB = FPQ_ADALN_DB && !X32 && !EMIT;
  uint16_t* lut = nullptr;                       // symmetric tables only: at most 2 x 512 buckets (E2M3)
  if constexpr (!HW4 && HW6 == 0) {
    __shared__ __attribute__((aligned(16))) uint16_t lut_s[1024];
    lut = lut_s;
  }
  __shared__ u32x4 planes[4][PV];                // A[8v..8v+3], A[8v+4..8v+7], B[8v..8v+3], B[8v+4..8v+7]
  // 16 groups x INS bytes per wavefront.  TIGHT: 15 groups x 272 = 4080 bytes in a 4096-byte slot, planes + images =
  // 31744 bytes: the most with which FIVE workgroups are resident on a CU (tools/probe/occupancy_census.hip,
  // profiles/r03_occupancy_census.txt: at 32256 - 32768 bytes the occupancy API still answers 5, the hardware places 4).
  // Group 15 does not exist in such a row: its lanes write nothing into the image (their slot is the next wavefront's
  // group 0), what the transform reads there is garbage confined to outputs the buffer range drops.
  __shared__ u32x4 images[W][TIGHT ? 256 : INS];
  const int vpr = (int)r.vec_per_row;            // host: (MAXC - 1) * 64 < vpr <= MAXC * 64
  FPQ_PHASE("workgroup_prologue");
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* img = (char*)images[wave];
  const char* pl = (const char*)planes;
  const RqLaneAddr la = rq_lane_addr<INS>(lane);
  const HadOperand ha = had_operand(lane);
  const int64_t L = ad.rows_per_batch;
  const int row_bytes = vpr * (X32 ? 32 : 16);

  // workgroup -> (batch entry, chunk): one 32-bit division (the 64-bit row / L of the first version of this kernel was
  // ~290 scalar instructions of software divide at the head of every workgroup)
  uint32_t id = blockIdx.x;
  const uint32_t n0 = (uint32_t)tiers.batches[0] * (uint32_t)tiers.per_batch[0];
  const uint32_t n1 = (uint32_t)tiers.batches[1] * (uint32_t)tiers.per_batch[1];
  int tier = 0;
  uint32_t b0 = 0;
  if (id >= n0 + n1) {
    tier = 2;
    id -= n0 + n1;
    b0 = (uint32_t)(tiers.batches[0] + tiers.batches[1]);
  } else if (id >= n0) {
    tier = 1;
    id -= n0;
    b0 = (uint32_t)tiers.batches[0];
  }
  const uint32_t pb = (uint32_t)tiers.per_batch[tier];
  const int rows_per_wg = tiers.rows[tier];
  const uint32_t mg = tiers.magic[tier];
  const uint32_t bq = mg ? (uint32_t)(((uint64_t)id * mg) >> 32) : id / pb;
  const int64_t b = b0 + bq;
  const int64_t lo = b * L + (int64_t)(id - bq * pb) * rows_per_wg;
  int64_t hi = lo + rows_per_wg;
  if (hi > (b + 1) * L) hi = (b + 1) * L;
  if (hi > rows) hi = rows;

  // Rows of 17 / 18 groups (d36: C = 2304): the one or two groups beyond the tile fill at most half of the 64-lane slot
  // that transforms them as butterflies.  A wavefront therefore pairs its rows: the first row of a pair parks its
  // modulated slot chunk (lanes 0 .. 31), the second row is loaded and modulated with its slot chunk on lanes 32 .. 63
  // (`hi_half`), and ONE slot pass - butterfly, maximum, scale, quantize, store - serves both (about 200 of a row's
  // ~700 vector instructions belong to that pass).  Not for the emitting form (tests), the per-token forms (the slot
  // enters the row's scale), fp32 rows (their slot chunk meets in the image).
  constexpr bool PAIRABLE = MAXC == 5 && !X32 && !EMIT && !TOKEN && !DB;
  const bool pair_ok = PAIRABLE && vpr - 256 <= 32;
  const int lane16_hi = ((lane + 32) & 63) * 16;     // the slot chunk's vector index of this lane in a `hi_half` row
  auto load_row = [&](u32x4 (&dst)[RV], int64_t row, bool hi_half = false) {
#ifdef FPQ_ADALN_NOMEM   // timing experiment: zero-record descriptors - every row load returns zeros, every row store is dropped,
                         // the instruction stream is unchanged: what the kernel costs without its HBM traffic
    const __amdgpu_buffer_rsrc_t src = rq_rsrc((const char*)x + row * row_bytes, 0);
#else
    const __amdgpu_buffer_rsrc_t src = rq_rsrc((const char*)x + row * row_bytes, (PAIR2 && row + 1 < hi ? 2 : 1) * row_bytes);
#endif
#pragma unroll
    for (int n = 0; n < RV; ++n) {
      int at = la.lane16 + n * 1024;
      if (PAIRABLE && n == 4 && hi_half) at = lane16_hi + 4096;   // lanes 0 .. 31 then point beyond the row: zeros
      dst[n] = __builtin_amdgcn_raw_buffer_load_b128(src, at, 0, kRqNt);
    }
  };

  // the rotation's signs of this lane's chunk (16-byte vector: chunk lane % 16; fp32 rows: half lane & 1 of chunk
  // (lane / 2) % 16) - only to take them off again for the h_out of the emitting form
  uint32_t sx[4] = {0, 0, 0, 0};
  if constexpr (EMIT || MAXC == 5 || MOD16) {   // (MOD16: the staging below puts them on the packed modulation words)
    const int lg = lane & 15;
    const uint32_t sb = (r.sign[lg >> 2] >> ((lg & 3) * 8)) & 0xFFu;
#pragma unroll
    for (int k = 0; k < 4; ++k) sx[k] = (((sb >> (2 * k)) & 1u) << 15) | (((sb >> (2 * k + 1)) & 1u) << 31);
  }
  uint32_t sx2[2] = {0, 0};
  if constexpr (EMIT && X32) {
    const int j0 = ((lane >> 1) & 15) * 8 + 4 * (lane & 1);
    const uint32_t db = (r.sign[j0 >> 5] >> (j0 & 31)) & 0xFu;
    sx2[0] = ((db & 1u) << 15) | (((db >> 1) & 1u) << 31);
    sx2[1] = (((db >> 2) & 1u) << 15) | (((db >> 3) & 1u) << 31);
  }
  const int pl_x = (lane & 1) * (PV * 16) + (lane >> 1) * 16;   // fp32 rows: plane `half`, vector lane / 2 (+ 32 n)
  const float inv_c = 1.0f / (float)ad.cols;
  const h2v_t ones = {(_Float16)1.0f, (_Float16)1.0f};

  // ---- the folded modulation of batch entry b:  A = half(scale + 1) * s * D,  B = shift * s * D;  zeros beyond the row ----
  // Two steps: the loads (raw words into registers), then arithmetic + LDS writes.  The loads are issued BEFORE the
  // wavefront's first row is requested and for all of a thread's plane vectors at once (round 4).  Until then the row came
  // first and the staging loop loaded, waited and stored one vector per trip: a wavefront's loads return in order and a
  // CU's L1 works its misses off in order, so the modulation words (L2 hits after the first workgroup of a batch entry)
  // sat behind every row request the CU had issued, and rows of 17 - 20 groups (320 plane vectors: two trips of a
  // 256-thread workgroup) paid the latency twice - tools/adaln_stamps.py: at [20 x 324 x 2304] fp32 no wavefront had its
  // first row before 4.7 us after its start (2.4 us at [100 x 64 x 1920]).
  constexpr int NST = (PV + 64 * NW - 1) / (64 * NW);   // plane vectors per thread
  struct StageRaw {
    u32x4 w[MOD16 ? 2 : 4];   // scale, shift (fp16: one vector each; fp32: two)
    u32x4 s[2];               // smoothing factors
  };
  auto stage_load = [&](int64_t b, StageRaw (&raw)[NST]) {
#pragma unroll
    for (int t = 0; t < NST; ++t) {
      const int v = threadIdx.x + t * 64 * NW;
      if (v < PV && (TIGHT || v < vpr)) {
        const int64_t col = (int64_t)v * 8;
        if constexpr (MOD16) {
          raw[t].w[0] = *(const u32x4*)((const _Float16*)ad.scale + b * ad.cols + col);
          raw[t].w[1] = *(const u32x4*)((const _Float16*)ad.shift + b * ad.cols + col);
        } else {
          const u32x4* ap = (const u32x4*)((const float*)ad.scale + b * ad.cols + col);
          const u32x4* bp = (const u32x4*)((const float*)ad.shift + b * ad.cols + col);
          raw[t].w[0] = ap[0];
          raw[t].w[1] = ap[1];
          raw[t].w[2] = bp[0];
          raw[t].w[3] = bp[1];
        }
        if (r.smooth) {
          const u32x4* sp = (const u32x4*)(r.smooth + col);
          raw[t].s[0] = sp[0];
          raw[t].s[1] = sp[1];
        }
      }
    }
  };
  auto stage_store = [&](const StageRaw (&raw)[NST]) {
#pragma unroll
    for (int t = 0; t < NST; ++t) {
      const int v = threadIdx.x + t * 64 * NW;
      if (v >= PV) break;
      float sc[8], sh[8];
      if (TIGHT || v < vpr) {
        float sm[8];
        if (r.smooth) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            sm[k] = u2f(raw[t].s[0][k]);
            sm[4 + k] = u2f(raw[t].s[1][k]);
          }
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) sm[k] = 1.0f;
        }
        if constexpr (MOD16) {
          // fp16 modulation: the sign vector D goes onto the PACKED words (one xor per pair; vector v of a row carries
          // the signs of chunk v % 16 = lane % 16: the lane constants sx), and one v_fma_mix_f32 per element widens and
          // applies the smoothing factor (h * s - 0 == h * s, signed zeros included) - about half the vector
          // instructions of convert, multiply, per-element sign flip, in a prologue every wavefront pays per two rows
          const u32x4 ws = raw[t].w[0], wh = raw[t].w[1];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const uint32_t s1p = pk_add_f16(ws[k], 0x3C003C00u) ^ sx[k];   // scale.add(1) is an fp16 op in the reference
            const uint32_t shp = wh[k] ^ sx[k];
            sc[2 * k] = fma_h_lo(s1p, sm[2 * k], -0.0f);
            sc[2 * k + 1] = fma_h_hi(s1p, sm[2 * k + 1], -0.0f);
            sh[2 * k] = fma_h_lo(shp, sm[2 * k], -0.0f);
            sh[2 * k + 1] = fma_h_hi(shp, sm[2 * k + 1], -0.0f);
          }
        } else {
          const u32x4 a0 = raw[t].w[0], a1 = raw[t].w[1], b0 = raw[t].w[2], b1 = raw[t].w[3];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            sc[k] = u2f(a0[k]) + 1.0f;
            sc[4 + k] = u2f(a1[k]) + 1.0f;
            sh[k] = u2f(b0[k]);
            sh[4 + k] = u2f(b1[k]);
          }
          if (r.smooth) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              sc[k] *= sm[k];
              sh[k] *= sm[k];
            }
          }
          // the rotation's sign vector D rides on the modulation: half(-t) == -half(t), so h * D = half(fma(ln, A*D, B*D))
          const int j0 = (v * 8) & 127;
          const uint32_t dbits = (r.sign[j0 >> 5] >> (j0 & 31)) & 0xFFu;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const uint32_t flip = ((dbits >> k) & 1u) << 31;
            sc[k] = u2f(fbits(sc[k]) ^ flip);
            sh[k] = u2f(fbits(sh[k]) ^ flip);
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) sc[k] = sh[k] = 0.0f;
      }
      planes[0][v] = u32x4{fbits(sc[0]), fbits(sc[1]), fbits(sc[2]), fbits(sc[3])};
      planes[1][v] = u32x4{fbits(sc[4]), fbits(sc[5]), fbits(sc[6]), fbits(sc[7])};
      planes[2][v] = u32x4{fbits(sh[0]), fbits(sh[1]), fbits(sh[2]), fbits(sh[3])};
      planes[3][v] = u32x4{fbits(sh[4]), fbits(sh[5]), fbits(sh[6]), fbits(sh[7])};
    }
  };

  // One row.  `cur` holds it; `next_row` >= 0: that row is requested - DB: into `nxt`, at once; otherwise into `cur`,
  // once the current row has left it.
#ifdef FPQ_ADALN_STAMPS
  unsigned long long st_sum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = 0, st_rows = 0, st_first = 0;
  const unsigned long long st_t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz, one clock for the whole chip
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last) : : "memory");   // phase 0 of the first row = the prologue
#endif
  u32x4 pend = {0, 0, 0, 0};      // PAIRABLE: the parked slot chunk of the pair's first row, and that row
  int64_t pend_row = 0;
  bool cur_hi = false;           // the row in `cur` was loaded with its slot chunk on lanes 32 .. 63
  auto do_row = [&](u32x4 (&cur)[RV], u32x4 (&nxt)[RV], int64_t row, int64_t next_row) {
    const int nrows = (PAIR2 && row + 1 < hi) ? 2 : 1;                 // PAIR2: rows in this unit (a batch entry of odd length ends with one)
    const bool hi_half = PAIRABLE && cur_hi;                           // second row of a pair
    const bool park = PAIRABLE && pair_ok && !hi_half && next_row >= 0;   // first row of a pair: its slot waits for the next row
    FPQ_STAMP(0);                                   // between rows (loop control; the first row: the prologue)
#ifdef FPQ_ADALN_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (++st_rows == 1) st_first = __builtin_amdgcn_s_memrealtime();   // the wavefront's first row has arrived
#endif
    FPQ_STAMP(1);                                   // waiting for the row (and, in this build, the previous row's stores)
#ifdef FPQ_ADALN_COPYONLY   // experiment: the kernel's memory access pattern alone (rows in, rows out, nothing computed; 2: no staging either)
    if constexpr (!X32 && !EMIT && !CODES && !TOKEN) {
      const __amdgpu_buffer_rsrc_t dst = rq_rsrc(out + row * vpr, vpr * 16);
      u32x4 keep[RV];
#pragma unroll
      for (int n = 0; n < RV; ++n) keep[n] = cur[n];
      if (next_row >= 0) load_row(DB ? nxt : cur, next_row);
#pragma unroll
      for (int n = 0; n < RV; ++n) __builtin_amdgcn_raw_buffer_store_b128(keep[n], dst, la.lane16 + n * 1024, 0, kRqNt);
      return;
    }
#endif
    if constexpr (DB) {
      FPQ_PHASE("prefetch_next_row");
      if (next_row >= 0) load_row(nxt, next_row);   // wave-uniform
    } else {
      (void)nxt;
    }
    // ---- LayerNorm statistics (see the second generation above; padding lanes hold zeros) ----
    FPQ_PHASE("ln_stats");
    float a1[RV], a2[RV];
#pragma unroll
    for (int c = 0; c < RV; ++c) a1[c] = a2[c] = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int c = 0; c < RV; ++c) {
        const uint32_t xw = cur[c][k];
        if constexpr (X32) {
          const float xf = u2f(xw);
          a1[c] += xf;
          a2[c] = __builtin_fmaf(xf, xf, a2[c]);
        } else {
          const h2v_t xv = __builtin_bit_cast(h2v_t, xw);
          a1[c] = __builtin_amdgcn_fdot2(xv, ones, a1[c], false);
          a2[c] = __builtin_amdgcn_fdot2(xv, xv, a2[c], false);
        }
      }
    float sum1 = a1[0], s2 = a2[0];
    float sum1b = 0.0f, s2b = 0.0f;                  // PAIR2: the second row's sums (chunks 2, 3)
    if constexpr (PAIR2) {
      sum1 += a1[1];
      s2 += a2[1];
      sum1b = a1[2] + a1[3];
      s2b = a2[2] + a2[3];
    } else {
#pragma unroll
      for (int c = 1; c < RV; ++c) {
        sum1 += a1[c];
        s2 += a2[c];
      }
    }
    FPQ_PHASE("ln_reduce_rstd");
#ifdef FPQ_ADALN_ABLATE_STATS   // measurement only (wrong results): what the kernel would cost with the row statistics given
    sum1 = 0.0f;
    s2 = (float)ad.cols;
    sum1b = 0.0f;
    s2b = (float)ad.cols;
#else
    wave_sum2_dpp(sum1, s2);
#endif
    const float mean = sum1 * inv_c;
    float var = __builtin_fmaf(-mean, mean, s2 * inv_c);
    float mean_b = 0.0f, var_b = 1.0f;
    if constexpr (PAIR2) {
#ifndef FPQ_ADALN_ABLATE_STATS
      wave_sum2_dpp(sum1b, s2b);
#endif
      mean_b = sum1b * inv_c;
      var_b = __builtin_fmaf(-mean_b, mean_b, s2b * inv_c);
      if (!(mean * mean < 64.0f * var) || !(mean_b * mean_b < 64.0f * var_b)) {   // either row: both centred (rare)
        float ca = 0.0f, cb = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float mu = c < 2 ? mean : mean_b;
            const float d0 = fma_h_lo(cur[c][k], 1.0f, -mu), d1 = fma_h_hi(cur[c][k], 1.0f, -mu);
            if (c < 2) ca = __builtin_fmaf(d1, d1, __builtin_fmaf(d0, d0, ca));
            else cb = __builtin_fmaf(d1, d1, __builtin_fmaf(d0, d0, cb));
          }
        var = wave_sum_dpp(ca) * inv_c;
        var_b = wave_sum_dpp(cb) * inv_c;
      }
    }
    if (!PAIR2 && !(mean * mean < (X32 ? 8.0f : 64.0f) * var)) {   // cancellation (or NaN / Inf): the centred second pass - rare
#pragma unroll
      for (int c = 0; c < RV; ++c) a2[c] = 0.0f;
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < RV; ++c) {
          if constexpr (X32) {
            const float d0 = u2f(cur[c][k]) - mean;
            a2[c] = __builtin_fmaf(d0, d0, a2[c]);
          } else {
            const float d0 = fma_h_lo(cur[c][k], 1.0f, -mean), d1 = fma_h_hi(cur[c][k], 1.0f, -mean);
            a2[c] = __builtin_fmaf(d0, d0, a2[c]);
            a2[c] = __builtin_fmaf(d1, d1, a2[c]);
          }
        }
#pragma unroll
      for (int c = 0; c < RV; ++c) {                      // the zero padding is not part of the row
        const int vl = (PAIRABLE && c == 4 && hi_half) ? ((lane + 32) & 63) : lane;
        if ((c * 64 + vl) * (X32 ? 4 : 8) >= (int)ad.cols) a2[c] = 0.0f;
      }
      s2 = a2[0];
#pragma unroll
      for (int c = 1; c < RV; ++c) s2 += a2[c];
      var = wave_sum_dpp(s2) * inv_c;
    }
    const float ve = var + ad.eps;
    float rstd = __builtin_amdgcn_rsqf(ve);
    rstd = __builtin_fmaf(rstd * __builtin_fmaf(-ve * rstd, rstd, 1.0f), 0.5f, rstd);
    const float nm = -mean * rstd;
    float rstd_b = 0.0f, nm_b = 0.0f;
    if constexpr (PAIR2) {
      const float vb = var_b + ad.eps;
      rstd_b = __builtin_amdgcn_rsqf(vb);
      rstd_b = __builtin_fmaf(rstd_b * __builtin_fmaf(-vb * rstd_b, rstd_b, 1.0f), 0.5f, rstd_b);
      nm_b = -mean_b * rstd_b;
    }
    FPQ_STAMP(2);                                   // statistics, reduction, rstd

    // ---- modulate into the operand image: h * D = half(fma(fma(x, rstd, nm), A, B)) ----
    FPQ_PHASE("modulate_to_image");
    const __amdgpu_buffer_rsrc_t h_dst = rq_rsrc(EMIT && h_out ? (const char*)(h_out + row * vpr) : nullptr, EMIT && h_out ? vpr * 16 : 0);
    const float rstd_a = rstd, nm_a = nm;
    u32x4 hw_slot = {0, 0, 0, 0};   // MAXC == 5: chunk 256 + lane of the row (groups 16 .. 19), transformed as butterflies
    if constexpr (X32) {
      auto half_chunk = [&](int n) {
        const u32x4 A = *(const u32x4*)(pl + pl_x + n * 512), B = *(const u32x4*)(pl + pl_x + n * 512 + 2 * PV * 16);
        const u32x4 w = cur[n];
        u32x2 hw2;
        hw2[0] = f2h2(__builtin_fmaf(__builtin_fmaf(u2f(w[0]), rstd, nm), u2f(A[0]), u2f(B[0])), __builtin_fmaf(__builtin_fmaf(u2f(w[1]), rstd, nm), u2f(A[1]), u2f(B[1])));
        hw2[1] = f2h2(__builtin_fmaf(__builtin_fmaf(u2f(w[2]), rstd, nm), u2f(A[2]), u2f(B[2])), __builtin_fmaf(__builtin_fmaf(u2f(w[3]), rstd, nm), u2f(A[3]), u2f(B[3])));
        if constexpr (EMIT)   // half lane & 1 of chunk 32 n + lane / 2 = bytes lane * 8 + n * 512 of the row
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{hw2[0] ^ sx2[0], hw2[1] ^ sx2[1]}, h_dst, lane * 8 + n * 512, 0, kRqNt);
        return hw2;
      };
      if constexpr (MAXC == 5) {   // the slot first: its halves meet in the (still free) image, one chunk per lane out
#pragma unroll
        for (int n = 8; n < RV; ++n) *(u32x2*)(img + lane * 8 + (n - 8) * 512) = half_chunk(n);
        __builtin_amdgcn_wave_barrier();
        hw_slot = *(const u32x4*)(img + la.lane16);
        __builtin_amdgcn_wave_barrier();
      }
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        u32x2 hw2 = {0, 0};
        if (n < RV) hw2 = half_chunk(n);
        *(u32x2*)(img + la.in_w8 + n * (2 * INS)) = hw2;
      }
    } else {
      auto chunk = [&](int c) {
        const int l16 = (PAIRABLE && c == 4 && hi_half) ? lane16_hi : la.lane16;
        const int pc = PAIR2 ? (c & 1) : c;                       // plane chunk: PAIR2 - both rows share the batch entry's modulation
        const float rstd = (PAIR2 && c >= 2) ? rstd_b : rstd_a, nm = (PAIR2 && c >= 2) ? nm_b : nm_a;
        const u32x4 A0 = *(const u32x4*)(pl + l16 + (0 * PV + pc * 64) * 16), A1 = *(const u32x4*)(pl + l16 + (1 * PV + pc * 64) * 16);
        const u32x4 B0 = *(const u32x4*)(pl + l16 + (2 * PV + pc * 64) * 16), B1 = *(const u32x4*)(pl + l16 + (3 * PV + pc * 64) * 16);
        const u32x4 w = cur[c];
        u32x4 hw;
        hw[0] = f2h2(__builtin_fmaf(fma_h_lo(w[0], rstd, nm), u2f(A0[0]), u2f(B0[0])), __builtin_fmaf(fma_h_hi(w[0], rstd, nm), u2f(A0[1]), u2f(B0[1])));
        hw[1] = f2h2(__builtin_fmaf(fma_h_lo(w[1], rstd, nm), u2f(A0[2]), u2f(B0[2])), __builtin_fmaf(fma_h_hi(w[1], rstd, nm), u2f(A0[3]), u2f(B0[3])));
        hw[2] = f2h2(__builtin_fmaf(fma_h_lo(w[2], rstd, nm), u2f(A1[0]), u2f(B1[0])), __builtin_fmaf(fma_h_hi(w[2], rstd, nm), u2f(A1[1]), u2f(B1[1])));
        hw[3] = f2h2(__builtin_fmaf(fma_h_lo(w[3], rstd, nm), u2f(A1[2]), u2f(B1[2])), __builtin_fmaf(fma_h_hi(w[3], rstd, nm), u2f(A1[3]), u2f(B1[3])));
        if constexpr (EMIT)
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{hw[0] ^ sx[0], hw[1] ^ sx[1], hw[2] ^ sx[2], hw[3] ^ sx[3]}, h_dst,
                                                 la.lane16 + c * 1024, 0, kRqNt);
        return hw;
      };
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        u32x4 hw = {0, 0, 0, 0};
        if (c < MAXC) hw = chunk(c);
        if (!TIGHT || c < 3 || lane < 48) *(u32x4*)(img + la.in_w + c * (4 * INS)) = hw;
      }
      if constexpr (MAXC == 5) hw_slot = chunk(4);
    }
    // the slot pass of this row: skipped when the chunk is parked; both halves of the lanes when it closes a pair
    bool do_slot = MAXC == 5, paired = false;
    if constexpr (PAIRABLE) {
      if (park) {
        pend = hw_slot;
        pend_row = row;
        do_slot = false;
      } else if (hi_half) {
        paired = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) hw_slot[k] = lane < 32 ? pend[k] : hw_slot[k];
      }
    }
    if constexpr (!DB) {
      FPQ_PHASE("prefetch_next_row");
      if (next_row >= 0) load_row(cur, next_row, park);   // wave-uniform; lands under the rest of this row
    }
    if constexpr (PAIRABLE) cur_hi = park;
    __builtin_amdgcn_wave_barrier();
    FPQ_STAMP(3);                                   // modulate (plane reads), image writes, next row requested

    // ---- transform on the matrix cores; this lane then holds 32 outputs of group lane % 16 ----
    uint32_t yw[8][2];
    const float mf = hadamard128_mfma(img, la.in_r, ha, r.c_h, yw);
    __builtin_amdgcn_wave_barrier();
    FPQ_STAMP(4);                                   // operand reads, 8 MFMAs, 4-point butterfly, maximum, c_h rounding
    if constexpr (EMIT) {
      if (y_out) rq_store_tile(img, yw, rq_rsrc(y_out + row * vpr, vpr * 16), la);
    }
    FPQ_PHASE("group_max_scale");
    uint32_t m = mul2_to_h2(mf, 0.0f, r.c_h) & 0xFFFFu;       // see rotate_quant_mfma_kernel
    if (__builtin_expect((yw[0][0] & 0x7C00u) == 0x7C00u, 0)) {
      m = 0;
#pragma unroll
      for (int c = 0; c < 8; ++c) m = pk_max_u16(m, pk_max_u16(yw[c][0] & 0x7FFF7FFFu, yw[c][1] & 0x7FFF7FFFu));
      const uint32_t lo16 = m & 0xFFFFu, hi16 = m >> 16;
      m = lo16 > hi16 ? lo16 : hi16;
    }
    // ---- groups 16 .. 19 (d36: C = 2304 = 18 groups): one chunk per lane, the transform as butterflies (a second tile
    // would run its epilogue for 64 lanes to serve 8 - 16 of them); quantized after the tile ----
    u32x4 y1 = {0, 0, 0, 0};
    uint32_t m1 = 0;
    if (MAXC == 5 && do_slot) {   // wave-uniform (always true unless this row's chunk is parked for its pair)
      const u32x4 hwa[1] = {hw_slot};
      float t1[1][8];
      fwht128_h_n<1>(hwa, t1, 1, lane & 15);
#pragma unroll
      for (int k = 0; k < 4; ++k) y1[k] = mul2_to_h2(t1[0][2 * k], t1[0][2 * k + 1], r.c_h);
      m1 = vec_absmax16(y1);
    }
    RowScale16 s, s1;
    if constexpr (TOKEN) {
      m = row_max_dpp<64>(m > m1 ? m : m1);   // fp6_quant_*_per_token_cuda on the rotated row: one scale for the whole row
      s = s1 = row_scale16(m, a.fpos.gmax, a.inv_gpos);
      if (r.code_scales && lane == 0) r.code_scales[row] = (uint16_t)(s.s16x2 & 0xFFFFu);
    } else {
      auto sw = __builtin_amdgcn_permlane16_swap(m, m, false, false);
      m = sw[0] > sw[1] ? sw[0] : sw[1];
      sw = __builtin_amdgcn_permlane32_swap(m, m, false, false);
      m = sw[0] > sw[1] ? sw[0] : sw[1];
      s = s1 = row_scale16(m, a.fpos.gmax, a.inv_gpos);
      if (MAXC == 5 && do_slot) {
        uint32_t ms[1] = {m1};
        row_max_dpp16_n<1>(ms, 1);
        s1 = row_scale16(ms[0], a.fpos.gmax, a.inv_gpos);
      }
      if constexpr (HW4) {
        scale_nan_if_not_finite(s);
        if constexpr (MAXC == 5) scale_nan_if_not_finite(s1);
      }
    }
    if constexpr (HW6 != 0) {
      scale_nan_if_not_finite(s);
      if constexpr (MAXC == 5) scale_nan_if_not_finite(s1);
    }
    FPQ_STAMP(5);                                   // group maximum across the quarters, scale and its reciprocal
    // MAXC == 5: this lane's chunk of groups 16 .. 19 - of this row, or (closing a pair) lanes 0 .. 31 the parked row's
    // and lanes 32 .. 63 this row's
    const int slot_idx = paired ? (lane & 31) : lane;
    const int64_t slot_row = (paired && lane < 32) ? pend_row : row;
    const int64_t slot_at = slot_row * vpr + 256 + slot_idx;
    const bool slot_live = do_slot && 256 + slot_idx < vpr;
    if constexpr (CODES && TOKEN) {   // per-token operands: E4M3 bytes or dense 6-bit codes, the row scale is out already
      if (r.code_bits == 6) {
        // row-major: the row's 6 vpr bytes behind its start; k-major (include/fpq.h): chunk ch of the row = chunk ch % 6 of K step ch / 6
        const uint32_t steps = (uint32_t)vpr >> 4;
        const __amdgpu_buffer_rsrc_t d6 = r.km_rows ? rq_rsrc(out, (int)(r.km_rows * steps * 96u)) : rq_rsrc((const uint8_t*)out + row * ((int64_t)vpr * 6), vpr * 6);
        rq_store_codes6((u32x4*)img, yw, s, lut, a.shift, d6, lane, [&](int ch) -> uint32_t {
          if (!r.km_rows) return (uint32_t)ch * 16u;
          const uint32_t st = (uint32_t)ch / 6u;
          return st < steps ? km6_off((uint32_t)row, st, (uint32_t)ch - 6u * st, r.km_rows) : 0xFFFFFFFFu;
        });
      }
      else rq_store_codes8((u32x4*)img, yw, s, lut, a.shift, rq_rsrc((const uint8_t*)out + row * ((int64_t)vpr * 8), vpr * 8), lane);
      if constexpr (MAXC == 5) {
        if (slot_live) {
          uint32_t cb[8];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const uint32_t rb = div_pair16(y1[k], s1.inv, s1.inv_lo, s1.inv, s1.inv_lo);
            const uint32_t u = pk_sub_u16(rb, pk_lshr_u16(rb, 15));
            cb[2 * k] = lut[(u & 0xFFFFu) >> a.shift];
            cb[2 * k + 1] = lut[u >> (16 + a.shift)];
          }
          if (r.code_bits == 6) {
            // 8 six-bit codes = 48 bits per lane, rows packed densely: the four lanes of a quad own 24 contiguous bytes;
            // lane q of the quad takes the (3 - q) upper 16-bit words of its own string and the q + 1 lower words of its
            // right neighbour's, so that lanes 0..2 each store 8 aligned bytes (cols % 32 == 0: a quad is live or dead
            // as a whole)
            const uint64_t own = (uint64_t)(cb[0] | (cb[1] << 6) | (cb[2] << 12) | (cb[3] << 18)) |
                                 ((uint64_t)(cb[4] | (cb[5] << 6) | (cb[6] << 12) | (cb[7] << 18)) << 24);
            const uint32_t nlo = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)own, 0xF9, 0xF, 0xF, false);   // quad_perm [1,2,3,3]
            const uint32_t nhi = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(own >> 32), 0xF9, 0xF, 0xF, false);
            const uint64_t nb = ((uint64_t)nhi << 32) | nlo;
            const int qp = lane & 3, sr = 16 * qp;
            const uint64_t w6 = (own >> sr) | (nb << (48 - sr));
            if (qp < 3) {
              uint8_t* dst = (uint8_t*)out + row * ((int64_t)vpr * 6) + (int64_t)4 * (64 * 6) + 24 * (lane >> 2) + 8 * qp;
              if (r.km_rows) {   // the same 8 bytes of the row, at byte wb: K step wb / 96, chunk (wb % 96) / 16 of the k-major image
                const uint32_t wb = 1536u + 24u * ((uint32_t)lane >> 2) + 8u * (uint32_t)qp, w = wb % 96u;
                dst = (uint8_t*)out + km6_off((uint32_t)row, wb / 96u, w >> 4, r.km_rows) + (w & 15u);
              }
              __builtin_nontemporal_store(u32x2{(uint32_t)w6, (uint32_t)(w6 >> 32)}, (u32x2*)dst);
            }
          } else {
            const u32x2 o2 = {cb[0] | (cb[1] << 8) | (cb[2] << 16) | (cb[3] << 24), cb[4] | (cb[5] << 8) | (cb[6] << 16) | (cb[7] << 24)};
            __builtin_nontemporal_store(o2, (u32x2*)out + slot_at);
          }
        }
      }
    } else if constexpr (CODES) {   // FP4 operands: codes + one fp16 scale per group (fpq_gemm_fp4.h)
      // row-major: the unit's groups (one row's, or two short rows' with PAIR2) are contiguous behind `row`; k-major: group
      // u = lane / 4 of the unit is group u % gpr of row + u / gpr, the range check of the row-major buffer becomes `live`
      const uint32_t gpr = (uint32_t)vpr >> 4;
      const __amdgpu_buffer_rsrc_t cdst = r.km_rows ? rq_rsrc(out, (int)(r.km_rows * gpr * 64u)) : rq_rsrc((const uint32_t*)out + row * vpr, nrows * vpr * 4);
      const auto coff = [&](int ln) -> uint32_t {
        if (!r.km_rows) return (uint32_t)ln * 16u;
        const uint32_t u = (uint32_t)ln >> 2, second = (nrows == 2 && u >= gpr) ? 1u : 0u;
        return rq_km4_off((uint32_t)row + second, u - second * gpr, u < (uint32_t)nrows * gpr, ln, r.km_rows);
      };
      const uint32_t tpad = (r.km_rows + 3u) & ~3u;   // k-major: the scales as fp32 [G][rows rounded up to 4] (include/fpq.h)
      const __amdgpu_buffer_rsrc_t sdst = r.km_rows ? rq_rsrc(r.code_scales, (int)(tpad * gpr * 4u)) : rq_rsrc(r.code_scales + row * (vpr >> 4), nrows * (vpr >> 4) * 2);
      const auto soff = [&](int u_) -> uint32_t {
        if (!r.km_rows) return (uint32_t)u_ * 2u;
        const uint32_t u = (uint32_t)u_, second = (nrows == 2 && u >= gpr) ? 1u : 0u;
        return u < (uint32_t)nrows * gpr ? ((((u - second * gpr) * tpad + (uint32_t)row + second) * 4u) | 0x80000000u) : 0xFFFFFFFFu;
      };
      if constexpr (HW4)
        rq_store_codes_hw(img, yw, s, cdst, sdst, lane, coff, soff);
      else
        rq_store_codes((u32x4*)img, yw, s, lut, a.shift, cdst, sdst, lane, coff, soff);
      if (MAXC == 5 && do_slot) {
        const uint32_t cd = HW4 ? codes_vec16_hw(y1, s1.inv) : codes_vec16(y1, lut, a.shift, s1.inv, s1.inv_lo);
        if (slot_live) {
          if (r.km_rows) {   // vector j = 256 + slot_idx of slot_row: 4 bytes at byte 4 (j & 3) of chunk (j & 15) >> 2 of group j >> 4
            const uint32_t j = 256u + (uint32_t)slot_idx;
            ((uint32_t*)out)[(km4_off((uint32_t)slot_row, j >> 4, (j & 15u) >> 2, r.km_rows) >> 2) + (j & 3u)] = cd;
            if ((lane & 15) == 0)
              ((float*)r.code_scales)[(int64_t)(j >> 4) * tpad + slot_row] = (float)__builtin_bit_cast(_Float16, (uint16_t)(s1.s16x2 & 0xFFFFu));
          } else {
            ((uint32_t*)out)[slot_at] = cd;
          }
          if (!r.km_rows && (lane & 15) == 0) r.code_scales[slot_at >> 4] = (uint16_t)(s1.s16x2 & 0xFFFFu);
        }
      }
    } else {
      if constexpr (HW6 != 0) {
        uint32_t q6[16], lv6[16];
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) q6[2 * c + rr] = div_pair16(yw[c][rr], s.inv, s.inv_lo, s.inv, s.inv_lo);
        fp6_levels_hw32<HW6 == 2>(q6, lv6);
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) yw[c][rr] = pk_fma0_f16(lv6[2 * c + rr], s.s16x2);
      } else
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          FPQ_PHASE("quant_divide");
          const uint32_t rb = div_pair16(yw[c][rr], s.inv, s.inv_lo, s.inv, s.inv_lo);
          if constexpr (HW4) {
            FPQ_PHASE("quant_level_hw");
            const uint32_t lv = e2m1_levels_hw(rb);
            FPQ_PHASE("quant_dequant_mul");
            yw[c][rr] = pk_fma0_f16(lv, s.s16x2);
          } else {
            FPQ_PHASE("quant_lookup");
            const uint32_t u = pk_sub_u16(rb, pk_lshr_u16(rb, 15));
            const uint32_t lv = rq_lut_pair(lut, u, a.shift);
            FPQ_PHASE("quant_dequant_mul");
            yw[c][rr] = pk_mul_f16(lv, s.s16x2);
          }
        }
      FPQ_STAMP(6);                                 // divide, level, dequantize
#ifdef FPQ_ADALN_NOMEM
      rq_store_tile<TIGHT>(img, yw, rq_rsrc(out + row * vpr, 0), la);
#else
      rq_store_tile<TIGHT>(img, yw, rq_rsrc(out + row * vpr, nrows * vpr * 16), la);
#endif
      FPQ_STAMP(7);                                 // output image round trip, stores issued
      FPQ_PHASE("row_end");
      if (MAXC == 5 && do_slot) {
        u32x4 o1;
        if constexpr (HW4) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            o1[k] = pk_fma0_f16(e2m1_levels_hw(div_pair16(y1[k], s1.inv, s1.inv_lo, s1.inv, s1.inv_lo)), s1.s16x2);
        } else if constexpr (HW6 != 0) {   // the slot chunk: 8 live values in a conversion of 32
          uint32_t q5[4], l5[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) q5[k] = div_pair16(y1[k], s1.inv, s1.inv_lo, s1.inv, s1.inv_lo);
          fp6_levels_hw32<HW6 == 2, 4>(q5, l5);
#pragma unroll
          for (int k = 0; k < 4; ++k) o1[k] = pk_fma0_f16(l5[k], s1.s16x2);
        } else {
          o1 = quant_vec16<false>(y1, lut, a.shift, s1.inv, s1.inv_lo, s1.s16x2, 0.f, 0.f, 0u);
        }
        if (slot_live) {
          if constexpr (EMIT) {
            if (h_out) __builtin_nontemporal_store(u32x4{hw_slot[0] ^ sx[0], hw_slot[1] ^ sx[1], hw_slot[2] ^ sx[2], hw_slot[3] ^ sx[3]}, h_out + slot_at);
            if (y_out) __builtin_nontemporal_store(y1, y_out + slot_at);
          }
          __builtin_nontemporal_store(o1, out + slot_at);
        }
      }
    }
  };

  // ---- the workgroup's rows: wavefront w takes rows lo + w, lo + w + 4, ... ----
  u32x4 cur[RV], alt[DB ? RV : 1];
  int64_t i = lo + RPU * wave;
#if !defined(FPQ_ADALN_COPYONLY) || FPQ_ADALN_COPYONLY < 2
  StageRaw sraw[NST];
  // fp32 rows (10 - 40 KiB of row requests per workgroup): the modulation's loads go first; fp16 rows: the row first, as in
  // round 3 (ten scale steps cold: fp32 rows 176.3 -> 172.9 us for d30 with the modulation first, fp16 rows
  // 137.1 -> 139.2; profiles/r04_adaln_stage_order.txt)
#ifdef FPQ_ADALN_ROW_FIRST
  constexpr bool MOD_FIRST = false;
#else
  constexpr bool MOD_FIRST = X32;
#endif
  if constexpr (MOD_FIRST) {
    stage_load(b, sraw);
    __builtin_amdgcn_sched_barrier(0);           // keep the modulation's loads in front of the row's in the instruction stream
    if (i < hi) load_row(cur, i);
  } else {
    if (i < hi) load_row(cur, i);
    __builtin_amdgcn_sched_barrier(0);
    stage_load(b, sraw);
  }
  if constexpr (!HW4 && HW6 == 0) lut16_stage(lut, tab, a.shift);
  stage_store(sraw);
  __syncthreads();
#else
  if (i < hi) load_row(cur, i);
#endif
  if constexpr (DB) {
    for (; i < hi; i += 2 * W) {                 // two register sets alternate: no row is copied between registers
      do_row(cur, alt, i, i + W < hi ? i + W : -1);
      if (i + W < hi) do_row(alt, cur, i + W, i + 2 * W < hi ? i + 2 * W : -1);
    }
  } else if (i < hi) {
    // first pass peeled: both edges into the loop then carry "loads, then this row's stores", and the wait for the
    // prefetched row leaves the stores in flight (fpq_rotate_mfma.h, rotate_quant_mfma_kernel)
    do_row(cur, cur, i, i + RPU * W < hi ? i + RPU * W : -1);
    for (i += RPU * W; i < hi; i += RPU * W) do_row(cur, cur, i, i + RPU * W < hi ? i + RPU * W : -1);
  }
#ifdef FPQ_ADALN_STAMPS
  if constexpr (!EMIT) {
    if (y_out && lane == 0) {
      unsigned long long* dst = (unsigned long long*)y_out + ((int64_t)blockIdx.x * W + wave) * 16;
#pragma unroll
      for (int k = 0; k < 8; ++k) dst[k] = st_sum[k];
      dst[8] = st_rows;
      dst[9] = st_last;
      dst[10] = st_t0;                                   // residency timeline: when and where this wavefront lived
      dst[11] = __builtin_amdgcn_s_memrealtime();
      unsigned hw_id, xcc_id;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
      dst[12] = hw_id;
      dst[13] = xcc_id;
      dst[14] = st_first;
    }
  }
#endif
}
