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
template <typename Tmod, int MAXC, bool CODES, bool EMIT, bool TOKEN = false, bool MFMA = false, bool X32 = false>
__global__ __launch_bounds__(kBlock, (X32 && MFMA && MAXC == 4 && !EMIT) ? FPQ_ADALN_X32_WAVES : 1) FPQ_ADALN_OCC void adaln_rq16_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ out,
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
  static_assert(!MFMA || MAXC <= 5, "matrix-core rotation: rows of at most 20 groups");
  u32x4* buf = nullptr;           // MFMA: this wavefront's operand / output image
  HadOperand ha = {};
  if constexpr (MFMA) {
    __shared__ u32x4 xpose[W][kRqTileVec];
    buf = xpose[wave];
    ha = had_operand(lane);
  } else if constexpr (X32) {
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
      uint32_t sx2[2] = {0, 0};
      if constexpr (EMIT && MFMA) {   // h_out wants the modulated row without the rotation's signs
        const int j0 = (k2 & 15) * 8 + 4 * hsel;
        const uint32_t db = (r.sign[j0 >> 5] >> (j0 & 31)) & 0xFu;
        sx2[0] = ((db & 1u) << 15) | (((db >> 1) & 1u) << 31);
        sx2[1] = (((db >> 2) & 1u) << 15) | (((db >> 3) & 1u) << 31);
      }
#pragma unroll
      for (int n = 0; n < (MFMA ? 8 : RV); ++n) {
        const int v = 32 * n + k2;                       // chunk of the row
        u32x2 hw2 = {0, 0};
        if (n < RV) {
          const u32x4 A = planes[hsel * vpr + v], B = planes[(2 + hsel) * vpr + v];   // beyond the row: in bounds, unused
          const u32x4 w = cur[n];
          hw2[0] = f2h2(__builtin_fmaf(__builtin_fmaf(u2f(w[0]), rstd, nm), u2f(A[0]), u2f(B[0])),
                        __builtin_fmaf(__builtin_fmaf(u2f(w[1]), rstd, nm), u2f(A[1]), u2f(B[1])));
          hw2[1] = f2h2(__builtin_fmaf(__builtin_fmaf(u2f(w[2]), rstd, nm), u2f(A[2]), u2f(B[2])),
                        __builtin_fmaf(__builtin_fmaf(u2f(w[3]), rstd, nm), u2f(A[3]), u2f(B[3])));
          if (n >= RV - 2 && !qlive[n]) hw2 = u32x2{0, 0};
          if constexpr (EMIT && MFMA) {
            if (h_out && qlive[n])
              __builtin_nontemporal_store(u32x2{hw2[0] ^ sx2[0], hw2[1] ^ sx2[1]}, (u32x2*)(h_out + row * vpr + v) + hsel);
          }
        }
        if constexpr (MFMA) {
          const int g = 2 * n + (k2 >> 4), pc = k2 & 15;
          img[(pc * 16 + (g ^ pc)) * 2 + hsel] = hw2;
        } else {
          img[v * 2 + hsel] = hw2;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if constexpr (MFMA) {
      // ---- modulate into the B-operand image (all 16 groups defined: vectors beyond the row are zeros), transform on
      // the matrix cores, quantize the 32 outputs this lane holds of its group, out through the same image ----
      const int lane_w = rq_opaque(lane);
#pragma unroll
      for (int c = 0; c < (X32 ? 0 : 4); ++c) {
        u32x4 hw = {0, 0, 0, 0};
        if (c < MAXC) {
          const int v = vidx[c];
          const u32x4 A0 = planes[v], B0 = planes[2 * vpr + v];
          const u32x4 A1 = planes[vpr + v], B1 = planes[3 * vpr + v];
          const u32x4 w = cur[c];
          hw[0] = f2h2(__builtin_fmaf(fma_h_lo(w[0], rstd, nm), u2f(A0[0]), u2f(B0[0])),
                       __builtin_fmaf(fma_h_hi(w[0], rstd, nm), u2f(A0[1]), u2f(B0[1])));
          hw[1] = f2h2(__builtin_fmaf(fma_h_lo(w[1], rstd, nm), u2f(A0[2]), u2f(B0[2])),
                       __builtin_fmaf(fma_h_hi(w[1], rstd, nm), u2f(A0[3]), u2f(B0[3])));
          hw[2] = f2h2(__builtin_fmaf(fma_h_lo(w[2], rstd, nm), u2f(A1[0]), u2f(B1[0])),
                       __builtin_fmaf(fma_h_hi(w[2], rstd, nm), u2f(A1[1]), u2f(B1[1])));
          hw[3] = f2h2(__builtin_fmaf(fma_h_lo(w[3], rstd, nm), u2f(A1[2]), u2f(B1[2])),
                       __builtin_fmaf(fma_h_hi(w[3], rstd, nm), u2f(A1[3]), u2f(B1[3])));
          if (c == MAXC - 1 && !last_live) hw = u32x4{0, 0, 0, 0};
          if constexpr (EMIT) {
            if (h_out && (c < MAXC - 1 || last_live))
              __builtin_nontemporal_store(u32x4{hw[0] ^ sx[0], hw[1] ^ sx[1], hw[2] ^ sx[2], hw[3] ^ sx[3]}, h_out + row * vpr + v);
          }
        }
        const int g = 4 * c + (lane_w >> 4), p = lane_w & 15;
        buf[p * 16 + (g ^ p)] = hw;
      }
      __builtin_amdgcn_wave_barrier();
      uint32_t yw[8][2];
      const float mf = hadamard128_mfma(buf, ha, r.c_h, lane, yw);
      __builtin_amdgcn_wave_barrier();
      if constexpr (EMIT) {
        if (y_out) rq_store_tile(buf, yw, rq_rsrc(y_out + row * vpr, vpr * 16), lane);
      }
      FPQ_PHASE("group_max_scale");
      uint32_t m = mul2_to_h2(mf, 0.0f, r.c_h) & 0xFFFFu;       // see rotate_quant_mfma_kernel
      if (__builtin_expect((yw[0][0] & 0x7C00u) == 0x7C00u, 0)) {
        m = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) m = pk_max_u16(m, pk_max_u16(yw[c][0] & 0x7FFF7FFFu, yw[c][1] & 0x7FFF7FFFu));
        const uint32_t lo = m & 0xFFFFu, hi = m >> 16;
        m = lo > hi ? lo : hi;
      }
      // ---- groups 16 .. 19 (vectors 256 .. vpr - 1; d36: C = 2304 = 18 groups): a second tile would run its epilogue
      // for 64 lanes to serve 8 - 16 of them; one chunk per lane with the transform as butterflies costs 40 % of a tile
      // and keeps every lane busy.  Rotated here, quantized after the tile (per token: with the row's one scale) ----
      u32x4 hw = {0, 0, 0, 0}, y1 = {0, 0, 0, 0};
      uint32_t m1 = 0;
      const int v = vidx[MAXC - 1];
      if constexpr (MAXC == 5) {
        if constexpr (X32) {   // the two half-chunk loads beyond the tile: halves meet in the (free again) image
          const int lane_x = rq_opaque(lane);
          const int hsel = lane_x & 1, k2 = lane_x >> 1;
#pragma unroll
          for (int n = 8; n < RV; ++n) {
            const int v2 = 32 * n + k2;
            const u32x4 A = planes[hsel * vpr + v2], B = planes[(2 + hsel) * vpr + v2];
            const u32x4 w = cur[n];
            u32x2 hw2;
            hw2[0] = f2h2(__builtin_fmaf(__builtin_fmaf(u2f(w[0]), rstd, nm), u2f(A[0]), u2f(B[0])),
                          __builtin_fmaf(__builtin_fmaf(u2f(w[1]), rstd, nm), u2f(A[1]), u2f(B[1])));
            hw2[1] = f2h2(__builtin_fmaf(__builtin_fmaf(u2f(w[2]), rstd, nm), u2f(A[2]), u2f(B[2])),
                          __builtin_fmaf(__builtin_fmaf(u2f(w[3]), rstd, nm), u2f(A[3]), u2f(B[3])));
            if (!qlive[n]) hw2 = u32x2{0, 0};
            ((u32x2*)buf)[(v2 - 256) * 2 + hsel] = hw2;
          }
          __builtin_amdgcn_wave_barrier();
          hw = buf[rq_opaque(lane)];
          __builtin_amdgcn_wave_barrier();
        } else {
          const u32x4 A0 = planes[v], B0 = planes[2 * vpr + v];
          const u32x4 A1 = planes[vpr + v], B1 = planes[3 * vpr + v];
          const u32x4 w = cur[4];
          hw[0] = f2h2(__builtin_fmaf(fma_h_lo(w[0], rstd, nm), u2f(A0[0]), u2f(B0[0])),
                       __builtin_fmaf(fma_h_hi(w[0], rstd, nm), u2f(A0[1]), u2f(B0[1])));
          hw[1] = f2h2(__builtin_fmaf(fma_h_lo(w[1], rstd, nm), u2f(A0[2]), u2f(B0[2])),
                       __builtin_fmaf(fma_h_hi(w[1], rstd, nm), u2f(A0[3]), u2f(B0[3])));
          hw[2] = f2h2(__builtin_fmaf(fma_h_lo(w[2], rstd, nm), u2f(A1[0]), u2f(B1[0])),
                       __builtin_fmaf(fma_h_hi(w[2], rstd, nm), u2f(A1[1]), u2f(B1[1])));
          hw[3] = f2h2(__builtin_fmaf(fma_h_lo(w[3], rstd, nm), u2f(A1[2]), u2f(B1[2])),
                       __builtin_fmaf(fma_h_hi(w[3], rstd, nm), u2f(A1[3]), u2f(B1[3])));
        }
        if (!last_live) hw = u32x4{0, 0, 0, 0};
        const u32x4 hwa[1] = {hw};
        float t1[1][8];
        fwht128_h_n<1>(hwa, t1, 1, lg);
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
        if constexpr (MAXC == 5) {
          uint32_t ms[1] = {m1};
          row_max_dpp16_n<1>(ms, 1);
          s1 = row_scale16(ms[0], a.fpos.gmax, a.inv_gpos);
        }
      }
      if constexpr (CODES && TOKEN) {   // per-token operands: E4M3 bytes or dense 6-bit codes, the row scale is out already
        if (r.code_bits == 6) rq_store_codes6(buf, yw, s, lut, a.shift, rq_rsrc((const uint8_t*)out + row * ((int64_t)vpr * 6), vpr * 6), lane);
        else rq_store_codes8(buf, yw, s, lut, a.shift, rq_rsrc((const uint8_t*)out + row * ((int64_t)vpr * 8), vpr * 8), lane);
        if constexpr (MAXC == 5) {
          if (last_live) token_codes_out(4, y1, s1, v);
        }
      } else if constexpr (CODES) {   // FP4 operands: codes + one fp16 scale per group (fpq_gemm_fp4.h)
        rq_store_codes(buf, yw, s, lut, a.shift, rq_rsrc((const uint32_t*)out + row * vpr, vpr * 4),
                       rq_rsrc(r.code_scales + row * (vpr >> 4), (vpr >> 4) * 2), lane);
        if constexpr (MAXC == 5) {
          const uint32_t cd = codes_vec16(y1, lut, a.shift, s1.inv, s1.inv_lo);
          if (last_live) {
            const int64_t at = row * vpr + v;
            ((uint32_t*)out)[at] = cd;
            if (lg == 0) r.code_scales[at >> 4] = (uint16_t)(s1.s16x2 & 0xFFFFu);
          }
        }
      } else {
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          FPQ_PHASE("quant_divide");
          const uint32_t rb = div_pair16(yw[c][rr], s.inv, s.inv_lo, s.inv, s.inv_lo);
          FPQ_PHASE("quant_lookup");
          const uint32_t u = pk_sub_u16(rb, pk_lshr_u16(rb, 15));
          const uint32_t lv = rq_lut_pair(lut, u, a.shift);
          FPQ_PHASE("quant_dequant_mul");
          yw[c][rr] = pk_mul_f16(lv, s.s16x2);
        }
      rq_store_tile(buf, yw, rq_rsrc(out + row * vpr, vpr * 16), lane);
      FPQ_PHASE("row_end");
      }
      if constexpr (MAXC == 5 && !CODES) {
        const u32x4 o1 = quant_vec16<false>(y1, lut, a.shift, s1.inv, s1.inv_lo, s1.s16x2, 0.f, 0.f, 0u);
        if (last_live) {
          const int64_t at = row * vpr + v;
          if constexpr (EMIT) {
            if (h_out) __builtin_nontemporal_store(u32x4{hw[0] ^ sx[0], hw[1] ^ sx[1], hw[2] ^ sx[2], hw[3] ^ sx[3]}, h_out + at);
            if (y_out) __builtin_nontemporal_store(y1, y_out + at);
          }
          __builtin_nontemporal_store(o1, out + at);
        }
      }
    } else {
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
          hw[j][0] = f2h2(__builtin_fmaf(fma_h_lo(w[0], rstd, nm), u2f(A0[0]), u2f(B0[0])),
                          __builtin_fmaf(fma_h_hi(w[0], rstd, nm), u2f(A0[1]), u2f(B0[1])));
          hw[j][1] = f2h2(__builtin_fmaf(fma_h_lo(w[1], rstd, nm), u2f(A0[2]), u2f(B0[2])),
                          __builtin_fmaf(fma_h_hi(w[1], rstd, nm), u2f(A0[3]), u2f(B0[3])));
          hw[j][2] = f2h2(__builtin_fmaf(fma_h_lo(w[2], rstd, nm), u2f(A1[0]), u2f(B1[0])),
                          __builtin_fmaf(fma_h_hi(w[2], rstd, nm), u2f(A1[1]), u2f(B1[1])));
          hw[j][3] = f2h2(__builtin_fmaf(fma_h_lo(w[3], rstd, nm), u2f(A1[2]), u2f(B1[2])),
                          __builtin_fmaf(fma_h_hi(w[3], rstd, nm), u2f(A1[3]), u2f(B1[3])));
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
    if constexpr (X32 && !MFMA) __builtin_amdgcn_wave_barrier();   // the row buffer is rewritten by the next row
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