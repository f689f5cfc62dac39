// fpq_fast16.h - the fp16 -> fp16 hot path (activations, KV cache).  Included by
// fpq_kernels.hip inside its anonymous namespace, after the generic helpers.
//
// Same results as the generic kernels (rows_subwave_kernel / rows_block_kernel),
// about a fifth of their VALU work per element:
//
//   * xn = fp16(x / s) costs 3 fp32 ops instead of an IEEE division sequence:
//         y  = x * inv            inv = v_rcp_f32(s)   (once per row)
//         e  = fma(-y, s, x)      exact residual
//         y2 = fma(e, inv, y)     = RN32(x/s * (1 - O(2^-44)))
//     x and s carry 11 significant bits, so x/s is either exactly a 12-bit number
//     (then y2 lands on it exactly and the fp16 tie rounds to even as IEEE does) or at
//     least 2^-23 (relative) away from every 12-bit number, i.e. from every fp16
//     rounding boundary (then y2 is on the same side).  Either way
//     fp16(y2) == fp16(RN32(x/s)) == what torch's fp16 division produces.
//     (Rounding the LAST fma straight to fp16 - v_fma_mixlo_f16 - would be wrong in
//     the tie case: the fp32 rounding is what snaps y2 onto the tie.)
//   * scale = fp16(amax / gmax) the same way with the compile-time-free constant 1/gmax.
//   * the minifloat rounding is a table in LDS indexed by the top bits of xn's fp16
//     pattern: every rounding threshold of every supported table is a multiple of
//     2^shift in that bit pattern (E2M1: shift 8 -> 256 entries; int_neg: shift 5),
//     so bucket -> level is exact.  "tie goes to the larger VALUE" becomes: subtract
//     1 from a negative pattern's magnitude before bucketing.  The entry is the fp16
//     pattern of the (signed) level with the table's zero as +0, so -0.0 never appears.
//     The table is filled per workgroup from the same closed form the generic kernels
//     use (quant_mag), which tests pin to the reference's scan.
//   * out = v_pk_mul_f16(level, scale): the exact product rounded once = fp16(fp32(q*s)).
//   * absmax on packed 16-bit integer patterns (NaN = largest pattern propagates like
//     torch.max), row reduction in DPP (rows of 8/16 lanes never leave a DPP row).
#pragma once

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef uint16_t us2_t __attribute__((ext_vector_type(2)));
typedef int16_t s2_t __attribute__((ext_vector_type(2)));

constexpr int kLutLdsEntries = 2048;   // shift >= 5: at most 2 x 1024 buckets

struct Lut16Args {
  Fmt fneg, fpos;      // identical for the symmetric tables
  float inv_gneg;      // RN32(1 / gmax) per side
  float inv_gpos;
  int shift;           // bucket width = 2^shift fp16 patterns
  uint32_t* nan_flag;  // dual format only: nullptr, or device word OR-ed with 1 when an input is NaN
  const void* clip_absmax;   // dual format per group only (CLIP kernels): device scalar max|x| in x's dtype, or nullptr
  float clip_strength;       // the reference's global clamp to +-strength * max|x| (tr/quant_utils.py:421-422)
  void* gelu_out;            // GELU kernels only: nullptr, or fp16 [rows, cols] receiving the GELU values the quantizer saw
};

__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
  us2_t r = __builtin_elementwise_max(__builtin_bit_cast(us2_t, a), __builtin_bit_cast(us2_t, b));
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) {
  s2_t r = __builtin_elementwise_max(__builtin_bit_cast(s2_t, a), __builtin_bit_cast(s2_t, b));
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pk_sub_u16(uint32_t a, uint32_t b) {
  us2_t r = __builtin_bit_cast(us2_t, a) - __builtin_bit_cast(us2_t, b);
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pk_lshr_u16(uint32_t a, int n) {
  us2_t r = __builtin_bit_cast(us2_t, a) >> (uint16_t)n;
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pk_ashr_i16(uint32_t a, int n) {
  s2_t r = __builtin_bit_cast(s2_t, a) >> (int16_t)n;
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pk_add_f16(uint32_t a, uint32_t b) {
  h2_t r = __builtin_bit_cast(h2_t, a) + __builtin_bit_cast(h2_t, b);
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pk_mul_f16(uint32_t a, uint32_t b) {
  h2_t r = __builtin_bit_cast(h2_t, a) * __builtin_bit_cast(h2_t, b);
  return __builtin_bit_cast(uint32_t, r);
}

// two fp32 values -> packed fp16 pair with ONE v_cvt_pk_f16_f32 (round to nearest even).  The
// asm barrier keeps both values as rounded fp32 numbers first (see f2h in fpq_kernels.hip).
__device__ __forceinline__ uint32_t f2h2(float lo, float hi) {
  asm volatile("" : "+v"(lo), "+v"(hi));
  typedef float f2c_t __attribute__((ext_vector_type(2)));
  h2_t r = __builtin_convertvector(f2c_t{lo, hi}, h2_t);
  return __builtin_bit_cast(uint32_t, r);
}

// (float)half_lo(w) * b and (float)half_hi(w) * b in one v_fma_mix_f32 each: the fp16 -> fp32
// widening rides on the multiply.  The addend is +0, which only differs from a plain product
// when the product is -0 (x = -0.0); every use below maps +-0 to the same table bucket.
__device__ __forceinline__ float mul_h_lo(uint32_t w, float b) {
  float d;
  asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(w), "v"(b));
  return d;
}
__device__ __forceinline__ float mul_h_hi(uint32_t w, float b) {
  float d;
  asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(w), "v"(b));
  return d;
}

// x - y * s with x an fp16 half of w (widened by the instruction), y and s fp32: the exact residual of the division
// step without a separate fp16 -> fp32 conversion of x
__device__ __forceinline__ float resid_h_lo(uint32_t w, float y, float s) {
  float d;
  asm("v_fma_mix_f32 %0, -%1, %2, %3 op_sel_hi:[0,0,1]" : "=v"(d) : "v"(y), "v"(s), "v"(w));
  return d;
}
__device__ __forceinline__ float resid_h_hi(uint32_t w, float y, float s) {
  float d;
  asm("v_fma_mix_f32 %0, -%1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(d) : "v"(y), "v"(s), "v"(w));
  return d;
}

// D = (float)half_lo(w) * b + c  /  (float)half_hi(w) * b + c (fp32 result)
__device__ __forceinline__ float fmaf_h_lo(uint32_t w, float b, float c) {
  float d;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(w), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float fmaf_h_hi(uint32_t w, float b, float c) {
  float d;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(w), "v"(b), "v"(c));
  return d;
}

// xn = half(x / s) for the two halves of w, as far as any table can tell: y = fp32(x * inv), inv = RN32(1 / s), then ONE
// v_cvt_pk_f16_f32 for the pair.
//   * inv is the correctly rounded reciprocal (row_scale16: one Newton step on v_rcp_f32 leaves (1/s)(1 - 2^-44), and
//     2^k / b for an 11-bit b is further than that from every fp32 rounding boundary), the product is rounded once:
//     |y / (x/s) - 1| <= (1 + 2^-24)^2 - 1 = 2^-23 + 2^-48.
//   * What the quantizers take from xn is its bucket: the level only changes where half(x/s) crosses a rounding
//     threshold T of the table, i.e. where x/s crosses T moved by the half fp16 ulp of the rounding to fp16: a number
//     c / 2^p with a 12-bit ODD c (T itself has a few bits).  x = a 2^i and s = b 2^j with 11-bit a, b: x/s = c / 2^p would
//     need a = g c for g = gcd(a, b) - impossible, c > 2047 >= a - so |x/s - c/2^p| / (x/s) = |2^q a - b c| / (2^q a) is
//     at least 1 / (4096 * 2047) = 2^-22.9993.  2^-23 + 2^-48 < 2^-22.9993: y is on the same side of every such number
//     as x/s, so half(y) falls into the bucket of half(x/s) - it may be the neighbouring fp16 value INSIDE a bucket,
//     which no caller can observe (they all go xn -> bucket -> level).
//   * The margin is 0.05 %, so the argument is backed by exhaustion: tests/test_gpu_parity.py::
//     test_every_fp16_pair_fast_path_vs_ieee_path pushes all 1.0e9 (group maximum, element) pairs through this path and
//     through the IEEE division for seven table configurations, and ..._hw_levels the same pairs through the
//     hardware-level form of the producers.
// Round 2 carried 1/s as inv + inv_lo and rounded x * inv + x * inv_lo straight to fp16 (the exact quotient, 25 cycles of
// the vector pipe per pair: v_fma_mixlo/hi_f16 cost 8.2 each, tools/probe/valu_issue_cost.hip); this form: 13.
// -DFPQ_DIV_WITH_LO restores the two-term reciprocal (21.8 cycles per pair) for A/B runs.
__device__ __forceinline__ uint32_t div_pair16(uint32_t w, float ih0, float il0, float ih1, float il1) {
#ifdef FPQ_DIV_WITH_LO
  const float t0 = mul_h_lo(w, il0), t1 = mul_h_hi(w, il1);
  return f2h2(fmaf_h_lo(w, ih0, t0), fmaf_h_hi(w, ih1, t1));
#else
  (void)il0;
  (void)il1;
  return f2h2(mul_h_lo(w, ih0), mul_h_hi(w, ih1));
#endif
}

// the two table entries of a packed pair of bucket patterns as ONE packed register: byte offsets straight from the
// pattern (bucket * 2 == (pattern >> (shift - 1)) & mask), 16-bit loads into the low / high half
__device__ __forceinline__ uint32_t lut_pair16(const uint16_t* lut, uint32_t u, int shift) {
  const uint32_t mask = ((1u << (16 - shift)) - 1u) << 1;
  const uint32_t off0 = (u >> (shift - 1)) & mask, off1 = (u >> (15 + shift)) & mask;
  h2_t q;
  q.x = *(const _Float16*)((const char*)lut + off0);
  q.y = *(const _Float16*)((const char*)lut + off1);
  return __builtin_bit_cast(uint32_t, q);
}

// max over the LPR lanes that own a row; LPR lanes are contiguous and LPR-aligned
template <int LPR>
__device__ __forceinline__ uint32_t row_max_dpp(uint32_t v) {
  auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
  if (LPR >= 2) v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  if (LPR >= 4) v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  if (LPR >= 8) v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));  // row_half_mirror
  if (LPR >= 16) v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true)); // row_mirror
  if (LPR >= 32) v = mx(v, (uint32_t)__shfl_xor((int)v, 16, 64));
  if (LPR >= 64) v = mx(v, (uint32_t)__shfl_xor((int)v, 32, 64));
  return v;
}

// fp16(a / g) for an fp16-valued a and a constant g given as inv_g = RN32(1/g):
// y = a*inv_g is within 2^-23 of a/g, one residual step makes it RN32(a/g(1+tiny)).
__device__ __forceinline__ uint32_t scale_bits_f16(uint32_t amax_bits, float g, float inv_g) {
  float a = h2f(amax_bits);
  if (!(a < __builtin_inff())) return amax_bits;   // inf -> inf, NaN -> NaN
  float y = a * inv_g;
  float e = __builtin_fmaf(-y, g, a);
  float y2 = __builtin_fmaf(e, inv_g, y);
  return f2h(y2);   // NaN/Inf propagate: amax = inf -> inf, NaN -> NaN
}

__device__ __forceinline__ void lut16_fill(uint16_t* lut, const Lut16Args& a) {
  const int n = 1 << (16 - a.shift);
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    uint32_t u = (uint32_t)i << a.shift;
    bool neg = (u >> 15) != 0;
    float r = h2f(u & 0x7FFFu);   // lowest magnitude pattern of the bucket
    float qm = quant_mag(r, 0u, neg ? a.fneg : a.fpos);
    uint32_t qb = f2h(qm);
    lut[i] = (uint16_t)((neg && qm != 0.0f) ? (qb | 0x8000u) : qb);
  }
}

// E2M1 levels from the FP4 conversion hardware instead of the bucket table:  level = f16_fp4(fp4_f32(float(xn) + 2^-14)).
// The reference's scan sends a tie to the LARGER value on both sides of zero (quant/quant_kernel.cu:25-37); the
// conversion rounds to nearest even.  fp16 values >= 0.25 are multiples of 2^-12 and every rounding boundary of E2M1 is a
// multiple of 0.25, so adding 2^-14 - exact in fp32 - removes every tie and sends it to the side the scan picks; below
// 0.25 nothing is a boundary.  tools/probe/cvt_fp4_probe.hip checks all 63 488 finite fp16 values against the scan on the
// hardware (profiles/r03_cvt_fp4_probe.txt), and the exhaustive (maximum, element) sweep of tests/test_gpu_parity.py runs
// through this path.  Levels that round to zero from below come back as -0: the dequantizing multiply is an fma with +0.
// Non-finite quotients (which saturate to 6 here, where the scan gives 0) only occur under a non-finite scale, and that
// is replaced by NaN (the reference's 0 * inf): scale_nan_if_not_finite.  No table, no LDS traffic.
__device__ __forceinline__ uint32_t e2m1_levels_hw(uint32_t xn2) {
  const float t0 = fmaf_h_lo(xn2, 1.0f, 0x1p-14f), t1 = fmaf_h_hi(xn2, 1.0f, 0x1p-14f);
  const uint32_t code = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(xn2, t0, t1, 1.0f, 0);   // byte 0 of a dead register
  return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(code, 1.0f, 0));
}
// ... and the hardware E2M1 CODES themselves (sign-magnitude nibbles, the operand format of the FP4 matrix cores,
// fpq_gemm_fp4.h): the two codes of a packed pair land in byte SEL of `acc`; four calls fill a register with the codes of
// 8 consecutive elements, low nibble first.  e2m1_codes_canon turns the -0 codes (negative values that round to zero)
// into +0, as the code table of the table-driven form has them (decode bit-equal to the fake-quantized values).
template <int SEL>
__device__ __forceinline__ uint32_t e2m1_codes_hw(uint32_t acc, uint32_t xn2) {
  const float t0 = fmaf_h_lo(xn2, 1.0f, 0x1p-14f), t1 = fmaf_h_hi(xn2, 1.0f, 0x1p-14f);
  return __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(acc, t0, t1, 1.0f, SEL);
}
__device__ __forceinline__ uint32_t e2m1_codes_canon(uint32_t w) {
  const uint32_t t = (w & 0x77777777u) + 0x77777777u;   // bit 3 of a nibble: its magnitude is not zero (no carry between nibbles)
  return w & (t | 0x77777777u);
}
// the 8 codes of one 16-byte vector (cf. codes_vec16)
__device__ __forceinline__ uint32_t codes_vec16_hw(const u32x4& w, float inv) {
  uint32_t c = 0;
  c = e2m1_codes_hw<0>(c, div_pair16(w[0], inv, 0.f, inv, 0.f));
  c = e2m1_codes_hw<1>(c, div_pair16(w[1], inv, 0.f, inv, 0.f));
  c = e2m1_codes_hw<2>(c, div_pair16(w[2], inv, 0.f, inv, 0.f));
  c = e2m1_codes_hw<3>(c, div_pair16(w[3], inv, 0.f, inv, 0.f));
  return e2m1_codes_canon(c);
}
__device__ __forceinline__ uint32_t pk_fma0_f16(uint32_t a, uint32_t b) {   // a * b + (+0): a -0 product becomes +0
  const h2_t z = {(_Float16)0.0f, (_Float16)0.0f};
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(__builtin_bit_cast(h2_t, a), __builtin_bit_cast(h2_t, b), z));
}
// FP6 levels without a table (round 4), for a lane that holds exactly 32 quotients (16 packed words): the FP6 conversion
// hardware takes two tuples of 16 floats, writes 32 six-bit codes (v_cvt_scalef32_2xpk16_{fp6,bf6}_f32: E2M3 / E3M2) and
// decodes them again (v_cvt_scalef32_pk32_f16_{fp6,bf6}).  It rounds to nearest-even; the reference's scan sends a tie to
// the LARGER value: float(xn) + 2^-17 - exact in fp32 for |xn| < 2^6, far below the distance of any fp16 value from a
// rounding threshold (thresholds are multiples of 2^-5 (E2M3) / 2^-6 (E3M2), fp16 values near them multiples of 2^-14 or
// coarser) - removes every tie towards the larger value and moves nothing else: tools/probe/cvt_fp6_probe.hip runs all
// 63 488 finite fp16 inputs through it against the scan (profiles/r03_cvt_fp6_probe.txt: 0 mismatches but -0 for inputs in
// (-step/2, 0), which the dequantizing fma with +0 repairs, as for E2M1).  Non-finite quotients saturate where the scan
// gives 0: they only occur under a non-finite scale, which scale_nan_if_not_finite turns into NaN (the reference's 0 * inf).
// ~2.2 pipe cycles per element each way (70 per instruction) + the bias: about the cost of the table lookups it
// replaces - what it buys is the staging of 2 KiB of table per workgroup, its barrier, and the LDS latency in every
// wavefront's short life (profiles/r04_pmc_token6.txt: the table form is not bound by vector issue, 39 % busy).
typedef uint32_t fp6_u16v_t __attribute__((ext_vector_type(16)));
typedef uint32_t fp6_u6v_t __attribute__((ext_vector_type(6)));
typedef float fp6_f16v_t __attribute__((ext_vector_type(16)));
// NW < 16: only the first NW packed words are live (a lane's fifth vector on rows of 17 - 20 groups: 8 values in a conversion
// of 32 - the instruction costs the same, but the table, its staging and its barrier still go).
template <bool BF6, int NW = 16>
__device__ __forceinline__ void fp6_levels_hw32(const uint32_t (&xn)[NW], uint32_t (&lv)[NW]) {
  // the two source tuples INTERLEAVE in the result: code 2 i comes from a[i], code 2 i + 1 from b[i] (seen on hardware with
  // distinct inputs - the probe's neighbouring patterns quantize alike and hid it): a = the low halves, b = the high halves
  fp6_f16v_t a, b;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (j < NW) {
      a[j] = fmaf_h_lo(xn[j], 1.0f, 0x1p-17f);
      b[j] = fmaf_h_hi(xn[j], 1.0f, 0x1p-17f);
    } else {
      a[j] = b[j] = 0.0f;
    }
  }
  fp6_u6v_t c;
  fp6_u16v_t d;
  if constexpr (BF6) {
    asm("v_cvt_scalef32_2xpk16_bf6_f32 %0, %1, %2, 1.0" : "=&v"(c) : "v"(a), "v"(b));   // early clobber: a multi-pass
    asm("v_cvt_scalef32_pk32_f16_bf6 %0, %1, 1.0" : "=&v"(d) : "v"(c));                  // instruction must not write over its sources
  } else {
    asm("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, 1.0" : "=&v"(c) : "v"(a), "v"(b));
    asm("v_cvt_scalef32_pk32_f16_fp6 %0, %1, 1.0" : "=&v"(d) : "v"(c));
  }
#pragma unroll
  for (int j = 0; j < NW; ++j) lv[j] = d[j];
}

// Quantize the 8 halves of one 16-byte vector.  s16x2 = scale replicated in both halves; inv_hi + inv_lo = 1 / scale
// (0 for a zero scale).  In DUAL mode each element picks the negative or positive side's scale.
__device__ __forceinline__ uint32_t fbits16(float f) { return __builtin_bit_cast(uint32_t, f); }

template <bool DUAL, bool HW4 = false>
__device__ __forceinline__ u32x4 quant_vec16(const u32x4& w, const uint16_t* lut, int shift, float ih_n, float il_n,
                                             uint32_t s16x2_n, float ih_p, float il_p, uint32_t s16x2_p) {
  static_assert(!(DUAL && HW4), "hardware levels: the symmetric E2M1 table only");
  u32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t wk = w[k];
    float h0 = ih_n, h1 = ih_n, l0 = il_n, l1 = il_n;
    uint32_t sc = s16x2_n;
    if (DUAL) {   // sign masks + v_bfi instead of compares and selects: 8 instructions per pair, not 12
      const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int)wk, 15, 1), m1 = (uint32_t)((int)wk >> 31);   // all ones: negative
      h0 = u2f((fbits16(ih_n) & m0) | (fbits16(ih_p) & ~m0));  l0 = u2f((fbits16(il_n) & m0) | (fbits16(il_p) & ~m0));
      h1 = u2f((fbits16(ih_n) & m1) | (fbits16(ih_p) & ~m1));  l1 = u2f((fbits16(il_n) & m1) | (fbits16(il_p) & ~m1));
      const uint32_t mp = pk_ashr_i16(wk, 15);
      sc = (s16x2_n & mp) | (s16x2_p & ~mp);
    }
    const uint32_t rb = div_pair16(wk, h0, l0, h1, l1);
    if constexpr (HW4) {
      o[k] = pk_fma0_f16(e2m1_levels_hw(rb), sc);
    } else {
      const uint32_t u = pk_sub_u16(rb, pk_lshr_u16(rb, 15));   // negative patterns: magnitude - 1
      o[k] = pk_mul_f16(lut_pair16(lut, u, shift), sc);
    }
  }
  return o;
}

// Same normalisation and bucketing as quant_vec16, but the table holds 4-bit hardware codes (OCP E2M1
// nibbles, fpq_gemm_fp4.h): returns the 8 codes of the vector packed low nibble first.
__device__ __forceinline__ uint32_t codes_vec16(const u32x4& w, const uint16_t* lut, int shift, float inv_hi, float inv_lo) {
  uint32_t packed = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t rb = div_pair16(w[k], inv_hi, inv_lo, inv_hi, inv_lo);
    const uint32_t u = pk_sub_u16(rb, pk_lshr_u16(rb, 15));
    const uint32_t c0 = lut[(u & 0xFFFFu) >> shift], c1 = lut[u >> (16 + shift)];
    packed |= (c0 | (c1 << 4)) << (8 * k);
  }
  return packed;
}

// |x| patterns of one vector reduced to a per-lane maximum (symmetric tables)
__device__ __forceinline__ uint32_t vec_absmax16(const u32x4& w) {
  uint32_t m = pk_max_u16(pk_max_u16(w[0] & 0x7FFF7FFFu, w[1] & 0x7FFF7FFFu),
                          pk_max_u16(w[2] & 0x7FFF7FFFu, w[3] & 0x7FFF7FFFu));
  uint32_t lo = m & 0xFFFFu, hi = m >> 16;
  return lo > hi ? lo : hi;
}

__device__ __forceinline__ uint32_t pk_min_f16(uint32_t a, uint32_t b) {
  uint32_t d;
  asm("v_pk_min_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ uint32_t pk_max_f16(uint32_t a, uint32_t b) {
  uint32_t d;
  asm("v_pk_max_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// torch.clamp(x, -c, c) on a packed pair; c2 = the bound in both halves (c >= 0, not NaN).  v_pk_min / max return the
// other operand for a NaN input, torch keeps the NaN: `keep_nan` restores it (only on the rare path that saw a NaN).
__device__ __forceinline__ uint32_t pk_clamp_f16(uint32_t w, uint32_t c2, bool keep_nan) {
  const uint32_t r = pk_min_f16(pk_max_f16(w, c2 ^ 0x80008000u), c2);
  if (!keep_nan) return r;
  const uint32_t nanm = pk_ashr_i16(pk_sub_u16(0x7C007C00u, w & 0x7FFF7FFFu), 15);   // 0xFFFF where the half is a NaN
  return (r & ~nanm) | (w & nanm);
}

// dual format: max|x| over x <= 0 and over x > 0 separately; NaN belongs to neither
__device__ __forceinline__ uint32_t vec_absmax16_dual(const u32x4& w, uint32_t& mneg, uint32_t& mpos) {
  uint32_t mn = 0, mp = 0, any_nan = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    uint32_t a = w[k] & 0x7FFF7FFFu;
    uint32_t nanm = pk_ashr_i16(pk_sub_u16(0x7C007C00u, a), 15);   // 0xFFFF where a > 0x7C00
    a &= ~nanm;
    any_nan |= nanm;
    uint32_t sm = pk_ashr_i16(w[k], 15);                            // 0xFFFF where the sign bit is set
    mn = pk_max_u16(mn, a & sm);
    mp = pk_max_u16(mp, a & ~sm);
  }
  uint32_t lo = mn & 0xFFFFu, hi = mn >> 16;
  mneg = lo > hi ? lo : hi;
  lo = mp & 0xFFFFu; hi = mp >> 16;
  mpos = lo > hi ? lo : hi;
  return any_nan;
}

// The same two maxima at two instructions per packed pair instead of ten, for inputs without NaN: as UNSIGNED 16-bit
// integers every negative value beats every positive one and grows with its magnitude, as SIGNED integers every negative
// value loses against +0 - so `un` = unsigned maximum and `sp` = signed maximum of the raw patterns carry max|x| over
// x < 0 and over x > 0.  A NaN (|pattern| > 0x7C00) rides along as the largest value of its side: callers test the
// reduced maxima for > 0x7C00 and, wavefront-uniformly, redo the row with vec_absmax16_dual (rare).
__device__ __forceinline__ void dual_max_acc(const u32x4& w, uint32_t& un, uint32_t& sp) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    un = pk_max_u16(un, w[k]);
    sp = pk_max_i16(sp, w[k]);
  }
}
__device__ __forceinline__ void dual_max_finish(uint32_t un, uint32_t sp, uint32_t& mneg, uint32_t& mpos) {
  const uint32_t ul = un & 0xFFFFu, uh = un >> 16;
  const uint32_t u = ul > uh ? ul : uh;
  mneg = (u & 0x8000u) ? (u & 0x7FFFu) : 0u;
  const int sl = (int)(sp << 16) >> 16, sh = (int)sp >> 16;
  const int sm = sl > sh ? sl : sh;
  mpos = sm > 0 ? (uint32_t)sm : 0u;
}
__device__ __forceinline__ bool dual_max_has_nan(uint32_t mneg, uint32_t mpos) {
  return __builtin_amdgcn_ballot_w64(mneg > 0x7C00u || mpos > 0x7C00u) != 0;
}

struct RowScale16 {
  float sf;         // (float)scale
  float inv;        // 1 / scale = inv + inv_lo, inv = the fp32 nearest (one Newton step on v_rcp_f32), inv_lo the rest;
  float inv_lo;     // both 0 for a zero scale: x / 0 -> level 0 -> +0
  uint32_t s16x2;
};

__device__ __forceinline__ RowScale16 row_scale16(uint32_t amax_bits, float g, float inv_g) {
  RowScale16 r;
  uint32_t sb = scale_bits_f16(amax_bits, g, inv_g);
  r.sf = h2f(sb);
  const float r0 = __builtin_amdgcn_rcpf(r.sf);
  const float r1 = __builtin_fmaf(__builtin_fmaf(-r.sf, r0, 1.0f), r0, r0);
  const float lo = __builtin_fmaf(-r.sf, r1, 1.0f) * r1;          // 1/s - r1, to ~2^-46
  const bool zero = r.sf == 0.0f;
  r.inv = zero ? 0.0f : r1;
  r.inv_lo = zero ? 0.0f : lo;
  r.s16x2 = sb | (sb << 16);
  return r;
}

__device__ __forceinline__ void scale_nan_if_not_finite(RowScale16& s) {
  if (!(s.sf < __builtin_inff())) s.s16x2 = 0x7E007E00u;
}

// dual format: a non-finite scale on either side poisons the whole row (0 * inf = NaN
// is added to every element by the reference's `q_neg*s_neg + q_pos*s_pos`)
__device__ __forceinline__ void dual_poison(RowScale16& n, RowScale16& p) {
  bool bad = !(fabsf(n.sf) < __builtin_inff()) || !(fabsf(p.sf) < __builtin_inff());
  if (bad) {
    n.sf = p.sf = __builtin_nanf("");
    n.s16x2 = p.s16x2 = 0x7E007E00u;
  }
}

// The bucket -> level table, prebuilt on the host (immutable, cached per table pair) and handed over BY VALUE in
// the kernel arguments: no device-side global state, and a workgroup pays a few 2-byte stores per lane instead of
// evaluating the closed form.  Kernel arguments are limited to 4 KiB, the widest table (int_neg / e2m3_pos, buckets
// of 32 patterns) has 2 x 1024 entries - but every table is constant beyond its largest rounding boundary (the
// largest level, for every finite pattern up to the reach limit) and zero for the inf / NaN patterns, so only the
// prefix of each sign half travels: <= 1216 entries for the widest pair.
constexpr int kLutArgEntries = 1280;
struct Lut16Tab {
  uint16_t e[kLutArgEntries];   // positive-half prefix, then negative-half prefix
  uint16_t n_pos, n_neg;        // prefix lengths (buckets)
  uint16_t fill_pos, fill_neg;  // every finite bucket beyond the prefix
};

// expand the by-value table into the workgroup's LDS image [positive half | negative half]
__device__ __forceinline__ void lut16_stage(uint16_t* lut, const Lut16Tab& tab, int shift) {
  const int nh = 1 << (15 - shift), nan0 = 0x7C00 >> shift;
  if (tab.n_pos == 0xFFFFu) {   // small table, stored whole: one plain copy (the common case, kept as cheap as it was)
    for (int i = threadIdx.x; i < 2 * nh; i += blockDim.x) lut[i] = tab.e[i];
    return;
  }
  for (int i = threadIdx.x; i < 2 * nh; i += blockDim.x) {
    const bool neg = i >= nh;
    const int j = neg ? i - nh : i;
    const int np = neg ? tab.n_neg : tab.n_pos, base = neg ? tab.n_pos : 0;
    const uint16_t fill = neg ? tab.fill_neg : tab.fill_pos;
    lut[i] = j < np ? tab.e[base + j] : (j < nan0 ? fill : (uint16_t)0);
  }
}

// host: compress a full image (2^(16-shift) entries); false when it does not have the constant-tail structure or
// the prefixes do not fit
inline bool lut16_compress(const uint16_t* full, int shift, Lut16Tab* out) {
  const int nh = 1 << (15 - shift), nan0 = 0x7C00 >> shift;
  if (2 * nh <= kLutArgEntries) {   // fits whole
    for (int i = 0; i < kLutArgEntries; ++i) out->e[i] = i < 2 * nh ? full[i] : 0;
    out->n_pos = out->n_neg = 0xFFFFu;
    out->fill_pos = out->fill_neg = 0;
    return true;
  }
  int np[2];
  uint16_t fill[2];
  for (int half = 0; half < 2; ++half) {
    const uint16_t* f = full + half * nh;
    for (int j = nan0; j < nh; ++j)
      if (f[j] != 0) return false;
    fill[half] = f[nan0 - 1];
    int n = nan0;
    while (n > 0 && f[n - 1] == fill[half]) --n;
    np[half] = n;
  }
  if (np[0] + np[1] > kLutArgEntries) return false;
  for (int i = 0; i < kLutArgEntries; ++i) out->e[i] = 0;
  for (int j = 0; j < np[0]; ++j) out->e[j] = full[j];
  for (int j = 0; j < np[1]; ++j) out->e[np[0] + j] = full[nh + j];
  out->n_pos = (uint16_t)np[0];
  out->n_neg = (uint16_t)np[1];
  out->fill_pos = fill[0];
  out->fill_neg = fill[1];
  return true;
}

inline void lut16_build_host(uint16_t* lut, const Lut16Args& a) {
  const int n = 1 << (16 - a.shift);
  for (int i = 0; i < n; ++i) {
    uint32_t u = (uint32_t)i << a.shift;
    bool neg = (u >> 15) != 0;
    float r = h2f(u & 0x7FFFu);
    float qm = quant_mag(r, 0u, neg ? a.fneg : a.fpos);
    uint32_t qb = f2h(qm);
    lut[i] = (uint16_t)((neg && qm != 0.0f) ? (qb | 0x8000u) : qb);
  }
}

// ---------------------------------------------------------------------------------
// GELU(tanh) of an fp16 tensor as torch computes it (fp32 arithmetic, one rounding to fp16): the FFN's activation between
// fc1 and fc2's input quantizer (tr/basic_var.py:120), fused into the fc1 GEMM's epilogue (fpq_gemm_fp4.h, GemmFc1) and into
// the stand-alone dual quantizer (rows16_lut_subwave_kernel<..., GELU>).
// torch: aten/src/ATen/native/cuda/ActivationGeluKernel.cu, GeluCUDAKernelImpl, approximate == tanh (opmath = float):
//   kBeta = M_SQRT2 * M_2_SQRTPI * 0.5, kKappa = 0.044715; x_cube = x * x * x; inner = kBeta * (x + kKappa * x_cube);
//   0.5 * x * (1 + tanh(inner))
// FPQ_GELU_FMA: the compiler that built torch contracts x + kKappa * x_cube into one fma (hipcc's default for HIP sources);
// this library is built with contraction off, so the fma is spelled out.
#ifndef FPQ_GELU_FMA
#define FPQ_GELU_FMA 1
#endif
__device__ __forceinline__ float tanh_devlib(float x) {   // __ocml_tanh_f32 (ROCm device library), restated
  const float y = __builtin_fabsf(x);
  float z;
  if (y < 0.625f) {
    const float y2 = x * x;
    float p = __builtin_fmaf(y2, -0x1.758e7ap-8f, 0x1.521192p-6f);
    p = __builtin_fmaf(y2, p, -0x1.b8389cp-5f);
    p = __builtin_fmaf(y2, p, 0x1.110704p-3f);
    p = __builtin_fmaf(y2, p, -0x1.555532p-2f);
    z = __builtin_fmaf(y2, y * p, y);
  } else {
    const float t = __builtin_expf(2.0f * y);
    z = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(t + 1.0f), 1.0f);
  }
  return __builtin_copysignf(z, x);
}
__device__ __forceinline__ float gelu_tanh_like_torch(float x) {
  const float kBeta = (float)(1.41421356237309504880 * 1.12837916709551257390 * 0.5), kKappa = 0.044715f;
  const float x3 = x * x * x;
#if FPQ_GELU_FMA
  const float inner = kBeta * __builtin_fmaf(kKappa, x3, x);
#else
  const float inner = kBeta * (x + kKappa * x3);
#endif
  return 0.5f * x * (1.0f + tanh_devlib(inner));
}

// The form the epilogue runs (9 vector instructions instead of ~30): gelu(x) = x w, w = (1 + tanh(u)) / 2 = 1 / (1 + 2^m),
// m = -2 log2(e) u = x (c0 + c1 x^2); `(w - 0.5) + 0.5` snaps w to the grid torch's own 1 + tanh(u) lives on (its tanh is an
// fp32 number just below 1 for the deep negatives, where 1 + tanh cancels: without the snap the more exact w is up to 2 fp16
// ulps away from what torch returns on 7 inputs in [-5.2, -4.7]).  tools/probe/gelu_probe.hip runs eight candidate forms over
// all 65536 fp16 inputs against torch on the GPU (profiles/r05_gelu_probe.txt): the torch-order form above is bit-equal
// to torch on every input; this one differs on 5 inputs, by one ulp each, NaN exactly where torch has NaN (NaN, -inf).
__device__ __forceinline__ float gelu_tanh_fast(float x) {
  const float kBeta = (float)(1.41421356237309504880 * 1.12837916709551257390 * 0.5), kKappa = 0.044715f;
  const float c0 = -2.8853900817779268f * kBeta, c1 = c0 * kKappa;   // float arithmetic, as in the probe
  const float m = x * __builtin_fmaf(x * x, c1, c0);
  const float w = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(m));
  return x * ((w - 0.5f) + 0.5f);
}

// ---------------------------------------------------------------------------------
// rows of 1..64 lanes x 16 bytes inside one wavefront (per-group 128, KV c=64, ...).
// A workgroup owns TILE = 256*U consecutive vectors (U*4 KiB, contiguous in HBM);
// workgroups are dispatched in address order, so the chip sweeps the tensor front
// to back.  All U loads are issued before the table is staged and the barrier.
// ---------------------------------------------------------------------------------
// HW4: E2M1 levels from the conversion hardware (above) - no table, no LDS, no barrier.
// CLIP (dual format per group): the reference's global clamp to +-strength * max|x| in front of the quantizer, the
// maximum read from a device scalar (fpq_absmax) - two packed instructions per pair on the fast path instead of the
// generic kernel (0.23 -> of 8 TB/s, profiles/r03_survey_shapes.txt).  Group maxima are taken on the raw words (a NaN
// is still seen) and clamped afterwards (clamping is monotonic).
// HW6 (1: E2M3, 2: E3M2; U = 4: a lane's four vectors are the 32 values of one FP6 conversion, each with the scale of its own
// row): levels from the FP6 conversion hardware (fp6_levels_hw32) - no table, no LDS, no barrier.
// GELU (dual format per group; round 5): x is the fc1 OUTPUT - every element goes through GELU(tanh) (gelu_tanh_fast above: fp32, one
// rounding to fp16, within one fp16 ulp of torch's on every input) before the quantizer sees it: `fc2.act_quant(act(y))` of the
// reference's FFN (tr/basic_var.py:120-121, tr/quant_utils.py:991) in one pass over y instead of two.
template <int LPR, bool DUAL, int U, bool TAB_ARG, bool NTL = true, bool NTS = true, bool HW4 = false, bool CLIP = false, int HW6 = 0,
          bool GELU = false>
__global__ __launch_bounds__(kBlock) void rows16_lut_subwave_kernel(const u32x4* __restrict__ x,
                                                                   u32x4* __restrict__ out, int64_t n_vec,
                                                                   Lut16Args a, Lut16Tab tab) {
  static_assert(HW6 == 0 || (U == 4 && !DUAL && !HW4 && !CLIP), "hardware FP6 levels: symmetric tables, four vectors per lane");
  uint16_t* lut = nullptr;
  if constexpr (!HW4 && HW6 == 0) {
    __shared__ __attribute__((aligned(16))) uint16_t lut_s[kLutLdsEntries];   // static: a compile-time LDS address (a dynamic base is not folded into the ds_read offsets)
    lut = lut_s;
  }
  static_assert(!CLIP || DUAL, "the global clamp belongs to the dual format");
  const int64_t tiles = (n_vec + (int64_t)kBlock * U - 1) / ((int64_t)kBlock * U);
  uint32_t clip2 = 0;
  bool clip_nan = false;
  if constexpr (CLIP) {   // c = half(strength * max|x|), as the reference's fp16 multiply
    const uint32_t cb = f2h(a.clip_strength * h2f(*(const uint16_t*)a.clip_absmax));
    clip_nan = (cb & 0x7FFFu) > 0x7C00u;   // a NaN in the tensor: clamp(x, NaN, NaN) is NaN everywhere -> every output +0
    clip2 = cb | (cb << 16);
  }
  bool first = true;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t v0 = tile * ((int64_t)kBlock * U) + threadIdx.x;
    u32x4 raw[U];
    bool live[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int64_t v = v0 + u * kBlock;
      live[u] = v < n_vec;
      raw[u] = live[u] ? (NTL ? __builtin_nontemporal_load(x + v) : x[v]) : u32x4{0, 0, 0, 0};
    }
    if constexpr (GELU) {
      static_assert(DUAL && !CLIP && HW6 == 0, "the fused activation belongs to fc2's dual-format input quantizer");
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int k = 0; k < 4; ++k) raw[u][k] = f2h2(gelu_tanh_fast(h2f(raw[u][k] & 0xFFFFu)), gelu_tanh_fast(h2f(raw[u][k] >> 16)));
        if (a.gelu_out && live[u]) __builtin_nontemporal_store(raw[u], (u32x4*)a.gelu_out + v0 + u * kBlock);
      }
    }
    if constexpr (HW6 != 0) {
      uint32_t q6[16], lv6[16], sc6[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t m = row_max_dpp<LPR>(vec_absmax16(raw[u]));
        RowScale16 s = row_scale16(m, a.fpos.gmax, a.inv_gpos);
        scale_nan_if_not_finite(s);
        sc6[u] = s.s16x2;
#pragma unroll
        for (int k = 0; k < 4; ++k) q6[4 * u + k] = div_pair16(raw[u][k], s.inv, s.inv_lo, s.inv, s.inv_lo);
      }
      fp6_levels_hw32<HW6 == 2>(q6, lv6);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const u32x4 o = {pk_fma0_f16(lv6[4 * u], sc6[u]), pk_fma0_f16(lv6[4 * u + 1], sc6[u]), pk_fma0_f16(lv6[4 * u + 2], sc6[u]),
                         pk_fma0_f16(lv6[4 * u + 3], sc6[u])};
        if (live[u]) {
          if (NTS) __builtin_nontemporal_store(o, out + v0 + u * kBlock);
          else out[v0 + u * kBlock] = o;
        }
      }
      continue;
    }
    if (first && !HW4) {
      if (TAB_ARG) {
        lut16_stage(lut, tab, a.shift);
      } else {
        lut16_fill(lut, a);
      }
      __syncthreads();
      first = false;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      u32x4 o;
      if constexpr (DUAL) {
        uint32_t un = 0, sg = 0, mn, mp;
        dual_max_acc(raw[u], un, sg);
        dual_max_finish(un, sg, mn, mp);
        mn = row_max_dpp<LPR>(mn);
        mp = row_max_dpp<LPR>(mp);
        u32x4 xw = raw[u];
        const bool has_nan = dual_max_has_nan(mn, mp);
        if constexpr (CLIP) {
          const uint32_t cb = clip2 & 0xFFFFu;
          mn = mn < cb ? mn : cb;
          mp = mp < cb ? mp : cb;
#pragma unroll
          for (int k = 0; k < 4; ++k) xw[k] = pk_clamp_f16(xw[k], clip2, has_nan);
        }
        if (__builtin_expect(has_nan, 0)) {   // a NaN somewhere in the wavefront's vectors: the exact rule
          if (vec_absmax16_dual(xw, mn, mp) && a.nan_flag) atomicOr(a.nan_flag, 1u);
          mn = row_max_dpp<LPR>(mn);
          mp = row_max_dpp<LPR>(mp);
        }
        RowScale16 sn = row_scale16(mn, a.fneg.gmax, a.inv_gneg), sp = row_scale16(mp, a.fpos.gmax, a.inv_gpos);
        dual_poison(sn, sp);
        o = quant_vec16<true>(xw, lut, a.shift, sn.inv, sn.inv_lo, sn.s16x2, sp.inv, sp.inv_lo, sp.s16x2);
        if (CLIP && clip_nan) o = u32x4{0, 0, 0, 0};
      } else {
        uint32_t m = row_max_dpp<LPR>(vec_absmax16(raw[u]));
        RowScale16 s = row_scale16(m, a.fpos.gmax, a.inv_gpos);
        if constexpr (HW4) scale_nan_if_not_finite(s);
        o = quant_vec16<false, HW4>(raw[u], lut, a.shift, s.inv, s.inv_lo, s.s16x2, 0.f, 0.f, 0u);
      }
      if (live[u]) {
        if (NTS) __builtin_nontemporal_store(o, out + v0 + u * kBlock);
        else out[v0 + u * kBlock] = o;
      }
    }
  }
}

// Several fp16 tensors, ONE launch (fpq_quant_rows_multi): the same rows as rows16_lut_subwave_kernel, the tensors'
// descriptors travel by value in the kernel arguments (no device-side table, no copy), blockIdx.y selects the tensor,
// workgroups beyond a shorter tensor's end exit at once.  For the places where the reference quantizes independent
// tensors back to back - the cached K and V of a step (tr/basic_var.py:192-200), the samples of a format search.
constexpr int kMaxMulti = 8;
struct Multi16 {
  const u32x4* x[kMaxMulti];
  u32x4* out[kMaxMulti];
  int64_t n_vec[kMaxMulti];
};

template <int LPR, int U>
__global__ __launch_bounds__(kBlock) void rows16_lut_multi_kernel(Multi16 m, Lut16Args a, Lut16Tab tab) {
  __shared__ __attribute__((aligned(16))) uint16_t lut[kLutLdsEntries];
  const int64_t n_vec = m.n_vec[blockIdx.y];
  const int64_t v0 = (int64_t)blockIdx.x * (kBlock * U) + threadIdx.x;
  if ((int64_t)blockIdx.x * (kBlock * U) >= n_vec) return;
  const u32x4* __restrict__ x = m.x[blockIdx.y];
  u32x4* __restrict__ out = m.out[blockIdx.y];
  u32x4 raw[U];
  bool live[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t v = v0 + u * kBlock;
    live[u] = v < n_vec;
    raw[u] = live[u] ? __builtin_nontemporal_load(x + v) : u32x4{0, 0, 0, 0};
  }
  lut16_stage(lut, tab, a.shift);
  __syncthreads();
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t mx = row_max_dpp<LPR>(vec_absmax16(raw[u]));
    const RowScale16 s = row_scale16(mx, a.fpos.gmax, a.inv_gpos);
    const u32x4 o = quant_vec16<false>(raw[u], lut, a.shift, s.inv, s.inv_lo, s.s16x2, 0.f, 0.f, 0u);
    if (live[u]) __builtin_nontemporal_store(o, out + v0 + u * kBlock);
  }
}

// ---------------------------------------------------------------------------------
// KV-cache step (the reference's tr/basic_var.py:186-209): ONE launch
//   (a) quantizes, in place, the cache entries the previous step appended (their first - and, the quantizer being
//       idempotent on its own output, only - quantization), rows of 8 * LPR halves, and
//   (b) copies the new k / v rows (any batch / token pitch, e.g. views of the fused qkv output) behind them.
// The cache is fp16 [2 (k, v), batch, max_len, row_elems]; blockIdx.z = k / v, blockIdx.y = batch entry,
// blockIdx.x < q_tiles: job (a), else job (b).  Every branch is uniform over the workgroup.
// ---------------------------------------------------------------------------------
struct KvStepArgs {
  u32x4* cache;
  int64_t slab_vec;        // 16-byte vectors per (k|v, batch entry) slab = max_len * row_vec
  int row_vec;             // vectors per token row
  int batch;
  int64_t q_first_vec;     // first vector of the rows to quantize (inside a slab)
  int64_t q_vecs;
  int q_tiles;
  const uint16_t* src[2];  // new k, new v: [batch, n_new, row_elems], rows contiguous
  int64_t src_batch_pitch, src_token_pitch;   // in elements
  int64_t new_first_vec;
  int64_t new_vecs;
};

template <int LPR, int U>
__global__ __launch_bounds__(kBlock) void kv16_step_kernel(KvStepArgs k, Lut16Args a, Lut16Tab tab) {
  __shared__ __attribute__((aligned(16))) uint16_t lut[kLutLdsEntries];   // static: a compile-time LDS address (a dynamic base is not folded into the ds_read offsets)
  const int kv = blockIdx.z, b = blockIdx.y;
  u32x4* slab = k.cache + ((int64_t)kv * k.batch + b) * k.slab_vec;
  if ((int)blockIdx.x < k.q_tiles) {
    u32x4* p = slab + k.q_first_vec;
    const int64_t v0 = (int64_t)blockIdx.x * (kBlock * U) + threadIdx.x;
    u32x4 raw[U];
    bool live[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t v = v0 + u * kBlock;
      live[u] = v < k.q_vecs;     // q_vecs is a multiple of LPR: a row is live or dead as a whole
      raw[u] = live[u] ? p[v] : u32x4{0, 0, 0, 0};
    }
    {
      lut16_stage(lut, tab, a.shift);
      __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t m = row_max_dpp<LPR>(vec_absmax16(raw[u]));
      const RowScale16 s = row_scale16(m, a.fpos.gmax, a.inv_gpos);
      const u32x4 o = quant_vec16<false>(raw[u], lut, a.shift, s.inv, s.inv_lo, s.s16x2, 0.f, 0.f, 0u);
      if (live[u]) p[v0 + u * kBlock] = o;
    }
  } else {
    const uint16_t* s = k.src[kv] + (int64_t)b * k.src_batch_pitch;
    u32x4* d = slab + k.new_first_vec;
    const int64_t v0 = (int64_t)((int)blockIdx.x - k.q_tiles) * (kBlock * U) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t v = v0 + u * kBlock;
      if (v < k.new_vecs) {
        const int64_t l = v / k.row_vec;
        const int c = (int)(v - l * k.row_vec);
        d[v] = *(const u32x4*)(s + l * k.src_token_pitch + c * 8);
      }
    }
  }
}

// Same rows, but LPR lanes own a row of 2*LPR vectors (two 16-byte vectors per lane, LPR*16 bytes
// apart): the per-row work - cross-lane max, scale, reciprocal - is paid once per 16 elements of a
// lane instead of once per 8.  Every load / store instruction still touches whole 128-byte lines
// (LPR = 8: eight lanes x 16 B).  Used for the 128-element groups.
template <int LPR, bool DUAL, bool TAB_ARG>
__global__ __launch_bounds__(kBlock) void rows16_lut_pair_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ out,
                                                                int64_t n_vec, Lut16Args a, Lut16Tab tab) {
  __shared__ __attribute__((aligned(16))) uint16_t lut[kLutLdsEntries];   // static: a compile-time LDS address (a dynamic base is not folded into the ds_read offsets)
  constexpr int64_t kTile = (int64_t)kBlock * 2;
  const int64_t tiles = (n_vec + kTile - 1) / kTile;
  const int in_row = threadIdx.x % LPR, row_in_tile = threadIdx.x / LPR;
  bool first = true;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t va = tile * kTile + (int64_t)row_in_tile * (2 * LPR) + in_row;
    const int64_t vb = va + LPR;
    const bool live = va < n_vec;   // n_vec is a multiple of 2*LPR: both vectors live or both dead
    u32x4 ra = live ? __builtin_nontemporal_load(x + va) : u32x4{0, 0, 0, 0};
    u32x4 rb = live ? __builtin_nontemporal_load(x + vb) : u32x4{0, 0, 0, 0};
    if (first) {
      if (TAB_ARG) {
        lut16_stage(lut, tab, a.shift);
      } else {
        lut16_fill(lut, a);
      }
      __syncthreads();
      first = false;
    }
    u32x4 oa, ob;
    if (DUAL) {
      uint32_t un = 0, sg = 0, mn, mp;
      dual_max_acc(ra, un, sg);
      dual_max_acc(rb, un, sg);
      dual_max_finish(un, sg, mn, mp);
      mn = row_max_dpp<LPR>(mn);
      mp = row_max_dpp<LPR>(mp);
      if (__builtin_expect(dual_max_has_nan(mn, mp), 0)) {   // the exact rule (NaN belongs to neither side)
        uint32_t mn2, mp2;
        const uint32_t nan_any = vec_absmax16_dual(ra, mn, mp) | vec_absmax16_dual(rb, mn2, mp2);
        if (nan_any && a.nan_flag) atomicOr(a.nan_flag, 1u);
        mn = row_max_dpp<LPR>(mn > mn2 ? mn : mn2);
        mp = row_max_dpp<LPR>(mp > mp2 ? mp : mp2);
      }
      RowScale16 sn = row_scale16(mn, a.fneg.gmax, a.inv_gneg), sp = row_scale16(mp, a.fpos.gmax, a.inv_gpos);
      dual_poison(sn, sp);
      oa = quant_vec16<true>(ra, lut, a.shift, sn.inv, sn.inv_lo, sn.s16x2, sp.inv, sp.inv_lo, sp.s16x2);
      ob = quant_vec16<true>(rb, lut, a.shift, sn.inv, sn.inv_lo, sn.s16x2, sp.inv, sp.inv_lo, sp.s16x2);
    } else {
      const uint32_t m1 = vec_absmax16(ra), m2 = vec_absmax16(rb);
      const uint32_t m = row_max_dpp<LPR>(m1 > m2 ? m1 : m2);
      RowScale16 s = row_scale16(m, a.fpos.gmax, a.inv_gpos);
      oa = quant_vec16<false>(ra, lut, a.shift, s.inv, s.inv_lo, s.s16x2, 0.f, 0.f, 0u);
      ob = quant_vec16<false>(rb, lut, a.shift, s.inv, s.inv_lo, s.s16x2, 0.f, 0.f, 0u);
    }
    if (live) {
      __builtin_nontemporal_store(oa, out + va);
      __builtin_nontemporal_store(ob, out + vb);
    }
  }
}

// ---------------------------------------------------------------------------------
// long fp16 rows (per-token 1920 / 7680 / 2304 / 9216, per-channel weights in fp16):
// one workgroup walks a contiguous run of rows; a row's vectors stay in registers
// (<= MAXC per lane) between the max reduction (DPP + LDS) and the rounding.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t block_max16(uint32_t v, uint32_t* sh) {
  v = row_max_dpp<64>(v);
  __syncthreads();   // sh may still be read from the previous reduction
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  uint32_t r = sh[0];
#pragma unroll
  for (int i = 1; i < kBlock / 64; ++i) r = r > sh[i] ? r : sh[i];
  return r;
}

// GELU: as in rows16_lut_subwave_kernel - x is the fc1 output, every element goes through GELU(tanh) first (the W6A6 run's
// `fc2.act_quant(act(y))` with the per-token INT-/E2M3+ quantizer, tr/quant_utils.py:614-646 bound at :930-931).
template <bool DUAL, int MAXC, bool TAB_ARG, bool GELU = false>
__global__ __launch_bounds__(kBlock) void rows16_lut_block_kernel(const uint16_t* __restrict__ x,
                                                                 uint16_t* __restrict__ out, int64_t rows,
                                                                 int64_t cols, int64_t rows_per_block, Lut16Args a,
                                                                 Lut16Tab tab) {
  __shared__ __attribute__((aligned(16))) uint16_t lut[kLutLdsEntries];   // static: a compile-time LDS address (a dynamic base is not folded into the ds_read offsets)
  __shared__ uint32_t sh[kBlock / 64];
  const int64_t vec_per_row = cols >> 3;
  // rows blockIdx.x, blockIdx.x + gridDim.x, ...: the workgroups in flight sweep the tensor front to back together
  // (consecutive rows per workgroup measured 0.68 of 8 TB/s on [65536 x 7680], this order 0.79-0.81)
  (void)rows_per_block;
  bool first = true;
  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    const u32x4* xr = (const u32x4*)(x + row * cols);
    u32x4* orow = (u32x4*)(out + row * cols);
    u32x4 raw[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      int64_t v = (int64_t)c * kBlock + threadIdx.x;
      raw[c] = (v < vec_per_row) ? __builtin_nontemporal_load(xr + v) : u32x4{0, 0, 0, 0};
    }
    if constexpr (GELU) {
      static_assert(DUAL, "the fused activation belongs to fc2's dual-format input quantizer");
#pragma unroll
      for (int c = 0; c < MAXC; ++c) {
        const int64_t v = (int64_t)c * kBlock + threadIdx.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) raw[c][k] = f2h2(gelu_tanh_fast(h2f(raw[c][k] & 0xFFFFu)), gelu_tanh_fast(h2f(raw[c][k] >> 16)));
        if (a.gelu_out && v < vec_per_row) __builtin_nontemporal_store(raw[c], (u32x4*)((uint16_t*)a.gelu_out + row * cols) + v);
      }
    }
    if (first) {
      if (TAB_ARG) {
        lut16_stage(lut, tab, a.shift);
      } else {
        lut16_fill(lut, a);
      }
      first = false;   // the barriers inside block_max16 order the table before its first use
    }
    RowScale16 sn, sp;
    if (DUAL) {
      uint32_t mn = 0, mp = 0;
      auto block_reduce = [&]() {   // both maxima through one LDS exchange (one barrier pair instead of two)
        mn = row_max_dpp<64>(mn);
        mp = row_max_dpp<64>(mp);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = (mn << 16) | mp;
        __syncthreads();
        mn = 0;
        mp = 0;
#pragma unroll
        for (int i = 0; i < kBlock / 64; ++i) {
          const uint32_t w = sh[i];
          mn = mn > (w >> 16) ? mn : (w >> 16);
          mp = mp > (w & 0xFFFFu) ? mp : (w & 0xFFFFu);
        }
      };
      {
        uint32_t un = 0, sg = 0;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) dual_max_acc(raw[c], un, sg);
        dual_max_finish(un, sg, mn, mp);
      }
      block_reduce();
      if (__builtin_expect(mn > 0x7C00u || mp > 0x7C00u, 0)) {   // a NaN in the row (the same for every thread): the exact rule
        mn = 0;
        mp = 0;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
          uint32_t a1, b1;
          if (vec_absmax16_dual(raw[c], a1, b1) && a.nan_flag) atomicOr(a.nan_flag, 1u);
          mn = mn > a1 ? mn : a1;
          mp = mp > b1 ? mp : b1;
        }
        block_reduce();
      }
      sn = row_scale16(mn, a.fneg.gmax, a.inv_gneg);
      sp = row_scale16(mp, a.fpos.gmax, a.inv_gpos);
      dual_poison(sn, sp);
    } else {
      uint32_t m = 0;
#pragma unroll
      for (int c = 0; c < MAXC; ++c) {
        uint32_t t = vec_absmax16(raw[c]);
        m = m > t ? m : t;
      }
      m = block_max16(m, sh);
      sn = row_scale16(m, a.fpos.gmax, a.inv_gpos);
      sp = sn;
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      int64_t v = (int64_t)c * kBlock + threadIdx.x;
      if (v < vec_per_row) {
        u32x4 o = quant_vec16<DUAL>(raw[c], lut, a.shift, sn.inv, sn.inv_lo, sn.s16x2, sp.inv, sp.inv_lo, sp.s16x2);
        __builtin_nontemporal_store(o, orow + v);
      }
    }
  }
}

// Medium rows (<= 64*MAXC vectors, e.g. per-token 1920 = 240 vectors): one WAVEFRONT per row, the
// row max by DPP/shuffles only - no LDS, no barrier in the row loop.
// HW6 (1: E2M3, 2: E3M2; rows of 193 .. 256 vectors = exactly 4 per lane, symmetric table): the levels from the FP6
// conversion hardware (fp6_levels_hw32) - no table, no LDS, no barrier.
template <bool DUAL, int MAXC, bool TAB_ARG, int HW6 = 0>
__global__ __launch_bounds__(kBlock) void rows16_lut_wave_kernel(const uint16_t* __restrict__ x,
                                                                uint16_t* __restrict__ out, int64_t rows, int64_t cols,
                                                                Lut16Args a, Lut16Tab tab) {
  static_assert(HW6 == 0 || (!DUAL && (MAXC == 4 || MAXC == 5)), "hardware FP6 levels: symmetric tables, 32 (+ 8) elements per lane");
  uint16_t* lut = nullptr;
  if constexpr (HW6 == 0) {
    __shared__ __attribute__((aligned(16))) uint16_t lut_s[kLutLdsEntries];   // static: a compile-time LDS address (a dynamic base is not folded into the ds_read offsets)
    lut = lut_s;
    if (TAB_ARG) {
      lut16_stage(lut, tab, a.shift);
    } else {
      lut16_fill(lut, a);
    }
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int64_t vpr = cols >> 3;
  constexpr int R = kBlock / 64;
  for (int64_t base = (int64_t)blockIdx.x * R; base < rows; base += (int64_t)gridDim.x * R) {
    const int64_t row = base + (threadIdx.x >> 6);
    if (row >= rows) continue;   // whole wavefront skips
    const u32x4* xr = (const u32x4*)(x + row * cols);
    u32x4* orow = (u32x4*)(out + row * cols);
    u32x4 raw[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int64_t v = (int64_t)c * 64 + lane;
      raw[c] = (v < vpr) ? __builtin_nontemporal_load(xr + v) : u32x4{0, 0, 0, 0};
    }
    RowScale16 sn, sp;
    if (DUAL) {
      uint32_t mn, mp;
      {
        uint32_t un = 0, sg = 0;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) dual_max_acc(raw[c], un, sg);
        dual_max_finish(un, sg, mn, mp);
      }
      mn = row_max_dpp<64>(mn);
      mp = row_max_dpp<64>(mp);
      if (__builtin_expect(mn > 0x7C00u || mp > 0x7C00u, 0)) {   // a NaN in the row (wavefront-uniform): the exact rule
        uint32_t nan_any = 0;
        mn = 0;
        mp = 0;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
          uint32_t a1, b1;
          nan_any |= vec_absmax16_dual(raw[c], a1, b1);
          mn = mn > a1 ? mn : a1;
          mp = mp > b1 ? mp : b1;
        }
        if (nan_any && a.nan_flag) atomicOr(a.nan_flag, 1u);
        mn = row_max_dpp<64>(mn);
        mp = row_max_dpp<64>(mp);
      }
      sn = row_scale16(mn, a.fneg.gmax, a.inv_gneg);
      sp = row_scale16(mp, a.fpos.gmax, a.inv_gpos);
      dual_poison(sn, sp);
    } else {
      uint32_t m = 0;
#pragma unroll
      for (int c = 0; c < MAXC; ++c) {
        uint32_t t = vec_absmax16(raw[c]);
        m = m > t ? m : t;
      }
      m = row_max_dpp<64>(m);
      sn = row_scale16(m, a.fpos.gmax, a.inv_gpos);
      sp = sn;
    }
    if constexpr (HW6 != 0) {
      scale_nan_if_not_finite(sn);
      uint32_t q[16], lv[16];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k) q[4 * c + k] = div_pair16(raw[c][k], sn.inv, sn.inv_lo, sn.inv, sn.inv_lo);
      fp6_levels_hw32<HW6 == 2>(q, lv);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int64_t v = (int64_t)c * 64 + lane;
        if (v < vpr)
          __builtin_nontemporal_store(u32x4{pk_fma0_f16(lv[4 * c], sn.s16x2), pk_fma0_f16(lv[4 * c + 1], sn.s16x2),
                                            pk_fma0_f16(lv[4 * c + 2], sn.s16x2), pk_fma0_f16(lv[4 * c + 3], sn.s16x2)}, orow + v);
      }
      if constexpr (MAXC == 5) {   // the fifth vector of lanes 0 .. (vpr - 256): a second conversion with 8 live values
        uint32_t q5[4], l5[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) q5[k] = div_pair16(raw[4][k], sn.inv, sn.inv_lo, sn.inv, sn.inv_lo);
        fp6_levels_hw32<HW6 == 2, 4>(q5, l5);
        const int64_t v = 256 + lane;
        if (v < vpr)
          __builtin_nontemporal_store(u32x4{pk_fma0_f16(l5[0], sn.s16x2), pk_fma0_f16(l5[1], sn.s16x2), pk_fma0_f16(l5[2], sn.s16x2),
                                            pk_fma0_f16(l5[3], sn.s16x2)}, orow + v);
      }
      continue;
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int64_t v = (int64_t)c * 64 + lane;
      if (v < vpr) {
        u32x4 o = quant_vec16<DUAL>(raw[c], lut, a.shift, sn.inv, sn.inv_lo, sn.s16x2, sp.inv, sp.inv_lo, sp.s16x2);
        __builtin_nontemporal_store(o, orow + v);
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// F1: the online rotate fused in front of the per-group quant.
//
// Reference (tr/basic_var.py:263,266 under the driver's fp16 autocast):
//     x1 = matmul(half(producer * s), half(Q))            fp16 GEMM, fp32 accumulation
//     q  = fp_quant_*_per_group_cuda(x1, 4, 128)
// with Q block-diagonal, every 128x128 block = diag(D) . H128 / sqrt(128)
// (rotate_utils/rotation_utils.py:69-104, hadamard_utils.py:63-99), so per 128-chunk
//     x1 = half( c_h * FWHT128(h * D) ),   c_h = half(float32(1/sqrt(128))) = 0.08837890625   (one rounding:
//          v_fma_mixlo/hi_f16; the butterfly of fp16 inputs is nearly always exact in fp32)
// Here: h = half(x * s) (s optional), sign flip by xor, 3 butterfly stages inside the lane's
// 8 values, 4 across the 16 lanes of the group (DPP quad_perm for lane^1, lane^2; ds_swizzle
// for lane^4, lane^8 - no LDS memory is touched), fp32 throughout, one rounding to fp16, and
// the result feeds the quantizer of this file without leaving registers.  A dense fp16 GEMM
// of 2*rows*C^2 FLOPs and a 2 B/elem round trip disappear.
//
// Parity contract: the quant stage is bit-exact for the rotated values this kernel
// produces (it can emit them); the rotated values are within 1 fp16 ulp of the
// fp64-accumulated product (the reference GEMM's own summation order is unspecified).
// ---------------------------------------------------------------------------------
struct RotArgs {
  const float* smooth;   // [cols] per-channel factor (GALT s) or nullptr
  uint32_t sign[4];      // bit j set <=> D[j] = -1, j = 0..127
  float c_h;             // (float)half(float32(1/sqrt(128)))
  int64_t vec_per_row;   // cols / 8
  uint16_t* code_scales; // nullptr: `out` receives fake-quantized fp16 values.  Otherwise `out` receives packed
                         // hardware E2M1 codes (4 bytes per 8 elements), this array one fp16 scale per 128-group,
                         // and the staged table is the code table (fpq_gemm_fp4.h)
  int code_bits;         // per-token code emission: 8 = E4M3 bytes (fpq_gemm_fp8.h), 6 = dense 6-bit E2M3 (fpq_gemm_fp6.h)
  uint32_t km_rows;      // != 0: FP4 / FP6 codes go to a k-major image of this many rows (include/fpq.h) instead of row-major
  FastDiv km_gpr;        // ... groups (K steps) per row, for the kernels that walk the tensor as a flat run of groups
};

__device__ __forceinline__ float xlane_xor4(float v) {
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x101F));
}
__device__ __forceinline__ float xlane_xor8(float v) {
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x201F));
}
__device__ __forceinline__ float xlane_xor1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float xlane_xor2(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
}

// FWHT over the 128 values held by 16 lanes x 8 registers; natural (Sylvester) order.
// The values travel as 4 float pairs (t[j], t[j+4]) so that the butterflies map onto the
// packed fp32 ALU (v_pk_add_f32 / v_pk_fma_f32: two results per issue slot).
typedef float f2_t __attribute__((ext_vector_type(2)));

#ifndef FPQ_FWHT_SWZ
#define FPQ_FWHT_SWZ 0   // 1: lane ^ 1 / lane ^ 2 exchanges through ds_swizzle (LDS crossbar) instead of DPP moves (VALU)
#endif
__device__ __forceinline__ float xlane_xor1s(float v) {
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x80B1));   // quad perm [1,0,3,2]
}
__device__ __forceinline__ float xlane_xor2s(float v) {
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x804E));   // quad perm [2,3,0,1]
}
__device__ __forceinline__ f2_t xlane2(f2_t v, int which) {
  f2_t r;
  if (which == 1) { r.x = xlane_xor1(v.x); r.y = xlane_xor1(v.y); }
#if FPQ_FWHT_SWZ
  else if (which == 2) { r.x = xlane_xor2s(v.x); r.y = xlane_xor2s(v.y); }
#endif
  else if (which == 2) { r.x = xlane_xor2(v.x); r.y = xlane_xor2(v.y); }
  else if (which == 4) { r.x = xlane_xor4(v.x); r.y = xlane_xor4(v.y); }
  else { r.x = xlane_xor8(v.x); r.y = xlane_xor8(v.y); }
  return r;
}

__device__ __forceinline__ void fwht128(float (&t)[8], int lane_in_group) {
  f2_t p[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) p[j] = f2_t{t[j], t[j + 4]};
  // index bit 0 and bit 1: whole pairs against whole pairs
  { f2_t a = p[0], b = p[1]; p[0] = a + b; p[1] = a - b; a = p[2]; b = p[3]; p[2] = a + b; p[3] = a - b; }
  { f2_t a = p[0], b = p[2]; p[0] = a + b; p[2] = a - b; a = p[1]; b = p[3]; p[1] = a + b; p[3] = a - b; }
  // index bit 2: inside each pair
#pragma unroll
  for (int j = 0; j < 4; ++j) p[j] = f2_t{p[j].x + p[j].y, p[j].x - p[j].y};
  // index bits 3..6 live in the lane number: partner + sign*mine (lower lane a+b, upper lane a-b)
  const float s1 = (lane_in_group & 1) ? -1.0f : 1.0f, s2 = (lane_in_group & 2) ? -1.0f : 1.0f;
  const float s4 = (lane_in_group & 4) ? -1.0f : 1.0f, s8 = (lane_in_group & 8) ? -1.0f : 1.0f;
#pragma unroll
  for (int j = 0; j < 4; ++j) p[j] = __builtin_elementwise_fma(p[j], f2_t{s1, s1}, xlane2(p[j], 1));
#pragma unroll
  for (int j = 0; j < 4; ++j) p[j] = __builtin_elementwise_fma(p[j], f2_t{s2, s2}, xlane2(p[j], 2));
#pragma unroll
  for (int j = 0; j < 4; ++j) p[j] = __builtin_elementwise_fma(p[j], f2_t{s4, s4}, xlane2(p[j], 4));
#pragma unroll
  for (int j = 0; j < 4; ++j) p[j] = __builtin_elementwise_fma(p[j], f2_t{s8, s8}, xlane2(p[j], 8));
#pragma unroll
  for (int j = 0; j < 4; ++j) { t[j] = p[j].x; t[j + 4] = p[j].y; }
}

// half(a * b) packed from two fp32 values: the rotated value half(c_h * FWHT(...)).  Two v_mul_f32 and one
// v_cvt_pk_f16_f32 (2 x 2.5 + 4.6 cycles of the vector pipe) instead of v_fma_mixlo/hi_f16 (2 x 8.2,
// profiles/r03_valu_issue_cost.txt).  The product is rounded to fp32 before it is rounded to fp16: the two roundings
// differ from one in ~2^-13 of the values, by one fp16 ulp; no torch op sequence is mirrored here (the reference's GEMM
// rounds its own fp32 accumulator) and the contract - within half an ulp (1.001) + the accumulation bound of the exact
// product, quantization exact on the values produced - holds as before.
__device__ __forceinline__ uint32_t mul2_to_h2(float lo, float hi, float b) { return f2h2(lo * b, hi * b); }

// D = s * (float)half(w) + (float)half(pw), lo / hi halves: both fp16 operands are widened by the instruction itself
__device__ __forceinline__ float fmix_hsh_lo(uint32_t w, float s, uint32_t pw) {
  float d;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(w), "v"(s), "v"(pw));
  return d;
}
__device__ __forceinline__ float fmix_hsh_hi(uint32_t w, float s, uint32_t pw) {
  float d;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(w), "v"(s), "v"(pw));
  return d;
}

// FWHT over the 128 fp16 values held by 16 lanes x 4 packed registers (signs already applied), natural (Sylvester)
// order, fp32 results (not yet scaled by c_h).  N independent groups are transformed stage by stage (stage-major
// source order: N dependency chains for the in-order issue to interleave).  The butterfly stages commute, so the
// first one is the lane-bit-0 stage, taken straight from the PACKED halves: one DPP move fetches two partner
// values, and v_fma_mix_f32 widens mine and the partner's while it adds (exact) - no separate fp16 -> fp32
// conversion exists.  Then the three in-register stages on float pairs, then lane bits 1, 2, 3 as in fwht128.
template <int N>
__device__ __forceinline__ void fwht128_h_n(const u32x4 (&w)[N], float (&t)[N][8], int n, int lane_in_group) {
  const float s1 = (lane_in_group & 1) ? -1.0f : 1.0f, s2 = (lane_in_group & 2) ? -1.0f : 1.0f;
  const float s4 = (lane_in_group & 4) ? -1.0f : 1.0f, s8 = (lane_in_group & 8) ? -1.0f : 1.0f;
  f2_t p[N][4];
#pragma unroll
  for (int q = 0; q < N; ++q) {
    if (q >= n) continue;
    uint32_t pw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#if FPQ_FWHT_SWZ
      pw[k] = (uint32_t)__builtin_amdgcn_ds_swizzle((int)w[q][k], 0x80B1);
#else
      pw[k] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[q][k], 0xB1, 0xF, 0xF, true);
#endif
#pragma unroll
    for (int k = 0; k < 4; ++k) {   // element 2k, 2k+1 -> pair register (k & 1 ? .. ): t index j pairs with j + 4
      const float lo = fmix_hsh_lo(w[q][k], s1, pw[k]), hi = fmix_hsh_hi(w[q][k], s1, pw[k]);
      // elements 2k and 2k+1: p[j] = {t[j], t[j + 4]}
      if (k < 2) {
        p[q][2 * k].x = lo;
        p[q][2 * k + 1].x = hi;
      } else {
        p[q][2 * k - 4].y = lo;
        p[q][2 * k - 3].y = hi;
      }
    }
  }
  // index bit 0 and bit 1: whole pairs against whole pairs; index bit 2: inside each pair
#pragma unroll
  for (int q = 0; q < N; ++q) {
    if (q >= n) continue;
    f2_t a = p[q][0], b = p[q][1];
    p[q][0] = a + b;
    p[q][1] = a - b;
    a = p[q][2];
    b = p[q][3];
    p[q][2] = a + b;
    p[q][3] = a - b;
  }
#pragma unroll
  for (int q = 0; q < N; ++q) {
    if (q >= n) continue;
    f2_t a = p[q][0], b = p[q][2];
    p[q][0] = a + b;
    p[q][2] = a - b;
    a = p[q][1];
    b = p[q][3];
    p[q][1] = a + b;
    p[q][3] = a - b;
  }
#pragma unroll
  for (int q = 0; q < N; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (q < n) p[q][j] = f2_t{p[q][j].x + p[q][j].y, p[q][j].x - p[q][j].y};
#pragma unroll
  for (int q = 0; q < N; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (q < n) p[q][j] = __builtin_elementwise_fma(p[q][j], f2_t{s2, s2}, xlane2(p[q][j], 2));
#pragma unroll
  for (int q = 0; q < N; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (q < n) p[q][j] = __builtin_elementwise_fma(p[q][j], f2_t{s4, s4}, xlane2(p[q][j], 4));
#pragma unroll
  for (int q = 0; q < N; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (q < n) p[q][j] = __builtin_elementwise_fma(p[q][j], f2_t{s8, s8}, xlane2(p[q][j], 8));
#pragma unroll
  for (int q = 0; q < N; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t[q][j] = p[q][j].x;
      t[q][j + 4] = p[q][j].y;
    }
}

// group-of-16-lanes maxima of N values, the DPP steps interleaved
template <int N>
__device__ __forceinline__ void row_max_dpp16_n(uint32_t (&v)[N], int n) {
  auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
#pragma unroll
  for (int step = 0; step < 4; ++step)
#pragma unroll
    for (int q = 0; q < N; ++q) {
      if (q >= n) continue;
      uint32_t o;
      if (step == 0) o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v[q], 0xB1, 0xF, 0xF, true);
      else if (step == 1) o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v[q], 0x4E, 0xF, 0xF, true);
      else if (step == 2) o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v[q], 0x141, 0xF, 0xF, true);
      else o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v[q], 0x140, 0xF, 0xF, true);
      v[q] = mx(v[q], o);
    }
}

template <typename Tin, bool EMIT, int U, bool CODES = false>
__global__ __launch_bounds__(kBlock) void rotate_quant16_kernel(const void* __restrict__ xv, u32x4* __restrict__ out,
                                                               u32x4* __restrict__ rot_out, int64_t n_vec,
                                                               RotArgs r, Lut16Args a, Lut16Tab tab) {
  __shared__ __attribute__((aligned(16))) uint16_t lut[kLutLdsEntries];   // static: a compile-time LDS address (a dynamic base is not folded into the ds_read offsets)
  const int lg = threadIdx.x & 15;
  // this lane's 8 sign bits -> xor masks on packed halves
  const uint32_t sb = (r.sign[lg >> 2] >> ((lg & 3) * 8)) & 0xFFu;
  uint32_t sx[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) sx[k] = (((sb >> (2 * k)) & 1u) << 15) | (((sb >> (2 * k + 1)) & 1u) << 31);
  const int64_t tiles = (n_vec + (int64_t)kBlock * U - 1) / ((int64_t)kBlock * U);
  bool first = true;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t v0 = tile * ((int64_t)kBlock * U) + threadIdx.x;
    u32x4 raw[U];
    bool live[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int64_t v = v0 + u * kBlock;
      live[u] = v < n_vec;
      u32x4 w = {0, 0, 0, 0};
      if (live[u]) {
        if constexpr (sizeof(Tin) == 2) {
          w = __builtin_nontemporal_load((const u32x4*)xv + v);
          if (r.smooth) {   // h = half(float(x) * s)
            const float* sp = r.smooth + (v % r.vec_per_row) * 8;
#pragma unroll
            for (int k = 0; k < 4; ++k)
              w[k] = f2h(h2f(w[k] & 0xFFFFu) * sp[2 * k]) | (f2h(h2f(w[k] >> 16) * sp[2 * k + 1]) << 16);
          }
        } else {            // fp32 producer output: h = half(x * s)
          u32x4 lo = __builtin_nontemporal_load((const u32x4*)xv + 2 * v);
          u32x4 hi = __builtin_nontemporal_load((const u32x4*)xv + 2 * v + 1);
          float f[8] = {u2f(lo[0]), u2f(lo[1]), u2f(lo[2]), u2f(lo[3]), u2f(hi[0]), u2f(hi[1]), u2f(hi[2]), u2f(hi[3])};
          if (r.smooth) {
            const float* sp = r.smooth + (v % r.vec_per_row) * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] *= sp[i];
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) w[k] = f2h(f[2 * k]) | (f2h(f[2 * k + 1]) << 16);
        }
      }
      raw[u] = w;
    }
    if (first) {
      lut16_stage(lut, tab, a.shift);
      __syncthreads();
      first = false;
    }
    // the U vectors of a lane go through the stages together (stage-major: U independent dependency chains)
    u32x4 ws[U], ys[U];
    float tt[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) ws[u][k] = raw[u][k] ^ sx[k];
    fwht128_h_n<U>(ws, tt, U, lg);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) ys[u][k] = mul2_to_h2(tt[u][2 * k], tt[u][2 * k + 1], r.c_h);
    uint32_t ms[U];
#pragma unroll
    for (int u = 0; u < U; ++u) ms[u] = vec_absmax16(ys[u]);
    row_max_dpp16_n<U>(ms, U);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const u32x4 y = ys[u];
      if (EMIT && live[u]) __builtin_nontemporal_store(y, rot_out + v0 + u * kBlock);
      RowScale16 s = row_scale16(ms[u], a.fpos.gmax, a.inv_gpos);
      if constexpr (CODES) {
        const uint32_t c = codes_vec16(y, lut, a.shift, s.inv, s.inv_lo);
        if (live[u]) {
          const int64_t v = v0 + u * kBlock;
          ((uint32_t*)out)[v] = c;
          if (lg == 0) r.code_scales[v >> 4] = (uint16_t)(s.s16x2 & 0xFFFFu);
        }
      } else {
        u32x4 o = quant_vec16<false>(y, lut, a.shift, s.inv, s.inv_lo, s.s16x2, 0.f, 0.f, 0u);
        if (live[u]) __builtin_nontemporal_store(o, out + v0 + u * kBlock);
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// F1, complete form: the whole producer of tr/basic_var.py:263 / :266 in one launch
//     h  = half( ((LN(x) * half(scale+1)) + shift) * s )      LN without affine, eps, fp32 math
//     x1 = half( c_h * FWHT128(h * D) )
//     q  = per-group(128) quant(x1)
// One workgroup per token row (C <= 4096): the row stays in registers from the first load
// to the final store; mean / variance by a two-pass block reduction (shuffles + LDS).
// fp32 op order follows the reference's chain of torch ops (mul, add_, mul - each rounded
// to fp32, no contraction); LayerNorm's own mean/rstd differ from torch's Welford kernel
// by fp32 rounding only, so h can differ from torch's by 1 fp16 ulp on rare elements:
// the contract for this entry point is the fuzzy one of SURVEY.md section 7 (quant stage
// bit-exact on the rotated values produced here; rotated values within 1 ulp).
// ---------------------------------------------------------------------------------
struct AdaLnArgs {
  const void* scale;      // [batches, cols]  (scale1 / scale2 of the block)
  const void* shift;      // [batches, cols]
  int mod_is_f16;         // dtype of scale / shift
  int64_t rows_per_batch; // L: row r uses batch r / L
  float eps;
  int64_t cols;
};

__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// One WAVEFRONT per token row (64 lanes x MAXC vectors of 8 channels, C <= 64*8*MAXC): no LDS
// traffic and no workgroup barrier after the table is staged; the modulation vectors of the
// row are requested together with x so that their latency hides behind the two reductions.
template <int LANES>
__device__ __forceinline__ float row_sum_f32(float v, float* sh) {
  v = wave_sum_f32(v);
  if constexpr (LANES == 64) {
    return v;
  } else {   // the whole workgroup owns the row
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = sh[0];
#pragma unroll
    for (int i = 1; i < kBlock / 64; ++i) r += sh[i];
    return r;
  }
}

template <typename Tin, typename Tmod, int LANES, int MAXC, bool CODES = false, bool TOKEN = false>
__global__ __launch_bounds__(kBlock) void adaln_rotate_quant16_kernel(const void* __restrict__ xv,
                                                                     u32x4* __restrict__ out, u32x4* __restrict__ h_out,
                                                                     u32x4* __restrict__ y_out, int64_t rows,
                                                                     AdaLnArgs ad, RotArgs r, Lut16Args a, Lut16Tab tab) {
  __shared__ __attribute__((aligned(16))) uint16_t lut[kLutLdsEntries];   // static: a compile-time LDS address (a dynamic base is not folded into the ds_read offsets)
  __shared__ float shf[kBlock / 64];
  constexpr bool MOD16 = sizeof(Tmod) == 2;
  const int lane = threadIdx.x & (LANES - 1);
  const int lg = lane & 15;
  const uint32_t sb = (r.sign[lg >> 2] >> ((lg & 3) * 8)) & 0xFFu;
  uint32_t sx[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) sx[k] = (((sb >> (2 * k)) & 1u) << 15) | (((sb >> (2 * k + 1)) & 1u) << 31);
  {
    lut16_stage(lut, tab, a.shift);
    __syncthreads();
  }
  const int64_t vpr = r.vec_per_row;
  const float inv_c = 1.0f / (float)ad.cols;
  // a workgroup walks rows blockIdx.x*R + sub, (blockIdx.x + gridDim.x)*R + sub, ... (R rows per pass);
  // the trip count is uniform over the workgroup, rows beyond the end are skipped by `row_live`
  constexpr int R = kBlock / LANES;
  for (int64_t base = (int64_t)blockIdx.x * R; base < rows; base += (int64_t)gridDim.x * R) {
  const int64_t row_raw = base + (threadIdx.x / LANES);
  const bool row_live = row_raw < rows;
  const int64_t row = row_live ? row_raw : rows - 1;   // dead wavefronts recompute the last row, stores masked
  const int64_t b = row / ad.rows_per_batch;

  float f[MAXC][8];
  constexpr int MV = MOD16 ? 1 : 2;
  u32x4 m_sc[MAXC][MV], m_sh[MAXC][MV];   // raw modulation vectors, requested together with x
  float s1 = 0.0f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int64_t v = (int64_t)c * LANES + lane;
    const bool live = v < vpr;
    const int64_t col = v * 8;
    if constexpr (sizeof(Tin) == 2) {
      u32x4 w = live ? __builtin_nontemporal_load((const u32x4*)xv + row * vpr + v) : u32x4{0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f[c][2 * k] = h2f(w[k] & 0xFFFFu);
        f[c][2 * k + 1] = h2f(w[k] >> 16);
      }
    } else {
      u32x4 lo = live ? __builtin_nontemporal_load((const u32x4*)xv + 2 * (row * vpr + v)) : u32x4{0, 0, 0, 0};
      u32x4 hi = live ? __builtin_nontemporal_load((const u32x4*)xv + 2 * (row * vpr + v) + 1) : u32x4{0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f[c][k] = u2f(lo[k]);
        f[c][4 + k] = u2f(hi[k]);
      }
    }
#pragma unroll
    for (int j = 0; j < MV; ++j) m_sc[c][j] = m_sh[c][j] = u32x4{0, 0, 0, 0};
    if (live) {
      const u32x4* ap = (const u32x4*)((const Tmod*)ad.scale + b * ad.cols + col);
      const u32x4* bp = (const u32x4*)((const Tmod*)ad.shift + b * ad.cols + col);
#pragma unroll
      for (int j = 0; j < MV; ++j) {
        m_sc[c][j] = ap[j];
        m_sh[c][j] = bp[j];
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s1 += f[c][i];
  }
  const float mean = row_sum_f32<LANES>(s1, shf) * inv_c;
  float s2 = 0.0f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const bool live = (int64_t)c * LANES + lane < vpr;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float d = f[c][i] - mean;
      f[c][i] = d;
      s2 = __builtin_fmaf(d, d, s2);
    }
    if (!live) s2 -= 8.0f * mean * mean;   // padding lanes hold zeros: take their (0 - mean)^2 back out
  }
  const float var = row_sum_f32<LANES>(s2, shf) * inv_c;
  const float rstd = 1.0f / __builtin_sqrtf(var + ad.eps);
  u32x4 ys[TOKEN ? MAXC : 1];
  uint32_t mrow = 0;
  (void)ys;
  (void)mrow;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int64_t v = (int64_t)c * LANES + lane;
    const bool live = v < vpr;
    float t[8];
    if (live) {
      const int64_t col = v * 8;
      float sc[8], sh[8], sm[8];
      if constexpr (MOD16) {   // scale.add(1) is an fp16 op in the reference
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t s1p = pk_add_f16(m_sc[c][0][k], 0x3C003C00u);   // scale + 1 as a (packed) fp16 add
          sc[2 * k] = h2f(s1p & 0xFFFFu);
          sc[2 * k + 1] = h2f(s1p >> 16);
          sh[2 * k] = h2f(m_sh[c][0][k] & 0xFFFFu);
          sh[2 * k + 1] = h2f(m_sh[c][0][k] >> 16);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          sc[k] = u2f(m_sc[c][0][k]) + 1.0f;
          sc[4 + k] = u2f(m_sc[c][MV - 1][k]) + 1.0f;
          sh[k] = u2f(m_sh[c][0][k]);
          sh[4 + k] = u2f(m_sh[c][MV - 1][k]);
        }
      }
      if (r.smooth) {
        const u32x4* sp = (const u32x4*)(r.smooth + col);
        u32x4 s0 = sp[0], s1v = sp[1];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          sm[k] = u2f(s0[k]);
          sm[4 + k] = u2f(s1v[k]);
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float ln = f[c][i] * rstd;
        float u1 = ln * sc[i];
        float u2 = u1 + sh[i];
        if (r.smooth) u2 = u2 * sm[i];
        t[i] = u2;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = 0.0f;
    }
    u32x4 hw;
#pragma unroll
    for (int k = 0; k < 4; ++k) hw[k] = f2h2(t[2 * k], t[2 * k + 1]);
    if (h_out && live && row_live) __builtin_nontemporal_store(hw, h_out + row * vpr + v);
#pragma unroll
    for (int k = 0; k < 4; ++k) t[2 * k] = t[2 * k + 1] = 0.0f;
    {
      u32x4 ws[1];
      float tt[1][8];
#pragma unroll
      for (int k = 0; k < 4; ++k) ws[0][k] = hw[k] ^ sx[k];
      fwht128_h_n<1>(ws, tt, 1, lg);
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] = tt[0][k];
    }
    u32x4 y;
#pragma unroll
    for (int k = 0; k < 4; ++k) y[k] = mul2_to_h2(t[2 * k], t[2 * k + 1], r.c_h);
    if (y_out && live && row_live) __builtin_nontemporal_store(y, y_out + row * vpr + v);
    if constexpr (TOKEN) {   // per-token scale: keep the rotated row, quantize after the row maximum is known
      ys[c] = y;
      const uint32_t mv = vec_absmax16(y);
      mrow = mrow > mv ? mrow : mv;
      continue;
    }
    uint32_t m = row_max_dpp<16>(vec_absmax16(y));
    RowScale16 s = row_scale16(m, a.fpos.gmax, a.inv_gpos);
    if constexpr (CODES) {
      const uint32_t cd = codes_vec16(y, lut, a.shift, s.inv, s.inv_lo);
      if (live && row_live) {
        ((uint32_t*)out)[row * vpr + v] = cd;
        if (lg == 0) r.code_scales[(row * vpr + v) >> 4] = (uint16_t)(s.s16x2 & 0xFFFFu);
      }
    } else {
      u32x4 o = quant_vec16<false>(y, lut, a.shift, s.inv, s.inv_lo, s.s16x2, 0.f, 0.f, 0u);
      if (live && row_live) __builtin_nontemporal_store(o, out + row * vpr + v);
    }
  }
  if constexpr (TOKEN) {
    // fp6_quant_*_per_token_cuda on the rotated row (tr/quant_utils.py:503-534): one scale for the whole row
    static_assert(!TOKEN || LANES == 64, "the per-token form keeps a row inside one wavefront");
    mrow = row_max_dpp<64>(mrow);
    const RowScale16 s = row_scale16(mrow, a.fpos.gmax, a.inv_gpos);
    if (r.code_scales && lane == 0 && row_live) r.code_scales[row] = (uint16_t)(s.s16x2 & 0xFFFFu);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int64_t v = (int64_t)c * LANES + lane;
      if (v < vpr && row_live) {
        if constexpr (CODES) {   // E4M3 bytes of the levels (fpq_gemm_fp8.h), 8 per vector
          const uint32_t wk0 = ys[c][0], wk1 = ys[c][1], wk2 = ys[c][2], wk3 = ys[c][3];
          uint32_t cb[8];
          const uint32_t ws[4] = {wk0, wk1, wk2, wk3};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const uint32_t wk = ws[k];
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
              uint8_t* dst = (uint8_t*)out + row * (vpr * 6) + (int64_t)c * (LANES * 6) + 24 * (lane >> 2) + 8 * qp;
              __builtin_nontemporal_store(u32x2{(uint32_t)w, (uint32_t)(w >> 32)}, (u32x2*)dst);
            }
          } else {
            const u32x2 o2 = {cb[0] | (cb[1] << 8) | (cb[2] << 16) | (cb[3] << 24), cb[4] | (cb[5] << 8) | (cb[6] << 16) | (cb[7] << 24)};
            __builtin_nontemporal_store(o2, (u32x2*)out + row * vpr + v);
          }
        } else {
          u32x4 o = quant_vec16<false>(ys[c], lut, a.shift, s.inv, s.inv_lo, s.s16x2, 0.f, 0.f, 0u);
          __builtin_nontemporal_store(o, out + row * vpr + v);
        }
      }
    }
  }
  }   // row loop
}
