// fpq_fast32.h - fp32 input, rows of 128 elements: the weight-calibration hot path
// (QuantizedLinear.from_float -> fp_quant_e{1,2,3}_per_group_cuda / fp6_quant_*_per_group_cuda on the fp32 weight,
// tr/quant_utils.py:828-855; the driver's later var.half() fused as the fp16 output form).
// Included by fpq_kernels.hip inside its anonymous namespace, after the generic helpers.
//
// Same results as rows_subwave_kernel<float, Tout, 32, false, U> (IEEE fp32 division + closed form, ~50 VALU ops per
// element, 8-byte stores for fp16 output), at ~17 VALU ops per element and 16-byte accesses only:
//
//   * 32 lanes own a group, each lane 4 consecutive floats (one fully coalesced 16-byte load per iteration); the
//     group maximum is 4 DPP steps + one v_permlane16_swap; fp16 results leave as 16-byte stores after a pair
//     exchange between neighbouring lanes (see the kernel).
//   * xn = RN32(x / s) is APPROXIMATED by  y0 = x*r, e = fma(-y0, s, x), y = fma(e, r, y0)  with r = one Newton step
//     on v_rcp_f32(s) (computed once per lane and group): |y - RN32(x/s)| <= 1 ulp for a normal, finite s.
//     The level only depends on which side of the rounding boundaries xn falls, every boundary has at most M+1
//     fractional mantissa bits, so y decides exactly like RN32(x/s) unless it lies within 3 ulp of a boundary.
//     That is checked per element (3 integer ops); should ANY lane of the wavefront see such a value - or a scale
//     outside [2^-90, 2^90], zero, infinite or NaN - the whole wavefront redoes its vectors with the generic IEEE
//     path (quant_sym<float>).  Probability ~1e-6 per element: a few thousand wavefronts per 1.3 G weights.
//   * the level comes from a table in LDS indexed by the fp32 bit pattern's top bits (one bucket = 2^(22-M)
//     patterns; boundaries are bucket edges; "tie to the larger value" = pattern - 1 for negative xn), the pattern
//     clamped to [one bucket below the smallest boundary, the bucket of the largest boundary] first, so the table has
//     <= 128 entries per sign and out-of-range values can never look "near a boundary".  Each workgroup fills it
//     from the closed form (quant_mag), one entry per thread.
//   * out = Tout(fp32(level * s)): the fp32 product exists before the fp16 rounding, as torch's mul + .half() do.
#pragma once

struct Lut32Args {
  Fmt f;
  uint32_t lo_clamp;   // (bucket below the smallest boundary) << bshift | half a bucket
  uint32_t hi_clamp;   // (bucket of the largest boundary) << bshift | half a bucket
  int bshift;          // 22 - M
  int nbits;           // the table holds 2^nbits buckets per sign, indexed by bucket mod 2^nbits
};

// One segment of a multi-tensor launch: `rows` groups of 128 elements, contiguous (device-resident table,
// include/fpq.h fpq_segment_t has the same layout)
struct Seg32 {
  const void* x;
  void* out;
  int64_t rows;
};

inline Lut32Args lut32_args(int table_id) {
  Lut32Args a;
  a.f = make_fmt(table_id);
  const int M = kTables[table_id].mbits;
  a.bshift = 22 - M;
  float pos[64];
  const int np = pos_levels(table_id, pos);
  const float b_lo = 0.5f * (pos[0] + pos[1]), b_hi = 0.5f * (pos[np - 2] + pos[np - 1]);
  const uint32_t lo_bucket = (f2u(b_lo) >> a.bshift) - 1u, hi_bucket = f2u(b_hi) >> a.bshift;
  const uint32_t half_bucket = 1u << (a.bshift - 1);
  a.lo_clamp = (lo_bucket << a.bshift) | half_bucket;
  a.hi_clamp = (hi_bucket << a.bshift) | half_bucket;
  int nb = 1;
  while ((1u << nb) < hi_bucket - lo_bucket + 1u) ++nb;
  a.nbits = nb;
  return a;
}

constexpr int kLut32MaxBits = 7;   // every symmetric table fits (E2M3: 113 buckets)

__device__ __forceinline__ void lut32_fill(float* lut, const Lut32Args& a) {
  // entry [sign][bucket mod 2^nbits]: the signed level of the bucket's lowest pattern (the table's zero is +0)
  const uint32_t lo_bucket = a.lo_clamp >> a.bshift, hi_bucket = a.hi_clamp >> a.bshift;
  const uint32_t n = hi_bucket - lo_bucket + 1u, wrap = (1u << a.nbits) - 1u;
  for (uint32_t i = threadIdx.x; i < 2u * n; i += blockDim.x) {
    const uint32_t neg = i >= n ? 1u : 0u, b = lo_bucket + (i - neg * n);
    float q = quant_mag(u2f(b << a.bshift), 0u, a.f);
    if (b == lo_bucket) q = 0.0f;   // everything below the smallest boundary
    lut[(neg << a.nbits) | (b & wrap)] = (neg && q != 0.0f) ? -q : q;
  }
}

// Four elements of one lane through the fast path: levels * s in p[].  `slow` (this lane already knows its scale is
// outside the fast path's range) is OR-ed with "one of my values is too close to a rounding boundary"; if ANY lane of
// the wavefront says so, the wavefront redoes these four with the IEEE path.
__device__ __forceinline__ void quant4_fast32(const u32x4& raw, float s, float r, bool slow, const Lut32Args& a,
                                              const float* lut, uint32_t low_mask, uint32_t idx_mask, float (&p)[4]) {
  uint32_t near = 0xFFFFFFFFu;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float xf = u2f(raw[i]);
    const float y0 = xf * r;
    const float e = __builtin_fmaf(-y0, s, xf);
    const float y = __builtin_fmaf(e, r, y0);
    const uint32_t yb = fbits(y);
    const uint32_t uu = (yb & 0x7FFFFFFFu) - (yb >> 31);          // negative: magnitude pattern - 1
    uint32_t c;                                                    // signed clamp: -0.0 gives pattern 0 - 1 = -1 -> lo
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(c) : "v"(uu), "s"(a.lo_clamp), "v"(a.hi_clamp));   // one SGPR per VOP3
    const uint32_t d = (c + 3u) & low_mask;
    near = near < d ? near : d;
    const uint32_t off = ((c >> (a.bshift - 2)) & idx_mask) | ((yb >> 31) << (a.nbits + 2));
    const float q = *(const float*)((const char*)lut + off);
    p[i] = q * s;
  }
  slow |= near <= 6u;
  if (__builtin_amdgcn_ballot_w64(slow) != 0) {   // rare
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = quant_sym<float>(u2f(raw[i]), s, a.f);
  }
}

// scale and its refined reciprocal from a row / group maximum; *slow: the fast path's range check
__device__ __forceinline__ void scale_fast32(uint32_t amax_bits, float gmax, float* s, float* r, bool* slow) {
  const float sv = u2f(amax_bits) / gmax;                       // IEEE: scale = absmax / max|table|
  float rv = __builtin_amdgcn_rcpf(sv);
  rv = __builtin_fmaf(__builtin_fmaf(-sv, rv, 1.0f), rv, rv);   // one Newton step
  *s = sv;
  *r = rv;
  *slow = (fbits(sv) - 0x12800000u) > (0x6C000000u - 0x12800000u);   // s outside [2^-90, 2^90], zero, inf or NaN
}

// max over the 32 lanes that own a group (lanes 32k .. 32k+31): 4 DPP steps inside each row of 16, then the two rows
// of a half-wave trade their maxima with ONE v_permlane16_swap (gfx950; no LDS crossbar, no address VGPR)
__device__ __forceinline__ uint32_t group32_max(uint32_t v) {
  v = row_max_dpp<16>(v);
  const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);   // r[0]: rows {0,0,2,2}, r[1]: rows {1,1,3,3}
  return r[0] > r[1] ? r[0] : r[1];
}

// 32 lanes own a group: every load instruction is ONE fully coalesced 16-byte-per-lane access (lane l reads floats
// 4l .. 4l+3 of its group; a 16-lanes-x-8-floats mapping leaves each instruction half of every 128-byte line and
// measured 0.50 of 8 TB/s where this one reaches the plain-copy rate).  fp16 results are 8 bytes per lane and load:
// iterations are processed in pairs (u, u+1) and the two lanes of a pair (2k, 2k+1) swap halves through one DPP
// quad_perm, so that the even lane stores the pair's 16 bytes of iteration u and the odd lane those of iteration
// u+1 - every store instruction is 16 bytes per lane again, two contiguous 512-byte runs.
template <typename Tout, int U>
__global__ __launch_bounds__(kBlock) void groups32_lut_kernel(const Seg32* __restrict__ segs, Seg32 one, Lut32Args a) {
  static_assert(U % 2 == 0, "iterations are stored in pairs");
  __shared__ float lut[2 << kLut32MaxBits];
  const Seg32 sg = segs ? segs[blockIdx.y] : one;   // wave-uniform: scalar loads
  const int64_t n_vec = sg.rows * 32;                // 16-byte input vectors (4 floats)
  if ((int64_t)blockIdx.x * (kBlock * U) >= n_vec) return;   // a shorter segment of a multi-tensor launch
  const int64_t v0 = (int64_t)blockIdx.x * (kBlock * U) + threadIdx.x;
  const u32x4* __restrict__ x = (const u32x4*)sg.x;
  u32x4 raw[U];
  bool live[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t v = v0 + u * kBlock;
    live[u] = v < n_vec;   // rows are whole: the 32 lanes of a group are live or dead together
    raw[u] = live[u] ? __builtin_nontemporal_load(x + v) : u32x4{0, 0, 0, 0};
  }
  lut32_fill(lut, a);
  __syncthreads();
  const uint32_t low_mask = (1u << a.bshift) - 1u;
  const uint32_t idx_mask = ((1u << a.nbits) - 1u) << 2;
  const bool odd = (threadIdx.x & 1) != 0;
  uint32_t hprev[2] = {0, 0};   // packed fp16 results of the pair's first iteration
#pragma unroll
  for (int u = 0; u < U; ++u) {
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t ab = raw[u][i] & 0x7FFFFFFFu;
      m = m > ab ? m : ab;
    }
    m = group32_max(m);
    float s, r;
    bool slow;
    scale_fast32(m, a.f.gmax, &s, &r, &slow);
    slow = slow && live[u];
    float p[4];
    quant4_fast32(raw[u], s, r, slow, a, lut, low_mask, idx_mask, p);   // dead lanes hold zeros: never "near"
    const int64_t v = v0 + u * kBlock;
    if constexpr (sizeof(Tout) == 4) {
      if (live[u]) __builtin_nontemporal_store(u32x4{fbits(p[0]), fbits(p[1]), fbits(p[2]), fbits(p[3])}, (u32x4*)sg.out + v);
    } else {
      const uint32_t h0 = f2h2(p[0], p[1]), h1 = f2h2(p[2], p[3]);
      if ((u & 1) == 0) {
        hprev[0] = h0;
        hprev[1] = h1;
      } else {
        // even lane keeps iteration u-1 and needs its right neighbour's u-1 halves; odd lane keeps iteration u and
        // needs its left neighbour's u halves: each lane sends what its partner stores
        const uint32_t s0 = odd ? hprev[0] : h0, s1 = odd ? hprev[1] : h1;
        const uint32_t r0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s0, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
        const uint32_t r1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s1, 0xB1, 0xF, 0xF, true);
        const u32x4 o = odd ? u32x4{r0, r1, h0, h1} : u32x4{hprev[0], hprev[1], r0, r1};
        const int64_t vs = odd ? v : v - kBlock;              // the iteration this lane stores
        if (odd ? live[u] : live[u - 1]) __builtin_nontemporal_store(o, (u32x4*)sg.out + (vs >> 1));
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// fp32 groups of 128 -> CODES + one fp32 scale per group (round 4): the quantize side of the calibration's packed
// exchange (fpq_quant_rows_codes[_segments] on fp32 weights; calibrate.ShardedCodesCalibration).  Until now that side ran
// codes128_kernel<float>: IEEE fp32 division + the closed form per element, ~50 vector instructions per element, 0.25 -
// 0.30 of 8 TB/s - twice the time of the fp16-slab form it is meant to beat.  Same mapping and the same
// approximate-then-verify quotient as groups32_lut_kernel above; the LDS table holds the CODE of each bucket (index in
// the sorted de-duplicated table: zero_code +- level index) instead of its level, the slow path is the generic code
// emitter's arithmetic.  Output: PACK - 4 nibbles = 2 bytes per lane and iteration (64 lanes x 2 B = one 128-byte line
// per store instruction), else 4 bytes per lane; lane 0 of a group writes the scale.
// ---------------------------------------------------------------------------------------------------------------
struct CodesSeg32 {   // = CodesSeg of fpq_kernels.hip = fpq_codes_segment_t (include/fpq.h)
  const void* x;
  uint8_t* codes;
  void* scales;
  int64_t rows;
};

__device__ __forceinline__ void lut32_fill_codes(uint32_t* lut, const Lut32Args& a) {
  const uint32_t lo_bucket = a.lo_clamp >> a.bshift, hi_bucket = a.hi_clamp >> a.bshift;
  const uint32_t n = hi_bucket - lo_bucket + 1u, wrap = (1u << a.nbits) - 1u;
  for (uint32_t i = threadIdx.x; i < 2u * n; i += blockDim.x) {
    const uint32_t neg = i >= n ? 1u : 0u, b = lo_bucket + (i - neg * n);
    const float q = b == lo_bucket ? 0.0f : quant_mag(u2f(b << a.bshift), 0u, a.f);   // everything below the smallest boundary: zero
    const int li = level_index(q, a.f);
    lut[(neg << a.nbits) | (b & wrap)] = (uint32_t)(neg ? a.f.zero_code - li : a.f.zero_code + li);
  }
}

__device__ __forceinline__ void codes4_fast32(const u32x4& raw, float s, float r, bool slow, const Lut32Args& a,
                                              const uint32_t* lut, uint32_t low_mask, uint32_t idx_mask, uint32_t (&c4)[4]) {
  uint32_t near = 0xFFFFFFFFu;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float xf = u2f(raw[i]);
    const float y0 = xf * r;
    const float e = __builtin_fmaf(-y0, s, xf);
    const float y = __builtin_fmaf(e, r, y0);
    const uint32_t yb = fbits(y);
    const uint32_t uu = (yb & 0x7FFFFFFFu) - (yb >> 31);          // negative: magnitude pattern - 1
    uint32_t c;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(c) : "v"(uu), "s"(a.lo_clamp), "v"(a.hi_clamp));
    const uint32_t d = (c + 3u) & low_mask;
    near = near < d ? near : d;
    const uint32_t off = ((c >> (a.bshift - 2)) & idx_mask) | ((yb >> 31) << (a.nbits + 2));
    c4[i] = *(const uint32_t*)((const char*)lut + off);
  }
  slow |= near <= 6u;
  if (__builtin_amdgcn_ballot_w64(slow) != 0) {   // rare: the generic emitter's arithmetic (codes128_body)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float xn = u2f(raw[i]) / s;
      const uint32_t neg = (xn < 0.0f) ? 1u : 0u;
      const int li = level_index(quant_mag(__builtin_fabsf(xn), neg, a.f), a.f);
      c4[i] = (uint32_t)(neg ? a.f.zero_code - li : a.f.zero_code + li);
    }
  }
}

template <bool PACK, int U>
__global__ __launch_bounds__(kBlock) void groups32_codes_kernel(const CodesSeg32* __restrict__ segs, CodesSeg32 one, Lut32Args a) {
  __shared__ uint32_t lut[2 << kLut32MaxBits];
  const CodesSeg32 sg = segs ? segs[blockIdx.y] : one;   // wave-uniform: scalar loads
  const int64_t n_vec = sg.rows * 32;                     // 16-byte input vectors (4 floats)
  if ((int64_t)blockIdx.x * (kBlock * U) >= n_vec) return;
  const int64_t v0 = (int64_t)blockIdx.x * (kBlock * U) + threadIdx.x;
  const u32x4* __restrict__ x = (const u32x4*)sg.x;
  u32x4 raw[U];
  bool live[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t v = v0 + u * kBlock;
    live[u] = v < n_vec;
    raw[u] = live[u] ? __builtin_nontemporal_load(x + v) : u32x4{0, 0, 0, 0};
  }
  lut32_fill_codes(lut, a);
  __syncthreads();
  const uint32_t low_mask = (1u << a.bshift) - 1u;
  const uint32_t idx_mask = ((1u << a.nbits) - 1u) << 2;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t ab = raw[u][i] & 0x7FFFFFFFu;
      m = m > ab ? m : ab;
    }
    m = group32_max(m);
    float s, r;
    bool slow;
    scale_fast32(m, a.f.gmax, &s, &r, &slow);
    slow = slow && live[u];
    uint32_t c4[4];
    codes4_fast32(raw[u], s, r, slow, a, lut, low_mask, idx_mask, c4);
    const int64_t v = v0 + u * kBlock;
    if (live[u]) {
      if ((threadIdx.x & 31) == 0) ((float*)sg.scales)[v >> 5] = s;
      if constexpr (PACK) ((uint16_t*)sg.codes)[v] = (uint16_t)(c4[0] | (c4[1] << 4) | (c4[2] << 8) | (c4[3] << 12));
      else ((uint32_t*)sg.codes)[v] = c4[0] | (c4[1] << 8) | (c4[2] << 16) | (c4[3] << 24);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Long fp32 rows, one scale per row (per-channel weights of the W6A6 runs: fp6_quant_e2m3_per_token_cuda on the fp32
// [out, in] weight, tr/quant_utils.py:808-811, result fp16).  LANES = 64: one wavefront per row (in <= 2048),
// LANES = 256: one workgroup per row.  A lane holds vectors c * LANES + lane of its row (4 floats each, every load a
// fully coalesced 16 bytes per lane), MAXC of them; same fast path as above with the row's scale; fp16 results leave
// as 16-byte stores after the neighbour exchange (vector pairs c, c + 1), a last unpaired vector as 8-byte stores.
// ---------------------------------------------------------------------------------------------------------------
template <typename Tout, int LANES, int MAXC>
__global__ __launch_bounds__(kBlock) void rows32_lut_kernel(const float* __restrict__ x, Tout* __restrict__ out,
                                                           int64_t rows, int64_t cols, Lut32Args a) {
  __shared__ float lut[2 << kLut32MaxBits];
  __shared__ uint32_t sh[kBlock / 64];
  lut32_fill(lut, a);
  __syncthreads();
  const uint32_t low_mask = (1u << a.bshift) - 1u;
  const uint32_t idx_mask = ((1u << a.nbits) - 1u) << 2;
  const int lane = threadIdx.x & (LANES - 1);
  const int vpr = (int)(cols >> 2);              // host: cols % 8 == 0, vpr <= LANES * MAXC
  const bool odd = (lane & 1) != 0;
  constexpr int R = kBlock / LANES;              // rows per workgroup pass
  for (int64_t base = (int64_t)blockIdx.x * R; base < rows; base += (int64_t)gridDim.x * R) {
    const int64_t row = base + threadIdx.x / LANES;
    const bool row_live = row < rows;            // uniform per wavefront (LANES >= 64)
    const u32x4* xr = (const u32x4*)(x + (row_live ? row : rows - 1) * cols);
    u32x4 raw[MAXC];
    uint32_t m = 0;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int v = c * LANES + lane;
      raw[c] = v < vpr ? __builtin_nontemporal_load(xr + v) : u32x4{0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t ab = raw[c][i] & 0x7FFFFFFFu;
        m = m > ab ? m : ab;
      }
    }
    m = row_max_dpp<16>(m);
    {
      auto r16 = __builtin_amdgcn_permlane16_swap(m, m, false, false);
      m = r16[0] > r16[1] ? r16[0] : r16[1];
      auto r32 = __builtin_amdgcn_permlane32_swap(m, m, false, false);
      m = r32[0] > r32[1] ? r32[0] : r32[1];
    }
    if constexpr (LANES == 256) {                 // rows never share a workgroup here: plain block maximum
      __syncthreads();
      if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
      __syncthreads();
      m = sh[0];
#pragma unroll
      for (int i = 1; i < kBlock / 64; ++i) m = m > sh[i] ? m : sh[i];
    }
    float s, r;
    bool slow;
    scale_fast32(m, a.f.gmax, &s, &r, &slow);
    uint32_t hprev[2] = {0, 0};
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int v = c * LANES + lane;
      const bool live = v < vpr && row_live;
      float p[4];
      quant4_fast32(raw[c], s, r, slow && live, a, lut, low_mask, idx_mask, p);
      if constexpr (sizeof(Tout) == 4) {
        if (live) __builtin_nontemporal_store(u32x4{fbits(p[0]), fbits(p[1]), fbits(p[2]), fbits(p[3])},
                                              (u32x4*)(out + row * cols) + v);
      } else {
        const uint32_t h0 = f2h2(p[0], p[1]), h1 = f2h2(p[2], p[3]);
        if ((c & 1) == 0 && c + 1 < MAXC) {
          hprev[0] = h0;
          hprev[1] = h1;
        } else if ((c & 1) == 1) {
          // as in groups32_lut_kernel: the even lane stores the pair's 16 bytes of vector c - 1, the odd lane those of c
          const uint32_t s0 = odd ? hprev[0] : h0, s1 = odd ? hprev[1] : h1;
          const uint32_t r0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s0, 0xB1, 0xF, 0xF, true);
          const uint32_t r1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s1, 0xB1, 0xF, 0xF, true);
          const u32x4 o = odd ? u32x4{r0, r1, h0, h1} : u32x4{hprev[0], hprev[1], r0, r1};
          const int vs = odd ? v : v - LANES;                  // vpr is even: the two lanes of a pair are live together
          if (vs < vpr && row_live) __builtin_nontemporal_store(o, (u32x4*)(out + row * cols) + (vs >> 1));
        } else {                                               // MAXC odd: the last vector leaves as 8 bytes per lane
          if (live) __builtin_nontemporal_store(u32x2{h0, h1}, (u32x2*)(out + row * cols) + v);
        }
      }
    }
  }
}
