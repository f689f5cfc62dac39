// fpq_gemm_fp4.h - F2: a REAL low-precision consumer for the quantized
// activations and weights.  Included by fpq_gemm.hip inside its anonymous namespace (the operand-emitting quantizer: fpq_codes_mx.h).
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

// Byte offset of (row, 16-byte chunk kb) in the LDS image of an operand tile (64 B per row).
// ds_read_b128 serves a wavefront in four fixed 16-lane groups - {0-3,12-15,20-27}, {4-11,16-19,28-31} and
// the same +32 (MI355X_MICROARCH.md, LDS) - and with lane = (kb << 4) | row each group mixes two k-blocks.
// Layout: per 16 rows one 1 KiB block = 4 planes (one per kb) of 16 slots; row r sits in slot
// (r&3)*4 + (r>>2) of its plane, planes 2 and 3 rotated by two slots.  Every read group then covers 16
// distinct 16-byte slots (conflict-free); the staging writes (4 lanes per row) are 2-way.
__device__ __forceinline__ int gemm_lds_off(int row, int kb) {
  const int r = row & 15;
  return ((row >> 4) << 10) + (kb << 8) + (((((r & 3) << 2) + (r >> 2) + ((kb >> 1) << 1)) & 15) << 4);
}

// Workgroup = WR x WC wavefronts, each owning MT x NT MFMA tiles (16x16): tile BM = 16*MT*WR tokens by
// BN = 16*NT*WC outputs.
// Optional tail of every GEMM epilogue below: out = residual + y * gate[t / rows_per_gate, :] with y the fp16 Linear
// output, each operation in fp16 with one rounding - what torch computes for the AdaLN blocks'
// `x = x + attn(...).mul_(gamma1)` / `x + ffn(...).mul(gamma2)` (tr/basic_var.py:264,267; gamma is [B, 1, C]).
struct GemmEpi {
  const _Float16* gate;    // [ceil(T / rows_per_gate), O] or nullptr
  const _Float16* resid;   // [T, O] or nullptr; may alias out
  int rows_per_gate;
  // K-MAJOR OPERAND IMAGES (include/fpq.h, "k-major operand images"; the FP4 and FP6 LDS-DMA kernels): 0 = row-major codes,
  // else the image rows of the weight side (outs rounded up to 64; the activation side has exactly T).
  int km_w_rows;
  // SPLIT OUTPUT (include/fpq.h, fpq_gemm_split_t; the FP4 LDS-DMA kernel's plain epilogue): sp_cols != 0: the outputs are sp_cols-wide
  // column parts with a destination each - token t = b * sp_rpb + l of part p goes to row b * sp_bstride[p] + sp_row0[p] + l of
  // sp_out[p] (sp_stride[p] elements per row).  mat_qkv writes q to its own tensor and k, v straight into the KV cache's slots.
  int sp_cols, sp_rpb;
  _Float16* sp_out[3];
  int64_t sp_stride[3], sp_bstride[3], sp_row0[3];
};
typedef _Float16 fpq_h2_t __attribute__((ext_vector_type(2)));
typedef _Float16 fpq_h4_t __attribute__((ext_vector_type(4)));

// The four 8-byte stores of a lane (rows t_first .. +3, outputs o .. o+3), or with -DFPQ_GEMM_WIDE_STORES two 16-byte ones: lanes
// q and q ^ 1 hold neighbouring outputs of the same four rows; the even lane trades its rows 2, 3 for the odd lane's rows 0, 1
// (one DPP quad_perm [1,0,3,2] per dword) and each then owns eight consecutive outputs of two rows.  outs % 8 == 0 and 4q % 8 == 0
// on the even lane: both 16-byte pieces are inside the row or both outside; `out` is 16-byte aligned (checked on the host).
#ifdef FPQ_GEMM_WIDE_STORES
#define FPQ_GEMM_ROWS_STORE(y_, t_first_, tc_, o_, oc_)                                                             \
  do {                                                                                                              \
    const bool odd_ = (lane & 1) != 0;                                                                              \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                                              \
      const u32x2 own_lo_ = __builtin_bit_cast(u32x2, (y_)[j_]), own_hi_ = __builtin_bit_cast(u32x2, (y_)[2 + j_]); \
      const u32x2 send_ = odd_ ? own_lo_ : own_hi_;                                                                 \
      u32x2 recv_;                                                                                                  \
      recv_[0] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)send_[0], 0xB1, 0xF, 0xF, false);                    \
      recv_[1] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)send_[1], 0xB1, 0xF, 0xF, false);                    \
      const u32x2 first_ = odd_ ? recv_ : own_lo_, second_ = odd_ ? own_hi_ : recv_;                                \
      const u32x4 w_ = u32x4{first_[0], first_[1], second_[0], second_[1]};                                         \
      const int r_ = (odd_ ? 2 : 0) + j_;                                                                           \
      const int trow_ = odd_ ? (tc_)[2 + j_] : (tc_)[j_];                                                           \
      if ((t_first_) + r_ < T && (o_) < O)                                                                          \
        __builtin_nontemporal_store(w_, (u32x4*)(out + (int64_t)trow_ * O + (oc_) - (odd_ ? 4 : 0)));               \
    }                                                                                                               \
  } while (0)
#else
#define FPQ_GEMM_ROWS_STORE(y_, t_first_, tc_, o_, oc_)                                                             \
  do {                                                                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                                \
        if ((t_first_) + i_ < T && (o_) < O)                                                                        \
          __builtin_nontemporal_store(__builtin_bit_cast(u32x2, (y_)[i_]), (u32x2*)(out + (int64_t)(tc_)[i_] * O + (oc_))); \
  } while (0)
#endif

// (macros, not functions: the kernels carry different target attributes and a callee is only inlined into a kernel
// with the same ones)
#define FPQ_GEMM_EPI_VEC(y, e, t, o, O)                                                                  \
  do {                                                                                                   \
    if ((e).gate) {                                                                                      \
      const u32x4 g_ = *(const u32x4*)((e).gate + (int64_t)((t) / (e).rows_per_gate) * (O) + (o));       \
      _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                   \
          (y)[i_] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(fpq_h2_t, (uint32_t)(y)[i_]) *      \
                                                     __builtin_bit_cast(fpq_h2_t, (uint32_t)g_[i_]));    \
    }                                                                                                    \
    if ((e).resid) {                                                                                     \
      const u32x4 r_ = *(const u32x4*)((e).resid + (int64_t)(t) * (O) + (o));                            \
      _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                   \
          (y)[i_] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(fpq_h2_t, (uint32_t)r_[i_]) +       \
                                                     __builtin_bit_cast(fpq_h2_t, (uint32_t)(y)[i_]));   \
    }                                                                                                    \
  } while (0)
#define FPQ_GEMM_EPI_ONE(y, e, t, o, O)                                                                  \
  do {                                                                                                   \
    if ((e).gate) (y) = (y) * (e).gate[(int64_t)((t) / (e).rows_per_gate) * (O) + (o)];                  \
    if ((e).resid) (y) = (e).resid[(int64_t)(t) * (O) + (o)] + (y);                                      \
  } while (0)

template <typename Tsw, int MT, int NT, int WR, int WC>
__global__ __launch_bounds__(64 * WR * WC) void gemm_fp4_kernel(const uint8_t* __restrict__ A,
                                                               const _Float16* __restrict__ sa,
                                                               const uint8_t* __restrict__ W, const Tsw* __restrict__ sw,
                                                               const _Float16* __restrict__ bias,
                                                               _Float16* out, int T, int O, int C, GemmEpi epi) {
  constexpr int BM = 16 * MT * WR, BN = 16 * NT * WC, NTHR = 64 * WR * WC;
  constexpr int ABYTES = BM * 64, BBYTES = BN * 64, STAGE = ABYTES + BBYTES;
  constexpr int HA = (BM * 4 + NTHR - 1) / NTHR, HB = (BN * 4 + NTHR - 1) / NTHR;   // staging chunks per thread
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int G = C >> 7, row_bytes = C >> 1;
  uint8_t* lAB = smem;                              // [2][A | B]
  float* lsa = (float*)(smem + 2 * STAGE);          // [G][BM]
  float* lsw = lsa + G * BM;                        // [G][BN]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WC, wn = wave % WC;
  // XCD-aware tile order (workgroups are dealt round-robin over the 8 XCDs, each with its own 4 MiB L2):
  // XCD k owns a contiguous band of `cpx` column tiles and walks all row tiles of that band, so its
  // slice of W stays L2-resident and an A tile is re-used by cpx consecutive workgroups.
  const int n_col = (O + BN - 1) / BN, n_row = (T + BM - 1) / BM;
  const int cpx = (n_col + 7) >> 3;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int col_blk = xcd * cpx + local % cpx, row_blk = local / cpx;
  if (col_blk >= n_col || row_blk >= n_row) return;   // uniform over the workgroup
  const int t0 = row_blk * BM, o0 = col_blk * BN;

  // scales of this tile, transposed to [g][row] and widened to fp32
  for (int i = tid; i < G * BM; i += NTHR) {
    const int r = i % BM, g = i / BM;
    lsa[i] = (t0 + r < T) ? (float)sa[(int64_t)(t0 + r) * G + g] : 0.0f;
  }
  for (int i = tid; i < G * BN; i += NTHR) {
    const int r = i % BN, g = i / BN;
    lsw[i] = (o0 + r < O) ? (float)sw[(int64_t)(o0 + r) * G + g] : 0.0f;
  }

  v4f_t acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = v4f_t{0, 0, 0, 0};

  // staging: chunk id = h*NTHR + tid -> (row = id >> 2, kb = id & 3): 4 lanes fetch one row's 64 bytes.
  // Software pipeline: group g+2 is in flight in registers and group g+1 is being written to the other
  // LDS buffer while group g is multiplied; one barrier per group.
  // Rows past the end of A / W are clamped to the last valid row: their products land in output rows /
  // columns the epilogue never stores, and unconditional loads keep the loop free of exec-mask branches.
  static_assert((BM * 4) % NTHR == 0 && (BN * 4) % NTHR == 0, "tile chunks must divide evenly over the threads");
  const uint8_t* pa[HA];
  const uint8_t* pb[HB];
  int la[HA], lb[HB];
#pragma unroll
  for (int h = 0; h < HA; ++h) {
    const int id = h * NTHR + tid, r = id >> 2, kb = id & 3;
    const int t = t0 + r < T ? t0 + r : T - 1;
    la[h] = gemm_lds_off(r, kb);
    pa[h] = A + (int64_t)t * row_bytes + kb * 16;
  }
#pragma unroll
  for (int h = 0; h < HB; ++h) {
    const int id = h * NTHR + tid, r = id >> 2, kb = id & 3;
    const int o = o0 + r < O ? o0 + r : O - 1;
    lb[h] = gemm_lds_off(r, kb);
    pb[h] = W + (int64_t)o * row_bytes + kb * 16;
  }
  u32x4 ga[HA], gb[HB];
  auto fetch = [&](int g) {
#pragma unroll
    for (int h = 0; h < HA; ++h) ga[h] = *(const u32x4*)(pa[h] + g * 64);
#pragma unroll
    for (int h = 0; h < HB; ++h) gb[h] = *(const u32x4*)(pb[h] + g * 64);
  };
  auto stage = [&](int buf) {
    uint8_t* base = lAB + buf * STAGE;
#pragma unroll
    for (int h = 0; h < HA; ++h) *(u32x4*)(base + la[h]) = ga[h];
#pragma unroll
    for (int h = 0; h < HB; ++h) *(u32x4*)(base + ABYTES + lb[h]) = gb[h];
  };
  fetch(0);
  stage(0);
  if (G > 1) fetch(1);
  __syncthreads();   // buffer 0 and the scale tiles are visible
  for (int g = 0; g < G; ++g) {
    const uint8_t* lA = lAB + (g & 1) * STAGE;
    const uint8_t* lB = lA + ABYTES;
    if (g + 1 < G) {   // the other buffer was last read in iteration g-1, a barrier has passed since
      stage((g + 1) & 1);
      if (g + 2 < G) fetch(g + 2);
    }
    v8i_t af[MT], bf[NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const u32x4 v = *(const u32x4*)(lA + gemm_lds_off((wm * MT + m) * 16 + (lane & 15), lane >> 4));
      af[m] = v8i_t{(int)v[0], (int)v[1], (int)v[2], (int)v[3], 0, 0, 0, 0};
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const u32x4 v = *(const u32x4*)(lB + gemm_lds_off((wn * NT + n) * 16 + (lane & 15), lane >> 4));
      bf[n] = v8i_t{(int)v[0], (int)v[1], (int)v[2], (int)v[3], 0, 0, 0, 0};
    }
    v4f_t sa4[MT];
    float sw1[NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) sa4[m] = *(const v4f_t*)(lsa + g * BM + (wm * MT + m) * 16 + 4 * (lane >> 4));
#pragma unroll
    for (int n = 0; n < NT; ++n) sw1[n] = lsw[g * BN + (wn * NT + n) * 16 + (lane & 15)];
    // NT independent MFMAs per tile row, then their scale-and-accumulate in the packed-fp32 VALU: the
    // matrix pipe works on row m+1 while the VALU finishes row m
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      v4f_t d[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n)
        d[n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af[m], bf[n], v4f_t{0, 0, 0, 0}, 4, 4, 0, 0, 0, 0);   // literal zero scales select the unscaled instruction (x1.0, probed)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const v4f_t p = sa4[m] * sw1[n];
        acc[m][n] = __builtin_elementwise_fma(d[n], p, acc[m][n]);
      }
    }
    __syncthreads();   // group g consumed, group g+1 staged
  }

  // epilogue: bias, fp16, transpose each wavefront tile through LDS for 16-byte row stores
  constexpr int WROWS = 16 * MT, WCOLS = 16 * NT, LDW = WCOLS + 8;
  _Float16* lo = (_Float16*)smem + wave * (WROWS * LDW);
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int col = n * 16 + (lane & 15);
      const int o = o0 + wn * WCOLS + col;
      const float b = (bias && o < O) ? (float)bias[o] : 0.0f;
#pragma unroll
      for (int i = 0; i < 4; ++i) lo[(m * 16 + 4 * (lane >> 4) + i) * LDW + col] = (_Float16)(acc[m][n][i] + b);
    }
  __syncthreads();
  constexpr int PIECES = WROWS * (WCOLS / 8);
#pragma unroll
  for (int pass = 0; pass < (PIECES + 63) / 64; ++pass) {
    const int piece = pass * 64 + lane;
    if (piece < PIECES) {
      const int r = piece / (WCOLS / 8), cpc = piece % (WCOLS / 8);
      const int t = t0 + wm * WROWS + r, o = o0 + wn * WCOLS + cpc * 8;
      if (t < T && o + 8 <= O) {
        u32x4 y = *(const u32x4*)(lo + r * LDW + cpc * 8);
        FPQ_GEMM_EPI_VEC(y, epi, t, o, O);
        *(u32x4*)(out + (int64_t)t * O + o) = y;
      } else if (t < T) {
        for (int e = 0; e < 8; ++e)
          if (o + e < O) {
            _Float16 y = lo[r * LDW + cpc * 8 + e];
            FPQ_GEMM_EPI_ONE(y, epi, t, o + e, O);
            out[(int64_t)t * O + o + e] = y;
          }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Second-generation kernel: operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4, no staging
// registers, no ds_write), wavefront tile 16*MT x 16*NT (128 x 64) so that one pair of fragment reads
// feeds more matrix work, two waves per SIMD.  What the measurements in tools/probe/ say about this chip:
//   * v_mfma_f32_16x16x128_f8f6f4 (FP4) issues every ~8.5 ns per SIMD (7.9 PFLOP/s chip-wide);
//   * VALU work does not hide behind it (mfma + 8 v_fma = 18.5 ns vs 8.5 + 12.1 separately), and
//     v_pk_fma_f32 costs as much as two v_fma_f32;
//   * the scaled form (v_mfma_scale_*) pays an extra VALU slot for its scale operands.
// So the per-group scale-and-accumulate (2 multiplies + 1 fma per output element and group) is the real
// bound of this formulation (~19 ns per MFMA, ~3.5 PFLOP/s), and everything else has to stay out of its way.
//
// LDS image of a 16-row x 64-byte block (1 KiB, written by ONE LDS-DMA instruction, lane j -> bytes 16j):
// lane j = 4*q + c fetches row q, 16-byte chunk c ^ pi[q >> 2], pi = (0,2,3,1): four consecutive lanes read
// one row's 64 contiguous bytes from memory, and the fragment reads (lane l: row l & 15, chunk l >> 4,
// ds_read_b128 served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) touch 16 distinct
// 16-byte slots per group (checked exhaustively offline): conflict-free on both sides.
#define FPQ_NOPK __attribute__((target("no-packed-fp32-ops")))   // callees must carry the kernel's target features to be inlined
// __syncthreads() spelled out (the header's inline function does not carry the attribute and would become a call)
#define FPQ_SYNC()                                             \
  do {                                                         \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     \
    __builtin_amdgcn_s_barrier();                              \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");     \
  } while (0)
FPQ_NOPK __device__ __forceinline__ int glds_chunk_perm(int q) { return (0x78 >> ((q >> 2) << 1)) & 3; }   // pi = 0,2,3,1

// Scale tiles -> LDS as fp32 [G][rows]: one thread per tile row (activation rows first, then weight rows) walks
// that row's G consecutive scales; rows past the end of the tensor read as zero (their outputs are never stored).
template <typename Tsw, int BM, int BN, int NTHR>
FPQ_NOPK __device__ __forceinline__ void load_scale_tiles(const _Float16* __restrict__ sa, const Tsw* __restrict__ sw, float* lsa,
                                                 float* lsw, int t0, int o0, int T, int O, int G, int tid) {
  // a row's scales in batches of GB loads, all in flight before the first is stored (round 4; batches of 5 were three
  // dependent trips to L2 for the 15 groups of C = 1920 in a prologue that every tile waits for)
#ifndef FPQ_GEMM_SCALE_BATCH
#define FPQ_GEMM_SCALE_BATCH 16
#endif
  constexpr int GB = FPQ_GEMM_SCALE_BATCH;
  for (int r = tid; r < BM + BN; r += NTHR) {
    if (r < BM) {   // wave-uniform: BM is a multiple of 64
      const bool ok = t0 + r < T;
      const _Float16* src = sa + (int64_t)(ok ? t0 + r : 0) * G;
      for (int g0 = 0; g0 < G; g0 += GB) {
        _Float16 v[GB];
#pragma unroll
#ifdef FPQ_GEMM_SCALE_FAKE   // timing experiment only (wrong results): what the scale tiles' global loads cost the prologue
        for (int i = 0; i < GB; ++i) v[i] = (_Float16)(1.0f + (float)(r & 1));
#else
        for (int i = 0; i < GB; ++i) v[i] = src[g0 + i < G ? g0 + i : G - 1];
#endif
#pragma unroll
        for (int i = 0; i < GB; ++i)
          if (g0 + i < G) lsa[(g0 + i) * BM + r] = ok ? (float)v[i] : 0.0f;
      }
    } else {
      const int c = r - BM;
      const bool ok = o0 + c < O;
      const Tsw* src = sw + (int64_t)(ok ? o0 + c : 0) * G;
      for (int g0 = 0; g0 < G; g0 += GB) {
        Tsw v[GB];
#pragma unroll
#ifdef FPQ_GEMM_SCALE_FAKE
        for (int i = 0; i < GB; ++i) v[i] = (Tsw)(0.01f + 0.001f * (float)(c & 3));
#else
        for (int i = 0; i < GB; ++i) v[i] = src[g0 + i < G ? g0 + i : G - 1];
#endif
#pragma unroll
        for (int i = 0; i < GB; ++i)
          if (g0 + i < G) lsw[(g0 + i) * BN + c] = ok ? (float)v[i] : 0.0f;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// The fc1 tail (round 5): everything between the fc1 GEMM and fc2's GEMM in the reference's FFN
//     h  = F.gelu(fc1(x), approximate="tanh")                              tr/basic_var.py:120      (fp16 tensor, fp32 arithmetic)
//     q  = fp_quant_e1m2_neg_e2m1_pos_per_group_cuda(h, 4, 128)            tr/quant_utils.py:415-452, bound at :991
// as the epilogue of gemm_fp4_glds_kernel: every tile is BN = 128 outputs wide and starts at a multiple of 128, i.e. it
// holds WHOLE quantization groups - one per token row - so both scales of a group are tile-local.
//   * y = half(acc + bias) exactly as the plain epilogue rounds it (the Linear output the reference's GELU sees);
//   * h = half(gelu(float(y))), gelu_tanh_fast below: within one fp16 ulp of torch's F.gelu(y, approximate="tanh") on every
//     fp16 input, checked exhaustively (tests/test_gpu_fc1_fused.py; torch's formula in torch's operation order with the
//     device library's tanh restated, gelu_tanh_like_torch, is bit-equal to torch and three times the instructions);
//   * per token row the maxima of the negative and of the positive side over the lane's four outputs, the 16 lanes of the
//     row (a DPP reduce-scatter) and the two wavefronts that share the group (LDS), one thread per row turns them into the two scales
//     (row_scale16, dual_poison: the arithmetic of rows16_lut_subwave_kernel<DUAL>), and every lane quantizes its own
//     values with quant_pair16_dual = one packed pair of quant_vec16<DUAL>: bit-equal to the stand-alone quantizer on h;
//   * "any NaN in the tensor => the whole result is zero" (the reference's global clamp, tr/quant_utils.py:421-422) keeps
//     its flag + fix-up launch (fpq_kernels.hip, zero_if_flag_kernel): a row that saw a NaN raises the flag.
// LDS: the bucket table behind the scale tiles (staged in the prologue); the maxima exchange and the row scales in the
// stage buffer the last K group does not use.
struct GemmNoFc1 {};
struct GemmFc1 {
  _Float16* h_out;       // nullptr, or fp16 [T, O] receiving h (the tensor the quantizer saw)
  uint32_t* nan_flag;    // nullptr, or the 8-byte scratch of fpq_quant_rows_dual
  Lut16Args a;           // the (e1m2_neg, e2m1_pos) bucket table's arguments ...
  Lut16Tab tab;          // ... and the table itself, by value (as the stand-alone quantizers take it)
};

// (the GELU itself - gelu_tanh_fast, gelu_tanh_like_torch - lives in fpq_fast16.h: the stand-alone fused quantizer uses it too)
#ifdef FPQ_FC1_GELU_TORCH_ORDER   // A/B builds: the bit-equal form
#define FPQ_FC1_GELU gelu_tanh_like_torch
#else
#define FPQ_FC1_GELU gelu_tanh_fast
#endif

// One step of the maxima's reduce-scatter over the 16 lanes of a DPP row: rows r and r + N / 2 are paired, a lane keeps the
// one its bit selects and hands the other to its partner (DPP control CTRL), taking the partner's in return: N rows in, N / 2
// out, each now the maximum over twice as many lanes.  N == 1: plain exchange-and-max.  Packed pairs, unsigned compare.
template <int N, int CTRL>
FPQ_NOPK __device__ __forceinline__ void pk_max_scatter_step(uint32_t* key, bool bit) {
  if constexpr (N == 1) {
    key[0] = pk_max_u16(key[0], (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key[0], CTRL, 0xF, 0xF, true));
  } else {
#pragma unroll
    for (int r = 0; r < N / 2; ++r) {
      const uint32_t keep = bit ? key[r + N / 2] : key[r], send = bit ? key[r] : key[r + N / 2];
      key[r] = pk_max_u16(keep, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)send, CTRL, 0xF, 0xF, true));
    }
  }
}

// one packed pair of quant_vec16<DUAL> (fpq_fast16.h): each half takes the reciprocal and the scale of its sign's side
FPQ_NOPK __device__ __forceinline__ uint32_t quant_pair16_dual(uint32_t wk, const uint16_t* lut, int shift, float ih_n, uint32_t s2_n,
                                                               float ih_p, uint32_t s2_p) {
  const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int)wk, 15, 1), m1 = (uint32_t)((int)wk >> 31);   // all ones: negative
  const float h0 = u2f((fbits16(ih_n) & m0) | (fbits16(ih_p) & ~m0)), h1 = u2f((fbits16(ih_n) & m1) | (fbits16(ih_p) & ~m1));
  const uint32_t mp = pk_ashr_i16(wk, 15);
  const uint32_t sc = (s2_n & mp) | (s2_p & ~mp);
  const uint32_t rb = div_pair16(wk, h0, 0.0f, h1, 0.0f);
  const uint32_t u = pk_sub_u16(rb, pk_lshr_u16(rb, 15));   // negative patterns: magnitude - 1
  return pk_mul_f16(lut_pair16(lut, u, shift), sc);
}

// target("no-packed-fp32-ops"): beside MFMAs a packed fp32 op costs as much as two scalar ones and blocks the issue
// port twice as long (tools/probe/valu_mfma_overlap.hip); with the feature off the compiler emits scalar
// v_mul_f32 / v_fma_f32 and schedules them - and the MFMA hazard wait states - itself.
// XE = GemmNoFc1: the plain Linear (+ gate / residual tail); XE = GemmFc1: the fc1 tail above.
template <typename Tsw, int MT, int NT, typename XE = GemmNoFc1>
__global__ __launch_bounds__(256, (MT * NT > 16 ? 2 : 3)) FPQ_NOPK void gemm_fp4_glds_kernel(const uint8_t* __restrict__ A,
                                                              const _Float16* __restrict__ sa,
                                                              const uint8_t* __restrict__ W, const Tsw* __restrict__ sw,
                                                              const _Float16* __restrict__ bias,
                                                              _Float16* out, int T, int O, int C, GemmEpi epi, XE xe) {
  constexpr bool FC1 = !__is_same(XE, GemmNoFc1);
  constexpr int WR = 2, WC = 2, BM = 16 * MT * WR, BN = 16 * NT * WC, NTHR = 256;
  constexpr int ABLK = BM / 16, BBLK = BN / 16, NBLK = ABLK + BBLK, STAGE = NBLK * 1024;
  static_assert(NBLK % 4 == 0, "blocks are dealt round-robin to the four wavefronts");
  constexpr int PIECES = NBLK / 4;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int G = C >> 7, row_bytes = C >> 1;
  float* lsa = (float*)(smem + 2 * STAGE);   // [G][BM]
  const int Gp = (G + 3) & ~3;               // the scale tiles' LDS-DMA pieces cover up to four groups each: room for a whole last piece
  float* lsw = lsa + Gp * BM;                // [G][BN]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int n_col = (O + BN - 1) / BN, n_row = (T + BM - 1) / BM;
  const int cpx = (n_col + 7) >> 3;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int col_blk = xcd * cpx + local % cpx, row_blk = local / cpx;
  if (col_blk >= n_col || row_blk >= n_row) return;   // uniform over the workgroup
  const int t0 = row_blk * BM, o0 = col_blk * BN;

  // LDS-DMA sources of this wavefront's pieces (block = wave + 4*i of the stage), group 0.
  // The weight rows are DEALT over a wavefront's NT = 4 blocks (round 4): row q of block n holds output 4*q + n of the
  // wavefront's 64, so the lane that holds column q of the NT result tiles holds four CONSECUTIVE outputs 4*q .. 4*q + 3
  // and the epilogue stores 8 bytes per lane, 128 contiguous bytes per 16 lanes, straight from the accumulators.
  static_assert(NT == 4, "the epilogue packs a lane's NT results of one row into one 8-byte store");
  // Addressing (round 4): a UNIFORM 64-bit base per operand (the tile's first row) + a 32-bit lane offset (row inside the
  // tile, clamped to the tensor's last row, and 16-byte chunk) that never changes: the step from group to group (64 bytes)
  // is scalar arithmetic and the load takes scalar base + vector offset.  Spelled in assembly - the compiler folds
  // base + offset into a loop-invariant per-lane pointer and adds the group's 64 bytes to each with a 64-bit vector
  // addition - and WITHOUT a memory clobber (with one, -4 % instead of +3 %: it pins every LDS read of the group behind it;
  // the loads write the stage nobody reads, and the barriers order them).  The compiler does not see these loads: the wait
  // for them is spelled out in front of the barrier that hands a stage to its readers (FPQ_GLDS_WAIT).
  // Issuing a stage's pieces between the MFMAs of the previous one instead of in front of them: 2.5 x slower (measured).
  // A persistent form (one workgroup per resident slot looping over its tiles, the next tile's stage 0 and scale tiles
  // requested in front of the current tile's stores): 5 - 10 % slower than one workgroup per tile (measured, round 4).
  // K-major images (epi.km_w_rows != 0): plane g holds every row's 64 bytes of group g, [G][rows][64], the 16-byte chunks of a row
  // already in the LDS image's order and the weight rows in dealt order - a piece is 1 KiB CONTIGUOUS (8 whole 128-byte lines)
  // instead of 16 rows' half lines row_bytes apart: 70 against 105 cycles to issue, profiles/r05_lds_dma_issue.txt.
  const bool km = epi.km_w_rows != 0;
  const int row_stride = km ? 64 : row_bytes;
  const int64_t a_step = km ? (int64_t)T * 64 : 64, w_step = km ? (int64_t)epi.km_w_rows * 64 : 64;
  const uint8_t* const gbase[2] = {A + (int64_t)t0 * row_stride, W + (int64_t)o0 * row_stride};
  uint32_t voff[PIECES];
  {
    const int q = lane >> 2, kb = km ? (lane & 3) : (lane & 3) ^ glds_chunk_perm(q);
    const int w_rows = km ? epi.km_w_rows : O;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int blk = wave + 4 * i;
      if (blk < ABLK) {
        const int t = t0 + blk * 16 + q;
        voff[i] = (uint32_t)((t < T ? t : T - 1) - t0) * (uint32_t)row_stride + (uint32_t)(kb * 16);
      } else {
        const int wb = blk - ABLK;
        const int o = km ? o0 + wb * 16 + q : o0 + (wb / NT) * (16 * NT) + NT * q + wb % NT;
        voff[i] = (uint32_t)((o < w_rows ? o : w_rows - 1) - o0) * (uint32_t)row_stride + (uint32_t)(kb * 16);
      }
    }
  }
  static_assert(ABLK % 4 == 0, "a wavefront's pieces i < ABLK / 4 are rows of A, the rest rows of W");
#define FPQ_GLDS_ISSUE(g, buf)                                                                                      \
  _Pragma("unroll") for (int i_ = 0; i_ < PIECES; ++i_)                                                             \
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"                                 \
                   :                                                                                                \
                   : "v"(voff[i_]), "s"(gbase[i_ < ABLK / 4 ? 0 : 1] + (g) * (i_ < ABLK / 4 ? a_step : w_step)),                                   \
                     "s"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(smem + (buf) * STAGE +     \
                                                                                       (wave + 4 * i_) * 1024))    \
                   : "m0")
#define FPQ_GLDS_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
  FPQ_GLDS_ISSUE(0, 0);

  if (km) {
    // K-MAJOR SCALE IMAGES (include/fpq.h): fp32 planes [G][rows rounded up to 4] (activations) and [G][km_w_rows] (weights, natural
    // output order) - the tile's scales of a group are 4 BM (4 BN) contiguous bytes, exactly one row of lsa (lsw): they come
    // in by LDS-DMA like the codes, 256 / BM (two) groups per 1 KiB piece, no register round trip, no conversion, no ds_write.
    // Row-major fp16 / fp32 scales cost 90 strided load instructions + as many LDS writes per tile: 8 % of the kernel
    // (profiles/r05_kmajor_ab.txt).  Pieces dealt round-robin over the four wavefronts; the main loop's first wait covers them.
    constexpr int APG = 256 / BM, LPG_A = BM / 4;          // groups per piece, lanes per group (activation side)
    const int Tpad = (T + 3) & ~3, n_a = (G + APG - 1) / APG, n_w = (G + 1) >> 1;
    const float* sa_km = (const float*)sa;
    const float* sw_km = (const float*)sw;
    for (int p = wave; p < n_a + n_w; p += 4) {
      const bool is_a = p < n_a;
      const int g0 = is_a ? p * APG : (p - n_a) * 2;
      const int sub = is_a ? lane / LPG_A : lane >> 5, l4 = is_a ? lane % LPG_A : lane & 31;
      const int grp = g0 + sub < G ? g0 + sub : G - 1;       // (a last piece's surplus groups re-read the last one; their LDS rows are padding)
      const int rows = is_a ? Tpad : epi.km_w_rows, r0 = is_a ? t0 : o0;
      int r4 = r0 + 4 * l4;
      r4 = r4 < rows - 4 ? r4 : rows - 4;
      const uint32_t vo = (uint32_t)(((grp - g0) * rows + (r4 - r0)) * 4);
      const float* sbase = (is_a ? sa_km : sw_km) + ((int64_t)g0 * rows + r0);
      const uint32_t dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(is_a ? lsa + g0 * BM : lsw + g0 * BN);
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(vo), "s"(sbase), "s"(dst) : "m0");
    }
  } else {
    load_scale_tiles<Tsw, BM, BN, NTHR>(sa, sw, lsa, lsw, t0, o0, T, O, G, tid);
    // A wait the COMPILER sees (the builtin, not assembly): its scoreboard still carries the scale loads above, whose last
    // waits it counted without knowing of the stage-0 pieces in the same queue - left like that, it protects their
    // destination registers with vmcnt waits inside the main loop, and those wait for the stage just requested.
    // (It has to stand in THIS branch: behind the join, under a second `if (!km)`, the compiler's scoreboard keeps the loads
    // pending on the path it cannot rule out, and the loop got its vmcnt(0) back - 8 % - profiles/r05_kmajor_ab.txt.)
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  }
  uint16_t* lut = nullptr;
  if constexpr (FC1) {   // the dual quantizer's bucket table, behind the scale tiles; visible after the first barrier of the main loop
    lut = (uint16_t*)(lsw + Gp * BN);
    lut16_stage(lut, xe.tab, xe.a.shift);
  }


  // the tile's bias (four consecutive outputs per lane, see the epilogue) is requested here, not between the last MFMA and
  // the stores
  constexpr int WROWS = 16 * MT, WCOLS = 16 * NT;
  const int o = o0 + wn * WCOLS + NT * (lane & 15);
  const int oc = o < O ? o : O - 4;
  fpq_h4_t bias_h = fpq_h4_t{0, 0, 0, 0};
  if (bias) bias_h = *(const fpq_h4_t*)(bias + oc);

  v4f_t acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = v4f_t{0, 0, 0, 0};

  // fragment read offset inside a block: row lane & 15, chunk lane >> 4
  const int frag_off = ((lane & 15) << 6) + ((((lane >> 4) ^ glds_chunk_perm(lane & 15)) & 3) << 4);
  const int a_off = wm * MT * 1024 + frag_off, b_off = (ABLK + wn * NT) * 1024 + frag_off;
  const int sa_off = wm * MT * 16 + 4 * (lane >> 4), sw_off = wn * NT * 16 + NT * (lane & 15);   // outputs 4q .. 4q+3: tile n holds 4q + n

  for (int g = 0; g < G; ++g) {
    FPQ_GLDS_WAIT();
    FPQ_SYNC();   // stage g has landed; stage g^1's readers are done
    if (g + 1 < G) { FPQ_GLDS_ISSUE(g + 1, (g + 1) & 1); }
    const uint8_t* st = smem + (g & 1) * STAGE;
    // Software pipeline over the tile rows: the ds_reads of row m+1 are issued first, then the NT MFMAs of row m,
    // then the scale-and-accumulate of row m-1 - VALU work that does not depend on the MFMAs in flight, so no
    // hazard wait states are needed and 8 of each MFMA's 16 cycles of vector issue are hidden behind it.
    // (Left to itself the compiler issues each row's reads right before their use and the VALU right behind its
    // own MFMAs, with s_nops in between.)
    u32x4 bq[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) bq[n] = *(const u32x4*)(st + b_off + n * 1024);
    const v4f_t sw1 = *(const v4f_t*)(lsw + g * BN + sw_off);
    u32x4 aq = *(const u32x4*)(st + a_off);
    v4f_t sa4 = *(const v4f_t*)(lsa + g * BM + sa_off);
    v4f_t d_prev[NT], sa4_prev = sa4;
#pragma unroll
    for (int n = 0; n < NT; ++n) d_prev[n] = v4f_t{0, 0, 0, 0};
#pragma unroll
    for (int m = 0; m <= MT; ++m) {
      u32x4 aq_n = aq;
      v4f_t sa4_n = sa4;
      if (m + 1 < MT) {
        aq_n = *(const u32x4*)(st + a_off + (m + 1) * 1024);
        sa4_n = *(const v4f_t*)(lsa + g * BM + sa_off + (m + 1) * 16);
      }
      __builtin_amdgcn_sched_barrier(0);
      v4f_t d[NT];
      if (m < MT) {
        const v8i_t af = v8i_t{(int)aq[0], (int)aq[1], (int)aq[2], (int)aq[3], 0, 0, 0, 0};
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const v8i_t bf = v8i_t{(int)bq[n][0], (int)bq[n][1], (int)bq[n][2], (int)bq[n][3], 0, 0, 0, 0};
          // literal zero scale operands select the unscaled instruction (x 1.0; tools/probe/mfma_fp4_probe.hip)
          d[n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af, bf, v4f_t{0, 0, 0, 0}, 4, 4, 0, 0, 0, 0);
        }
      }
      if (m > 0) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float t = d_prev[n][i] * sa4_prev[i];
            acc[m - 1][n][i] = __builtin_fmaf(t, sw1[n], acc[m - 1][n][i]);
          }
      }
      if (m > 0 && m < MT) {
        // issue order inside the row: one MFMA, then the 8 VALU ops of one finished tile, NT times - an in-order
        // wavefront that issues its MFMAs back to back just waits for the matrix pipe with its VALU idle
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (m < MT) {
#pragma unroll
        for (int n = 0; n < NT; ++n) d_prev[n] = d[n];
        sa4_prev = sa4;
      }
      aq = aq_n;
      sa4 = sa4_n;
    }
  }
  // epilogue, from the registers (round 4): a lane's results of row i in the NT tiles are four consecutive outputs -> + bias,
  // one rounding to fp16, gate / residual, one 8-byte store; 16 lanes write 128 contiguous bytes.  Rounds 1-3 turned the
  // tile through LDS for 16-byte row stores (64 x (add, convert, ds_write_b16), two barriers, the re-read: ~330 of the
  // ~2850 vector instructions of a wavefront's tile at K = 1920).  outs % 8 == 0 and o % 4 == 0: o < O means o + 4 <= O.
  // Loads are unconditional on clamped addresses (a lane past the edge reads what a neighbour reads and stores nothing):
  // a load inside a divergent branch is waited for inside it.
  v4f_t b4 = v4f_t{0, 0, 0, 0};
#pragma unroll
  for (int n = 0; n < NT; ++n) b4[n] = (float)bias_h[n];
  if constexpr (FC1) {
    // (see the comment above GemmFc1)  Row r of the tile = wm * WROWS + m * 16 + 4 * (lane >> 4) + i.
    uint32_t* xch = (uint32_t*)(smem + (G & 1) * STAGE);     // [2 (wn)][BM]: packed (max|h| over h < 0) | (max h over h > 0) << 16
    u32x4* rsc = (u32x4*)(xch + 2 * BM);                     // [BM]: {1 / s_neg, 1 / s_pos, s_neg x 2, s_pos x 2}
    static_assert(2 * BM * 4 + BM * 16 <= STAGE, "exchange + row scales fit the idle stage buffer");
    uint32_t hw[MT][4][2];
    // The two maxima of a row as ONE packed key, both halves compared unsigned: low half = the unsigned maximum of the fp16
    // patterns (the most negative value, or a negative NaN), high half = their signed maximum with the sign bit flipped (the
    // largest positive value, or a positive NaN) - dual_max_acc of fpq_fast16.h, finished by the row's thread further down.
    constexpr int NR = 4 * MT;
    uint32_t key[NR];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float g[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) g[n] = FPQ_FC1_GELU((float)(_Float16)(acc[m][n][i] + b4[n]));
        const uint32_t w0 = f2h2(g[0], g[1]), w1 = f2h2(g[2], g[3]);
        hw[m][i][0] = w0;
        hw[m][i][1] = w1;
        const uint32_t un = pk_max_u16(w0, w1), sg = pk_max_i16(w0, w1) ^ 0x80008000u;
        key[4 * m + i] = pk_max_u16(__builtin_amdgcn_perm(sg, un, 0x05040100u), __builtin_amdgcn_perm(sg, un, 0x07060302u));
      }
      if (xe.h_out) {
        const int t_first = t0 + wm * WROWS + m * 16 + 4 * (lane >> 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (t_first + i < T && o < O)
            __builtin_nontemporal_store(u32x2{hw[m][i][0], hw[m][i][1]}, (u32x2*)(xe.h_out + (int64_t)(t_first + i) * O + oc));
      }
    }
    {
      // maxima over the 16 lanes that share a row: a reduce-scatter (the NR rows of a lane are halved four times, each step
      // one DPP exchange per surviving row: 30 exchanges for 32 rows, where reducing every row over all lanes takes 128) -
      // afterwards lane j of a DPP row holds the finished keys of NR / 16 rows (one row per lane pair for NR = 8), row index
      // = the lane's bits, highest first, then the position in `key`
      const int j = lane & 15;
      pk_max_scatter_step<NR, 0x128>(key, (j & 8) != 0);                               // row_ror:8       partner j ^ 8
      pk_max_scatter_step<(NR >= 2 ? NR / 2 : 1), 0x141>(key, (j & 4) != 0);           // row_half_mirror partner 7 - (j & 7)
      pk_max_scatter_step<(NR >= 4 ? NR / 4 : 1), 0x4E>(key, (j & 2) != 0);            // quad_perm [2,3,0,1]
      pk_max_scatter_step<(NR >= 8 ? NR / 8 : 1), 0xB1>(key, (j & 1) != 0);            // quad_perm [1,0,3,2]
      constexpr int NF = NR >= 16 ? NR / 16 : 1;                                       // finished rows per lane
      constexpr int HALVINGS = NR >= 16 ? 4 : 3;                                       // NR = 8: the last step is a plain exchange
      const int rho0 = (HALVINGS == 4 ? j : (j >> 1)) * NF;                            // row m * 4 + i of key[0]
      uint32_t* dst = xch + wn * BM + wm * WROWS + (rho0 >> 2) * 16 + 4 * (lane >> 4) + (rho0 & 3);
      if constexpr (NF == 2) *(u32x2*)dst = u32x2{key[0], key[1]};                     // rows i, i + 1 of one tile row block
      else dst[0] = key[0];
    }
    FPQ_SYNC();
    if (tid < BM) {   // one thread per token row: the group's two scales
      const uint32_t k = pk_max_u16(xch[tid], xch[BM + tid]);                          // the two wavefronts that share the group
      const uint32_t un = k & 0xFFFFu, sgv = (k >> 16) ^ 0x8000u;
      const uint32_t mn = (un & 0x8000u) ? (un & 0x7FFFu) : 0u, mp = (sgv & 0x8000u) ? 0u : sgv;   // dual_max_finish
      if ((mn > 0x7C00u || mp > 0x7C00u) && xe.nan_flag && t0 + tid < T)   // a NaN in this group (the builtin: atomicOr() is a header function without this kernel's target attribute - it would become a call)
        __hip_atomic_fetch_or(xe.nan_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      RowScale16 sn = row_scale16(mn, xe.a.fneg.gmax, xe.a.inv_gneg), sp = row_scale16(mp, xe.a.fpos.gmax, xe.a.inv_gpos);
      dual_poison(sn, sp);
      rsc[tid] = u32x4{fbits16(sn.inv), fbits16(sp.inv), sn.s16x2, sp.s16x2};
    }
    FPQ_SYNC();
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int r_first = wm * WROWS + m * 16 + 4 * (lane >> 4);
      u32x2 q[4];
      int tq[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32x4 sc = rsc[r_first + i];
        q[i][0] = quant_pair16_dual(hw[m][i][0], lut, xe.a.shift, u2f(sc[0]), sc[2], u2f(sc[1]), sc[3]);
        q[i][1] = quant_pair16_dual(hw[m][i][1], lut, xe.a.shift, u2f(sc[0]), sc[2], u2f(sc[1]), sc[3]);
        tq[i] = t0 + r_first + i;
      }
      FPQ_GEMM_ROWS_STORE(q, t0 + r_first, tq, o, oc);
    }
    return;
  }
  const bool gate_far = epi.gate && epi.rows_per_gate >= WROWS;
  int gq0 = 0, gr0 = 0, gq_last = 0;
  if (epi.gate) {
    const int first = t0 + wm * WROWS + 4 * (lane >> 4);
    gq0 = first / epi.rows_per_gate;
    gr0 = first - gq0 * epi.rows_per_gate;
    gq_last = (T - 1) / epi.rows_per_gate;
  }
  // split output: the tile lies inside ONE part (sp_cols % 128 == 0); its destination, and the batch entry / row of the wavefront's first row
  const int part = epi.sp_cols ? o0 / epi.sp_cols : 0;
  _Float16* const sp_base = part == 0 ? epi.sp_out[0] : part == 1 ? epi.sp_out[1] : epi.sp_out[2];
  const int64_t sp_stride = part == 0 ? epi.sp_stride[0] : part == 1 ? epi.sp_stride[1] : epi.sp_stride[2];
  const int64_t sp_bstride = part == 0 ? epi.sp_bstride[0] : part == 1 ? epi.sp_bstride[1] : epi.sp_bstride[2];
  const int64_t sp_row0 = part == 0 ? epi.sp_row0[0] : part == 1 ? epi.sp_row0[1] : epi.sp_row0[2];
  const bool sp_far = epi.sp_cols && epi.sp_rpb >= WROWS;
  int sb0 = 0, sr0 = 0;
  if (epi.sp_cols) {
    const int first = t0 + wm * WROWS + 4 * (lane >> 4);
    sb0 = first / epi.sp_rpb;
    sr0 = first - sb0 * epi.sp_rpb;
  }
  // (requesting the gate / residual rows one tile row ahead of their use - the compiler may not move a load above a store
  // that could alias it, and the residual may BE the output - was measured: 13 % slower with the fused tail, round 4)
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int t_first = t0 + wm * WROWS + m * 16 + 4 * (lane >> 4);
    fpq_h4_t y[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int n = 0; n < NT; ++n) y[i][n] = (_Float16)(acc[m][n][i] + b4[n]);
    int tc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) tc[i] = t_first + i < T ? t_first + i : T - 1;
    if (epi.gate) {
      fpq_h4_t gt[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // a gate row spans at least the wavefront's rows: ONE division per tile (gq0, gr0 below) and a comparison per row,
        // instead of a division per row - 32 of them were a fifth of a tile's vector instructions
        const int off = gr0 + m * 16 + i;
        int gq = gate_far ? gq0 + (off >= epi.rows_per_gate ? 1 : 0) : tc[i] / epi.rows_per_gate;
        gq = gq < gq_last ? gq : gq_last;
        gt[i] = *(const fpq_h4_t*)(epi.gate + (int64_t)gq * O + oc);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) y[i] = y[i] * gt[i];
    }
    if (epi.resid) {
      fpq_h4_t rs[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) rs[i] = *(const fpq_h4_t*)(epi.resid + (int64_t)tc[i] * O + oc);
#pragma unroll
      for (int i = 0; i < 4; ++i) y[i] = rs[i] + y[i];
    }
    if (epi.sp_cols) {   // uniform: four 8-byte stores to the part's rows (one division per tile while a batch entry spans the wavefront's rows)
      const int oc_l = oc - part * epi.sp_cols;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int off = sr0 + m * 16 + i;
        int bb, ll;
        if (sp_far) {
          const int wrap = off >= epi.sp_rpb ? 1 : 0;
          bb = sb0 + wrap;
          ll = off - wrap * epi.sp_rpb;
        } else {
          bb = tc[i] / epi.sp_rpb;
          ll = tc[i] - bb * epi.sp_rpb;
        }
        if (t_first + i < T && o < O)
          __builtin_nontemporal_store(__builtin_bit_cast(u32x2, y[i]), (u32x2*)(sp_base + ((int64_t)bb * sp_bstride + sp_row0 + ll) * sp_stride + oc_l));
      }
      continue;
    }
    // (non-temporal: a round of tiles writes as much as an XCD's L2 holds - the operands should stay there; +1-2 %)
    FPQ_GEMM_ROWS_STORE(y, t_first, tc, o, oc);
  }
}

#undef FPQ_GLDS_ISSUE
#undef FPQ_GLDS_WAIT

template <int MT, int NT>
struct GemmGldsCfg {
  static constexpr int BM = 32 * MT, BN = 32 * NT;
  static size_t lds(int G) {   // two stages + the scale tiles, groups rounded up to four (the plain epilogue uses no LDS)
    return 2 * (size_t)(BM + BN) * 64 + (size_t)((G + 3) & ~3) * (BM + BN) * 4;
  }
  static size_t lds_fc1(int G, int shift) { return lds(G) + ((size_t)2 << (16 - shift)); }   // + the dual quantizer's bucket table
};

template <int MT, int NT, int WR, int WC>
struct GemmCfg {
  static constexpr int BM = 16 * MT * WR, BN = 16 * NT * WC, NTHR = 64 * WR * WC;
  static size_t lds(int G) {
    size_t main = 2 * (size_t)(BM + BN) * 64 + (size_t)G * (BM + BN) * 4;
    size_t epi = (size_t)WR * WC * (16 * MT) * (16 * NT + 8) * 2;
    return main > epi ? main : epi;
  }
};
