// fpq_gemm_fp8.h - a REAL low-precision consumer for the per-token / per-channel configurations (W6A6, run.sh:7).
// Included by fpq_gemm.hip inside its anonymous namespace, after fpq_gemm_fp4.h (the operand-emitting quantizers: fpq_codes_fp8.h).
//
// With ONE scale per activation row and ONE per weight row the scales leave the K-sum:
//     y[t,o] = s_a[t] * s_w[o] * sum_k La[t,k] * Lw[o,k]
// so the whole contraction is matrix-core work with nothing to do per group (the FP4 per-group kernel spends two
// thirds of its issue slots on the group scales).  Every level of every FP4 / FP6 table of the reference
// (E2M3: multiples of 1/8 up to 7.5, E3M2: 1/16 .. 28, E2M1, E1M2, E3M0) is exactly an OCP FP8 E4M3 number, so the
// operands are stored as one E4M3 byte per element and multiplied by v_mfma_f32_16x16x128_f8f6f4 (products exact,
// fp32 accumulation; K = 1920 sums of E2M3 products stay below 2^24 * 2^-6, i.e. exact).  Probed on hardware
// (tools/probe/mfma_fp8_probe.hip, mfma_fp6_probe.hip): lane l supplies row l & 15, k-block l >> 4 = 32 consecutive
// bytes; the FP6-packed form of the same instruction issues at the same ~19 ns per SIMD as the FP8 form on this
// chip, so the simpler byte layout costs nothing in matrix throughput.
//
// LDS image of a 16-row x 128-byte block: physical 16-byte chunk pc of row r holds logical chunk pc ^ a(r),
// a(r) = ((r >> 1) & 1) + 4 * (r >> 3) (found by exhaustive search): the two ds_read_b128 of every fragment
// (chunks 2kb, 2kb+1 of row l & 15) are conflict-free in all four lane groups, and the LDS-DMA that fills the block
// (two 1 KiB pieces of 8 rows, lane j -> row j >> 3, physical chunk j & 7) reads whole 128-byte lines from memory.
#pragma once

// Epilogue of the row-scaled GEMMs, from the accumulators (round 4, as in gemm_fp4_glds_kernel): the weight rows are dealt
// over a wavefront's NT = 4 tiles on their way into LDS (row q of tile n = output 4q + n of the wavefront's 64), so a lane's
// results of one token in the four tiles are four consecutive outputs: row scale x column scale, bias, one rounding to fp16,
// gate / residual, one non-temporal 8-byte store; 16 lanes write 128 contiguous bytes.  No turn through LDS, no barrier.
// outs % 8 == 0 and o % 4 == 0: o < O means o + 4 <= O.  Loads are unconditional on clamped addresses (a lane past the edge
// reads what a neighbour reads and stores nothing).  The two scale vectors of the tile wait in LDS as fp32 since the
// prologue (FPQ_GEMM_ROWS_STAGE_SCALES: their loads ride with stage 0; fetched in the epilogue, four dependent 2-byte loads
// per tile row stood between the last MFMA and the stores).  Macros: the kernels carry target attributes a callee would need too.
#define FPQ_GEMM_ROWS_STAGE_SCALES(stage_bytes)                                                                     \
  float* lsr = (float*)(smem + 2 * (stage_bytes));   /* [BM] row scales, [BN] column scales, [BN] bias */           \
  float* lsc = lsr + BM;                                                                                            \
  float* lsb = lsc + BN;                                                                                            \
  for (int r_ = tid; r_ < BM + BN; r_ += 256) {                                                                     \
    if (r_ < BM) {                                                                                                  \
      lsr[r_] = (float)sa[t0 + r_ < T ? t0 + r_ : T - 1];                                                           \
    } else {                                                                                                        \
      const int oc0_ = o0 + r_ - BM < O ? o0 + r_ - BM : O - 1;                                                     \
      lsc[r_ - BM] = (float)sw[oc0_];                                                                               \
      lsb[r_ - BM] = bias ? (float)bias[oc0_] : 0.0f;                                                               \
    }                                                                                                               \
  }
#define FPQ_GEMM_ROWS_EPILOGUE()                                                                                    \
  do {                                                                                                              \
    static_assert(NT == 4, "the epilogue packs a lane's NT results of one row into one 8-byte store");              \
    constexpr int WROWS_ = 16 * MT, WCOLS_ = 16 * NT;                                                               \
    const int o_ = o0 + wn * WCOLS_ + NT * (lane & 15);                                                             \
    const int oc_ = o_ < O ? o_ : O - 4;                                                                            \
    const v4f_t sc_ = *(const v4f_t*)(lsc + wn * WCOLS_ + NT * (lane & 15));                                        \
    const v4f_t b_ = *(const v4f_t*)(lsb + wn * WCOLS_ + NT * (lane & 15));                                         \
    /* a gate row that spans the wavefront's rows: one division per tile and a comparison per row (gemm_fp4_glds_kernel) */ \
    const bool gate_far_ = epi.gate && epi.rows_per_gate >= WROWS_;                                                 \
    int gq0_ = 0, gr0_ = 0, gq_last_ = 0;                                                                           \
    if (epi.gate) {                                                                                                 \
      const int first_ = t0 + wm * WROWS_ + 4 * (lane >> 4);                                                        \
      gq0_ = first_ / epi.rows_per_gate;                                                                            \
      gr0_ = first_ - gq0_ * epi.rows_per_gate;                                                                     \
      gq_last_ = (T - 1) / epi.rows_per_gate;                                                                       \
    }                                                                                                               \
    _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                                \
      const int t_first_ = t0 + wm * WROWS_ + m * 16 + 4 * (lane >> 4);                                             \
      int tc_[4];                                                                                                   \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) tc_[i] = t_first_ + i < T ? t_first_ + i : T - 1;               \
      const v4f_t sr_ = *(const v4f_t*)(lsr + wm * WROWS_ + m * 16 + 4 * (lane >> 4));                              \
      fpq_h4_t y_[4];                                                                                               \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                 \
          _Pragma("unroll") for (int n = 0; n < 4; ++n) y_[i][n] = (_Float16)(acc[m][n][i] * (sr_[i] * sc_[n]) + b_[n]); \
      if (epi.gate) {                                                                                               \
        fpq_h4_t g_[4];                                                                                             \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                             \
          const int off_ = gr0_ + m * 16 + i;                                                                       \
          int gq_ = gate_far_ ? gq0_ + (off_ >= epi.rows_per_gate ? 1 : 0) : tc_[i] / epi.rows_per_gate;            \
          gq_ = gq_ < gq_last_ ? gq_ : gq_last_;                                                                    \
          g_[i] = *(const fpq_h4_t*)(epi.gate + (int64_t)gq_ * O + oc_);                                            \
        }                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) y_[i] = y_[i] * g_[i];                                        \
      }                                                                                                             \
      if (epi.resid) {                                                                                              \
        fpq_h4_t r_[4];                                                                                             \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) r_[i] = *(const fpq_h4_t*)(epi.resid + (int64_t)tc_[i] * O + oc_); \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) y_[i] = r_[i] + y_[i];                                        \
      }                                                                                                             \
      FPQ_GEMM_ROWS_STORE(y_, t_first_, tc_, o_, oc_);                                                              \
    }                                                                                                               \
  } while (0)

FPQ_NOPK __device__ __forceinline__ int fp8_chunk_swz(int r) { return ((r >> 1) & 1) + ((r >> 3) << 2); }

template <typename Tsa, typename Tsw, int MT, int NT>
__global__ __launch_bounds__(256, (MT * NT > 16 ? 1 : 2)) FPQ_NOPK void gemm_fp8_rows_kernel(const uint8_t* __restrict__ A,
                                                                       const Tsa* __restrict__ sa,
                                                                       const uint8_t* __restrict__ W,
                                                                       const Tsw* __restrict__ sw,
                                                                       const _Float16* __restrict__ bias,
                                                                       _Float16* out, int T, int O, int C, GemmEpi epi) {
  constexpr int WR = 2, WC = 2, BM = 16 * MT * WR, BN = 16 * NT * WC;
  constexpr int ABLK = BM / 16, BBLK = BN / 16, NBLK = ABLK + BBLK, STAGE = NBLK * 2048;
  constexpr int NPIECE = 2 * NBLK;                 // 1 KiB LDS-DMA pieces per stage (8 rows each)
  static_assert(NPIECE % 4 == 0, "pieces are dealt round-robin to the four wavefronts");
  constexpr int PIECES = NPIECE / 4;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int steps = C >> 7;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int n_col = (O + BN - 1) / BN, n_row = (T + BM - 1) / BM;
  const int cpx = (n_col + 7) >> 3;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int col_blk = xcd * cpx + local % cpx, row_blk = local / cpx;
  if (col_blk >= n_col || row_blk >= n_row) return;   // uniform over the workgroup
  const int t0 = row_blk * BM, o0 = col_blk * BN;

  // LDS-DMA sources: scalar base per operand + 32-bit lane offset, in assembly with explicit waits (as in gemm_fp4_glds_kernel)
  const uint8_t* const gbase[2] = {A + (int64_t)t0 * C, W + (int64_t)o0 * C};
  uint32_t voff[PIECES];
  constexpr int APIECES = 2 * ABLK / 4;            // a wavefront's pieces i < APIECES are rows of A (ABLK % 2 == 0)
  static_assert((2 * ABLK) % 4 == 0, "the A / W boundary falls between two rounds of the four wavefronts");
#pragma unroll
  for (int i = 0; i < PIECES; ++i) {
    const int piece = wave + 4 * i;                 // block piece >> 1, half piece & 1
    const int r = ((piece & 1) << 3) + (lane >> 3); // row inside the 16-row block
    const int c = (lane & 7) ^ fp8_chunk_swz(r);    // logical chunk this lane fetches
    const int blk = piece >> 1;
    if (blk < ABLK) {
      const int t = t0 + blk * 16 + r;
      voff[i] = (uint32_t)((t < T ? t : T - 1) - t0) * (uint32_t)C + (uint32_t)(c * 16);
    } else {
      const int wb = blk - ABLK;                     // weight rows dealt over a wavefront's NT tiles (FPQ_GEMM_ROWS_EPILOGUE)
      const int o = o0 + (wb / NT) * (16 * NT) + NT * r + wb % NT;
      voff[i] = (uint32_t)((o < O ? o : O - 1) - o0) * (uint32_t)C + (uint32_t)(c * 16);
    }
  }
#define FPQ_GLDS8_ISSUE(s, buf)                                                                                     \
  _Pragma("unroll") for (int i_ = 0; i_ < PIECES; ++i_)                                                             \
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"                                 \
                   :                                                                                                \
                   : "v"(voff[i_]), "s"(gbase[i_ < APIECES ? 0 : 1] + (s) * 128),                                   \
                     "s"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(smem + (buf) * STAGE +     \
                                                                                       (wave + 4 * i_) * 1024))    \
                   : "m0")
  FPQ_GLDS8_ISSUE(0, 0);
  FPQ_GEMM_ROWS_STAGE_SCALES(STAGE);

  v4f_t acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = v4f_t{0, 0, 0, 0};

  // fragment offsets inside a block: row lane & 15, logical chunks 2*(lane >> 4) and +1
  const int fr = lane & 15, fc = (lane >> 4) << 1;
  const int off0 = (fr << 7) + (((fc) ^ fp8_chunk_swz(fr)) << 4), off1 = (fr << 7) + (((fc + 1) ^ fp8_chunk_swz(fr)) << 4);
  const int a_base = wm * MT * 2048, b_base = (ABLK + wn * NT) * 2048;

  for (int s = 0; s < steps; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the compiler does not see the LDS-DMA loads above
    FPQ_SYNC();   // stage s has landed; the other buffer's readers are done
    if (s + 1 < steps) { FPQ_GLDS8_ISSUE(s + 1, (s + 1) & 1); }
    const uint8_t* st = smem + (s & 1) * STAGE;
    v8i_t bf[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const u32x4 lo = *(const u32x4*)(st + b_base + n * 2048 + off0), hi = *(const u32x4*)(st + b_base + n * 2048 + off1);
      bf[n] = v8i_t{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const u32x4 lo = *(const u32x4*)(st + a_base + m * 2048 + off0), hi = *(const u32x4*)(st + a_base + m * 2048 + off1);
      const v8i_t af = v8i_t{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
#pragma unroll
      for (int n = 0; n < NT; ++n)
        acc[m][n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af, bf[n], acc[m][n], 0, 0, 0, 0, 0, 0);   // fp8 e4m3, unscaled
    }
  }
#undef FPQ_GLDS8_ISSUE
  FPQ_GEMM_ROWS_EPILOGUE();
}

template <int MT, int NT>
struct GemmFp8Cfg {
  static constexpr int BM = 32 * MT, BN = 32 * NT;
  static size_t lds() {
    return 2 * (size_t)(BM + BN) * 128 + (size_t)(BM + 2 * BN) * 4;   // two stages + row scales, column scales, bias as fp32
  }
};
