// fpq_gemm_fp6.h - the row-scaled GEMM of fpq_gemm_fp8.h with the operands in the 6-bit packed form of the same
// matrix instruction (cbsz = blgp = 2, FP6 E2M3): 96 bytes per row and 128-element K step instead of 128, i.e. 25 %
// less LDS-DMA and fragment traffic in a kernel that is bound by exactly that.  Included after fpq_gemm_fp8.h.
//
// Operand layout, probed on hardware (tools/probe/mfma_fp6_probe.hip): lane l supplies row l & 15, k-block l >> 4 =
// 32 consecutive elements as a 192-bit little-endian string, element j in bits [6j, 6j+6): sign, 2 exponent bits
// (bias 1), 3 mantissa bits - which is just the dense 6-bit packing of the row, 24 bytes per k-block.
//
// LDS image: 32-row "super-blocks" of 96-byte rows (3072 B = three 1 KiB LDS-DMA pieces; piece p, lane j -> 16-byte
// chunk p*64 + j of the super-block = row ci / 6, physical chunk ci % 6).  Physical chunk pc of row r holds logical
// chunk (pc - rot(r)) mod 6, rot(r) = (r >> 3) & 1 (found by exhaustive search): the three ds_read_b64 of a fragment
// (bytes 24*kb + 8t of row l & 15) are conflict-free in both 32-lane groups.
#pragma once

FPQ_NOPK __device__ __forceinline__ int fp6_rot(int r) { return (r >> 3) & 1; }
// The fragment reads MUST stay ds_read_b64 (two 32-lane groups, 64 banks: the layout above is conflict-free for exactly that
// instruction).  Left as plain loads, the compiler pairs the reads of tile rows 1536 bytes apart into ds_read2st64_b64 - two
// accesses served in 16-lane groups over 32 banks (MI355X_MICROARCH.md, LDS) - which conflict: rounds 1 - 4 ran with six of
// them per K step, SQ_LDS_BANK_CONFLICT = 48 cycles per step = 1.5 per MFMA (profiles/r04_pmc_gemm6.txt; the FP4 / FP8 GEMMs
// read 16 bytes with ds_read_b128, which has no paired form).  A volatile access is not merged.
// (spelled with the LDS address space: a volatile access through a generic pointer becomes a flat load)
typedef const volatile __attribute__((address_space(3))) u32x2* fpq_lds_v64_ptr;
#define FPQ_LDS_READ64(ptr) (*(fpq_lds_v64_ptr)(const __attribute__((address_space(3))) void*)(ptr))

// -DFPQ_GEMM6_STAMPS: diagnostic build (tools/build_variant.sh --gemm stamps6 -DFPQ_GEMM6_STAMPS, tools/gemm6_stamps.py): s_memtime
// stamps between the phases of a K step, summed per wavefront in scalar registers and written once at the end to a buffer of
// their own (fpq_debug_gemm6_stamp_buffer; no output value depends on them): where a wavefront's step time goes.  Each stamp
// waits for the scalar-memory counter, which the LDS reads share: read the SHARES.  No stamp executes in a regular build.
#ifdef FPQ_GEMM6_STAMPS
__device__ unsigned long long* g_gemm6_stamps;   // [wavefronts][8]: wait for the stage, barrier, LDS-DMA issue, fragments + MFMAs, prologue, epilogue, steps
#define FPQ_ST6(k)                                                   \
  do {                                                               \
    __builtin_amdgcn_sched_barrier(0);                               \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();      \
    __builtin_amdgcn_sched_barrier(0);                               \
    st6_sum[k] += t_ - st6_last;                                     \
    st6_last = t_;                                                   \
  } while (0)
#else
#define FPQ_ST6(k) do { } while (0)
#endif

template <typename Tsa, typename Tsw, int MT, int NT>
__global__ __launch_bounds__(256, 2) FPQ_NOPK void gemm_fp6_rows_kernel(const uint8_t* __restrict__ A,
                                                                       const Tsa* __restrict__ sa,
                                                                       const uint8_t* __restrict__ W,
                                                                       const Tsw* __restrict__ sw,
                                                                       const _Float16* __restrict__ bias,
                                                                       _Float16* out, int T, int O, int C, GemmEpi epi) {
  constexpr int WR = 2, WC = 2, BM = 16 * MT * WR, BN = 16 * NT * WC;
  static_assert(BM % 32 == 0 && BN % 32 == 0, "tiles are made of 32-row super-blocks");
  constexpr int ASB = BM / 32, BSB = BN / 32, NSB = ASB + BSB, STAGE = NSB * 3072;
  constexpr int NPIECE = 3 * NSB;
  static_assert(NPIECE % 4 == 0 && NPIECE / 4 < 16, "pieces are dealt round-robin to the four wavefronts");
  constexpr int PIECES = NPIECE / 4;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int steps = C >> 7, row_bytes = (C >> 2) * 3;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int n_col = (O + BN - 1) / BN, n_row = (T + BM - 1) / BM;
  const int cpx = (n_col + 7) >> 3;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int col_blk = xcd * cpx + local % cpx, row_blk = local / cpx;
  if (col_blk >= n_col || row_blk >= n_row) return;   // uniform over the workgroup
  const int t0 = row_blk * BM, o0 = col_blk * BN;
#ifdef FPQ_GEMM6_STAMPS
  unsigned long long st6_sum[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long st6_last = __builtin_amdgcn_s_memtime();
#endif

  // LDS-DMA sources: scalar base per operand + 32-bit lane offset, in assembly with explicit waits (as in gemm_fp4_glds_kernel)
  static_assert((3 * ASB) % 4 == 0, "the A / W boundary falls between two rounds of the four wavefronts");
  constexpr int APIECES = 3 * ASB / 4;              // a wavefront's pieces i < APIECES are rows of A
  // K-major images (epi.km_w_rows != 0, as in gemm_fp4_glds_kernel): plane s holds every row's 96 bytes of K step s, [steps][rows][96],
  // chunks already rotated as in the LDS image, weight rows in dealt order: a super-block's three pieces are 3 KiB contiguous
  // (70 against 100 - 150 cycles to issue a piece; profiles/r05_lds_dma_issue.txt, r05_gemm6_stamps.txt).
  const bool km = epi.km_w_rows != 0;
  const int row_stride = km ? 96 : row_bytes;
  const int64_t a_step = km ? (int64_t)T * 96 : 96, w_step = km ? (int64_t)epi.km_w_rows * 96 : 96;
  const uint8_t* const gbase[2] = {A + (int64_t)t0 * row_stride, W + (int64_t)o0 * row_stride};
  const int w_rows = km ? epi.km_w_rows : O;
  uint32_t voff[PIECES];
#pragma unroll
  for (int i = 0; i < PIECES; ++i) {
    const int piece = wave + 4 * i;                 // super-block piece / 3, part piece % 3
    const int sb = piece / 3, ci = (piece % 3) * 64 + lane;
    const int r = ci / 6, pc = ci - 6 * r;          // row inside the super-block, physical chunk
    int c = pc - fp6_rot(r);
    c = c < 0 ? c + 6 : c;                          // logical chunk this lane fetches
    c = km ? pc : c;                                // (the image holds the rotated order)
    if (sb < ASB) {
      const int t = t0 + sb * 32 + r;
      voff[i] = (uint32_t)((t < T ? t : T - 1) - t0) * (uint32_t)row_stride + (uint32_t)(c * 16);
    } else {
      const int ti = 2 * (sb - ASB) + (r >> 4);      // 16-row tile of the weight side; its rows are dealt over a wavefront's
      const int o = km ? o0 + (sb - ASB) * 32 + r : o0 + (ti / NT) * (16 * NT) + NT * (r & 15) + ti % NT;   // NT tiles (FPQ_GEMM_ROWS_EPILOGUE)
      voff[i] = (uint32_t)((o < w_rows ? o : w_rows - 1) - o0) * (uint32_t)row_stride + (uint32_t)(c * 16);
    }
  }
#define FPQ_GLDS6_ONE(s, buf, i_)                                                                                   \
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"                                     \
               :                                                                                                    \
               : "v"(voff[i_]), "s"(gbase[(i_) < APIECES ? 0 : 1] + (s) * ((i_) < APIECES ? a_step : w_step)),                                      \
                 "s"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(smem + (buf) * STAGE +         \
                                                                                   (wave + 4 * (i_)) * 1024))      \
               : "m0")
#define FPQ_GLDS6_ISSUE(s, buf) _Pragma("unroll") for (int i_ = 0; i_ < PIECES; ++i_) FPQ_GLDS6_ONE(s, buf, i_)
  FPQ_GLDS6_ISSUE(0, 0);
  FPQ_GEMM_ROWS_STAGE_SCALES(STAGE);

  v4f_t acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = v4f_t{0, 0, 0, 0};

  // the three 8-byte pieces of this lane's fragment inside a 16-row half of a super-block
  const int fr = lane & 15, kb = lane >> 4;
  int foff[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int b = kb * 24 + 8 * t;
    int pc = (b >> 4) + fp6_rot(fr);
    pc = pc >= 6 ? pc - 6 : pc;
    foff[t] = fr * 96 + pc * 16 + (b & 15);
  }
  const int a_base = wm * MT * 1536, b_base = ASB * 3072 + wn * NT * 1536;   // tile row mt -> 1536 * mt (two per super-block)

  // Two LDS stages, one barrier per step; the LDS-DMA pieces of step s+1 are issued in one burst behind the barrier.
  // History of that choice on mat_qkv [65536 x 1920 -> 5760] (ms): with the compiler's form of the load (per-lane 64-bit
  // pointers) the burst measured 0.81 and one piece after every third MFMA 0.75 - 0.79, so rounds 1 - 3 interleaved;
  // with scalar base + lane offset in assembly (round 4) the burst is the faster one: 0.627 interleaved, 0.603 burst.
  // Also measured: register staging (global_load + ds_write) 0.85 - 0.88, a three-stage ring with counted vmcnt 0.92,
  // requesting tile row m + 1's fragment before the MFMAs of row m (no gain: the SIMD's second wavefront covers it).
  FPQ_ST6(4);
  for (int s = 0; s < steps; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the compiler does not see the LDS-DMA loads
    FPQ_ST6(0);
    FPQ_SYNC();   // stage s has landed; the other buffer's readers are done
    FPQ_ST6(1);
    const uint8_t* st = smem + (s & 1) * STAGE;
#ifndef FPQ_GEMM6_INTERLEAVE   // (-DFPQ_GEMM6_INTERLEAVE: one piece behind every third MFMA instead - re-measured on k-major operands, profiles/r05_kmajor_ab.txt)
    if (s + 1 < steps) { FPQ_GLDS6_ISSUE(s + 1, (s + 1) & 1); }
#endif
    FPQ_ST6(2);
    v8i_t bf[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const uint8_t* p = st + b_base + n * 1536;
      const u32x2 q0 = FPQ_LDS_READ64(p + foff[0]), q1 = FPQ_LDS_READ64(p + foff[1]), q2 = FPQ_LDS_READ64(p + foff[2]);
      bf[n] = v8i_t{(int)q0[0], (int)q0[1], (int)q1[0], (int)q1[1], (int)q2[0], (int)q2[1], 0, 0};
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const uint8_t* p = st + a_base + m * 1536;
      const u32x2 q0 = FPQ_LDS_READ64(p + foff[0]), q1 = FPQ_LDS_READ64(p + foff[1]), q2 = FPQ_LDS_READ64(p + foff[2]);
      const v8i_t af = v8i_t{(int)q0[0], (int)q0[1], (int)q1[0], (int)q1[1], (int)q2[0], (int)q2[1], 0, 0};
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        acc[m][n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af, bf[n], acc[m][n], 2, 2, 0, 0, 0, 0);   // fp6 e2m3, unscaled
        // a scheduling fence after every third MFMA - where the interleaved form issued its pieces: without the fences the
        // compiler's order of the rows' ds_reads and MFMAs is 5 % slower (0.635 against 0.603 ms), measured both ways
        // (a fence after every MFMA, every second or every fourth measures the same; in the FP8 kernel fences cost 2 %)
        if ((m * NT + n) % 3 == 2) {
          __builtin_amdgcn_sched_barrier(0);
#ifdef FPQ_GEMM6_INTERLEAVE
          if ((m * NT + n) / 3 < PIECES && s + 1 < steps) {
            FPQ_GLDS6_ONE(s + 1, (s + 1) & 1, (m * NT + n) / 3);
            __builtin_amdgcn_sched_barrier(0);
          }
#endif
        }
      }
    }
#ifdef FPQ_GEMM6_INTERLEAVE
    if (s + 1 < steps) {
#pragma unroll
      for (int i_ = (MT * NT) / 3; i_ < PIECES; ++i_) FPQ_GLDS6_ONE(s + 1, (s + 1) & 1, i_);
    }
#endif
    FPQ_ST6(3);
  }
#undef FPQ_GLDS6_ISSUE
#undef FPQ_GLDS6_ONE
  FPQ_GEMM_ROWS_EPILOGUE();
#ifdef FPQ_GEMM6_STAMPS
  FPQ_ST6(5);
  if (lane == 0 && g_gemm6_stamps) {
    unsigned long long* dst = g_gemm6_stamps + ((int64_t)blockIdx.x * 4 + wave) * 8;
#pragma unroll
    for (int k = 0; k < 6; ++k) dst[k] = st6_sum[k];
    dst[6] = (unsigned long long)steps;
    dst[7] = 1;
  }
#endif
}

template <int MT, int NT>
struct GemmFp6Cfg {
  static constexpr int BM = 32 * MT, BN = 32 * NT;
  static size_t lds() {
    return 2 * (size_t)(BM + BN) * 96 + (size_t)(BM + 2 * BN) * 4;   // two stages + row scales, column scales, bias as fp32
  }
};
