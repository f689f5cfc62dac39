// fpq_rotate_mfma.h - the online rotate in front of the per-group quantizer with the 128-point Hadamard transform
// on the matrix cores.  Included by fpq_kernels.hip after fpq_fast16.h.
//
// Why: the butterfly form (rotate_quant16_kernel) is bound by vector-instruction issue, not by memory
// (profiles/r02_pmc_rotate_butterfly.txt: 22 VALU instructions per element, 8.5 of them the butterfly, the vector pipe ~80 %
// busy at 0.70 of 8 TB/s where the plain quantizer streams at 0.80).  The reference computes this step as a GEMM in
// the first place (tr/basic_var.py:263,266: torch.matmul(x, Q), Q = blockdiag(diag(D) H128 / sqrt(128))); H128 is
// +-1, exact as fp16 MFMA operands.
//
// One wavefront transforms 16 groups (2048 elements, 4 KiB) at a time:
//   1. four fully coalesced 16-byte loads per lane (vector i * 64 + lane of the tile); signs D applied by xor;
//   2. transpose through LDS into the B-operand layout of v_mfma_f32_16x16x32_f16: lane (g = lane % 16,
//      quarter = lane / 16) needs, for k-step s, the 8 consecutive inputs 32 s + 8 quarter .. of group g = chunk
//      p = 4 s + quarter.  Chunk (g, p) lives at slot p * 16 + (g ^ p) (16 bytes each): the writes (16 lanes = the 16
//      chunks of one group) and the reads (16 lanes = one chunk of 16 groups) both touch 16 different slots mod 16;
//   3. H128[o][b] = (-1)^popcount(o & b) factors over the bits.  Bits 0 .. 4 go through the matrix cores: with
//      o = i + 16 u (+ 32 t), b = k' + 8 quarter + 32 s the exponent is popcount(i & 7 & k') + i3 q0 + u q1 (+ t0 s0
//      + t1 s1): a per-lane +-1 pattern A_u (u = 0, 1), and P[u][s] = A_u . B_s is EIGHT independent v_mfma
//      (16 outputs x 16 groups each, 4 accumulator registers).  Bits 5, 6 are a 4-point butterfly over s on the
//      accumulators: y[u][t] = sum_s (-1)^(t0 s0 + t1 s1) P[u][s], 64 fp32 adds per lane;
//   4. the accumulators leave lane (g, quarter) holding outputs 4 quarter + 16 (u + 2 t) + {0..3} of group g: 32 of
//      its 128; the group maximum is an in-lane v_max3 chain + two v_permlane swaps, the scale is computed once
//      per lane and 32 elements (butterfly form: once per 8), the quantizer runs on packed pairs of adjacent outputs;
//   5. results go back through LDS (8-byte pieces in, 16-byte vectors out; chunk p' of group g at slot
//      g * 16 + (p' ^ g)) and leave as the same coalesced 16-byte stores as the loads.
// Wavefronts are persistent and (fp16 input) issue the loads of their next tile before transforming the current one.
//
// Numerics: +-1 operands - the matrix cores add and subtract the inputs themselves; fp32 sums of fp16 values of like
// magnitude are exact, cancelling outputs included (+-c_h operands, the reference's own GEMM, show the rounding of
// the fp32 accumulator as several fp16 ulp on small outputs), and the one rounding is half(sum * c_h).  Same contract
// as the butterfly form: <= 1 fp16 ulp from the fp64 product, quantization bit-exact on the rotated values produced
// (tests/test_gpu_parity.py::test_rotate_quant_fused).
#pragma once

typedef _Float16 rq_h8_t __attribute__((ext_vector_type(8)));
typedef float rq_f4_t __attribute__((ext_vector_type(4)));

constexpr int kRqTileVec = 256;   // 16 groups x 16 vectors of 8 halves per wavefront

// The wavefront's LDS image (round 3: padded rows instead of xor swizzles).  Round 2 kept chunk (g, p) at slot
// p * 16 + (g ^ p): conflict-free, but every one of the ~24 LDS accesses of a tile had its own xor-ed address - 3 of the
// 14.6 (rotate) / 21.8 (adaLN) vector instructions per element were address arithmetic (profiles/r03_adaln_isa_census.txt).
// Now a group is a row of the image and every access of a phase is ONE lane-constant base + an immediate offset:
//   operand image, 288 bytes per group (18 x 16): chunk p of group g at g * 288 + 16 p.
//       row-order writes (vector i * 64 + lane of the tile = chunk lane % 16 of group 4 i + lane / 16): base + i * 1152,
//       consecutive lanes, consecutive 16 bytes;  B-operand reads (chunk 4 s + lane / 16 of group lane % 16): base + 64 s,
//       dword (8 g + 4 quarter) mod 64: the 16 lanes of each ds_read_b128 lane group ({0-3, 12-15, 20-27}, ...) hit 16
//       different 16-byte slots of the 256-byte bank row.
//   output image, 272 bytes per group (17 x 16): outputs 8 p' .. 8 p' + 7 of group g at g * 272 + 16 p'.
//       8-byte pieces in (piece c of lane (g, quarter) = outputs 16 c + 4 quarter ..): base + 32 c (two lanes per bank
//       pair - the store's own transfer time covers it);  16-byte vectors out in row order: base + i * 1088.
constexpr int kRqInStride = 288, kRqOutStride = 272;
constexpr int kRqSmoothMax = 2560;   // channels of a smoothing vector kept in LDS by the rotate kernel (VAR: 1024 .. 2304)
constexpr int kRqImageBytes = 16 * kRqInStride;           // 4608 per wavefront
constexpr int kRqImageVec = kRqImageBytes / 16;

struct RqLaneAddr {
  int in_w;    // row-order 16-byte writes of the operand image
  int in_w8;   // the same image written by half-chunks (fp32 input: 8 bytes per lane and load)
  int in_r;    // B-operand reads
  int out_w;   // 8-byte pieces of the output image
  int out_r;   // row-order 16-byte reads of the output image
  int lane16;  // lane * 16: global offset of this lane's vector inside a 1 KiB row-order slab
};
// INS: bytes per group of the operand image.  288 is conflict-free; 272 (the output image's stride: one image size for
// both) costs the B-operand reads one extra LDS cycle per instruction and saves 256 bytes per wavefront - which is what
// lets the adaLN producer fit a fifth workgroup per CU at C = 1920 (fpq_adaln.h).
template <int INS = kRqInStride>
__device__ __forceinline__ RqLaneAddr rq_lane_addr(int lane) {
  const int g = lane & 15, q = lane >> 4;
  RqLaneAddr a;
  a.in_w = q * INS + g * 16;
  a.in_w8 = (lane >> 5) * INS + ((lane >> 1) & 15) * 16 + (lane & 1) * 8;
  a.in_r = g * INS + q * 16;
  a.out_w = g * kRqOutStride + q * 8;
  a.out_r = q * kRqOutStride + g * 16;
  a.lane16 = lane * 16;
  return a;
}

// -DFPQ_ISA_CENSUS: phase markers for tools/isa_census.py --phases (a comment line in the assembly between two
// scheduling barriers, so that every instruction is counted in the phase it belongs to); nothing in a regular build
#ifdef FPQ_ISA_CENSUS
#define FPQ_PHASE(name)                        \
  do {                                         \
    __builtin_amdgcn_sched_barrier(0);         \
    asm volatile("; PHASE " name);             \
    __builtin_amdgcn_sched_barrier(0);         \
  } while (0)
#else
#define FPQ_PHASE(name) do { } while (0)
#endif

// The LDS addresses of a phase are lane constants + xor patterns: hoisted out of the tile loop they would pin
// registers for the sake of a few xors per tile.  An empty asm makes the lane index opaque at the start of a phase.
__device__ __forceinline__ int rq_opaque(int lane) {
  asm volatile("" : "+v"(lane));
  return lane;
}

struct HadOperand {
  u32x4 a[2];   // [u]: output bit 4
};

// The A operands are lane constants: (-1)^(popcount(i & 7 & k') + i3 q0 + u q1) as +-1.0 pairs, i = lane % 16, quarter
// q = lane / 16, eight k' per register quad.  Built at compile time into a 2 KiB table in the code object (immutable):
// a wavefront fetches its 32 bytes per lane with two loads at kernel start instead of ~30 vector instructions - which in
// the adaLN producer are paid per workgroup, i.e. per 3 - 4 rows.
struct HadTable {
  uint32_t w[64][8];
};
constexpr uint32_t had_parity(uint32_t v) { return (v ^ (v >> 1) ^ (v >> 2)) & 1u; }
constexpr HadTable make_had_table() {
  HadTable t = {};
  for (int lane = 0; lane < 64; ++lane) {
    const uint32_t i = (uint32_t)lane & 15u, quarter = (uint32_t)lane >> 4;
    const uint32_t f0 = ((i >> 3) & quarter & 1u) ? 0x80008000u : 0u;   // i3 q0
    const uint32_t f1 = f0 ^ ((quarter & 2u) ? 0x80008000u : 0u);       // + q1 for u = 1
    for (uint32_t w = 0; w < 4; ++w) {
      const uint32_t b0 = had_parity((i & 7u) & (2u * w)), b1 = had_parity((i & 7u) & (2u * w + 1u));
      const uint32_t base = 0x3C003C00u ^ (b0 << 15) ^ (b1 << 31);      // +-1.0, +-1.0
      t.w[lane][w] = base ^ f0;
      t.w[lane][4 + w] = base ^ f1;
    }
  }
  return t;
}
__device__ const HadTable kHadTable = make_had_table();

__device__ __forceinline__ HadOperand had_operand(int lane) {
  const u32x4* p = (const u32x4*)kHadTable.w[lane];
  HadOperand h;
  h.a[0] = p[0];
  h.a[1] = p[1];
  return h;
}

// 16 groups of 128 sign-applied fp16 inputs (operand image in `img`) -> this lane's 32 rotated outputs of group
// lane % 16 as packed fp16 words yw[c][r], c = u + 2 t: outputs 16 c + 4 (lane / 16) + 2 r, + 1; returns the
// maximum |sum| before the scaling by c_h.  `in_r` = RqLaneAddr::in_r.
__device__ __forceinline__ float hadamard128_mfma(const char* img, int in_r, const HadOperand& ha, float c_h,
                                                  uint32_t (&yw)[8][2]) {
  FPQ_PHASE("mfma");
  rq_f4_t acc[2][4];
  const rq_f4_t zero = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const rq_h8_t b = __builtin_bit_cast(rq_h8_t, *(const u32x4*)(img + in_r + 64 * s));
#pragma unroll
    for (int u = 0; u < 2; ++u)
      acc[u][s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(rq_h8_t, ha.a[u]), b, zero, 0, 0, 0);
  }
  // The accumulators' first readers must be instructions the compiler sees (the adds below): its hazard recognizer
  // inserts the wait states a matrix-core result needs before a VALU read, but does not look into inline assembly
  // (a v_fma_mixlo_f16 straight on the accumulators read them two k-steps stale).
  FPQ_PHASE("butterfly4_max_round");
  float m = 0.0f;
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 4; v += 2) {
      float y[4][2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const float p0 = acc[u][0][v + e], p1 = acc[u][1][v + e], p2 = acc[u][2][v + e], p3 = acc[u][3][v + e];
        const float q00 = p0 + p1, q10 = p0 - p1;   // Q[t0][s1]
        const float q01 = p2 + p3, q11 = p2 - p3;
        y[0][e] = q00 + q01;   // t = t0 + 2 t1
        y[1][e] = q10 + q11;
        y[2][e] = q00 - q01;
        y[3][e] = q10 - q11;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(y[t][0])), __builtin_fabsf(y[t][1]));
        yw[u + 2 * t][v >> 1] = mul2_to_h2(y[t][0], y[t][1], c_h);
      }
    }
  return m;
}

// A tile is addressed through a buffer resource whose range is the live part of the tile: loads beyond it return
// zeros, stores beyond it are dropped - no per-vector predicate, no 64-bit address arithmetic in the vector pipe.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rq_rsrc(const void* tile_ptr, int live_bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)tile_ptr, 0, live_bytes < 0 ? 0 : live_bytes, 0x00020000);
}
constexpr int kRqNt = 2;   // cache policy: non-temporal

// this lane's 16 words (8-byte pieces c) into the output image, then 4 coalesced 16-byte vectors per lane out.
// SKIP15: the image holds 15 groups only (4080 of a 4096-byte slot) - the four lanes of group 15 write nothing (their
// pieces would land in the next wavefront's image); what the row-order reads fetch beyond the slot goes to stores that
// the buffer range drops.
template <bool SKIP15 = false>
__device__ __forceinline__ void rq_store_tile(char* img, const uint32_t (&w)[8][2], __amdgpu_buffer_rsrc_t dst,
                                              const RqLaneAddr& la) {
  FPQ_PHASE("store_tile");
  if (!SKIP15 || (la.lane16 & 0xF0) != 0xF0) {
#pragma unroll
    for (int c = 0; c < 8; ++c) *(u32x2*)(img + la.out_w + 32 * c) = u32x2{w[c][0], w[c][1]};
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const u32x4 o = *(const u32x4*)(img + la.out_r + i * (4 * kRqOutStride));
    __builtin_amdgcn_raw_buffer_store_b128(o, dst, la.lane16 + i * 1024, 0, kRqNt);
  }
  __builtin_amdgcn_wave_barrier();
}

// one tile's raw input, every load a fully coalesced 16 bytes per lane.  fp16: vector i * 64 + lane, i = 0 .. 3, = chunk
// (8 elements) i * 64 + lane.  fp32: 4-float vector n * 64 + lane, n = 0 .. 7, = half lane & 1 of chunk 32 n + lane / 2
// (32 bytes per lane would leave each load instruction half of every 128-byte line, fpq_fast32.h): the lane converts
// its half-chunks and they meet their other halves in the LDS image.
template <typename Tin>
struct RqRaw {
  u32x4 w[sizeof(Tin) == 2 ? 4 : 8];
};

template <typename Tin>
__device__ __forceinline__ void rq_load_tile(__amdgpu_buffer_rsrc_t src, int lane, RqRaw<Tin>& raw) {
#pragma unroll
  for (int i = 0; i < (sizeof(Tin) == 2 ? 4 : 8); ++i)
    raw.w[i] = __builtin_amdgcn_raw_buffer_load_b128(src, lane * 16 + i * 1024, 0, kRqNt);
}

// lut_pair16 with one instruction less: bit shift-1 of both patterns is cleared at once, after which the byte offsets
// are a bit-field extract and a shift
__device__ __forceinline__ uint32_t rq_lut_pair(const uint16_t* lut, uint32_t u, int shift) {
  const uint32_t u2 = u & ~(0x10001u << (shift - 1));
  const uint32_t off0 = __builtin_amdgcn_ubfe(u2, (uint32_t)(shift - 1), (uint32_t)(17 - shift)), off1 = u2 >> (15 + shift);
  h2_t q;
  q.x = *(const _Float16*)((const char*)lut + off0);
  q.y = *(const _Float16*)((const char*)lut + off1);
  return __builtin_bit_cast(uint32_t, q);
}

// FP4 operand output (fpq_gemm_fp4.h: E2M1 nibbles, low nibble first, + one fp16 scale per group) from the lane's 16
// rotated words.  `lut` holds the code table.  A lane's eight 4-element pieces are 16 bits each and belong at bytes
// 8 c + 2 quarter of its group's 64: 2-byte LDS writes (dword index xor-swizzled with bits 2, 3 of the group: the 64
// lanes of a write hit 32 different dwords, two lanes each), then every lane reads 16 bytes = chunk lane % 4 of group
// lane / 4 and the tile's 1 KiB of codes leaves as ONE coalesced store per lane; lanes 0 .. 15 store the scales.
// code_off(lane): where this lane's 16 bytes (chunk lane % 4 of the tile's group lane / 4) go inside codes_dst - lane * 16 for
// row-major codes (the tile's 16 groups are 1 KiB in a row), rq_km4_off for a k-major image (codes_dst then spans the whole
// image); 0xFFFFFFFF: outside every buffer, the store is dropped.  A callable evaluated on an opaque copy of the lane index right
// in front of the store: computed ahead of the conversion it costs the one register the kernel does not have (tests/test_no_spill.py).
__device__ __forceinline__ uint32_t rq_km4_off(uint32_t t, uint32_t g, bool live, int lane, uint32_t km_rows) {
  return live ? km4_off(t, g, (uint32_t)lane & 3u, km_rows) : 0xFFFFFFFFu;
}
// 16 bytes of operand codes out, non-temporal in both layouts (measured for the k-major image too: with plain stores the adaLN
// producer is 8 - 13 % slower at 10 000+ rows, profiles/r05_kmajor_ab.txt - unlike the stand-alone FP6 quantizer, fpq_codes_fp6.h).
__device__ __forceinline__ void rq_store_code_chunk(const u32x4& o, __amdgpu_buffer_rsrc_t dst, uint32_t off) {
  __builtin_amdgcn_raw_buffer_store_b128(o, dst, off, 0, kRqNt);
}
// the tile's 16 group scales (lane u < 16 holds group u's): fp16 at byte 2 u of scales_dst (row-major [rows][G]), or - scale_off(u)
// with bit 31 set - as fp32 at that byte offset of the k-major scale image [G][rows rounded up to 4] (0xFFFFFFFF: no such group)
template <typename OffFn>
__device__ __forceinline__ void rq_store_scale(const RowScale16& s, __amdgpu_buffer_rsrc_t scales_dst, int lane, OffFn scale_off) {
  if (lane < 16) {
    const uint32_t off = scale_off(lane);
    if (off & 0x80000000u) {
      if (off != 0xFFFFFFFFu)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, (float)__builtin_bit_cast(_Float16, (uint16_t)(s.s16x2 & 0xFFFFu))),
                                              scales_dst, off & 0x7FFFFFFFu, 0, 0);
    } else {
      __builtin_amdgcn_raw_buffer_store_b16((uint16_t)(s.s16x2 & 0xFFFFu), scales_dst, off, 0, 0);
    }
  }
}
template <typename OffFn, typename ScaleFn>
__device__ __forceinline__ void rq_store_codes(u32x4* buf, const uint32_t (&yw)[8][2], const RowScale16& s,
                                               const uint16_t* lut, int shift, __amdgpu_buffer_rsrc_t codes_dst,
                                               __amdgpu_buffer_rsrc_t scales_dst, int lane, OffFn code_off, ScaleFn scale_off) {
  lane = rq_opaque(lane);
  const int g = lane & 15, quarter = lane >> 4;
  const int swz = ((g >> 2) & 3) << 2;
  uint16_t* b16 = (uint16_t*)buf;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    uint32_t w[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const uint32_t rb = div_pair16(yw[c][rr], s.inv, s.inv_lo, s.inv, s.inv_lo);
      const uint32_t u = pk_sub_u16(rb, pk_lshr_u16(rb, 15));
      w[rr] = rq_lut_pair(lut, u, shift);                      // two codes: bits 0 .. 3 and 16 .. 19
    }
    const uint32_t piece = (w[0] & 0xFu) | ((w[0] >> 12) & 0xF0u) | ((w[1] & 0xFu) << 8) | ((w[1] >> 4) & 0xF000u);
    b16[g * 32 + (((2 * c + (quarter >> 1)) ^ swz) << 1) + (quarter & 1)] = (uint16_t)piece;
  }
  __builtin_amdgcn_wave_barrier();
  lane = rq_opaque(lane);
  const int gg = lane >> 2, j = lane & 3;
  const u32x4 o = buf[gg * 4 + (j ^ ((gg >> 2) & 3))];
  rq_store_code_chunk(o, codes_dst, code_off(lane));
  rq_store_scale(s, scales_dst, lane, scale_off);
  __builtin_amdgcn_wave_barrier();
}

// The same output with the codes taken from the FP4 conversion hardware (fpq_fast16.h: e2m1_codes_hw): no table, no
// lookups, no nibble packing - the conversion writes the two codes of a pair into a byte of the lane's code registers.
// The lane's eight 16-bit pieces (4 elements each) go to bytes 8 c + 2 quarter of its group's 64; groups are rows of 80
// bytes in the image (20 dwords: the 64 lanes of a 2-byte write hit 32 dwords, two lanes each sharing one), every
// address a lane constant + an immediate.
constexpr int kRqCodeStride = 80;
template <typename OffFn, typename ScaleFn>
__device__ __forceinline__ void rq_store_codes_hw(char* img, const uint32_t (&yw)[8][2], const RowScale16& s,
                                                  __amdgpu_buffer_rsrc_t codes_dst, __amdgpu_buffer_rsrc_t scales_dst,
                                                  int lane, OffFn code_off, ScaleFn scale_off) {
  const int g = lane & 15, quarter = lane >> 4;
  const int cw = g * kRqCodeStride + 2 * quarter;
#pragma unroll
  for (int k = 0; k < 4; ++k) {        // pieces 2 k and 2 k + 1: outputs 32 k + 4 quarter .. + 3 and 32 k + 16 + 4 quarter .. + 3
    uint32_t w = 0;
    w = e2m1_codes_hw<0>(w, div_pair16(yw[2 * k][0], s.inv, 0.f, s.inv, 0.f));
    w = e2m1_codes_hw<1>(w, div_pair16(yw[2 * k][1], s.inv, 0.f, s.inv, 0.f));
    w = e2m1_codes_hw<2>(w, div_pair16(yw[2 * k + 1][0], s.inv, 0.f, s.inv, 0.f));
    w = e2m1_codes_hw<3>(w, div_pair16(yw[2 * k + 1][1], s.inv, 0.f, s.inv, 0.f));
    w = e2m1_codes_canon(w);
    *(uint16_t*)(img + cw + 16 * k) = (uint16_t)w;
    *(uint16_t*)(img + cw + 16 * k + 8) = (uint16_t)(w >> 16);
  }
  __builtin_amdgcn_wave_barrier();
  const u32x4 o = *(const u32x4*)(img + (lane >> 2) * kRqCodeStride + (lane & 3) * 16);   // chunk lane % 4 of group lane / 4
  rq_store_code_chunk(o, codes_dst, code_off(rq_opaque(lane)));
  rq_store_scale(s, scales_dst, rq_opaque(lane), scale_off);
  __builtin_amdgcn_wave_barrier();
}

// Per-token operand outputs of the adaLN producer (one scale per row, fpq_gemm_fp8.h / fpq_gemm_fp6.h): `lut` holds the
// code table, `s` the row's scale.  E4M3 bytes: a lane's eight pieces are one dword each, at dword 4 c + quarter of its
// group's 32 (index xor-swizzled with bits 1 .. 3 of the group: 64 lanes, 64 banks); the tile's 2 KiB leave as two
// 16-byte stores per lane.
__device__ __forceinline__ void rq_store_codes8(u32x4* buf, const uint32_t (&yw)[8][2], const RowScale16& s,
                                                const uint16_t* lut, int shift, __amdgpu_buffer_rsrc_t dst, int lane) {
  lane = rq_opaque(lane);
  const int g = lane & 15, quarter = lane >> 4;
  const int swz = ((g >> 1) & 7) << 2;
  uint32_t* b32 = (uint32_t*)buf;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    uint32_t w[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const uint32_t rb = div_pair16(yw[c][rr], s.inv, s.inv_lo, s.inv, s.inv_lo);
      const uint32_t u = pk_sub_u16(rb, pk_lshr_u16(rb, 15));
      w[rr] = rq_lut_pair(lut, u, shift);                      // two code bytes: bits 0 .. 7 and 16 .. 23
    }
    b32[g * 32 + ((4 * c + quarter) ^ swz)] = __builtin_amdgcn_perm(w[1], w[0], 0x06040200u);
  }
  __builtin_amdgcn_wave_barrier();
  lane = rq_opaque(lane);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n = i * 64 + lane, gg = n >> 3, j = n & 7;
    const u32x4 o = buf[gg * 8 + (j ^ ((gg >> 1) & 7))];
    __builtin_amdgcn_raw_buffer_store_b128(o, dst, lane * 16 + i * 1024, 0, kRqNt);
  }
  __builtin_amdgcn_wave_barrier();
}

// Dense 6-bit codes (E2M3): a piece is 24 bits at byte 12 c + 3 quarter of its group's 96: one 2-byte and one 1-byte LDS
// write, in the order the address parity asks for; the tile's 1.5 KiB leave as 16-byte stores (lanes 0 .. 31 two).
template <typename OffFn>
__device__ __forceinline__ void rq_store_codes6(u32x4* buf, const uint32_t (&yw)[8][2], const RowScale16& s,
                                                const uint16_t* lut, int shift, __amdgpu_buffer_rsrc_t dst, int lane,
                                                OffFn off) {
  lane = rq_opaque(lane);
  const int g = lane & 15, quarter = lane >> 4;
  const bool odd = (quarter & 1) != 0;
  uint8_t* b8 = (uint8_t*)buf;
  const int base = g * 96 + 3 * quarter;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    uint32_t w[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const uint32_t rb = div_pair16(yw[c][rr], s.inv, s.inv_lo, s.inv, s.inv_lo);
      const uint32_t u = pk_sub_u16(rb, pk_lshr_u16(rb, 15));
      w[rr] = rq_lut_pair(lut, u, shift);                      // two 6-bit codes: bits 0 .. 5 and 16 .. 21
    }
    const uint32_t p24 = (w[0] & 0x3Fu) | ((w[0] >> 10) & 0xFC0u) | ((w[1] & 0x3Fu) << 12) | ((w[1] >> 16 & 0x3Fu) << 18);
    const int at = base + 12 * c;                               // odd quarters sit at odd addresses
    b8[odd ? at : at + 2] = (uint8_t)(odd ? p24 : p24 >> 16);
    *(uint16_t*)(b8 + (odd ? at + 1 : at)) = (uint16_t)(odd ? p24 >> 8 : p24);
  }
  __builtin_amdgcn_wave_barrier();
  lane = rq_opaque(lane);
  rq_store_code_chunk(buf[lane], dst, off(lane));   // chunk `lane` of the tile's 96: row-major lane * 16
  if (lane < 32) rq_store_code_chunk(buf[64 + lane], dst, off(64 + lane));   // chunk 64 + lane: row-major 1024 + lane * 16
  __builtin_amdgcn_wave_barrier();
}

#ifndef FPQ_ROT_WAVES
#define FPQ_ROT_WAVES 6
#endif

// CODES: `out` receives packed E2M1 codes (4 bytes per 8 elements), r.code_scales one fp16 scale per group; the staged
// table is the code table
// HW4 (E2M1 values or FP4 operands): levels / codes from the FP4 conversion hardware (fpq_fast16.h) - no table lookups
// SMOOTH (the GALT vector applied in front of the rotation: 32 more floats per lane and tile in flight) takes 5 wavefronts
// per SIMD = 96 registers: at 80 it spilled 3 - 8 of them (tests/test_no_spill.py reads every kernel's metadata).
template <typename Tin, bool EMIT, bool SMOOTH, bool CODES = false, bool HW4 = false>
__global__ __launch_bounds__(kBlock, SMOOTH ? FPQ_ROT_WAVES - 1 : FPQ_ROT_WAVES) void rotate_quant_mfma_kernel(const void* __restrict__ xv,
                                                                                 u32x4* __restrict__ out,
                                                                                 u32x4* __restrict__ rot_out,
                                                                                 int64_t n_vec, RotArgs r, Lut16Args a,
                                                                                 Lut16Tab tab) {
  uint16_t* lut = nullptr;
  if constexpr (!HW4) {
    __shared__ __attribute__((aligned(16))) uint16_t lut_s[kLutLdsEntries];
    lut = lut_s;
  }
  __shared__ u32x4 xpose[kBlock / 64][kRqImageVec];   // 4.5 KiB per wavefront, private to it
  // SMOOTH: the vector (<= kRqSmoothMax channels: every VAR width) is staged in LDS once per workgroup; a tile's chunk of
  // it is two ds_read_b128 at (column chunk) * 32 bytes.  Round 2 read it from global memory behind a 64-bit modulo per
  // vector, inside the tile's dependency chain: 280 us per [65536 x 1920] against 80 without a vector
  // (profiles/r03_pmc_rotate_smooth_before.txt, the first measurement of this instantiation).  Wider rows keep global loads.
  constexpr int kSmoothLds = SMOOTH ? kRqSmoothMax : 4;
  __shared__ __attribute__((aligned(16))) float smooth_s[kSmoothLds];
  const uint32_t vpr32 = (uint32_t)r.vec_per_row;
  const bool smooth_in_lds = SMOOTH && vpr32 * 8u <= (uint32_t)kRqSmoothMax;
  if constexpr (SMOOTH) {
    if (smooth_in_lds)
      for (uint32_t i = threadIdx.x; i < vpr32 * 2u; i += kBlock) ((u32x4*)smooth_s)[i] = ((const u32x4*)r.smooth)[i];
  }
#ifndef FPQ_ROT_PREFETCH16
#define FPQ_ROT_PREFETCH16 1
#endif
#ifndef FPQ_ROT_PREFETCH32   // fp32 input: 32 more registers take the kernel from 8 to 5 wavefronts per SIMD, measured 72 - 75 us against 64 - 65 without
#define FPQ_ROT_PREFETCH32 0
#endif
  constexpr bool PREFETCH = (sizeof(Tin) == 2 ? FPQ_ROT_PREFETCH16 != 0 : FPQ_ROT_PREFETCH32 != 0) && !EMIT;   // the emitting form is for tests and calibration dumps
  constexpr int VW = sizeof(Tin) == 2 ? 1 : 2;           // 16-byte words per input vector
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* img = (char*)xpose[wave];
  u32x4* buf = xpose[wave];                        // the code-emitting epilogues keep their own (smaller) layouts
  const int n_tiles = (int)((n_vec + kRqTileVec - 1) / kRqTileVec);
  const int tile_step = (int)gridDim.x * (kBlock / 64);
  int tile = (int)blockIdx.x * (kBlock / 64) + wave;
  const int lg = lane & 15;
  const uint32_t sb = (r.sign[lg >> 2] >> ((lg & 3) * 8)) & 0xFFu;
  uint32_t sx[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) sx[k] = (((sb >> (2 * k)) & 1u) << 15) | (((sb >> (2 * k + 1)) & 1u) << 31);
  auto rem_of = [&](int t) {
    const int64_t left = n_vec - (int64_t)t * kRqTileVec;
    return (int)(left < 0 ? 0 : left > kRqTileVec ? kRqTileVec : left);
  };
  auto in_rsrc = [&](int t) {
    return rq_rsrc((const u32x4*)xv + (int64_t)t * (kRqTileVec * VW), rem_of(t) * (16 * VW));
  };

  RqRaw<Tin> raw;
  if (PREFETCH) rq_load_tile<Tin>(in_rsrc(tile), lane, raw);
  if constexpr (!HW4) lut16_stage(lut, tab, a.shift);
  if constexpr (!HW4 || SMOOTH) __syncthreads();   // the table / the smoothing vector (workgroup-wide, once); everything below is private to the wavefront
  const HadOperand ha = had_operand(lane);
  // column chunk of vector (tile, i, lane) = (tile * 256 + i * 64 + lane) mod vpr: the tile's part once per pass in the
  // scalar unit's width, the rest in 32 bits
  auto smooth_chunk = [&](uint32_t base_mod, int off) {
    const uint32_t c = (base_mod + (uint32_t)off) % vpr32;
    u32x4 s0, s1;
    if (smooth_in_lds) {
      s0 = *(const u32x4*)(smooth_s + c * 8);
      s1 = *(const u32x4*)(smooth_s + c * 8 + 4);
    } else {
      s0 = *(const u32x4*)(r.smooth + (size_t)c * 8);
      s1 = *(const u32x4*)(r.smooth + (size_t)c * 8 + 4);
    }
    struct { u32x4 a, b; } out = {s0, s1};
    return out;
  };

  // The body as a lambda, run once in front of the loop: the compiler merges its s_waitcnt bookkeeping over the edges
  // into the loop header, and the entry edge (prologue loads, nothing after them) would make the wait for the
  // prefetched tile "at most 3 .. 0 operations outstanding" - which on the back edge drains the four stores issued
  // just before it, every iteration.  With the first pass peeled both edges carry "loads, then four stores" and the
  // wait becomes vmcnt(7 .. 4): the stores of a tile complete under the next tile's work.
  auto pass = [&]() {
    const int64_t base_vec = (int64_t)tile * kRqTileVec;
    const int rem = rem_of(tile);
    if (!PREFETCH) rq_load_tile<Tin>(in_rsrc(tile), lane, raw);
    // 1. + 2.: (smooth,) sign, into the operand image (lane-constant addresses are rebuilt per phase from an opaque
    // lane index: two or three instructions, instead of registers held across the whole tile at 6 wavefronts per SIMD)
    const RqLaneAddr la1 = rq_lane_addr(rq_opaque(lane));
    const uint32_t base_mod = SMOOTH ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)base_vec % (uint64_t)vpr32)) : 0u;
    if constexpr (sizeof(Tin) == 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        u32x4 w = raw.w[i];
        if (SMOOTH) {   // h = half(float(x) * s): the widening rides on the multiply (x * s - 0 == x * s, signed zeros included)
          const auto sv = smooth_chunk(base_mod, i * 64 + lane);
          w[0] = f2h2(fmaf_h_lo(w[0], u2f(sv.a[0]), -0.0f), fmaf_h_hi(w[0], u2f(sv.a[1]), -0.0f));
          w[1] = f2h2(fmaf_h_lo(w[1], u2f(sv.a[2]), -0.0f), fmaf_h_hi(w[1], u2f(sv.a[3]), -0.0f));
          w[2] = f2h2(fmaf_h_lo(w[2], u2f(sv.b[0]), -0.0f), fmaf_h_hi(w[2], u2f(sv.b[1]), -0.0f));
          w[3] = f2h2(fmaf_h_lo(w[3], u2f(sv.b[2]), -0.0f), fmaf_h_hi(w[3], u2f(sv.b[3]), -0.0f));
        }
        *(u32x4*)(img + la1.in_w + i * (4 * kRqInStride)) = u32x4{w[0] ^ sx[0], w[1] ^ sx[1], w[2] ^ sx[2], w[3] ^ sx[3]};
      }
    } else {            // fp32 producer output: h = half(x * s), half-chunks of 8 bytes
      const int lane_w = rq_opaque(lane);
      const int hsel = lane_w & 1, k2 = lane_w >> 1;
      const uint32_t sb2 = (r.sign[(k2 & 15) >> 2] >> (((k2 & 15) & 3) * 8 + 4 * hsel)) & 0xFu;
      const uint32_t sy[2] = {((sb2 & 1u) << 15) | (((sb2 >> 1) & 1u) << 31), (((sb2 >> 2) & 1u) << 15) | (((sb2 >> 3) & 1u) << 31)};
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        float f[4] = {u2f(raw.w[n][0]), u2f(raw.w[n][1]), u2f(raw.w[n][2]), u2f(raw.w[n][3])};
        if (SMOOTH) {
          const auto sv = smooth_chunk(base_mod, 32 * n + k2);
          const u32x4 sp = hsel ? sv.b : sv.a;
#pragma unroll
          for (int k = 0; k < 4; ++k) f[k] *= u2f(sp[k]);
        }
        // half h = lane & 1 of chunk 32 n + lane / 2 = chunk (lane / 2) % 16 of group 2 n + lane / 32
        *(u32x2*)(img + la1.in_w8 + n * (2 * kRqInStride)) =
            u32x2{(f2h(f[0]) | (f2h(f[1]) << 16)) ^ sy[0], (f2h(f[2]) | (f2h(f[3]) << 16)) ^ sy[1]};
      }
    }
    if (PREFETCH)   // the next tile, in flight during 3. - 5.
      rq_load_tile<Tin>(in_rsrc(tile + tile_step), lane, raw);
    __builtin_amdgcn_wave_barrier();

    // 3. + 4.
    uint32_t yw[8][2];
    const float mf = hadamard128_mfma(img, rq_lane_addr(rq_opaque(lane)).in_r, ha, r.c_h, yw);
    __builtin_amdgcn_wave_barrier();
    if (EMIT) rq_store_tile(img, yw, rq_rsrc(rot_out + base_vec, rem * 16), rq_lane_addr(rq_opaque(lane)));
    // rounding is monotonic: half(c_h max |sum|) == max |half(c_h sum)|.  v_max drops NaN where the maximum of the integer patterns
    // (and torch's amax) keeps it; every output of a group contains every input, so one non-finite input makes ALL
    // outputs of the group non-finite: one output per lane tells whether the (rare) pattern scan is needed.
    uint32_t m = mul2_to_h2(mf, 0.0f, r.c_h) & 0xFFFFu;
    if (__builtin_expect((yw[0][0] & 0x7C00u) == 0x7C00u, 0)) {
      m = 0;
#pragma unroll
      for (int c = 0; c < 8; ++c) m = pk_max_u16(m, pk_max_u16(yw[c][0] & 0x7FFF7FFFu, yw[c][1] & 0x7FFF7FFFu));
      const uint32_t lo = m & 0xFFFFu, hi = m >> 16;
      m = lo > hi ? lo : hi;
    }
    {   // the other three quarters of the group: lanes ^ 16, ^ 32, ^ 48
      auto sw = __builtin_amdgcn_permlane16_swap(m, m, false, false);
      m = sw[0] > sw[1] ? sw[0] : sw[1];
      sw = __builtin_amdgcn_permlane32_swap(m, m, false, false);
      m = sw[0] > sw[1] ? sw[0] : sw[1];
    }
    RowScale16 s = row_scale16(m, a.fpos.gmax, a.inv_gpos);
    if constexpr (HW4) scale_nan_if_not_finite(s);
    if constexpr (CODES) {
      // row-major: the tile's 16 groups are 1 KiB in a row behind base_vec; k-major: every group has its own (row, group) slot
      const __amdgpu_buffer_rsrc_t cdst = r.km_rows ? rq_rsrc(out, (int)(r.km_rows * r.km_gpr.d * 64u)) : rq_rsrc((const uint32_t*)out + base_vec, rem * 4);
      const auto coff = [&](int ln) -> uint32_t {
        if (!r.km_rows) return (uint32_t)ln * 16u;
        const uint32_t gi = (uint32_t)(base_vec >> 4) + ((uint32_t)ln >> 2), t = fast_div_q(gi, r.km_gpr);
        return rq_km4_off(t, gi - t * r.km_gpr.d, (int64_t)gi * 16 < n_vec, ln, r.km_rows);
      };
      // scales: fp16 [rows][G] behind the tile's first group, or (k-major) fp32 [G][rows rounded up to 4]
      const uint32_t tpad = (r.km_rows + 3u) & ~3u;
      const __amdgpu_buffer_rsrc_t sdst = r.km_rows ? rq_rsrc(r.code_scales, (int)(tpad * r.km_gpr.d * 4u)) : rq_rsrc(r.code_scales + (base_vec >> 4), rem / 8);
      const auto soff = [&](int u) -> uint32_t {
        if (!r.km_rows) return (uint32_t)u * 2u;
        const uint32_t gi = (uint32_t)(base_vec >> 4) + (uint32_t)u, t = fast_div_q(gi, r.km_gpr);
        return (int64_t)gi * 16 < n_vec ? ((((gi - t * r.km_gpr.d) * tpad + t) * 4u) | 0x80000000u) : 0xFFFFFFFFu;
      };
      if constexpr (HW4)
        rq_store_codes_hw(img, yw, s, cdst, sdst, rq_opaque(lane), coff, soff);
      else
        rq_store_codes(buf, yw, s, lut, a.shift, cdst, sdst, lane, coff, soff);
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const uint32_t rb = div_pair16(yw[c][rr], s.inv, s.inv_lo, s.inv, s.inv_lo);
          if constexpr (HW4) {
            yw[c][rr] = pk_fma0_f16(e2m1_levels_hw(rb), s.s16x2);
          } else {
            const uint32_t u = pk_sub_u16(rb, pk_lshr_u16(rb, 15));
            yw[c][rr] = pk_mul_f16(rq_lut_pair(lut, u, a.shift), s.s16x2);
          }
        }
      // 5.
      rq_store_tile(img, yw, rq_rsrc(out + base_vec, rem * 16), rq_lane_addr(rq_opaque(lane)));
    }
  };
  if (tile < n_tiles) {
    pass();
    for (tile += tile_step; tile < n_tiles; tile += tile_step) pass();
  }
}
