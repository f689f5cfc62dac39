// fpq_common.h - format descriptors, dtype traits and the closed-form rounding shared by the two
// translation units of libfpq_hip.so (fpq_kernels.hip: quantizers; fpq_gemm.hip: matrix-core consumers).
// Everything lives in an anonymous namespace: each translation unit gets its own copy.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "fpq.h"

// ---------------------------------------------------------------------------------
// Process-wide experiment switches (include/fpq.h, fpq_set_option).  ONE table of ints for both translation units
// (defined in fpq_kernels.hip), filled from the environment once when the library is loaded and changed afterwards only
// through fpq_set_option: the launch paths read an int, never the environment.
// ---------------------------------------------------------------------------------
#define FPQ_OPTION_LIST(X)                                                                                             \
  X(FPQ_NO_HW4, 1)          /* E2M1: keep the bucket table instead of the FP4 conversion hardware */                   \
  X(FPQ_NO_HW6, 1)          /* E2M3 / E3M2: keep the bucket table instead of the FP6 conversion hardware */            \
  X(FPQ_NO_FAST32, 1)       /* fp32 rows: IEEE division + closed form instead of approximate-then-verify */            \
  X(FPQ_ADALN_NO_PAIR2, 1)  /* adaLN producer at C = 1024: one row per tile */                                         \
  X(FPQ_ROT_BUTTERFLY, 1)   /* the butterfly form of the 128-point transform instead of the matrix cores */            \
  X(FPQ_ADALN_V1, 1)        /* first-generation adaLN kernel */                                                        \
  X(FPQ_ADALN_NO_TIGHT, 1)                                                                                             \
  X(FPQ_NO_WAVE_ROWS, 1)    /* long fp16 rows: one workgroup per row even when a wavefront would hold it */            \
  X(FPQ_GEMM_CFG, 0)        /* FP4 GEMM tiling: 0..2 register-staged, 10 / 20 / 30 LDS-DMA 256x128 / 128x128 / 64x128 */ \
  X(FPQ_GEMM6_CFG, 0)       /* FP6 GEMM tiling: 0 128x128, 1 256x128 */                                                 \
  X(FPQ_GEMM8_CFG, 0)       /* FP8 GEMM tiling: 0 128x128, 1 256x128 */                                                 \
  X(FPQ_ROT_WGS, 0)         /* rotate_quant: workgroups per generation */                                              \
  X(FPQ_ADALN_ROWS, 0)      /* adaLN producer: rows per workgroup */                                                   \
  X(FPQ_ADALN_TAIL, 0)      /* adaLN producer: rows at the end of the grid cut into finer tiers */                     \
  X(FPQ_ADALN_GRID, 0)      /* first-generation adaLN kernel: grid cap */                                              \
  X(FPQ_ADALN_LANES, 0)     /* adaLN producer: 64 (wavefront per row) or 256 (workgroup per row) */                    \
  X(FPQ_BIGTAB_RPB, 0)                                                                                                 \
  X(FPQ_BIGTAB_U, 0)                                                                                                   \
  X(FPQ_BIGTAB_CAP, 0)
enum FpqOptId {
#define FPQ_OPT_ENUM(name, is_flag) OPT_##name,
  FPQ_OPTION_LIST(FPQ_OPT_ENUM)
#undef FPQ_OPT_ENUM
  FPQ_OPT_COUNT
};
extern "C" __attribute__((visibility("hidden"))) int fpq_option_table[FPQ_OPT_COUNT];
// a flag: set and not zero; a number: its value, or `dflt` while unset
static inline int fpq_opt_raw(int id) { return __atomic_load_n(&fpq_option_table[id], __ATOMIC_RELAXED); }
static inline bool fpq_flag(int id) { const int v = fpq_opt_raw(id); return v != FPQ_OPTION_DEFAULT && v != 0; }
static inline bool fpq_opt_set(int id) { return fpq_opt_raw(id) != FPQ_OPTION_DEFAULT; }
static inline int fpq_opt(int id, int dflt) { const int v = fpq_opt_raw(id); return v == FPQ_OPTION_DEFAULT ? dflt : v; }

// Helpers of fpq_kernels.hip that fpq_gemm.hip's fused fc1 epilogue needs too (internal to the library, not exported):
// the bucket table + arguments of a dual-format quantizer (copied into caller-provided Lut16Args / Lut16Tab objects, whose
// layout the two translation units share through fpq_fast16.h), and the "any NaN => the whole result is zero" fix-up launch.
extern "C" __attribute__((visibility("hidden"))) int fpq_internal_dual_lut(int neg_table, int pos_table, void* args_out, size_t args_bytes,
                                                                           void* tab_out, size_t tab_bytes);
extern "C" __attribute__((visibility("hidden"))) int fpq_internal_zero_if_flag(void* out, int64_t n_bytes, void* scratch, void* stream);

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int kBlock = 256;   // 4 wavefronts of 64
constexpr int kMaxBlocks = 256 * 8;  // 256 CUs x 8 resident workgroups

// ---------------------------------------------------------------------------------
// Format descriptors (wave-uniform kernel arguments -> SGPRs)
// ---------------------------------------------------------------------------------
struct Fmt {
  float kmin;       // smallest normal level 2^emin (levels below it are equally spaced)
  float inv_step0;  // 1 / spacing below kmin
  float step0;      // spacing below kmin = kmin / 2^M
  float gmax;       // largest level
  float limit;      // 102400 + gmax: beyond it the scan selects nothing
  uint32_t half_add;   // 1 << (22 - M)
  uint32_t keep_mask;  // ~((1 << (23 - M)) - 1)
  int32_t zero_code;   // index of 0.0 in the sorted de-duplicated SYMMETRIC table
  int32_t mshift;      // 23 - M
  uint32_t kmin_code_base;  // (bits(kmin) >> mshift) - 2^M : level index = (bits>>mshift) - base
  int32_t argmin;     // 1: torch.argmin semantics of the reference's pure-torch path (ties to the
                      //    SMALLER value, NaN/Inf -> table[0], no reach limit) instead of the scan's
  float preclamp;     // > 0: x = clamp(x, -preclamp, preclamp) first (the reference's clamp(x,-3,3))
};

struct TableInfo {
  const char* name;
  int symmetric;  // usable with fpq_quant_rows
  float kmin;
  int mbits;
  float gmax;
  int n_pos;  // number of non-negative levels (incl. 0)
};

// E2M1: bias 1 -> subnormal step .5 below 1.0.  E1M2: bias 1 -> step .25 everywhere.
// E3M0: bias 3 -> smallest normal .25.  E2M3: bias 1.  E3M2: bias 3.
// INT_NEG: integers 0..32 = fixed point, expressed as kmin = 32, M = 5 (step 1).
const TableInfo kTables[FPQ_NUM_TABLES] = {
    {"e2m1", 1, 1.0f, 1, 6.0f, 8},      {"e1m2", 1, 1.0f, 2, 1.75f, 8},
    {"e3m0", 1, 0.25f, 0, 16.0f, 8},    {"e2m3", 1, 1.0f, 3, 7.5f, 32},
    {"e3m2", 1, 0.25f, 2, 28.0f, 32},   {"e1m2_neg", 0, 1.0f, 2, 1.75f, 8},
    {"e2m1_pos", 0, 1.0f, 1, 6.0f, 8},  {"int_neg", 0, 32.0f, 5, 32.0f, 33},
    {"e2m3_pos", 0, 1.0f, 3, 7.5f, 32}, {"e2m1_neg", 0, 1.0f, 1, 6.0f, 8},
};

inline uint32_t f2u(float f) {
  uint32_t u;
  __builtin_memcpy(&u, &f, 4);
  return u;
}

Fmt make_fmt(int id) {
  const TableInfo& t = kTables[id];
  Fmt f;
  f.kmin = t.kmin;
  f.step0 = t.kmin / (float)(1 << t.mbits);
  f.inv_step0 = 1.0f / f.step0;
  f.gmax = t.gmax;
  f.limit = 102400.0f + t.gmax;
  f.half_add = 1u << (22 - t.mbits);
  f.keep_mask = ~((1u << (23 - t.mbits)) - 1u);
  f.zero_code = t.n_pos - 1;
  f.mshift = 23 - t.mbits;
  f.kmin_code_base = (f2u(t.kmin) >> f.mshift) - (1u << t.mbits);
  f.argmin = 0;
  f.preclamp = 0.0f;
  return f;
}

// positive levels of a table, ascending (host)
int pos_levels(int id, float* out) {
  const TableInfo& t = kTables[id];
  int n = 0;
  if (id == FPQ_INT_NEG) {
    for (int v = 0; v <= 32; ++v) out[n++] = (float)v;
    return n;
  }
  float step0 = t.kmin / (float)(1 << t.mbits);
  for (int m = 0; m < (1 << t.mbits); ++m) out[n++] = m * step0;
  for (float base = t.kmin; base <= t.gmax; base *= 2.0f)
    for (int m = 0; m < (1 << t.mbits); ++m) {
      float v = base * (1.0f + (float)m / (float)(1 << t.mbits));
      if (v <= t.gmax) out[n++] = v;
    }
  return n;
}

// ---------------------------------------------------------------------------------
// Device helpers
// ---------------------------------------------------------------------------------
__host__ __device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
__host__ __device__ __forceinline__ uint32_t fbits(float f) { return __builtin_bit_cast(uint32_t, f); }

__host__ __device__ __forceinline__ float h2f(uint32_t hbits) {
  _Float16 h = __builtin_bit_cast(_Float16, (uint16_t)hbits);
  return (float)h;
}
__host__ __device__ __forceinline__ uint32_t f2h(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  // The value must exist as a rounded fp32 before it is narrowed: torch materialises
  // the fp32 product and then casts (two roundings when x is fp32).  Without the
  // barrier LLVM folds fmul + fptrunc into v_fma_mixlo_f16, which rounds once.
  asm volatile("" : "+v"(f));
#endif
  _Float16 h = (_Float16)f;  // v_cvt_f16_f32, round to nearest even
  return (uint32_t)__builtin_bit_cast(uint16_t, h);
}

// Magnitude of the nearest level for r = |xn| (r >= 0 or NaN); neg01 = 1 when xn < 0
// (a tie then resolves to the smaller magnitude).  NaN, Inf and r > limit give 0.
__host__ __device__ __forceinline__ float quant_mag(float r, uint32_t neg01, const Fmt& f) {
  // binades at or above kmin: keep M mantissa bits, round half up (or half down)
  uint32_t nb = (fbits(r) + f.half_add - neg01) & f.keep_mask;
  float qn = u2f(nb);
  // below kmin: equally spaced levels
  float t = r * f.inv_step0;  // exact (power of two)
  float fl = __builtin_floorf(t);
  float fr = t - fl;          // exact
  bool up = neg01 ? (fr > 0.5f) : (fr >= 0.5f);
  float qs = (fl + (up ? 1.0f : 0.0f)) * f.step0;
  float q = (r >= f.kmin) ? qn : qs;
  q = __builtin_fminf(q, f.gmax);
  return (r <= f.limit) ? q : 0.0f;
}

// index of a level magnitude q (exactly a level) among the non-negative levels
__device__ __forceinline__ int level_index(float q, const Fmt& f) {
  int hi = (int)((fbits(q) >> f.mshift) - f.kmin_code_base);
  int lo = (int)(q * f.inv_step0);
  return (q >= f.kmin) ? hi : lo;
}

// --- dtype traits: T = storage type of x, all arithmetic in fp32 with T's roundings ---
template <typename T>
struct DT;
template <>
struct DT<_Float16> {
  static constexpr int kVec = 8;  // elements per 16-byte lane load
  static __device__ __forceinline__ float get(const u32x4& v, int i) {
    uint32_t w = v[i >> 1];
    return h2f((i & 1) ? (w >> 16) : (w & 0xFFFFu));
  }
  static __device__ __forceinline__ void put(u32x4& v, int i, float p) {
    uint32_t h = f2h(p);
    uint32_t w = v[i >> 1];
    v[i >> 1] = (i & 1) ? ((w & 0x0000FFFFu) | (h << 16)) : ((w & 0xFFFF0000u) | h);
  }
  static __device__ __forceinline__ uint32_t absbits(float xf) { return f2h(xf) & 0x7FFFu; }
  static __device__ __forceinline__ float from_absbits(uint32_t b) { return h2f(b); }
  static __device__ __forceinline__ float round(float v) { return h2f(f2h(v)); }
  static __device__ __forceinline__ bool bits_nan(uint32_t b) { return b > 0x7C00u; }
};
template <>
struct DT<float> {
  static constexpr int kVec = 4;
  static __device__ __forceinline__ float get(const u32x4& v, int i) { return u2f(v[i]); }
  static __device__ __forceinline__ void put(u32x4& v, int i, float p) { v[i] = fbits(p); }
  static __device__ __forceinline__ uint32_t absbits(float xf) { return fbits(xf) & 0x7FFFFFFFu; }
  static __device__ __forceinline__ float from_absbits(uint32_t b) { return u2f(b); }
  static __device__ __forceinline__ float round(float v) { return v; }
  static __device__ __forceinline__ bool bits_nan(uint32_t b) { return b > 0x7F800000u; }
};

template <typename T>
__device__ __forceinline__ float load_scalar(const T* p) {
  return (float)(*p);
}
template <typename T>
__device__ __forceinline__ void store_scalar(T* p, float v) {
  asm volatile("" : "+v"(v));  // see f2h
  *p = (T)v;
}

// scale = (T)(absmax / gmax), returned widened to fp32
template <typename T>
__device__ __forceinline__ float scale_of(uint32_t amax_bits, float gmax) {
  return DT<T>::round(DT<T>::from_absbits(amax_bits) / gmax);
}

// One element of a symmetric-table row.  s = scale (already rounded to T).
template <typename T>
__device__ __forceinline__ float quant_sym(float xf, float s, const Fmt& f) {
  float xn = DT<T>::round(xf / s);
  uint32_t neg = (xn < 0.0f) ? 1u : 0u;
  if (f.argmin) {
    // tr/quant_utils.py:209-230: first minimal index = the smaller value on a tie, i.e. down in
    // magnitude for xn > 0 and up for xn < 0; an all-NaN / all-Inf distance row gives index 0
    float r = fabsf(xn);
    Fmt g = f;
    g.limit = __builtin_inff();
    float qm = quant_mag(r, neg ^ 1u, g);
    float q = (neg && qm != 0.0f) ? -qm : qm;
    if (!(r < __builtin_inff())) q = -f.gmax;
    return q * s;
  }
  float qm = quant_mag(fabsf(xn), neg, f);
  float q = (neg && qm != 0.0f) ? -qm : qm;  // the table's zero is +0.0
  return q * s;                              // fp32 product; 0*inf and 0*nan poison the row
}

// One element of a dual-format row (neg table for x <= 0, pos table for x > 0).
template <typename T>
__device__ __forceinline__ float quant_dual(float xf, float sn, float sp, const Fmt& fn, const Fmt& fp) {
  bool isn = xf <= 0.0f, isp = xf > 0.0f;  // NaN: neither
  if (fn.argmin) {
    // fp_quant_e1m2_neg_e2m1_pos_per_group (tr/quant_utils.py:381-412), the pure-torch twin: BOTH halves of
    // every element go through quantize_to_nearest_grid (the other half's input is 0), argmin takes the first
    // minimal index (the smaller value on a tie) and index 0 for a NaN / +-Inf input - so a group without
    // negatives (scale_neg = 0, 0/0) contributes table_neg[0] = -gmax_neg to every element, as in the reference.
    Fmt gn = fn, gp = fp;
    gn.limit = gp.limit = __builtin_inff();
    const float a = DT<T>::round((isn ? xf : 0.0f) / sn);   // <= 0 or NaN
    const float b = DT<T>::round((isp ? xf : 0.0f) / sp);   // >= 0 or NaN
    const float ra = fabsf(a);
    float qa = quant_mag(ra, 0u, gn);                        // tie -> smaller value = larger magnitude
    qa = (qa != 0.0f) ? -qa : 0.0f;
    if (!(ra < __builtin_inff())) qa = -fn.gmax;
    float qb = quant_mag(b, 1u, gp);                         // tie -> smaller value
    if (!(fabsf(b) < __builtin_inff())) qb = 0.0f;           // table_pos[0]
    const float q = qa + qb;
    return q * (isn ? sn : sp);
  }
  float qn = 0.0f, qp = 0.0f;
  if (isn) {
    float xn = DT<T>::round(xf / sn);
    float qm = quant_mag(fabsf(xn), 1u, fn);
    qn = (qm != 0.0f) ? -qm : 0.0f;
  }
  if (isp) {
    float xn = DT<T>::round(xf / sp);
    qp = quant_mag(xn, 0u, fp);
  }
  float a = qn * sn;
  float b = qp * sp;
  return a + b;
}

// clamp to +-clip with torch.clamp(Tensor bounds) NaN rules
__device__ __forceinline__ float clamp_like_torch(float xf, float clip, bool clip_nan) {
  if (clip_nan) return __builtin_nanf("");
  if (xf != xf) return xf;
  return fminf(fmaxf(xf, -clip), clip);
}

template <int LANES>
__device__ __forceinline__ uint32_t lanes_max(uint32_t v) {
#pragma unroll
  for (int m = LANES / 2; m >= 1; m >>= 1) {
    uint32_t o = (uint32_t)__shfl_xor((int)v, m, 64);
    v = v > o ? v : o;
  }
  return v;
}

struct DualArgs {
  Fmt fneg, fpos;
  const void* clip_absmax;  // device scalar or nullptr
  float clip_strength;
  uint32_t* nan_flag;       // nullptr, or device word that is OR-ed with 1 when an input element is NaN
};

template <typename T>
__device__ __forceinline__ float clip_value(const DualArgs& d, bool* is_nan) {
  float am = load_scalar<T>((const T*)d.clip_absmax);
  float c = DT<T>::round(d.clip_strength * am);
  *is_nan = (c != c);
  return c;
}


__device__ __forceinline__ uint32_t block_max(uint32_t v, uint32_t* sh) {
  v = lanes_max<64>(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();  // protect sh from the previous use
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  uint32_t r = sh[0];
#pragma unroll
  for (int i = 1; i < kBlock / 64; ++i) r = r > sh[i] ? r : sh[i];
  return r;
}

// ---------------------------------------------------------------------------------
// K-major operand images (include/fpq.h): where a producer's 16-byte chunk goes
// ---------------------------------------------------------------------------------
// n / d for n < 2^31 and 1 <= d < 2^31 by one multiply-high: m = ceil(2^(32 + s) / d) with s = floor(log2 d) (exact for
// n < 2^(32 + s) / d, which is > 2^31); a power of two has m == 0 and is a shift.
struct FastDiv {
  uint32_t d, m, s;
};
inline FastDiv fast_div(uint32_t d) {
  FastDiv f{d, 0, 0};
  while ((2u << f.s) <= d && f.s < 31) ++f.s;
  if ((d & (d - 1)) != 0) f.m = (uint32_t)((((uint64_t)1 << (32 + f.s)) + d - 1) / d);
  return f;
}
__device__ __forceinline__ uint32_t fast_div_q(uint32_t n, const FastDiv& f) {
  return f.m ? __umulhi(n, f.m) >> f.s : n >> f.s;
}
// the permutations of the GEMMs' LDS images (fpq_gemm_fp4.h glds_chunk_perm, fpq_gemm_fp6.h fp6_rot), restated for the producers
__host__ __device__ __forceinline__ uint32_t km4_perm(uint32_t row) { return (0x78u >> (((row & 15u) >> 2) << 1)) & 3u; }
__host__ __device__ __forceinline__ uint32_t km6_rot(uint32_t row) { return (row >> 3) & 1u; }
// FP4 (64 bytes per row and group): byte offset of logical chunk c (0..3) of group g of row t in an image of `rows` rows
__device__ __forceinline__ uint32_t km4_off(uint32_t t, uint32_t g, uint32_t c, uint32_t rows) {
  return ((g * rows + t) << 6) + ((c ^ km4_perm(t)) << 4);
}
// FP6 (96 bytes per row and K step): byte offset of logical chunk c (0..5) of step s of row t
__device__ __forceinline__ uint32_t km6_off(uint32_t t, uint32_t s, uint32_t c, uint32_t rows) {
  uint32_t p = c + km6_rot(t);
  p = p >= 6 ? p - 6 : p;
  return (s * rows + t) * 96u + (p << 4);
}

// ---------------------------------------------------------------------------------
// Host-side launch helpers
// ---------------------------------------------------------------------------------
inline int grid_for(int64_t work_items_of_block, int64_t cap = kMaxBlocks) {
  int64_t g = work_items_of_block < 1 ? 1 : work_items_of_block;
  return (int)(g > cap ? cap : g);
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? FPQ_OK : FPQ_ERR_LAUNCH; }

}  // namespace
